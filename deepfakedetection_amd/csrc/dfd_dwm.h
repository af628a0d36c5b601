// dfd_dwm.h — depthwise convolution on the matrix cores (bf16): shared geometry and staging of dfd_dwmm*.hip.
//
// Why matrix cores for a depthwise convolution.  The VALU kernels (dfd_dwfwd.hip / dfd_dwbwd.hip) issue, per kernel row of a quad,
// 80 packed FMAs next to 64 bf16 -> f32 unpacks, ~20 address / select operations and 16 LDS reads: the FMAs are half of the issued
// work and the whole tap loop runs at ~8 % of the f32 vector peak.  The same VALU also has to apply the
// producer's BatchNorm + SiLU to every staged element (two quarter-rate transcendentals each), and the two compete.
// Here the tap loop moves to the MFMA pipe, which runs beside the VALU:
//
//     y[p][c] = sum_t a[p + t][c] * w[t][c]      as      D[c][p] = sum_k A[c][k] B[k][p],   k = (tap of a PAIR, 16 channels)
//
// with A[c][(t, c')] = w[t][c] if c' == c else 0 — a 16-channel group's weights on a diagonal, 2 taps per 16x16x32 MFMA.  Only 1/16 of
// the multiplies are useful, which still equals the best the vector unit could do (32 useful MACs per cycle and SIMD) — but it costs
// ONE ds_read_b128 (the B fragment: 16 pixels x 16 channels x 2 taps, bf16 as staged, no unpack) and one address add per 512 useful
// MACs, and leaves the VALU to the prologue.  Zero products are exact, so a channel's result is the f32 sum of its own taps; the one
// observable difference from the VALU form is the order of that sum (pairs inside the MFMA) — and that a non-finite activation makes
// the other 15 channels of its group NaN too (0 * inf), which only happens in a run that has already diverged.
//
// LDS image of a staged tile: one PLANE per 16-channel group, 32 bytes per pixel.  A B fragment's 16 lanes of one LDS pass then read
// 16 consecutive pixels at 32-byte pitch (+16 for the upper channel half), which is conflict-free for ds_read_b128's lane groups
// (MI355X_MICROARCH.md, LDS table); for stride 2 the even and odd input columns are stored as separate half-rows so that the 16
// outputs of a run still read consecutive slots.  Planes are padded to an odd multiple of 32 bytes modulo 128 so the four groups
// a pixel is split into land on different banks when it is staged.
#pragma once
#include "dfd_dwq.h"

#define DWM_MAXPIX 4096              // 20-bit magic divisions are checked up to here

struct DwMGeom {
    int N, H, W, C, Ho, Wo, pt, pl;
    int NI;                          // images per work item (> 1 only when one tile covers the image)
    int TH, TW, THW, NPV, R;         // centre tile, pixels per image, pixels per item, runs of 16
    int IH, IW, IHW;                 // staged extent per image
    int P;                           // row pitch of the LDS image in pixels (stride 2: pitch of a half-row)
    int IMGP;                        // LDS pixels per image
    int plane;                       // LDS pixels per plane, == 1 (mod 4)
    int tiles_y, tiles_x, nwork;
    unsigned tw_magic, thw_magic, iw_magic, ihw_magic, w_magic, hw_magic;
    int whole;                       // one tile = the whole image: the zero border is written once, only real pixels are staged
    int remap;
    int in_bytes, out_bytes, opitch; // LDS bytes of one image / one output tile; bytes per pixel of the output tile
};

// (the planner knobs, DFD_TUNE_* / dfd_tune_get, live in dfd_common.h)

// host entry points of the matrix-core kernels; DFD_EUNSUPPORTED: not served (switched off, f32, C % 16, ...) — the caller runs
// the vector-unit kernel of dfd_dwfwd.hip / dfd_dwbwd.hip
int dfd_dw_fwd_mm(const void* x, const float* in_bnstate, int in_act, const float* w, void* y, const dfd_dwconv_shape* s,
                  float* partials, int pcap, int* nparts, hipStream_t st);

bool dfd_dwm_geom(const dfd_dwconv_shape* s, bool centre_is_input, bool pro, int npair, int extra_lds, DwMGeom* g);

__device__ __forceinline__ int dwm_div(int v, unsigned magic) { return (int)(((unsigned)v * magic) >> 20); }

// LDS pixel index of staged position (iy, ix) of image slot `img`
template <int S>
__device__ __forceinline__ int dwm_lpix(const DwMGeom& g, int img, int iy, int ix) {
    if constexpr (S == 1) return img * g.IMGP + iy * g.P + ix;
    else return img * g.IMGP + (2 * iy + (ix & 1)) * g.P + (ix >> 1);
}
// offset (in LDS pixels) of tap (i, j) relative to the run base of an output pixel
template <int S>
__device__ __forceinline__ int dwm_tapoff(const DwMGeom& g, int i, int j) {
    if constexpr (S == 1) return i * g.P + j;
    else return (2 * i + (j & 1)) * g.P + (j >> 1);
}
// run base of centre pixel (qy, qx) of image slot img: the LDS pixel of tap (0, 0)
template <int S>
__device__ __forceinline__ int dwm_base(const DwMGeom& g, int img, int qy, int qx) {
    if constexpr (S == 1) return img * g.IMGP + qy * g.P + qx;
    else return img * g.IMGP + 4 * qy * g.P + qx;
}

// The diagonal weight fragments of one 16-channel group: lane (m = lane & 15, kg = lane >> 4) of pair pr holds A[m][8 kg .. 8 kg + 7],
// k = (tap 2 pr + (kg >> 1), channel (kg & 1) * 8 + j) — one non-zero bf16 at most.  wsrc(t) = the f32 weight of tap t of channel m.
template <int NPAIR, int KK, typename F>
__device__ __forceinline__ void dwm_weight_frags(uint4 (&wf)[NPAIR], int lane, bool active, F wsrc) {
    const int m = lane & 15, kg = lane >> 4;
    const bool diag = active && ((kg & 1) == (m >> 3));
    const int d = (m & 7) >> 1, sh = 16 * (m & 1);
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const int tt = 2 * pr + (kg >> 1);
        const float wv = (diag && tt < KK) ? wsrc(tt) : 0.f;
        const unsigned h = ((unsigned)f2bf(wv)) << sh;
        wf[pr] = make_uint4(d == 0 ? h : 0u, d == 1 ? h : 0u, d == 2 ? h : 0u, d == 3 ? h : 0u);
    }
}
