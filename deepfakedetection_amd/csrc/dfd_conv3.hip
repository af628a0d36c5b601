// dfd_conv3.hip — dense 3x3 stride-1 convolution (pad 1) as a DIRECT MFMA kernel, bf16.
//
// FasterViT's ConvBlocks (levels 0-1: 56x56x64 and 28x28x128 at 224 px) run two such convolutions forward and — the data
// gradient of a stride-1 convolution being the forward convolution of the output gradient with the flipped, transposed weight
// — two more backward (reference call sites: the third-party module's forward / backward at trainers/fastervit.py:271, :274).
// The implicit-GEMM form (k_pw_nt<.., CONV>) gathers its operand chunk by chunk with two magic divisions per 16 bytes and
// applies the producer's BatchNorm + GELU to every gathered copy (nine times per element): 133-516 us per call, 5.2 of
// FasterViT-0's 30.6 ms, at ~450 TFLOP/s and 1.6 TB/s — neither roofline.  Here:
//   * a workgroup owns an 8 x 16 tile of output pixels and 64 output channels (grid.y = Cout / 64); the input tile with halo
//     (10 x 18 pixels x C channels) is staged in LDS ONCE, the producer's BN + activation applied once per element, zero padding
//     in the activated domain; pixel pitch C*2 + 32 bytes = p sixteen-byte units with p = 2 (mod 4): ds_read_b128 is served
//     in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, .. (MI355X_MICROARCH.md, LDS), i.e. pixels {0-3, 12-15} of
//     one k-group with pixels {4-11} of the next: an odd p always collides some pair (C*2 + 16: 7 two-way conflicts per
//     group, every read at half rate), p = 2 (mod 4) keeps the two k-groups on opposite parities and each of them distinct;
//   * wave w owns output channels 16 w .. 16 w + 15 of the group: its WEIGHTS — 9 taps x C/32 k-steps of 16x32 fragments —
//     stay in registers for the whole persistent loop (72 VGPRs at C = 64, 144 at C = 128): no weight traffic after the prologue;
//   * per (tap, k-step) and 16-pixel row of the tile: one ds_read_b128 (the pixel fragment, shifted by the tap) + one
//     v_mfma_f32_16x16x32_bf16 with the weight as the A operand, so a lane ends up with 4 consecutive output channels of one
//     pixel: 8-byte stores;
//   * BatchNorm partial sums (sum, sum of squares of the ROUNDED outputs) per channel: registers over the persistent loop, a
//     16-lane butterfly at the end, one partial row per workgroup slot (waves own disjoint channels: no cross-wave step).
// Bound: LDS read rate (1 KB per MFMA) at ~2x the MFMA time — ~55-60 us per convolution at batch 256 against 100+ MB of
// activations (input + output once): HBM ~4 TB/s equivalent.  Shapes outside (k 3, stride 1, pad 1, C in {64, 96, 128}, Cout % 64
// == 0, bf16) keep the implicit-GEMM path (DFD_EUNSUPPORTED from the launcher, checked by dfd_conv_fwd).
#include "dfd_common.h"
#include "dfd_pw.h"

#define C3_TW 16
#define C3_IW (C3_TW + 2)
#define C3_OP 72                 // elements per pixel of the staged output tile (64 channels + 8)

template <int CK, int ACT, bool PRO, bool STATS, int TH>        // CK = C / 32; TH = tile rows (8, or 7 where 7 divides H and 8 does not)
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_conv3_direct(const unsigned short* __restrict__ x, const float* __restrict__ in_bn, const unsigned short* __restrict__ w,
               unsigned short* __restrict__ y, float* __restrict__ partials, int N, int H, int W, int Cout, int tiles_y, int tiles_x) {
    constexpr int C = 32 * CK, CP = C + 16;             // pixel pitch in elements: C / 8 + 2 sixteen-byte units, = 2 mod 4 (see below)
    extern __shared__ __attribute__((aligned(16))) unsigned short tile[];         // [(TH + 2)][C3_IW][CP], then the output tile
    unsigned short* otile = tile + (TH + 2) * C3_IW * CP;                            // [TH * C3_TW][C3_OP]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int co0 = blockIdx.y * 64 + wave * 16;        // this wave's 16 output channels
    // resident weights: A operand fragment (row = output channel co0 + c16, k = input channels 32 ks + 8 g ..) per tap
    bf16x8_t wf[9][CK];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < CK; ++ks)
            wf[t][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(w + (long)(co0 + c16) * 9 * C + t * C + 32 * ks + 8 * g));
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int tiles = tiles_y * tiles_x;
    const long nwork = (long)N * tiles;
    constexpr int VPP = C / 8;                          // 16-byte vectors per pixel
    for (long work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int n = (int)(work / tiles), tr = (int)(work - (long)n * tiles);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int h0 = ty * TH, w0 = tx * C3_TW;
        __syncthreads();                                 // previous tile's fragment reads are done
        // ---- stage the input tile with halo; the producer's BN + activation once per element; zeros outside the image
        // the tile's loads are issued in one batch (64 channels: 6 vectors per lane) before their first use, or in three of 3-4
        // (96 / 128 channels: beside 108-144 registers of resident weights a batch of 6 spilled and measured 14-27 % slower)
        {
            constexpr int NV = ((TH + 2) * C3_IW * VPP + DFD_THREADS - 1) / DFD_THREADS;
            constexpr int NBATCH = CK >= 3 ? 3 : 1, NVB = (NV + NBATCH - 1) / NBATCH;
#pragma unroll 1
            for (int bt = 0; bt < NBATCH; ++bt) {       // a real loop: unrolled, the compiler hoists every batch's loads to the top
                uint4 r[NVB];
                bool ok[NVB];
#pragma unroll
                for (int u = 0; u < NVB; ++u) {
                    const int i = threadIdx.x + (bt * NVB + u) * DFD_THREADS;
                    const int pix = i / VPP, v = i - pix * VPP;
                    const int iy = pix / C3_IW, ix = pix - iy * C3_IW;
                    const int gy = h0 - 1 + iy, gx = w0 - 1 + ix;
                    ok[u] = i < (TH + 2) * C3_IW * VPP && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    r[u] = make_uint4(0, 0, 0, 0);
                    if (ok[u]) r[u] = *reinterpret_cast<const uint4*>(x + (((long)n * H + gy) * W + gx) * C + v * 8);
                }
#pragma unroll
                for (int u = 0; u < NVB; ++u) {
                    const int i = threadIdx.x + (bt * NVB + u) * DFD_THREADS;
                    if (i >= (TH + 2) * C3_IW * VPP) continue;
                    const int pix = i / VPP, v = i - pix * VPP;
                    uint4 q = r[u];
                    if constexpr (PRO) {
                        if (ok[u]) {
                            float f[8], sc[8], sh[8];
                            Vec<bf16>::unpack(q, f);
                            load_f32<8>(in_bn + v * 8, sc);
                            load_f32<8>(in_bn + C + v * 8, sh);
                            bn_act_array<ACT, 8>(f, sc, sh);
                            q = Vec<bf16>::pack(f);
                        }
                    }
                    *reinterpret_cast<uint4*>(tile + pix * CP + v * 8) = q;
                }
            }
        }
        __syncthreads();
        // ---- 8 rows of 16 pixels x this wave's 16 channels
        f32x4_t acc[TH];
#pragma unroll
        for (int r = 0; r < TH; ++r) acc[r] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // blocks of (tap, k-step): the next block's eight pixel fragments (B operand: column = pixel c16 of a tile row, k =
        // channels 32 ks + 8 g ..) are read while the current block's MFMAs issue; the scheduling barrier keeps the compiler from
        // hoisting every read of the tile to the top (that spills the resident weights)
        constexpr int NB = 9 * CK;
        bf16x8_t pf[2][TH];
        const unsigned short* lane_base = tile + c16 * CP + 8 * g;
#pragma unroll
        for (int r = 0; r < TH; ++r) pf[0][r] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lane_base + r * C3_IW * CP));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b + 1 < NB) {
                const int t = (b + 1) / CK, ks = (b + 1) - t * CK, kh = t / 3, kw = t - 3 * kh;
                const unsigned short* base = lane_base + (kh * C3_IW + kw) * CP + 32 * ks;
#pragma unroll
                for (int r = 0; r < TH; ++r) pf[(b + 1) & 1][r] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(base + r * C3_IW * CP));
            }
            const int t = b / CK, ks = b - t * CK;
#pragma unroll
            for (int r = 0; r < TH; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], pf[b & 1][r], acc[r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- store: lane holds output channels co0 + 4 g .. + 3 of pixel (h0 + r, w0 + c16).  The four waves' 32-byte quarters
        // of a pixel's 128-byte line meet in LDS and leave as 16-byte pieces of whole lines (8-byte stores straight from the
        // accumulators wrote quarter lines at four different times: measured slower, here and in the GEMM kernel)
        const int ox = w0 + c16;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const unsigned lo = pack_bf2(acc[r][0], acc[r][1]), hi = pack_bf2(acc[r][2], acc[r][3]);
            *reinterpret_cast<uint2*>(otile + (r * C3_TW + c16) * C3_OP + wave * 16 + 4 * g) = make_uint2(lo, hi);
            if constexpr (STATS) {
                if (h0 + r < H && ox < W) {
                    const float v0 = __uint_as_float(lo << 16), v1 = __uint_as_float(lo & 0xffff0000u);
                    const float v2 = __uint_as_float(hi << 16), v3 = __uint_as_float(hi & 0xffff0000u);
                    s1[0] += v0; s1[1] += v1; s1[2] += v2; s1[3] += v3;
                    s2[0] = fmaf(v0, v0, s2[0]); s2[1] = fmaf(v1, v1, s2[1]); s2[2] = fmaf(v2, v2, s2[2]); s2[3] = fmaf(v3, v3, s2[3]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < (TH * C3_TW * 8 + DFD_THREADS - 1) / DFD_THREADS; ++u) {
            const int idx = threadIdx.x + u * DFD_THREADS, pix = idx >> 3, ch = idx & 7;
            const int oy = h0 + (pix >> 4), oxx = w0 + (pix & 15);
            if (pix < TH * C3_TW && oy < H && oxx < W)
                *reinterpret_cast<uint4*>(y + (((long)n * H + oy) * W + oxx) * Cout + blockIdx.y * 64 + ch * 8) =
                    *reinterpret_cast<const uint4*>(otile + pix * C3_OP + ch * 8);
        }
    }
    if constexpr (STATS) {
        // the 16 lanes of a group g hold the same 4 channels for 16 different pixels: butterfly over the low 4 lane bits
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 8; o >= 1; o >>= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
        if (c16 == 0) {
            float* p = partials + (long)blockIdx.x * 2 * Cout;
#pragma unroll
            for (int j = 0; j < 4; ++j) { p[co0 + 4 * g + j] = s1[j]; p[Cout + co0 + 4 * g + j] = s2[j]; }
        }
    }
}

// DFD_EUNSUPPORTED: the shape is not this kernel's (the caller falls back to the implicit GEMM)
int dfd_conv3_direct(const void* x, const dfd_dwconv_shape* s, const float* in_bnstate, int in_act, const void* w_nk, int Cout,
                     void* y, float* partials, int pcap, int* nparts, hipStream_t st) {
    if (s->k != 3 || s->stride != 1 || s->pad_top != 1 || s->pad_left != 1 || s->Ho != s->H || s->Wo != s->W) return DFD_EUNSUPPORTED;
    if (Cout % 64 || !(s->C == 64 || s->C == 96 || s->C == 128)) return DFD_EUNSUPPORTED;
    const int th = (s->H % 8 != 0 && s->H % 7 == 0) ? 7 : 8;      // 28 x 28 (and 14, 7): 7-row tiles cover the image exactly
    const int tiles_y = (s->H + th - 1) / th, tiles_x = (s->W + C3_TW - 1) / C3_TW;
    const long nwork = (long)s->N * tiles_y * tiles_x;
    const bool stats = partials != nullptr;
    // two workgroups per CU, persistent: 512 over ALL 64-channel column blocks.  (Until round 4 it was 512 PER block: with Cout = 128 half of
    // the workgroups queued behind the other half and every one loaded its 147 KB of weights for half the work; FasterViT-0 19.74 -> 19.53 ms.)
    long gx = 512 / (Cout / 64);
    if (stats) { const long cap = pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS; if (gx > cap) gx = cap; }
    if (gx > nwork) gx = nwork;
    if (stats) *nparts = (int)gx;
    const dim3 grid((unsigned)gx, Cout / 64);
    const size_t lds = (size_t)(th + 2) * C3_IW * (s->C + 16) * 2 + (size_t)th * C3_TW * C3_OP * 2;
#define C3_LAUNCH(CK, PRO, STATS)                                                                                          \
    do {                                                                                                                   \
        if (th == 7) hipLaunchKernelGGL((k_conv3_direct<CK, ACT, PRO, STATS, 7>), grid, dim3(DFD_THREADS), lds, st, (const unsigned short*)x, \
                       in_bnstate, (const unsigned short*)w_nk, (unsigned short*)y, partials, s->N, s->H, s->W, Cout, tiles_y, tiles_x); \
        else hipLaunchKernelGGL((k_conv3_direct<CK, ACT, PRO, STATS, 8>), grid, dim3(DFD_THREADS), lds, st, (const unsigned short*)x, \
                       in_bnstate, (const unsigned short*)w_nk, (unsigned short*)y, partials, s->N, s->H, s->W, Cout, tiles_y, tiles_x); \
    } while (0)
#define C3_CK(CK)                                                                                                          \
    do {                                                                                                                   \
        if (!in_bnstate) {                                                                                                 \
            constexpr int ACT = DFD_ACT_NONE;                                                                              \
            if (stats) C3_LAUNCH(CK, false, true); else C3_LAUNCH(CK, false, false);                                       \
        } else {                                                                                                           \
            DISPATCH_ACT_PW(in_act, { if (stats) C3_LAUNCH(CK, true, true); else C3_LAUNCH(CK, true, false); });           \
        }                                                                                                                  \
    } while (0)
    if (s->C == 64) C3_CK(2);
    else if (s->C == 96) C3_CK(3);
    else C3_CK(4);
#undef C3_CK
#undef C3_LAUNCH
    return DFD_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradient of the same layers: dw[co][tap][ci] = sum over output pixels of P(dy)[pix][co] * act(x)[pix + tap][ci]
// (P = the BatchNorm-backward map c0 dy + c1 y + c2 of the layer's own BN, or nothing).  The implicit TN GEMM
// (k_pw_tn<.., CONV>) re-gathers and re-activates x for each of the 9 taps and runs at ~150 TFLOP/s (212-390 us per layer).
// Here the OUTPUT stays in registers instead: a workgroup owns 64 output channels x 64 input channels x 9 taps (36 16x16
// accumulator tiles per wave: wave w owns input-channel tile w, all four output-channel tiles and all taps) for the whole
// persistent loop over 4 x 32 pixel tiles; per tile the activated x (with halo) and the mapped dy are staged in LDS once and
// both MFMA operands — k = 32 pixels of one tile row — come from transposed LDS reads (ds_read_tr16_b64: 4 pixels x 16
// channels per 16 lanes), the tap being nothing but an address offset.  Per 32-pixel step: 8 + 18 transposed reads for 36
// MFMAs.  Each workgroup writes its 64 x 9 x 64 partial block at the end; dfd_sum_partials adds the rows in a fixed order.
typedef __attribute__((address_space(3))) short4_t c3_lds_short4;
#define C3W_TH 4
#define C3W_TW 32
#define C3W_IH (C3W_TH + 2)
#define C3W_IW (C3W_TW + 2)
#define C3W_PITCH 80             // elements per staged pixel (64 channels + 16: pixels 40 banks apart, conflict-free 8-byte reads)

template <int ACT, bool PROX, bool PROP>
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_conv3_wgrad(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ y2, const float* __restrict__ coefp,
              const unsigned short* __restrict__ x, const float* __restrict__ in_bn, float* __restrict__ slab,
              int N, int H, int W, int C, int Cout, int tiles_y, int tiles_x) {
    __shared__ __attribute__((aligned(16))) unsigned short xt[C3W_IH * C3W_IW * C3W_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short dt[C3W_TH * C3W_TW * C3W_PITCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4, q = c16 >> 2, p = c16 & 3;
    const int cgroups = C >> 6;
    const int cig = blockIdx.y % cgroups, cog = blockIdx.y / cgroups;
    const int ci0 = cig * 64, co0 = cog * 64;
    f32x4_t acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[t][o] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int tiles = tiles_y * tiles_x;
    const long nwork = (long)N * tiles;
    for (long work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int n = (int)(work / tiles), tr = (int)(work - (long)n * tiles);
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int h0 = ty * C3W_TH, w0 = tx * C3W_TW;
        __syncthreads();
        // ---- every global load of the tile is issued before the first use (one exposed memory latency per tile, not eleven)
        constexpr int XV = (C3W_IH * C3W_IW * 8 + DFD_THREADS - 1) / DFD_THREADS, DV = C3W_TH * C3W_TW * 8 / DFD_THREADS;
        uint4 xr[XV], dr[DV], d2[PROP ? DV : 1];
        bool xok[XV], dok[DV];
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS, pix = i >> 3, v = i & 7;
            const int iy = pix / C3W_IW, ix = pix - iy * C3W_IW;
            const int gy = h0 - 1 + iy, gx = w0 - 1 + ix;
            xok[u] = i < C3W_IH * C3W_IW * 8 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            xr[u] = make_uint4(0, 0, 0, 0);
            if (xok[u]) xr[u] = *reinterpret_cast<const uint4*>(x + (((long)n * H + gy) * W + gx) * C + ci0 + v * 8);
        }
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS, pix = i >> 3, v = i & 7;
            const int iy = pix / C3W_TW, ix = pix - iy * C3W_TW;
            const int gy = h0 + iy, gx = w0 + ix;
            dok[u] = gy < H && gx < W;
            dr[u] = make_uint4(0, 0, 0, 0);
            if (dok[u]) dr[u] = *reinterpret_cast<const uint4*>(dy + (((long)n * H + gy) * W + gx) * Cout + co0 + v * 8);
        }
        // ---- x tile with halo (this workgroup's 64 input channels), producer's BN + activation once per element
#pragma unroll
        for (int u = 0; u < XV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS, pix = i >> 3, v = i & 7;
            if (i >= C3W_IH * C3W_IW * 8) continue;
            uint4 r = xr[u];
            if constexpr (PROX) {
                if (xok[u]) {
                    float f[8], sc[8], sh[8];
                    Vec<bf16>::unpack(r, f);
                    load_f32<8>(in_bn + ci0 + v * 8, sc);
                    load_f32<8>(in_bn + C + ci0 + v * 8, sh);
                    bn_act_array<ACT, 8>(f, sc, sh);
                    r = Vec<bf16>::pack(f);
                }
            }
            *reinterpret_cast<uint4*>(xt + pix * C3W_PITCH + v * 8) = r;
        }
        // ---- dy tile (this workgroup's 64 output channels), BatchNorm-backward map once per element; zero outside the image
        if constexpr (PROP) {                            // the map's second operand: loaded once the x registers are free again
#pragma unroll
            for (int u = 0; u < DV; ++u) {
                const int i = threadIdx.x + u * DFD_THREADS, pix = i >> 3, v = i & 7;
                const int iy = pix / C3W_TW, ix = pix - iy * C3W_TW;
                d2[u] = make_uint4(0, 0, 0, 0);
                if (dok[u]) d2[u] = *reinterpret_cast<const uint4*>(y2 + (((long)n * H + h0 + iy) * W + w0 + ix) * Cout + co0 + v * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS, pix = i >> 3, v = i & 7;
            uint4 r = dr[u];
            if constexpr (PROP) {
                if (dok[u]) {
                    float f[8], f2[8], k0[8], k1[8], k2[8];
                    Vec<bf16>::unpack(r, f);
                    Vec<bf16>::unpack(d2[u], f2);
                    load_f32<8>(coefp + co0 + v * 8, k0);
                    load_f32<8>(coefp + Cout + co0 + v * 8, k1);
                    load_f32<8>(coefp + 2 * Cout + co0 + v * 8, k2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaf(k0[j], f[j], fmaf(k1[j], f2[j], k2[j]));
                    r = Vec<bf16>::pack(f);
                }
            }
            *reinterpret_cast<uint4*>(dt + pix * C3W_PITCH + v * 8) = r;
        }
        __syncthreads();
        // ---- k-step = one tile row of 32 pixels; slot (g, j) = pixel column 16 (j >> 2) + 4 g + (j & 3)
        // transposed read: lane 4 q + p of group g supplies the address of pixel 4 g + q, channels 4 p .. 4 p + 3
        const unsigned short* da = dt + (4 * g + q) * C3W_PITCH + 4 * p;
        const unsigned short* xa = xt + (4 * g + q) * C3W_PITCH + 16 * wave + 4 * p;
#pragma unroll
        for (int ks = 0; ks < C3W_TH; ++ks) {
            bf16x8_t af[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const unsigned short* a = da + ks * C3W_TW * C3W_PITCH + 16 * o;
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_short4*)(a));
                const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_short4*)(a + 16 * C3W_PITCH));
                af[o] = __builtin_bit_cast(bf16x8_t, (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kh = t / 3, kw = t - 3 * kh;
                const unsigned short* b = xa + ((ks + kh) * C3W_IW + kw) * C3W_PITCH;
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_short4*)(b));
                const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((c3_lds_short4*)(b + 16 * C3W_PITCH));
                const bf16x8_t bfrag = __builtin_bit_cast(bf16x8_t, (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[t][o] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[o], bfrag, acc[t][o], 0, 0, 0);
            }
        }
    }
    // ---- partial block: lane holds dw[co0 + 16 o + 4 g + r][tap][ci0 + 16 wave + c16]
    float* out = slab + (long)blockIdx.x * Cout * 9 * C;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(long)(co0 + 16 * o + 4 * g + r) * 9 * C + t * C + ci0 + 16 * wave + c16] = acc[t][o][r];
}

static bool conv3_wgrad_shape_ok(const dfd_dwconv_shape* s, int Cout) {
    return s->k == 3 && s->stride == 1 && s->pad_top == 1 && s->pad_left == 1 && s->Ho == s->H && s->Wo == s->W && s->C % 64 == 0 &&
           Cout % 64 == 0 && s->C <= 512 && Cout <= 512;
}
static int conv3_wgrad_rows(const dfd_dwconv_shape* s, int Cout) {
    const int gy = (s->C / 64) * (Cout / 64);
    const long nwork = (long)s->N * ((s->H + C3W_TH - 1) / C3W_TH) * ((s->W + C3W_TW - 1) / C3W_TW);
    long gx = 512 / gy;
    if (gx < 1) gx = 1;
    if (gx > nwork) gx = nwork;
    return (int)gx;
}
// workspace floats' bytes the direct kernel needs (0: shape not served)
size_t dfd_conv3_wgrad_ws(const dfd_dwconv_shape* s, int Cout) {
    if (!conv3_wgrad_shape_ok(s, Cout)) return 0;
    const int gx = conv3_wgrad_rows(s, Cout);
    return (size_t)(gx + gx / 32 + 2) * Cout * 9 * s->C * 4;
}
// DFD_EUNSUPPORTED: the shape / prologue is not this kernel's (the caller falls back to the implicit TN GEMM)
int dfd_conv3_wgrad(const void* p, const dfd_prologue* pro_p, int Cout, const void* x, const dfd_dwconv_shape* s,
                    const float* in_bnstate, int in_act, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    if (!conv3_wgrad_shape_ok(s, Cout)) return DFD_EUNSUPPORTED;
    const int mp = pro_p ? pro_p->mode : DFD_PRO_NONE;
    if (!(mp == DFD_PRO_NONE || mp == DFD_PRO_AFFINE2)) return DFD_EUNSUPPORTED;
    if (dfd_conv3_wgrad_ws(s, Cout) > ws_bytes) return DFD_EWORKSPACE;
    const int gx = conv3_wgrad_rows(s, Cout);
    const int tiles_y = (s->H + C3W_TH - 1) / C3W_TH, tiles_x = (s->W + C3W_TW - 1) / C3W_TW;
    const dim3 grid(gx, (s->C / 64) * (Cout / 64));
    const unsigned short* y2 = mp == DFD_PRO_AFFINE2 ? (const unsigned short*)pro_p->a2 : nullptr;
    const float* coefp = mp == DFD_PRO_AFFINE2 ? pro_p->coef : nullptr;
#define C3W_LAUNCH(PROX, PROP)                                                                                              \
    hipLaunchKernelGGL((k_conv3_wgrad<ACT, PROX, PROP>), grid, dim3(DFD_THREADS), 0, st, (const unsigned short*)p, y2, coefp, \
                       (const unsigned short*)x, in_bnstate, ws, s->N, s->H, s->W, s->C, Cout, tiles_y, tiles_x)
    if (!in_bnstate) {
        constexpr int ACT = DFD_ACT_NONE;
        if (mp == DFD_PRO_AFFINE2) C3W_LAUNCH(false, true); else C3W_LAUNCH(false, false);
    } else {
        DISPATCH_ACT_PW(in_act, { if (mp == DFD_PRO_AFFINE2) C3W_LAUNCH(true, true); else C3W_LAUNCH(true, false); });
    }
#undef C3W_LAUNCH
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    // dw is in GEMM layout: the caller's dfd_conv_weight_perm reads it next, so the sum is not left to an open batch
    return dfd_launch_sum_partials(ws, gx, (long)Cout * 9 * s->C, dw, accumulate, st, false);
}
