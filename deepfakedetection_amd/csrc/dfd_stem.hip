// dfd_stem.hip — the 3-channel stem convolution (3x3, stride 2) and its weight gradient on the bf16 matrix cores.
//
// Every model family starts with it (reference: the first convolution of the timm / lukemelas EfficientNet, of timm's
// EfficientFormerV2 stem and of fastervit's PatchEmbed; reached from trainers/efficientnet.py:266, efficientformer.py:244,
// fastervit.py:271): [N][H][W][3] f32 image -> [N][H/2][W/2][Cout] bf16 with BatchNorm statistics, and dw[Cout][3][3][3].
// The f32-FMA kernels in dfd_misc.hip spend 27 FMAs x Cout per pixel on the vector ALU: 188-352 us forward and 380-547 us
// backward at batch 256 x 224^2, 0.2-0.3 of what the bytes cost.  As GEMMs both are tiny in K / N (27 taps, padded to 32 =
// ONE 16x16x32 MFMA step) so the matrix cores make the arithmetic free and the kernels stream:
//   forward   y[pix][co]  = sum_k w[co][k] patch[pix][k]      A = weights (registers for the whole kernel), B = patches
//   backward  dw[co][k]   = sum_pix dz[pix][co] patch[pix][k] A = dz^T (transposed LDS reads), B = patches, K = pixels
// Input rows are staged in LDS ONCE as bf16 (the rounding the f32 kernels apply per use) in padded coordinates — element
// e = (ix + pad_left) * 3 + ci, zeros left and right — so a patch row of output pixel ox is the 9 contiguous elements from
// 6 ox (4-byte aligned) and image borders need no branches.  Tap order inside the kernels: k = 8 kh + r for r = kw * 3 + ci < 8,
// k = 24 + kh for r = 8 (forward: one 16-byte run per lane) / k = 9 kh + r (backward: per-pixel gathers).
// Same products as the f32 kernels (bf16 operands, f32 accumulation), another summation order.
// Shapes outside (bf16, k 3, stride 2, Cout % 16 == 0, Cout <= 64, W % 4 == 0) keep the dfd_misc.hip kernels.
#include "dfd_common.h"
#include "dfd_pw.h"

typedef __attribute__((address_space(3))) short4_t st_lds_short4;

#define STM_ROWS 4                    // output rows per forward workgroup step (one per wave)
#define STM_BROWS 2                   // output rows per backward workgroup step (two waves per row)

__device__ __forceinline__ unsigned short st_f2bf(float f) { return f2bf(f); }

// stage input rows [iy0, iy0 + nrows) of image n as bf16 into rows of pitch RP elements (data at element 3 * pad_left)
template <int MAXV>
__device__ __forceinline__ void stem_stage_rows(const float* __restrict__ x, unsigned short* __restrict__ xrow, int RP, long n, int iy0,
                                                int nrows, int H, int W, int pad_left) {
    const int vpr = (W * 3) >> 2;                        // float4 per image row
    const int total = nrows * vpr;
    float4 v[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int i = threadIdx.x + u * DFD_THREADS;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total) {
            const int r = i / vpr, c = i - r * vpr, iy = iy0 + r;
            if (iy >= 0 && iy < H) v[u] = *reinterpret_cast<const float4*>(x + ((n * H + iy) * (long)W) * 3 + 4 * c);
        }
    }
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int i = threadIdx.x + u * DFD_THREADS;
        if (i < total) {
            const int r = i / vpr, c = i - r * vpr;
            unsigned short* d = xrow + r * RP + 3 * pad_left + 4 * c;
            d[0] = st_f2bf(v[u].x); d[1] = st_f2bf(v[u].y); d[2] = st_f2bf(v[u].z); d[3] = st_f2bf(v[u].w);
        }
    }
    for (int i = threadIdx.x + MAXV * DFD_THREADS; i < total; i += DFD_THREADS) {     // wider images than MAXV covers
        const int r = i / vpr, c = i - r * vpr, iy = iy0 + r;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H) q = *reinterpret_cast<const float4*>(x + ((n * H + iy) * (long)W) * 3 + 4 * c);
        unsigned short* d = xrow + r * RP + 3 * pad_left + 4 * c;
        d[0] = st_f2bf(q.x); d[1] = st_f2bf(q.y); d[2] = st_f2bf(q.z); d[3] = st_f2bf(q.w);
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <int CT>                                        // Cout / 16
__global__ void __launch_bounds__(DFD_THREADS)
k_stem_fwd_mfma(const float* __restrict__ x, const float* __restrict__ w, unsigned short* __restrict__ y, dfd_stem_shape s, int RP,
                float* __restrict__ partials, int stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
    constexpr int RS = 2 * (STM_ROWS - 1) + 3;           // staged input rows per step
    unsigned short* xrow = smem_s;                        // [RS][RP]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4;
    const int Co = 16 * CT;
    // weights as the A operand: row = output channel, slot (g, j): g < 3: kernel row g, j = kw * 3 + ci; g == 3: j < 3: row j, kw 2, ci 2
    bf16x8_t wf[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const float* wc = w + (long)(16 * t + c16) * 27;
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (g < 3) { const int kw = j / 3, ci = j - 3 * kw; v = wc[(ci * 3 + g) * 3 + kw]; }
            else if (j < 3) v = wc[(2 * 3 + j) * 3 + 2];
            e[j] = st_f2bf(v);
        }
        wf[t] = __builtin_bit_cast(bf16x8_t, (short8_t){(short)e[0], (short)e[1], (short)e[2], (short)e[3], (short)e[4], (short)e[5], (short)e[6], (short)e[7]});
    }
    for (int i = threadIdx.x; i < RS * RP; i += DFD_THREADS) xrow[i] = 0;       // pads stay zero for the whole kernel
    float s1[CT][4], s2[CT][4];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }
    const int groups = (s.Ho + STM_ROWS - 1) / STM_ROWS;  // steps per image
    const long nsteps = (long)s.N * groups;
    const int mtiles = (s.Wo + 15) >> 4;
    for (long step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const long n = step / groups;
        const int oy0 = (int)(step - n * groups) * STM_ROWS;
        __syncthreads();                                  // the previous step's reads are done
        stem_stage_rows<6>(x, xrow, RP, n, oy0 * 2 - s.pad_top, RS, s.H, s.W, s.pad_left);
        __syncthreads();
        const int oy = oy0 + wave;
        if (oy < s.Ho) {
            const unsigned short* r0 = xrow + (2 * wave) * RP;
            unsigned short* yrow = y + ((n * s.Ho + oy) * (long)s.Wo) * Co;
            for (int mt = 0; mt < mtiles; ++mt) {
                const int ox = 16 * mt + c16;
                uint4 q;
                if (g < 3) {
                    const unsigned* p = reinterpret_cast<const unsigned*>(r0 + g * RP + 6 * ox);
                    q = make_uint4(p[0], p[1], p[2], p[3]);
                } else {
                    const unsigned short* p = r0 + 6 * ox + 8;
                    q = make_uint4((unsigned)p[0] | ((unsigned)p[RP] << 16), (unsigned)p[2 * RP], 0u, 0u);
                }
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, q);
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    f32x4_t a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t], pf, (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    if (ox < s.Wo) {
                        const unsigned lo = pack_bf2(a[0], a[1]), hi = pack_bf2(a[2], a[3]);
                        *reinterpret_cast<uint2*>(yrow + (long)ox * Co + 16 * t + 4 * g) = make_uint2(lo, hi);
                        const float v0 = __uint_as_float(lo << 16), v1 = __uint_as_float(lo & 0xffff0000u);
                        const float v2 = __uint_as_float(hi << 16), v3 = __uint_as_float(hi & 0xffff0000u);
                        s1[t][0] += v0; s1[t][1] += v1; s1[t][2] += v2; s1[t][3] += v3;
                        s2[t][0] = fmaf(v0, v0, s2[t][0]); s2[t][1] = fmaf(v1, v1, s2[t][1]);
                        s2[t][2] = fmaf(v2, v2, s2[t][2]); s2[t][3] = fmaf(v3, v3, s2[t][3]);
                    }
                }
            }
        }
    }
    if (stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem_s);    // [4 waves][2][Co]
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = s1[t][r], b = s2[t][r];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
                if (c16 == 0) { red[(wave * 2) * Co + 16 * t + 4 * g + r] = a; red[(wave * 2 + 1) * Co + 16 * t + 4 * g + r] = b; }
            }
        __syncthreads();
        if ((int)threadIdx.x < 2 * Co) {
            float sum = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) sum += red[wv * 2 * Co + threadIdx.x];
            partials[(long)blockIdx.x * 2 * Co + threadIdx.x] = sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------------- weight gradient
template <int CT, bool MAP>
__global__ void __launch_bounds__(DFD_THREADS)
k_stem_wgrad_mfma(const float* __restrict__ x, const unsigned short* __restrict__ dz, const unsigned short* __restrict__ yraw,
                  const float* __restrict__ coef, dfd_stem_shape s, int RP, int WoP, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_s[];
    constexpr int RS = 2 * (STM_BROWS - 1) + 3;
    constexpr int Co = 16 * CT, PITCH = Co + 16;
    unsigned short* xrow = smem_s;                        // [RS][RP]
    unsigned short* dzt = smem_s + ((RS * RP + 7) & ~7);  // [STM_BROWS][WoP][PITCH]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c16 = lane & 15, g = lane >> 4, q = c16 >> 2, p = c16 & 3;
    const int row = wave >> 1, half = wave & 1;
    for (int i = threadIdx.x; i < RS * RP; i += DFD_THREADS) xrow[i] = 0;
    for (int i = threadIdx.x; i < STM_BROWS * WoP * PITCH / 2; i += DFD_THREADS) reinterpret_cast<unsigned*>(dzt)[i] = 0u;   // pixels >= Wo stay zero
    f32x4_t acc[CT][2];
#pragma unroll
    for (int t = 0; t < CT; ++t) { acc[t][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    // the lane's channel vector is fixed (256 % (Co / 8) == 0): BN-backward coefficients loaded once
    constexpr int CV = Co / 8;
    constexpr bool FIXED_V = DFD_THREADS % CV == 0;       // 48 channels: six vectors per pixel, coefficients fetched per item
    float ka[8], kb[8], kc[8];
    if constexpr (MAP && FIXED_V) {
        const int v = threadIdx.x % CV;
        load_f32<8>(coef + v * 8, ka); load_f32<8>(coef + Co + v * 8, kb); load_f32<8>(coef + 2 * Co + v * 8, kc);
    }
    const int groups = (s.Ho + STM_BROWS - 1) / STM_BROWS;
    const long nsteps = (long)s.N * groups;
    const int ksteps = WoP >> 5;
    // tap of this lane's B column in tile nt: k = c16 + 16 nt = 9 kh + r (k < 27)
    int boff[2];
    bool bon[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int k = c16 + 16 * nt, kh = k / 9, r = k - 9 * kh;
        bon[nt] = k < 27;
        boff[nt] = bon[nt] ? (2 * row + kh) * RP + r : 0;
    }
    for (long step = blockIdx.x; step < nsteps; step += gridDim.x) {
        const long n = step / groups;
        const int oy0 = (int)(step - n * groups) * STM_BROWS;
        __syncthreads();
        // ---- dz rows (BN-backward map, rounded to bf16) -> [row][pixel][PITCH]; loads first, then the map and the stores
        constexpr int DV = 8;
        const int dtotal = STM_BROWS * s.Wo * CV;
        uint4 d0[DV], d1[MAP ? DV : 1];
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS;
            d0[u] = make_uint4(0, 0, 0, 0);
            if constexpr (MAP) d1[u] = make_uint4(0, 0, 0, 0);
            if (i < dtotal) {
                const int pix = i / CV, v = i - pix * CV, r = pix / s.Wo, ox = pix - r * s.Wo;
                if (oy0 + r < s.Ho) {
                    const long off = ((n * s.Ho + oy0 + r) * (long)s.Wo + ox) * Co + v * 8;
                    d0[u] = *reinterpret_cast<const uint4*>(dz + off);
                    if constexpr (MAP) d1[u] = *reinterpret_cast<const uint4*>(yraw + off);
                }
            }
        }
        stem_stage_rows<4>(x, xrow, RP, n, oy0 * 2 - s.pad_top, RS, s.H, s.W, s.pad_left);
        auto put = [&](int i, uint4 a, uint4 b) {
            const int pix = i / CV, v = i - pix * CV, r = pix / s.Wo, ox = pix - r * s.Wo;
            if constexpr (MAP) {
                if (oy0 + r < s.Ho) {
                    float f[8], f2[8];
                    Vec<bf16>::unpack(a, f);
                    Vec<bf16>::unpack(b, f2);
                    if constexpr (!FIXED_V) { load_f32<8>(coef + v * 8, ka); load_f32<8>(coef + Co + v * 8, kb); load_f32<8>(coef + 2 * Co + v * 8, kc); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = fmaf(ka[j], f[j], fmaf(kb[j], f2[j], kc[j]));
                    a = Vec<bf16>::pack(f);
                }
            }
            *reinterpret_cast<uint4*>(dzt + (r * WoP + ox) * PITCH + v * 8) = a;
        };
#pragma unroll
        for (int u = 0; u < DV; ++u) {
            const int i = threadIdx.x + u * DFD_THREADS;
            uint4 b2 = d0[u];
            if constexpr (MAP) b2 = d1[u];
            if (i < dtotal) put(i, d0[u], b2);
        }
        for (int i = threadIdx.x + DV * DFD_THREADS; i < dtotal; i += DFD_THREADS) {       // wider images than DV covers
            const int pix = i / CV, v = i - pix * CV, r = pix / s.Wo, ox = pix - r * s.Wo;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (oy0 + r < s.Ho) {
                const long off = ((n * s.Ho + oy0 + r) * (long)s.Wo + ox) * Co + v * 8;
                a = *reinterpret_cast<const uint4*>(dz + off);
                if constexpr (MAP) b = *reinterpret_cast<const uint4*>(yraw + off);
            }
            put(i, a, b);
        }
        __syncthreads();
        // ---- k-step = 32 pixels of the wave's row; slot (g, j) = pixel 32 ks + 16 (j >> 2) + 4 g + (j & 3)
        const unsigned short* drow = dzt + row * WoP * PITCH;
        for (int ks = half; ks < ksteps; ks += 2) {
            bf16x8_t bfr[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                unsigned short e[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int px = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
                    e[j] = xrow[boff[nt] + 6 * px];
                }
                uint4 qq = make_uint4((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                                      (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16));
                if (!bon[nt]) qq = make_uint4(0, 0, 0, 0);
                bfr[nt] = __builtin_bit_cast(bf16x8_t, qq);
            }
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const unsigned short* a = drow + (32 * ks + 4 * g + q) * PITCH + 16 * t + 4 * p;
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_short4*)(a));
                const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((st_lds_short4*)(a + 16 * PITCH));
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[0], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[1], acc[t][1], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' partial sums through LDS (fixed order), written in torch's tap order (ci, kh, kw)
    __syncthreads();
    float* red = reinterpret_cast<float*>(dzt);           // [4][Co][32]
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * Co + 16 * t + 4 * g + r) * 32 + 16 * nt + c16] = acc[t][nt][r];
    __syncthreads();
    float* o = ws + (long)blockIdx.x * Co * 27;
    for (int i = threadIdx.x; i < Co * 27; i += DFD_THREADS) {
        const int co = i / 27, k = i - co * 27, kh = k / 9, r = k - 9 * kh, kw = r / 3, ci = r - 3 * kw;
        float sum = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) sum += red[(wv * Co + co) * 32 + k];
        o[co * 27 + (ci * 3 + kh) * 3 + kw] = sum;
    }
}

// ----------------------------------------------------------------------------------------------------------------- host
static bool stem_mfma_ok(const dfd_stem_shape* s, const void* x) {
    return s->k == 3 && s->stride == 2 && s->Cout % 16 == 0 && s->Cout <= 64 && s->W % 4 == 0 && ((size_t)x & 15) == 0 &&
           s->W <= 1024 && s->Wo <= 512 && s->pad_left <= 2 && s->pad_top <= 2;
}
static int stem_row_pitch(const dfd_stem_shape* s, int pix_round) {
    const int wo = ((s->Wo + pix_round - 1) / pix_round) * pix_round;
    int need = 6 * wo + 16;
    const int data = (s->W + s->pad_left) * 3 + 8;
    if (need < data) need = data;
    return (need + 7) & ~7;
}
int dfd_stem_fwd_mfma(const float* x, const float* w, void* y, const dfd_stem_shape* s, float* partials, int pcap, int* nparts,
                      hipStream_t st) {
    if (!stem_mfma_ok(s, x)) return DFD_EUNSUPPORTED;
    const int RP = stem_row_pitch(s, 16);
    constexpr int RS = 2 * (STM_ROWS - 1) + 3;
    size_t lds = (size_t)RS * RP * 2;
    if (lds < (size_t)4 * 2 * s->Cout * 4) lds = (size_t)4 * 2 * s->Cout * 4;
    if (lds > 64 * 1024) return DFD_EUNSUPPORTED;
    const long nsteps = (long)s->N * ((s->Ho + STM_ROWS - 1) / STM_ROWS);
    long P = 1024;
    if (partials && P > pcap) P = pcap;
    if (P > DFD_MAX_PARTIALS) P = DFD_MAX_PARTIALS;
    if (P > nsteps) P = nsteps;
    if (partials) *nparts = (int)P;
#define STF(CT) hipLaunchKernelGGL((k_stem_fwd_mfma<CT>), dim3((unsigned)P), dim3(DFD_THREADS), lds, st, x, w, (unsigned short*)y, *s, RP, \
                                   partials, partials ? 1 : 0)
    switch (s->Cout / 16) { case 1: STF(1); break; case 2: STF(2); break; case 3: STF(3); break; default: STF(4); break; }
#undef STF
    return DFD_CHECK_LAUNCH();
}
// launches only the partial-slab kernel; *rows = slab rows written ([rows][Cout][27]); the caller sums them
int dfd_stem_wgrad_mfma(const float* x, const void* dz, const void* yraw, const float* coef, const dfd_stem_shape* s, float* ws,
                        int max_rows, int* rows, hipStream_t st) {
    if (!stem_mfma_ok(s, x)) return DFD_EUNSUPPORTED;
    const int RP = stem_row_pitch(s, 32), WoP = ((s->Wo + 31) / 32) * 32;
    constexpr int RS = 2 * (STM_BROWS - 1) + 3;
    const int Co = s->Cout;
    size_t dzt = (size_t)STM_BROWS * WoP * (Co + 16) * 2;
    if (dzt < (size_t)4 * Co * 32 * 4) dzt = (size_t)4 * Co * 32 * 4;
    const size_t lds = (((size_t)RS * RP + 7) & ~(size_t)7) * 2 + dzt;
    if (lds > 64 * 1024) return DFD_EUNSUPPORTED;
    const long nsteps = (long)s->N * ((s->Ho + STM_BROWS - 1) / STM_BROWS);
    long P = 768;
    if (P > max_rows) P = max_rows;
    if (P > nsteps) P = nsteps;
    *rows = (int)P;
#define STW(CT)                                                                                                                     \
    do {                                                                                                                            \
        if (coef) hipLaunchKernelGGL((k_stem_wgrad_mfma<CT, true>), dim3((unsigned)P), dim3(DFD_THREADS), lds, st, x,               \
                                     (const unsigned short*)dz, (const unsigned short*)yraw, coef, *s, RP, WoP, ws);                \
        else hipLaunchKernelGGL((k_stem_wgrad_mfma<CT, false>), dim3((unsigned)P), dim3(DFD_THREADS), lds, st, x,                   \
                                (const unsigned short*)dz, (const unsigned short*)yraw, coef, *s, RP, WoP, ws);                     \
    } while (0)
    switch (Co / 16) { case 1: STW(1); break; case 2: STW(2); break; case 3: STW(3); break; default: STW(4); break; }
#undef STW
    return DFD_CHECK_LAUNCH();
}
