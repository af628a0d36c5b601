// dfd_dwconv.hip — depthwise k x k convolution (k in {3,5}, stride in {1,2}) over
// NHWC activations for gfx950: data gradient and weight gradient (forward: dfd_dwfwd.hip).
//
// All three kernels share one structure:
//   * a workgroup owns a power-of-two chunk of CVB channel vectors (16 B each) and
//     walks a list of (image, spatial tile) work items (persistent grid), so its
//     per-channel statistics stay in registers until the end;
//   * the producer's BatchNorm + activation (forward) or the BN-backward affine map
//     dy = a*dz + b*y + c (backward) is applied ONCE per element while the tile with
//     its halo is staged into LDS; padding is written as zeros in the staged domain;
//   * lanes are (channel vector fastest, pixel next): a wave reads/writes contiguous
//     CVB*16-byte segments of consecutive pixels, and LDS reads are 16 B per lane.
// HBM traffic per launch is ~1 read of the input tensor (+halo) and 1 write of the
// output tensor; see DESIGN.md for the byte counts used in the roofline.
#include "dfd_common.h"

struct DwGeom {
    int N, H, W, C, Ho, Wo, pt, pl;
    int CV;            // channel vectors in C
    int cvb_log2;      // log2(channel vectors per workgroup)
    int TH, tw_log2;   // tile of "centre" pixels: TH x (1 << tw_log2)
    int tiles_y, tiles_x, nwork;
    int IH, IW;        // staged tile extent
    unsigned iw_magic; // ceil(2^20 / IW): pix / IW == (pix * iw_magic) >> 20 for pix < 4096
    int CH, CW;        // extent of the centre-pixel grid the tiles cover (Ho,Wo or H,W)
};

#define DW_NP 4  // centre pixels per lane per tile

// ---------------------------------------------------------------------------
// staging: fill tile[pix][vl] for pix in [0, IH*IW) from image n of src at rows
// gy0.., cols gx0.. (SH x SW image), applying MODE:
//   0 raw, 1 act(scale*x+shift), 2 a*x + b*x2 + c
// ---------------------------------------------------------------------------
template <typename T, int ACT, int MODE>
__device__ __forceinline__ void stage_tile(uint4* __restrict__ tile, const T* __restrict__ src, const T* __restrict__ src2,
                                           const float (&k0)[Vec<T>::N], const float (&k1)[Vec<T>::N],
                                           const float (&k2)[Vec<T>::N], long img_base, int SH, int SW, int C, int c0,
                                           bool cvalid, int gy0, int gx0, int IH, int IW, unsigned magic, int cvb_log2) {
    constexpr int V = Vec<T>::N;
    const int total = (IH * IW) << cvb_log2;
    for (int base = threadIdx.x; base < total; base += DFD_THREADS * 4) {
        uint4 raw[4], raw2[4];
        bool inb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * DFD_THREADS;
            const int pix = idx >> cvb_log2;
            const int iy = (int)(((unsigned)pix * magic) >> 20);
            const int ix = pix - iy * IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            inb[u] = cvalid && idx < total && gy >= 0 && gy < SH && gx >= 0 && gx < SW;
            if (inb[u]) {
                const long off = img_base + ((long)gy * SW + gx) * C + c0;
                raw[u] = *reinterpret_cast<const uint4*>(src + off);
                if constexpr (MODE == 2) raw2[u] = *reinterpret_cast<const uint4*>(src2 + off);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * DFD_THREADS;
            if (idx >= total) continue;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (inb[u]) {
                if constexpr (MODE == 0) {
                    q = raw[u];
                } else {
                    float v[V], v2[V];
                    if constexpr (V == 8) Vec<bf16>::unpack(raw[u], v);
                    else { v[0] = __uint_as_float(raw[u].x); v[1] = __uint_as_float(raw[u].y); v[2] = __uint_as_float(raw[u].z); v[3] = __uint_as_float(raw[u].w); }
                    if constexpr (MODE == 2) {
                        if constexpr (V == 8) Vec<bf16>::unpack(raw2[u], v2);
                        else { v2[0] = __uint_as_float(raw2[u].x); v2[1] = __uint_as_float(raw2[u].y); v2[2] = __uint_as_float(raw2[u].z); v2[3] = __uint_as_float(raw2[u].w); }
                    }
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        if constexpr (MODE == 1) v[j] = act_fwd<ACT>(fmaf(k0[j], v[j], k1[j]));
                        else v[j] = fmaf(k0[j], v[j], fmaf(k1[j], v2[j], k2[j]));
                    }
                    if constexpr (V == 8) q = Vec<bf16>::pack(v);
                    else q = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
                }
            }
            tile[idx] = q;
        }
    }
}

template <typename T> __device__ __forceinline__ void unpack_q(const uint4& q, float (&v)[Vec<T>::N]) {
    if constexpr (Vec<T>::N == 8) Vec<bf16>::unpack(q, v);
    else { v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w); }
}

// weights of the workgroup's channel chunk -> LDS as [tap][CVB*V] f32, rounded to T
template <typename T, int K>
__device__ __forceinline__ void stage_weights(float* __restrict__ wl, const float* __restrict__ w, int C, int chunk_c0, int cvb) {
    constexpr int V = Vec<T>::N;
    const int per_tap = cvb * V;
    for (int i = threadIdx.x; i < K * K * per_tap; i += DFD_THREADS) {
        const int tap = i / per_tap, cc = i - tap * per_tap;
        const int c = chunk_c0 + cc;
        wl[i] = c < C ? round_to<T>(w[(long)c * K * K + tap]) : 0.f;
    }
}

// ---------------------------------------------------------------------------
// data gradient: centre pixels are INPUT pixels (h, w); the staged tile is the
// dy region that can reach them.  g.IH/IW = staged dy extent, CH/CW = H/W.
// ---------------------------------------------------------------------------
template <typename T, int K, int S, int ACT, bool COEF, bool EPI>
__global__ void __launch_bounds__(DFD_THREADS, 4)
k_dw_bwd_data(const T* __restrict__ dz, const T* __restrict__ yraw, const float* __restrict__ coef,
              const float* __restrict__ w, const T* __restrict__ xin, const float* __restrict__ in_bnstate,
              T* __restrict__ dzin, DwGeom g, float* __restrict__ partials, int tile_bytes) {
    constexpr int V = Vec<T>::N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);
    float* wl = reinterpret_cast<float*>(smem + tile_bytes);
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2;
    const int t = threadIdx.x, vl = t & (cvb - 1), pl = t >> g.cvb_log2;
    const int vglob = blockIdx.y * cvb + vl;
    const bool cvalid = vglob < g.CV;
    const int c0 = vglob * V;
    const int TW = 1 << g.tw_log2;

    stage_weights<T, K>(wl, w, g.C, blockIdx.y * cvb * V, cvb);
    float ka[V], kb[V], kc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { ka[j] = 1.f; kb[j] = 0.f; kc[j] = 0.f; }
    if (COEF && cvalid) { load_f32<V>(coef + c0, ka); load_f32<V>(coef + g.C + c0, kb); load_f32<V>(coef + 2 * g.C + c0, kc); }
    float sc[V], sh[V], mean[V], rstd[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = 1.f; sh[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f; }
    if (EPI && cvalid) {
        load_f32<V>(in_bnstate + c0, sc); load_f32<V>(in_bnstate + g.C + c0, sh);
        load_f32<V>(in_bnstate + 2 * g.C + c0, mean); load_f32<V>(in_bnstate + 3 * g.C + c0, rstd);
    }
    int hyl[DW_NP], hxl[DW_NP];
#pragma unroll
    for (int i = 0; i < DW_NP; ++i) {
        const int p = pl + i * PL;
        hyl[i] = p >> g.tw_log2;
        hxl[i] = p & (TW - 1);
    }
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    const int tiles = g.tiles_y * g.tiles_x;
    for (int work = blockIdx.x; work < g.nwork; work += gridDim.x) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int h0 = ty * g.TH, w0 = tx * TW;
        // first dy row/col that any centre pixel of this tile can touch (floor division)
        const int ny = h0 + g.pt - (K - 1), nx = w0 + g.pl - (K - 1);
        const int sy0 = (S == 1) ? ny : (ny >= 0 ? ny / 2 : -((-ny + 1) / 2));
        const int sx0 = (S == 1) ? nx : (nx >= 0 ? nx / 2 : -((-nx + 1) / 2));
        __syncthreads();
        stage_tile<T, DFD_ACT_NONE, COEF ? 2 : 0>(tile, dz, yraw, ka, kb, kc, (long)n * g.Ho * g.Wo * g.C, g.Ho, g.Wo, g.C,
                                                  c0, cvalid, sy0, sx0, g.IH, g.IW, g.iw_magic, g.cvb_log2);
        __syncthreads();
#pragma unroll 1
        for (int i = 0; i < DW_NP; ++i) {
            const int h = h0 + hyl[i], wq = w0 + hxl[i];
            if (!(cvalid && hyl[i] < g.TH && h < g.H && wq < g.W)) continue;
            float acc[V];
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] = 0.f;
            const int by = h + g.pt, bx = wq + g.pl;
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int numy = by - kh;
                if (S == 2 && (numy & 1)) continue;
                const int ry = (S == 1 ? numy : (numy >> 1)) - sy0;
#pragma unroll
                for (int kw = 0; kw < K; ++kw) {
                    const int numx = bx - kw;
                    if (S == 2 && (numx & 1)) continue;
                    const int rx = (S == 1 ? numx : (numx >> 1)) - sx0;
                    float wv[V], dv[V];
                    load_f32<V>(wl + ((kh * K + kw) * cvb + vl) * V, wv);
                    unpack_q<T>(tile[((ry * g.IW + rx) << g.cvb_log2) + vl], dv);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] = fmaf(dv[j], wv[j], acc[j]);
                }
            }
            const long off = (((long)n * g.H + h) * g.W + wq) * g.C + c0;
            if constexpr (EPI) {
                float xv[V];
                Vec<T>::load(xin + off, xv);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float z = fmaf(sc[j], xv[j], sh[j]);
                    const float d = round_to<T>(acc[j] * act_grad<ACT>(z));
                    acc[j] = d;
                    s1[j] += d;
                    s2[j] = fmaf(d, (xv[j] - mean[j]) * rstd[j], s2[j]);
                }
            }
            Vec<T>::store(dzin + off, acc);
        }
    }
    if constexpr (EPI) {
        __syncthreads();
        float acc2[2 * V];
#pragma unroll
        for (int j = 0; j < V; ++j) { acc2[j] = s1[j]; acc2[V + j] = s2[j]; }
        reduce_rowlanes<2 * V>(acc2, reinterpret_cast<float*>(smem), cvb, PL, vl, pl, true);
        if (pl == 0 && cvalid) {
            float* p = partials + (long)blockIdx.x * 2 * g.C;
            float a0[V], a1[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { a0[j] = acc2[j]; a1[j] = acc2[V + j]; }
            store_f32<V>(p + c0, a0);
            store_f32<V>(p + g.C + c0, a1);
        }
    }
}

// ---------------------------------------------------------------------------
// weight gradient: lanes are (channel vector, tap, pixel subgroup); both the
// activated input tile (with halo) and the dy tile live in LDS.
// partial layout: [workgroup][C][K*K]
// ---------------------------------------------------------------------------
template <typename T, int K, int S, int ACT, bool PRO, bool COEF>
__global__ void __launch_bounds__(DFD_THREADS)
k_dw_bwd_weight(const T* __restrict__ dz, const T* __restrict__ yraw, const float* __restrict__ coef,
                const T* __restrict__ xin, const float* __restrict__ in_bnstate, DwGeom g,
                float* __restrict__ partials, int tile_bytes) {
    constexpr int V = Vec<T>::N;
    constexpr int KK = K * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);
    uint4* dyt = reinterpret_cast<uint4*>(smem + tile_bytes);
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2;
    const int t = threadIdx.x, vl = t & (cvb - 1), gl = t >> g.cvb_log2;
    const int vglob = blockIdx.y * cvb + vl;
    const bool cvalid = vglob < g.CV;
    const int c0 = vglob * V;
    const int TW = 1 << g.tw_log2;
    const int NSG = PL / KK;
    const int tap = gl % KK, sg = gl / KK;
    const bool lane_on = sg < NSG;
    const int kh = tap / K, kw = tap - kh * K;

    float sc[V], sh[V], zz[V], ka[V], kb[V], kc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = 1.f; sh[j] = 0.f; zz[j] = 0.f; ka[j] = 1.f; kb[j] = 0.f; kc[j] = 0.f; }
    if (PRO && cvalid) { load_f32<V>(in_bnstate + c0, sc); load_f32<V>(in_bnstate + g.C + c0, sh); }
    if (COEF && cvalid) { load_f32<V>(coef + c0, ka); load_f32<V>(coef + g.C + c0, kb); load_f32<V>(coef + 2 * g.C + c0, kc); }

    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const int npix = g.TH << g.tw_log2;
    const int tiles = g.tiles_y * g.tiles_x;
    const unsigned tw_magic = (1u << 20) / TW;  // exact: TW is a power of two
    for (int work = blockIdx.x; work < g.nwork; work += gridDim.x) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int oy0 = ty * g.TH, ox0 = tx * TW;
        __syncthreads();
        stage_tile<T, ACT, PRO ? 1 : 0>(tile, xin, xin, sc, sh, zz, (long)n * g.H * g.W * g.C, g.H, g.W, g.C, c0, cvalid,
                                        oy0 * S - g.pt, ox0 * S - g.pl, g.IH, g.IW, g.iw_magic, g.cvb_log2);
        stage_tile<T, DFD_ACT_NONE, COEF ? 2 : 0>(dyt, dz, yraw, ka, kb, kc, (long)n * g.Ho * g.Wo * g.C, g.Ho, g.Wo, g.C,
                                                  c0, cvalid, oy0, ox0, g.TH, TW, tw_magic, g.cvb_log2);
        __syncthreads();
        if (lane_on) {
            for (int p = sg; p < npix; p += NSG) {
                const int oy = p >> g.tw_log2, ox = p & (TW - 1);
                float dv[V], av[V];
                unpack_q<T>(dyt[(p << g.cvb_log2) + vl], dv);
                unpack_q<T>(tile[((((oy * S + kh) * g.IW) + ox * S + kw) << g.cvb_log2) + vl], av);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(dv[j], av[j], acc[j]);
            }
        }
    }
    // reduce over pixel subgroups through LDS: red[sg][tap][vl][V]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane_on) {
#pragma unroll
        for (int j = 0; j < V; ++j) red[((sg * KK + tap) * cvb + vl) * V + j] = acc[j];
    }
    __syncthreads();
    if (lane_on && sg == 0 && cvalid) {
        float* p = partials + (long)blockIdx.x * g.C * KK;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float s = 0.f;
            for (int q = 0; q < NSG; ++q) s += red[((q * KK + tap) * cvb + vl) * V + j];
            p[(long)(c0 + j) * KK + tap] = s;
        }
    }
}

// out[i] (+)= sum_p partials[p][i], in a fixed order.  Two stages when there are many
// partial rows: groups of SUM_GROUP rows are summed by independent workgroups into the
// rows that FOLLOW the slab in the workspace ([P .. P + ceil(P/SUM_GROUP))), then those.
#define SUM_GROUP 32
__global__ void k_sum_partials(const float* __restrict__ partials, int P, long L, float* __restrict__ out,
                               long out_stride, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int p0 = blockIdx.y * SUM_GROUP;
    const int p1 = (p0 + SUM_GROUP < P) ? p0 + SUM_GROUP : P;
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += partials[(long)p * L + i];
    float* o = out + (long)blockIdx.y * out_stride + i;
    *o = (accumulate ? *o : 0.f) + s;
}
// `partials` must have room for P + ceil(P / SUM_GROUP) rows of L floats.
int dfd_launch_sum_partials(float* partials, int P, long L, float* out, int accumulate, hipStream_t st) {
    const int threads = 256;
    const unsigned gx = (unsigned)((L + threads - 1) / threads);
    if (P > SUM_GROUP) {
        const int G = (P + SUM_GROUP - 1) / SUM_GROUP;          // <= 32 for P <= 1024
        float* mid = partials + (long)P * L;
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, G), dim3(threads), 0, st, partials, P, L, mid, L, 0);
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, mid, G, L, out, 0, accumulate);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, partials, P, L, out, 0, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}

// ---------------------------------------------------------------------------
// host: geometry
// ---------------------------------------------------------------------------
static int ilog2_pow2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// centre grid CHxCW, halo relation: staged extent for a TH x TW centre tile
//   fwd / wgrad : IH = (TH-1)*S + K          (input pixels under the outputs)
//   bwd_data    : IH = (TH + K - 2)/S + 2    (dy pixels reaching the inputs)
static bool make_geom(const dfd_dwconv_shape* s, int vec, bool centre_is_input, int max_cvb, size_t extra_per_pix_bytes,
                      DwGeom* g, int* tile_bytes) {
    if (!s || s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Ho <= 0 || s->Wo <= 0 || s->C <= 0 || s->C % 8) return false;
    if (!(s->k == 3 || s->k == 5) || !(s->stride == 1 || s->stride == 2)) return false;
    if (s->pad_top < 0 || s->pad_left < 0 || s->pad_top >= s->k || s->pad_left >= s->k) return false;
    // every output must be reachable: (Ho-1)*S - pt <= H-1  (trailing padding is implicit zeros)
    if ((s->Ho - 1) * s->stride - s->pad_top > s->H - 1 || (s->Wo - 1) * s->stride - s->pad_left > s->W - 1) return false;
    g->N = s->N; g->H = s->H; g->W = s->W; g->C = s->C; g->Ho = s->Ho; g->Wo = s->Wo; g->pt = s->pad_top; g->pl = s->pad_left;
    g->CV = s->C / vec;
    // power-of-two channel-vector chunk: least padding first, then the widest
    int best = 1, best_waste = 1 << 30;
    for (int c = max_cvb; c >= 4; c >>= 1) {
        const int chunks = (g->CV + c - 1) / c;
        const int waste = chunks * c - g->CV;
        if (waste < best_waste) { best = c; best_waste = waste; }
    }
    if (g->CV < 4) best = g->CV >= 2 ? 2 : 1;
    g->cvb_log2 = ilog2_pow2(best);
    const int cvb = 1 << g->cvb_log2, PL = DFD_THREADS / cvb;
    g->CH = centre_is_input ? s->H : s->Ho;
    g->CW = centre_is_input ? s->W : s->Wo;
    int TW = 16;
    if (g->CW <= 8) TW = 8;
    if (g->CW <= 4) TW = 4;
    int TH = (DW_NP * PL) / TW;
    if (TH > g->CH) TH = g->CH;
    if (TH < 1) TH = 1;
    const int K = s->k, S = s->stride;
    for (;;) {
        g->TH = TH; g->tw_log2 = ilog2_pow2(TW);
        if (centre_is_input) { g->IH = (TH + K - 2) / S + 2; g->IW = (TW + K - 2) / S + 2; }
        else { g->IH = (TH - 1) * S + K; g->IW = (TW - 1) * S + K; }
        size_t bytes = (size_t)g->IH * g->IW * cvb * 16 + (size_t)TH * TW * cvb * extra_per_pix_bytes;
        if ((bytes <= 40 * 1024 && g->IH * g->IW < 4096) || TH == 1) { *tile_bytes = (int)((size_t)g->IH * g->IW * cvb * 16); break; }
        TH = TH > 2 ? TH / 2 : 1;
    }
    if (g->IH * g->IW >= 4096) return false;
    g->iw_magic = ((1u << 20) + g->IW - 1) / g->IW;
    g->tiles_y = (g->CH + g->TH - 1) / g->TH;
    g->tiles_x = (g->CW + TW - 1) / TW;
    g->nwork = s->N * g->tiles_y * g->tiles_x;
    return true;
}

static inline int dw_grid_x(int nwork, int nchunks, int pcap) {
    int cap = pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS;
    // aim at ~2048 resident workgroups in total
    int want = 2048 / (nchunks > 0 ? nchunks : 1);
    if (want < 64) want = 64;
    if (want > cap) want = cap;
    if (want > nwork) want = nwork;
    return want < 1 ? 1 : want;
}

#define DISPATCH_KS(KV, SV, ...)                                                        \
    if (KV == 3 && SV == 1) { constexpr int K = 3, S = 1; __VA_ARGS__; }                \
    else if (KV == 3 && SV == 2) { constexpr int K = 3, S = 2; __VA_ARGS__; }           \
    else if (KV == 5 && SV == 1) { constexpr int K = 5, S = 1; __VA_ARGS__; }           \
    else if (KV == 5 && SV == 2) { constexpr int K = 5, S = 2; __VA_ARGS__; }           \
    else return DFD_EUNSUPPORTED;

template <typename T>
static int dw_bwd_data_t(const void* dz, const void* y, const float* coef, const float* w, const void* xin,
                         const float* in_bnstate, int in_act, void* dzin, const dfd_dwconv_shape* s, float* partials,
                         int pcap, int* nparts, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwGeom g; int tile_bytes;
    if (!make_geom(s, V, true, 16, 0, &g, &tile_bytes)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    const bool epi = xin != nullptr;
    if (epi && (!in_bnstate || !partials || !nparts || pcap < 1)) return DFD_EINVAL;
    const int gx = dw_grid_x(g.nwork, nchunks, epi ? pcap : DFD_MAX_PARTIALS);
    if (epi) *nparts = gx;
    size_t lds = (size_t)tile_bytes + (size_t)s->k * s->k * cvb * V * 4;
    const size_t red = (size_t)DFD_THREADS * 2 * V * 4;
    if (lds < red) lds = red;
    dim3 grid(gx, nchunks);
    const bool hc = coef != nullptr;
    if (hc && !y) return DFD_EINVAL;
    if (!epi) in_act = DFD_ACT_NONE;
#define LAUNCH_BD(COEF, EPI) \
    hipLaunchKernelGGL((k_dw_bwd_data<T, K, S, ACT, COEF, EPI>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, (const T*)y, \
                       coef, w, (const T*)xin, in_bnstate, (T*)dzin, g, partials, tile_bytes)
    DISPATCH_KS(s->k, s->stride, {
        if (epi) {
            DISPATCH_ACT(in_act, { if (hc) LAUNCH_BD(true, true); else LAUNCH_BD(false, true); });
        } else {
            constexpr int ACT = DFD_ACT_NONE;
            if (hc) LAUNCH_BD(true, false); else LAUNCH_BD(false, false);
        }
    });
#undef LAUNCH_BD
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_dwconv_bwd_data(int dtype, const void* dz, const void* y, const float* coef, const float* w,
                                   const void* xin, const float* in_bnstate, int in_act, void* dzin,
                                   const dfd_dwconv_shape* s, float* partials, int pcap, int* nparts,
                                   dfd_stream stream) {
    if (!dz || !w || !dzin || !s) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16)
        return dw_bwd_data_t<bf16>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, s, partials, pcap, nparts, st);
    if (dtype == DFD_F32)
        return dw_bwd_data_t<float>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, s, partials, pcap, nparts, st);
    return DFD_EINVAL;
}

static int dw_wgrad_parts(const DwGeom& g, int k, int nchunks) {
    int want = 1024 / (nchunks > 0 ? nchunks : 1);
    if (want < 32) want = 32;
    // keep the partial slab under 16 MiB
    const long per = (long)g.C * k * k * 4;
    long cap = (16l << 20) / per;
    if (cap < 32) cap = 32;
    if (want > cap) want = (int)cap;
    if (want > g.nwork) want = g.nwork;
    return want < 1 ? 1 : want;
}

template <typename T>
static bool dw_wgrad_geom(const dfd_dwconv_shape* s, DwGeom* g, int* tile_bytes) {
    const int max_cvb = (s && s->k == 5) ? 8 : 16;  // needs 256/cvb >= k*k tap lanes
    return make_geom(s, Vec<T>::N, false, max_cvb, 16, g, tile_bytes);
}

extern "C" size_t dfd_dwconv_bwd_weight_ws(const dfd_dwconv_shape* s) {
    DwGeom g; int tile_bytes;
    if (!dw_wgrad_geom<bf16>(s, &g, &tile_bytes)) return 0;
    const size_t per = (size_t)g.C * s->k * s->k * 4;
    const size_t a = (size_t)16 << 20, b = per * 32;
    return (a > b ? a : b) + per * 34;      // + second-stage rows of the partial-slab reduction
}

template <typename T>
static int dw_bwd_weight_t(const void* dz, const void* y, const float* coef, const void* xin, const float* in_bnstate,
                           int in_act, float* dw, const dfd_dwconv_shape* s, int accumulate, float* ws,
                           size_t ws_bytes, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwGeom g; int tile_bytes;
    if (!dw_wgrad_geom<T>(s, &g, &tile_bytes)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    const int KK = s->k * s->k;
    const int P = dw_wgrad_parts(g, s->k, nchunks);
    if ((size_t)(P + P / 32 + 2) * g.C * KK * 4 > ws_bytes) return DFD_EWORKSPACE;
    const int TW = 1 << g.tw_log2;
    size_t lds = (size_t)tile_bytes + (size_t)g.TH * TW * cvb * 16;
    const size_t red = (size_t)((DFD_THREADS / cvb) / KK) * KK * cvb * V * 4;
    if (lds < red) lds = red;
    dim3 grid(P, nchunks);
    const bool pro = in_bnstate != nullptr, hc = coef != nullptr;
    if (hc && !y) return DFD_EINVAL;
    if (!pro) in_act = DFD_ACT_NONE;
#define LAUNCH_BW(PRO, COEF) \
    hipLaunchKernelGGL((k_dw_bwd_weight<T, K, S, ACT, PRO, COEF>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, \
                       (const T*)y, coef, (const T*)xin, in_bnstate, g, ws, tile_bytes)
    DISPATCH_KS(s->k, s->stride, {
        if (pro) {
            DISPATCH_ACT(in_act, { if (hc) LAUNCH_BW(true, true); else LAUNCH_BW(true, false); });
        } else {
            constexpr int ACT = DFD_ACT_NONE;
            if (hc) LAUNCH_BW(false, true); else LAUNCH_BW(false, false);
        }
    });
#undef LAUNCH_BW
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, P, (long)g.C * KK, dw, accumulate, st);
}

extern "C" int dfd_dwconv_bwd_weight(int dtype, const void* dz, const void* y, const float* coef, const void* xin,
                                     const float* in_bnstate, int in_act, float* dw, const dfd_dwconv_shape* s,
                                     int accumulate, float* ws, size_t ws_bytes, dfd_stream stream) {
    if (!dz || !xin || !dw || !s || !ws) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16)
        return dw_bwd_weight_t<bf16>(dz, y, coef, xin, in_bnstate, in_act, dw, s, accumulate, ws, ws_bytes, st);
    if (dtype == DFD_F32)
        return dw_bwd_weight_t<float>(dz, y, coef, xin, in_bnstate, in_act, dw, s, accumulate, ws, ws_bytes, st);
    return DFD_EINVAL;
}
