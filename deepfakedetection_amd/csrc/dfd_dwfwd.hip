// dfd_dwfwd.hip — depthwise k x k convolution, forward, second-generation kernel.
//
// Measured motivation (profiles/, rocprofv3 PMC on MI355X): the first kernel spent ~440
// VALU lane-ops per 16-byte output vector and ran at ~60 % VALU utilisation, i.e. it was
// issue-bound at ~1.3 TB/s although the data path is HBM-bound by arithmetic intensity.
// This version cuts the instruction count:
//   * a lane computes a QUAD of four horizontally adjacent outputs, so each staged input
//     vector is read from LDS and unpacked ONCE per kernel row and feeds up to 4*K taps;
//   * all arithmetic is on float2 pairs -> v_pk_fma_f32 / v_pk_mul_f32 (two lanes' worth
//     of FMAs per instruction);
//   * tiles are chosen per layer by a small cost model (staged pixels vs. idle lanes vs.
//     LDS bytes) instead of a fixed 8x16, so 14x14 / 7x7 images are one tile, not two
//     ragged ones, and wide images use full-width strips with little halo;
//   * the workgroup walks (image, tile) work items persistently as before: one partial
//     (sum, sumsq) row per workgroup, BN + SiLU of the producer applied once per staged
//     element, zero padding written in the activated domain.
#include "dfd_dwq.h"
#include "dfd_dwm.h"
#include <climits>

#ifndef DW_FWD_PAIRS
#define DW_FWD_PAIRS 0      // 1: 3x3 stride-1 bf16 forward with two outputs per lane, five workgroups per CU — measured 3-12 % SLOWER
#endif                      //    on every EfficientNet-B0 / EfficientFormerV2 layer, so off; -DDW_FWD_PAIRS=1 builds it

// EACT >= 0 (eval / inference, where this layer's own BatchNorm is an affine map known up front): the epilogue stores
// act(BN(y)) instead of the raw output and leaves per-(tile, image) channel sums of it for the squeeze-excite pooling
// (pool_parts [tiles][N][C], the layout dfd_se_fwd's per-image kernel adds up) — the consumer needs no prologue and
// the pooling pass disappears.
// OPL = outputs per lane: 4 (a quad) or 2 (half a quad: the same tiles and staging, each lane item is a PAIR of adjacent
// outputs — 16 accumulator registers instead of 32 and K + S instead of K + 3 S staged columns per kernel row, so that five
// workgroups fit a CU (first option; per-layer A/B in DESIGN 9).
template <typename T, int K, int S, int ACT, bool PRO, bool STATS, int EACT = -1, int OPL = 4>
__global__ void __launch_bounds__(DFD_THREADS, OPL == 2 ? 5 : 4)
k_dw_fwd_q(const T* __restrict__ x, const float* __restrict__ bnstate, const float* __restrict__ w, T* __restrict__ y,
           DwQGeom g, float* __restrict__ partials, int tile_bytes, const float* __restrict__ ebn = nullptr,
           float* __restrict__ pool_parts = nullptr) {
    constexpr int V = Vec<T>::N, N2 = V / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);
    float* wl = reinterpret_cast<float*>(smem + tile_bytes);      // [tap][cvb*V] f32, rounded to T
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2;
    const int t = threadIdx.x, vl = t & (cvb - 1), lane = t >> g.cvb_log2;
    int bx, by;
    dwq_block(bx, by, g.remap);
    const int vglob = bx * cvb + vl;           // channel chunk is the FAST grid index:
    const bool cvalid = vglob < g.CV;                  // neighbouring chunks of one tile run together
    const int c0 = vglob * V;

    // all of a lane's (strided, cache-line-per-lane) weight loads are requested before the first is stored: a plain
    // loop pays one memory round trip per pass at the head of every workgroup
    {
        constexpr int NW = (K * K * 16 * V + DFD_THREADS - 1) / DFD_THREADS;       // cvb <= 16
        float wr[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = t + u * DFD_THREADS;
            wr[u] = 0.f;
            if (i < K * K * cvb * V) {
                const int tap = i / (cvb * V), cc = i - tap * (cvb * V);
                const int c = bx * cvb * V + cc;
                if (c < g.C) wr[u] = w[(long)c * K * K + tap];
            }
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = t + u * DFD_THREADS;
            if (i < K * K * cvb * V) wl[i] = round_to<T>(wr[u]);
        }
    }
    float* cf = wl + K * K * cvb * V;                              // scale, shift : [2][cvb*V]
    for (int i = t; i < 2 * cvb * V; i += DFD_THREADS) {
        const int which = i / (cvb * V), cc = i - which * (cvb * V), c = bx * cvb * V + cc;
        cf[i] = (PRO && c < g.C) ? bnstate[which * g.C + c] : (which == 0 ? 1.f : 0.f);
    }
    f2 s1[N2], s2[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) { s1[j] = (f2){0.f, 0.f}; s2[j] = (f2){0.f, 0.f}; }
    f2 esc[N2], esh[N2], pool[N2];
    if constexpr (EACT >= 0) {
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            esc[j] = cvalid ? *reinterpret_cast<const f2*>(ebn + c0 + 2 * j) : (f2){0.f, 0.f};
            esh[j] = cvalid ? *reinterpret_cast<const f2*>(ebn + g.C + c0 + 2 * j) : (f2){0.f, 0.f};
            pool[j] = (f2){0.f, 0.f};
        }
    }

    const int tiles = g.tiles_y * g.tiles_x, TW = 4 * g.QW;
    for (int work = by; work < g.nwork; work += gridDim.y) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int oy0 = ty * g.TH, ox0 = tx * TW;
        __syncthreads();
        {
            f2 sc[N2], sh[N2];                       // short-lived: the coefficients live in LDS
            lds_row<N2>(cf + vl * V, sc);
            lds_row<N2>(cf + cvb * V + vl * V, sh);
            stage_q<T, ACT, PRO, StageDepth<K, S>::X>(tile, x, sc, sh, (long)n * g.H * g.W * g.C, g.H, g.W, g.C, c0, cvalid, oy0 * S - g.pt,
                                 ox0 * S - g.pl, g.IH, g.IW, g.iw_magic, g.cvb_log2);
        }
        __syncthreads();
        if (EACT < 0 && !cvalid) continue;
        constexpr int HPQ = 4 / OPL;                     // lane items per quad
#pragma unroll 1
        for (int it = cvalid ? lane : HPQ * g.NQ; it < HPQ * g.NQ; it += PL) {
            const int q = OPL == 4 ? it : (it >> 1), sub = OPL == 4 ? 0 : (it & 1) * OPL;
            const int qy = (int)(((unsigned)q * g.qw_magic) >> 20), qx = q - qy * g.QW;
            const int oy = oy0 + qy, ox = ox0 + 4 * qx + sub;
            if (oy >= g.Ho || ox >= g.Wo) continue;
            f2 acc[OPL][N2];
#pragma unroll
            for (int o = 0; o < OPL; ++o)
#pragma unroll
                for (int j = 0; j < N2; ++j) acc[o][j] = (f2){0.f, 0.f};
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                const uint4* row = tile + ((((qy * S + kh) * g.IW) + (qx * 4 + sub) * S) << g.cvb_log2) + vl;
                f2 wv[K][N2];
#pragma unroll
                for (int kw = 0; kw < K; ++kw) lds_row<N2>(wl + ((kh * K + kw) * cvb + vl) * V, wv[kw]);
                // walk the INPUT columns of the quad's window: one vector is unpacked at a time and
                // feeds every (output, tap) pair it belongs to (keeps the live set at acc + weights)
                constexpr int NCOL = (OPL - 1) * S + K;
#pragma unroll
                for (int c = 0; c < NCOL; ++c) {
                    f2 xc[N2];
                    unpack2(row[c << g.cvb_log2], xc);
#pragma unroll
                    for (int o = 0; o < OPL; ++o) {
                        const int kw = c - S * o;
                        if (kw >= 0 && kw < K) {
#pragma unroll
                            for (int j = 0; j < N2; ++j) acc[o][j] = __builtin_elementwise_fma(xc[j], wv[kw][j], acc[o][j]);
                        }
                    }
                }
            }
            T* dst = y + (((long)n * g.Ho + oy) * g.Wo + ox) * g.C + c0;
#pragma unroll
            for (int o = 0; o < OPL; ++o) {
                if (ox + o < g.Wo) {
#pragma unroll
                    for (int j = 0; j < N2; ++j) {
                        f2 r = round2<T>(acc[o][j]);
                        if constexpr (EACT >= 0) {
                            const f2 z = __builtin_elementwise_fma(esc[j], r, esh[j]);
                            r = round2<T>(act_fwd2<EACT>(z));
                            pool[j] += r;
                        }
                        acc[o][j] = r;
                        if constexpr (STATS) { s1[j] += r; s2[j] = __builtin_elementwise_fma(r, r, s2[j]); }
                    }
                    *reinterpret_cast<uint4*>(dst + (long)o * g.C) = pack2(acc[o]);
                }
            }
        }
        if constexpr (EACT >= 0) {
            // this work item's channel sums of the activated output: row lanes combined in a fixed order, one row per
            // (tile, image) — the per-image squeeze-excite kernel adds the tiles of an image in tile order
            float pa[V];
#pragma unroll
            for (int j = 0; j < N2; ++j) { pa[2 * j] = pool[j].x; pa[2 * j + 1] = pool[j].y; pool[j] = (f2){0.f, 0.f}; }
            reduce_rowlanes<V>(pa, cf + 2 * cvb * V, cvb, PL, vl, lane, cvalid);        // its own region: the weights stay put
            if (lane == 0 && cvalid) store_f32<V>(pool_parts + ((long)tr * g.N + n) * g.C + c0, pa);
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        float acc2[2 * V];
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            acc2[2 * j] = s1[j].x; acc2[2 * j + 1] = s1[j].y;
            acc2[V + 2 * j] = s2[j].x; acc2[V + 2 * j + 1] = s2[j].y;
        }
        reduce_rowlanes<2 * V>(acc2, reinterpret_cast<float*>(smem), cvb, PL, vl, lane, true);
        if (lane == 0 && cvalid) {
            float* p = partials + (long)by * 2 * g.C;
            float a0[V], a1[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { a0[j] = acc2[j]; a1[j] = acc2[V + j]; }
            store_f32<V>(p + c0, a0);
            store_f32<V>(p + g.C + c0, a1);
        }
    }
}

// ---------------------------------------------------------------------------
// host: tile selection by a small cost model
// ---------------------------------------------------------------------------
static int ilog2p(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

bool dfd_dwq_geom(const dfd_dwconv_shape* s, int vec, int max_cvb, bool centre_is_input, size_t extra_lds,
                  int extra_centre, int lane_div, DwQGeom* g, int* tile_bytes, int halo_tiles, long lds_budget) {
    if (!s || s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Ho <= 0 || s->Wo <= 0 || s->C <= 0 || s->C % 8) return false;
    if (!(s->k == 3 || s->k == 5) || !(s->stride == 1 || s->stride == 2)) return false;
    if (s->pad_top < 0 || s->pad_left < 0 || s->pad_top >= s->k || s->pad_left >= s->k) return false;
    if ((s->Ho - 1) * s->stride - s->pad_top > s->H - 1 || (s->Wo - 1) * s->stride - s->pad_left > s->W - 1) return false;
    g->N = s->N; g->H = s->H; g->W = s->W; g->C = s->C; g->Ho = s->Ho; g->Wo = s->Wo; g->pt = s->pad_top; g->pl = s->pad_left;
    g->CV = s->C / vec;
    int best = 1, best_waste = 1 << 30;
    for (int c = max_cvb; c >= 4; c >>= 1) {
        const int waste = ((g->CV + c - 1) / c) * c - g->CV;
        if (waste < best_waste) { best = c; best_waste = waste; }
    }
    // 3x3 layers: a wide chunk (256-byte pieces) beats a perfectly divisible narrow one as long as
    // few lanes idle (measured: C = 480, cvb 16 vs 4: 57 vs 72 us); 5x5 layers keep the narrow chunk
    // because their tap table would eat the tile's LDS
    if (s->k == 3) {
        for (int c = max_cvb; c > best; c >>= 1) {
            const int padded = ((g->CV + c - 1) / c) * c;
            if ((padded - g->CV) * 12 <= padded) { best = c; break; }
        }
    }
    if (g->CV < 4) best = g->CV >= 2 ? 2 : 1;
    g->cvb_log2 = ilog2p(best);
    const int cvb = best, PL = DFD_THREADS / cvb, K = s->k, S = s->stride;
    // centre grid: outputs (forward) or inputs (data gradient); the staged tile is the other side
    const int CH = centre_is_input ? s->H : s->Ho, CW = centre_is_input ? s->W : s->Wo;
    auto ext = [&](int centre) {            // staged extent along one axis for `centre` centre pixels
        if (!centre_is_input) return (centre - 1) * S + K;
        return S == 1 ? centre + K - 1 : (centre + K - 2) / 2 + 2;
    };
    const int maxQW = (CW + 3) / 4;
    double best_cost = 1e300;
    int bTH = 1, bQW = 1;
    for (int QW = 1; QW <= maxQW; ++QW) {
        const int TW = 4 * QW, IW = ext(TW);
        for (int TH = 1; TH <= CH && TH <= 64; ++TH) {
            const int IH = ext(TH);
            const long px = (long)IH * IW;
            if (px >= 4096) break;
            const long lds = px * cvb * 16 * halo_tiles + (long)extra_lds + (long)TH * TW * cvb * extra_centre;
            if (lds > lds_budget && !(TH == 1 && QW == 1)) break;
            const long tiles = (long)((CH + TH - 1) / TH) * ((CW + TW - 1) / TW);
            const long lanes = PL / lane_div > 0 ? PL / lane_div : 1;      // lanes that share the quads of a tile
            const long rounds = ((long)TH * QW + lanes - 1) / lanes;
            const double stage = (double)((px * cvb + DFD_THREADS - 1) / DFD_THREADS) * 120.0;
            const double quad = 4.0 * K * K * (vec / 2) + (double)K * (K + 3) * (vec / 2) * 2 + 80.0;
            const double cost = (double)tiles * (stage + rounds * quad + 250.0);
            if (cost < best_cost) { best_cost = cost; bTH = TH; bQW = QW; }
        }
    }
    g->TH = bTH; g->QW = bQW; g->NQ = bTH * bQW;
    g->qw_magic = ((1u << 20) + bQW - 1) / bQW;
    g->IH = ext(bTH); g->IW = ext(4 * bQW);
    if ((long)g->IH * g->IW >= 4096 || g->NQ >= 4096) return false;
    g->iw_magic = ((1u << 20) + g->IW - 1) / g->IW;
    g->tiles_y = (CH + bTH - 1) / bTH;
    g->tiles_x = (CW + 4 * bQW - 1) / (4 * bQW);
    if ((long)s->N * g->tiles_y * g->tiles_x > INT_MAX / 2) return false;        // int work counters in the kernels
    g->nwork = s->N * g->tiles_y * g->tiles_x;
    // the kernels divide by QW and IW with 20-bit magic numbers ((v * magic) >> 20): check them over the
    // whole range they are used on (v < NQ and v < IH*IW, both < 4096) instead of trusting the bound
    for (unsigned v = 0; v < (unsigned)g->NQ; ++v)
        if (((v * g->qw_magic) >> 20) != v / (unsigned)bQW) return false;
    for (unsigned v = 0; v < (unsigned)(g->IH * g->IW); ++v)
        if (((v * g->iw_magic) >> 20) != v / (unsigned)g->IW) return false;
    {   // the weight-gradient kernel stages the centre tile with a magic for TW = 4*QW
        const unsigned TW = 4u * (unsigned)bQW, tw_magic = ((1u << 20) + TW - 1) / TW;
        for (unsigned v = 0; v < (unsigned)bTH * TW; ++v)
            if (((v * tw_magic) >> 20) != v / TW) return false;
    }
    *tile_bytes = g->IH * g->IW * cvb * 16;
    g->remap = 0;
    return true;
}

DwqOcc dwq_occupancy(int op, const dfd_dwconv_shape* s) {
    const int mode = dfd_tune_get(DFD_TUNE_DWQ_WIDE);
    bool wide = false;
    if (mode > 0) wide = true;
    else if (mode < 0 && s) {
        const bool k5s1 = s->k == 5 && s->stride == 1;
        if (op == 0) wide = k5s1;
        else if (op == 1) wide = k5s1 || (s->k == 3 && s->stride == 2);
        else wide = k5s1 && s->H * s->W > 64;          // (7x7: the weight gradient's own staging measured 52 -> 55 us wide)
    }
    const int target = dfd_tune_get(op == 0 ? DFD_TUNE_DWQ_GRID_FWD : (op == 1 ? DFD_TUNE_DWQ_GRID_BWD : DFD_TUNE_DWQ_GRID_WGRAD));
    DwqOcc o;
    // narrow: 39 KB x 4 leaves 4 KB of the CU's 160 (36 KB until round 4: EfficientFormerV2-S1 16.39 -> 16.26 ms at 39 / 40 KB, B0 -0.03 ms);
    // wide: 44 / 48 / 50 / 52 KB measured within 0.04 ms of each other on B0
    o.lds_budget = wide ? 48 * 1024 : 39 * 1024;
    o.grid = wide ? target * 3 / 4 : target;
    return o;
}

template <typename T>
static int dw_fwd_q_t(const void* x, const float* in_bnstate, int in_act, const float* w, void* y,
                      const dfd_dwconv_shape* s, float* partials, int pcap, int* nparts, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwQGeom g; int tile_bytes;
    // two outputs per lane (five workgroups per CU: tile + tables within 31 KB): bf16 3x3 stride-1 layers (DW_FWD_PAIRS)
    const bool pairs = DW_FWD_PAIRS && sizeof(T) == 2 && s && s->k == 3 && s->stride == 1;
    const size_t tables = (size_t)(s ? s->k * s->k + 2 : 0) * 16 * V * 4;
    const DwqOcc occ = dwq_occupancy(0, s);
    if (!dfd_dwq_geom(s, V, 16, false, tables, 0, 1, &g, &tile_bytes, 1, pairs ? 31 * 1024 : occ.lds_budget)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    // XCD-aware order (dwq_block): measured per layer — helps the forward kernel when an image spans
    // several tiles and the channel chunks share cache lines (block 2: 219 -> 201 us), hurts when one
    // tile is one image (14x14, 7x7 layers: +8..17 %)
    g.remap = (nchunks > 1 && g.tiles_y * g.tiles_x >= 2 &&
               ((s->C * (int)sizeof(T)) % 128 != 0 || (cvb * 16) % 128 != 0)) ? 1 : 0;
    const bool stats = partials != nullptr;
    int cap = stats ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    int gy = occ.grid / nchunks;
    if (gy < dfd_tune_get(DFD_TUNE_DWQ_GRID_MIN)) gy = dfd_tune_get(DFD_TUNE_DWQ_GRID_MIN);
    if (gy > cap) gy = cap;
    if (gy > g.nwork) gy = g.nwork;
    if (stats) *nparts = gy;
    size_t lds = (size_t)tile_bytes + (size_t)(s->k * s->k + 2) * cvb * V * 4;
    const size_t red = (size_t)DFD_THREADS * 2 * V * 4;
    if (lds < red) lds = red;
    dim3 grid(nchunks, gy);
    const bool pro = in_bnstate != nullptr;
#define LAUNCH_FWD(PRO, STATS)                                                                                               \
    do {                                                                                                                     \
        if constexpr (DW_FWD_PAIRS && sizeof(T) == 2 && K == 3 && S == 1) {                                                  \
            if (pairs) {                                                                                                     \
                hipLaunchKernelGGL((k_dw_fwd_q<T, K, S, ACT, PRO, STATS, -1, 2>), grid, dim3(DFD_THREADS), lds, st, (const T*)x, \
                                   in_bnstate, w, (T*)y, g, partials, tile_bytes);                                           \
                break;                                                                                                       \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((k_dw_fwd_q<T, K, S, ACT, PRO, STATS>), grid, dim3(DFD_THREADS), lds, st, (const T*)x, in_bnstate, w, \
                           (T*)y, g, partials, tile_bytes);                                                                  \
    } while (0)
    DISPATCH_KS(s->k, s->stride, {
        if (pro) {
            DISPATCH_ACT_DW(in_act, { if (stats) LAUNCH_FWD(true, true); else LAUNCH_FWD(true, false); });
        } else {
            constexpr int ACT = DFD_ACT_NONE;
            if (stats) LAUNCH_FWD(false, true); else LAUNCH_FWD(false, false);
        }
    });
#undef LAUNCH_FWD
    return DFD_CHECK_LAUNCH();
}

// ---- eval / inference form: y = act(BN(conv(x))) stored, per-(tile, image) channel sums for the SE pooling
template <typename T>
static int dw_fwd_eval_t(const void* x, const float* w, const float* out_bnstate, int out_act, void* y,
                         const dfd_dwconv_shape* s, float* pool_parts, int* ntiles, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwQGeom g; int tile_bytes;
    if (!dfd_dwq_geom(s, V, 16, false, (size_t)(s ? s->k * s->k + 2 : 0) * 16 * V * 4 + (size_t)DFD_THREADS * V * 4, 0, 1, &g, &tile_bytes)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    g.remap = (nchunks > 1 && g.tiles_y * g.tiles_x >= 2 &&
               ((s->C * (int)sizeof(T)) % 128 != 0 || (cvb * 16) % 128 != 0)) ? 1 : 0;
    int gy = DFD_DW_GRID / nchunks;
    if (gy < 64) gy = 64;
    if (gy > DFD_MAX_PARTIALS) gy = DFD_MAX_PARTIALS;
    if (gy > g.nwork) gy = g.nwork;
    *ntiles = g.tiles_y * g.tiles_x;
    const size_t lds = (size_t)tile_bytes + (size_t)(s->k * s->k + 2) * cvb * V * 4 + (size_t)DFD_THREADS * V * 4;
    dim3 grid(nchunks, gy);
    if (out_act != DFD_ACT_SILU) return DFD_EUNSUPPORTED;          // the EfficientNet blocks; widen when another net needs it
    DISPATCH_KS(s->k, s->stride, {
        hipLaunchKernelGGL((k_dw_fwd_q<T, K, S, DFD_ACT_NONE, false, false, DFD_ACT_SILU>), grid, dim3(DFD_THREADS), lds, st,
                           (const T*)x, (const float*)nullptr, w, (T*)y, g, (float*)nullptr, tile_bytes, out_bnstate, pool_parts);
    });
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_dwconv_fwd_eval_tiles(int dtype, const dfd_dwconv_shape* s) {
    DwQGeom g; int tile_bytes;
    const int V = dtype == DFD_BF16 ? Vec<bf16>::N : Vec<float>::N;
    if (!dfd_dwq_geom(s, V, 16, false, (size_t)(s ? s->k * s->k + 2 : 0) * 16 * V * 4 + (size_t)DFD_THREADS * V * 4, 0, 1, &g, &tile_bytes)) return 0;
    return g.tiles_y * g.tiles_x;
}
extern "C" int dfd_dwconv_fwd_eval(int dtype, const void* x, const float* w, const float* out_bnstate, int out_act, void* y,
                                   const dfd_dwconv_shape* s, float* pool_parts, int* ntiles, dfd_stream stream) {
    if (!x || !w || !out_bnstate || !y || !s || !pool_parts || !ntiles) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) return dw_fwd_eval_t<bf16>(x, w, out_bnstate, out_act, y, s, pool_parts, ntiles, st);
    if (dtype == DFD_F32) return dw_fwd_eval_t<float>(x, w, out_bnstate, out_act, y, s, pool_parts, ntiles, st);
    return DFD_EINVAL;
}

extern "C" int dfd_dwconv_fwd(int dtype, const void* x, const float* in_bnstate, int in_act, const float* w, void* y,
                              const dfd_dwconv_shape* s, float* partials, int pcap, int* nparts, dfd_stream stream) {
    if (!x || !w || !y || !s) return DFD_EINVAL;
    if (partials && (!nparts || pcap < 1)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        const int rc = dfd_dw_fwd_mm(x, in_bnstate, in_act, w, y, s, partials, pcap, nparts, st);      // matrix-core form (dfd_dwmm.hip)
        if (rc != DFD_EUNSUPPORTED) return rc;
        return dw_fwd_q_t<bf16>(x, in_bnstate, in_act, w, y, s, partials, pcap, nparts, st);
    }
    if (dtype == DFD_F32) return dw_fwd_q_t<float>(x, in_bnstate, in_act, w, y, s, partials, pcap, nparts, st);
    return DFD_EINVAL;
}
