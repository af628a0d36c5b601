// dfd_dwbwd.hip — depthwise k x k convolution, data gradient, quad/packed-math kernel.
//
//   dy   = a*dz + b*y + c            (BN-backward affine map, applied once while staging)
//   da[h][w] = sum_{kh,kw} dy[(h+pt-kh)/S][(w+pl-kw)/S] * W[kh][kw]   (divisible terms only)
//   dzin = da * act'(scale*xin + shift)                                (epilogue, optional)
//   partial sums (dzin, dzin*xhat) per channel for the producer's BatchNorm backward
//
// Centre pixels are INPUT pixels; a lane owns a quad of four adjacent input columns.  The dy
// region that can reach a centre tile is staged in LDS.  Stride 1 is a correlation with the
// flipped kernel, so the column walk of the forward kernel applies unchanged.  Stride 2 only
// touches taps of matching parity; because tiles start at even columns the (output, tap) ->
// staged-column map is a compile-time table once the parity of pad_left is a template
// parameter (PLP).
#include "dfd_dwq.h"
#ifndef DFD_ABLATE
#define DFD_ABLATE 0   // timing experiments (scripts/build_variant.sh): 1 = staging only, 2 = taps only
#endif

// WPC = workgroups per CU the instance is compiled for: 4, or 3 for the wide occupancy class of the 5x5 stride-1 layers (dwq_occupancy), where only
// three are resident anyway: 168 instead of 128 registers per lane.  The 4-per-CU 5x5 instances with the epilogue spill 20-48 bytes per lane; the
// 3-per-CU ones take 139-159 registers, no scratch, and twice the staging depth: data gradient 28x28 C240 174 -> 160 us, 14x14 C480 / C672 72 -> 67 /
// 111 -> 98, 7x7 49 -> 48; weight gradient 14x14 71 -> 68 / 101 -> 99 (B0 step -0.05 ms).  The forward has no spills at 4 per CU and gained nothing.
template <typename T, int K, int S, int PLP, int ACT, bool COEF, bool EPI, int WPC = 4>
__global__ void __launch_bounds__(DFD_THREADS, WPC)
k_dw_bwd_data_q(const T* __restrict__ dz, const T* __restrict__ yraw, const float* __restrict__ coef,
                const float* __restrict__ w, const T* __restrict__ xin, const float* __restrict__ in_bnstate,
                T* __restrict__ dzin, DwQGeom g, float* __restrict__ partials, int tile_bytes) {
    constexpr int V = Vec<T>::N, N2 = V / 2;
    // stride 2: E = pad_left + (w0 - 2*sx0) is K-1 or K, whichever has pad_left's parity
    constexpr int E = (((K - 1) & 1) == PLP) ? K - 1 : K;
    constexpr int NCOL = (S == 1) ? K + 3 : (3 + E) / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* tile = reinterpret_cast<uint4*>(smem);
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2, cvbV = cvb * V;
    float* wl = reinterpret_cast<float*>(smem + tile_bytes);      // [tap][cvb*V]
    float* cf = wl + K * K * cvbV;                                // ka, kb, kc, scale, shift : [5][cvb*V]
    const int t = threadIdx.x, vl = t & (cvb - 1), lane = t >> g.cvb_log2;
    int bx, by;
    dwq_block(bx, by, g.remap);
    const int vglob = bx * cvb + vl;
    const bool cvalid = vglob < g.CV;
    const int c0 = vglob * V;
    const int chunk_c0 = bx * cvbV;

    {   // every weight load of the lane in flight at once (see dfd_dwfwd.hip)
        constexpr int NW = (K * K * 16 * V + DFD_THREADS - 1) / DFD_THREADS;       // cvb <= 16
        float wr[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = t + u * DFD_THREADS;
            wr[u] = 0.f;
            if (i < K * K * cvbV) {
                const int tap = i / cvbV, cc = i - tap * cvbV;
                const int c = chunk_c0 + cc;
                if (c < g.C) wr[u] = w[(long)c * K * K + tap];
            }
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            const int i = t + u * DFD_THREADS;
            if (i < K * K * cvbV) wl[i] = round_to<T>(wr[u]);
        }
    }
    for (int i = t; i < 5 * cvbV; i += DFD_THREADS) {
        const int which = i / cvbV, cc = i - which * cvbV;
        const int c = chunk_c0 + cc;
        float v = (which == 0 || which == 3) ? 1.f : 0.f;
        if (c < g.C) {
            if (which < 3) { if (COEF) v = coef[which * g.C + c]; }
            else if (EPI) v = in_bnstate[(which - 3) * g.C + c];
        }
        cf[i] = v;
    }
    f2 s1[N2], s2[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) { s1[j] = (f2){0.f, 0.f}; s2[j] = (f2){0.f, 0.f}; }

    const int tiles = g.tiles_y * g.tiles_x, TW = 4 * g.QW;
    for (int work = by; work < g.nwork; work += gridDim.y) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int h0 = ty * g.TH, w0 = tx * TW;
        const int ny = h0 + g.pt - (K - 1), nx = w0 + g.pl - (K - 1);
        const int sy0 = (S == 1) ? ny : (ny >= 0 ? ny / 2 : -((-ny + 1) / 2));
        const int sx0 = (S == 1) ? nx : (nx >= 0 ? nx / 2 : -((-nx + 1) / 2));
        __syncthreads();
#if DFD_ABLATE != 2
        stage_dy<T, COEF, (WPC == 3 ? 2 : 1) * StageDepth<K, S>::DY>(tile, dz, yraw, cf, cvbV, vl, (long)n * g.Ho * g.Wo * g.C, g.Ho, g.Wo, g.C, c0, cvalid, sy0, sx0,
                          g.IH, g.IW, g.iw_magic, g.cvb_log2);
#endif
        __syncthreads();
        if (!cvalid) continue;
#if DFD_ABLATE == 1
        continue;
#endif
#pragma unroll 1
        for (int q = lane; q < g.NQ; q += PL) {
            const int qy = (int)(((unsigned)q * g.qw_magic) >> 20), qx = q - qy * g.QW;
            const int h = h0 + qy, wq = w0 + 4 * qx;
            if (h >= g.H || wq >= g.W) continue;
            f2 acc[4][N2];
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int j = 0; j < N2; ++j) acc[o][j] = (f2){0.f, 0.f};
            // the epilogue's xin vectors are requested now so that HBM latency hides under the taps
            const long off = (((long)n * g.H + h) * g.W + wq) * g.C + c0;
            uint4 xr[4];
            if constexpr (EPI) {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    xr[o] = (wq + o < g.W) ? *reinterpret_cast<const uint4*>(xin + off + (long)o * g.C) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                int r;
                if constexpr (S == 1) {
                    r = qy + (K - 1 - kh);                     // h + pt - kh - sy0
                } else {
                    const int num = h + g.pt - kh;
                    if (num & 1) continue;
                    r = (num >> 1) - sy0;
                }
                const uint4* row = tile + (((r * g.IW) + qx * (S == 1 ? 4 : 2)) << g.cvb_log2) + vl;
                f2 wv[K][N2];
#pragma unroll
                for (int kw = 0; kw < K; ++kw) lds_row<N2>(wl + (kh * K + kw) * cvbV + vl * V, wv[kw]);
#pragma unroll
                for (int c = 0; c < NCOL; ++c) {
                    f2 xc[N2];
                    unpack2(row[c << g.cvb_log2], xc);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        // S=1: column c holds dy[.. w+o+pl-kw] with kw = o + (K-1) - c
                        // S=2: column c holds it with kw = o + E - 2c
                        const int kw = (S == 1) ? o + (K - 1) - c : o + E - 2 * c;
                        if (kw >= 0 && kw < K) {
#pragma unroll
                            for (int j = 0; j < N2; ++j) acc[o][j] = __builtin_elementwise_fma(xc[j], wv[kw][j], acc[o][j]);
                        }
                    }
                }
            }
            f2 scv[N2], shv[N2];                     // BN(scale, shift) of the producer: two 16-byte LDS reads each
            if constexpr (EPI) {
                lds_row<N2>(cf + 3 * cvbV + vl * V, scv);
                lds_row<N2>(cf + 4 * cvbV + vl * V, shv);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (wq + o < g.W) {
                    if constexpr (EPI) {
                        f2 xv[N2];
                        unpack2(xr[o], xv);
#pragma unroll
                        for (int j = 0; j < N2; ++j) {
                            const f2 z = __builtin_elementwise_fma(scv[j], xv[j], shv[j]);
                            f2 gr;
                            if constexpr (ACT == DFD_ACT_SILU) {
                                // act_grad<SILU> on the pair: only exp and rcp are per element, the rest is packed math
                                // (same operations in the same order as the scalar form: identical bits)
                                const f2 e = (f2){__expf(-z.x), __expf(-z.y)};
                                const f2 d1 = e + (f2){1.f, 1.f};
                                const f2 sg = (f2){__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)};
                                gr = sg * ((f2){1.f, 1.f} + z * ((f2){1.f, 1.f} - sg));
                            } else {
                                gr = act_grad2<ACT>(z);
                            }
                            const f2 d = round2<T>(acc[o][j] * gr);
                            acc[o][j] = d;
                            s1[j] += d;
                            s2[j] = __builtin_elementwise_fma(d, xv[j], s2[j]);      // sum dz*x ; xhat applied at the end
                        }
                    }
                    *reinterpret_cast<uint4*>(dzin + off + (long)o * g.C) = pack2(acc[o]);
                }
            }
        }
    }
    if constexpr (EPI) {
        __syncthreads();
        float acc2[2 * V];
        // sum dz*xhat = rstd * (sum dz*x - mean * sum dz)
#pragma unroll
        for (int j = 0; j < N2; ++j) {
            float m0 = 0.f, m1 = 0.f, r0 = 1.f, r1 = 1.f;
            if (cvalid) {
                m0 = in_bnstate[2 * g.C + c0 + 2 * j]; m1 = in_bnstate[2 * g.C + c0 + 2 * j + 1];
                r0 = in_bnstate[3 * g.C + c0 + 2 * j]; r1 = in_bnstate[3 * g.C + c0 + 2 * j + 1];
            }
            acc2[2 * j] = s1[j].x; acc2[2 * j + 1] = s1[j].y;
            acc2[V + 2 * j] = r0 * (s2[j].x - m0 * s1[j].x);
            acc2[V + 2 * j + 1] = r1 * (s2[j].y - m1 * s1[j].y);
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

template <typename T>
static int dw_bwd_data_q_t(const void* dz, const void* y, const float* coef, const float* w, const void* xin,
                           const float* in_bnstate, int in_act, void* dzin, const dfd_dwconv_shape* s, float* partials,
                           int pcap, int* nparts, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwQGeom g; int tile_bytes;
    const int kk = s ? s->k * s->k : 0;
    const DwqOcc occ = dwq_occupancy(1, s);
    if (!dfd_dwq_geom(s, V, 16, true, (size_t)(kk + 5) * 16 * V * 4, 0, 1, &g, &tile_bytes, 1, occ.lds_budget)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    // chunk-workgroups share cache lines unless both the pixel stride and the chunk width are whole lines
    g.remap = (s->stride == 1 && nchunks > 1 && ((s->C * (int)sizeof(T)) % 128 != 0 || (cvb * 16) % 128 != 0)) ? 1 : 0;
    const bool epi = xin != nullptr, hc = coef != nullptr;
    if (epi && (!in_bnstate || !partials || !nparts || pcap < 1)) return DFD_EINVAL;
    if (hc && !y) return DFD_EINVAL;
    int cap = epi ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    int gy = occ.grid / nchunks;
    if (gy < dfd_tune_get(DFD_TUNE_DWQ_GRID_MIN)) gy = dfd_tune_get(DFD_TUNE_DWQ_GRID_MIN);
    if (gy > cap) gy = cap;
    if (gy > g.nwork) gy = g.nwork;
    if (epi) *nparts = gy;
    size_t lds = (size_t)tile_bytes + (size_t)(kk + 5) * cvb * V * 4;
    const size_t red = (size_t)DFD_THREADS * 2 * V * 4;
    if (lds < red) lds = red;
    dim3 grid(nchunks, gy);
    const int plp = s->pad_left & 1;
    if (!epi) in_act = DFD_ACT_NONE;
    const bool wide3 = occ.lds_budget > 40 * 1024;
#define LAUNCH_BD(PLP, COEF, EPI) \
    do { \
        if constexpr (sizeof(T) == 2 && K == 5 && S == 1) { \
            if (wide3) { \
                hipLaunchKernelGGL((k_dw_bwd_data_q<T, K, S, PLP, ACT, COEF, EPI, 3>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, \
                                   (const T*)y, coef, w, (const T*)xin, in_bnstate, (T*)dzin, g, partials, tile_bytes); \
                break; \
            } \
        } \
        hipLaunchKernelGGL((k_dw_bwd_data_q<T, K, S, PLP, ACT, COEF, EPI>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, \
                           (const T*)y, coef, w, (const T*)xin, in_bnstate, (T*)dzin, g, partials, tile_bytes); \
    } while (0)
#define LAUNCH_BD_P(COEF, EPI) \
    if (S == 2 && plp) LAUNCH_BD(1, COEF, EPI); else LAUNCH_BD(0, COEF, EPI)
    DISPATCH_KS(s->k, s->stride, {
        if (epi) {
            DISPATCH_ACT_DW(in_act, { if (hc) { LAUNCH_BD_P(true, true); } else { LAUNCH_BD_P(false, true); } });
        } else {
            constexpr int ACT = DFD_ACT_NONE;
            if (hc) { LAUNCH_BD_P(true, false); } else { LAUNCH_BD_P(false, false); }
        }
    });
#undef LAUNCH_BD_P
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
        return dw_bwd_data_q_t<bf16>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, s, partials, pcap, nparts, st);
    if (dtype == DFD_F32)
        return dw_bwd_data_q_t<float>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, s, partials, pcap, nparts, st);
    return DFD_EINVAL;
}

// ===========================================================================
// weight gradient:  dw[c][kh][kw] = sum_{n,oy,ox} dy[n][oy][ox][c] * a[n][oy*S-pt+kh][ox*S-pl+kw][c]
// with a = act(scale*xin+shift) (or xin) and dy = affine2(dz, y).
// Both tiles live in LDS.  Pixel lanes are split into K kernel-row roles: lane (kh, run)
// walks quads of four adjacent outputs, unpacks the four dy vectors and the 3*S+K input
// vectors of its kernel row once each, and accumulates its K taps in registers over the
// whole persistent loop.  partial layout: [workgroup][C][K*K].
// ===========================================================================
template <typename T, int K, int S, int ACT, bool PRO, bool COEF, int WPC = 4>      // WPC: see k_dw_bwd_data_q (here it only lifts the register cap: the 4-per-CU 5x5 instances spill 32-40 bytes)
__global__ void __launch_bounds__(DFD_THREADS, WPC)
k_dw_bwd_weight_q(const T* __restrict__ dz, const T* __restrict__ yraw, const float* __restrict__ coef,
                  const T* __restrict__ xin, const float* __restrict__ in_bnstate, DwQGeom g,
                  float* __restrict__ partials, int tile_bytes) {
    constexpr int V = Vec<T>::N, N2 = V / 2;
    constexpr int NCOL = 3 * S + K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2, cvbV = cvb * V;
    const int TW = 4 * g.QW;
    uint4* tile = reinterpret_cast<uint4*>(smem);                              // activated input, with halo
    uint4* dyt = reinterpret_cast<uint4*>(smem + tile_bytes);                  // dy, [TH][TW][cvb]
    float* cf = reinterpret_cast<float*>(smem + tile_bytes + (size_t)g.TH * TW * cvb * 16);   // ka, kb, kc [3][cvbV]
    const int t = threadIdx.x, vl = t & (cvb - 1), lane = t >> g.cvb_log2;
    int bx, by;
    dwq_block(bx, by, g.remap);
    const int vglob = bx * cvb + vl;
    const bool cvalid = vglob < g.CV;
    const int c0 = vglob * V, chunk_c0 = bx * cvbV;
    const int NPR = PL / K;                          // quad-walking lanes per kernel row
    const int kh = lane % K, prl = lane / K;
    const bool lane_on = prl < NPR;

    for (int i = t; i < 5 * cvbV; i += DFD_THREADS) {          // ka, kb, kc, scale, shift
        const int which = i / cvbV, cc = i - which * cvbV, c = chunk_c0 + cc;
        float v = (which == 0 || which == 3) ? 1.f : 0.f;
        if (c < g.C) {
            if (which < 3) { if (COEF) v = coef[which * g.C + c]; }
            else if (PRO) v = in_bnstate[(which - 3) * g.C + c];
        }
        cf[i] = v;
    }
    f2 accw[K][N2];
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int j = 0; j < N2; ++j) accw[kw][j] = (f2){0.f, 0.f};

    const int tiles = g.tiles_y * g.tiles_x;
    const unsigned tw_magic = ((1u << 20) + TW - 1) / TW;
    for (int work = by; work < g.nwork; work += gridDim.y) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int oy0 = ty * g.TH, ox0 = tx * TW;
        __syncthreads();
        {
            f2 sc[N2], sh[N2];                       // short-lived: the coefficients live in LDS
            lds_row<N2>(cf + 3 * cvbV + vl * V, sc);
            lds_row<N2>(cf + 4 * cvbV + vl * V, sh);
            stage_q<T, ACT, PRO, (K == 3 ? 8 : 4)>(tile, xin, sc, sh, (long)n * g.H * g.W * g.C, g.H, g.W, g.C, c0, cvalid, oy0 * S - g.pt,
                                 ox0 * S - g.pl, g.IH, g.IW, g.iw_magic, g.cvb_log2);
        }
        stage_dy<T, COEF, StageDepth<K, S>::DY>(dyt, dz, yraw, cf, cvbV, vl, (long)n * g.Ho * g.Wo * g.C, g.Ho, g.Wo, g.C, c0, cvalid, oy0, ox0,
                          g.TH, TW, tw_magic, g.cvb_log2);
        __syncthreads();
        if (!cvalid || !lane_on) continue;
#pragma unroll 1
        for (int q = prl; q < g.NQ; q += NPR) {
            const int qy = (int)(((unsigned)q * g.qw_magic) >> 20), qx = q - qy * g.QW;
            if (oy0 + qy >= g.Ho || ox0 + 4 * qx >= g.Wo) continue;      // dy is zero there anyway
            f2 dv[4][N2];
            const uint4* drow = dyt + (((qy * TW) + 4 * qx) << g.cvb_log2) + vl;
#pragma unroll
            for (int o = 0; o < 4; ++o) unpack2(drow[o << g.cvb_log2], dv[o]);
            const uint4* row = tile + ((((qy * S + kh) * g.IW) + qx * 4 * S) << g.cvb_log2) + vl;
#pragma unroll
            for (int c = 0; c < NCOL; ++c) {
                f2 xc[N2];
                unpack2(row[c << g.cvb_log2], xc);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int kw = c - S * o;
                    if (kw >= 0 && kw < K) {
#pragma unroll
                        for (int j = 0; j < N2; ++j) accw[kw][j] = __builtin_elementwise_fma(dv[o][j], xc[j], accw[kw][j]);
                    }
                }
            }
        }
    }
    // combine the run lanes of each kernel row through LDS: red[prl][kh][kw][vl][V]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane_on) {
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                float* d = red + ((((prl * K + kh) * K + kw) * cvb + vl) * V + 2 * j);
                d[0] = accw[kw][j].x;
                d[1] = accw[kw][j].y;
            }
    }
    __syncthreads();
    // every run lane takes its share of the K * V (tap column, channel) sums of its kernel row; each sum runs over the
    // run lanes in ascending order (fixed order), and no thread is left summing K * V * NPR values alone
    if (lane_on && cvalid) {
        float* p = partials + (long)by * g.C * K * K;
        for (int e = prl; e < K * V; e += NPR) {
            const int kw = e / V, j = e - kw * V;
            float s = 0.f;
            for (int r = 0; r < NPR; ++r) s += red[(((r * K + kh) * K + kw) * cvb + vl) * V + j];
            p[(long)(c0 + j) * K * K + kh * K + kw] = s;
        }
    }
}

static bool dw_wgrad_q_geom(const dfd_dwconv_shape* s, int vec, DwQGeom* g, int* tile_bytes) {
    const int max_cvb = (s && s->k == 5) ? 8 : 16;          // needs 256/cvb >= K lanes per run
    if (!dfd_dwq_geom(s, vec, max_cvb, false, (size_t)5 * 16 * vec * 4, 16, s ? s->k : 1, g, tile_bytes, 1, dwq_occupancy(2, s).lds_budget)) return false;
    const int cvb = 1 << g->cvb_log2, nchunks = (g->CV + cvb - 1) / cvb, esz = vec == 8 ? 2 : 4;
    g->remap = (nchunks > 1 && ((s->C * esz) % 128 != 0 || (cvb * 16) % 128 != 0)) ? 1 : 0;
    return true;
}

static int dw_wgrad_q_parts(const DwQGeom& g, int k, int nchunks, int grid) {
    int want = grid / (nchunks > 0 ? nchunks : 1);
    if (want < 32) want = 32;
    const long per = (long)g.C * k * k * 4;
    long cap = (16l << 20) / per;
    if (cap < 32) cap = 32;
    if (want > cap) want = (int)cap;
    if (want > DFD_MAX_PARTIALS) want = DFD_MAX_PARTIALS;
    if (want > g.nwork) want = g.nwork;
    return want < 1 ? 1 : want;
}

extern "C" size_t dfd_dwconv_bwd_weight_ws(const dfd_dwconv_shape* s) {
    if (!s || s->C <= 0 || s->k <= 0) return 0;
    const size_t per = (size_t)s->C * s->k * s->k * 4;
    const size_t a = (size_t)16 << 20, b = per * 32;
    return (a > b ? a : b) + per * 36;      // partial slab + second-stage rows of the reduction
}

template <typename T>
static int dw_bwd_weight_q_t(const void* dz, const void* y, const float* coef, const void* xin, const float* in_bnstate,
                             int in_act, float* dw, const dfd_dwconv_shape* s, int accumulate, float* ws,
                             size_t ws_bytes, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    DwQGeom g; int tile_bytes;
    if (!dw_wgrad_q_geom(s, V, &g, &tile_bytes)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb, KK = s->k * s->k;
    const int P = dw_wgrad_q_parts(g, s->k, nchunks, dwq_occupancy(2, s).grid);
    if ((size_t)(P + P / 32 + 2) * g.C * KK * 4 > ws_bytes) return DFD_EWORKSPACE;
    size_t lds = (size_t)tile_bytes + (size_t)g.TH * 4 * g.QW * cvb * 16 + (size_t)5 * cvb * V * 4;
    const size_t red = (size_t)((DFD_THREADS / cvb) / s->k) * KK * cvb * V * 4;
    if (lds < red) lds = red;
    dim3 grid(nchunks, P);
    const bool pro = in_bnstate != nullptr, hc = coef != nullptr;
    if (hc && !y) return DFD_EINVAL;
    if (!pro) in_act = DFD_ACT_NONE;
    const bool wide3 = dwq_occupancy(2, s).lds_budget > 40 * 1024;
#define LAUNCH_BW(PRO, COEF) \
    do { \
        if constexpr (sizeof(T) == 2 && K == 5 && S == 1) { \
            if (wide3) { \
                hipLaunchKernelGGL((k_dw_bwd_weight_q<T, K, S, ACT, PRO, COEF, 3>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, \
                                   (const T*)y, coef, (const T*)xin, in_bnstate, g, ws, tile_bytes); \
                break; \
            } \
        } \
        hipLaunchKernelGGL((k_dw_bwd_weight_q<T, K, S, ACT, PRO, COEF>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, \
                           (const T*)y, coef, (const T*)xin, in_bnstate, g, ws, tile_bytes); \
    } while (0)
    DISPATCH_KS(s->k, s->stride, {
        if (pro) {
            DISPATCH_ACT_DW(in_act, { if (hc) LAUNCH_BW(true, true); else LAUNCH_BW(true, false); });
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
        return dw_bwd_weight_q_t<bf16>(dz, y, coef, xin, in_bnstate, in_act, dw, s, accumulate, ws, ws_bytes, st);
    if (dtype == DFD_F32)
        return dw_bwd_weight_q_t<float>(dz, y, coef, xin, in_bnstate, in_act, dw, s, accumulate, ws, ws_bytes, st);
    return DFD_EINVAL;
}
