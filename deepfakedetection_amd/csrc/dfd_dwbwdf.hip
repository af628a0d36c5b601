// dfd_dwbwdf.hip — depthwise 3x3 stride-1 convolution: data gradient AND weight gradient in ONE kernel.
//
// The two backward kernels of dfd_dwbwd.hip each stage the same operands — dy = a*dz + b*y + c (the BatchNorm-backward map of
// the conv's own BN, two tensors) and the producer's activated output act(scale*xin + shift) — so a block's backward read
// (dz, y, xin) twice from HBM: 6.6 + 5.4 GB of the 47 GB an EfficientNet-B0 step moves (profiles/r02_summary.md: traffic 1.8x
// the algorithmic bytes on both), and EfficientFormerV2's 21 ConvMlp blocks pay the same.  Here one workgroup owns a tile of
// INPUT pixels and the OUTPUT pixels with the same coordinates (stride 1):
//      stage   dy tile   = outputs that touch the owned inputs          (owned rows/cols + K-1 halo, BN-backward map applied)
//              act tile  = inputs the owned outputs read                (owned rows/cols + K-1 halo, BN + activation applied)
//      phase D da[h][w]  = sum_{kh,kw} dy[h+pt-kh][w+pl-kw] W[kh][kw];  dzin = da * act'(scale*xin + shift);  sums for BN backward
//              (quad walk of k_dw_bwd_data_q; the raw xin of the owned pixels is re-read from L2 for the derivative)
//      phase W dW[kh][kw] += sum_{owned outputs} dy[oy][ox] * act[oy-pt+kh][ox-pl+kw]      (K kernel-row roles of k_dw_bwd_weight_q)
// Every tensor crosses HBM once (plus halo).  Per element the arithmetic is that of the two separate kernels (same staging
// maps, same tap order per output: dzin is bit-identical); the sums over pixels (BatchNorm partials, weight gradient) follow
// THIS kernel's tiling and workgroup slots, in a fixed order: reproducible run to run.
// Reference call site: the backward of the third-party MBConv / ConvMlp blocks (trainers/efficientnet.py:302,
// trainers/efficientformer_v2.py:246).  K = 3, S = 1 only (EfficientNet-B0 blocks 2, 6, 7, 15; every EfficientFormerV2 ConvMlp);
// other shapes keep the two-kernel path (DFD_EUNSUPPORTED).
#include "dfd_dwq.h"

#ifndef DWF_WGS
#define DWF_WGS 3
#endif
#ifndef DWF_LDS
#define DWF_LDS (50 * 1024)
#endif
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS, DWF_WGS)
k_dw_bwd_fused_q(const T* __restrict__ dz, const T* __restrict__ yraw, const float* __restrict__ coef,
                 const float* __restrict__ w, const T* __restrict__ xin, const float* __restrict__ in_bnstate,
                 T* __restrict__ dzin, DwQGeom g, float* __restrict__ partials, float* __restrict__ wparts, int tile_bytes) {
    constexpr int K = 3, V = Vec<T>::N, N2 = V / 2;
    constexpr int NCOL_D = K + 3;                    // dy columns a quad of 4 inputs reads (stride 1)
    constexpr int NCOL_W = 3 + K;                    // input columns a quad of 4 outputs reads
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* dyt = reinterpret_cast<uint4*>(smem);                     // dy, halo extent [IH][IW][cvb]
    uint4* act = reinterpret_cast<uint4*>(smem + tile_bytes);        // activated input, halo extent
    const int cvb = 1 << g.cvb_log2, PL = DFD_THREADS >> g.cvb_log2, cvbV = cvb * V;
    float* wl = reinterpret_cast<float*>(smem + 2 * (size_t)tile_bytes);      // [tap][cvb*V]
    float* cf = wl + K * K * cvbV;                                           // ka, kb, kc, scale, shift : [5][cvb*V]
    const int t = threadIdx.x, vl = t & (cvb - 1), lane = t >> g.cvb_log2;
    int bx, by;
    dwq_block(bx, by, g.remap);
    const int vglob = bx * cvb + vl;
    const bool cvalid = vglob < g.CV;
    const int c0 = vglob * V;
    const int chunk_c0 = bx * cvbV;
    {
        constexpr int NW = (K * K * 16 * V + DFD_THREADS - 1) / DFD_THREADS;
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
        if (c < g.C) v = which < 3 ? coef[which * g.C + c] : in_bnstate[(which - 3) * g.C + c];
        cf[i] = v;
    }
    f2 s1[N2], s2[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) { s1[j] = (f2){0.f, 0.f}; s2[j] = (f2){0.f, 0.f}; }
    // weight-gradient roles: lane (kh, run) as in k_dw_bwd_weight_q
    const int NPR = PL / K;
    const int kh_w = lane % K, prl = lane / K;
    const bool lane_on = prl < NPR;
    f2 accw[K][N2];
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
#pragma unroll
        for (int j = 0; j < N2; ++j) accw[kw][j] = (f2){0.f, 0.f};

    const int tiles = g.tiles_y * g.tiles_x, TW = 4 * g.QW;
    for (int work = by; work < g.nwork; work += gridDim.y) {
        const int n = work / tiles, tr = work - n * tiles;
        const int ty = tr / g.tiles_x, tx = tr - ty * g.tiles_x;
        const int h0 = ty * g.TH, w0 = tx * TW;
        const int sy0 = h0 + g.pt - (K - 1), sx0 = w0 + g.pl - (K - 1);      // first staged dy row / column
        const int ay0 = h0 - g.pt, ax0 = w0 - g.pl;                          // first staged input row / column
        __syncthreads();
        stage_dy<T, true, StageDepth<K, 1>::DY>(dyt, dz, yraw, cf, cvbV, vl, (long)n * g.Ho * g.Wo * g.C, g.Ho, g.Wo, g.C, c0, cvalid, sy0, sx0,
                                                g.IH, g.IW, g.iw_magic, g.cvb_log2);
        {
            f2 sc[N2], sh[N2];
            lds_row<N2>(cf + 3 * cvbV + vl * V, sc);
            lds_row<N2>(cf + 4 * cvbV + vl * V, sh);
            stage_q<T, ACT, true, 8>(act, xin, sc, sh, (long)n * g.H * g.W * g.C, g.H, g.W, g.C, c0, cvalid, ay0, ax0, g.IH, g.IW,
                                     g.iw_magic, g.cvb_log2);
        }
        __syncthreads();
        if (!cvalid) continue;
        // ---------------- phase D: data gradient of the owned input pixels (k_dw_bwd_data_q, S = 1, epilogue on)
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
            const long off = (((long)n * g.H + h) * g.W + wq) * g.C + c0;
            uint4 xr[4];
#pragma unroll
            for (int o = 0; o < 4; ++o)
                xr[o] = (wq + o < g.W) ? *reinterpret_cast<const uint4*>(xin + off + (long)o * g.C) : make_uint4(0, 0, 0, 0);
#pragma unroll 1
            for (int kh = 0; kh < K; ++kh) {
                const int r = qy + (K - 1 - kh);
                const uint4* row = dyt + (((r * g.IW) + qx * 4) << g.cvb_log2) + vl;
                f2 wv[K][N2];
#pragma unroll
                for (int kw = 0; kw < K; ++kw) lds_row<N2>(wl + (kh * K + kw) * cvbV + vl * V, wv[kw]);
#pragma unroll
                for (int c = 0; c < NCOL_D; ++c) {
                    f2 xc[N2];
                    unpack2(row[c << g.cvb_log2], xc);
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int kw = o + (K - 1) - c;
                        if (kw >= 0 && kw < K) {
#pragma unroll
                            for (int j = 0; j < N2; ++j) acc[o][j] = __builtin_elementwise_fma(xc[j], wv[kw][j], acc[o][j]);
                        }
                    }
                }
            }
            f2 scv[N2], shv[N2];
            lds_row<N2>(cf + 3 * cvbV + vl * V, scv);
            lds_row<N2>(cf + 4 * cvbV + vl * V, shv);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (wq + o < g.W) {
                    f2 xv[N2];
                    unpack2(xr[o], xv);
#pragma unroll
                    for (int j = 0; j < N2; ++j) {
                        const f2 z = __builtin_elementwise_fma(scv[j], xv[j], shv[j]);
                        f2 gr;
                        if constexpr (ACT == DFD_ACT_SILU) {
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
                        s2[j] = __builtin_elementwise_fma(d, xv[j], s2[j]);
                    }
                    *reinterpret_cast<uint4*>(dzin + off + (long)o * g.C) = pack2(acc[o]);
                }
            }
        }
        // ---------------- phase W: weight gradient of the owned output pixels (k_dw_bwd_weight_q)
        if (!lane_on) continue;
#pragma unroll 1
        for (int q = prl; q < g.NQ; q += NPR) {
            const int qy = (int)(((unsigned)q * g.qw_magic) >> 20), qx = q - qy * g.QW;
            if (h0 + qy >= g.Ho || w0 + 4 * qx >= g.Wo) continue;
            f2 dv[4][N2];
            // owned output (h0 + qy, w0 + 4 qx + o) sits at row qy + (K-1) - pt, column 4 qx + o + (K-1) - pl of the dy tile;
            // columns beyond Wo inside the tile were staged as zeros
            const uint4* drow = dyt + ((((qy + (K - 1) - g.pt) * g.IW) + 4 * qx + (K - 1) - g.pl) << g.cvb_log2) + vl;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (w0 + 4 * qx + o < g.Wo) unpack2(drow[o << g.cvb_log2], dv[o]);
                else {
#pragma unroll
                    for (int j = 0; j < N2; ++j) dv[o][j] = (f2){0.f, 0.f};
                }
            }
            const uint4* row = act + ((((qy + kh_w) * g.IW) + qx * 4) << g.cvb_log2) + vl;
#pragma unroll
            for (int c = 0; c < NCOL_W; ++c) {
                f2 xc[N2];
                unpack2(row[c << g.cvb_log2], xc);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int kw = c - o;
                    if (kw >= 0 && kw < K) {
#pragma unroll
                        for (int j = 0; j < N2; ++j) accw[kw][j] = __builtin_elementwise_fma(dv[o][j], xc[j], accw[kw][j]);
                    }
                }
            }
        }
    }
    // ---- BN-backward partial sums of dzin (as k_dw_bwd_data_q)
    __syncthreads();
    {
        float acc2[2 * V];
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
    // ---- weight-gradient partial row of this workgroup slot (as k_dw_bwd_weight_q): red[prl][kh][kw][vl][V]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    if (lane_on) {
#pragma unroll
        for (int kw = 0; kw < K; ++kw)
#pragma unroll
            for (int j = 0; j < N2; ++j) {
                float* d = red + ((((prl * K + kh_w) * K + kw) * cvb + vl) * V + 2 * j);
                d[0] = accw[kw][j].x;
                d[1] = accw[kw][j].y;
            }
    }
    __syncthreads();
    if (lane_on && cvalid) {
        float* p = wparts + (long)by * g.C * K * K;
        for (int e = prl; e < K * V; e += NPR) {
            const int kw = e / V, j = e - kw * V;
            float s = 0.f;
            for (int r = 0; r < NPR; ++r) s += red[(((r * K + kh_w) * K + kw) * cvb + vl) * V + j];
            p[(long)(c0 + j) * K * K + kh_w * K + kw] = s;
        }
    }
}

template <typename T>
static int dw_bwd_fused_t(const void* dz, const void* y, const float* coef, const float* w, const void* xin,
                          const float* in_bnstate, int in_act, void* dzin, float* dw, const dfd_dwconv_shape* s, float* partials,
                          int pcap, int* nparts, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int V = Vec<T>::N, K = 3;
    DwQGeom g; int tile_bytes;
    const int kk = K * K;
    // two halo tiles; three workgroups per CU (registers): 160 KB / 3 of LDS each
    if (!dfd_dwq_geom(s, V, 16, true, (size_t)(kk + 5) * 16 * V * 4, 0, 1, &g, &tile_bytes, 2, DWF_LDS)) return DFD_EINVAL;
    const int cvb = 1 << g.cvb_log2, nchunks = (g.CV + cvb - 1) / cvb;
    if ((DFD_THREADS / cvb) < K) return DFD_EUNSUPPORTED;
    g.remap = (nchunks > 1 && ((s->C * (int)sizeof(T)) % 128 != 0 || (cvb * 16) % 128 != 0)) ? 1 : 0;
    int cap = pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS;
    int gy = DFD_DW_GRID / nchunks;
    if (gy < 64) gy = 64;
    if (gy > cap) gy = cap;
    {   // the weight-gradient slab (one row per workgroup slot) must fit the caller's workspace: <= max(16 MB, 32 rows)
        const long per = (long)g.C * kk * 4;
        long rows = (16l << 20) / per;
        if (rows < 32) rows = 32;
        if (gy > rows) gy = (int)rows;
    }
    if (gy > g.nwork) gy = g.nwork;
    // the weight-gradient slab: gy rows of C*K*K floats + the second reduction stage behind them
    if ((size_t)(gy + gy / 32 + 2) * g.C * kk * 4 > ws_bytes) return DFD_EWORKSPACE;
    *nparts = gy;
    size_t lds = 2 * (size_t)tile_bytes + (size_t)(kk + 5) * cvb * V * 4;
    const size_t red1 = (size_t)DFD_THREADS * 2 * V * 4;
    const size_t red2 = (size_t)((DFD_THREADS / cvb) / K) * kk * cvb * V * 4;
    if (lds < red1) lds = red1;
    if (lds < red2) lds = red2;
    dim3 grid(nchunks, gy);
#define LAUNCH_BF(ACTV) \
    hipLaunchKernelGGL((k_dw_bwd_fused_q<T, ACTV>), grid, dim3(DFD_THREADS), lds, st, (const T*)dz, (const T*)y, coef, w, \
                       (const T*)xin, in_bnstate, (T*)dzin, g, partials, ws, tile_bytes)
    switch (in_act) {
        case DFD_ACT_SILU: LAUNCH_BF(DFD_ACT_SILU); break;
        case DFD_ACT_GELU: LAUNCH_BF(DFD_ACT_GELU); break;
        case DFD_ACT_RELU: LAUNCH_BF(DFD_ACT_RELU); break;
        case DFD_ACT_NONE: LAUNCH_BF(DFD_ACT_NONE); break;
        default: return DFD_EUNSUPPORTED;
    }
#undef LAUNCH_BF
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, gy, (long)g.C * kk, dw, accumulate, st);
}

extern "C" int dfd_dwconv_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const float* w, const void* xin,
                                    const float* in_bnstate, int in_act, void* dzin, float* dw, const dfd_dwconv_shape* s,
                                    float* partials, int pcap, int* nparts, int accumulate, float* ws, size_t ws_bytes,
                                    dfd_stream stream) {
    if (!dz || !y || !coef || !w || !xin || !in_bnstate || !dzin || !dw || !s || !partials || !nparts || pcap < 1 || !ws) return DFD_EINVAL;
    if (s->k != 3 || s->stride != 1) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16)
        return dw_bwd_fused_t<bf16>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, dw, s, partials, pcap, nparts, accumulate, ws, ws_bytes, st);
    if (dtype == DFD_F32)
        return dw_bwd_fused_t<float>(dz, y, coef, w, xin, in_bnstate, in_act, dzin, dw, s, partials, pcap, nparts, accumulate, ws, ws_bytes, st);
    return DFD_EINVAL;
}
