// dfd_pwntw.hip — 1x1 convolution, NT form, for layers whose weight panel fits LDS:
//
//     out[M][Nout] = P(a)[M][K] * w[Nout][K]^T          (K, Nout small; M = N*H*W large)
//
// These are the EfficientNet layers that carry most of the bytes (blocks 0-5: 3.2M..200K rows,
// 16..240 channels).  They are HBM-bound with almost no MFMA work, so what limits them is how many
// independent memory streams a CU keeps in flight.  Design (one wave = one stream, no workgroup
// barrier in the main loop):
//   * the BN x K weight panel of the workgroup is staged into LDS ONCE (zero padded, row stride
//     nk*128+16 bytes so the 16 rows of a fragment read land on different 16-byte bank groups);
//   * each wave owns 32 rows of a 128-row M tile and ALL BN output channels.  The activation operand
//     never touches LDS: a lane's MFMA fragment is 16 contiguous bytes of one row (8 bf16 / 4 f32 of
//     consecutive k), so it is loaded straight from global memory in fragment layout, two stages
//     ahead, and the producer's BN/activation/gate (or the BN-backward affine map) is applied to the
//     fragment registers;
//   * weights are the MFMA "A" operand: a lane's 4 accumulators are 4 consecutive output channels of
//     one row.  The 32 x BN result is transposed through a wave-private LDS patch (no barrier) and
//     leaves as 16-byte row-major stores; BN statistics are taken from the rounded values on the way;
//   * persistent over M tiles: (sum, sumsq) stay in registers, ONE partial row per workgroup.
#include "dfd_pw.h"

// KSN = 32-byte... k sub-steps per stage (1: K <= 4 chunks, a stage is 64 rows x 64 B per wave;
//       2: a stage is 32 rows x 128 B per wave).  Either way a lane has 4 x 16 B in flight per stage.
// NCH = column chunks a workgroup may carry WITH statistics (one accumulator set per chunk and lane)
// EACT >= 0: eval epilogue act(scale * y + shift) with this layer's own BatchNorm coefficients (see k_pw_nt)
template <typename T, int BN, int KSN, int NCH, int PRO, int ACT, bool RES, bool STATS, int EACT = -1>
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_pw_ntw(const T* __restrict__ a, ProArgs pa, const T* __restrict__ w, T* __restrict__ out, const T* __restrict__ res,
         int M, int K, int Nout, int m_tiles, int n_tiles, int gx, int wstride, int eoff, int coff, int nchunks,
         float* __restrict__ partials, const float* __restrict__ ebn = nullptr) {
    constexpr int E = El<T>::EPC;
    constexpr int BK = KSN * 4 * E;         // k elements per stage
    constexpr int JN = 4 / KSN;             // 16-row fragments per wave and stage
    constexpr int WR = 16 * JN;             // rows per wave and stage
    constexpr int BM = 4 * WR;              // rows per workgroup tile
    constexpr int NT = BN / 16;
    constexpr int OROW = BN * (int)sizeof(T) + 16;
    constexpr int CPR = BN / E;             // 16-byte chunks per output row
    constexpr int RL = 64 / CPR;            // rows per epilogue pass of one wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int total = gx * n_tiles;
    int lin = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int pb = lin / n_tiles, nt = lin - pb * n_tiles;
    const int n0 = nt * BN * nchunks;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = (K + BK - 1) / BK;
    const T* a2 = reinterpret_cast<const T*>(pa.a2);

    // ---- weight panel -> LDS (once)
    {
        const int chunks = nk * KSN * 4;
        // eight 16-byte loads per lane in flight: the panel is up to 16 passes of the workgroup, and a load-store loop
        // pays one memory round trip per pass before the first MFMA of every workgroup
        const int total = BN * nchunks * chunks;
        for (int base = t; base < total; base += DFD_THREADS * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * DFD_THREADS;
                v[u] = make_uint4(0, 0, 0, 0);
                if (idx < total) {
                    const int n = idx / chunks, c = idx - n * chunks;
                    if (n0 + n < Nout && c * E < K) v[u] = *reinterpret_cast<const uint4*>(w + (long)(n0 + n) * K + c * E);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = base + u * DFD_THREADS;
                if (idx < total) {
                    const int n = idx / chunks, c = idx - n * chunks;
                    *reinterpret_cast<uint4*>(smem + n * wstride + c * 16) = v[u];
                }
            }
        }
        // per-channel prologue coefficients, zero padded to the K tiles: rows of nk*BK floats
        if constexpr (PRO != DFD_PRO_NONE) {
            constexpr int NROW = PRO == DFD_PRO_AFFINE2 ? 3 : 2;
            float* cl = reinterpret_cast<float*>(smem + coff);
            const int kp = nk * BK;
            for (int idx = t; idx < NROW * kp; idx += DFD_THREADS) {
                const int r = idx / kp, k = idx - r * kp;
                cl[idx] = k < K ? pa.coef[r * K + k] : 0.f;
            }
        }
    }
    __syncthreads();

    const int ec = lane % CPR, er = lane / CPR;
    float s1[NCH][E], s2[NCH][E];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < E; ++j) { s1[c][j] = 0.f; s2[c][j] = 0.f; }

    struct Regs { uint4 a[JN][KSN], a2[JN][KSN]; };
    Regs R0, R1;
    float gt[JN][KSN][E];

    auto g_load = [&](Regs& R, int mt, int kt) {
        const int rbase = mt * BM + wave * WR + frow;
#pragma unroll
        for (int j = 0; j < JN; ++j)
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                // Unconditional loads from a row / chunk clamped into the operand (as in k_pw_nt): inside divergent branches the
                // compiler cannot count the loads in flight and waited vmcnt(0) right behind the request — the two register sets
                // pipelined nothing.  Rows past M are never stored; chunks past K become zeros where the fragment is formed
                // (`prologue`), a stage later — a select on the load's own destination would wait for the load again.
                const int row = rbase + j * 16, kk = kt * BK + (ks * 4 + fk) * E;
                const long off = (long)(row < M ? row : M - 1) * K + (kk < K ? kk : 0);
                R.a[j][ks] = *reinterpret_cast<const uint4*>(a + off);
                if constexpr (PRO == DFD_PRO_AFFINE2) R.a2[j][ks] = *reinterpret_cast<const uint4*>(a2 + off);

            }
    };
    auto gate_load = [&](int mt, int kt) {
        if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
            const int rbase = mt * BM + wave * WR + frow;
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                int row = rbase + j * 16;
                if (row >= M) row = M - 1;
                const float* grow = pa.gate + (long)pro_image(pa, row) * K;
#pragma unroll
                for (int ks = 0; ks < KSN; ++ks) {
                    int kk = kt * BK + (ks * 4 + fk) * E;
                    if (kk >= K) kk = 0;
                    load_f32<E>(grow + kk, gt[j][ks]);
                }
            }
        }
    };
    auto prologue = [&](const Regs& R, uint4 (&fa)[JN][KSN], int kt) {
        const float* cl = reinterpret_cast<const float*>(smem + coff);
        const int kp = nk * BK;
#pragma unroll
        for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
            for (int j = 0; j < JN; ++j) {
                if constexpr (PRO == DFD_PRO_NONE) {
                    fa[j][ks] = (kt * BK + (ks * 4 + fk) * E < K) ? R.a[j][ks] : make_uint4(0, 0, 0, 0);
                } else {
                    const int kk = kt * BK + (ks * 4 + fk) * E;
                    float v[E], c0[E], c1[E];
                    load_f32<E>(cl + kk, c0);
                    load_f32<E>(cl + kp + kk, c1);
                    q_to_f(R.a[j][ks], v);
                    if constexpr (PRO == DFD_PRO_AFFINE2) {
                        float v2[E], c2[E];
                        load_f32<E>(cl + 2 * kp + kk, c2);
                        q_to_f(R.a2[j][ks], v2);
#pragma unroll
                        for (int x = 0; x < E; ++x) v[x] = fmaf(c0[x], v[x], fmaf(c1[x], v2[x], c2[x]));
                    } else {
                        bn_act_array<ACT, E>(v, c0, c1);
                        if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
                            // the activated tensor is rounded to T before the gate multiply, as an
                            // unfused pipeline would store it
#pragma unroll
                            for (int x = 0; x < E; ++x) v[x] = round_to<T>(v[x]) * gt[j][ks][x];
                        }
                    }
                    fa[j][ks] = kk < K ? f_to_q(v) : make_uint4(0, 0, 0, 0);
                }
            }
    };

    f32x4_t acc[NT][JN];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < JN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    unsigned char* eb = smem + eoff + wave * WR * OROW;
    auto advance = [&](int& m_, int& k_) { if (++k_ == nk) { k_ = 0; m_ += gx; } };

    // nchunks > 1 only with nk == 1 (host): the prologued fragments are then reused for every BN-wide
    // column chunk of the panel, so the activation tensor is read and transformed exactly once
    auto chunk = [&](const uint4 (&fa)[JN][KSN], int mt, int kt, int nc, float (&t1)[E], float (&t2)[E]) {
        {
            const unsigned char* wbase = smem + (nc * BN + frow) * wstride + (kt * KSN * 4 + fk) * 16;
#pragma unroll
            for (int ks = 0; ks < KSN; ++ks) {
                if (kt * BK + ks * 4 * E < K) {
                    const unsigned char* wp = wbase + ks * 64;
#pragma unroll
                    for (int i = 0; i < NT; ++i) {
                        const uint4 fw = *reinterpret_cast<const uint4*>(wp + i * 16 * wstride);
#pragma unroll
                        for (int j = 0; j < JN; ++j) {
                            if constexpr (sizeof(T) == 2) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                    __builtin_bit_cast(bf16x8_t, fw), __builtin_bit_cast(bf16x8_t, fa[j][ks]), acc[i][j], 0, 0, 0);
                            } else {
                                const f32x4_t wv = __builtin_bit_cast(f32x4_t, fw);
                                const f32x4_t av = __builtin_bit_cast(f32x4_t, fa[j][ks]);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], av[0], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], av[1], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], av[2], acc[i][j], 0, 0, 0);
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], av[3], acc[i][j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            if (kt != nk - 1) return;
            // ---- rows of this wave done: accumulators -> private LDS patch [WR][BN] -> row stores
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < JN; ++j) {
                    unsigned char* p = eb + (j * 16 + frow) * OROW + (i * 16 + fk * 4) * (int)sizeof(T);
                    if constexpr (sizeof(T) == 2) {
                        uint2 q;
                        q.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
                        q.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
                        *reinterpret_cast<uint2*>(p) = q;
                    } else {
                        *reinterpret_cast<float4*>(p) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    }
                    acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n = n0 + nc * BN + ec * E;
            const int mw = mt * BM + wave * WR;
            if (n < Nout) {
                float es[E], eh[E];
                if constexpr (EACT >= 0) { load_f32<E>(ebn + n, es); load_f32<E>(ebn + Nout + n, eh); }
                // rows in groups of four: the group's residual vectors are requested together, from clamped rows, before any
                // is used (inside the bounds check each load was waited for before the next row's was issued)
                constexpr int NPS = WR / RL;
#pragma unroll
                for (int g0 = 0; g0 < NPS; g0 += 4) {
                    float rq[4][E];
                    if constexpr (RES) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (g0 + u < NPS) {
                                const int m = mw + er + (g0 + u) * RL;
                                Vec<T>::load(res + (long)(m < M ? m : M - 1) * Nout + n, rq[u]);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (g0 + u >= NPS) continue;
                        const int r = er + (g0 + u) * RL;
                        const int m = mw + r;
                        if (m < M) {
                            float v[E];
                            q_to_f(*reinterpret_cast<const uint4*>(eb + r * OROW + ec * 16), v);
                            if constexpr (EACT >= 0) {
#pragma unroll
                                for (int x = 0; x < E; ++x) v[x] = round_to<T>(act_fwd<EACT>(fmaf(es[x], v[x], eh[x])));
                            }
                            if constexpr (RES) {
#pragma unroll
                                for (int x = 0; x < E; ++x) v[x] = round_to<T>(v[x] + rq[u][x]);
                            }
                            if constexpr (STATS) {
#pragma unroll
                                for (int x = 0; x < E; ++x) { t1[x] += v[x]; t2[x] = fmaf(v[x], v[x], t2[x]); }
                            }
                            Vec<T>::store(out + (long)m * Nout + n, v);
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    auto compute = [&](const uint4 (&fa)[JN][KSN], int mt, int kt) {
        if constexpr (STATS) {
#pragma unroll
            for (int nc = 0; nc < NCH; ++nc)
                if (nc < nchunks) chunk(fa, mt, kt, nc, s1[nc], s2[nc]);
        } else {
            for (int nc = 0; nc < nchunks; ++nc) chunk(fa, mt, kt, nc, s1[0], s2[0]);
        }
    };

    // ---- flat pipeline over (M tile, K tile): stage s is consumed from one register set while the
    // loads of stage s+1 (other set) and, right after the prologue, stage s+2 (this set) are in flight
    int mt = pb, kt = 0;
    if (mt < m_tiles) {
        int mt1 = mt, kt1 = 0;
        advance(mt1, kt1);
        gate_load(mt, 0);
        g_load(R0, mt, 0);
        if (mt1 < m_tiles) g_load(R1, mt1, kt1);
        auto step = [&](Regs& rcur) -> bool {
            uint4 fa[JN][KSN];
            prologue(rcur, fa, kt);
            int mt2 = mt1, kt2 = kt1;
            advance(mt2, kt2);
            const bool has1 = mt1 < m_tiles;
            if (has1) {
                gate_load(mt1, kt1);
                if (mt2 < m_tiles) g_load(rcur, mt2, kt2);
            }
            compute(fa, mt, kt);
            if (!has1) return false;
            mt = mt1; kt = kt1;
            mt1 = mt2; kt1 = kt2;
            return true;
        };
        for (;;) {
            if (!step(R0)) break;
            if (!step(R1)) break;
        }
    }
    if constexpr (STATS) {
        __syncthreads();                    // every wave is done with the panel and its patch
        float* red = reinterpret_cast<float*>(smem);
        const int rl = t / CPR;
#pragma unroll
        for (int nc = 0; nc < NCH; ++nc) {
            if (nc >= nchunks) break;
            float acc2[2 * E];
#pragma unroll
            for (int j = 0; j < E; ++j) { acc2[j] = s1[nc][j]; acc2[E + j] = s2[nc][j]; }
            reduce_rowlanes<2 * E>(acc2, red, CPR, DFD_THREADS / CPR, ec, rl, true);
            const int n = n0 + nc * BN + ec * E;
            if (rl == 0 && n < Nout) {
                float* p = partials + (long)pb * 2 * Nout;
                float a0[E], a1[E];
#pragma unroll
                for (int j = 0; j < E; ++j) { a0[j] = acc2[j]; a1[j] = acc2[E + j]; }
                store_f32<E>(p + n, a0);
                store_f32<E>(p + Nout + n, a1);
            }
        }
    }
}

// ===========================================================================
// host dispatch
// ===========================================================================
#define NTW_MAX_PANEL (64 * 1024)

// co-resident workgroups per CU of one instantiation: a workgroup is one wave per SIMD, so the
// register file allows 512 / VGPRs of them (allocation granule 8), LDS allows 160 KB / lds
template <typename KernelT>
static int ntw_blocks_per_cu(KernelT kern, int lds) {
    struct NtwTag;
    const int regs = dfd_kernel_regs_once<NtwTag>(kern, 256);   // one flag per instantiation of this template
    int by_regs = 512 / ((regs + 7) / 8 * 8);
    if (by_regs > 8) by_regs = 8;
    int by_lds = (160 * 1024) / (lds + 512);
    int nb = by_regs < by_lds ? by_regs : by_lds;
    return nb < 1 ? 1 : nb;
}

template <typename T, int BN, int KSN>
static int ntw_launch(const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M,
                      int K, int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn, int eact) {
    constexpr int E = El<T>::EPC;
    constexpr int BK = KSN * 4 * E;
    constexpr int WR = 16 * (4 / KSN), BM = 4 * WR;
    constexpr int OROW = BN * (int)sizeof(T) + 16;
    const int nk = (K + BK - 1) / BK;
    const int wstride = nk * KSN * 64 + 16;
    const bool stats = partials != nullptr;
    // column chunks handled by one workgroup: all of them when one K tile covers K (the prologued
    // fragments are reused), except with statistics (one accumulator set per lane)
    const int mode = pro ? pro->mode : DFD_PRO_NONE;
    int nchunks = 1;
    if (nk == 1 && !stats) nchunks = (Nout + BN - 1) / BN;
    if (nchunks > 4) return DFD_EUNSUPPORTED;
    const int panel = BN * nchunks * wstride;
    if (panel > NTW_MAX_PANEL) return DFD_EUNSUPPORTED;
    const int coff = (panel + 15) / 16 * 16;
    const int eoff = coff + 3 * nk * BK * 4;
    constexpr int RED = DFD_THREADS * 2 * E * 4;
    int lds = eoff + 4 * WR * OROW;
    if (lds < RED) lds = RED;

    const int m_tiles = (M + BM - 1) / BM, n_tiles = (Nout + BN * nchunks - 1) / (BN * nchunks);
    const ProArgs pa = pro_args(pro);
    const int act = (pro && (mode == DFD_PRO_BN_ACT || mode == DFD_PRO_BN_ACT_GATE)) ? pro->act : DFD_ACT_NONE;
    const bool has_res = residual != nullptr;
    const int cap = partials ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;

    // the weight panel is staged once per workgroup: only worth it when a workgroup then streams
    // several M tiles past it (otherwise the K-tiled kernel with its shared M tile is the better fit)
    if ((long)m_tiles * n_tiles < 1024 && nk * n_tiles > 2) return DFD_EUNSUPPORTED;

#define LAUNCH_NTW_(NCHV, PRO, RES, STATS)                                                                                      \
    do {                                                                                                                  \
        auto kern = k_pw_ntw<T, BN, KSN, NCHV, PRO, ACT, RES, STATS>;                                                        \
        int gx = (256 * ntw_blocks_per_cu(kern, lds)) / n_tiles;                                                          \
        if (gx < 32) gx = 32;                                                                                             \
        if (gx > cap) gx = cap;                                                                                           \
        if (gx > m_tiles) gx = m_tiles;                                                                                   \
        if (partials) *nparts = gx;                                                                                       \
        hipLaunchKernelGGL(kern, dim3(gx * n_tiles), dim3(DFD_THREADS), lds, st, (const T*)a, pa, (const T*)w, (T*)out,   \
                           (const T*)residual, M, K, Nout, m_tiles, n_tiles, gx, wstride, eoff, coff, nchunks, partials,  \
                           (const float*)nullptr);                                                                        \
    } while (0)
    // (statistics over several column chunks in one workgroup, NCH = 4, measured slower than
    // separate column tiles: not instantiated)
#define LAUNCH_NTW(PRO, RES, STATS) LAUNCH_NTW_(1, PRO, RES, STATS)
    if (ebn) {          // eval epilogue: plain operand, no residual, no statistics
        if (mode != DFD_PRO_NONE || has_res || stats || eact != DFD_ACT_SILU) return DFD_EUNSUPPORTED;
        auto kern = k_pw_ntw<T, BN, KSN, 1, DFD_PRO_NONE, DFD_ACT_NONE, false, false, DFD_ACT_SILU>;
        int gx = (256 * ntw_blocks_per_cu(kern, lds)) / n_tiles;
        if (gx < 32) gx = 32;
        if (gx > cap) gx = cap;
        if (gx > m_tiles) gx = m_tiles;
        hipLaunchKernelGGL(kern, dim3(gx * n_tiles), dim3(DFD_THREADS), lds, st, (const T*)a, pa, (const T*)w, (T*)out,
                           (const T*)nullptr, M, K, Nout, m_tiles, n_tiles, gx, wstride, eoff, coff, nchunks, (float*)nullptr, ebn);
        return DFD_CHECK_LAUNCH();
    }
    if (mode == DFD_PRO_AFFINE2) {
        constexpr int ACT = DFD_ACT_NONE;
        if (stats) return DFD_EUNSUPPORTED;
        if (has_res) LAUNCH_NTW(DFD_PRO_AFFINE2, true, false); else LAUNCH_NTW(DFD_PRO_AFFINE2, false, false);
    } else if (mode == DFD_PRO_NONE) {
        constexpr int ACT = DFD_ACT_NONE;
        if (has_res && stats) return DFD_EUNSUPPORTED;
        if (has_res) LAUNCH_NTW(DFD_PRO_NONE, true, false);
        else if (stats) LAUNCH_NTW(DFD_PRO_NONE, false, true);
        else LAUNCH_NTW(DFD_PRO_NONE, false, false);
    } else {
        if (has_res) return DFD_EUNSUPPORTED;
        if (mode == DFD_PRO_BN_ACT) {
            DISPATCH_ACT_PW(act, { if (stats) LAUNCH_NTW(DFD_PRO_BN_ACT, false, true); else LAUNCH_NTW(DFD_PRO_BN_ACT, false, false); });
        } else {
            DISPATCH_ACT_PW(act, { if (stats) LAUNCH_NTW(DFD_PRO_BN_ACT_GATE, false, true); else LAUNCH_NTW(DFD_PRO_BN_ACT_GATE, false, false); });
        }
    }
#undef LAUNCH_NTW
#undef LAUNCH_NTW_
    return DFD_CHECK_LAUNCH();
}

template <typename T>
static int ntw_t(const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M, int K,
                 int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn, int eact) {
#define NTW_GO(BN, KSN) return ntw_launch<T, BN, KSN>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st, ebn, eact)
    constexpr int E = El<T>::EPC;
    if (K <= 4 * E) {
        // one 64-byte k sub-step covers K: 64-row stages, column tiles of at most 64
        if (Nout <= 32) NTW_GO(32, 1);
        NTW_GO(64, 1);
    }
    if (Nout <= 32) NTW_GO(32, 2);
    if (Nout <= 64) NTW_GO(64, 2);
    // one K tile and no statistics: 64-wide column chunks looped inside the workgroup
    if (K <= 8 * E && !partials && Nout <= 256) NTW_GO(64, 2);
    NTW_GO(128, 2);
#undef NTW_GO
}

int dfd_pw_ntw(int dtype, const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M,
               int K, int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn, int eact) {
    if (dtype == DFD_BF16) return ntw_t<bf16>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st, ebn, eact);
    if (dtype == DFD_F32) return ntw_t<float>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st, ebn, eact);
    return DFD_EINVAL;
}
