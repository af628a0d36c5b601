// dfd_pwconv.hip — 1x1 convolutions as MFMA GEMMs over the NHWC row matrix.
//
//   forward / data-gradient ("NT"):  out[M][Nout] = P(a)[M][K] * w[Nout][K]^T
//   weight gradient        ("TN"):  dw[Ni][Nj]   = sum_m P(p)[m][i] * Q(q)[m][j]
//
// gfx950 specifics:
//   * bf16: v_mfma_f32_16x16x32_bf16, f32: v_mfma_f32_16x16x4_f32 (exact f32).
//   * LDS tiles have 128-byte rows of eight 16-byte chunks, XOR-swizzled
//     (chunk ^= row & 7) so the ds_read_b128 fragment reads are conflict-free.
//   * the weights are the MFMA "A" operand and the activations the "B" operand, so a
//     lane's four accumulator registers are four CONSECUTIVE output channels of one
//     row; the tile goes back through LDS and is written with 16-byte row-major stores.
//   * the producer's BN + activation (+ SE gate), or the BN-backward affine map of two
//     tensors, is applied in registers between the global load and the LDS write:
//     the normalised/activated tensor never exists in HBM.
//   * the TN kernel reads both operands with ds_read_b64_tr_b16 (hardware transpose)
//     from tiles stored exactly as they sit in memory ([m][channel]).
//   * workgroups are persistent over M tiles: per-channel (sum, sumsq) stay in
//     registers and leave as ONE partial row per workgroup (reproducible reductions).
#include "dfd_pw.h"

// ===========================================================================
// NT kernel
// ===========================================================================
// LDS layout of the NT kernel: A double buffer at 0, then (past the epilogue overlay, so a
// resident weight tile survives it) the B double buffer.
template <typename T, int BN, int BM> struct NtLds {
    static constexpr int A_BYTES = BM * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int OROW = BN * (int)sizeof(T) + 16;                   // epilogue row stride
    static constexpr int OBYTES = BM * OROW;
    static constexpr int BOFF = ((2 * A_BYTES > OBYTES ? 2 * A_BYTES : OBYTES) + 15) / 16 * 16;
    static constexpr int TOTAL = BOFF + 2 * B_BYTES;
};

// BM = rows of a workgroup tile: 128, or 64 when M is so small that 128-row tiles leave CUs idle
// CONV: the A operand is gathered from an NHWC image (implicit GEMM of a dense k x k convolution, ConvArgs): a lane's
// 16-byte chunk is 8 channels of one tap, zero outside the image AFTER the producer's BN + activation
// EACT >= 0 (eval / inference): the epilogue stores act(scale * y + shift) with this layer's own BatchNorm coefficients
// (ebn: float[2+][Nout]) instead of the raw output, so that the consumer needs no prologue
template <typename T, int BN, int BM, int PRO, int ACT, bool RES, bool STATS, bool CONV = false, int EACT = -1>
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_pw_nt(const T* __restrict__ a, ProArgs pa, const T* __restrict__ w, T* __restrict__ out, const T* __restrict__ res,
        int M, int K, int Nout, int m_tiles, int n_tiles, int gx, float* __restrict__ partials, ConvArgs cv,
        const float* __restrict__ ebn = nullptr) {
    constexpr int E = El<T>::EPC;
    constexpr int BK = El<T>::BK;
    constexpr int NTW = BN / 32;            // 16-wide n tiles per wave
    using L = NtLds<T, BN, BM>;
    constexpr int RF = BM / 32;             // 16-row fragments per wave
    constexpr int AI = BM / 32;             // A staging items per thread and K tile (BM rows x 8 chunks / 256)
    constexpr int OROW = L::OROW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware remap: workgroups that share an M tile (different n tiles) sit on one XCD
    const int total = gx * n_tiles;
    int lin = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int pb = lin / n_tiles, nt = lin - pb * n_tiles;
    const int n0 = nt * BN;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fk = lane >> 4;

    // epilogue mapping
    constexpr int CPR = BN / E;             // 16-B chunks per output row
    constexpr int RL = DFD_THREADS / CPR;
    const int ec = t % CPR, er = t / CPR;
    float s1[E], s2[E];
#pragma unroll
    for (int j = 0; j < E; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

    const int nk = (K + BK - 1) / BK;
    const bool b_resident = nk == 1;        // the whole weight tile fits one K step: stage it once
    const T* a2 = reinterpret_cast<const T*>(pa.a2);
    if (pb >= m_tiles) return;

    // staging geometry of a K tile: kc valid 16-byte chunks, kc4 (4 or 8) = chunks an MFMA sub-step
    // may read (written as zeros beyond kc).  A thread keeps ONE chunk column c per stage and walks
    // rows r = t / kc4 + i * (256 / kc4), so its per-channel coefficients are loaded once per stage.
    auto tile_kc = [&](int kt, int& kc, int& sh) {
        int rem = (K - kt * BK + E - 1) / E;
        kc = rem > 8 ? 8 : rem;
        sh = kc > 4 ? 3 : 2;
    };

    // two register sets: the loads of stage s+2 are issued while stage s computes and stage s+1
    // waits in the other set for its turn through the prologue into LDS
    struct Regs { uint4 a[AI], a2[AI], b[BN / 32]; unsigned ok; };
    Regs R0, R1;
    // Full K stages (all 8 chunks): a lane's rows and chunk are the same for every such stage of a row
    // tile, so its byte offsets inside the tile are computed ONCE per tile (weights: once per kernel) and a stage's addresses
    // are a wave-uniform base + that offset — the per-stage clamps and 64-bit multiply-adds (~90 VALU instructions per lane
    // and stage, as many issue cycles as the stage's 32 MFMAs) are gone from the K loop.
    // (a K that is not a multiple of the tile: every stage but the last is full, the last one takes the general path)
    const bool kany = !CONV && K >= BK;
    auto stage_full = [&](int kt) { return kany && (kt + 1) * BK <= K; };
    unsigned voffA[AI], voffB[BN / 32];
    if (kany) {
        const int c = t & 7, rb0 = t >> 3;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int r = rb0 + i * 32;
            const int dn = n0 + r < Nout ? r : Nout - 1 - n0;
            voffB[i] = (unsigned)(dn * K + c * E) * (unsigned)sizeof(T);
        }
    }
    auto g_load = [&](Regs& R, int mt, int kt, bool with_b) {
        if (stage_full(kt)) {
            const int m0 = mt * BM, k0 = kt * BK;
            if (kt == 0) {                               // wave-uniform: first stage of a row tile
                const int c = t & 7, rb0 = t >> 3;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    const int r = rb0 + i * 32;
                    const int dm = m0 + r < M ? r : M - 1 - m0;
                    voffA[i] = (unsigned)(dm * K + c * E) * (unsigned)sizeof(T);
                }
            }
            const char* baseA = reinterpret_cast<const char*>(a) + ((long)m0 * K + k0) * (long)sizeof(T);
#pragma unroll
            for (int i = 0; i < AI; ++i) R.a[i] = *reinterpret_cast<const uint4*>(baseA + voffA[i]);
            if constexpr (PRO == DFD_PRO_AFFINE2) {
                const char* baseA2 = reinterpret_cast<const char*>(a2) + ((long)m0 * K + k0) * (long)sizeof(T);
#pragma unroll
                for (int i = 0; i < AI; ++i) R.a2[i] = *reinterpret_cast<const uint4*>(baseA2 + voffA[i]);
            }
            if (with_b) {
                const char* baseB = reinterpret_cast<const char*>(w) + ((long)n0 * K + k0) * (long)sizeof(T);
#pragma unroll
                for (int i = 0; i < BN / 32; ++i) R.b[i] = *reinterpret_cast<const uint4*>(baseB + voffB[i]);
            }
            return;
        }
        int kc, sh;
        tile_kc(kt, kc, sh);
        const int c = t & ((1 << sh) - 1), rb0 = t >> sh, rstep = DFD_THREADS >> sh;
        const int m0 = mt * BM, k0 = kt * BK;
        if constexpr (CONV) {
            const int k = k0 + c * E, tap = udiv(k, cv.c), ch = k - tap * cv.C;
            const int dy = udiv(tap, cv.kk), dx = tap - dy * cv.ks;
            R.ok = 0u;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int r = rb0 + i * rstep, m = m0 + r;
                R.a[i] = make_uint4(0, 0, 0, 0);
                if (r < BM && c < kc && m < M) {
                    const int n = udiv(m, cv.howo), rem = m - n * (cv.Ho * cv.Wo);
                    const int oy = udiv(rem, cv.wo), ox = rem - oy * cv.Wo;
                    const int iy = oy * cv.stride - cv.pt + dy, ix = ox * cv.stride - cv.pl + dx;
                    if ((unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W) {
                        R.a[i] = *reinterpret_cast<const uint4*>(a + (((long)n * cv.H + iy) * cv.W + ix) * cv.C + ch);
                        R.ok |= 1u << i;
                    }
                }
            }
        } else {
        // Unconditional loads from a row / chunk clamped into the operand; what does not belong to the tile is replaced by
        // zeros at the LDS store (s_store).  With the loads inside divergent branches the compiler cannot count how many are
        // in flight: it waited vmcnt(0) in front of the MFMAs of every K step, i.e. for the stage it had just requested,
        // and the two-stage pipeline hid nothing.
        const int cc = c < kc ? c : 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = rb0 + i * rstep;
            const int rr = r < BM ? r : rb0;                    // (narrow K tiles: 64 rows per pass, the tile has fewer passes)
            const int m = m0 + rr < M ? m0 + rr : M - 1;
            const long off = (long)m * K + k0 + cc * E;
            R.a[i] = *reinterpret_cast<const uint4*>(a + off);
            if constexpr (PRO == DFD_PRO_AFFINE2) R.a2[i] = *reinterpret_cast<const uint4*>(a2 + off);
        }
        }
        if (with_b) {
            const int cc = c < kc ? c : 0;
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int r = rb0 + i * rstep;
                const int n = n0 + r < Nout ? n0 + r : Nout - 1;
                R.b[i] = *reinterpret_cast<const uint4*>(w + (long)n * K + k0 + cc * E);
            }
        }
    };
    // per-channel prologue coefficients of the stage that is stored next
    float c0[E], c1[E], c2[E];
    auto c_load = [&](int kt) {
        if constexpr (PRO != DFD_PRO_NONE) {
            int kc, sh;
            tile_kc(kt, kc, sh);
            const int c = t & ((1 << sh) - 1);
            int k = c < kc ? kt * BK + c * E : 0, rowlen = K;
            if constexpr (CONV) { k -= udiv(k, cv.c) * cv.C; rowlen = cv.C; }      // coefficients are per input channel
            load_f32<E>(pa.coef + k, c0);
            load_f32<E>(pa.coef + rowlen + k, c1);
            if constexpr (PRO == DFD_PRO_AFFINE2) load_f32<E>(pa.coef + 2 * rowlen + k, c2);
        }
    };
    auto s_store = [&](const Regs& R, int mt, int kt, int buf, bool with_b) {
        if (stage_full(kt)) {                            // all 8 chunks, 32 rows per pass: no per-item range checks, no branches
            const int c = t & 7, rb0 = t >> 3;
            unsigned char* ab = smem + buf * L::A_BYTES;
            unsigned char* bb = smem + L::BOFF + buf * L::B_BYTES;
            const int m0 = mt * BM, k0 = kt * BK;
            const bool full_m = m0 + BM <= M, full_n = n0 + BN <= Nout;      // wave-uniform: only edge tiles select zeros
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const int r = rb0 + i * 32;
                const bool valid = full_m || m0 + r < M;
                uint4 q = R.a[i];
                if constexpr (PRO != DFD_PRO_NONE) {
                    const float* gk = nullptr;
                    if constexpr (PRO == DFD_PRO_BN_ACT_GATE) gk = pa.gate + (long)pro_image(pa, valid ? m0 + r : M - 1) * K + k0 + c * E;
                    q = apply_pro_c<T, PRO, ACT, E>(R.a[i], R.a2[i], c0, c1, c2, gk);
                }
                if (!valid) q = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(ab + r * 128 + ((c ^ (r & 7)) << 4)) = q;
            }
            if (with_b) {
#pragma unroll
                for (int i = 0; i < BN / 32; ++i) {
                    const int r = rb0 + i * 32;
                    *reinterpret_cast<uint4*>(bb + r * 128 + ((c ^ (r & 7)) << 4)) = (full_n || n0 + r < Nout) ? R.b[i] : make_uint4(0, 0, 0, 0);
                }
            }
            return;
        }
        int kc, sh;
        tile_kc(kt, kc, sh);
        const int c = t & ((1 << sh) - 1), rb0 = t >> sh, rstep = DFD_THREADS >> sh;
        unsigned char* ab = smem + buf * L::A_BYTES;
        unsigned char* bb = smem + L::BOFF + buf * L::B_BYTES;
        const int m0 = mt * BM, k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = rb0 + i * rstep;
            if (r >= BM) continue;
            const bool valid = c < kc && m0 + r < M && (!CONV || ((R.ok >> i) & 1u));
            uint4 q = (CONV || valid) ? R.a[i] : make_uint4(0, 0, 0, 0);
            if (PRO != DFD_PRO_NONE && valid) {
                const float* gk = nullptr;
                if constexpr (PRO == DFD_PRO_BN_ACT_GATE) gk = pa.gate + (long)pro_image(pa, m0 + r) * K + k0 + c * E;
                q = apply_pro_c<T, PRO, ACT, E>(R.a[i], R.a2[i], c0, c1, c2, gk);
            }
            *reinterpret_cast<uint4*>(ab + r * 128 + ((c ^ (r & 7)) << 4)) = q;
        }
        if (with_b) {
#pragma unroll
            for (int i = 0; i < BN / 32; ++i) {
                const int r = rb0 + i * rstep;
                if (r < BN)
                    *reinterpret_cast<uint4*>(bb + r * 128 + ((c ^ (r & 7)) << 4)) = (c < kc && n0 + r < Nout) ? R.b[i] : make_uint4(0, 0, 0, 0);
            }
        }
    };

    // flat software pipeline over (M tile, K tile), two stages deep: while stage s runs its MFMAs
    // (and epilogue), the loads of stage s+2 are issued and stage s+1 sits in registers; its
    // prologue + LDS store follow the MFMAs, so a global load has a full stage to land
    f32x4_t acc[NTW][RF];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < RF; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool next_b = !b_resident;
    auto advance = [&](int& m_, int& k_) { if (++k_ == nk) { k_ = 0; m_ += gx; } };
    auto compute = [&](int mt, int kt, int buf) {
        {
                const unsigned char* ab = smem + buf * L::A_BYTES;
                const unsigned char* bb = smem + L::BOFF + (b_resident ? 0 : buf) * L::B_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int klim = kt * BK + ks * (BK / 2);
                    if (klim < K) {
                        const int c = ks * 4 + fk;
                        uint4 fa[RF], fw[NTW];
#pragma unroll
                        for (int i = 0; i < RF; ++i) {
                            const int r = wm * (BM / 2) + i * 16 + frow;
                            fa[i] = *reinterpret_cast<const uint4*>(ab + r * 128 + ((c ^ (r & 7)) << 4));
                        }
#pragma unroll
                        for (int i = 0; i < NTW; ++i) {
                            const int r = wn * (BN / 2) + i * 16 + frow;
                            fw[i] = *reinterpret_cast<const uint4*>(bb + r * 128 + ((c ^ (r & 7)) << 4));
                        }
#pragma unroll
                        for (int i = 0; i < NTW; ++i)
#pragma unroll
                            for (int j = 0; j < RF; ++j) {
                                if constexpr (sizeof(T) == 2) {
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                        __builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fa[j]), acc[i][j], 0, 0, 0);
                                } else {
                                    const f32x4_t wv = __builtin_bit_cast(f32x4_t, fw[i]);
                                    const f32x4_t av = __builtin_bit_cast(f32x4_t, fa[j]);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[0], av[0], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[1], av[1], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[2], av[2], acc[i][j], 0, 0, 0);
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[3], av[3], acc[i][j], 0, 0, 0);
                                }
                            }
                    }
                }
            }
            if (kt == nk - 1) {
                // ---- tile done: accumulators -> LDS overlay [m][n] -> 16-byte row-major stores
                const int m0 = mt * BM;
                __syncthreads();            // every wave is done reading the A buffers the overlay covers
#pragma unroll
                for (int i = 0; i < NTW; ++i)
#pragma unroll
                    for (int j = 0; j < RF; ++j) {
                        const int m = wm * (BM / 2) + j * 16 + frow;
                        const int n = wn * (BN / 2) + i * 16 + fk * 4;
                        unsigned char* p = smem + m * OROW + n * (int)sizeof(T);
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
                __syncthreads();
                const int n = n0 + ec * E;
                if (n < Nout) {
                    float es[E], eh[E];
                    if constexpr (EACT >= 0) { load_f32<E>(ebn + n, es); load_f32<E>(ebn + Nout + n, eh); }
                    // rows in groups of four: the group's residual vectors are requested together, before any of them is
                    // used (one at a time, each load was waited for before the next row's was issued)
                    constexpr int NR = (BM + RL - 1) / RL;
#pragma unroll
                    for (int g0 = 0; g0 < NR; g0 += 4) {
                        float rq[4][E];
                        if constexpr (RES) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (g0 + u < NR) {
                                    const int m = m0 + er + (g0 + u) * RL;
                                    Vec<T>::load(res + (long)(m < M ? m : M - 1) * Nout + n, rq[u]);
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (g0 + u >= NR) continue;
                            const int r = er + (g0 + u) * RL;
                            const int m = m0 + r;
                            if (r >= BM || m >= M) continue;
                            float v[E];
                            q_to_f(*reinterpret_cast<const uint4*>(smem + r * OROW + ec * 16), v);
                            if constexpr (EACT >= 0) {
#pragma unroll
                                for (int j = 0; j < E; ++j) v[j] = round_to<T>(act_fwd<EACT>(fmaf(es[j], v[j], eh[j])));
                            }
                            if constexpr (RES) {
#pragma unroll
                                for (int j = 0; j < E; ++j) v[j] = round_to<T>(v[j] + rq[u][j]);
                            }
                            if constexpr (STATS) {
#pragma unroll
                                for (int j = 0; j < E; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
                            }
                            Vec<T>::store(out + (long)m * Nout + n, v);
                        }
                    }
                }
                __syncthreads();            // overlay reads done before the next stage lands in the A buffers
            }
    };
    int mt = pb, kt = 0, buf = 0;
    int mt1 = mt, kt1 = 0;
    advance(mt1, kt1);
    g_load(R0, mt, 0, true);
    c_load(0);
    if (mt1 < m_tiles) g_load(R1, mt1, kt1, next_b);
    s_store(R0, mt, 0, 0, true);
    __syncthreads();
    // one step: stage (mt,kt) is in LDS buffer `buf`, stage (mt1,kt1) is in `rnext`, `rfree` is free
    auto step = [&](Regs& rfree, const Regs& rnext) -> bool {
        int mt2 = mt1, kt2 = kt1;
        advance(mt2, kt2);
        // coefficients first: they are consumed first (the next s_store), and a wait for the YOUNGEST load drains every
        // older one — requested after the prefetch, they forced the stage just requested to land before the MFMAs
        if (mt1 < m_tiles) c_load(kt1);
        if (mt1 < m_tiles && mt2 < m_tiles) g_load(rfree, mt2, kt2, next_b);
        compute(mt, kt, buf);
        if (mt1 >= m_tiles) return false;
        const int nbuf = (kt == nk - 1) ? 0 : (buf ^ 1);
        s_store(rnext, mt1, kt1, nbuf, next_b);
        __syncthreads();
        mt = mt1; kt = kt1; buf = nbuf;
        mt1 = mt2; kt1 = kt2;
        return true;
    };
    for (;;) {
        if (!step(R0, R1)) break;
        if (!step(R1, R0)) break;
    }
    if constexpr (STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        float acc2[2 * E];
#pragma unroll
        for (int j = 0; j < E; ++j) { acc2[j] = s1[j]; acc2[E + j] = s2[j]; }
        reduce_rowlanes<2 * E>(acc2, red, CPR, RL, ec, er, true);
        const int n = n0 + ec * E;
        if (er == 0 && n < Nout) {
            float* p = partials + (long)pb * 2 * Nout;
            float a0[E], a1[E];
#pragma unroll
            for (int j = 0; j < E; ++j) { a0[j] = acc2[j]; a1[j] = acc2[E + j]; }
            store_f32<E>(p + n, a0);
            store_f32<E>(p + Nout + n, a1);
        }
    }
}

// ===========================================================================
// TN kernel (weight gradient)
// tile rows are m (the reduction index); bf16: 64 rows x 256 B, f32: 32 rows x (512+64) B
// ===========================================================================
#define TN_B 128   // output tile edge (channels of p and of q)
#ifndef TN_TARGET_WGS
#define TN_TARGET_WGS 512   // workgroups a weight-gradient launch aims for (tiles x row splits): the default of dfd_tune key 12
#endif
#ifndef TN_MIN_STEPS
// reduction steps (of 64 rows) a row split must have at least.  8 starved the small-M layers of workgroups — 1,568 rows (batch 32 at
// 7x7) or 12,544 (batch 256) gave 3 / 24 splits for 18 output tiles and each split walked 8 exposed load -> barrier -> MFMA steps.
// Measured 8 / 4 / 2 / 1: EfficientNet-B0 batch 32 4.62 / 4.55 / 4.43 / 4.44 ms, batch 256 13.45 / 13.49 / 13.32 / 13.42,
// FasterViT-0 20.5 / - / 19.78 / 19.77, EfficientFormerV2-S1 18.0 / - / 18.03 / 18.02 (round 4).
#define TN_MIN_STEPS 2
#endif

#define TN_F32_ROW 576  // (128 + 16) floats: kq rows land on disjoint bank halves

// CONV: q is gathered from an NHWC image as the im2col matrix of a dense k x k convolution would hold it (row m = output
// pixel, column = (tap, channel)); its prologue coefficients are per input channel
template <typename T, int PROP, int PROQ, int ACT, bool CONV = false>
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_pw_tn(const T* __restrict__ p, ProArgs pp, int Ni, const T* __restrict__ q, ProArgs pq, int Nj, int M,
        int i_tiles, int j_tiles, int rows_per_split, int gate_imgs, float* __restrict__ ws, ConvArgs cv) {
    constexpr int E = El<T>::EPC;
    constexpr int BMK = (sizeof(T) == 2) ? 64 : 32;         // reduction rows per step
    constexpr int ROWB = (sizeof(T) == 2) ? 256 : TN_F32_ROW;
    constexpr int TILE = BMK * ROWB;
    constexpr int CPRW = TN_B / E;                           // chunks per tile row: 16 / 32
    constexpr int RPP = DFD_THREADS / CPRW;                  // rows per staging pass: 16 / 8
    constexpr int NPASS = BMK / RPP;                         // 4
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware order: the output tiles of one row split read the same rows of p and q, so they
    // go to one XCD (one L2), back to back; consecutive hardware ids are dealt over the 8 XCDs
    int tile_id = blockIdx.x, split = blockIdx.y;
    if (gridDim.x > 1) {
        const int nx = gridDim.x, total = nx * gridDim.y, lin = blockIdx.y * nx + blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = lin & 7, sl = lin >> 3;
        const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + sl;
        split = logical / nx;
        tile_id = logical - split * nx;
    }
    const int it = tile_id / j_tiles, jt = tile_id - it * j_tiles;
    const int i0 = it * TN_B, j0 = jt * TN_B;
    const int mbeg = split * rows_per_split;
    int mend = mbeg + rows_per_split;
    if (mend > M) mend = M;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const T* p2 = reinterpret_cast<const T*>(pp.a2);

    // Staging work is spread over the VALID 16-byte chunks only (channel counts of 16..48 are
    // common: with a fixed chunk-column per lane most lanes would idle through the prologue).
    // Items are numbered P rows first, then Q rows, so at most one wave mixes the two prologues.
    int cp = (Ni - i0 + E - 1) / E; if (cp > CPRW) cp = CPRW;
    int cq = (Nj - j0 + E - 1) / E; if (cq > CPRW) cq = CPRW;
    // per operand: item i of this lane is flat index t + 256*i over (row, valid chunk)
    constexpr int MAXI = (BMK * CPRW) / DFD_THREADS;           // 4
    int mp_[MAXI], mq_[MAXI];                                  // row | chunk << 8 | ok << 16
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int idx = t + DFD_THREADS * i;
        const bool okp = idx < BMK * cp, okq = idx < BMK * cq;
        const int rp_ = okp ? idx / cp : 0, rq_ = okq ? idx / cq : 0;
        mp_[i] = rp_ | ((okp ? idx - rp_ * cp : 0) << 8) | ((int)okp << 16);
        mq_[i] = rq_ | ((okq ? idx - rq_ * cq : 0) << 8) | ((int)okq << 16);
    }
    // chunks beyond cp / cq are never written: clear the tiles once so partly valid 16-wide MFMA
    // tiles read zeros there
    if (cp < CPRW || cq < CPRW) {
        for (int i = t; i < 4 * TILE / 16; i += DFD_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    }
    // the prologue coefficients of this output tile's channels, and the gate rows of every image the row
    // split touches, live in LDS: in the staging path a per-chunk global load sits between the prefetched
    // operand and its LDS store, where nothing hides its latency (this kernel has no registers to spare)
    float* ctab = reinterpret_cast<float*>(smem + 4 * TILE);          // [5][TN_B]: p c0,c1,c2 ; q scale,shift
    float* gtab = ctab + 5 * TN_B;                                    // [gate_imgs][TN_B]
    int img0 = 0;
    if constexpr (PROP != DFD_PRO_NONE || PROQ != DFD_PRO_NONE) {
        for (int i = t; i < 5 * TN_B; i += DFD_THREADS) {
            const int r = i / TN_B, c = i - r * TN_B;
            float v = 0.f;
            if (r < 3) { if (PROP == DFD_PRO_AFFINE2 && i0 + c < Ni) v = pp.coef[r * Ni + i0 + c]; }
            else if (PROQ != DFD_PRO_NONE && j0 + c < Nj) {
                if constexpr (CONV) { const int k = j0 + c; v = pq.coef[(r - 3) * cv.C + (k - udiv(k, cv.c) * cv.C)]; }
                else v = pq.coef[(r - 3) * Nj + j0 + c];
            }
            ctab[i] = v;
        }
        if constexpr (PROQ == DFD_PRO_BN_ACT_GATE) {
            if (gate_imgs > 0 && mend > mbeg) {
                img0 = pro_image(pq, mbeg);
                const int nimg = pro_image(pq, mend - 1) - img0 + 1;      // <= gate_imgs (host)
                for (int i = t; i < nimg * TN_B; i += DFD_THREADS) {
                    const int r = i / TN_B, c = i - r * TN_B;
                    gtab[i] = (j0 + c < Nj) ? pq.gate[(long)(img0 + r) * Nj + j0 + c] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 rp[MAXI], rp2[MAXI], rq[MAXI];
    unsigned qok = 0u;                                          // CONV: which of this step's q chunks lie inside the image
    auto g_load = [&](int mb) {
        if constexpr (CONV) qok = 0u;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            rp[i] = make_uint4(0, 0, 0, 0); rq[i] = make_uint4(0, 0, 0, 0);
            if (PROP == DFD_PRO_AFFINE2) rp2[i] = make_uint4(0, 0, 0, 0);
            const int a_ = mp_[i], b_ = mq_[i];
            const int ma = mb + (a_ & 255), mbq = mb + (b_ & 255);
            if ((a_ >> 16) && ma < mend) {
                const long off = (long)ma * Ni + i0 + ((a_ >> 8) & 255) * E;
                rp[i] = *reinterpret_cast<const uint4*>(p + off);
                if constexpr (PROP == DFD_PRO_AFFINE2) rp2[i] = *reinterpret_cast<const uint4*>(p2 + off);
            }
            if ((b_ >> 16) && mbq < mend) {
                if constexpr (CONV) {
                    const int k = j0 + ((b_ >> 8) & 255) * E, tap = udiv(k, cv.c), ch = k - tap * cv.C;
                    const int dy = udiv(tap, cv.kk), dx = tap - dy * cv.ks;
                    const int n = udiv(mbq, cv.howo), rem = mbq - n * (cv.Ho * cv.Wo);
                    const int oy = udiv(rem, cv.wo), ox = rem - oy * cv.Wo;
                    const int iy = oy * cv.stride - cv.pt + dy, ix = ox * cv.stride - cv.pl + dx;
                    if ((unsigned)iy < (unsigned)cv.H && (unsigned)ix < (unsigned)cv.W) {
                        rq[i] = *reinterpret_cast<const uint4*>(q + (((long)n * cv.H + iy) * cv.W + ix) * cv.C + ch);
                        qok |= 1u << i;
                    }
                } else {
                    rq[i] = *reinterpret_cast<const uint4*>(q + (long)mbq * Nj + j0 + ((b_ >> 8) & 255) * E);
                }
            }
        }
    };
    auto s_store = [&](int mb, int buf) {
        unsigned char* pb_ = smem + buf * 2 * TILE;
        unsigned char* qb_ = pb_ + TILE;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int a_ = mp_[i], b_ = mq_[i];
            if (a_ >> 16) {
                const int r = a_ & 255, chk = (a_ >> 8) & 255;
                uint4 v = rp[i];
                if (PROP != DFD_PRO_NONE && mb + r < mend) {
                    float c0[E], c1[E], c2[E];
                    load_f32<E>(ctab + chk * E, c0);
                    load_f32<E>(ctab + TN_B + chk * E, c1);
                    load_f32<E>(ctab + 2 * TN_B + chk * E, c2);
                    v = apply_pro_v<T, PROP, DFD_ACT_NONE, E>(rp[i], rp2[i], c0, c1, c2, c2);
                }
                if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(pb_ + tn_off_bf16(r, chk)) = v;
                else *reinterpret_cast<uint4*>(pb_ + r * ROWB + chk * 16) = v;
            }
            if (b_ >> 16) {
                const int r = b_ & 255, chk = (b_ >> 8) & 255, m = mb + r;
                uint4 v = rq[i];
                if (PROQ != DFD_PRO_NONE && m < mend && (!CONV || ((qok >> i) & 1u))) {
                    float c0[E], c1[E], gt[E];
                    load_f32<E>(ctab + 3 * TN_B + chk * E, c0);
                    load_f32<E>(ctab + 4 * TN_B + chk * E, c1);
                    if constexpr (PROQ == DFD_PRO_BN_ACT_GATE) {
                        const int img = pro_image(pq, m);
                        if (gate_imgs > 0) load_f32<E>(gtab + (img - img0) * TN_B + chk * E, gt);
                        else load_f32<E>(pq.gate + (long)img * Nj + j0 + chk * E, gt);
                    }
                    v = apply_pro_v<T, PROQ, ACT, E>(rq[i], rq[i], c0, c1, c1, gt);
                }
                if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(qb_ + tn_off_bf16(r, chk)) = v;
                else *reinterpret_cast<uint4*>(qb_ + r * ROWB + chk * 16) = v;
            }
        }
    };
    // 16-wide MFMA tiles of this wave that hold any valid channel
    int na = (Ni - i0 - wi * 64 + 15) / 16; na = na < 0 ? 0 : (na > 4 ? 4 : na);
    int nb = (Nj - j0 - wj * 64 + 15) / 16; nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);

    const int nsteps = (mend > mbeg) ? (mend - mbeg + BMK - 1) / BMK : 0;
    if (nsteps > 0) {
        g_load(mbeg);
        s_store(mbeg, 0);
    }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int buf = st & 1;
        if (st + 1 < nsteps) g_load(mbeg + (st + 1) * BMK);
        const unsigned char* pb_ = smem + buf * 2 * TILE;
        const unsigned char* qb_ = pb_ + TILE;
        if constexpr (sizeof(T) == 2) {
            // two k-substeps of 32 rows; fragment = 8 consecutive m for one channel:
            // two transposed 4x16 block reads per operand tile
            const int g = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
            auto mfma_step = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    short8_t fa[4], fb[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        if (!FULL && x >= na && x >= nb) continue;
                        const int chA = ((wi * 64 + x * 16) >> 3) + (pcol >> 1);
                        const int chB = ((wj * 64 + x * 16) >> 3) + (pcol >> 1);
                        const int m_lo = ks * 32 + 8 * g + qrow, m_hi = m_lo + 4;
                        const int sub = (pcol & 1) << 3;
                        short4_t a_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(pb_ + tn_off_bf16(m_lo, chA) + sub));
                        short4_t a_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(pb_ + tn_off_bf16(m_hi, chA) + sub));
                        short4_t b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(qb_ + tn_off_bf16(m_lo, chB) + sub));
                        short4_t b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(qb_ + tn_off_bf16(m_hi, chB) + sub));
                        fa[x] = (short8_t){a_lo[0], a_lo[1], a_lo[2], a_lo[3], a_hi[0], a_hi[1], a_hi[2], a_hi[3]};
                        fb[x] = (short8_t){b_lo[0], b_lo[1], b_lo[2], b_lo[3], b_hi[0], b_hi[1], b_hi[2], b_hi[3]};
                    }
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            if (FULL || (a < na && b < nb))
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[a]),
                                                                                  __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
                }
            };
            // a wave with any valid tile runs all 16 MFMAs in straight-line code: tiles past Ni / Nj only feed accumulators
            // that are never written out.  Guarding each MFMA (`a < na && b < nb`) put every one in its own basic block behind
            // an lgkmcnt(0), i.e. one exposed LDS round trip per MFMA on every edge tile — most tiles of 80..192-wide outputs.
            if (na > 0 && nb > 0) mfma_step(std::true_type{});
        } else {
            const int col = lane & 15, kq = lane >> 4;
#pragma unroll
            for (int ks = 0; ks < BMK / 4; ++ks) {
                float fa[4], fb[4];
                const int m = ks * 4 + kq;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    fa[x] = *reinterpret_cast<const float*>(pb_ + m * ROWB + (wi * 64 + x * 16 + col) * 4);
                    fb[x] = *reinterpret_cast<const float*>(qb_ + m * ROWB + (wj * 64 + x * 16 + col) * 4);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if (a < na && b < nb) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        if (st + 1 < nsteps) s_store(mbeg + (st + 1) * BMK, buf ^ 1);
        __syncthreads();
    }
    // D: col (lane&15) = j index, row 4*(lane>>4)+r = i index
    float* o = ws + (long)split * Ni * Nj;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + wj * 64 + b * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 64 + a * 16 + (lane >> 4) * 4 + r;
                if (i < Ni && j < Nj) o[(long)i * Nj + j] = acc[a][b][r];
            }
        }
}

// ===========================================================================
// weight preparation: f32 master [N][K] -> T [N][K] and T [K][N]
// ===========================================================================
template <typename T>
__global__ void k_prep_weights(const float* __restrict__ w, T* __restrict__ w_nk, T* __restrict__ w_kn, int N, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long)n * K);
    const float v = w[i];
    if constexpr (sizeof(T) == 2) {
        if (w_nk) reinterpret_cast<unsigned short*>(w_nk)[i] = f2bf(v);
        if (w_kn) reinterpret_cast<unsigned short*>(w_kn)[(long)k * N + n] = f2bf(v);
    } else {
        if (w_nk) reinterpret_cast<float*>(w_nk)[i] = v;
        if (w_kn) reinterpret_cast<float*>(w_kn)[(long)k * N + n] = v;
    }
}

extern "C" int dfd_pw_prep_weights(int dtype, const float* w, void* w_nk, void* w_kn, int N, int K, dfd_stream stream) {
    if (!w || N < 1 || K < 1 || (!w_nk && !w_kn)) return DFD_EINVAL;
    const long total = (long)N * K;
    const int threads = 256;
    const unsigned grid = (unsigned)((total + threads - 1) / threads);
    if (dtype == DFD_BF16)
        hipLaunchKernelGGL((k_prep_weights<bf16>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, w, (bf16*)w_nk, (bf16*)w_kn, N, K);
    else if (dtype == DFD_F32)
        hipLaunchKernelGGL((k_prep_weights<float>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, w, (float*)w_nk, (float*)w_kn, N, K);
    else
        return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

// ---- all derived weights of a network in ONE launch per 32 jobs (instead of one k_prep_weights per layer
// and one transpose per squeeze-excite block): job = f32 [N][K] -> `nk` (same layout, element type T or
// f32) and / or `kn` ([K][N]).  Jobs travel by value in the kernel arguments, so a captured hipGraph needs
// no table upload.
#define PREP_JOBS_PER_LAUNCH 32
struct PrepJobs { dfd_prep_job j[PREP_JOBS_PER_LAUNCH]; };
__global__ void __launch_bounds__(256)
k_prep_weights_multi(PrepJobs jobs) {
    const dfd_prep_job& jb = jobs.j[blockIdx.y];
    const long total = (long)jb.N * jb.K;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = ((long)blockIdx.x * 4 + u) * 256 + threadIdx.x;
        if (i >= total) return;
        const int n = (int)(i / jb.K), k = (int)(i - (long)n * jb.K);
        const float v = jb.src[i];
        if (jb.dtype == DFD_BF16) {
            if (jb.nk) reinterpret_cast<unsigned short*>(jb.nk)[i] = f2bf(v);
            if (jb.kn) reinterpret_cast<unsigned short*>(jb.kn)[(long)k * jb.N + n] = f2bf(v);
        } else {
            if (jb.nk) reinterpret_cast<float*>(jb.nk)[i] = v;
            if (jb.kn) reinterpret_cast<float*>(jb.kn)[(long)k * jb.N + n] = v;
        }
    }
}
extern "C" int dfd_prep_weights_multi(const dfd_prep_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) {
        const dfd_prep_job& j = jobs[i];
        if (!j.src || (!j.nk && !j.kn) || j.N < 1 || j.K < 1 || (j.dtype != DFD_BF16 && j.dtype != DFD_F32)) return DFD_EINVAL;
    }
    for (int base = 0; base < njobs; base += PREP_JOBS_PER_LAUNCH) {
        PrepJobs pj;
        const int cnt = njobs - base < PREP_JOBS_PER_LAUNCH ? njobs - base : PREP_JOBS_PER_LAUNCH;
        long most = 0;
        for (int i = 0; i < cnt; ++i) {
            pj.j[i] = jobs[base + i];
            const long t = (long)pj.j[i].N * pj.j[i].K;
            if (t > most) most = t;
        }
        hipLaunchKernelGGL(k_prep_weights_multi, dim3((unsigned)((most + 1023) / 1024), cnt), dim3(256), 0, (hipStream_t)stream, pj);
    }
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// host dispatch
// ===========================================================================
template <typename T, int BN, int BM>
static int pw_nt_launch(const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M,
                        int K, int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn = nullptr,
                        int eact = DFD_ACT_NONE) {
    const int m_tiles = (M + BM - 1) / BM, n_tiles = (Nout + BN - 1) / BN;
    int cap = partials ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    int gx = 2048 / n_tiles;
    if (gx < 32) gx = 32;
    if (gx > cap) gx = cap;
    if (gx > m_tiles) gx = m_tiles;
    if (partials) *nparts = gx;
    constexpr int RED = DFD_THREADS * 2 * El<T>::EPC * 4;
    int lds = NtLds<T, BN, BM>::TOTAL;
    if (lds < RED) lds = RED;
    const ProArgs pa = pro_args(pro);
    const int mode = pro ? pro->mode : DFD_PRO_NONE;
    const int act = (pro && (mode == DFD_PRO_BN_ACT || mode == DFD_PRO_BN_ACT_GATE)) ? pro->act : DFD_ACT_NONE;
    const bool has_res = residual != nullptr, stats = partials != nullptr;
    dim3 grid(gx * n_tiles);
    if (ebn) {          // eval epilogue: plain operand, no residual, no statistics (the EfficientNet expand convolution)
        if (mode != DFD_PRO_NONE || has_res || stats || eact != DFD_ACT_SILU) return DFD_EUNSUPPORTED;
        hipLaunchKernelGGL((k_pw_nt<T, BN, BM, DFD_PRO_NONE, DFD_ACT_NONE, false, false, false, DFD_ACT_SILU>), grid, dim3(DFD_THREADS),
                           lds, st, (const T*)a, pa, (const T*)w, (T*)out, (const T*)nullptr, M, K, Nout, m_tiles, n_tiles, gx,
                           (float*)nullptr, ConvArgs{}, ebn);
        return DFD_CHECK_LAUNCH();
    }
#define LAUNCH_NT(PRO, RES, STATS)                                                                                        \
    hipLaunchKernelGGL((k_pw_nt<T, BN, BM, PRO, ACT, RES, STATS>), grid, dim3(DFD_THREADS), lds, st, (const T*)a, pa, (const T*)w, \
                       (T*)out, (const T*)residual, M, K, Nout, m_tiles, n_tiles, gx, partials, ConvArgs{}, (const float*)nullptr)
    // combinations used by the engine: forward = {NONE, BN_ACT, BN_ACT_GATE} x stats, no residual;
    // data gradient = AFFINE2 (+ residual), no stats; plain = NONE
    if (mode == DFD_PRO_AFFINE2) {
        constexpr int ACT = DFD_ACT_NONE;
        if (stats) return DFD_EUNSUPPORTED;
        if (has_res) LAUNCH_NT(DFD_PRO_AFFINE2, true, false); else LAUNCH_NT(DFD_PRO_AFFINE2, false, false);
    } else if (mode == DFD_PRO_NONE) {
        constexpr int ACT = DFD_ACT_NONE;
        if (has_res && stats) return DFD_EUNSUPPORTED;
        if (has_res) LAUNCH_NT(DFD_PRO_NONE, true, false);
        else if (stats) LAUNCH_NT(DFD_PRO_NONE, false, true);
        else LAUNCH_NT(DFD_PRO_NONE, false, false);
    } else {
        if (has_res) return DFD_EUNSUPPORTED;
        if (mode == DFD_PRO_BN_ACT) {
            DISPATCH_ACT_PW(act, { if (stats) LAUNCH_NT(DFD_PRO_BN_ACT, false, true); else LAUNCH_NT(DFD_PRO_BN_ACT, false, false); });
        } else {
            DISPATCH_ACT_PW(act, { if (stats) LAUNCH_NT(DFD_PRO_BN_ACT_GATE, false, true); else LAUNCH_NT(DFD_PRO_BN_ACT_GATE, false, false); });
        }
    }
#undef LAUNCH_NT
    return DFD_CHECK_LAUNCH();
}

template <typename T>
static int pw_nt_t(const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M, int K,
                   int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn = nullptr,
                   int eact = DFD_ACT_NONE) {
    if constexpr (sizeof(T) == 2) {                      // plain product, many rows, deep K: 256 x 256 tiles with LDS-DMA staging
        if (!pro && !residual && !partials && !ebn) {
            const int rc = dfd_gemm_nt_dma(a, w, out, M, K, Nout, st);
            if (rc != DFD_EUNSUPPORTED) return rc;
        }
    }
    // 128-row tiles unless they would leave most CUs without a workgroup (the 7x7 / 14x14 layers)
    const long tiles128 = (long)((M + 127) / 128) * ((Nout + 127) / 128);
    const bool small = tiles128 < 256;      // measured: a win below one tile per CU, a loss above
#define PW_NT_GO(BNV)                                                                                                   \
    return small ? pw_nt_launch<T, BNV, 64>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st, ebn, eact)  \
                 : pw_nt_launch<T, BNV, 128>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st, ebn, eact)
    if (Nout <= 32) PW_NT_GO(32);
    if (Nout <= 64) PW_NT_GO(64);
    PW_NT_GO(128);
#undef PW_NT_GO
}

// ---- dense k x k convolution as an implicit GEMM on the NT kernel (the A operand is gathered, nothing is materialised)
template <typename T, int BN, int BM>
static int conv_nt_launch(const void* x, const dfd_dwconv_shape* s, const float* in_bnstate, int in_act, const void* w,
                          void* out, int Cout, float* partials, int pcap, int* nparts, hipStream_t st) {
    const int M = s->N * s->Ho * s->Wo, K = s->k * s->k * s->C, Nout = Cout;
    const int m_tiles = (M + BM - 1) / BM, n_tiles = (Nout + BN - 1) / BN;
    int cap = partials ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    int gx = 2048 / n_tiles;
    if (gx < 32) gx = 32;
    if (gx > cap) gx = cap;
    if (gx > m_tiles) gx = m_tiles;
    if (partials) *nparts = gx;
    constexpr int RED = DFD_THREADS * 2 * El<T>::EPC * 4;
    int lds = NtLds<T, BN, BM>::TOTAL;
    if (lds < RED) lds = RED;
    ProArgs pa{nullptr, in_bnstate, nullptr, 1, 0u, -1};
    ConvArgs cv{s->H, s->W, s->C, s->Ho, s->Wo, s->k, s->stride, s->pad_top, s->pad_left,
                make_magic(s->Ho * s->Wo), make_magic(s->Wo), make_magic(s->C), make_magic(s->k)};
    const bool stats = partials != nullptr;
    dim3 grid(gx * n_tiles);
#define LAUNCH_CV(PRO, STATS)                                                                                               \
    hipLaunchKernelGGL((k_pw_nt<T, BN, BM, PRO, ACT, false, STATS, true>), grid, dim3(DFD_THREADS), lds, st, (const T*)x, pa,   \
                       (const T*)w, (T*)out, (const T*)nullptr, M, K, Nout, m_tiles, n_tiles, gx, partials, cv, (const float*)nullptr)
    if (!in_bnstate) {
        constexpr int ACT = DFD_ACT_NONE;
        if (stats) LAUNCH_CV(DFD_PRO_NONE, true); else LAUNCH_CV(DFD_PRO_NONE, false);
    } else {
        DISPATCH_ACT_PW(in_act, { if (stats) LAUNCH_CV(DFD_PRO_BN_ACT, true); else LAUNCH_CV(DFD_PRO_BN_ACT, false); });
    }
#undef LAUNCH_CV
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_conv_fwd(int dtype, const void* x, const dfd_dwconv_shape* s, const float* in_bnstate, int in_act,
                            const void* w_nk, int Cout, void* y, float* partials, int pcap, int* nparts, dfd_stream stream) {
    if (!x || !s || !w_nk || !y || Cout < 8 || Cout % 8 || s->C < 8 || s->C % 8 || s->k < 1 || s->k > 7 || s->stride < 1) return DFD_EINVAL;
    if (s->N < 1 || s->Ho < 1 || s->Wo < 1 || (long)s->N * s->Ho * s->Wo > (1l << 30)) return DFD_EINVAL;
    if (partials && (!nparts || pcap < 1)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long M = (long)s->N * s->Ho * s->Wo;
    if (dtype == DFD_BF16) {                             // 3x3 stride 1: direct kernel, weights resident in registers
        const int rc = dfd_conv3_direct(x, s, in_bnstate, in_act, w_nk, Cout, y, partials, pcap, nparts, st);
        if (rc != DFD_EUNSUPPORTED) return rc;
    }
    const bool small = ((M + 127) / 128) * ((Cout + 127) / 128) < 256;
#define CONV_GO(TT, BNV) return small ? conv_nt_launch<TT, BNV, 64>(x, s, in_bnstate, in_act, w_nk, y, Cout, partials, pcap, nparts, st) \
                                      : conv_nt_launch<TT, BNV, 128>(x, s, in_bnstate, in_act, w_nk, y, Cout, partials, pcap, nparts, st)
    if (dtype == DFD_BF16) { if (Cout <= 64) CONV_GO(bf16, 64); CONV_GO(bf16, 128); }
    if (dtype == DFD_F32) { if (Cout <= 64) CONV_GO(float, 64); CONV_GO(float, 128); }
#undef CONV_GO
    return DFD_EINVAL;
}

extern "C" int dfd_pwconv_fwd(int dtype, const void* a, const dfd_prologue* pro, const void* w, void* out,
                              const void* residual, int M, int K, int Nout, float* partials, int pcap, int* nparts,
                              dfd_stream stream) {
    if (!a || !w || !out || M < 1 || K < 8 || Nout < 8 || K % 8 || Nout % 8 || !pro_ok(pro)) return DFD_EINVAL;
    if (partials && (!nparts || pcap < 1)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // small weight panels (the large-M layers): wave-autonomous kernel with the panel resident in LDS
    const int rc = dfd_pw_ntw(dtype, a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st);
    if (rc != DFD_EUNSUPPORTED) return rc;
    // mid-size layers: LDS-DMA ring kernel (plain products that fill the chip with 256-wide tiles stay with dfd_gemm.hip)
    if (dtype == DFD_BF16 && !(!pro && !residual && !partials && dfd_gemm_plan(M, K, Nout))) {
        const int rd = dfd_pw_ntd(dtype, a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st);
        if (rd != DFD_EUNSUPPORTED) return rd;
    }
    if (dtype == DFD_BF16) return pw_nt_t<bf16>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st);
    if (dtype == DFD_F32) return pw_nt_t<float>(a, pro, w, out, residual, M, K, Nout, partials, pcap, nparts, st);
    return DFD_EINVAL;
}

extern "C" int dfd_pwconv_fwd_eval(int dtype, const void* a, const void* w, const float* out_bnstate, int out_act, void* out,
                                   int M, int K, int Nout, dfd_stream stream) {
    if (!a || !w || !out || !out_bnstate || M < 1 || K < 8 || Nout < 8 || K % 8 || Nout % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rc = dfd_pw_ntw(dtype, a, nullptr, w, out, nullptr, M, K, Nout, nullptr, 0, nullptr, st, out_bnstate, out_act);
    if (rc != DFD_EUNSUPPORTED) return rc;
    if (dtype == DFD_BF16) return pw_nt_t<bf16>(a, nullptr, w, out, nullptr, M, K, Nout, nullptr, 0, nullptr, st, out_bnstate, out_act);
    if (dtype == DFD_F32) return pw_nt_t<float>(a, nullptr, w, out, nullptr, M, K, Nout, nullptr, 0, nullptr, st, out_bnstate, out_act);
    return DFD_EINVAL;
}

static void tn_plan(int dtype, int M, int Ni, int Nj, int* i_tiles, int* j_tiles, int* splits, int* rows_per_split) {
    const int bmk = dtype == DFD_BF16 ? 64 : 32;
    *i_tiles = (Ni + TN_B - 1) / TN_B;
    *j_tiles = (Nj + TN_B - 1) / TN_B;
    const int tiles = *i_tiles * *j_tiles;
    int s = dfd_tune_get(DFD_TUNE_TN_WGS) / tiles;
    if (s < 1) s = 1;
    // keep the slab under 64 MiB and give each split at least 8 reduction steps
    const long per = (long)Ni * Nj * 4;
    long cap = (64l << 20) / per;
    if (cap < 1) cap = 1;
    if (s > cap) s = (int)cap;
    int max_s = (M + bmk * TN_MIN_STEPS - 1) / (bmk * TN_MIN_STEPS);
    if (max_s < 1) max_s = 1;
    if (s > max_s) s = max_s;
    int rps = (M + s - 1) / s;
    rps = ((rps + bmk - 1) / bmk) * bmk;
    s = (M + rps - 1) / rps;
    *splits = s;
    *rows_per_split = rps;
}

extern "C" size_t dfd_pwconv_wgrad_ws(int M, int Ni, int Nj) {
    if (M < 1 || Ni < 1 || Nj < 1) return 0;
    int it, jt, s, rps;
    size_t best = 0;
    for (int dt = 0; dt < 2; ++dt) {
        tn_plan(dt, M, Ni, Nj, &it, &jt, &s, &rps);
        const size_t b = (size_t)(s + s / 32 + 2) * Ni * Nj * 4;
        if (b > best) best = b;
    }
    // the wave-autonomous kernels (dfd_pwtnw.hip: weight gradient, fused expand backward) write one slab per workgroup, up to 512
    const int narrow = Ni < Nj ? Ni : Nj, wide = Ni < Nj ? Nj : Ni;
    if (narrow <= 48 && wide <= 240) {
        const size_t b = (size_t)(512 + 512 / 32 + 2) * Ni * Nj * 4;
        if (b > best) best = b;
    }
    return best;
}

template <typename T>
static int pw_tn_t(const void* p, const dfd_prologue* pro_p, int Ni, const void* q, const dfd_prologue* pro_q, int Nj,
                   int M, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    int it, jt, splits, rps;
    tn_plan(sizeof(T) == 2 ? DFD_BF16 : DFD_F32, M, Ni, Nj, &it, &jt, &splits, &rps);
    if ((size_t)(splits + splits / 32 + 2) * Ni * Nj * 4 > ws_bytes) return DFD_EWORKSPACE;
    constexpr int BMK = (sizeof(T) == 2) ? 64 : 32;
    constexpr int ROWB = (sizeof(T) == 2) ? 256 : TN_F32_ROW;
    int lds = 2 * 2 * BMK * ROWB + 5 * TN_B * 4;
    const ProArgs pp = pro_args(pro_p), pq = pro_args(pro_q);
    const int mp = pro_p ? pro_p->mode : DFD_PRO_NONE, mq = pro_q ? pro_q->mode : DFD_PRO_NONE;
    // gate rows in LDS when the images one row split touches fit beside two resident workgroups' tiles
    int gate_imgs = 0;
    if (mq == DFD_PRO_BN_ACT_GATE) {
        const int need = (rps + pq.HW - 2) / pq.HW + 1;
        if ((size_t)lds + (size_t)need * TN_B * 4 <= (size_t)80 * 1024) gate_imgs = need;
        lds += gate_imgs * TN_B * 4;
    }
    const int act = (pro_q && (mq == DFD_PRO_BN_ACT || mq == DFD_PRO_BN_ACT_GATE)) ? pro_q->act : DFD_ACT_NONE;
    if (!(mp == DFD_PRO_NONE || mp == DFD_PRO_AFFINE2)) return DFD_EUNSUPPORTED;
    if (mq == DFD_PRO_AFFINE2) return DFD_EUNSUPPORTED;
    dim3 grid(it * jt, splits);
#define LAUNCH_TN(PP, PQ)                                                                                                \
    hipLaunchKernelGGL((k_pw_tn<T, PP, PQ, ACT>), grid, dim3(DFD_THREADS), lds, st, (const T*)p, pp, Ni, (const T*)q, pq, Nj, \
                       M, it, jt, rps, gate_imgs, ws, ConvArgs{})
    if (mq == DFD_PRO_NONE) {
        constexpr int ACT = DFD_ACT_NONE;
        if (mp == DFD_PRO_AFFINE2) LAUNCH_TN(DFD_PRO_AFFINE2, DFD_PRO_NONE); else LAUNCH_TN(DFD_PRO_NONE, DFD_PRO_NONE);
    } else if (mq == DFD_PRO_BN_ACT) {
        DISPATCH_ACT_PW(act, { if (mp == DFD_PRO_AFFINE2) LAUNCH_TN(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT); else LAUNCH_TN(DFD_PRO_NONE, DFD_PRO_BN_ACT); });
    } else {
        DISPATCH_ACT_PW(act, { if (mp == DFD_PRO_AFFINE2) LAUNCH_TN(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT_GATE); else LAUNCH_TN(DFD_PRO_NONE, DFD_PRO_BN_ACT_GATE); });
    }
#undef LAUNCH_TN
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, splits, (long)Ni * Nj, dw, accumulate, st);
}

// ---- weight gradient of a dense k x k convolution: dw[Cout][k*k*C] = sum_m P(p)[m][co] * im2col(Q(x))[m][(tap, c)],
// the im2col operand gathered inside the TN kernel
template <typename T>
static int conv_tn_t(const void* p, const dfd_prologue* pro_p, int Cout, const void* x, const dfd_dwconv_shape* s,
                     const float* in_bnstate, int in_act, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    const int M = s->N * s->Ho * s->Wo, Ni = Cout, Nj = s->k * s->k * s->C;
    int it, jt, splits, rps;
    tn_plan(sizeof(T) == 2 ? DFD_BF16 : DFD_F32, M, Ni, Nj, &it, &jt, &splits, &rps);
    if ((size_t)(splits + splits / 32 + 2) * Ni * Nj * 4 > ws_bytes) return DFD_EWORKSPACE;
    constexpr int BMK = (sizeof(T) == 2) ? 64 : 32;
    constexpr int ROWB = (sizeof(T) == 2) ? 256 : TN_F32_ROW;
    const int lds = 2 * 2 * BMK * ROWB + 5 * TN_B * 4;
    const ProArgs pp = pro_args(pro_p);
    ProArgs pq{nullptr, in_bnstate, nullptr, 1, 0u, -1};
    const int mp = pro_p ? pro_p->mode : DFD_PRO_NONE;
    if (!(mp == DFD_PRO_NONE || mp == DFD_PRO_AFFINE2)) return DFD_EUNSUPPORTED;
    ConvArgs cv{s->H, s->W, s->C, s->Ho, s->Wo, s->k, s->stride, s->pad_top, s->pad_left,
                make_magic(s->Ho * s->Wo), make_magic(s->Wo), make_magic(s->C), make_magic(s->k)};
    dim3 grid(it * jt, splits);
#define LAUNCH_CTN(PP, PQ)                                                                                                  \
    hipLaunchKernelGGL((k_pw_tn<T, PP, PQ, ACT, true>), grid, dim3(DFD_THREADS), lds, st, (const T*)p, pp, Ni, (const T*)x, pq, \
                       Nj, M, it, jt, rps, 0, ws, cv)
    if (!in_bnstate) {
        constexpr int ACT = DFD_ACT_NONE;
        if (mp == DFD_PRO_AFFINE2) LAUNCH_CTN(DFD_PRO_AFFINE2, DFD_PRO_NONE); else LAUNCH_CTN(DFD_PRO_NONE, DFD_PRO_NONE);
    } else {
        DISPATCH_ACT_PW(in_act, { if (mp == DFD_PRO_AFFINE2) LAUNCH_CTN(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT); else LAUNCH_CTN(DFD_PRO_NONE, DFD_PRO_BN_ACT); });
    }
#undef LAUNCH_CTN
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    // dw is in GEMM layout: the caller's dfd_conv_weight_perm reads it next, so the sum is not left to an open batch
    return dfd_launch_sum_partials(ws, splits, (long)Ni * Nj, dw, accumulate, st, false);
}

extern "C" size_t dfd_conv_wgrad_ws(const dfd_dwconv_shape* s, int Cout) {
    if (!s || Cout < 1) return 0;
    const size_t a = dfd_pwconv_wgrad_ws(s->N * s->Ho * s->Wo, Cout, s->k * s->k * s->C), b = dfd_conv3_wgrad_ws(s, Cout);
    return a > b ? a : b;
}
extern "C" int dfd_conv_wgrad(int dtype, const void* p, const dfd_prologue* pro_p, int Cout, const void* x,
                              const dfd_dwconv_shape* s, const float* in_bnstate, int in_act, float* dw, int accumulate,
                              float* ws, size_t ws_bytes, dfd_stream stream) {
    if (!p || !x || !s || !dw || !ws || Cout < 8 || Cout % 8 || s->C < 8 || s->C % 8 || !pro_ok(pro_p)) return DFD_EINVAL;
    if (s->N < 1 || s->Ho < 1 || s->Wo < 1 || (long)s->N * s->Ho * s->Wo > (1l << 30)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {                             // 3x3 stride 1: direct kernel, output block resident in registers
        const int rc = dfd_conv3_wgrad(p, pro_p, Cout, x, s, in_bnstate, in_act, dw, accumulate, ws, ws_bytes, st);
        if (rc != DFD_EUNSUPPORTED) return rc;
        return conv_tn_t<bf16>(p, pro_p, Cout, x, s, in_bnstate, in_act, dw, accumulate, ws, ws_bytes, st);
    }
    if (dtype == DFD_F32) return conv_tn_t<float>(p, pro_p, Cout, x, s, in_bnstate, in_act, dw, accumulate, ws, ws_bytes, st);
    return DFD_EINVAL;
}

extern "C" int dfd_pwconv_wgrad(int dtype, const void* p, const dfd_prologue* pro_p, int Ni, const void* q,
                                const dfd_prologue* pro_q, int Nj, int M, float* dw, int accumulate, float* ws,
                                size_t ws_bytes, dfd_stream stream) {
    if (!p || !q || !dw || !ws || M < 1 || Ni < 8 || Nj < 8 || Ni % 8 || Nj % 8 || !pro_ok(pro_p) || !pro_ok(pro_q))
        return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        // large-M layers with a narrow operand: wave-autonomous kernel (dfd_pwtnw.hip)
        const int rc = dfd_pw_tnw(p, pro_p, Ni, q, pro_q, Nj, M, dw, accumulate, ws, ws_bytes, st);
        if (rc != DFD_EUNSUPPORTED) return rc;
    }
    if (dtype == DFD_BF16) return pw_tn_t<bf16>(p, pro_p, Ni, q, pro_q, Nj, M, dw, accumulate, ws, ws_bytes, st);
    if (dtype == DFD_F32) return pw_tn_t<float>(p, pro_p, Ni, q, pro_q, Nj, M, dw, accumulate, ws, ws_bytes, st);
    return DFD_EINVAL;
}
