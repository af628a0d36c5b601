// dfd_pwntd.hip — 1x1 convolution, NT form, for the MID-SIZE layers: 12.5 k .. 64 k rows with K or N in the hundreds
//
//     out[M][Nout] = P(a)[M][K] * w[Nout][K]^T          (EfficientNet-B0 blocks 5-15 + head at batch 256: 14 x 14 and 7 x 7 maps)
//
// These launches carry little data (30-100 MB) and little arithmetic (2-12 GFLOP); what they cost in k_pw_nt (dfd_pwconv.hip) is
// LATENCY: a 128 x 128 tile walks K in 64-wide steps behind a two-deep register prefetch, one or two workgroups per CU, and every
// step waits for loads requested one step earlier — 20-60 us per launch where the bytes need 5-12.  This kernel keeps the
// operands of two to three K steps in flight per workgroup without holding them in registers:
//   * one 64-row x BN-column tile per workgroup (BN = 16 NT <= 192: narrow outputs are ONE column tile, so the activation operand
//     is read and its prologue evaluated once), 4 waves, wave w owns rows 16 w .. 16 w + 15 and ALL columns;
//   * a ring of NS stages in LDS, a stage = [64 rows of a | (AFFINE2: of a2) | BN rows of w] x 64 k in 128-byte rows whose 16-byte
//     chunks are XOR-swizzled by the row, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass; the
//     swizzle is applied to the per-lane SOURCE address because a DMA piece's destination is lane-linear, as in dfd_gemm.hip);
//   * the DMA is issued as inline assembly.  hipcc orders every ds_read behind a pending LDS-DMA it knows of with s_waitcnt
//     vmcnt(0), which would serialise the ring; hidden from the compiler the requests are waited for by counted
//     waits written here: stage kt is awaited with vmcnt(J x stages requested after it).  The compiler's own vmcnt bookkeeping
//     stays safe (the counter retires in order: a wait it computes without knowing the extra requests can only wait longer);
//   * the producer's BN + activation + squeeze-excite gate (or the BN-backward map of two tensors) is applied to the A fragment
//     between its ds_read and the MFMAs — each element once (a fragment belongs to one wave), with the per-channel coefficients
//     and the tile's gate rows copied to LDS by the same DMA queue at the start;
//   * one tile per workgroup (no persistence): the epilogue overlays the ring — accumulators -> wave-private patch -> 16-byte
//     row-major stores (+ residual), BN statistics from the rounded values, ONE partial row per row tile.
// Same arithmetic as k_pw_nt per element (prologue via apply_pro_v, products accumulated over k in ascending 32-wide MFMA steps);
// the statistics' partial rows are cut differently, so sums agree to rounding, not bit for bit.
#include "dfd_pw.h"
#include "dfd_dwm.h"

#define ND_BM 64
#define ND_RED_BYTES (DFD_THREADS * 16 * 4)

typedef __attribute__((address_space(3))) unsigned char nd_lds_u8;

// one LDS-DMA piece: lane l's 16 bytes at `src` land at lds_addr + 16 l (M0 carries the wave-uniform LDS address).  M0 is not on the
// clobber list — hipcc rejects it as a reserved register — and the compiler itself never touches M0 in these kernels (checked in the
// -S output: every `m0` is one of the s_mov_b32 below); re-check after adding anything that makes it use M0 (LDS-DMA builtins, movrel).
__device__ __forceinline__ void nd_dma16(const void* src, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void nd_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }
__device__ __forceinline__ unsigned nd_lds_addr(const void* p) {
    return (unsigned)reinterpret_cast<uintptr_t>((nd_lds_u8*)p);
}

template <int NT, int PRO> struct NdGeom {
    static constexpr int BN = 16 * NT;
    static constexpr int WPC = (2 * NT + 3) / 4;                        // W pieces (8 rows x 128 B) per wave and stage
    static constexpr int JA = PRO == DFD_PRO_AFFINE2 ? 4 : 2;           // A (+ A2) pieces per wave and stage
    static constexpr int J = JA + WPC;                                  // DMA requests per wave and stage
    static constexpr int WOFF = JA * 4096;                              // byte offset of the W rows inside a stage
    static constexpr int STAGE = WOFF + WPC * 4096;
    static constexpr int OROW = BN * 2 + 16;                            // epilogue patch row
    static constexpr int EPI = (4 * 16 * OROW + 15) / 16 * 16 + ND_RED_BYTES;
};

template <int NT, int PRO, int ACT, bool RES, bool STATS>
__global__ void __launch_bounds__(DFD_THREADS, 2)
k_pw_ntd(const unsigned short* __restrict__ a, ProArgs pa, const unsigned short* __restrict__ w, unsigned short* __restrict__ out,
         const unsigned short* __restrict__ res, int M, int K, int Nout, int m_tiles, int n_tiles, int ns, int coff, int goff,
         float* __restrict__ partials, int dbg) {
    using G = NdGeom<NT, PRO>;
    constexpr int BN = G::BN, WPC = G::WPC, J = G::J, OROW = G::OROW;
    constexpr int NROW = PRO == DFD_PRO_AFFINE2 ? 3 : 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware order: the column tiles of one row tile run back to back on one XCD (they share the A rows in its L2)
    const int total = m_tiles * n_tiles;
    int lin = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mt = lin / n_tiles, nt = lin - mt * n_tiles;
    const int m0 = mt * ND_BM, n0 = nt * BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = (K + 63) >> 6;
    const unsigned lds0 = nd_lds_addr(smem);
    const unsigned short* a2 = reinterpret_cast<const unsigned short*>(pa.a2);

    // ---- the tile's tables -> LDS, first in the DMA queue (every later wait covers them): prologue coefficients [NROW][K] and the
    // gate rows of the images this row tile touches
    int img0 = 0;
    if constexpr (PRO != DFD_PRO_NONE) {
        const int chunks = NROW * K / 4;                                // 16-byte chunks (K % 4 == 0)
        for (int p = wave; p * 64 < chunks; p += 4) {
            int c = p * 64 + lane;
            if (c >= chunks) c = chunks - 1;
            nd_dma16(pa.coef + c * 4, lds0 + coff + p * 1024);
        }
        if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
            img0 = pro_image(pa, m0);
            const int mlast = m0 + ND_BM - 1 < M ? m0 + ND_BM - 1 : M - 1;
            const int gch = (pro_image(pa, mlast) - img0 + 1) * K / 4;
            const float* gsrc = pa.gate + (long)img0 * K;
            for (int p = wave; p * 64 < gch; p += 4) {
                int c = p * 64 + lane;
                if (c >= gch) c = gch - 1;
                nd_dma16(gsrc + c * 4, lds0 + goff + p * 1024);
            }
        }
    }

    // ---- DMA sources: piece p = j * 4 + wave fills rows 8 p .. 8 p + 7 of its region; lane l -> row 8 p + (l >> 3), slot l & 7,
    // which holds source chunk (l & 7) ^ (row & 7); byte offsets of the rows relative to the operand
    const int cs = (lane & 7) ^ ((lane >> 3) & 7);
    unsigned offA[2], offW[WPC];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        const int m = m0 + r < M ? m0 + r : M - 1;
        offA[j] = (unsigned)m * (unsigned)K * 2u;
    }
#pragma unroll
    for (int j = 0; j < WPC; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        const int n = n0 + r < Nout ? n0 + r : Nout - 1;
        offW[j] = (unsigned)n * (unsigned)K * 2u;
    }
    auto request = [&](int kt, int buf) {
        int k = kt * 64 + cs * 8;
        if (k >= K) k = K - 8;                                          // (chunks past K: any valid bytes; the A fragment is zeroed there)
        const unsigned sb = lds0 + buf * G::STAGE + wave * 1024;
        const char* ab = reinterpret_cast<const char*>(a) + k * 2;
        const char* wb = reinterpret_cast<const char*>(w) + k * 2;
        if (!(dbg & 8)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) nd_dma16(ab + offA[j], sb + j * 4096);
        }
        if constexpr (PRO == DFD_PRO_AFFINE2) {
            const char* a2b = reinterpret_cast<const char*>(a2) + k * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) nd_dma16(a2b + offA[j], sb + 8192 + j * 4096);
        }
        if (!(dbg & 4)) {
#pragma unroll
        for (int j = 0; j < WPC; ++j) nd_dma16(wb + offW[j], sb + G::WOFF + j * 4096);
        }
    };

    f32x4_t acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // this lane's fragment row and (BN_ACT_GATE) its gate row in the LDS table
    const int arow = wave * 16 + frow;
    const float* ctab = reinterpret_cast<const float*>(smem + coff);
    const float* grow = nullptr;
    if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
        const int m = m0 + arow < M ? m0 + arow : M - 1;
        grow = reinterpret_cast<const float*>(smem + goff) + (pro_image(pa, m) - img0) * K;
    }

    const int D = ns - 1;                                               // stages in flight
    for (int s = 0; s < D && s < nk; ++s) request(s, s);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most `ahead` younger stages are outstanding
        const int ahead = nk - 1 - kt < D - 1 ? nk - 1 - kt : D - 1;
        if (ahead >= 2) nd_wait_vm<2 * J>();
        else if (ahead == 1) nd_wait_vm<J>();
        else nd_wait_vm<0>();
        __syncthreads();                                                // ... everybody's pieces have, and stage kt - 1's buffer is free
        if (kt + D < nk) {
            int nb = buf + D;
            if (nb >= ns) nb -= ns;
            request(kt + D, nb);
        }
        const unsigned char* sa = smem + buf * G::STAGE;
        const unsigned char* sw = sa + G::WOFF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cc = ks * 4 + fk;
            const int kk = kt * 64 + cc * 8;
            const int sl = (cc ^ (frow & 7)) << 4;
            // every read of the sub-step is requested before the first use: the A fragment(s), the coefficient / gate vectors, then
            // all NT weight fragments — the prologue's arithmetic runs while the weight fragments are still on their way
            uint4 fa = *reinterpret_cast<const uint4*>(sa + arow * 128 + sl);
            uint4 fa2 = fa;
            float c0[8], c1[8], c2[8], gt[8];
            if constexpr (PRO != DFD_PRO_NONE) {
                if constexpr (PRO == DFD_PRO_AFFINE2) fa2 = *reinterpret_cast<const uint4*>(sa + 8192 + arow * 128 + sl);
                load_f32<8>(ctab + kk, c0);
                load_f32<8>(ctab + K + kk, c1);
                if constexpr (PRO == DFD_PRO_AFFINE2) load_f32<8>(ctab + 2 * K + kk, c2);
                if constexpr (PRO == DFD_PRO_BN_ACT_GATE) load_f32<8>(grow + kk, gt);
            }
            uint4 fw[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *reinterpret_cast<const uint4*>(sw + (i * 16 + frow) * 128 + sl);
            __builtin_amdgcn_sched_barrier(0);                          // (left alone the compiler reads two fragments at a time, each MFMA behind an LDS round trip)
            if constexpr (PRO != DFD_PRO_NONE) {
                if constexpr (PRO != DFD_PRO_AFFINE2) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) c2[x] = 0.f;            // (unused by apply_pro_v in these modes)
                }
                if constexpr (PRO != DFD_PRO_BN_ACT_GATE) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) gt[x] = 0.f;
                }
                if (!(dbg & 1)) fa = apply_pro_v<bf16, PRO, ACT, 8>(fa, fa2, c0, c1, c2, gt);
            }
            if (kk >= K) fa = make_uint4(0, 0, 0, 0);
            if (!(dbg & 2)) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[i]), __builtin_bit_cast(bf16x8_t, fa), acc[i], 0, 0, 0);
            } else {
                acc[0][0] += __uint_as_float(fa.x ^ fw[NT - 1].y);        // (timing only: keeps the reads alive)
            }
        }
        if (++buf == ns) buf = 0;
    }

    // ---- epilogue (overlays the ring): accumulators -> this wave's patch [16][BN] -> 16-byte row-major stores
    __syncthreads();
    if (dbg & 16) { if (acc[0][0] == 123.456f) out[0] = 1; return; }
    unsigned char* eb = smem + wave * 16 * OROW;
#pragma unroll
    for (int i = 0; i < NT; ++i)
        *reinterpret_cast<uint2*>(eb + frow * OROW + (i * 16 + fk * 4) * 2) = make_uint2(pack_bf2(acc[i][0], acc[i][1]), pack_bf2(acc[i][2], acc[i][3]));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int CPR = BN / 8;                                         // 16-byte chunks per output row
    constexpr int RL = 64 / CPR;                                        // rows per pass of one wave
    constexpr int NPS = (16 + RL - 1) / RL;
    const int ec = lane % CPR, er = lane / CPR;
    const bool lact = er < RL;
    const int n = n0 + ec * 8;
    const int mw = m0 + wave * 16;
    float s1[8], s2[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) { s1[x] = 0.f; s2[x] = 0.f; }
    {
        uint4 rq[NPS];
        if constexpr (RES) {                                            // all residual vectors of this lane requested before any is used
#pragma unroll
            for (int u = 0; u < NPS; ++u) {
                int m = mw + er + u * RL;
                if (m >= M) m = M - 1;
                rq[u] = *reinterpret_cast<const uint4*>(res + (long)m * Nout + (n < Nout ? n : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < NPS; ++u) {
            const int r = er + u * RL, m = mw + r;
            if (lact && r < 16 && m < M && n < Nout) {
                float v[8];
                Vec<bf16>::unpack(*reinterpret_cast<const uint4*>(eb + r * OROW + ec * 16), v);
                if constexpr (RES) {
                    float q[8];
                    Vec<bf16>::unpack(rq[u], q);
#pragma unroll
                    for (int x = 0; x < 8; ++x) v[x] = round_to<bf16>(v[x] + q[x]);
                }
                if constexpr (STATS) {
#pragma unroll
                    for (int x = 0; x < 8; ++x) { s1[x] += v[x]; s2[x] = fmaf(v[x], v[x], s2[x]); }
                }
                *reinterpret_cast<uint4*>(out + (long)m * Nout + n) = Vec<bf16>::pack(v);
            }
        }
    }
    if constexpr (STATS) {
        // (sum, sumsq) of the tile: lane sums -> LDS -> 16 values per column chunk added over (wave, row lane) in a fixed order
        float* red = reinterpret_cast<float*>(smem + (4 * 16 * OROW + 15) / 16 * 16);
        if (lact) {
            float* p = red + ((wave * RL + er) * CPR + ec) * 16;
            store_f32<8>(p, s1);
            store_f32<8>(p + 8, s2);
        }
        __syncthreads();
        for (int idx = t; idx < CPR * 16; idx += DFD_THREADS) {
            const int c = idx >> 4, x = idx & 15;
            float s = 0.f;
#pragma unroll 4
            for (int r = 0; r < 4 * RL; ++r) s += red[(r * CPR + c) * 16 + x];
            const int col = n0 + c * 8 + (x & 7);
            if (col < Nout) partials[(long)mt * 2 * Nout + (x >> 3) * Nout + col] = s;
        }
    }
}

// ===========================================================================
// host dispatch
// ===========================================================================
template <int NT, int PRO, int ACT, bool RES, bool STATS>
static int ntd_launch(const void* a, const ProArgs& pa, const void* w, void* out, const void* residual, int M, int K, int Nout,
                      float* partials, int imgs, hipStream_t st) {
    using G = NdGeom<NT, PRO>;
    constexpr int NROW = PRO == DFD_PRO_AFFINE2 ? 3 : 2;
    const int m_tiles = (M + ND_BM - 1) / ND_BM, n_tiles = (Nout + G::BN - 1) / G::BN;
    const int nk = (K + 63) / 64;
    // tables behind the ring: whole DMA pieces + 256 bytes (a fragment of the last, partial K step reads up to 63 floats past a row)
    const int ctab = PRO == DFD_PRO_NONE ? 0 : (NROW * K * 4 + 1023) / 1024 * 1024 + 256;
    const int gtab = PRO == DFD_PRO_BN_ACT_GATE ? (imgs * K * 4 + 1023) / 1024 * 1024 + 256 : 0;
    const int tables = ctab + gtab;
    // stages: three with two workgroups per CU when that fits (80 KB each), else two, else one workgroup per CU with up to four
    int ns = dfd_tune_get(DFD_TUNE_NTD_NS);
    if (ns < 2 || ns > 4) {
        const int half = 80 * 1024 - 512;
        if (3 * G::STAGE + tables <= half) ns = 3;
        else if (2 * G::STAGE + tables <= half) ns = 2;
        else { ns = (160 * 1024 - 1024 - tables) / G::STAGE; if (ns > 4) ns = 4; }
    }
    if (ns > nk + 1) ns = nk + 1;
    if (ns < 2) ns = 2;
    int ring = ns * G::STAGE;
    if (ring < G::EPI) ring = G::EPI;
    const int lds = ring + tables;
    if (lds > 160 * 1024) return DFD_EUNSUPPORTED;
    auto kern = k_pw_ntd<NT, PRO, ACT, RES, STATS>;
    struct NdTag;
    dfd_allow_lds_once<NdTag>(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_tiles)), dim3(DFD_THREADS), lds, st, (const unsigned short*)a, pa,
                       (const unsigned short*)w, (unsigned short*)out, (const unsigned short*)residual, M, K, Nout, m_tiles, n_tiles, ns,
                       ring, ring + ctab, partials, dfd_tune_get(DFD_TUNE_DEBUG) >> 8);
    return DFD_CHECK_LAUNCH();
}

template <int NT>
static int ntd_nt(const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M, int K, int Nout,
                  float* partials, int* nparts, hipStream_t st) {
    const ProArgs pa = pro_args(pro);
    const int mode = pro ? pro->mode : DFD_PRO_NONE;
    const bool has_res = residual != nullptr, stats = partials != nullptr;
    int rc = DFD_EUNSUPPORTED;
    if (mode == DFD_PRO_NONE && !has_res) {
        rc = stats ? ntd_launch<NT, DFD_PRO_NONE, DFD_ACT_NONE, false, true>(a, pa, w, out, residual, M, K, Nout, partials, 0, st)
                   : ntd_launch<NT, DFD_PRO_NONE, DFD_ACT_NONE, false, false>(a, pa, w, out, residual, M, K, Nout, partials, 0, st);
    } else if (mode == DFD_PRO_BN_ACT_GATE && pro->act == DFD_ACT_SILU && stats && !has_res) {
        const int imgs = (ND_BM + pa.HW - 2) / pa.HW + 1;               // images a 64-row tile can touch
        rc = ntd_launch<NT, DFD_PRO_BN_ACT_GATE, DFD_ACT_SILU, false, true>(a, pa, w, out, residual, M, K, Nout, partials, imgs, st);
    } else if (mode == DFD_PRO_AFFINE2 && !stats) {
        rc = has_res ? ntd_launch<NT, DFD_PRO_AFFINE2, DFD_ACT_NONE, true, false>(a, pa, w, out, residual, M, K, Nout, partials, 0, st)
                     : ntd_launch<NT, DFD_PRO_AFFINE2, DFD_ACT_NONE, false, false>(a, pa, w, out, residual, M, K, Nout, partials, 0, st);
    }
    if (rc == DFD_OK && stats) *nparts = (M + ND_BM - 1) / ND_BM;
    return rc;
}

// column-tile width (in 16-column fragments) for an output of Nout channels: the narrowest instance that covers it in the fewest tiles
static int ntd_pick(int Nout) {
    const int wmax = dfd_tune_get(DFD_TUNE_NTD_MAXN) > 0 ? dfd_tune_get(DFD_TUNE_NTD_MAXN) : 192;   // (A/B: narrower column tiles)
    const int tiles = (Nout + wmax - 1) / wmax;
    const int per = ((Nout + tiles - 1) / tiles + 15) / 16;
    return per <= 4 ? 4 : per <= 6 ? 6 : per <= 8 ? 8 : 12;
}

int dfd_pw_ntd(int dtype, const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M, int K,
               int Nout, float* partials, int pcap, int* nparts, hipStream_t st) {
    if (!(dfd_tune_get(DFD_TUNE_PW_NTD) & 1) || dtype != DFD_BF16) return DFD_EUNSUPPORTED;
    const int m_tiles = (M + ND_BM - 1) / ND_BM;
    const int cap = partials ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    // rows: enough tiles to give most CUs one, few enough for one partial row per tile; K: at least one full step
    if (m_tiles < dfd_tune_get(DFD_TUNE_NTD_MINT) || m_tiles > cap || K < 64 || (long)M * K >= (1l << 30)) return DFD_EUNSUPPORTED;
    switch (ntd_pick(Nout)) {
        case 4: return ntd_nt<4>(a, pro, w, out, residual, M, K, Nout, partials, nparts, st);
        case 6: return ntd_nt<6>(a, pro, w, out, residual, M, K, Nout, partials, nparts, st);
        case 8: return ntd_nt<8>(a, pro, w, out, residual, M, K, Nout, partials, nparts, st);
        default: return ntd_nt<12>(a, pro, w, out, residual, M, K, Nout, partials, nparts, st);
    }
}
// which column-tile width serves this shape (0: not this kernel) — tests assert the path they name
extern "C" int dfd_pw_ntd_plan(int M, int K, int Nout) {
    const int m_tiles = (M + ND_BM - 1) / ND_BM;
    if (!(dfd_tune_get(DFD_TUNE_PW_NTD) & 1) || m_tiles < dfd_tune_get(DFD_TUNE_NTD_MINT) || m_tiles > DFD_MAX_PARTIALS || K < 64 || (long)M * K >= (1l << 30)) return 0;
    return 16 * ntd_pick(Nout);
}
