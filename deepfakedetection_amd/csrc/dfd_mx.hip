// dfd_mx.hip — fp8 weights for the FasterViT Linear layers (BASELINE config 5: "bf16/fp8 weights ... hierarchical attention on
// CDNA4 fp8 MFMA"; reference call sites trainers/fastervit.py:271, :235, :371-375 — the forward of the third-party module).
//
// Format: OCP microscaling (MX) FP8.  Elements are OCP e4m3fn (gfx950's native fp8, NOT MI300X's fnuz), every 32 consecutive
// elements along K share one E8M0 scale byte (value 2^(byte-127)):
//      e    = clamp(floor(log2(max|v|)) - 8, -127, 127)        (8 = exponent of e4m3's largest power of two, 256 <= 448)
//      q[i] = e4m3_rne_saturate(v[i] * 2^-e),   scale byte = e + 127,   max|v| == 0 -> byte 0, q = 0
// which is the only fp8 form that runs above the bf16 MFMA rate on this part: v_mfma_scale_f32_16x16x128_f8f6f4 consumes both
// operands in that format (there is no bf16 x fp8 MFMA), so the activation operand is quantised the same way on the fly
// (k_mx_quant_rows, after the producer's prologue) — W8A8 with f32 accumulation; backward stays bf16 on the DEQUANTISED
// weights (straight-through for the master weights).
//
// Kernels
//   k_mx_quant_weights_multi   f32 master [N][K] -> q [N][K] u8, scale [N][K/32] u8, and the dequantised [K][N] copy in the
//                              activation dtype for the data-gradient GEMM; one launch for up to 32 weights (jobs by value)
//   k_mx_quant_rows<T, ACT>    activations [M][K] (T = bf16 | f32) [-> act(c0*x + c1)] -> q, scale; one thread per 32-block,
//                              64 B in, 33 B out: a streaming pass at HBM rate
//   k_mx_gemm                  out[M][N] = deq(aq) . deq(wq)^T, f32 accumulate, T out.  4 waves as 2 (M) x 2 (N), wave tile
//                              32 x 64, fragments straight from global memory in MFMA layout (see mx_load: a lane holds two
//                              16-byte pieces of row l & 15, the wave one cache line per row and k-step) plus a scale byte.
//                              The WEIGHT is
//                              passed as the MFMA's A operand and the activation as B, so a lane's four accumulator registers
//                              are four consecutive output columns of one row: 8-byte stores, 32 B contiguous per row.
//                              Workgroup ids are dealt round-robin over the 8 XCDs by hardware: id -> tile is remapped so
//                              that all N tiles of one M tile run on ONE XCD back to back and share its L2 copy of the
//                              activation rows (the weight, <= 1 MB, is resident in every L2).
// Roofline: HBM — algorithmic bytes M*K + M*K/32 (A) + N*K (W) + M*N*2 (out); the MFMA work (2*M*N*K at the MX rate, 2x bf16)
// is < 10 % of the kernel's time at every FasterViT shape (K = 256..2048, N = 256..2048).
#include "dfd_common.h"
#include "dfd_pw.h"

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// ------------------------------------------------------------------------------------------------ element conversion
__device__ __forceinline__ int mx_block_exp(float amax) {
    // floor(log2(amax)) - 8 from the f32 exponent field (subnormal / zero amax: field 0 -> clamps to -127)
    const int field = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    int e = field - 127 - 8;
    return e < -127 ? -127 : (e > 127 ? 127 : e);
}
__device__ __forceinline__ float mx_inv_scale(int e) {      // 2^-e as an f32 (e in [-127, 127]; 2^127 and 2^-126.. are normal,
    return __uint_as_float((unsigned)(127 - e) << 23);      // e = 127 gives field 0 = 0.0: such a block holds inf/nan only)
}
// four f32 (already multiplied by 2^-e) -> four e4m3fn bytes, RNE, saturating at +-448
__device__ __forceinline__ unsigned mx_pack4(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
    b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f);
    d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
// e4m3fn byte -> f32 (exact)
__device__ __forceinline__ float e4m3_to_f(unsigned b) {
    const unsigned s = (b & 0x80u) << 24, ex = (b >> 3) & 0xfu, m = b & 7u;
    float v;
    if (ex == 0) v = (float)m * 0.001953125f;                          // subnormal: m * 2^-9
    else v = __uint_as_float(((ex + 120u) << 23) | (m << 20));         // 2^(ex-7) * (1 + m/8)
    return __uint_as_float(__float_as_uint(v) | s);
}

// quantise 32 f32 values held by one thread; returns the scale byte, q = 8 packed words
// A block that holds a NaN or an infinity becomes a NaN block — scale byte 0xff (the OCP MX NaN scale) and every element the
// e4m3fn NaN 0x7f — so that a diverged run still shows non-finite logits with fp8 weights, as it does in bf16 (fmaxf / fminf
// drop NaNs: without this the block came out as +-448 * 2^e and the loss stayed finite).
__device__ __forceinline__ unsigned mx_quant32(const float (&v)[32], unsigned (&q)[8]) {
    float amax = 0.f;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        amax = __builtin_fmaxf(amax, __builtin_fabsf(v[i]));
        bad |= !(__builtin_fabsf(v[i]) <= 3.4028234663852886e38f);      // NaN or infinity
    }
    if (bad) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] = 0x7f7f7f7fu;
        return 0xffu;
    }
    const int e = mx_block_exp(amax);
    const float inv = mx_inv_scale(e);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = mx_pack4(v[4 * i] * inv, v[4 * i + 1] * inv, v[4 * i + 2] * inv, v[4 * i + 3] * inv);
    return (unsigned)(e + 127);
}

// ------------------------------------------------------------------------------------------------ weights
#define MX_JOBS_PER_LAUNCH 32
struct MxJobs { dfd_mx_job j[MX_JOBS_PER_LAUNCH]; };
__global__ void __launch_bounds__(256)
k_mx_quant_weights_multi(MxJobs jobs) {
    const dfd_mx_job& jb = jobs.j[blockIdx.y];
    const int kb_per_row = jb.K >> 5;
    const long blocks = (long)jb.N * kb_per_row;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= blocks) return;
    const int n = (int)(t / kb_per_row), kb = (int)(t - (long)n * kb_per_row);
    const float* src = jb.src + (long)n * jb.K + kb * 32;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
        const float4 f = *reinterpret_cast<const float4*>(src + i);
        v[i] = f.x; v[i + 1] = f.y; v[i + 2] = f.z; v[i + 3] = f.w;
    }
    unsigned q[8];
    const unsigned sb = mx_quant32(v, q);
    uint4* dq = reinterpret_cast<uint4*>(jb.q + (long)n * jb.K + kb * 32);
    dq[0] = make_uint4(q[0], q[1], q[2], q[3]);
    dq[1] = make_uint4(q[4], q[5], q[6], q[7]);
    jb.scale[(long)n * kb_per_row + kb] = (uint8_t)sb;
    if (jb.kn) {
        // 2^(sb - 127): sb == 0 is 2^-127 (a subnormal f32), sb == 0xff the NaN scale
        const float sc = sb == 0xffu ? __uint_as_float(0x7fc00000u) : (sb ? __uint_as_float(sb << 23) : __uint_as_float(0x00400000u));
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float d = sb == 0xffu ? sc : e4m3_to_f((q[i >> 2] >> (8 * (i & 3))) & 0xffu) * sc;
            const long at = (long)(kb * 32 + i) * jb.N + n;
            if (jb.kn_dtype == DFD_BF16) reinterpret_cast<unsigned short*>(jb.kn)[at] = f2bf(d);   // exact: 4 significant bits
            else reinterpret_cast<float*>(jb.kn)[at] = d;
        }
    }
}

extern "C" int dfd_mx_quant_weights_multi(const dfd_mx_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) {
        const dfd_mx_job& j = jobs[i];
        if (!j.src || !j.q || !j.scale || j.N < 1 || j.K < 32 || (j.K & 31)) return DFD_EINVAL;
        if (j.kn && j.kn_dtype != DFD_BF16 && j.kn_dtype != DFD_F32) return DFD_EINVAL;
    }
    for (int base = 0; base < njobs; base += MX_JOBS_PER_LAUNCH) {
        MxJobs pj;
        const int cnt = njobs - base < MX_JOBS_PER_LAUNCH ? njobs - base : MX_JOBS_PER_LAUNCH;
        long most = 0;
        for (int i = 0; i < cnt; ++i) {
            pj.j[i] = jobs[base + i];
            const long t = (long)pj.j[i].N * (pj.j[i].K >> 5);
            if (t > most) most = t;
        }
        hipLaunchKernelGGL(k_mx_quant_weights_multi, dim3((unsigned)((most + 255) / 256), cnt), dim3(256), 0, (hipStream_t)stream, pj);
    }
    return DFD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ activations
template <typename T, int ACT, bool PRO>
__global__ void __launch_bounds__(256)
k_mx_quant_rows(const T* __restrict__ a, const float* __restrict__ coef, uint8_t* __restrict__ q, uint8_t* __restrict__ scale,
                long blocks, int K) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= blocks) return;
    const int kb_per_row = K >> 5;
    const long m = t / kb_per_row;
    const int k0 = (int)(t - m * kb_per_row) * 32;
    float v[32];
    if constexpr (sizeof(T) == 2) {
        const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(a) + m * K + k0);
        const uint4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];         // four loads in flight
        float u[8];
        Vec<bf16>::unpack(r0, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = u[i];
        Vec<bf16>::unpack(r1, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[8 + i] = u[i];
        Vec<bf16>::unpack(r2, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[16 + i] = u[i];
        Vec<bf16>::unpack(r3, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[24 + i] = u[i];
    } else {
        const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a) + m * K + k0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 f = p[i];
            v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
        }
    }
    if constexpr (PRO) {
        // the producer's BatchNorm / bias + activation, as the GEMM prologues apply it: act(c0[k] * x + c1[k]), then
        // rounded to T (the value an unfused pipeline would have stored and the bf16 GEMM multiplies)
        float c0[32], c1[32];
        load_f32<32>(coef + k0, c0);
        load_f32<32>(coef + K + k0, c1);
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = round_to<T>(act_fwd<ACT>(fmaf(c0[i], v[i], c1[i])));
    }
    unsigned w[8];
    const unsigned sb = mx_quant32(v, w);
    uint4* dq = reinterpret_cast<uint4*>(q + m * K + k0);
    dq[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dq[1] = make_uint4(w[4], w[5], w[6], w[7]);
    scale[t] = (uint8_t)sb;
}

extern "C" int dfd_mx_quant_rows(int dtype, const void* a, const dfd_prologue* pro, uint8_t* q, uint8_t* scale, long M, int K,
                                 dfd_stream stream) {
    if (!a || !q || !scale || M < 1 || K < 32 || (K & 31)) return DFD_EINVAL;
    const int mode = pro ? pro->mode : DFD_PRO_NONE;
    if (mode != DFD_PRO_NONE && mode != DFD_PRO_BN_ACT) return DFD_EUNSUPPORTED;
    if (mode == DFD_PRO_BN_ACT && !pro->coef) return DFD_EINVAL;
    const long blocks = M * (K >> 5);
    const unsigned grid = (unsigned)((blocks + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    const float* coef = pro ? pro->coef : nullptr;
#define MXQ(T)                                                                                                        \
    if (mode == DFD_PRO_NONE) {                                                                                       \
        hipLaunchKernelGGL((k_mx_quant_rows<T, DFD_ACT_NONE, false>), dim3(grid), dim3(256), 0, st, (const T*)a, coef, q, scale, blocks, K); \
    } else {                                                                                                          \
        DISPATCH_ACT_PW(pro->act, hipLaunchKernelGGL((k_mx_quant_rows<T, ACT, true>), dim3(grid), dim3(256), 0, st, (const T*)a, coef, q, \
                                                      scale, blocks, K));                                            \
    }
    if (dtype == DFD_BF16) { MXQ(bf16) }
    else if (dtype == DFD_F32) { MXQ(float) }
    else return DFD_EINVAL;
#undef MXQ
    return DFD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------------ GEMM
#define MX_WM 2                 // 16-row blocks of the activation per wave
#define MX_WN 4                 // 16-column blocks of the weight per wave
#define MX_BM (2 * 16 * MX_WM)  // 64 rows per workgroup
#define MX_BN (2 * 16 * MX_WN)  // 128 columns per workgroup

struct MxFrag { i32x8_t v; int s; };

// Operand layout of v_mfma_scale_f32_16x16x128_f8f6f4, measured on MI355X with one-hot data (scripts/probes/mx_layout_probe.hip;
// neither guide documents it): lane l = 16 g + i (i = row of A / column of B, g = 0..3) holds 32 bytes; its bytes 0-15 and
// 16-31 belong to DIFFERENT 32-element scale blocks, each shared with the lane group next to it:
//      k(g, byte b) = 64 (g >> 1) + 32 (b >> 4) + 16 (g & 1) + (b & 15)
// and the scale register of lane group g is applied to block  k / 32 == ((g & 1) << 1) | (g >> 1)   (groups 0, 2, 1, 3 carry
// the scales of blocks 0, 1, 2, 3).  With K-contiguous rows in memory a lane therefore loads two 16-byte pieces 32 bytes
// apart; the wave still covers one full 128-byte line per row and k-step.
__device__ __forceinline__ MxFrag mx_load(const uint8_t* __restrict__ q, const uint8_t* __restrict__ sc, long row, int K, int ks, int g) {
    MxFrag f;
    const uint8_t* base = q + row * K + ks * 128 + 64 * (g >> 1) + 16 * (g & 1);
    const uint4 lo = *reinterpret_cast<const uint4*>(base), hi = *reinterpret_cast<const uint4*>(base + 32);
    f.v = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    f.s = (int)sc[row * (K >> 5) + ks * 4 + (((g & 1) << 1) | (g >> 1))];
    return f;
}

template <typename T>
__global__ void __launch_bounds__(256)
k_mx_gemm(const uint8_t* __restrict__ aq, const uint8_t* __restrict__ as, const uint8_t* __restrict__ wq, const uint8_t* __restrict__ ws,
          T* __restrict__ out, int M, int K, int N, int m_tiles, int n_tiles) {
    // XCD-aware order: workgroup id -> (xcd, slot); each XCD owns a contiguous range of tiles, n fastest
    const int total = m_tiles * n_tiles;
    const int id = blockIdx.x;
    const int per = (total + 7) >> 3;
    const int tile = (id & 7) * per + (id >> 3);
    if (tile >= total) return;
    const int mt = tile / n_tiles, nt = tile - mt * n_tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i16 = lane & 15, kb = lane >> 4;
    const int m0 = mt * MX_BM + (wave >> 1) * (16 * MX_WM);
    const int n0 = nt * MX_BN + (wave & 1) * (16 * MX_WN);
    long arow[MX_WM], wrow[MX_WN];
#pragma unroll
    for (int i = 0; i < MX_WM; ++i) { const int r = m0 + 16 * i + i16; arow[i] = r < M ? r : M - 1; }      // clamped: always valid
#pragma unroll
    for (int j = 0; j < MX_WN; ++j) { const int c = n0 + 16 * j + i16; wrow[j] = c < N ? c : N - 1; }
    f32x4_t acc[MX_WN][MX_WM];
#pragma unroll
    for (int j = 0; j < MX_WN; ++j)
#pragma unroll
        for (int i = 0; i < MX_WM; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int ksteps = K >> 7;
    MxFrag a[MX_WM], w[MX_WN];
#pragma unroll
    for (int i = 0; i < MX_WM; ++i) a[i] = mx_load(aq, as, arow[i], K, 0, kb);
#pragma unroll
    for (int j = 0; j < MX_WN; ++j) w[j] = mx_load(wq, ws, wrow[j], K, 0, kb);
    for (int ks = 0; ks < ksteps; ++ks) {
        MxFrag an[MX_WM], wn[MX_WN];
        const int kn = ks + 1 < ksteps ? ks + 1 : ks;                   // last step reloads itself (unused)
#pragma unroll
        for (int i = 0; i < MX_WM; ++i) an[i] = mx_load(aq, as, arow[i], K, kn, kb);
#pragma unroll
        for (int j = 0; j < MX_WN; ++j) wn[j] = mx_load(wq, ws, wrow[j], K, kn, kb);
#pragma unroll
        for (int j = 0; j < MX_WN; ++j)
#pragma unroll
            for (int i = 0; i < MX_WM; ++i)
                // D[n][m] += W[n][k] * A[m][k]: weight = A operand (rows n), activation = B operand (columns m);
                // cbsz = blgp = 0: both e4m3; opsel 0: scale byte 0 of each lane's scale register
                acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w[j].v, a[i].v, acc[j][i], 0, 0, 0, w[j].s, 0, a[i].s);
#pragma unroll
        for (int i = 0; i < MX_WM; ++i) a[i] = an[i];
#pragma unroll
        for (int j = 0; j < MX_WN; ++j) w[j] = wn[j];
    }
    // C/D layout of the 16x16 MFMA: column = lane & 15 (= activation row m), rows (lane >> 4) * 4 + reg (= weight rows n)
#pragma unroll
    for (int i = 0; i < MX_WM; ++i) {
        const int m = m0 + 16 * i + i16;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < MX_WN; ++j) {
            const int n = n0 + 16 * j + kb * 4;
            if (n >= N) continue;                                        // N % 4 == 0 (checked by the host)
            const f32x4_t c = acc[j][i];
            if constexpr (sizeof(T) == 2) {
                uint2 pk;
                pk.x = pack_bf2(c[0], c[1]);
                pk.y = pack_bf2(c[2], c[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + (long)m * N + n) = pk;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (long)m * N + n) = make_float4(c[0], c[1], c[2], c[3]);
            }
        }
    }
}

extern "C" int dfd_mx_gemm(const uint8_t* aq, const uint8_t* ascale, const uint8_t* wq, const uint8_t* wscale, int dtype_out, void* out,
                           long M, int K, int N, dfd_stream stream) {
    if (!aq || !ascale || !wq || !wscale || !out || M < 1 || N < 4 || K < 128) return DFD_EINVAL;
    if ((K & 127) || (N & 3) || M > (1l << 30)) return DFD_EUNSUPPORTED;
    const int m_tiles = (int)((M + MX_BM - 1) / MX_BM), n_tiles = (N + MX_BN - 1) / MX_BN;
    const long total = (long)m_tiles * n_tiles;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    hipStream_t st = (hipStream_t)stream;
    if (dtype_out == DFD_BF16)
        hipLaunchKernelGGL((k_mx_gemm<bf16>), dim3(grid), dim3(256), 0, st, aq, ascale, wq, wscale, (bf16*)out, (int)M, K, N, m_tiles, n_tiles);
    else if (dtype_out == DFD_F32)
        hipLaunchKernelGGL((k_mx_gemm<float>), dim3(grid), dim3(256), 0, st, aq, ascale, wq, wscale, (float*)out, (int)M, K, N, m_tiles, n_tiles);
    else
        return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
