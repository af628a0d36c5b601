// dfd_common.h — shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * activations are NHWC, i.e. a [rows = N*H*W][C] row-major matrix, element
//     type T = float or bf16 (storage: unsigned short); C % 8 == 0.
//   * one lane moves one 16-byte channel vector: 8 bf16 or 4 f32 (VEC).
//   * statistics, BN coefficients, SE gates, weights' master copies are f32.
//   * wave = 64 lanes, workgroup = 256 threads (4 waves) unless noted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/dfd_hip.h"

#define DFD_THREADS 256

// Host-side kernel-attribute caches ("kernel-attribute caches behind std::call_once"): the forward runs on the main
// thread and the backward on autograd's thread, and the first launch of an instantiation may happen inside a stream capture.
// `Tag` names the call site, so every (site, kernel instantiation) pair owns one flag.
template <typename Tag, typename KernelT>
static inline void dfd_allow_lds_once(KernelT kern, int bytes) {
    static std::once_flag once;
    std::call_once(once, [&] { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
}
template <typename Tag, typename KernelT>
static inline int dfd_kernel_regs_once(KernelT kern, int fallback) {
    static std::once_flag once;
    static int regs = 0;
    std::call_once(once, [&] {
        hipFuncAttributes attr;
        regs = (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(kern)) == hipSuccess && attr.numRegs > 0) ? attr.numRegs : fallback;
    });
    return regs;
}

struct bf16 { unsigned short x; };  // storage tag; arithmetic is always f32

__device__ __forceinline__ float bf2f(unsigned short h) {
    return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
}
// round an f32 to what it would be after a store/load through T
template <typename T> __device__ __forceinline__ float round_to(float f);
template <> __device__ __forceinline__ float round_to<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_to<bf16>(float f) { return bf2f(f2bf(f)); }

// ---------------------------------------------------------------------------
// 16-byte channel vectors
// ---------------------------------------------------------------------------
template <typename T> struct Vec;
template <> struct Vec<float> {
    static constexpr int N = 4;
    __device__ __forceinline__ static void load(const float* p, float (&v)[4]) {
        float4 q = *reinterpret_cast<const float4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    }
    __device__ __forceinline__ static void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ __forceinline__ static void zero(float* p) {
        *reinterpret_cast<float4*>(p) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};
template <> struct Vec<bf16> {
    static constexpr int N = 8;
    __device__ __forceinline__ static void unpack(const uint4& q, float (&v)[8]) {
        v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
        v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
        v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
        v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
    }
    __device__ __forceinline__ static uint4 pack(const float (&v)[8]) {
        uint4 q;
        q.x = pack_bf2(v[0], v[1]); q.y = pack_bf2(v[2], v[3]);
        q.z = pack_bf2(v[4], v[5]); q.w = pack_bf2(v[6], v[7]);
        return q;
    }
    __device__ __forceinline__ static void load(const bf16* p, float (&v)[8]) {
        uint4 q = *reinterpret_cast<const uint4*>(p);
        unpack(q, v);
    }
    __device__ __forceinline__ static void store(bf16* p, const float (&v)[8]) {
        *reinterpret_cast<uint4*>(p) = pack(v);
    }
    __device__ __forceinline__ static void zero(bf16* p) {
        *reinterpret_cast<uint4*>(p) = make_uint4(0, 0, 0, 0);
    }
};

// load VEC consecutive f32 parameters (per-channel coefficients)
template <int N> __device__ __forceinline__ void load_f32(const float* p, float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 4) {
        float4 q = *reinterpret_cast<const float4*>(p + i);
        v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
    }
}
template <int N> __device__ __forceinline__ void store_f32(float* p, const float (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; i += 4)
        *reinterpret_cast<float4*>(p + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
}

// ---------------------------------------------------------------------------
// activations (DFD_ACT_*)
// ---------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float z) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-z));
}
// Exact (erf-based) GELU, nn.GELU() of timm / fastervit.  The normal CDF comes from the complementary error
// function in the Abramowitz-Stegun 7.1.26 form  erfc(x) = (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x), x >= 0
// (absolute error <= 1.5e-7, the size of an f32 ulp near 1): one v_rcp, one v_exp and seven FMAs per element instead
// of libm's erff (~35 instructions), and exp(-z^2/2) is shared with the density in the derivative.  Working with the
// complement keeps the RELATIVE accuracy in the negative tail, where 1 + erf cancels.
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& ez) {
    const float x = fabsf(z) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    ez = __expf(-x * x);                                               // = exp(-z^2 / 2)
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float half_erfc = 0.5f * poly * ez;                          // 0.5 * erfc(|z| / sqrt 2)
    cdf = z >= 0.f ? 1.0f - half_erfc : half_erfc;
}
template <int ACT> __device__ __forceinline__ float act_fwd(float z) {
    if constexpr (ACT == DFD_ACT_SILU) return z * sigmoid_f(z);
    else if constexpr (ACT == DFD_ACT_RELU) return z > 0.f ? z : 0.f;
    else if constexpr (ACT == DFD_ACT_GELU) { float cdf, ez; gelu_parts(z, cdf, ez); return z * cdf; }
    else return z;
}
// d act(z) / dz
template <int ACT> __device__ __forceinline__ float act_grad(float z) {
    if constexpr (ACT == DFD_ACT_SILU) {
        float s = sigmoid_f(z);
        return s * (1.0f + z * (1.0f - s));
    } else if constexpr (ACT == DFD_ACT_RELU) {
        return z > 0.f ? 1.f : 0.f;
    } else if constexpr (ACT == DFD_ACT_GELU) {
        float cdf, ez;
        gelu_parts(z, cdf, ez);
        return fmaf(z * 0.39894228040143268f, ez, cdf);               // Phi(z) + z * phi(z)
    } else {
        return 1.f;
    }
}

// The same activations on a PAIR of values: every multiply / add / FMA is written on float2 so that it becomes one
// v_pk_*_f32 instruction for two elements; only exp and rcp (quarter rate either way) stay per element.  Operation for
// operation the scalar sequence above, so the results are bit-identical to act_fwd / act_grad — a GELU costs ~16 regular
// VALU instructions + 2 transcendentals per element as scalars, 8 + 2 in pairs, and the fused BN + GELU prologues of the
// EfficientFormerV2 / FasterViT kernels are bound by exactly that.
typedef float dfd_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dfd_f2 splat2(float v) { return (dfd_f2){v, v}; }
__device__ __forceinline__ void gelu_parts2(dfd_f2 z, dfd_f2& cdf, dfd_f2& ez) {
    const dfd_f2 x = (dfd_f2){fabsf(z.x), fabsf(z.y)} * splat2(0.70710678118654752f);
    const dfd_f2 den = __builtin_elementwise_fma(splat2(0.3275911f), x, splat2(1.0f));
    const dfd_f2 t = (dfd_f2){__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    const dfd_f2 ea = (-x * x) * splat2(1.44269504088896340736f);            // __expf(v) = exp2(v * log2 e)
    ez = (dfd_f2){__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
    dfd_f2 poly = __builtin_elementwise_fma(t, splat2(1.061405429f), splat2(-1.453152027f));
    poly = __builtin_elementwise_fma(t, poly, splat2(1.421413741f));
    poly = __builtin_elementwise_fma(t, poly, splat2(-0.284496736f));
    poly = __builtin_elementwise_fma(t, poly, splat2(0.254829592f));
    poly = t * poly;
    const dfd_f2 half_erfc = splat2(0.5f) * poly * ez;
    const dfd_f2 upper = splat2(1.0f) - half_erfc;
    cdf = (dfd_f2){z.x >= 0.f ? upper.x : half_erfc.x, z.y >= 0.f ? upper.y : half_erfc.y};
}
template <int ACT> __device__ __forceinline__ dfd_f2 act_fwd2(dfd_f2 z) {
    if constexpr (ACT == DFD_ACT_GELU) { dfd_f2 cdf, ez; gelu_parts2(z, cdf, ez); return z * cdf; }
#ifndef DFD_SILU_SCALAR
    else if constexpr (ACT == DFD_ACT_SILU) {
        // z * rcp(1 + exp(-z)) with the multiply / add / multiply on pairs (v_pk_*_f32); operation for operation act_fwd<SILU>
        const dfd_f2 t = z * splat2(-1.44269504088896340736f);               // __expf(-z) = exp2(-z * log2 e)
        const dfd_f2 d = (dfd_f2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + splat2(1.0f);
        return z * (dfd_f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    }
#endif
    else return (dfd_f2){act_fwd<ACT>(z.x), act_fwd<ACT>(z.y)};
}
template <int ACT> __device__ __forceinline__ dfd_f2 act_grad2(dfd_f2 z) {
    if constexpr (ACT == DFD_ACT_GELU) {
        dfd_f2 cdf, ez;
        gelu_parts2(z, cdf, ez);
        return __builtin_elementwise_fma(z * splat2(0.39894228040143268f), ez, cdf);
    }
#ifndef DFD_SILU_SCALAR
    else if constexpr (ACT == DFD_ACT_SILU) {
        // s * (1 + z * (1 - s)), s = rcp(1 + exp(-z)): act_grad<SILU> operation for operation, on pairs
        const dfd_f2 t = z * splat2(-1.44269504088896340736f);
        const dfd_f2 d = (dfd_f2){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + splat2(1.0f);
        const dfd_f2 sg = (dfd_f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        return sg * (splat2(1.0f) + z * (splat2(1.0f) - sg));
    }
#endif
    else return (dfd_f2){act_grad<ACT>(z.x), act_grad<ACT>(z.y)};
}
// act(scale * v + shift) / act'(..) over an array of N values, in pairs
template <int ACT, int N>
__device__ __forceinline__ void bn_act_array(float (&v)[N], const float (&sc)[N], const float (&sh)[N]) {
    static_assert(N % 2 == 0, "pairs");
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        const dfd_f2 z = __builtin_elementwise_fma((dfd_f2){sc[j], sc[j + 1]}, (dfd_f2){v[j], v[j + 1]}, (dfd_f2){sh[j], sh[j + 1]});
        const dfd_f2 a = act_fwd2<ACT>(z);
        v[j] = a.x; v[j + 1] = a.y;
    }
}

// ---------------------------------------------------------------------------
// channel mapping shared by the row-streaming kernels: thread t owns channel
// vector (vchunk*cvb + t % cvb) and row lane t / cvb; cvb divides C/VEC.
// ---------------------------------------------------------------------------
struct ChanMap {
    int cvb;   // channel vectors per workgroup
    int rpb;   // row lanes per workgroup = 256 / cvb
    int nvc;   // number of channel-vector chunks = (C/VEC) / cvb
};
static inline ChanMap make_chanmap(int C, int vec) {
    int cv = C / vec;
    int cvb = cv;
    if (cv > 64) {
        cvb = 1;
        for (int d = 64; d >= 1; --d) if (cv % d == 0) { cvb = d; break; }
    }
    ChanMap m;
    m.cvb = cvb;
    m.rpb = DFD_THREADS / cvb;
    m.nvc = cv / cvb;
    return m;
}

// block reduction of NV per-thread f32 partial sums over the row lanes that share a channel vector; the result
// lands in the rl == 0 threads.  red must hold DFD_THREADS * NV floats.
// Value i of channel vector vl is summed over the row lanes in ascending order (fixed order: bitwise reproducible)
// by the thread (vl, rl = i mod rpb), so the NV * rpb dependent LDS reads the rl == 0 threads used to do alone
// (up to 512 per thread, ~20 us at the end of EVERY workgroup with statistics) are spread over the whole workgroup.
template <int NV>
__device__ __forceinline__ void reduce_rowlanes(float (&acc)[NV], float* red, int cvb, int rpb, int vl, int rl, bool active) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * DFD_THREADS + t] = active ? acc[i] : 0.f;
    __syncthreads();
    if (active) {
        for (int i = rl; i < NV; i += rpb) {
            const float* col = red + i * DFD_THREADS + vl;
            float s = 0.f;
            int r = 0;
            for (; r + 4 <= rpb; r += 4) {                    // four loads in flight, added in row order
                const float a0 = col[r * cvb], a1 = col[(r + 1) * cvb], a2 = col[(r + 2) * cvb], a3 = col[(r + 3) * cvb];
                s += a0; s += a1; s += a2; s += a3;
            }
            for (; r < rpb; ++r) s += col[r * cvb];
            red[i * DFD_THREADS + vl] = s;                    // slot (i, row lane 0, vl): this thread was its only reader
        }
    }
    __syncthreads();
    if (rl == 0 && active) {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = red[i * DFD_THREADS + vl];
    }
    __syncthreads();
}

// stem convolution on the matrix cores (dfd_stem.hip, bf16); DFD_EUNSUPPORTED: shape not served, the f32-FMA kernels run
int dfd_stem_fwd_mfma(const float* x, const float* w, void* y, const dfd_stem_shape* s, float* partials, int pcap, int* nparts,
                      hipStream_t st);
int dfd_stem_wgrad_mfma(const float* x, const void* dz, const void* yraw, const float* coef, const dfd_stem_shape* s, float* ws,
                        int max_rows, int* rows, hipStream_t st);

#define DFD_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DFD_OK : DFD_ELAUNCH)

// dispatch a runtime activation code to a constexpr int ACT inside the body
#define DISPATCH_ACT(ACTV, ...)                                                      \
    switch (ACTV) {                                                                  \
        case DFD_ACT_NONE: { constexpr int ACT = DFD_ACT_NONE; __VA_ARGS__; } break; \
        case DFD_ACT_SILU: { constexpr int ACT = DFD_ACT_SILU; __VA_ARGS__; } break; \
        case DFD_ACT_RELU: { constexpr int ACT = DFD_ACT_RELU; __VA_ARGS__; } break; \
        case DFD_ACT_GELU: { constexpr int ACT = DFD_ACT_GELU; __VA_ARGS__; } break; \
        default: return DFD_EUNSUPPORTED;                                            \
    }

// knobs (dfd_tune): A/B switches and sizes read by the host-side planners; set once at start-up, before any launch
enum { DFD_TUNE_DW_MFMA = 0, DFD_TUNE_DW_LDS_KB = 1, DFD_TUNE_DW_GRID = 2, DFD_TUNE_DEBUG = 3, DFD_TUNE_PW_NTD = 4, DFD_TUNE_NTD_NS = 5, DFD_TUNE_NTD_MAXN = 6, DFD_TUNE_NTD_MINT = 7,
       DFD_TUNE_DWQ_GRID_FWD = 8, DFD_TUNE_DWQ_GRID_BWD = 9, DFD_TUNE_DWQ_GRID_WGRAD = 10, DFD_TUNE_DWQ_GRID_MIN = 11, DFD_TUNE_TN_WGS = 12, DFD_TUNE_DWQ_WIDE = 13,
       DFD_TUNE_COUNT = 16 };
int dfd_tune_get(int key);

// sums P partial rows of L floats; the buffer needs room for P + ceil(P/32) rows (two-stage reduction)
// deferrable: inside dfd_sum_batch_begin/_end the sum is recorded and launched with the batch (weight gradients);
// false: launched now, the next launch may read `out`
int dfd_launch_sum_partials(float* partials, int P, long L, float* out, int accumulate, hipStream_t st, bool deferrable = true);
