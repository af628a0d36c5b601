// dfd_pw.h — definitions shared by the 1x1-convolution GEMM kernels (dfd_pwconv.hip, dfd_pwntw.hip)
#pragma once
#include "dfd_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define PW_BM 128

struct ProArgs {
    const void* a2;
    const float* coef;
    const float* gate;
    int HW;
    unsigned hw_magic;      // row -> image: umulhi(m, hw_magic) >> hw_shift  (hw_shift < 0: HW == 1)
    int hw_shift;
};
__device__ __forceinline__ int pro_image(const ProArgs& pa, int m) {
    return pa.hw_shift < 0 ? m : (int)(__umulhi((unsigned)m, pa.hw_magic) >> pa.hw_shift);
}

// exact unsigned division by a launch constant: x / d == umulhi(x, mul) >> sh  (sh < 0: d == 1), x < 2^31
struct Magic { unsigned mul; int sh; };
static inline Magic make_magic(int d) {
    Magic m{0u, -1};
    if (d > 1) {
        int s = 0;
        while ((1ll << s) < d) ++s;
        m.mul = (unsigned)(((1ull << (31 + s)) + (unsigned long long)d - 1) / (unsigned long long)d);
        m.sh = s - 1;
    }
    return m;
}
__device__ __forceinline__ int udiv(int x, const Magic& m) { return m.sh < 0 ? x : (int)(__umulhi((unsigned)x, m.mul) >> m.sh); }

// implicit-GEMM view of a dense k x k convolution: row m = output pixel (n, oy, ox), column k = (tap, channel)
struct ConvArgs {
    int H, W, C, Ho, Wo, ks, stride, pt, pl;
    Magic howo, wo, c, kk;          // divisions by Ho*Wo, Wo, C, ks
};

template <typename T> struct El;
template <> struct El<bf16> { static constexpr int EPC = 8; static constexpr int BK = 64; };   // elements per 16-B chunk, K tile
template <> struct El<float> { static constexpr int EPC = 4; static constexpr int BK = 32; };

__device__ __forceinline__ void q_to_f(const uint4& q, float (&v)[8]) { Vec<bf16>::unpack(q, v); }
__device__ __forceinline__ void q_to_f(const uint4& q, float (&v)[4]) {
    v[0] = __uint_as_float(q.x); v[1] = __uint_as_float(q.y); v[2] = __uint_as_float(q.z); v[3] = __uint_as_float(q.w);
}
__device__ __forceinline__ uint4 f_to_q(const float (&v)[8]) { return Vec<bf16>::pack(v); }
__device__ __forceinline__ uint4 f_to_q(const float (&v)[4]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}

// apply the prologue to one 16-byte chunk (EPC consecutive k of one row)
template <typename T, int PRO, int ACT>
__device__ __forceinline__ uint4 apply_pro(uint4 q, uint4 q2, const float* __restrict__ coef, const float* __restrict__ gate_row,
                                           int k, int K) {
    constexpr int E = El<T>::EPC;
    if constexpr (PRO == DFD_PRO_NONE) {
        return q;
    } else {
        float v[E], c0[E], c1[E];
        q_to_f(q, v);
        load_f32<E>(coef + k, c0);
        load_f32<E>(coef + K + k, c1);
        if constexpr (PRO == DFD_PRO_AFFINE2) {
            float v2[E], c2[E];
            q_to_f(q2, v2);
            load_f32<E>(coef + 2 * K + k, c2);
#pragma unroll
            for (int j = 0; j < E; ++j) v[j] = fmaf(c0[j], v[j], fmaf(c1[j], v2[j], c2[j]));
        } else {
            bn_act_array<ACT, E>(v, c0, c1);
            if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
                float gt[E];
                load_f32<E>(gate_row + k, gt);
                // the activated tensor is rounded to T before the gate multiply, as an
                // unfused pipeline would store it
#pragma unroll
                for (int j = 0; j < E; ++j) v[j] = round_to<T>(v[j]) * gt[j];
            }
        }
        return f_to_q(v);
    }
}

// same with the per-channel coefficient vectors already in registers
template <typename T, int PRO, int ACT, int E>
__device__ __forceinline__ uint4 apply_pro_c(uint4 q, uint4 q2, const float (&c0)[E], const float (&c1)[E], const float (&c2)[E],
                                             const float* __restrict__ gate_k) {
    if constexpr (PRO == DFD_PRO_NONE) {
        return q;
    } else {
        float v[E];
        q_to_f(q, v);
        if constexpr (PRO == DFD_PRO_AFFINE2) {
            float v2[E];
            q_to_f(q2, v2);
#pragma unroll
            for (int j = 0; j < E; ++j) v[j] = fmaf(c0[j], v[j], fmaf(c1[j], v2[j], c2[j]));
        } else {
            bn_act_array<ACT, E>(v, c0, c1);
            if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
                float gt[E];
                load_f32<E>(gate_k, gt);
#pragma unroll
                for (int j = 0; j < E; ++j) v[j] = round_to<T>(v[j]) * gt[j];
            }
        }
        return f_to_q(v);
    }
}

// same with the gate values in registers too (read from an LDS table or from global memory by the caller)
template <typename T, int PRO, int ACT, int E>
__device__ __forceinline__ uint4 apply_pro_v(uint4 q, uint4 q2, const float (&c0)[E], const float (&c1)[E], const float (&c2)[E],
                                             const float (&gt)[E]) {
    if constexpr (PRO == DFD_PRO_NONE) {
        return q;
    } else {
        float v[E];
        q_to_f(q, v);
        if constexpr (PRO == DFD_PRO_AFFINE2) {
            float v2[E];
            q_to_f(q2, v2);
#pragma unroll
            for (int j = 0; j < E; ++j) v[j] = fmaf(c0[j], v[j], fmaf(c1[j], v2[j], c2[j]));
        } else {
            bn_act_array<ACT, E>(v, c0, c1);
            if constexpr (PRO == DFD_PRO_BN_ACT_GATE) {
#pragma unroll
                for (int j = 0; j < E; ++j) v[j] = round_to<T>(v[j]) * gt[j];
            }
        }
        return f_to_q(v);
    }
}

// activations instantiated for the GEMM prologues (EfficientNet: SiLU; EfficientFormerV2 / FasterViT: GELU)
#define DISPATCH_ACT_PW(ACTV, ...)                                                   \
    switch (ACTV) {                                                                  \
        case DFD_ACT_NONE: { constexpr int ACT = DFD_ACT_NONE; __VA_ARGS__; } break; \
        case DFD_ACT_SILU: { constexpr int ACT = DFD_ACT_SILU; __VA_ARGS__; } break; \
        case DFD_ACT_GELU: { constexpr int ACT = DFD_ACT_GELU; __VA_ARGS__; } break; \
        default: return DFD_EUNSUPPORTED;                                            \
    }
static inline bool pro_ok(const dfd_prologue* p) {
    if (!p) return true;
    switch (p->mode) {
        case DFD_PRO_NONE: return true;
        case DFD_PRO_BN_ACT: return p->coef != nullptr;
        case DFD_PRO_BN_ACT_GATE: return p->coef && p->gate && p->HW > 0;
        case DFD_PRO_AFFINE2: return p->coef && p->a2;
        default: return false;
    }
}
static inline ProArgs pro_args(const dfd_prologue* p) {
    ProArgs a{nullptr, nullptr, nullptr, 1, 0u, -1};
    if (p) { a.a2 = p->a2; a.coef = p->coef; a.gate = p->gate; a.HW = p->HW > 0 ? p->HW : 1; }
    if (a.HW > 1) {
        // exact for every m < 2^31: magic = ceil(2^(31+s) / HW), s = ceil(log2 HW)
        int sh = 0;
        while ((1ll << sh) < a.HW) ++sh;
        a.hw_magic = (unsigned)(((1ull << (31 + sh)) + (unsigned long long)a.HW - 1) / (unsigned long long)a.HW);
        a.hw_shift = sh - 1;
    }
    return a;
}


// byte offset of 16-B chunk `ch` of row m in a bf16 [64][128] tile, swizzled for tr reads
__device__ __forceinline__ int tn_off_bf16(int m, int ch) {
    const int f = ((m & 3) | (((m >> 3) & 1) << 2)) << 1;
    return m * 256 + ((ch ^ f) << 4);
}

// wave-autonomous NT kernel for layers whose weight panel stays resident in LDS (dfd_pwntw.hip);
// returns DFD_EUNSUPPORTED when the shape does not qualify (the caller then uses the tiled kernel)
int dfd_pw_ntw(int dtype, const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M,
               int K, int Nout, float* partials, int pcap, int* nparts, hipStream_t st, const float* ebn = nullptr,
               int eact = DFD_ACT_NONE);

// mid-size layers (12.5 k .. 64 k rows): LDS-DMA ring, one 64-row tile per workgroup (dfd_pwntd.hip, bf16 only); DFD_EUNSUPPORTED
// when the shape or the prologue / epilogue combination is not served
int dfd_pw_ntd(int dtype, const void* a, const dfd_prologue* pro, const void* w, void* out, const void* residual, int M, int K,
               int Nout, float* partials, int pcap, int* nparts, hipStream_t st);

// direct 3x3 stride-1 pad-1 dense convolution with resident weights (dfd_conv3.hip, bf16 only); DFD_EUNSUPPORTED when the shape
// does not qualify (dfd_conv_fwd then runs the implicit GEMM)
int dfd_conv3_direct(const void* x, const dfd_dwconv_shape* s, const float* in_bnstate, int in_act, const void* w_nk, int Cout,
                     void* y, float* partials, int pcap, int* nparts, hipStream_t st);
// its weight gradient (output block resident in registers); ws bytes it needs (0: shape not served)
size_t dfd_conv3_wgrad_ws(const dfd_dwconv_shape* s, int Cout);
int dfd_conv3_wgrad(const void* p, const dfd_prologue* pro_p, int Cout, const void* x, const dfd_dwconv_shape* s,
                    const float* in_bnstate, int in_act, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st);

// plain bf16 NT product on 256 x 256 tiles with LDS-DMA staging (dfd_gemm.hip); DFD_EUNSUPPORTED: shape not served
int dfd_gemm_nt_dma(const void* a, const void* w, void* out, int M, int K, int N, hipStream_t st);

// wave-autonomous TN (weight-gradient) kernel for large-M layers with a narrow and a wide operand
// (dfd_pwtnw.hip, bf16 only); DFD_EUNSUPPORTED when the shape does not qualify
int dfd_pw_tnw(const void* p, const dfd_prologue* pro_p, int Ni, const void* q, const dfd_prologue* pro_q, int Nj, int M,
               float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st);
