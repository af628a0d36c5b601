// dfd_misc.hip — the small kernels around the conv stack: squeeze-excite MLP, stem
// convolution, classifier head, label-smoothed cross entropy, softmax/argmax and the
// fused multi-tensor AdamW step.  All f32 except the stem's output / gradient tensors.
#include "dfd_common.h"
#include "dfd_se.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int ACT> __device__ __forceinline__ float act_rt(float z) { return act_fwd<ACT>(z); }

// ===========================================================================
// squeeze-excite MLP
// ===========================================================================
#define SE_MAX_C 4096
#define SE_MAX_R 128

// ---- forward, two short kernels instead of one long per-image loop:
//   A: grid (ceil(R/4), N): one wave per hidden unit r: hpre[n][r] = b1[r] + <pooled[n,:], w1[r,:]>
//   B: grid (ceil(C/256), N): one lane per channel c: gate = sigmoid(b2[c] + sum_r w2t[r][c]*act(hpre[n][r]))
__device__ __forceinline__ float wave_dot(const float* __restrict__ a, const float* __restrict__ b, int C, int lane) {
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {                 // C % 8 == 0: float4 loads
        const float4 x = *reinterpret_cast<const float4*>(a + c);
        const float4 y = *reinterpret_cast<const float4*>(b + c);
        s = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, s))));
    }
    return wave_sum(s);
}

__global__ void __launch_bounds__(DFD_THREADS)
k_se_hidden(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ b1, int C, int R,
            float* __restrict__ hpre) {
    const int n = blockIdx.y, lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float s = wave_dot(pooled + (long)n * C, w1 + (long)r * C, C, lane);
    if (lane == 0) hpre[(long)n * R + r] = s + (b1 ? b1[r] : 0.f);
}

template <int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_se_gate(const float* __restrict__ hpre, const float* __restrict__ w2t, const float* __restrict__ b2, int C, int R,
          float* __restrict__ gate) {
    __shared__ float sh[SE_MAX_R];
    const int n = blockIdx.y, t = threadIdx.x, c = blockIdx.x * DFD_THREADS + t;
    if (t < R) sh[t] = act_rt<ACT>(hpre[(long)n * R + t]);
    __syncthreads();
    if (c >= C) return;
    float s = b2 ? b2[c] : 0.f;
    for (int r = 0; r < R; ++r) s = fmaf(w2t[(long)r * C + c], sh[r], s);
    gate[(long)n * C + c] = sigmoid_f(s);
}

// ---- backward, per-image part:
//   A: grid (ceil(R/4), N): wave per r: dh = <dgate*g*(1-g), w2t[r,:]>; dhpre = dh*act'(hpre); also
//      writes ws_g (by the r-block 0) / ws_dh / ws_h for the weight-gradient kernel
//   B: grid (ceil(C/256), N): lane per c: dpooled[n][c] = sum_r dhpre[n][r] * w1[r][c]
template <int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_se_bwd_hidden(const float* __restrict__ dgate, const float* __restrict__ gate, const float* __restrict__ hpre,
                const float* __restrict__ w2t, int N, int C, int R, float* __restrict__ ws) {
    const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = blockIdx.x * 4 + wave;
    float* ws_g = ws;
    float* ws_dh = ws + (long)N * C;
    float* ws_h = ws_dh + (long)N * R;
    if (r >= R) return;
    const float* dg = dgate + (long)n * C;
    const float* gt = gate + (long)n * C;
    const float* wr = w2t + (long)r * C;
    const bool writer = blockIdx.x == 0 && wave == 0;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 d = *reinterpret_cast<const float4*>(dg + c);
        const float4 g = *reinterpret_cast<const float4*>(gt + c);
        const float4 w = *reinterpret_cast<const float4*>(wr + c);
        const float4 v = make_float4(d.x * g.x * (1.f - g.x), d.y * g.y * (1.f - g.y), d.z * g.z * (1.f - g.z), d.w * g.w * (1.f - g.w));
        if (writer) *reinterpret_cast<float4*>(ws_g + (long)n * C + c) = v;
        s = fmaf(v.x, w.x, fmaf(v.y, w.y, fmaf(v.z, w.z, fmaf(v.w, w.w, s))));
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float z = hpre[(long)n * R + r];
        ws_dh[(long)n * R + r] = s * act_grad<ACT>(z);
        ws_h[(long)n * R + r] = act_rt<ACT>(z);
    }
}

__global__ void __launch_bounds__(DFD_THREADS)
k_se_bwd_pooled(const float* __restrict__ ws, const float* __restrict__ w1, int N, int C, int R, float* __restrict__ dpooled) {
    __shared__ float sd[SE_MAX_R];
    const float* ws_dh = ws + (long)N * C;
    const int n = blockIdx.y, t = threadIdx.x, c = blockIdx.x * DFD_THREADS + t;
    if (t < R) sd[t] = ws_dh[(long)n * R + t];
    __syncthreads();
    if (c >= C) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s = fmaf(sd[r], w1[(long)r * C + c], s);
    dpooled[(long)n * C + c] = s;
}

// weight gradients: grid (C/64, R); thread (c, nl) sums every 4th image, the four image
// lanes are combined through LDS in a fixed order.
__global__ void __launch_bounds__(DFD_THREADS)
k_se_fc_bwd_w(const float* __restrict__ pooled, const float* __restrict__ ws, int N, int C, int R,
              float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
              int accumulate) {
    __shared__ float red[3 * 4 * 64];
    se_fc_bwd_w_body(pooled, ws, N, C, R, dw1, db1, dw2, db2, accumulate, blockIdx.x, blockIdx.y, red);
}

// ---- one workgroup per image: the whole squeeze-excite MLP behind the pooling kernel (and the sum of its
// H*W split vectors), forward and backward.  Same summation orders as the short kernels above.
int dfd_pool_launch(int dtype, bool bwd, const void* D, const void* y, const float* bnstate, int act, float* out, int N,
                    int HW, int C, void* ws, size_t ws_bytes, hipStream_t st, int* splits, float* mul);

#define SE_IMG_THREADS 1024
// four rows of `w` against the LDS vector at once (independent load streams); each dot keeps wave_dot's order
__device__ __forceinline__ void lds_dot4(const float* sp, const float* __restrict__ w, int C, int R, int r0, int rstep, int lane,
                                         float (&s)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 x = *reinterpret_cast<const float4*>(sp + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + k * rstep;
            if (r < R) {
                const float4 y = *reinterpret_cast<const float4*>(w + (long)r * C + c);
                s[k] = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, s[k]))));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = wave_sum(s[k]);
}

template <int ACT>
__global__ void __launch_bounds__(SE_IMG_THREADS)
k_se_fwd_img(const float* __restrict__ parts, int splits, float mul, float* __restrict__ pooled, const float* __restrict__ w1,
             const float* __restrict__ b1, const float* __restrict__ w2t, const float* __restrict__ b2, int N, int C, int R,
             float* __restrict__ hpre, float* __restrict__ gate) {
    __shared__ __attribute__((aligned(16))) float sp[SE_MAX_C];
    __shared__ float sh[SE_MAX_R];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int NW = SE_IMG_THREADS / 64;
    for (int c = t; c < C; c += SE_IMG_THREADS) {
        float v;
        if (parts) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z) s += parts[((long)z * N + n) * C + c];
            v = s * mul;
            pooled[(long)n * C + c] = v;
        } else {
            v = pooled[(long)n * C + c];
        }
        sp[c] = v;
    }
    __syncthreads();
    for (int rb = wave; rb < R; rb += 4 * NW) {
        float s4[4];
        lds_dot4(sp, w1, C, R, rb, NW, lane, s4);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rb + k * NW;
                if (r < R) {
                    const float z = s4[k] + (b1 ? b1[r] : 0.f);
                    hpre[(long)n * R + r] = z;
                    sh[r] = act_rt<ACT>(z);
                }
            }
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += SE_IMG_THREADS) {
        float s = b2 ? b2[c] : 0.f;
        int r = 0;
        for (; r + 8 <= R; r += 8) {                          // eight weight rows in flight, same FMA order
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w2t[(long)(r + u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(wv[u], sh[r + u], s);
        }
        for (; r < R; ++r) s = fmaf(w2t[(long)r * C + c], sh[r], s);
        gate[(long)n * C + c] = sigmoid_f(s);
    }
}

template <int ACT>
__global__ void __launch_bounds__(SE_IMG_THREADS)
k_se_bwd_img(const float* __restrict__ parts, int splits, const float* __restrict__ dgate_in, const float* __restrict__ gate,
             const float* __restrict__ hpre, const float* __restrict__ w1, const float* __restrict__ w2t, int N, int C, int R,
             float* __restrict__ ws, float* __restrict__ dpooled) {
    __shared__ __attribute__((aligned(16))) float sp[SE_MAX_C];
    __shared__ float sd[SE_MAX_R];
    float* ws_g = ws;
    float* ws_dh = ws + (long)N * C;
    float* ws_h = ws_dh + (long)N * R;
    constexpr int NW = SE_IMG_THREADS / 64;
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int c = t; c < C; c += SE_IMG_THREADS) {
        float d;
        if (splits > 1) {
            d = 0.f;
            for (int z = 0; z < splits; ++z) d += parts[((long)z * N + n) * C + c];
        } else {
            d = dgate_in[(long)n * C + c];
        }
        const float g = gate[(long)n * C + c];
        const float v = d * g * (1.f - g);
        ws_g[(long)n * C + c] = v;
        sp[c] = v;
    }
    __syncthreads();
    for (int rb = wave; rb < R; rb += 4 * NW) {
        float s4[4];
        lds_dot4(sp, w2t, C, R, rb, NW, lane, s4);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = rb + k * NW;
                if (r < R) {
                    const float z = hpre[(long)n * R + r];
                    const float dh = s4[k] * act_grad<ACT>(z);
                    ws_dh[(long)n * R + r] = dh;
                    ws_h[(long)n * R + r] = act_rt<ACT>(z);
                    sd[r] = dh;
                }
            }
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += SE_IMG_THREADS) {
        float s = 0.f;
        int r = 0;
        for (; r + 8 <= R; r += 8) {
            float wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = w1[(long)(r + u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(sd[r + u], wv[u], s);
        }
        for (; r < R; ++r) s = fmaf(sd[r], w1[(long)r * C + c], s);
        dpooled[(long)n * C + c] = s;
    }
}

__global__ void k_transpose_f32(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // index into out [cols][rows]
    if (i >= rows * cols) return;
    const int c = i / rows, r = i - c * rows;
    out[i] = in[(long)r * cols + c];
}

extern "C" int dfd_se_fc_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                             int N, int C, int R, int act, float* hpre, float* gate, float* w2t, dfd_stream stream) {
    if (!pooled || !w1 || !hpre || !gate || !w2t || N < 1 || C < 1 || R < 1) return DFD_EINVAL;
    if (C > SE_MAX_C || R > SE_MAX_R) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // w2 is [C][R]; every use sums over r for many c (or over c for few r): keep an [R][C] copy
    // (w2 == NULL: the caller has already filled w2t, e.g. with dfd_prep_weights_multi)
    if (w2) hipLaunchKernelGGL(k_transpose_f32, dim3((C * R + 255) / 256), dim3(256), 0, st, w2, w2t, C, R);
    hipLaunchKernelGGL(k_se_hidden, dim3((R + 3) / 4, N), dim3(DFD_THREADS), 0, st, pooled, w1, b1, C, R, hpre);
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_se_gate<ACT>), dim3((C + DFD_THREADS - 1) / DFD_THREADS, N), dim3(DFD_THREADS), 0, st, hpre, w2t,
                           b2, C, R, gate);
    });
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_se_fwd(int dtype, const void* y, const float* bnstate, int act_in, int N, int HW, int C, const float* w1,
                          const float* b1, const float* w2, const float* b2, int R, int act, float* pooled, float* hpre,
                          float* gate, float* w2t, void* ws, size_t ws_bytes, dfd_stream stream) {
    if (!y || !bnstate || !w1 || !pooled || !hpre || !gate || !w2t || N < 1 || C < 1 || R < 1 || C % 4) return DFD_EINVAL;
    if (C > SE_MAX_C || R > SE_MAX_R) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (w2) hipLaunchKernelGGL(k_transpose_f32, dim3((C * R + 255) / 256), dim3(256), 0, st, w2, w2t, C, R);
    int splits = 1;
    float mul = 1.f;
    const int rc = dfd_pool_launch(dtype, false, nullptr, y, bnstate, act_in, pooled, N, HW, C, ws, ws_bytes, st, &splits, &mul);
    if (rc != DFD_OK) return rc;
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_se_fwd_img<ACT>), dim3(N), dim3(SE_IMG_THREADS), 0, st, splits > 1 ? (const float*)ws : nullptr, splits,
                           mul, pooled, w1, b1, w2t, b2, N, C, R, hpre, gate);
    });
    return DFD_CHECK_LAUNCH();
}

// the same branch from per-tile channel sums that the producer already formed (dfd_dwconv_fwd_eval): no pooling pass
extern "C" int dfd_se_fwd_parts(const float* parts, int splits, int N, int HW, int C, const float* w1, const float* b1,
                                const float* w2, const float* b2, int R, int act, float* pooled, float* hpre, float* gate,
                                float* w2t, dfd_stream stream) {
    if (!parts || splits < 1 || !w1 || !pooled || !hpre || !gate || !w2t || N < 1 || HW < 1 || C < 1 || R < 1 || C % 4) return DFD_EINVAL;
    if (C > SE_MAX_C || R > SE_MAX_R) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (w2) hipLaunchKernelGGL(k_transpose_f32, dim3((C * R + 255) / 256), dim3(256), 0, st, w2, w2t, C, R);
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_se_fwd_img<ACT>), dim3(N), dim3(SE_IMG_THREADS), 0, st, parts, splits, 1.f / (float)HW, pooled, w1, b1,
                           w2t, b2, N, C, R, hpre, gate);
    });
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_se_bwd(int dtype, const void* D, const void* y, const float* bnstate, int act_in, int N, int HW, int C,
                          const float* gate, const float* hpre, const float* pooled, const float* w1, const float* w2t, int R,
                          int act, float* dgate, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int accumulate,
                          void* pool_ws, size_t pool_ws_bytes, float* ws, dfd_stream stream) {
    if (!D || !y || !bnstate || !gate || !hpre || !pooled || !w1 || !w2t || !dgate || !dpooled || !ws || N < 1 || C < 1 ||
        R < 1 || C % 4)
        return DFD_EINVAL;
    if (C > SE_MAX_C || R > SE_MAX_R) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int splits = 1;
    float mul = 1.f;
    const int rc = dfd_pool_launch(dtype, true, D, y, bnstate, act_in, dgate, N, HW, C, pool_ws, pool_ws_bytes, st, &splits, &mul);
    if (rc != DFD_OK) return rc;
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_se_bwd_img<ACT>), dim3(N), dim3(SE_IMG_THREADS), 0, st, (const float*)pool_ws, splits, dgate, gate, hpre,
                           w1, w2t, N, C, R, ws, dpooled);
    });
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    if (dw1 && dw2) {
        hipLaunchKernelGGL(k_se_fc_bwd_w, dim3((C + 63) / 64, R), dim3(DFD_THREADS), 0, st, pooled, ws, N, C, R, dw1, db1,
                           dw2, db2, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_se_fc_bwd(const float* dgate, const float* gate, const float* hpre, const float* pooled,
                             const float* w1, const float* w2t, int N, int C, int R, int act, float* dpooled,
                             float* dw1, float* db1, float* dw2, float* db2, int accumulate, float* ws,
                             dfd_stream stream) {
    if (!dgate || !gate || !hpre || !pooled || !w1 || !w2t || !dpooled || !ws || N < 1 || C < 1 || R < 1) return DFD_EINVAL;
    if (C > SE_MAX_C || R > SE_MAX_R) return DFD_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_se_bwd_hidden<ACT>), dim3((R + 3) / 4, N), dim3(DFD_THREADS), 0, st, dgate, gate, hpre, w2t, N, C,
                           R, ws);
    });
    hipLaunchKernelGGL(k_se_bwd_pooled, dim3((C + DFD_THREADS - 1) / DFD_THREADS, N), dim3(DFD_THREADS), 0, st, ws, w1, N, C, R,
                       dpooled);
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    if (dw1 && dw2) {
        hipLaunchKernelGGL(k_se_fc_bwd_w, dim3((C + 63) / 64, R), dim3(DFD_THREADS), 0, st, pooled, ws, N, C, R, dw1, db1,
                           dw2, db2, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// stem convolution: [N][H][W][3] f32 -> [N][Ho][Wo][Co] T
// ===========================================================================
#define STEM_MAX_CO 64
template <typename T, int K>
__global__ void __launch_bounds__(DFD_THREADS)
k_stem_fwd(const float* __restrict__ x, const float* __restrict__ w, T* __restrict__ y, dfd_stem_shape s, ChanMap cm,
           float* __restrict__ partials, int stats) {
    constexpr int V = Vec<T>::N;
    constexpr int TAPS = 3 * K * K;
    __shared__ __attribute__((aligned(16))) float wl[TAPS * STEM_MAX_CO];   // [ci][kh][kw][Co]
    __shared__ float red[DFD_THREADS * 2 * V];
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int Co = s.Cout, c0 = vl * V;
    for (int i = t; i < TAPS * Co; i += DFD_THREADS) {
        const int tap = i / Co, co = i - tap * Co;          // tap = (ci*K + kh)*K + kw
        wl[tap * Co + co] = round_to<T>(w[(long)co * TAPS + tap]);
    }
    __syncthreads();
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    const long npix = (long)s.N * s.Ho * s.Wo;
    if (active) {
        const long step = (long)gridDim.x * cm.rpb;
        for (long pix = (long)blockIdx.x * cm.rpb + rl; pix < npix; pix += step) {
            const int ox = (int)(pix % s.Wo);
            const long tq = pix / s.Wo;
            const int oy = (int)(tq % s.Ho);
            const long n = tq / s.Ho;
            // packed float2 accumulators (v_pk_fma_f32); a kernel row of an interior pixel is nine
            // contiguous floats, fetched as two 16-byte loads and one 4-byte load
            typedef float f2v __attribute__((ext_vector_type(2)));
            f2v acc2[V / 2];
#pragma unroll
            for (int j = 0; j < V / 2; ++j) acc2[j] = (f2v){0.f, 0.f};
            const int ix0 = ox * s.stride - s.pad_left;
            const bool row_inside = ix0 >= 0 && ix0 + K <= s.W && K == 3;
#pragma unroll
            for (int kh = 0; kh < K; ++kh) {
                const int iy = oy * s.stride - s.pad_top + kh;
                if (iy < 0 || iy >= s.H) continue;
                float xr[3 * K];
                const float* prow = x + ((n * s.H + iy) * (long)s.W + ix0) * 3;
                if (row_inside) {
                    struct __attribute__((packed, aligned(4))) F4 { float a, b, c, d; };
                    const F4 q0 = *reinterpret_cast<const F4*>(prow), q1 = *reinterpret_cast<const F4*>(prow + 4);
                    xr[0] = q0.a; xr[1] = q0.b; xr[2] = q0.c; xr[3] = q0.d;
                    xr[4] = q1.a; xr[5] = q1.b; xr[6] = q1.c; xr[7] = q1.d;
                    xr[8] = prow[8];
                } else {
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const int ix = ix0 + kw;
                        const bool ok = ix >= 0 && ix < s.W;
#pragma unroll
                        for (int ci = 0; ci < 3; ++ci) xr[kw * 3 + ci] = ok ? prow[kw * 3 + ci] : 0.f;
                    }
                }
#pragma unroll
                for (int kw = 0; kw < K; ++kw)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const float xv = round_to<T>(xr[kw * 3 + ci]);
                        const f2v xx = (f2v){xv, xv};
                        // the lane's V weights as 16-byte LDS reads (float2 pairs become ds_read2_b64: twice the LDS
                        // cycles and a bank conflict at this 32-byte lane stride, see dfd_dwq.h)
                        const float* wp = wl + ((ci * K + kh) * K + kw) * Co + c0;
                        f2v wv[V / 2];
#pragma unroll
                        for (int j4 = 0; j4 < V / 4; ++j4) {
                            const float4 q = *reinterpret_cast<const float4*>(wp + 4 * j4);
                            wv[2 * j4] = (f2v){q.x, q.y};
                            wv[2 * j4 + 1] = (f2v){q.z, q.w};
                        }
#pragma unroll
                        for (int j = 0; j < V / 2; ++j) acc2[j] = __builtin_elementwise_fma(xx, wv[j], acc2[j]);
                    }
            }
            float acc[V];
#pragma unroll
            for (int j = 0; j < V / 2; ++j) { acc[2 * j] = acc2[j].x; acc[2 * j + 1] = acc2[j].y; }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float r = round_to<T>(acc[j]);
                acc[j] = r;
                s1[j] += r;
                s2[j] = fmaf(r, r, s2[j]);
            }
            Vec<T>::store(y + pix * Co + c0, acc);
        }
    }
    if (stats) {
        float acc2[2 * V];
#pragma unroll
        for (int j = 0; j < V; ++j) { acc2[j] = s1[j]; acc2[V + j] = s2[j]; }
        reduce_rowlanes<2 * V>(acc2, red, cm.cvb, cm.rpb, vl, rl, active);
        if (rl == 0) {
            float* p = partials + (long)blockIdx.x * 2 * Co;
            float a0[V], a1[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { a0[j] = acc2[j]; a1[j] = acc2[V + j]; }
            store_f32<V>(p + c0, a0);
            store_f32<V>(p + Co + c0, a1);
        }
    }
}

// weight gradient = a skinny GEMM  dw[co][tap] = sum_pix dy[pix][co] * patch[pix][tap].
// A workgroup stages chunks of STEM_CHUNK output pixels in LDS: dy [pix][Co] (after the
// BN-backward affine map) and the 3*K*K input patch [pix][32] (taps padded to 32).  Threads
// form pixel groups of (Co/4) x 4 lanes; a lane owns a 4(co) x 8(tap) register tile, reads
// one 16-byte dy vector and two 16-byte patch vectors per pixel (3 LDS reads per 32 FMAs)
// and keeps its sums in registers over the whole persistent loop.
// partial layout: [workgroup][Co][TAPS]
#define STEM_CHUNK 128
#define STEM_TAPP 32
template <typename T, int K>
__global__ void __launch_bounds__(DFD_THREADS)
k_stem_wgrad(const float* __restrict__ x, const T* __restrict__ dz, const T* __restrict__ yraw,
             const float* __restrict__ coef, dfd_stem_shape s, float* __restrict__ ws) {
    constexpr int V = Vec<T>::N;
    constexpr int TAPS = 3 * K * K;
    static_assert(TAPS <= STEM_TAPP, "patch padding");
    __shared__ __attribute__((aligned(16))) float sdy[STEM_CHUNK * STEM_MAX_CO];
    __shared__ __attribute__((aligned(16))) float sx[STEM_CHUNK * STEM_TAPP];
    const int t = threadIdx.x, Co = s.Cout;
    const int COG = Co / 4, TPG = COG * 4, NG = DFD_THREADS / TPG;      // lanes per pixel group, groups
    const int grp = t / TPG, r = t - grp * TPG, cog = r >> 2, tg = r & 3;
    const bool on = grp < NG;
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int i = t; i < STEM_CHUNK * STEM_TAPP; i += DFD_THREADS) sx[i] = 0.f;      // pad taps stay zero
    const long npix = (long)s.N * s.Ho * s.Wo;
    const int CV = Co / V;
    // a lane's channel vector is the same for every item it stages (256 % CV == 0): its BN-backward coefficients are
    // loaded once, not three dependent global loads per item inside the staging loop
    const bool fixed_v = coef != nullptr && DFD_THREADS % CV == 0;
    float ka0[V], kb0[V], kc0[V];
    if (fixed_v) {
        const int v = t % CV;
        load_f32<V>(coef + v * V, ka0);
        load_f32<V>(coef + Co + v * V, kb0);
        load_f32<V>(coef + 2 * Co + v * V, kc0);
    }
    for (long base = (long)blockIdx.x * STEM_CHUNK; base < npix; base += (long)gridDim.x * STEM_CHUNK) {
        const int ox_b = (int)(base % s.Wo);
        const long tq_b = base / s.Wo;
        const int oy_b = (int)(tq_b % s.Ho);
        const long n_b = tq_b / s.Ho;
        __syncthreads();
        for (int i = t; i < STEM_CHUNK * CV; i += DFD_THREADS) {
            const int p = i / CV, v = i - p * CV;
            const long pix = base + p;
            float d[V];
#pragma unroll
            for (int j = 0; j < V; ++j) d[j] = 0.f;
            if (pix < npix) {
                Vec<T>::load(dz + pix * Co + v * V, d);
                if (coef) {
                    float yv[V], ka[V], kb[V], kc[V];
                    Vec<T>::load(yraw + pix * Co + v * V, yv);
                    if (fixed_v) {
#pragma unroll
                        for (int j = 0; j < V; ++j) { ka[j] = ka0[j]; kb[j] = kb0[j]; kc[j] = kc0[j]; }
                    } else {
                        load_f32<V>(coef + v * V, ka);
                        load_f32<V>(coef + Co + v * V, kb);
                        load_f32<V>(coef + 2 * Co + v * V, kc);
                    }
#pragma unroll
                    for (int j = 0; j < V; ++j) d[j] = round_to<T>(fmaf(ka[j], d[j], fmaf(kb[j], yv[j], kc[j])));
                }
            }
            store_f32<V>(sdy + p * Co + v * V, d);
        }
        for (int i = t; i < STEM_CHUNK * K; i += DFD_THREADS) {          // one (pixel, kernel row) per item
            const int p = i / K, kh = i - p * K;
            const long pix = base + p;
            float v[3 * K];
#pragma unroll
            for (int j = 0; j < 3 * K; ++j) v[j] = 0.f;
            if (pix < npix) {
                // (n, oy, ox) of the chunk's first pixel are wave-uniform (scalar divisions, once per
                // chunk); the item's offset p < 128 is folded in with small 32-bit arithmetic
                const unsigned lin = (unsigned)ox_b + (unsigned)p;
                const unsigned dy_ = lin / (unsigned)s.Wo;                   // 32-bit, lin < Wo + STEM_CHUNK
                const int ox = (int)(lin - dy_ * (unsigned)s.Wo);
                int oy = oy_b + (int)dy_;
                long n = n_b;
                while (oy >= s.Ho) { oy -= s.Ho; ++n; }
                const int iy = oy * s.stride - s.pad_top + kh, ix0 = ox * s.stride - s.pad_left;
                if (iy >= 0 && iy < s.H) {
                    const float* row = x + ((n * s.H + iy) * (long)s.W) * 3;
#pragma unroll
                    for (int kw = 0; kw < K; ++kw) {
                        const int ix = ix0 + kw;
                        if (ix >= 0 && ix < s.W) {
#pragma unroll
                            for (int ci = 0; ci < 3; ++ci) v[kw * 3 + ci] = round_to<T>(row[(long)ix * 3 + ci]);
                        }
                    }
                }
            }
            // torch's tap order is (ci, kh, kw)
#pragma unroll
            for (int kw = 0; kw < K; ++kw)
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) sx[p * STEM_TAPP + (ci * K + kh) * K + kw] = v[kw * 3 + ci];
        }
        __syncthreads();
        if (on) {
            for (int p = grp; p < STEM_CHUNK; p += NG) {
                const float4 d = *reinterpret_cast<const float4*>(sdy + p * Co + cog * 4);
                const float4 xa = *reinterpret_cast<const float4*>(sx + p * STEM_TAPP + tg * 8);
                const float4 xb = *reinterpret_cast<const float4*>(sx + p * STEM_TAPP + tg * 8 + 4);
                const float dv[4] = {d.x, d.y, d.z, d.w};
                typedef float f2v __attribute__((ext_vector_type(2)));
                const f2v xp[4] = {(f2v){xa.x, xa.y}, (f2v){xa.z, xa.w}, (f2v){xb.x, xb.y}, (f2v){xb.z, xb.w}};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f2v dd = (f2v){dv[i], dv[i]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f2v a = (f2v){acc[i][2 * j], acc[i][2 * j + 1]};
                        a = __builtin_elementwise_fma(dd, xp[j], a);
                        acc[i][2 * j] = a.x; acc[i][2 * j + 1] = a.y;
                    }
                }
            }
        }
    }
    // combine the pixel groups through LDS (reuse sdy: NG * Co * 32 floats <= chunk buffer)
    __syncthreads();
    float* red = sdy;
    if (on) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(grp * Co + cog * 4 + i) * STEM_TAPP + tg * 8 + j] = acc[i][j];
    }
    __syncthreads();
    float* o = ws + (long)blockIdx.x * Co * TAPS;
    for (int i = t; i < Co * TAPS; i += DFD_THREADS) {
        const int co = i / TAPS, tap = i - co * TAPS;
        float sum = 0.f;
        for (int gq = 0; gq < NG; ++gq) sum += red[(gq * Co + co) * STEM_TAPP + tap];
        o[i] = sum;
    }
}

static bool stem_ok(const dfd_stem_shape* s) {
    if (!s || s->N < 1 || s->H < 1 || s->W < 1 || s->Ho < 1 || s->Wo < 1) return false;
    if (s->Cout < 8 || s->Cout % 8 || s->Cout > STEM_MAX_CO) return false;
    if (s->k != 3 || s->stride < 1 || s->stride > 2) return false;
    if (s->pad_top < 0 || s->pad_left < 0 || s->pad_top >= s->k || s->pad_left >= s->k) return false;
    return true;
}

template <typename T>
static int stem_fwd_t(const float* x, const float* w, void* y, const dfd_stem_shape* s, float* partials, int pcap,
                      int* nparts, hipStream_t st) {
    const ChanMap cm = make_chanmap(s->Cout, Vec<T>::N);
    const long npix = (long)s->N * s->Ho * s->Wo;
    int cap = partials ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    long want = (npix + (long)cm.rpb * 4 - 1) / ((long)cm.rpb * 4);
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if (partials) *nparts = (int)want;
    hipLaunchKernelGGL((k_stem_fwd<T, 3>), dim3((unsigned)want), dim3(DFD_THREADS), 0, st, x, w, (T*)y, *s, cm, partials,
                       partials ? 1 : 0);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_stem_conv_fwd(int dtype, const float* x, const float* w, void* y, const dfd_stem_shape* s,
                                 float* partials, int pcap, int* nparts, dfd_stream stream) {
    if (!x || !w || !y) return DFD_EINVAL;
    if (!stem_ok(s)) return s && s->k != 3 ? DFD_EUNSUPPORTED : DFD_EINVAL;
    if (partials && (!nparts || pcap < 1)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        const int rc = dfd_stem_fwd_mfma(x, w, y, s, partials, pcap, nparts, st);
        if (rc != DFD_EUNSUPPORTED) return rc;
        return stem_fwd_t<bf16>(x, w, y, s, partials, pcap, nparts, st);
    }
    if (dtype == DFD_F32) return stem_fwd_t<float>(x, w, y, s, partials, pcap, nparts, st);
    return DFD_EINVAL;
}

static int stem_wgrad_blocks(const dfd_stem_shape* s) {
    const long npix = (long)s->N * s->Ho * s->Wo;
    long b = (npix + STEM_CHUNK * 4 - 1) / (STEM_CHUNK * 4);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}
extern "C" size_t dfd_stem_conv_wgrad_ws(const dfd_stem_shape* s) {
    if (!stem_ok(s)) return 0;
    return (size_t)(1024 + 34) * s->Cout * 27 * 4;
}
extern "C" int dfd_stem_conv_wgrad(int dtype, const float* x, const void* dz, const void* y, const float* coef,
                                   float* dw, const dfd_stem_shape* s, int accumulate, float* ws, size_t ws_bytes,
                                   dfd_stream stream) {
    if (!x || !dz || !dw || !ws || (coef && !y)) return DFD_EINVAL;
    if (!stem_ok(s)) return DFD_EINVAL;
    const int P = stem_wgrad_blocks(s);
    if ((size_t)(P + P / 32 + 2) * s->Cout * 27 * 4 > ws_bytes) return DFD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        int rows = 0;
        const int rc = dfd_stem_wgrad_mfma(x, dz, y, coef, s, ws, 1024, &rows, st);
        if (rc == DFD_OK) return dfd_launch_sum_partials(ws, rows, (long)s->Cout * 27, dw, accumulate, st);
        if (rc != DFD_EUNSUPPORTED) return rc;
    }
    if (dtype == DFD_BF16)
        hipLaunchKernelGGL((k_stem_wgrad<bf16, 3>), dim3(P), dim3(DFD_THREADS), 0, st, x, (const bf16*)dz, (const bf16*)y, coef, *s, ws);
    else if (dtype == DFD_F32)
        hipLaunchKernelGGL((k_stem_wgrad<float, 3>), dim3(P), dim3(DFD_THREADS), 0, st, x, (const float*)dz, (const float*)y, coef, *s, ws);
    else
        return DFD_EINVAL;
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    // ws is [P][Co][27] and torch's weight is [Co][3][3][3] = [Co][27]
    return dfd_launch_sum_partials(ws, P, (long)s->Cout * 27, dw, accumulate, st);
}

// ===========================================================================
// classifier head
// ===========================================================================
__global__ void k_dropout(const float* __restrict__ x, const float* __restrict__ u, float p, float inv_keep,
                          float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = u[i] >= p ? x[i] * inv_keep : 0.f;
}
extern "C" int dfd_dropout(const float* x, const float* u, float p, float* out, int n, dfd_stream stream) {
    if (!x || !u || !out || n < 1 || p < 0.f || p >= 1.f) return DFD_EINVAL;
    hipLaunchKernelGGL(k_dropout, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, u, p, 1.0f / (1.0f - p), out, n);
    return DFD_CHECK_LAUNCH();
}

// grid (N, ceil(J / 16)): a workgroup owns 16 outputs of one row, a wave four of them (independent dot products,
// each summed in wave_dot's order)
__global__ void __launch_bounds__(DFD_THREADS)
k_linear_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
             int K, int J) {
    const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xr = x + (long)n * K;
    const int j0 = blockIdx.y * 16 + wave * 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
        const float xv = xr[k];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < J) s[u] = fmaf(xv, w[(long)(j0 + u) * K + k], s[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float r = wave_sum(s[u]);
        if (lane == 0 && j0 + u < J) out[(long)n * J + j0 + u] = r + (b ? b[j0 + u] : 0.f);
    }
}
// few input features (the coordinate MLPs' Linear(2, 512)): one lane per output, no cross-lane reduction
__global__ void __launch_bounds__(DFD_THREADS)
k_linear_fwd_smallk(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                    float* __restrict__ out, long total, int K, int J) {
    const long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x;
    if (i >= total) return;
    const long n = i / J;
    const int j = (int)(i - n * J);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(x[n * K + k], w[(long)j * K + k], s);
    out[i] = s + (b ? b[j] : 0.f);
}
extern "C" int dfd_linear_fwd(const float* x, const float* w, const float* b, float* out, int N, int K, int J,
                              dfd_stream stream) {
    if (!x || !w || !out || N < 1 || K < 1 || J < 1) return DFD_EINVAL;
    if (K <= 16) {
        const long total = (long)N * J;
        hipLaunchKernelGGL(k_linear_fwd_smallk, dim3((unsigned)((total + DFD_THREADS - 1) / DFD_THREADS)), dim3(DFD_THREADS), 0,
                           (hipStream_t)stream, x, w, b, out, total, K, J);
        return DFD_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_linear_fwd, dim3(N, (J + 15) / 16), dim3(DFD_THREADS), 0, (hipStream_t)stream, x, w, b, out, K, J);
    return DFD_CHECK_LAUNCH();
}

// grid (ceil(K / 64), N), 256 threads: wave q sums every 4th output feature for its 64 input features; the four
// partial sums are combined in wave order
__global__ void __launch_bounds__(DFD_THREADS)
k_linear_bwd_x(const float* __restrict__ dout, const float* __restrict__ w, float* __restrict__ dx, int K, int J) {
    __shared__ float red[4][64];
    const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, k = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (k < K) {
        int j = wave;
        for (; j + 12 < J; j += 16) {
            float d[4], wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { d[u] = dout[(long)n * J + j + 4 * u]; wv[u] = w[(long)(j + 4 * u) * K + k]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) s = fmaf(d[u], wv[u], s);
        }
        for (; j < J; j += 4) s = fmaf(dout[(long)n * J + j], w[(long)j * K + k], s);
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && k < K) dx[(long)n * K + k] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ void k_linear_bwd_w(const float* __restrict__ dout, const float* __restrict__ x, float* __restrict__ dw,
                               float* __restrict__ db, int N, int K, int J, int accumulate) {
    const int j = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) {
        float s = 0.f;
        int n = 0;
        for (; n + 8 <= N; n += 8) {           // eight rows in flight; products added in row order
            float xv[8], dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { xv[u] = x[(long)(n + u) * K + k]; dv[u] = dout[(long)(n + u) * J + j]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s = fmaf(dv[u], xv[u], s);
        }
        for (; n < N; ++n) s = fmaf(dout[(long)n * J + j], x[(long)n * K + k], s);
        float* p = dw + (long)j * K + k;
        *p = (accumulate ? *p : 0.f) + s;
    }
    if (db && blockIdx.x == 0) {                      // the workgroup is one wave: lanes stride over the rows
        float s = 0.f;
        for (int n = threadIdx.x; n < N; n += 64) s += dout[(long)n * J + j];
        s = wave_sum(s);
        if (threadIdx.x == 0) db[j] = (accumulate ? db[j] : 0.f) + s;
    }
}
extern "C" int dfd_linear_bwd(const float* dout, const float* x, const float* w, float* dx, float* dw, float* db,
                              int N, int K, int J, int accumulate, dfd_stream stream) {
    if (!dout || N < 1 || K < 1 || J < 1) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        if (!w) return DFD_EINVAL;
        hipLaunchKernelGGL(k_linear_bwd_x, dim3((K + 63) / 64, N), dim3(DFD_THREADS), 0, st, dout, w, dx, K, J);
    }
    if (dw) {
        if (!x) return DFD_EINVAL;
        hipLaunchKernelGGL(k_linear_bwd_w, dim3((K + 63) / 64, J), dim3(64), 0, st, dout, x, dw, db, N, K, J, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// label-smoothed cross entropy (mean reduction), softmax / argmax
// ===========================================================================
__device__ __forceinline__ float block_reduce(float v, float* sm, bool is_max) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    float r = sm[0];
    for (int i = 1; i < DFD_THREADS / 64; ++i) r = is_max ? fmaxf(r, sm[i]) : r + sm[i];
    return r;
}

__global__ void __launch_bounds__(DFD_THREADS)
k_ce_rows(const float* __restrict__ logits, const int64_t* __restrict__ targets, int N, int J, float eps, float gscale,
          float* __restrict__ row_loss, float* __restrict__ dlogits) {
    __shared__ float sm[4];
    const int n = blockIdx.x, t = threadIdx.x;
    const float* lr = logits + (long)n * J;
    float mx = -INFINITY;
    for (int j = t; j < J; j += DFD_THREADS) mx = fmaxf(mx, lr[j]);
    mx = block_reduce(mx, sm, true);
    float se = 0.f, sl = 0.f;
    for (int j = t; j < J; j += DFD_THREADS) { const float d = lr[j] - mx; se += __expf(d); sl += d; }
    se = block_reduce(se, sm, false);
    sl = block_reduce(sl, sm, false);
    const float lse = __logf(se);
    const int tgt = (int)targets[n];
    // sum_j logp_j = sl - J*lse ; logp_t = lr[t] - mx - lse
    if (t == 0) {
        const float logp_t = lr[tgt] - mx - lse;
        const float smooth = -(sl - (float)J * lse) / (float)J;
        row_loss[n] = (1.f - eps) * (-logp_t) + eps * smooth;
    }
    if (dlogits) {
        const float inv = 1.f / se, sc = gscale / (float)N;
        for (int j = t; j < J; j += DFD_THREADS) {
            const float p = __expf(lr[j] - mx) * inv;
            const float q = eps / (float)J + (j == tgt ? 1.f - eps : 0.f);
            dlogits[(long)n * J + j] = (p - q) * sc;
        }
    }
}
__global__ void k_mean_rows(const float* __restrict__ row_loss, int N, float* __restrict__ loss) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < N; i += DFD_THREADS) s += row_loss[i];
    s = block_reduce(s, sm, false);
    if (threadIdx.x == 0) *loss = s / (float)N;
}
extern "C" int dfd_ce_loss(const float* logits, const int64_t* targets, int N, int J, float label_smoothing,
                           float grad_scale, float* row_loss, float* loss, float* dlogits, dfd_stream stream) {
    if (!logits || !targets || !row_loss || !loss || N < 1 || J < 1) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ce_rows, dim3(N), dim3(DFD_THREADS), 0, st, logits, targets, N, J, label_smoothing, grad_scale,
                       row_loss, dlogits);
    hipLaunchKernelGGL(k_mean_rows, dim3(1), dim3(DFD_THREADS), 0, st, row_loss, N, loss);
    return DFD_CHECK_LAUNCH();
}

__global__ void __launch_bounds__(DFD_THREADS)
k_softmax_argmax(const float* __restrict__ logits, int J, float* __restrict__ probs, int64_t* __restrict__ preds) {
    __shared__ float sm[4];
    __shared__ int si[4];
    const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* lr = logits + (long)n * J;
    float mx = -INFINITY;
    for (int j = t; j < J; j += DFD_THREADS) mx = fmaxf(mx, lr[j]);
    mx = block_reduce(mx, sm, true);
    float se = 0.f;
    for (int j = t; j < J; j += DFD_THREADS) se += __expf(lr[j] - mx);
    se = block_reduce(se, sm, false);
    const float inv = 1.f / se;
    float bp = -1.f; int bi = 0x7fffffff;
    for (int j = t; j < J; j += DFD_THREADS) {
        const float p = __expf(lr[j] - mx) * inv;
        if (probs) probs[(long)n * J + j] = p;
        if (p > bp) { bp = p; bi = j; }
    }
    // arg max of the probabilities, lowest index on ties
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float op = __shfl_xor(bp, o);
        const int oi = __shfl_xor(bi, o);
        if (op > bp || (op == bp && oi < bi)) { bp = op; bi = oi; }
    }
    __syncthreads();
    if (lane == 0) { sm[wave] = bp; si[wave] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int i = 1; i < 4; ++i)
            if (sm[i] > bp || (sm[i] == bp && si[i] < bi)) { bp = sm[i]; bi = si[i]; }
        preds[n] = bi;
    }
}
extern "C" int dfd_softmax_argmax(const float* logits, int N, int J, float* probs, int64_t* preds, dfd_stream stream) {
    if (!logits || !preds || N < 1 || J < 1) return DFD_EINVAL;
    hipLaunchKernelGGL(k_softmax_argmax, dim3(N), dim3(DFD_THREADS), 0, (hipStream_t)stream, logits, J, probs, preds);
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// fused multi-tensor AdamW (torch.optim.AdamW semantics, decoupled weight decay)
// ===========================================================================
__global__ void __launch_bounds__(DFD_THREADS)
k_adamw(const int64_t* __restrict__ table, const float* __restrict__ hp) {
    const int64_t* row = table + (long)blockIdx.x * DFD_ADAMW_TABLE_COLS;
    float* p = reinterpret_cast<float*>(row[0]);
    const float* g = reinterpret_cast<const float*>(row[1]);
    float* m = reinterpret_cast<float*>(row[2]);
    float* v = reinterpret_cast<float*>(row[3]);
    const int cnt = (int)row[4];
    const float lr = hp[0], b1 = hp[1], b2 = hp[2], eps = hp[3], wd = hp[4], bc1 = hp[5], bc2 = hp[6], gs = hp[7];
    const float step = lr / bc1, rs2 = 1.0f / sqrtf(bc2), decay = 1.f - lr * wd;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gs;
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        pp = pp * decay - step * (mm / (sqrtf(vv) * rs2 + eps));
    };
    const bool vec = ((row[0] | row[1] | row[2] | row[3]) & 15) == 0;
    const int nv = vec ? cnt / 4 : 0;
    for (int i = threadIdx.x; i < nv; i += DFD_THREADS) {
        float4 pq = reinterpret_cast<float4*>(p)[i], mq = reinterpret_cast<float4*>(m)[i], vq = reinterpret_cast<float4*>(v)[i];
        const float4 gq = reinterpret_cast<const float4*>(g)[i];
        upd(pq.x, gq.x, mq.x, vq.x); upd(pq.y, gq.y, mq.y, vq.y); upd(pq.z, gq.z, mq.z, vq.z); upd(pq.w, gq.w, mq.w, vq.w);
        reinterpret_cast<float4*>(p)[i] = pq; reinterpret_cast<float4*>(m)[i] = mq; reinterpret_cast<float4*>(v)[i] = vq;
    }
    for (int i = nv * 4 + threadIdx.x; i < cnt; i += DFD_THREADS) upd(p[i], g[i], m[i], v[i]);
}
extern "C" int dfd_adamw_step(const int64_t* table, int nchunks, const float* hp, dfd_stream stream) {
    if (!table || !hp || nchunks < 1) return DFD_EINVAL;
    hipLaunchKernelGGL(k_adamw, dim3(nchunks), dim3(DFD_THREADS), 0, (hipStream_t)stream, table, hp);
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// input tail: uint8 NHWC batch -> [horizontal flip] -> /255 -> (x - mean) / std -> [erase box] -> f32 NHWC
// (RandomHorizontalFlip, ToTensor, Normalize, RandomErasing(value=0) of trainers/efficientnet.py:111-234,
//  in that order; the random decisions arrive as per-image parameters).  One thread = one pixel (3 channels).
// ===========================================================================
__global__ void __launch_bounds__(DFD_THREADS)
k_image_prep(const unsigned char* __restrict__ src, float* __restrict__ dst, int N, int H, int W, float m0, float m1, float m2,
             float s0, float s1, float s2, const unsigned char* __restrict__ flip, const int* __restrict__ erase) {
    const long total = (long)N * H * W;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const int sx = (flip && flip[n]) ? W - 1 - x : x;
        const unsigned char* p = src + (((long)n * H + y) * W + sx) * 3;
        float r = ((float)p[0] / 255.0f - m0) / s0;
        float g = ((float)p[1] / 255.0f - m1) / s1;
        float b = ((float)p[2] / 255.0f - m2) / s2;
        if (erase) {
            const int* e = erase + 4 * n;                      // top, left, height, width (height 0: none)
            if (e[2] > 0 && y >= e[0] && y < e[0] + e[2] && x >= e[1] && x < e[1] + e[3]) { r = 0.f; g = 0.f; b = 0.f; }
        }
        float* o = dst + i * 3;
        o[0] = r; o[1] = g; o[2] = b;
    }
}
extern "C" int dfd_image_prep(const unsigned char* src, float* dst, int N, int H, int W, const float* mean3,
                              const float* std3, const unsigned char* flip, const int* erase, dfd_stream stream) {
    if (!src || !dst || !mean3 || !std3 || N < 1 || H < 1 || W < 1) return DFD_EINVAL;
    if (std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f) return DFD_EINVAL;
    const long total = (long)N * H * W;
    long grid = (total + DFD_THREADS - 1) / DFD_THREADS;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_image_prep, dim3((unsigned)grid), dim3(DFD_THREADS), 0, (hipStream_t)stream, src, dst, N, H, W, mean3[0],
                       mean3[1], mean3[2], std3[0], std3[1], std3[2], flip, erase);
    return DFD_CHECK_LAUNCH();
}
