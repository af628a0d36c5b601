// dfd_vit.hip — the kernels the EfficientFormerV2 / FasterViT engines need beyond the MBConv set:
//   * small strided batched GEMM (attention Q.K^T, P.V and their gradients; one workgroup per (batch, head),
//     operands staged in LDS as f32) — these contractions are 49..196 tokens x 16..128 features, far below one
//     MFMA tile per wave in at least one dimension, and latency- not throughput-bound;
//   * the attention row kernel: [talking-head mix ->] softmax [-> talking-head mix], forward and backward,
//     with all heads of a (batch, query) row in registers;
//   * learned attention-bias gather / scatter, im2col / col2im for the dense 3x3 convolutions, the
//     BatchNorm + add + activation passes around the attention core, bilinear x2 upsampling, LayerNorm;
//   * bookkeeping kernels that keep ATen off the model path: Philox uniforms, counter bumps, axpby.
// Layout conventions: see dfd_common.h (NHWC activations, 16 bytes per lane, f32 statistics).
#include "dfd_common.h"

// ===========================================================================
// bookkeeping
// ===========================================================================
__global__ void k_axpby(const float* __restrict__ x, const float* __restrict__ y, float a, float b,
                        const float* __restrict__ a_dev, float* __restrict__ out, long n) {
    const float aa = a_dev ? a * a_dev[0] : a;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = aa * x[i] + (y ? b * y[i] : 0.f);
}
extern "C" int dfd_axpby(const float* x, const float* y, float a, float b, const float* a_dev, float* out, long n,
                         dfd_stream stream) {
    if (!x || !out || n < 1) return DFD_EINVAL;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_axpby, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, a, b, a_dev, out, n);
    return DFD_CHECK_LAUNCH();
}

template <typename T>
__global__ void k_add(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long nvec) {
    constexpr int V = Vec<T>::N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
        float u[V], v[V];
        Vec<T>::load(a + i * V, u);
        Vec<T>::load(b + i * V, v);
#pragma unroll
        for (int j = 0; j < V; ++j) u[j] += v[j];
        Vec<T>::store(out + i * V, u);
    }
}
extern "C" int dfd_add(int dtype, const void* a, const void* b, void* out, long n, dfd_stream stream) {
    if (!a || !b || !out || n < 8 || n % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = (n / 8 + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_add<bf16>), dim3((unsigned)grid), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, n / 8);
    } else if (dtype == DFD_F32) {
        long grid = (n / 4 + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_add<float>), dim3((unsigned)grid), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n / 4);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

// Philox4x32-10 (Salmon et al., SC'11): key = seed, counter = (offset lo, offset hi, stream id, block index)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// keep <= 0: out[i] = U[0,1); keep > 0: out[i] = floor(keep + u) / keep (drop-path row scale)
__global__ void k_rand(const uint64_t* __restrict__ state, uint32_t stream_id, float keep, float* __restrict__ out, long n) {
    const uint64_t seed = state[0], off = state[1];
    const long nblk = (n + 3) / 4;
    for (long b = (long)blockIdx.x * 256 + threadIdx.x; b < nblk; b += (long)gridDim.x * 256) {
        uint32_t c[4] = {(uint32_t)off, (uint32_t)(off >> 32), stream_id, (uint32_t)b};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long i = b * 4 + j;
            if (i >= n) break;
            const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
            out[i] = keep > 0.f ? floorf(keep + u) / keep : u;
        }
    }
}
extern "C" int dfd_rand(const uint64_t* rng_state, uint32_t stream_id, float keep, float* out, long n, dfd_stream stream) {
    if (!rng_state || !out || n < 1 || keep > 1.f) return DFD_EINVAL;
    long grid = ((n + 3) / 4 + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_rand, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, rng_state, stream_id, keep, out, n);
    return DFD_CHECK_LAUNCH();
}
// once per forward pass: every BatchNorm's num_batches_tracked += 1, Philox offset += 1
__global__ void k_step_tick(const int64_t* __restrict__ counter_ptrs, int n, uint64_t* __restrict__ rng_state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *reinterpret_cast<int64_t*>(counter_ptrs[i]) += 1;
    if (i == 0 && rng_state) rng_state[1] += 1;
}
extern "C" int dfd_step_tick(const int64_t* counter_ptrs, int ncounters, uint64_t* rng_state, dfd_stream stream) {
    if (ncounters < 0 || (ncounters > 0 && !counter_ptrs)) return DFD_EINVAL;
    const int n = ncounters > 0 ? ncounters : 1;
    hipLaunchKernelGGL(k_step_tick, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, counter_ptrs, ncounters, rng_state);
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// elementwise passes over [rows][C]
// ===========================================================================
static inline int grid_vec(long total, int cap) {
    long b = (total + DFD_THREADS * 4 - 1) / (DFD_THREADS * 4);
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

// out = a[c]*dz + b[c]*y + c[c]
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_affine2_apply(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ coef, T* __restrict__ out,
                long total_vec, int CV, int C) {
    constexpr int V = Vec<T>::N;
    const long stride = (long)gridDim.x * DFD_THREADS;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total_vec; i += stride) {
        const int c0 = (int)(i % CV) * V;
        float d[V], yv[V], ka[V], kb[V], kc[V];
        Vec<T>::load(dz + i * V, d);
        Vec<T>::load(y + i * V, yv);
        load_f32<V>(coef + c0, ka);
        load_f32<V>(coef + C + c0, kb);
        load_f32<V>(coef + 2 * C + c0, kc);
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = fmaf(ka[j], d[j], fmaf(kb[j], yv[j], kc[j]));
        Vec<T>::store(out + i * V, d);
    }
}
extern "C" int dfd_affine2_apply(int dtype, const void* dz, const void* y, const float* coef, void* out, long rows, int C,
                                 dfd_stream stream) {
    if (!dz || !y || !coef || !out || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        const int CV = C / 8; const long tot = rows * CV;
        hipLaunchKernelGGL((k_affine2_apply<bf16>), dim3(grid_vec(tot, 16384)), dim3(DFD_THREADS), 0, st, (const bf16*)dz, (const bf16*)y, coef, (bf16*)out, tot, CV, C);
    } else if (dtype == DFD_F32) {
        const int CV = C / 4; const long tot = rows * CV;
        hipLaunchKernelGGL((k_affine2_apply<float>), dim3(grid_vec(tot, 16384)), dim3(DFD_THREADS), 0, st, (const float*)dz, (const float*)y, coef, (float*)out, tot, CV, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

// out = act(scale*y + shift + other)   (bnstate == NULL: y used as is; other == NULL: no addend)
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_bn_add_act(const T* __restrict__ y, const float* __restrict__ bnstate, const T* __restrict__ other, T* __restrict__ out,
             long total_vec, int CV, int C) {
    constexpr int V = Vec<T>::N;
    const long stride = (long)gridDim.x * DFD_THREADS;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total_vec; i += stride) {
        const int c0 = (int)(i % CV) * V;
        float v[V], o[V];
        Vec<T>::load(y + i * V, v);
        if (bnstate) {
            float sc[V], sh[V];
            load_f32<V>(bnstate + c0, sc);
            load_f32<V>(bnstate + C + c0, sh);
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = fmaf(sc[j], v[j], sh[j]);
        }
        if (other) {
            Vec<T>::load(other + i * V, o);
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] += o[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = act_fwd<ACT>(v[j]);
        Vec<T>::store(out + i * V, v);
    }
}
template <typename T>
static int bn_add_act_t(const void* y, const float* bnstate, const void* other, int act, void* out, long rows, int C, hipStream_t st) {
    const int CV = C / Vec<T>::N;
    const long tot = rows * CV;
    const int grid = grid_vec(tot, 16384);
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_bn_add_act<T, ACT>), dim3(grid), dim3(DFD_THREADS), 0, st, (const T*)y, bnstate, (const T*)other, (T*)out, tot, CV, C);
    });
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_add_act(int dtype, const void* y, const float* bnstate, const void* other, int act, void* out, long rows,
                              int C, dfd_stream stream) {
    if (!y || !out || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) return bn_add_act_t<bf16>(y, bnstate, other, act, out, rows, C, st);
    if (dtype == DFD_F32) return bn_add_act_t<float>(y, bnstate, other, act, out, rows, C, st);
    return DFD_EINVAL;
}

// d = g * act'(scale*y + shift + other); writes d and, when bnstate is given, the partial sums (d, d*xhat)
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_bn_add_act_bwd(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ bnstate,
                 const T* __restrict__ other, T* __restrict__ d, long rows, int C, ChanMap cm, float* __restrict__ partials) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * 2 * V];
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int c0 = (blockIdx.y * cm.cvb + vl) * V;
    float sc[V], sh[V], mean[V], rstd[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { sc[j] = 1.f; sh[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f; }
    if (bnstate) {
        load_f32<V>(bnstate + c0, sc);
        load_f32<V>(bnstate + C + c0, sh);
        load_f32<V>(bnstate + 2 * C + c0, mean);
        load_f32<V>(bnstate + 3 * C + c0, rstd);
    }
    float acc[2 * V];
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) acc[j] = 0.f;
    if (active) {
        const long step = (long)gridDim.x * cm.rpb;
        for (long r = (long)blockIdx.x * cm.rpb + rl; r < rows; r += step) {
            float gv[V], yv[V], ov[V];
            Vec<T>::load(g + r * C + c0, gv);
            Vec<T>::load(y + r * C + c0, yv);
#pragma unroll
            for (int j = 0; j < V; ++j) ov[j] = 0.f;
            if (other) Vec<T>::load(other + r * C + c0, ov);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float z = fmaf(sc[j], yv[j], sh[j]) + ov[j];
                const float dd = round_to<T>(gv[j] * act_grad<ACT>(z));
                gv[j] = dd;
                acc[j] += dd;
                acc[V + j] += dd * (yv[j] - mean[j]) * rstd[j];
            }
            Vec<T>::store(d + r * C + c0, gv);
        }
    }
    if (partials) {
        reduce_rowlanes<2 * V>(acc, red, cm.cvb, cm.rpb, vl, rl, active);
        if (rl == 0) {
            float* p = partials + (long)blockIdx.x * 2 * C;
            float a0[V], a1[V];
#pragma unroll
            for (int j = 0; j < V; ++j) { a0[j] = acc[j]; a1[j] = acc[V + j]; }
            store_f32<V>(p + c0, a0);
            store_f32<V>(p + C + c0, a1);
        }
    }
}
static inline int vit_pick_parts(long rows, int rpb, int pcap) {
    int cap = pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS;
    long want = (rows + (long)rpb * 4 - 1) / ((long)rpb * 4);
    if (want < 1) want = 1;
    if (want > cap) want = cap;
    return (int)want;
}
template <typename T>
static int bn_add_act_bwd_t(const void* g, const void* y, const float* bnstate, const void* other, int act, void* d, long rows,
                            int C, float* partials, int pcap, int* nparts, hipStream_t st) {
    const ChanMap cm = make_chanmap(C, Vec<T>::N);
    const int P = vit_pick_parts(rows, cm.rpb, partials ? pcap : DFD_MAX_PARTIALS);
    if (partials) *nparts = P;
    dim3 grid(P, cm.nvc);
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_bn_add_act_bwd<T, ACT>), grid, dim3(DFD_THREADS), 0, st, (const T*)g, (const T*)y, bnstate,
                           (const T*)other, (T*)d, rows, C, cm, partials);
    });
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_add_act_bwd(int dtype, const void* g, const void* y, const float* bnstate, const void* other, int act,
                                  void* d, long rows, int C, float* partials, int pcap, int* nparts, dfd_stream stream) {
    if (!g || !y || !d || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    if (partials && (!nparts || pcap < 1 || !bnstate)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) return bn_add_act_bwd_t<bf16>(g, y, bnstate, other, act, d, rows, C, partials, pcap, nparts, st);
    if (dtype == DFD_F32) return bn_add_act_bwd_t<float>(g, y, bnstate, other, act, d, rows, C, partials, pcap, nparts, st);
    return DFD_EINVAL;
}

// per-channel (sum, sumsq) partial slabs of a plain tensor (BatchNorm on a materialised input)
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_channel_stats(const T* __restrict__ x, long rows, int C, ChanMap cm, float* __restrict__ partials) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * 2 * V];
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int c0 = (blockIdx.y * cm.cvb + vl) * V;
    float acc[2 * V];
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) acc[j] = 0.f;
    if (active) {
        const long step = (long)gridDim.x * cm.rpb;
        for (long r = (long)blockIdx.x * cm.rpb + rl; r < rows; r += step) {
            float v[V];
            Vec<T>::load(x + r * C + c0, v);
#pragma unroll
            for (int j = 0; j < V; ++j) { acc[j] += v[j]; acc[V + j] = fmaf(v[j], v[j], acc[V + j]); }
        }
    }
    reduce_rowlanes<2 * V>(acc, red, cm.cvb, cm.rpb, vl, rl, active);
    if (rl == 0) {
        float* p = partials + (long)blockIdx.x * 2 * C;
        float a0[V], a1[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { a0[j] = acc[j]; a1[j] = acc[V + j]; }
        store_f32<V>(p + c0, a0);
        store_f32<V>(p + C + c0, a1);
    }
}
extern "C" int dfd_channel_stats(int dtype, const void* x, long rows, int C, float* partials, int pcap, int* nparts,
                                 dfd_stream stream) {
    if (!x || !partials || !nparts || pcap < 1 || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        const ChanMap cm = make_chanmap(C, 8);
        const int P = vit_pick_parts(rows, cm.rpb, pcap);
        *nparts = P;
        hipLaunchKernelGGL((k_channel_stats<bf16>), dim3(P, cm.nvc), dim3(DFD_THREADS), 0, st, (const bf16*)x, rows, C, cm, partials);
    } else if (dtype == DFD_F32) {
        const ChanMap cm = make_chanmap(C, 4);
        const int P = vit_pick_parts(rows, cm.rpb, pcap);
        *nparts = P;
        hipLaunchKernelGGL((k_channel_stats<float>), dim3(P, cm.nvc), dim3(DFD_THREADS), 0, st, (const float*)x, rows, C, cm, partials);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_sum_rows(float* partials, int P, long L, float* out, int accumulate, dfd_stream stream) {
    if (!partials || !out || P < 1 || L < 1) return DFD_EINVAL;
    // the caller's next launch may read `out`: never left to an open batch
    return dfd_launch_sum_partials(partials, P, L, out, accumulate, (hipStream_t)stream, false);
}
// the same, but inside dfd_sum_batch_begin / _end the sum is recorded and launched with the batch: for results nobody reads
// before the end of the block (a parameter gradient summed straight into its gradient-arena slot)
extern "C" int dfd_sum_rows_deferred(float* partials, int P, long L, float* out, int accumulate, dfd_stream stream) {
    if (!partials || !out || P < 1 || L < 1) return DFD_EINVAL;
    return dfd_launch_sum_partials(partials, P, L, out, accumulate, (hipStream_t)stream, true);
}

// ===========================================================================
// bilinear x2 upsampling (align_corners = False), fused with the activation that follows it
// (timm Attention2d: x = upsample(attn@v + v_local); x = act(x)).
// out[Y] = w0*s[i0] + w1*s[i1]:  Y = 2m: i0 = max(m-1, 0), i1 = m, (0.25, 0.75);  Y = 2m+1: i0 = m, i1 = min(m+1, h-1), (0.75, 0.25)
// ===========================================================================
__device__ __forceinline__ void up2_taps(int Y, int h, int& i0, int& i1, float& w0, float& w1) {
    const int m = Y >> 1;
    if (Y & 1) { i0 = m; i1 = m + 1 < h ? m + 1 : h - 1; w0 = 0.75f; w1 = 0.25f; }
    else { i0 = m > 0 ? m - 1 : 0; i1 = m; w0 = 0.25f; w1 = 0.75f; }
}
template <typename T>
__device__ __forceinline__ void up2_sample(const T* __restrict__ s_img, int h, int w, int C, int c0, int Y, int X, float (&u)[Vec<T>::N]) {
    constexpr int V = Vec<T>::N;
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    up2_taps(Y, h, y0, y1, wy0, wy1);
    up2_taps(X, w, x0, x1, wx0, wx1);
    float a[V], b[V], c[V], d[V];
    Vec<T>::load(s_img + ((long)y0 * w + x0) * C + c0, a);
    Vec<T>::load(s_img + ((long)y0 * w + x1) * C + c0, b);
    Vec<T>::load(s_img + ((long)y1 * w + x0) * C + c0, c);
    Vec<T>::load(s_img + ((long)y1 * w + x1) * C + c0, d);
#pragma unroll
    for (int j = 0; j < V; ++j) u[j] = wy0 * (wx0 * a[j] + wx1 * b[j]) + wy1 * (wx0 * c[j] + wx1 * d[j]);
}
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_up2_act_fwd(const T* __restrict__ s, T* __restrict__ out, int N, int h, int w, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V, H2 = 2 * h, W2 = 2 * w;
    const long total = (long)N * H2 * W2 * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int X = (int)(tq % W2); tq /= W2;
        const int Y = (int)(tq % H2);
        const long n = tq / H2;
        float u[V];
        up2_sample<T>(s + n * h * w * (long)C, h, w, C, cv * V, Y, X, u);
#pragma unroll
        for (int j = 0; j < V; ++j) u[j] = act_fwd<ACT>(u[j]);
        Vec<T>::store(out + i * V, u);
    }
}
// pass 1 of the backward (optional, needs a workspace the size of g): gq[Y][X] = g[Y][X] * act'(up(s)[Y][X]); pass 2 is then
// the activation-free transposed interpolation of gq.  One activation derivative per OUTPUT element instead of one per tap.
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_up2_act_grad(const T* __restrict__ g, const T* __restrict__ s, T* __restrict__ gq, int N, int h, int w, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V, H2 = 2 * h, W2 = 2 * w;
    const long total = (long)N * H2 * W2 * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int X = (int)(tq % W2); tq /= W2;
        const int Y = (int)(tq % H2);
        const long n = tq / H2;
        float u[V], gv[V];
        up2_sample<T>(s + n * h * w * (long)C, h, w, C, cv * V, Y, X, u);
        Vec<T>::load(g + i * V, gv);
#pragma unroll
        for (int j = 0; j < V; ++j) gv[j] *= act_grad<ACT>(u[j]);
        Vec<T>::store(gq + i * V, gv);
    }
}
// ds[m][k] = sum over the (at most 4x4) outputs (Y, X) that read s[m][k]:  wy * wx * g[Y][X] * act'(up(s)[Y][X])
template <typename T, int ACT>
__global__ void __launch_bounds__(DFD_THREADS)
k_up2_act_bwd(const T* __restrict__ g, const T* __restrict__ s, T* __restrict__ ds, int N, int h, int w, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V, H2 = 2 * h, W2 = 2 * w;
    const long total = (long)N * h * w * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int k = (int)(tq % w); tq /= w;
        const int m = (int)(tq % h);
        const long n = tq / h;
        const T* s_img = s + n * h * w * (long)C;
        const T* g_img = g + n * H2 * W2 * (long)C;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int Y = 2 * m - 1; Y <= 2 * m + 2; ++Y) {
            if (Y < 0 || Y >= H2) continue;
            int y0, y1; float wy0, wy1;
            up2_taps(Y, h, y0, y1, wy0, wy1);
            const float wy = (y0 == m ? wy0 : 0.f) + (y1 == m ? wy1 : 0.f);
            if (wy == 0.f) continue;
            for (int X = 2 * k - 1; X <= 2 * k + 2; ++X) {
                if (X < 0 || X >= W2) continue;
                int x0, x1; float wx0, wx1;
                up2_taps(X, w, x0, x1, wx0, wx1);
                const float wx = (x0 == k ? wx0 : 0.f) + (x1 == k ? wx1 : 0.f);
                if (wx == 0.f) continue;
                float gv[V], u[V];
                Vec<T>::load(g_img + ((long)Y * W2 + X) * C + cv * V, gv);
                if constexpr (ACT != DFD_ACT_NONE) {
                    up2_sample<T>(s_img, h, w, C, cv * V, Y, X, u);
#pragma unroll
                    for (int j = 0; j < V; ++j) gv[j] *= act_grad<ACT>(u[j]);
                }
                const float ww = wy * wx;
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] = fmaf(ww, gv[j], acc[j]);
            }
        }
        Vec<T>::store(ds + i * V, acc);
    }
}
template <typename T>
static int up2_t(bool bwd, const void* a, const void* s, int act, void* out, int N, int h, int w, int C, hipStream_t st, void* ws = nullptr) {
    const long tot = (long)N * h * w * (C / Vec<T>::N) * (bwd ? 1 : 4);
    long grid = (tot + DFD_THREADS - 1) / DFD_THREADS;
    if (grid > 16384) grid = 16384;
    if (bwd && ws && act != DFD_ACT_NONE) {
        long grid4 = (tot * 4 + DFD_THREADS - 1) / DFD_THREADS;
        if (grid4 > 16384) grid4 = 16384;
        DISPATCH_ACT(act, {
            hipLaunchKernelGGL((k_up2_act_grad<T, ACT>), dim3((unsigned)grid4), dim3(DFD_THREADS), 0, st, (const T*)a, (const T*)s, (T*)ws, N, h, w, C);
        });
        hipLaunchKernelGGL((k_up2_act_bwd<T, DFD_ACT_NONE>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const T*)ws, (const T*)s, (T*)out, N, h, w, C);
        return DFD_CHECK_LAUNCH();
    }
    DISPATCH_ACT(act, {
        if (bwd) hipLaunchKernelGGL((k_up2_act_bwd<T, ACT>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const T*)a, (const T*)s, (T*)out, N, h, w, C);
        else hipLaunchKernelGGL((k_up2_act_fwd<T, ACT>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const T*)s, (T*)out, N, h, w, C);
    });
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_up2_act_fwd(int dtype, const void* s, int act, void* out, int N, int h, int w, int C, dfd_stream stream) {
    if (!s || !out || N < 1 || h < 1 || w < 1 || C < 8 || C % 8) return DFD_EINVAL;
    if (dtype == DFD_BF16) return up2_t<bf16>(false, nullptr, s, act, out, N, h, w, C, (hipStream_t)stream);
    if (dtype == DFD_F32) return up2_t<float>(false, nullptr, s, act, out, N, h, w, C, (hipStream_t)stream);
    return DFD_EINVAL;
}
extern "C" int dfd_up2_act_bwd(int dtype, const void* g, const void* s, int act, void* ds, int N, int h, int w, int C, void* ws,
                               dfd_stream stream) {
    if (!g || !s || !ds || N < 1 || h < 1 || w < 1 || C < 8 || C % 8) return DFD_EINVAL;
    if (dtype == DFD_BF16) return up2_t<bf16>(true, g, s, act, ds, N, h, w, C, (hipStream_t)stream, ws);
    if (dtype == DFD_F32) return up2_t<float>(true, g, s, act, ds, N, h, w, C, (hipStream_t)stream, ws);
    return DFD_EINVAL;
}

// ---------------------------------------------------------------------------
// strided pixel subsample: mode 0: out[n,i,j,:] = a[n,i,j,:] + bias[:] + x[n, i*s, j*s, :]   (LocalGlobalQuery: local + pool)
//                          mode 1: x[n, i*s, j*s, :] += a[n,i,j,:]                          (its gradient into x)
// ---------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void __launch_bounds__(DFD_THREADS)
k_subsample(const T* __restrict__ a, const float* __restrict__ bias, T* __restrict__ x, T* __restrict__ out, int N, int H, int W,
            int h, int w, int stride, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V;
    const long total = (long)N * h * w * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int jx = (int)(tq % w); tq /= w;
        const int iy = (int)(tq % h);
        const long n = tq / h;
        T* xp = x + ((n * H + (long)iy * stride) * W + (long)jx * stride) * C + cv * V;
        float av[V], xv[V];
        Vec<T>::load(a + i * V, av);
        Vec<T>::load(xp, xv);
        if constexpr (MODE == 0) {
            float b[V];
#pragma unroll
            for (int j = 0; j < V; ++j) b[j] = 0.f;
            if (bias) load_f32<V>(bias + cv * V, b);
#pragma unroll
            for (int j = 0; j < V; ++j) av[j] = av[j] + b[j] + xv[j];
            Vec<T>::store(out + i * V, av);
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) xv[j] += av[j];
            Vec<T>::store(xp, xv);
        }
    }
}
extern "C" int dfd_subsample_add(int dtype, const void* a, const float* bias, const void* x, void* out, int N, int H, int W,
                                 int stride, int C, dfd_stream stream) {
    if (!a || !x || !out || N < 1 || H < 1 || W < 1 || stride < 1 || C < 8 || C % 8) return DFD_EINVAL;
    const int h = (H - 1) / stride + 1, w = (W - 1) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = ((long)N * h * w * (C / 8) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_subsample<bf16, 0>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)a, bias, (bf16*)x, (bf16*)out, N, H, W, h, w, stride, C);
    } else if (dtype == DFD_F32) {
        long grid = ((long)N * h * w * (C / 4) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_subsample<float, 0>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)a, bias, (float*)x, (float*)out, N, H, W, h, w, stride, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_subsample_add_bwd(int dtype, const void* g, void* dx, int N, int H, int W, int stride, int C, dfd_stream stream) {
    if (!g || !dx || N < 1 || H < 1 || W < 1 || stride < 1 || C < 8 || C % 8) return DFD_EINVAL;
    const int h = (H - 1) / stride + 1, w = (W - 1) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = ((long)N * h * w * (C / 8) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_subsample<bf16, 1>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)g, nullptr, (bf16*)dx, nullptr, N, H, W, h, w, stride, C);
    } else if (dtype == DFD_F32) {
        long grid = ((long)N * h * w * (C / 4) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_subsample<float, 1>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)g, nullptr, (float*)dx, nullptr, N, H, W, h, w, stride, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// small strided batched GEMM:  C[b,h,m,n] = alpha * sum_k A[b,h,m,k] * B[b,h,k,n] (+ bias[h,m,n])
// element (b, h, r, c) of an operand sits at base + b*sb + h*sh + r*sr + c*sc (in elements), so the
// attention operands are read straight out of the NHWC projection outputs and the result is written
// straight into the NHWC layout the next 1x1 convolution reads.  One workgroup per (b, h): both
// operands are staged once in LDS as f32; thread e owns output elements e, e+256, ...
// ===========================================================================
template <int DT> struct ElemIO;
template <> struct ElemIO<DFD_F32> {
    __device__ static float ld(const void* p, long i) { return reinterpret_cast<const float*>(p)[i]; }
    __device__ static void st(void* p, long i, float v) { reinterpret_cast<float*>(p)[i] = v; }
};
template <> struct ElemIO<DFD_BF16> {
    __device__ static float ld(const void* p, long i) { return bf2f(reinterpret_cast<const unsigned short*>(p)[i]); }
    __device__ static void st(void* p, long i, float v) { reinterpret_cast<unsigned short*>(p)[i] = f2bf(v); }
};
struct MatDesc { long sb, sh, sr, sc; };

// 4x4 register tiles: LDS holds A transposed ([k][Mp]) and B ([k][Np]) with Mp, Np padded to multiples of 4 (pad = 0),
// so a thread reads one float4 of each per k and does 16 FMAs; staging walks the operand's unit-stride dimension with
// the 32 low lanes (no integer divisions anywhere).
template <int DT>
__device__ __forceinline__ void bg_stage(float* __restrict__ dst, int ld, const void* __restrict__ src, long base, long s_outer, long s_inner,
                                         int n_outer, int n_inner, bool transpose, int round) {
    // element (o, i) at base + o*s_outer + i*s_inner with i the fast (ideally unit-stride) index;
    // stored at dst[o*ld + i] (transpose == false) or dst[i*ld + o] (transpose == true)
    // four outer rows per pass, their loads issued back to back from clamped (always valid) coordinates and only then
    // used: one load at a time inside the bounds check, each was waited for (vmcnt(0)) before the next was requested
    if constexpr (DT == DFD_BF16) {
        // unit-stride bf16 rows whose length and pitch are multiples of 8 (q, k, v, dO of the attention layers: head_dim
        // contiguous): 16-byte loads, two items per thread in flight — 1/8 of the load instructions of the element path
        const unsigned short* sp = reinterpret_cast<const unsigned short*>(src) + base;
        if (s_inner == 1 && (n_inner & 7) == 0 && (s_outer & 7) == 0 && (reinterpret_cast<unsigned long>(sp) & 15) == 0) {
            const int nv = n_inner >> 3, total = n_outer * nv;
            for (int idx0 = threadIdx.x; idx0 < total; idx0 += 2 * DFD_THREADS) {
                uint4 q[2];
                int oo[2], iv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int idx = idx0 + u * DFD_THREADS < total ? idx0 + u * DFD_THREADS : total - 1;
                    oo[u] = idx / nv;
                    iv[u] = idx - oo[u] * nv;
                    q[u] = *reinterpret_cast<const uint4*>(sp + (long)oo[u] * s_outer + iv[u] * 8);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (idx0 + u * DFD_THREADS >= total) continue;
                    float v[8];
                    Vec<bf16>::unpack(q[u], v);         // bf16 -> f32 is exact; `round` (to bf16) is the identity here
                    if (!transpose) {
                        float* d = dst + oo[u] * ld + iv[u] * 8;        // ld % 4 == 0: 16-byte aligned
                        *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) dst[(iv[u] * 8 + j) * ld + oo[u]] = v[j];
                    }
                }
            }
            return;
        }
    }
    const int li = threadIdx.x & 31, lo = threadIdx.x >> 5;
    constexpr int RS = DFD_THREADS / 32, U = 4;
    for (int o0 = lo; o0 < n_outer; o0 += RS * U)
        for (int i = li; i < ((n_inner + 31) & ~31); i += 32) {
            const int ic = i < n_inner ? i : n_inner - 1;
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = o0 + u * RS;
                v[u] = ElemIO<DT>::ld(src, base + (long)(o < n_outer ? o : n_outer - 1) * s_outer + ic * s_inner);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int o = o0 + u * RS;
                if (o < n_outer && i < n_inner) {
                    const float x = round ? bf2f(f2bf(v[u])) : v[u];
                    dst[transpose ? i * ld + o : o * ld + i] = x;
                }
            }
        }
}
template <int DA, int DB, int DC>
__global__ void __launch_bounds__(DFD_THREADS)
k_bgemm(const void* __restrict__ A, MatDesc da, const void* __restrict__ B, MatDesc db, void* __restrict__ C, MatDesc dc,
        const float* __restrict__ bias, float alpha, int nh, int M, int N, int K, int round_a, int round_b) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Mp = (M + 3) & ~3, Np = (N + 3) & ~3;
    float* sa = sm;                       // [K][Mp]
    float* sb = sm + (long)K * Mp;        // [K][Np]
    const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
    const long a0 = (long)b * da.sb + (long)h * da.sh, b0 = (long)b * db.sb + (long)h * db.sh, c0 = (long)b * dc.sb + (long)h * dc.sh;
    const int t = threadIdx.x;
    // zero the pad columns (read as operands of discarded outputs only, but keep them finite)
    for (int i = t; i < K * (Mp - M); i += DFD_THREADS) { const int k = i / (Mp - M); sa[k * Mp + M + (i - k * (Mp - M))] = 0.f; }
    for (int i = t; i < K * (Np - N); i += DFD_THREADS) { const int k = i / (Np - N); sb[k * Np + N + (i - k * (Np - N))] = 0.f; }
    // A is [m][k]: fast index = whichever has the smaller stride
    if (da.sc <= da.sr) bg_stage<DA>(sa, Mp, A, a0, da.sr, da.sc, M, K, true, round_a);        // outer m, inner k -> sa[k][m]
    else bg_stage<DA>(sa, Mp, A, a0, da.sc, da.sr, K, M, false, round_a);                     // outer k, inner m -> sa[k][m]
    if (db.sc <= db.sr) bg_stage<DB>(sb, Np, B, b0, db.sr, db.sc, K, N, false, round_b);       // outer k, inner n -> sb[k][n]
    else bg_stage<DB>(sb, Np, B, b0, db.sc, db.sr, N, K, true, round_b);                      // outer n, inner k -> sb[k][n]
    __syncthreads();
    const int tm = Mp >> 2, tn = Np >> 2;
    const bool n_fast = dc.sc <= dc.sr;               // neighbouring threads along C's unit-stride dimension
    for (int tile = t; tile < tm * tn; tile += DFD_THREADS) {
        int im, in;
        if (n_fast) { im = tile / tn; in = tile - im * tn; } else { in = tile / tm; im = tile - in * tm; }
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        const float* pa = sa + 4 * im;
        const float* pb = sb + 4 * in;
        for (int k = 0; k < K; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(pa + k * Mp);
            const float4 bv = *reinterpret_cast<const float4*>(pb + k * Np);
            const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
        // the tile's 16 bias values are requested together (clamped coordinates) before the first one is used
        float bz[4][4];
        if (bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 4 * im + i < M ? 4 * im + i : M - 1, n = 4 * in + j < N ? 4 * in + j : N - 1;
                    bz[i][j] = bias[((long)h * M + m) * N + n];
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = 4 * im + i;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 4 * in + j;
                if (n >= N) continue;
                float v = alpha * acc[i][j];
                if (bias) v += bz[i][j];
                ElemIO<DC>::st(C, c0 + m * dc.sr + n * dc.sc, v);
            }
        }
    }
}
// long-K variant for tiny outputs (M, N <= 16: talking-head weight gradients, K = Nq*Nk): K is walked in
// chunks of 256 with one accumulator per thread
#define BG_KC 256
template <int DA, int DB, int DC>
__global__ void __launch_bounds__(DFD_THREADS)
k_bgemm_small(const void* __restrict__ A, MatDesc da, const void* __restrict__ B, MatDesc db, void* __restrict__ C, MatDesc dc,
              const float* __restrict__ bias, float alpha, int nh, int M, int N, int K) {
    __shared__ float sa[16 * (BG_KC + 1)];
    __shared__ float sb[BG_KC * 16];
    const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
    const long a0 = (long)b * da.sb + (long)h * da.sh, b0 = (long)b * db.sb + (long)h * db.sh, c0 = (long)b * dc.sb + (long)h * dc.sh;
    const int t = threadIdx.x;
    const int m = t / N, n = t - m * N;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += BG_KC) {
        const int kc = K - k0 < BG_KC ? K - k0 : BG_KC;
        __syncthreads();
        if (da.sc <= da.sr) {
            for (int i = t; i < M * kc; i += DFD_THREADS) { const int r = i / kc, c = i - r * kc; sa[r * (BG_KC + 1) + c] = ElemIO<DA>::ld(A, a0 + r * da.sr + (long)(k0 + c) * da.sc); }
        } else {
            for (int i = t; i < M * kc; i += DFD_THREADS) { const int c = i / M, r = i - c * M; sa[r * (BG_KC + 1) + c] = ElemIO<DA>::ld(A, a0 + r * da.sr + (long)(k0 + c) * da.sc); }
        }
        if (db.sc <= db.sr) {
            for (int i = t; i < kc * N; i += DFD_THREADS) { const int r = i / N, c = i - r * N; sb[r * N + c] = ElemIO<DB>::ld(B, b0 + (long)(k0 + r) * db.sr + c * db.sc); }
        } else {
            for (int i = t; i < kc * N; i += DFD_THREADS) { const int c = i / kc, r = i - c * kc; sb[r * N + c] = ElemIO<DB>::ld(B, b0 + (long)(k0 + r) * db.sr + c * db.sc); }
        }
        __syncthreads();
        if (t < M * N) {
            const float* ar = sa + m * (BG_KC + 1);
            for (int k = 0; k < kc; ++k) acc = fmaf(ar[k], sb[k * N + n], acc);
        }
    }
    if (t < M * N) {
        float v = alpha * acc;
        if (bias) v += bias[((long)h * M + m) * N + n];
        ElemIO<DC>::st(C, c0 + m * dc.sr + n * dc.sc, v);
    }
}

// K-parallel variant for tiny outputs with a long reduction (M, N <= 8: the talking-head weight gradients,
// dW[g][h] = sum over 2401 positions): thread t walks k = t, t + 256, ... with all M*N sums in registers, reading
// both operands straight from global memory (coalesced along k), then the 256 partial sums of every output are
// combined through LDS in a fixed order.
template <int DA, int DB, int DC>
__global__ void __launch_bounds__(DFD_THREADS)
k_bgemm_kred(const void* __restrict__ A, MatDesc da, const void* __restrict__ B, MatDesc db, void* __restrict__ C, MatDesc dc,
             const float* __restrict__ bias, float alpha, int nh, int M, int N, int K) {
    __shared__ float red[64 * DFD_THREADS / 4];          // 64 outputs x 64 lanes per pass (four passes of 64 threads)
    const int bh = blockIdx.x, b = bh / nh, h = bh - b * nh;
    const long a0 = (long)b * da.sb + (long)h * da.sh, b0 = (long)b * db.sb + (long)h * db.sh, c0 = (long)b * dc.sb + (long)h * dc.sh;
    const int t = threadIdx.x;
    float acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    for (int k = t; k < K; k += DFD_THREADS) {
        float av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = i < M ? ElemIO<DA>::ld(A, a0 + i * da.sr + (long)k * da.sc) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = j < N ? ElemIO<DB>::ld(B, b0 + (long)k * db.sr + j * db.sc) : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    // wave-level sums first (fixed butterfly), then the four wave results through LDS
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[i][j];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            acc[i][j] = v;
        }
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[wave * 64 + i * 8 + j] = acc[i][j];
    }
    __syncthreads();
    if (t < 64) {
        const int m = t >> 3, n = t & 7;
        if (m < M && n < N) {
            float v = alpha * (red[t] + red[64 + t] + red[128 + t] + red[192 + t]);
            if (bias) v += bias[((long)h * M + m) * N + n];
            ElemIO<DC>::st(C, c0 + m * dc.sr + n * dc.sc, v);
        }
    }
}

extern "C" int dfd_bgemm(int dt_a, const void* A, const dfd_mat* sa, int dt_b, const void* B, const dfd_mat* sb, int dt_c,
                         void* C, const dfd_mat* sc, const float* bias, float alpha, int nb, int nh, int M, int N, int K,
                         int round_a, int round_b, dfd_stream stream) {
    if (!A || !B || !C || !sa || !sb || !sc || nb < 1 || nh < 1 || M < 1 || N < 1 || K < 1) return DFD_EINVAL;
    const size_t lds = ((size_t)K * ((M + 3) & ~3) + (size_t)K * ((N + 3) & ~3)) * 4;
    const bool small = M <= 16 && N <= 16 && lds > 48 * 1024 && !round_a && !round_b;
    const bool kred = M <= 8 && N <= 8 && K >= 512 && !round_a && !round_b;
    if (!small && lds > 150 * 1024) return DFD_EUNSUPPORTED;
    const MatDesc da{sa->sb, sa->sh, sa->sr, sa->sc}, db{sb->sb, sb->sh, sb->sr, sb->sc}, dc{sc->sb, sc->sh, sc->sr, sc->sc};
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(nb * nh));
#define BG(DA, DB, DC)                                                                                                  \
    do {                                                                                                                \
        if (kred) {                                                                                                     \
            hipLaunchKernelGGL((k_bgemm_kred<DA, DB, DC>), grid, dim3(DFD_THREADS), 0, st, A, da, B, db, C, dc, bias, alpha, nh, M, N, K); \
            break;                                                                                                      \
        }                                                                                                               \
        if (small) {                                                                                                    \
            hipLaunchKernelGGL((k_bgemm_small<DA, DB, DC>), grid, dim3(DFD_THREADS), 0, st, A, da, B, db, C, dc, bias, alpha, nh, M, N, K); \
            break;                                                                                                      \
        }                                                                                                               \
        auto kern = k_bgemm<DA, DB, DC>;                                                                                \
        struct BgTag;                                                                                                   \
        dfd_allow_lds_once<BgTag>(kern, 150 * 1024);    /* once per instantiation, for the largest panel it accepts */  \
        hipLaunchKernelGGL(kern, grid, dim3(DFD_THREADS), lds, st, A, da, B, db, C, dc, bias, alpha, nh, M, N, K, round_a, round_b); \
    } while (0)
    const int code = dt_a * 4 + dt_b * 2 + dt_c;
    switch (code) {
        case 0: BG(DFD_F32, DFD_F32, DFD_F32); break;
        case 1: BG(DFD_F32, DFD_F32, DFD_BF16); break;
        case 2: BG(DFD_F32, DFD_BF16, DFD_F32); break;
        case 3: BG(DFD_F32, DFD_BF16, DFD_BF16); break;
        case 4: BG(DFD_BF16, DFD_F32, DFD_F32); break;
        case 5: BG(DFD_BF16, DFD_F32, DFD_BF16); break;
        case 6: BG(DFD_BF16, DFD_BF16, DFD_F32); break;
        case 7: BG(DFD_BF16, DFD_BF16, DFD_BF16); break;
        default: return DFD_EINVAL;
    }
#undef BG
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// attention rows: S [B][H][Nq][Nk] f32 -> [T1 = W1*S + b1 ->] P = softmax_j(T1) [-> T2 = W2*P + b2]
// (timm Attention2d.talking_head1/2 are 1x1 convolutions over the HEAD dimension.)
// thread = one (query i, key j) element with all H heads in registers; a workgroup owns R
// consecutive (b, i) rows, R * Nk <= 256.
// ===========================================================================
#define ATT_MAX_H 16
template <bool TALK>
__global__ void __launch_bounds__(DFD_THREADS)
k_attn_softmax_fwd(const float* __restrict__ S, const float* __restrict__ w1, const float* __restrict__ b1,
                   const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ P, float* __restrict__ T2,
                   long rows /* B*Nq */, int H, int Nq, int Nk, int R) {
    __shared__ float st[ATT_MAX_H * 256];          // [h][slot]
    __shared__ float smx[ATT_MAX_H * 64], ssum[ATT_MAX_H * 64];   // [h][row-in-block]
    __shared__ float red[DFD_THREADS];
    __shared__ float sw[2 * ATT_MAX_H * ATT_MAX_H + 2 * ATT_MAX_H];
    const int t = threadIdx.x;
    if (TALK) {
        for (int i = t; i < H * H; i += DFD_THREADS) { sw[i] = w1[i]; sw[ATT_MAX_H * ATT_MAX_H + i] = w2[i]; }
        if (t < H) { sw[2 * ATT_MAX_H * ATT_MAX_H + t] = b1 ? b1[t] : 0.f; sw[2 * ATT_MAX_H * ATT_MAX_H + ATT_MAX_H + t] = b2 ? b2[t] : 0.f; }
        __syncthreads();
    }
    const int rl = t / Nk, j = t - rl * Nk;
    const long row = (long)blockIdx.x * R + rl;
    const bool on = rl < R && row < rows;
    const long b = on ? row / Nq : 0;
    const int i = on ? (int)(row - b * Nq) : 0;
    const long plane = (long)Nq * Nk;
    const long base = (b * H) * plane + (long)i * Nk + j;       // + h*plane
    float v[ATT_MAX_H];
    if (on) {
        float s[ATT_MAX_H];
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) s[h] = h < H ? S[base + h * plane] : 0.f;
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) {
            if (h >= H) { v[h] = 0.f; continue; }
            if (TALK) {
                float a = sw[2 * ATT_MAX_H * ATT_MAX_H + h];
#pragma unroll
                for (int g = 0; g < ATT_MAX_H; ++g) if (g < H) a = fmaf(sw[h * H + g], s[g], a);
                v[h] = a;
            } else v[h] = s[h];
            st[h * 256 + t] = v[h];
        }
    }
    __syncthreads();
    // per (h, row) max and sum of exp: TEAM threads per pair (H * R <= 256 pairs), each over the keys k = m (mod TEAM); partial results
    // through LDS, combined in member order.  (One thread per pair walked the Nk keys twice on its own: with 196 keys and one row per
    // workgroup 8 of 256 threads did 392 dependent LDS reads each — 143 us per call in EfficientFormerV2-S1.)
    {
        const int NP = H * R;
        int TEAM = 1;
        while (TEAM < 32 && TEAM * 2 * NP <= DFD_THREADS) TEAM *= 2;
        const int pr = t / TEAM, m = t - pr * TEAM;
        const int h = pr < NP ? pr / R : 0, r2 = pr < NP ? pr - h * R : 0;
        const bool rowok = pr < NP && (long)blockIdx.x * R + r2 < rows;
        const float* q = st + h * 256 + r2 * Nk;
        float mx = -INFINITY;
        if (rowok) for (int k = m; k < Nk; k += TEAM) mx = fmaxf(mx, q[k]);
        red[t] = mx;
        __syncthreads();
        if (rowok) for (int u = 0; u < TEAM; ++u) mx = fmaxf(mx, red[pr * TEAM + u]);
        float se = 0.f;
        if (rowok) for (int k = m; k < Nk; k += TEAM) se += __expf(q[k] - mx);
        __syncthreads();                                                  // every partial maximum has been read
        red[t] = se;
        __syncthreads();
        if (rowok && m == 0) {
            float sum = 0.f;
            for (int u = 0; u < TEAM; ++u) sum += red[pr * TEAM + u];
            smx[h * 64 + r2] = mx;
            ssum[h * 64 + r2] = 1.0f / sum;
        }
    }
    __syncthreads();
    if (on) {
        float p[ATT_MAX_H];
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) {
            if (h >= H) { p[h] = 0.f; continue; }
            p[h] = __expf(v[h] - smx[h * 64 + rl]) * ssum[h * 64 + rl];
            P[base + h * plane] = p[h];
        }
        if (TALK) {
#pragma unroll
            for (int h = 0; h < ATT_MAX_H; ++h) {
                if (h >= H) continue;
                float a = sw[2 * ATT_MAX_H * ATT_MAX_H + ATT_MAX_H + h];
#pragma unroll
                for (int g = 0; g < ATT_MAX_H; ++g) if (g < H) a = fmaf(sw[ATT_MAX_H * ATT_MAX_H + h * H + g], p[g], a);
                T2[base + h * plane] = a;
            }
        }
    }
}
// backward: dT2 -> dP = W2^T dT2 -> dT1 = P * (dP - sum_j dP*P) -> dS = W1^T dT1.  dT1 is also written
// (TALK) for the talking-head weight gradients, which are plain contractions done by dfd_bgemm.
template <bool TALK>
__global__ void __launch_bounds__(DFD_THREADS)
k_attn_softmax_bwd(const float* __restrict__ dT2, const float* __restrict__ P, const float* __restrict__ w1,
                   const float* __restrict__ w2, float* __restrict__ dT1, float* __restrict__ dS, long rows, int H, int Nq,
                   int Nk, int R) {
    __shared__ float st[ATT_MAX_H * 256];
    __shared__ float sdot[ATT_MAX_H * 64];
    __shared__ float red[DFD_THREADS];
    __shared__ float sw[2 * ATT_MAX_H * ATT_MAX_H];
    const int t = threadIdx.x;
    if (TALK) {
        for (int i = t; i < H * H; i += DFD_THREADS) { sw[i] = w1[i]; sw[ATT_MAX_H * ATT_MAX_H + i] = w2[i]; }
        __syncthreads();
    }
    const int rl = t / Nk, j = t - rl * Nk;
    const long row = (long)blockIdx.x * R + rl;
    const bool on = rl < R && row < rows;
    const long b = on ? row / Nq : 0;
    const int i = on ? (int)(row - b * Nq) : 0;
    const long plane = (long)Nq * Nk;
    const long base = (b * H) * plane + (long)i * Nk + j;
    float dp[ATT_MAX_H], p[ATT_MAX_H];
    if (on) {
        float d2[ATT_MAX_H];
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) { d2[h] = h < H ? dT2[base + h * plane] : 0.f; p[h] = h < H ? P[base + h * plane] : 0.f; }
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) {
            if (h >= H) { dp[h] = 0.f; continue; }
            if (TALK) {
                float a = 0.f;
#pragma unroll
                for (int g = 0; g < ATT_MAX_H; ++g) if (g < H) a = fmaf(sw[ATT_MAX_H * ATT_MAX_H + g * H + h], d2[g], a);   // W2^T
                dp[h] = a;
            } else dp[h] = d2[h];
            st[h * 256 + t] = dp[h] * p[h];
        }
    }
    __syncthreads();
    {   // row sums of dP * P: TEAM threads per (h, row) pair, as in the forward kernel
        const int NP = H * R;
        int TEAM = 1;
        while (TEAM < 32 && TEAM * 2 * NP <= DFD_THREADS) TEAM *= 2;
        const int pr = t / TEAM, m = t - pr * TEAM;
        const int h = pr < NP ? pr / R : 0, r2 = pr < NP ? pr - h * R : 0;
        const bool rowok = pr < NP && (long)blockIdx.x * R + r2 < rows;
        const float* q = st + h * 256 + r2 * Nk;
        float s = 0.f;
        if (rowok) for (int k = m; k < Nk; k += TEAM) s += q[k];
        red[t] = s;
        __syncthreads();
        if (rowok && m == 0) {
            float sum = 0.f;
            for (int u = 0; u < TEAM; ++u) sum += red[pr * TEAM + u];
            sdot[h * 64 + r2] = sum;
        }
    }
    __syncthreads();
    if (on) {
        float d1[ATT_MAX_H];
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) {
            d1[h] = h < H ? p[h] * (dp[h] - sdot[h * 64 + rl]) : 0.f;
            if (TALK && h < H) dT1[base + h * plane] = d1[h];
        }
#pragma unroll
        for (int h = 0; h < ATT_MAX_H; ++h) {
            if (h >= H) continue;
            float a = d1[h];
            if (TALK) {
                a = 0.f;
#pragma unroll
                for (int g = 0; g < ATT_MAX_H; ++g) if (g < H) a = fmaf(sw[g * H + h], d1[g], a);     // W1^T
            }
            dS[base + h * plane] = a;
        }
    }
}
extern "C" int dfd_attn_softmax_fwd(const float* S, const float* th_w1, const float* th_b1, const float* th_w2,
                                    const float* th_b2, float* P, float* T2, int B, int H, int Nq, int Nk, dfd_stream stream) {
    if (!S || !P || B < 1 || H < 1 || H > ATT_MAX_H || Nq < 1 || Nk < 1 || Nk > 256) return DFD_EINVAL;
    const bool talk = th_w1 != nullptr;
    if (talk && (!th_w2 || !T2)) return DFD_EINVAL;
    int R = 256 / Nk;
    if (R > 64) R = 64;
    while (H * R > 256) --R;
    if (R < 1) return DFD_EUNSUPPORTED;
    const long rows = (long)B * Nq;
    const unsigned grid = (unsigned)((rows + R - 1) / R);
    if (talk) hipLaunchKernelGGL((k_attn_softmax_fwd<true>), dim3(grid), dim3(DFD_THREADS), 0, (hipStream_t)stream, S, th_w1, th_b1, th_w2, th_b2, P, T2, rows, H, Nq, Nk, R);
    else hipLaunchKernelGGL((k_attn_softmax_fwd<false>), dim3(grid), dim3(DFD_THREADS), 0, (hipStream_t)stream, S, th_w1, th_b1, th_w2, th_b2, P, T2, rows, H, Nq, Nk, R);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_attn_softmax_bwd(const float* dT2, const float* P, const float* th_w1, const float* th_w2, float* dT1,
                                    float* dS, int B, int H, int Nq, int Nk, dfd_stream stream) {
    if (!dT2 || !P || !dS || B < 1 || H < 1 || H > ATT_MAX_H || Nq < 1 || Nk < 1 || Nk > 256) return DFD_EINVAL;
    const bool talk = th_w1 != nullptr;
    if (talk && (!th_w2 || !dT1)) return DFD_EINVAL;
    int R = 256 / Nk;
    if (R > 64) R = 64;
    while (H * R > 256) --R;
    if (R < 1) return DFD_EUNSUPPORTED;
    const long rows = (long)B * Nq;
    const unsigned grid = (unsigned)((rows + R - 1) / R);
    if (talk) hipLaunchKernelGGL((k_attn_softmax_bwd<true>), dim3(grid), dim3(DFD_THREADS), 0, (hipStream_t)stream, dT2, P, th_w1, th_w2, dT1, dS, rows, H, Nq, Nk, R);
    else hipLaunchKernelGGL((k_attn_softmax_bwd<false>), dim3(grid), dim3(DFD_THREADS), 0, (hipStream_t)stream, dT2, P, th_w1, th_w2, dT1, dS, rows, H, Nq, Nk, R);
    return DFD_CHECK_LAUNCH();
}

// learned attention bias: full[h][l] = table[h][idx[l]];  dtable[h][t] = sum_{l: idx[l]==t} dfull[h][l] (fixed order)
__global__ void k_bias_gather(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ full, int H, int T, long L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * L) return;
    const int h = (int)(i / L);
    const long l = i - (long)h * L;
    full[i] = table[(long)h * T + idx[l]];
}
__global__ void __launch_bounds__(256) k_bias_scatter(const float* __restrict__ dfull, const int* __restrict__ idx, float* __restrict__ dtable, int H,
                                                       int T, long L, int accumulate) {
    // one workgroup per (h, t): threads stride over l, each wave reduces by a butterfly, the four wave sums are added in wave order (fixed).
    // (One WAVE per (h, t) walked all L = Nq * Nk index entries in L / 64 dependent steps: 25 us per call at 196 x 49.)
    __shared__ float sm[4];
    const int pair = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = pair / T, tt = pair - h * T;
    float s = 0.f;
    for (long l = threadIdx.x; l < L; l += 256) if (idx[l] == tt) s += dfull[(long)h * L + l];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) sm[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) dtable[pair] = (accumulate ? dtable[pair] : 0.f) + (((sm[0] + sm[1]) + sm[2]) + sm[3]);
}
extern "C" int dfd_bias_gather(const float* table, const int* idx, float* full, int H, int T, long L, dfd_stream stream) {
    if (!table || !idx || !full || H < 1 || T < 1 || L < 1) return DFD_EINVAL;
    hipLaunchKernelGGL(k_bias_gather, dim3((unsigned)(((long)H * L + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, idx, full, H, T, L);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bias_scatter(const float* dfull, const int* idx, float* dtable, int H, int T, long L, int accumulate,
                                dfd_stream stream) {
    if (!dfull || !idx || !dtable || H < 1 || T < 1 || L < 1) return DFD_EINVAL;
    hipLaunchKernelGGL(k_bias_scatter, dim3((unsigned)(H * T)), dim3(256), 0, (hipStream_t)stream, dfull, idx, dtable, H, T, L, accumulate);
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// dense k x k convolution as im2col + the 1x1 GEMM kernels (EfficientFormerV2 Stem4.conv2 and Downsample.conv,
// FasterViT PatchEmbed.conv2 / ConvBlock / Downsample): col[(n,oy,ox)][(kh,kw,c)] = act(bn(x))[n, oy*s-p+kh, ox*s-p+kw, c]
// ===========================================================================
template <typename T, int ACT, bool PRO>
__global__ void __launch_bounds__(DFD_THREADS)
k_im2col(const T* __restrict__ x, const float* __restrict__ bnstate, T* __restrict__ col, dfd_dwconv_shape s) {
    constexpr int V = Vec<T>::N;
    const int CV = s.C / V, KK = s.k * s.k;
    const long total = (long)s.N * s.Ho * s.Wo * KK * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int tap = (int)(tq % KK); tq /= KK;
        const int ox = (int)(tq % s.Wo); tq /= s.Wo;
        const int oy = (int)(tq % s.Ho);
        const long n = tq / s.Ho;
        const int kh = tap / s.k, kw = tap - kh * s.k;
        const int iy = oy * s.stride - s.pad_top + kh, ix = ox * s.stride - s.pad_left + kw;
        float v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = 0.f;
        if (iy >= 0 && iy < s.H && ix >= 0 && ix < s.W) {
            Vec<T>::load(x + ((n * s.H + iy) * (long)s.W + ix) * s.C + cv * V, v);
            if constexpr (PRO) {
                float sc[V], sh[V];
                load_f32<V>(bnstate + cv * V, sc);
                load_f32<V>(bnstate + s.C + cv * V, sh);
                bn_act_array<ACT, V>(v, sc, sh);
            }
        }
        Vec<T>::store(col + i * V, v);
    }
}
// dx[n,y,x,c] = sum over taps (kh,kw) with (y+p-kh) % s == 0, (x+p-kw) % s == 0 of dcol[(n,oy,ox)][(kh,kw,c)]
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_col2im(const T* __restrict__ dcol, T* __restrict__ dx, dfd_dwconv_shape s) {
    constexpr int V = Vec<T>::N;
    const int CV = s.C / V, KK = s.k * s.k;
    const long total = (long)s.N * s.H * s.W * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        const int ix = (int)(tq % s.W); tq /= s.W;
        const int iy = (int)(tq % s.H);
        const long n = tq / s.H;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        for (int kh = 0; kh < s.k; ++kh) {
            const int ty = iy + s.pad_top - kh;
            if (ty < 0 || ty % s.stride) continue;
            const int oy = ty / s.stride;
            if (oy >= s.Ho) continue;
            for (int kw = 0; kw < s.k; ++kw) {
                const int tx = ix + s.pad_left - kw;
                if (tx < 0 || tx % s.stride) continue;
                const int ox = tx / s.stride;
                if (ox >= s.Wo) continue;
                float v[V];
                Vec<T>::load(dcol + ((((n * s.Ho + oy) * (long)s.Wo + ox) * KK) + kh * s.k + kw) * s.C + cv * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += v[j];
            }
        }
        Vec<T>::store(dx + i * V, acc);
    }
}
static bool conv_shape_ok(const dfd_dwconv_shape* s) {
    if (!s || s->N < 1 || s->H < 1 || s->W < 1 || s->Ho < 1 || s->Wo < 1 || s->C < 8 || s->C % 8) return false;
    if (s->k < 1 || s->k > 7 || s->stride < 1 || s->stride > 4 || s->pad_top < 0 || s->pad_left < 0) return false;
    if ((long)(s->Ho - 1) * s->stride - s->pad_top > s->H - 1 || (long)(s->Wo - 1) * s->stride - s->pad_left > s->W - 1) return false;
    return true;
}
template <typename T>
static int im2col_t(const void* x, const float* bnstate, int act, void* col, const dfd_dwconv_shape* s, hipStream_t st) {
    const long tot = (long)s->N * s->Ho * s->Wo * s->k * s->k * (s->C / Vec<T>::N);
    long grid = (tot + DFD_THREADS * 2 - 1) / (DFD_THREADS * 2);
    if (grid > 32768) grid = 32768;
    if (bnstate) {
        DISPATCH_ACT(act, { hipLaunchKernelGGL((k_im2col<T, ACT, true>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const T*)x, bnstate, (T*)col, *s); });
    } else {
        hipLaunchKernelGGL((k_im2col<T, DFD_ACT_NONE, false>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const T*)x, bnstate, (T*)col, *s);
    }
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_im2col(int dtype, const void* x, const float* in_bnstate, int in_act, void* col, const dfd_dwconv_shape* s,
                          dfd_stream stream) {
    if (!x || !col || !conv_shape_ok(s)) return DFD_EINVAL;
    if (dtype == DFD_BF16) return im2col_t<bf16>(x, in_bnstate, in_act, col, s, (hipStream_t)stream);
    if (dtype == DFD_F32) return im2col_t<float>(x, in_bnstate, in_act, col, s, (hipStream_t)stream);
    return DFD_EINVAL;
}
extern "C" int dfd_col2im(int dtype, const void* dcol, void* dx, const dfd_dwconv_shape* s, dfd_stream stream) {
    if (!dcol || !dx || !conv_shape_ok(s)) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = ((long)s->N * s->H * s->W * (s->C / 8) + 255) / 256; if (grid > 32768) grid = 32768;
        hipLaunchKernelGGL((k_col2im<bf16>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)dcol, (bf16*)dx, *s);
    } else if (dtype == DFD_F32) {
        long grid = ((long)s->N * s->H * s->W * (s->C / 4) + 255) / 256; if (grid > 32768) grid = 32768;
        hipLaunchKernelGGL((k_col2im<float>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)dcol, (float*)dx, *s);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
// torch conv weight [O][I][k][k] f32 <-> GEMM layout [O][(kh,kw,i)]:
//   to_gemm = 1: dst[o][(kh*k+kw)*I + i] = src[o][i][kh][kw];   to_gemm = 0: the inverse (weight gradient back to torch's layout)
//   to_gemm = 2: dst[i][((k-1-kh)*k + (k-1-kw))*O + o] = src[o][i][kh][kw]  (transposed + flipped: data gradient of a stride-1 conv)
__global__ void k_conv_weight_perm(const float* __restrict__ src, float* __restrict__ dst, int O, int I, int k, int to_gemm, int accumulate) {
    const long tot = (long)O * I * k * k;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= tot) return;
    // e indexes the torch layout
    long tq = e;
    const int kw = (int)(tq % k); tq /= k;
    const int kh = (int)(tq % k); tq /= k;
    const int i = (int)(tq % I);
    const long o = tq / I;
    const long g = (o * k * k + (long)kh * k + kw) * I + i;
    if (to_gemm == 2) {
        // data-gradient operand of a stride-1 convolution: [I][(flipped tap, o)], i.e. the weight of the forward
        // convolution that maps the output gradient (O channels) back to the input (I channels)
        dst[((long)i * k * k + (long)(k - 1 - kh) * k + (k - 1 - kw)) * O + o] = src[e];
    } else if (to_gemm) dst[g] = src[e];
    else dst[e] = (accumulate ? dst[e] : 0.f) + src[g];
}
extern "C" int dfd_conv_weight_perm(const float* src, float* dst, int O, int I, int k, int to_gemm, int accumulate, dfd_stream stream) {
    if (!src || !dst || O < 1 || I < 1 || k < 1) return DFD_EINVAL;
    const long tot = (long)O * I * k * k;
    hipLaunchKernelGGL(k_conv_weight_perm, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, O, I, k, to_gemm, accumulate);
    return DFD_CHECK_LAUNCH();
}

// ===========================================================================
// LayerNorm over the channel dimension of [rows][C] (FasterViT token LN / LayerNorm2d), one wave per row.
//   fwd: y = (x - mu) * rstd * gamma + beta; saves (mu, rstd) per row
//   bwd: dx = rstd * (gh - mean(gh) - xhat * mean(gh * xhat)), gh = g * gamma; partial slabs of (g*xhat, g) per workgroup
// ===========================================================================
// Register-resident form (NV = 16-byte vectors per lane, 1 or 2): the row stays in registers between the statistics and the
// output (one read of x — and of g in the backward — instead of three / two), gamma / beta are loaded once per lane, the
// backward keeps its dgamma / dbeta partial sums in registers instead of 32 LDS read-modify-writes per row, and SHORT rows
// share a wave: with CV = C / V vectors per row, lpr = the power of two >= CV lanes form a row group and a wave walks
// 64 / lpr rows at once (C = 256 in bf16 is 32 vectors: two rows per wave instead of 32 idle lanes).  NV = 0: rows longer
// than 128 vectors — the streaming form.
__device__ __forceinline__ int ln_lpr(int CV) { int l = 1; while (l < CV && l < 64) l <<= 1; return l; }
template <typename T, int NV>
__global__ void __launch_bounds__(DFD_THREADS)
k_layernorm_fwd(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                T* __restrict__ y, float* __restrict__ stats, long rows, int C) {
    constexpr int V = Vec<T>::N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CV = C / V;
    if constexpr (NV > 0) {
        const int lpr = NV == 1 ? ln_lpr(CV) : 64, rpw = 64 / lpr;        // lanes per row, rows per wave
        const int sub = lane / lpr, cl = lane - sub * lpr;
        float gm[NV][V], bt[NV][V];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cv = cl + 64 * i;
            if (cv < CV) { load_f32<V>(gamma + cv * V, gm[i]); load_f32<V>(beta + cv * V, bt[i]); }
        }
        for (long r0 = ((long)blockIdx.x * 4 + wave) * rpw; r0 < rows; r0 += (long)gridDim.x * 4 * rpw) {
            const long r = r0 + sub;
            const bool ron = r < rows;
            float v[NV][V];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int cv = cl + 64 * i;
                if (ron && cv < CV) {
                    Vec<T>::load(x + r * C + cv * V, v[i]);
#pragma unroll
                    for (int j = 0; j < V; ++j) s += v[i][j];
                }
            }
            for (int o = lpr >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            const float mu = s / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (ron && cl + 64 * i < CV) {
#pragma unroll
                    for (int j = 0; j < V; ++j) { const float d = v[i][j] - mu; q = fmaf(d, d, q); }
                }
            }
            for (int o = lpr >> 1; o >= 1; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = 1.0f / sqrtf(q / (float)C + eps);
            if (ron && cl == 0 && stats) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int cv = cl + 64 * i;
                if (ron && cv < CV) {
#pragma unroll
                    for (int j = 0; j < V; ++j) v[i][j] = fmaf((v[i][j] - mu) * rstd, gm[i][j], bt[i][j]);
                    Vec<T>::store(y + r * C + cv * V, v[i]);
                }
            }
        }
    } else {
        for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
            float s = 0.f;
            for (int cv = lane; cv < CV; cv += 64) {
                float v[V];
                Vec<T>::load(x + r * C + cv * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) s += v[j];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            const float mu = s / (float)C;
            float q = 0.f;
            for (int cv = lane; cv < CV; cv += 64) {
                float v[V];
                Vec<T>::load(x + r * C + cv * V, v);
#pragma unroll
                for (int j = 0; j < V; ++j) { const float d = v[j] - mu; q = fmaf(d, d, q); }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = 1.0f / sqrtf(q / (float)C + eps);
            if (lane == 0 && stats) { stats[2 * r] = mu; stats[2 * r + 1] = rstd; }
            for (int cv = lane; cv < CV; cv += 64) {
                float v[V], gm[V], bt[V];
                Vec<T>::load(x + r * C + cv * V, v);
                load_f32<V>(gamma + cv * V, gm);
                load_f32<V>(beta + cv * V, bt);
#pragma unroll
                for (int j = 0; j < V; ++j) v[j] = fmaf((v[j] - mu) * rstd, gm[j], bt[j]);
                Vec<T>::store(y + r * C + cv * V, v);
            }
        }
    }
}
// residual (optional): dx = LN-backward(g) + residual — the skip connection's gradient, which used to be a separate add kernel
template <typename T, int NV>
__global__ void __launch_bounds__(DFD_THREADS)
k_layernorm_bwd(const T* __restrict__ g, const T* __restrict__ x, const float* __restrict__ gamma,
                const float* __restrict__ stats, const T* __restrict__ residual, T* __restrict__ dx, float* __restrict__ partials,
                long rows, int C) {
    constexpr int V = Vec<T>::N;
    extern __shared__ float sacc[];            // [4 waves][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int CV = C / V;
    float* mine = sacc + (long)wave * 2 * C;
    if constexpr (NV > 0) {
        const int lpr = NV == 1 ? ln_lpr(CV) : 64, rpw = 64 / lpr;
        const int sub = lane / lpr, cl = lane - sub * lpr;
        float gm[NV][V], dg[NV][V], db[NV][V];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cv = cl + 64 * i;
            if (cv < CV) load_f32<V>(gamma + cv * V, gm[i]);
#pragma unroll
            for (int j = 0; j < V; ++j) { dg[i][j] = 0.f; db[i][j] = 0.f; }
        }
        for (long r0 = ((long)blockIdx.x * 4 + wave) * rpw; r0 < rows; r0 += (long)gridDim.x * 4 * rpw) {
            const long r = r0 + sub;
            const bool ron = r < rows;
            const float mu = ron ? stats[2 * r] : 0.f, rstd = ron ? stats[2 * r + 1] : 0.f;
            float gv[NV][V], xh[NV][V];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int cv = cl + 64 * i;
                if (ron && cv < CV) {
                    Vec<T>::load(g + r * C + cv * V, gv[i]);
                    Vec<T>::load(x + r * C + cv * V, xh[i]);
#pragma unroll
                    for (int j = 0; j < V; ++j) {
                        xh[i][j] = (xh[i][j] - mu) * rstd;
                        const float gh = gv[i][j] * gm[i][j];
                        s1 += gh; s2 = fmaf(gh, xh[i][j], s2);
                        dg[i][j] += gv[i][j] * xh[i][j];
                        db[i][j] += gv[i][j];
                    }
                }
            }
            for (int o = lpr >> 1; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 / (float)C, m2 = s2 / (float)C;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int cv = cl + 64 * i;
                if (ron && cv < CV) {
                    float o[V];
#pragma unroll
                    for (int j = 0; j < V; ++j) o[j] = rstd * (gv[i][j] * gm[i][j] - m1 - xh[i][j] * m2);
                    if (residual) {
                        float rv[V];
                        Vec<T>::load(residual + r * C + cv * V, rv);
#pragma unroll
                        for (int j = 0; j < V; ++j) o[j] = round_to<T>(o[j]) + rv[j];      // the unfused pipeline stored dx, then added
                    }
                    Vec<T>::store(dx + r * C + cv * V, o);
                }
            }
        }
        // the row groups of a wave hold partial sums of the SAME channels: add them in group order (xor lpr, 2 lpr, ...)
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j)
                for (int o = lpr; o < 64; o <<= 1) { dg[i][j] += __shfl_xor(dg[i][j], o); db[i][j] += __shfl_xor(db[i][j], o); }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int cv = cl + 64 * i;
            if (sub == 0 && cv < CV) {
#pragma unroll
                for (int j = 0; j < V; ++j) { mine[cv * V + j] = dg[i][j]; mine[C + cv * V + j] = db[i][j]; }
            }
        }
    } else {
        for (int i = lane; i < 2 * C; i += 64) mine[i] = 0.f;
        for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
            const float mu = stats[2 * r], rstd = stats[2 * r + 1];
            float s1 = 0.f, s2 = 0.f;
            for (int cv = lane; cv < CV; cv += 64) {
                float gv[V], xv[V], gm[V];
                Vec<T>::load(g + r * C + cv * V, gv);
                Vec<T>::load(x + r * C + cv * V, xv);
                load_f32<V>(gamma + cv * V, gm);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (xv[j] - mu) * rstd, gh = gv[j] * gm[j];
                    s1 += gh; s2 = fmaf(gh, xh, s2);
                    mine[cv * V + j] += gv[j] * xh;            // dgamma partial (this lane owns these channels)
                    mine[C + cv * V + j] += gv[j];             // dbeta partial
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
            const float m1 = s1 / (float)C, m2 = s2 / (float)C;
            for (int cv = lane; cv < CV; cv += 64) {
                float gv[V], xv[V], gm[V];
                Vec<T>::load(g + r * C + cv * V, gv);
                Vec<T>::load(x + r * C + cv * V, xv);
                load_f32<V>(gamma + cv * V, gm);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float xh = (xv[j] - mu) * rstd;
                    gv[j] = rstd * (gv[j] * gm[j] - m1 - xh * m2);
                }
                if (residual) {
                    float rv[V];
                    Vec<T>::load(residual + r * C + cv * V, rv);
#pragma unroll
                    for (int j = 0; j < V; ++j) gv[j] = round_to<T>(gv[j]) + rv[j];
                }
                Vec<T>::store(dx + r * C + cv * V, gv);
            }
        }
    }
    __syncthreads();
    float* p = partials + (long)blockIdx.x * 2 * C;
    for (int i = threadIdx.x; i < 2 * C; i += DFD_THREADS) p[i] = sacc[i] + sacc[2 * C + i] + sacc[4 * C + i] + sacc[6 * C + i];
}
#define LN_DISPATCH(KERN, T, ...)                                                                         \
    do {                                                                                                  \
        const int nv_ = (C / Vec<T>::N + 63) / 64;                                                        \
        if (nv_ <= 1) hipLaunchKernelGGL((KERN<T, 1>), __VA_ARGS__);                                      \
        else if (nv_ == 2) hipLaunchKernelGGL((KERN<T, 2>), __VA_ARGS__);                                 \
        else hipLaunchKernelGGL((KERN<T, 0>), __VA_ARGS__);                                               \
    } while (0)
static inline int ln_rows_per_wave(int dtype, int C) {
    const int cv = C / (dtype == DFD_BF16 ? 8 : 4);
    if (cv > 64) return 1;
    int l = 1;
    while (l < cv) l <<= 1;
    return 64 / l;
}
extern "C" int dfd_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, float eps, void* y,
                                 float* stats, long rows, int C, dfd_stream stream) {
    if (!x || !gamma || !beta || !y || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    const int rpw = ln_rows_per_wave(dtype, C);
    long grid = (rows + 4 * rpw - 1) / (4 * rpw);
    if (grid > 8192) grid = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) LN_DISPATCH(k_layernorm_fwd, bf16, dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)x, gamma, beta, eps, (bf16*)y, stats, rows, C);
    else if (dtype == DFD_F32) LN_DISPATCH(k_layernorm_fwd, float, dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)x, gamma, beta, eps, (float*)y, stats, rows, C);
    else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
// partials: [nparts][2][C] with row 0 = dgamma, row 1 = dbeta partial sums; sum them with dfd_sum_rows (or dfd_sum_rows_deferred)
// residual (may be NULL): added to dx — the gradient of the skip connection around the normalised branch
extern "C" int dfd_layernorm_bwd(int dtype, const void* g, const void* x, const float* gamma, const float* stats, const void* residual,
                                 void* dx, float* partials, int pcap, int* nparts, long rows, int C, dfd_stream stream) {
    if (!g || !x || !gamma || !stats || !dx || !partials || !nparts || pcap < 1 || rows < 1 || C < 8 || C % 8) return DFD_EINVAL;
    if ((size_t)8 * C * 4 > 64 * 1024) return DFD_EUNSUPPORTED;
    const int rpw = ln_rows_per_wave(dtype, C);
    long grid = (rows + 16 * rpw - 1) / (16 * rpw);
    if (grid > pcap) grid = pcap;
    if (grid > DFD_MAX_PARTIALS) grid = DFD_MAX_PARTIALS;
    if (grid < 1) grid = 1;
    *nparts = (int)grid;
    const size_t lds = (size_t)8 * C * 4;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) LN_DISPATCH(k_layernorm_bwd, bf16, dim3((unsigned)grid), dim3(DFD_THREADS), lds, st, (const bf16*)g, (const bf16*)x, gamma, stats, (const bf16*)residual, (bf16*)dx, partials, rows, C);
    else if (dtype == DFD_F32) LN_DISPATCH(k_layernorm_bwd, float, dim3((unsigned)grid), dim3(DFD_THREADS), lds, st, (const float*)g, (const float*)x, gamma, stats, (const float*)residual, (float*)dx, partials, rows, C);
    else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
#undef LN_DISPATCH

// ===========================================================================
// token bookkeeping for windowed attention (fastervit faster_vit.py: window_partition / window_reverse,
// ct_dewindow / ct_window, torch.cat of carrier and window tokens and its split): every one of them is a row
// copy  dst[didx[r]] = src[sidx[r]]  over [rows][C] matrices (NULL index array = identity).  When each
// destination row is written once and each source row read at most once — true for all of the above — the
// backward pass is the same kernel with the index arrays swapped.
// ===========================================================================
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_copy_rows(const T* __restrict__ src, const int* __restrict__ sidx, T* __restrict__ dst, const int* __restrict__ didx, long n, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V;
    const long total = n * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const long r = i / CV;
        const int cv = (int)(i - r * CV);
        const long s = sidx ? sidx[r] : r, d = didx ? didx[r] : r;
        *reinterpret_cast<uint4*>(dst + d * C + cv * V) = *reinterpret_cast<const uint4*>(src + s * C + cv * V);
    }
}
extern "C" int dfd_copy_rows(int dtype, const void* src, const int* sidx, void* dst, const int* didx, long n, int C, dfd_stream stream) {
    if (!src || !dst || n < 1 || C < 8 || C % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = (n * (C / 8) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_copy_rows<bf16>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)src, sidx, (bf16*)dst, didx, n, C);
    } else if (dtype == DFD_F32) {
        long grid = (n * (C / 4) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_copy_rows<float>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)src, sidx, (float*)dst, didx, n, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}

// out[r][:] = x[r][:] + table[r % T][:]   (PosEmbMLPSwinv1D: one learned vector per token position, f32 table)
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_add_rowtable(const T* __restrict__ x, const float* __restrict__ table, T* __restrict__ out, long rows, int Tn, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V;
    const long total = rows * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const long r = i / CV;
        const int cv = (int)(i - r * CV);
        float v[V], t[V];
        Vec<T>::load(x + i * V, v);
        load_f32<V>(table + (r % Tn) * C + cv * V, t);
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] += t[j];
        Vec<T>::store(out + i * V, v);
    }
}
extern "C" int dfd_add_rowtable(int dtype, const void* x, const float* table, void* out, long rows, int T, int C, dfd_stream stream) {
    if (!x || !table || !out || rows < 1 || T < 1 || C < 8 || C % 8) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == DFD_BF16) {
        long grid = (rows * (C / 8) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_add_rowtable<bf16>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)x, table, (bf16*)out, rows, T, C);
    } else if (dtype == DFD_F32) {
        long grid = (rows * (C / 4) + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL((k_add_rowtable<float>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)x, table, (float*)out, rows, T, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
// dtable[t][c] = sum_w g[w*T + t][c]   (f32 output).  grid (t, channel chunk, split of the window range): a lane owns one
// 16-byte channel vector and every rpb-th window of its split, four loads in flight; the row lanes are combined in a
// fixed order, the splits by dfd_launch_sum_partials.
#define ROWTAB_MAX_SPLITS 32
template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_rowtable_grad(const T* __restrict__ g, float* __restrict__ parts, long nw, int Tn, int C, ChanMap cm, long w_per) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * V];
    const int tid = threadIdx.x, vl = tid % cm.cvb, rl = tid / cm.cvb;
    const bool active = rl < cm.rpb;
    const int t = blockIdx.x, c0 = (blockIdx.y * cm.cvb + vl) * V;
    const long w_beg = (long)blockIdx.z * w_per;
    long w_end = w_beg + w_per;
    if (w_end > nw) w_end = nw;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (active) {
        const T* base = g + (long)t * C + c0;
        const long step = (long)Tn * C;
        long w = w_beg + rl;
        for (; w + 3 * cm.rpb < w_end; w += 4 * cm.rpb) {
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) Vec<T>::load(base + (w + (long)u * cm.rpb) * step, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) acc[j] += v[u][j];
        }
        for (; w < w_end; w += cm.rpb) {
            float v[V];
            Vec<T>::load(base + w * step, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        }
    }
    reduce_rowlanes<V>(acc, red, cm.cvb, cm.rpb, vl, rl, active);
    if (rl == 0) store_f32<V>(parts + ((long)blockIdx.z * Tn + t) * C + c0, acc);
}
static int rowtable_splits(long nw, int Tn, int C, int vec) {
    const ChanMap cm = make_chanmap(C, vec);
    long s = 2048 / ((long)Tn * cm.nvc > 0 ? (long)Tn * cm.nvc : 1);
    const long max_s = nw / ((long)cm.rpb * 4);
    if (s > max_s) s = max_s;
    if (s > ROWTAB_MAX_SPLITS) s = ROWTAB_MAX_SPLITS;
    return s < 1 ? 1 : (int)s;
}
extern "C" size_t dfd_rowtable_grad_ws(int T, int C) { return (size_t)(ROWTAB_MAX_SPLITS + 2) * T * C * sizeof(float); }
extern "C" int dfd_rowtable_grad(int dtype, const void* g, float* dtable, long rows, int T, int C, int accumulate, float* ws,
                                 size_t ws_bytes, dfd_stream stream) {
    if (!g || !dtable || !ws || rows < 1 || T < 1 || rows % T || C < 1 || (dtype != DFD_BF16 && dtype != DFD_F32)) return DFD_EINVAL;
    const int vec = dtype == DFD_BF16 ? Vec<bf16>::N : Vec<float>::N;
    if (C % vec) return DFD_EINVAL;
    if (ws_bytes < dfd_rowtable_grad_ws(T, C)) return DFD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long nw = rows / T;
    const ChanMap cm = make_chanmap(C, vec);
    const int splits = rowtable_splits(nw, T, C, vec);
    const long w_per = (nw + splits - 1) / splits;
    const dim3 grid(T, cm.nvc, splits);
    if (dtype == DFD_BF16) hipLaunchKernelGGL((k_rowtable_grad<bf16>), grid, dim3(DFD_THREADS), 0, st, (const bf16*)g, ws, nw, T, C, cm, w_per);
    else hipLaunchKernelGGL((k_rowtable_grad<float>), grid, dim3(DFD_THREADS), 0, st, (const float*)g, ws, nw, T, C, cm, w_per);
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, splits, (long)T * C, dtable, accumulate, st, false);   // consumed by the table's MLP backward
}

// average pooling k x k, stride s, no padding (TokenInitializer's AvgPool2d(5, 3)) and its gradient
template <typename T, bool BWD>
__global__ void __launch_bounds__(DFD_THREADS)
k_avgpool(const T* __restrict__ in, T* __restrict__ out, int N, int H, int W, int Ho, int Wo, int k, int s, int C) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V;
    const float inv = 1.0f / (float)(k * k);
    const long total = BWD ? (long)N * H * W * CV : (long)N * Ho * Wo * CV;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total; i += (long)gridDim.x * DFD_THREADS) {
        const int cv = (int)(i % CV);
        long tq = i / CV;
        float acc[V];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] = 0.f;
        if constexpr (!BWD) {
            const int ox = (int)(tq % Wo); tq /= Wo;
            const int oy = (int)(tq % Ho);
            const long n = tq / Ho;
            for (int dy = 0; dy < k; ++dy)
                for (int dx = 0; dx < k; ++dx) {
                    float v[V];
                    Vec<T>::load(in + ((n * H + oy * s + dy) * (long)W + ox * s + dx) * C + cv * V, v);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += v[j];
                }
        } else {
            const int x = (int)(tq % W); tq /= W;
            const int y = (int)(tq % H);
            const long n = tq / H;
            for (int oy = 0; oy < Ho; ++oy) {
                if (y < oy * s || y >= oy * s + k) continue;
                for (int ox = 0; ox < Wo; ++ox) {
                    if (x < ox * s || x >= ox * s + k) continue;
                    float v[V];
                    Vec<T>::load(in + ((n * Ho + oy) * (long)Wo + ox) * C + cv * V, v);
#pragma unroll
                    for (int j = 0; j < V; ++j) acc[j] += v[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] *= inv;
        Vec<T>::store(out + i * V, acc);
    }
}
static int avgpool_go(int dtype, bool bwd, const void* in, void* out, int N, int H, int W, int k, int s, int C, hipStream_t st) {
    if (!in || !out || N < 1 || H < k || W < k || k < 1 || s < 1 || C < 8 || C % 8) return DFD_EINVAL;
    const int Ho = (H - k) / s + 1, Wo = (W - k) / s + 1;
    const long items = (bwd ? (long)N * H * W : (long)N * Ho * Wo);
    if (dtype == DFD_BF16) {
        long grid = (items * (C / 8) + 255) / 256; if (grid > 16384) grid = 16384;
        if (bwd) hipLaunchKernelGGL((k_avgpool<bf16, true>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)in, (bf16*)out, N, H, W, Ho, Wo, k, s, C);
        else hipLaunchKernelGGL((k_avgpool<bf16, false>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const bf16*)in, (bf16*)out, N, H, W, Ho, Wo, k, s, C);
    } else if (dtype == DFD_F32) {
        long grid = (items * (C / 4) + 255) / 256; if (grid > 16384) grid = 16384;
        if (bwd) hipLaunchKernelGGL((k_avgpool<float, true>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)in, (float*)out, N, H, W, Ho, Wo, k, s, C);
        else hipLaunchKernelGGL((k_avgpool<float, false>), dim3((unsigned)grid), dim3(DFD_THREADS), 0, st, (const float*)in, (float*)out, N, H, W, Ho, Wo, k, s, C);
    } else return DFD_EINVAL;
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_avgpool_fwd(int dtype, const void* x, void* out, int N, int H, int W, int k, int stride, int C, dfd_stream stream) {
    return avgpool_go(dtype, false, x, out, N, H, W, k, stride, C, (hipStream_t)stream);
}
extern "C" int dfd_avgpool_bwd(int dtype, const void* g, void* dx, int N, int H, int W, int k, int stride, int C, dfd_stream stream) {
    return avgpool_go(dtype, true, g, dx, N, H, W, k, stride, C, (hipStream_t)stream);
}

// relative-position attention bias of PosEmbMLPSwinv2D:
//   full[h][ng + i][ng + j] = 16 * sigmoid(table[idx[i*nl + j]][h]),  zero in the first ng rows / columns (carrier tokens)
__global__ void k_relpos_bias_fwd(const float* __restrict__ table, const int* __restrict__ idx, float* __restrict__ full, int H, int nl, int ng) {
    const int S = nl + ng;
    const long total = (long)H * S * S;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % S), i = (int)((e / S) % S), h = (int)(e / ((long)S * S));
    float v = 0.f;
    if (i >= ng && j >= ng) v = 16.0f * sigmoid_f(table[(long)idx[(i - ng) * nl + (j - ng)] * H + h]);
    full[e] = v;
}
// dtable[t][h] = sum_{(i,j): idx == t} dfull[h][ng+i][ng+j] * 16 * s * (1 - s),  s = sigmoid(table[t][h]); one wave per (t, h)
__global__ void k_relpos_bias_bwd(const float* __restrict__ dfull, const float* __restrict__ table, const int* __restrict__ idx,
                                  float* __restrict__ dtable, int H, int Tn, int nl, int ng) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= Tn * H) return;
    const int t = wave / H, h = wave - t * H;
    const int S = nl + ng;
    float s = 0.f;
    for (int l = lane; l < nl * nl; l += 64) {
        if (idx[l] != t) continue;
        const int i = l / nl, j = l - i * nl;
        s += dfull[((long)h * S + ng + i) * S + ng + j];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float sg = sigmoid_f(table[(long)t * H + h]);
        dtable[(long)t * H + h] = s * 16.0f * sg * (1.0f - sg);
    }
}
extern "C" int dfd_relpos_bias_fwd(const float* table, const int* idx, float* full, int H, int n_local, int n_global, dfd_stream stream) {
    if (!table || !idx || !full || H < 1 || n_local < 1 || n_global < 0) return DFD_EINVAL;
    const long S = n_local + n_global, total = (long)H * S * S;
    hipLaunchKernelGGL(k_relpos_bias_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, idx, full, H, n_local, n_global);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_relpos_bias_bwd(const float* dfull, const float* table, const int* idx, float* dtable, int H, int T, int n_local,
                                   int n_global, dfd_stream stream) {
    if (!dfull || !table || !idx || !dtable || H < 1 || T < 1 || n_local < 1 || n_global < 0) return DFD_EINVAL;
    const long waves = (long)T * H;
    hipLaunchKernelGGL(k_relpos_bias_bwd, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dfull, table, idx, dtable, H, T, n_local, n_global);
    return DFD_CHECK_LAUNCH();
}
