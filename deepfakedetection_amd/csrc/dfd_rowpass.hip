// dfd_rowpass.hip — HBM-streaming passes over the NHWC row matrix [N*HW][C]:
// BN apply / BN backward reductions / activation backward / SE squeeze.
// Every kernel moves 16 B per lane; consecutive lanes read consecutive channel
// vectors of a row, so a wave reads whole contiguous row segments.
// Per-channel reductions keep the channel assignment of a thread fixed while it
// strides over rows, accumulate in registers, reduce once through LDS and write ONE
// partial row per workgroup (fixed summation order => reproducible results).
#include "dfd_common.h"
#include "dfd_se.h"
#include "dfd_sum.h"

// ------------------------------------------------------------------ bn_act_apply
template <typename T, int ACT, bool HAS_RES, bool HAS_RS>
__global__ void __launch_bounds__(DFD_THREADS)
k_bn_act_apply(const T* __restrict__ y, const float* __restrict__ bnstate, const T* __restrict__ res,
               const float* __restrict__ rs, T* __restrict__ out, long total_vec, int CV, int C, long vec_per_img) {
    constexpr int V = Vec<T>::N;
    const long stride = (long)gridDim.x * DFD_THREADS;
    const long first = (long)blockIdx.x * DFD_THREADS + threadIdx.x;
    int cv = (int)(first % CV);                       // channel vector of element i, kept incrementally
    const int cstep = (int)(stride % CV);
    for (long i = first; i < total_vec; i += stride, cv = cv + cstep >= CV ? cv + cstep - CV : cv + cstep) {
        const int c0 = cv * V;
        float v[V], sc[V], sh[V];
        Vec<T>::load(y + i * V, v);
        load_f32<V>(bnstate + c0, sc);
        load_f32<V>(bnstate + C + c0, sh);
        float r = 1.f;
        if constexpr (HAS_RS) r = rs[i / vec_per_img];
        float q[V];
        if constexpr (HAS_RES) Vec<T>::load(res + i * V, q);
        bn_act_array<ACT, V>(v, sc, sh);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float z = v[j];
            if constexpr (HAS_RS) z *= r;
            if constexpr (HAS_RES) z += q[j];
            v[j] = z;
        }
        Vec<T>::store(out + i * V, v);
    }
}

template <typename T>
__global__ void __launch_bounds__(DFD_THREADS)
k_scale_rows(const T* __restrict__ x, const float* __restrict__ rs, T* __restrict__ out, long total_vec, long vec_per_img) {
    constexpr int V = Vec<T>::N;
    const long stride = (long)gridDim.x * DFD_THREADS;
    for (long i = (long)blockIdx.x * DFD_THREADS + threadIdx.x; i < total_vec; i += stride) {
        float v[V];
        Vec<T>::load(x + i * V, v);
        const float r = rs[i / vec_per_img];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] *= r;
        Vec<T>::store(out + i * V, v);
    }
}

// ------------------------------------------------------------------ bn_bwd_reduce
// partials[pb][0][c] = sum g*rs ; partials[pb][1][c] = sum g*rs*xhat
template <typename T, bool HAS_RS, bool HAS_Y = true>      // HAS_Y false: only sum g (second sum 0): layers without statistics
__global__ void __launch_bounds__(DFD_THREADS)
k_bn_bwd_reduce(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ bnstate,
                const float* __restrict__ rs, long rows, int HW, int C, ChanMap cm, float* __restrict__ partials) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * 2 * V];
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int c0 = (blockIdx.y * cm.cvb + vl) * V;
    float mean[V], rstd[V];
    if constexpr (HAS_Y) {
        load_f32<V>(bnstate + 2 * C + c0, mean);
        load_f32<V>(bnstate + 3 * C + c0, rstd);
    }
    float acc[2 * V];
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) acc[j] = 0.f;
    if (active) {
        const long step = (long)gridDim.x * cm.rpb;
        long r = (long)blockIdx.x * cm.rpb + rl;
        // four rows (eight vectors) in flight per lane
        for (; r + 3 * step < rows; r += 4 * step) {
            float gv[4][V], yv[4][V], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Vec<T>::load(g + (r + u * step) * C + c0, gv[u]);
                if constexpr (HAS_Y) Vec<T>::load(y + (r + u * step) * C + c0, yv[u]);
                sv[u] = 1.f;
                if constexpr (HAS_RS) sv[u] = rs[(r + u * step) / HW];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    float gg = gv[u][j];
                    if constexpr (HAS_RS) gg *= sv[u];
                    acc[j] += gg;
                    if constexpr (HAS_Y) acc[V + j] += gg * (yv[u][j] - mean[j]) * rstd[j];
                }
        }
        for (; r < rows; r += step) {
            float gv[V], yv[V];
            Vec<T>::load(g + r * C + c0, gv);
            if constexpr (HAS_Y) Vec<T>::load(y + r * C + c0, yv);
            float s = 1.f;
            if constexpr (HAS_RS) s = rs[r / HW];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float gg = gv[j];
                if constexpr (HAS_RS) gg *= s;
                acc[j] += gg;
                if constexpr (HAS_Y) acc[V + j] += gg * (yv[j] - mean[j]) * rstd[j];
            }
        }
    }
    reduce_rowlanes<2 * V>(acc, red, cm.cvb, cm.rpb, vl, rl, active);
    if (rl == 0) {
        float* p = partials + (long)blockIdx.x * (HAS_Y ? 2 : 1) * C;      // HAS_Y false: rows of C sums (dfd_bias_grad)
        float a0[V], a1[V];
#pragma unroll
        for (int j = 0; j < V; ++j) { a0[j] = acc[j]; a1[j] = acc[V + j]; }
        store_f32<V>(p + c0, a0);
        if constexpr (HAS_Y) store_f32<V>(p + C + c0, a1);
    }
}

// ------------------------------------------------------------------ act_bn_bwd
// MODE 0: da = D ; MODE 1: da = D*gate[n,c] + dpool[n,c]*invHW ; MODE 2: da = dpool[n,c]*invHW
struct SeWgradJob { const float* pooled; const float* ws; int N, C, R; float* dw1; float* db1; float* dw2; float* db2; int accumulate; };

template <typename T, int ACT, int MODE>
__global__ void __launch_bounds__(DFD_THREADS)
k_act_bn_bwd(const T* __restrict__ D, const T* __restrict__ y, const float* __restrict__ gate,
             const float* __restrict__ dpool, const float* __restrict__ bnstate, T* __restrict__ dz,
             long rows, int HW, int C, float invHW, ChanMap cm, float* __restrict__ partials, int main_x, SeWgradJob job,
             SumJobs sum1, SumJobs sum2) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * 2 * V];
    if ((int)blockIdx.x >= main_x) {
        // passenger workgroups: work that only the optimizer reads, carried here instead of being ~6-13 us launches of its own on the
        // dependency chain — the squeeze-excite FC weight gradients of this block (dfd_act_bn_bwd_se), stage 1 / stage 2 of the
        // batched weight-gradient slab sums handed over by earlier blocks (dfd_sum_batch_end_deferred)
        int id = ((int)blockIdx.x - main_x) * (int)gridDim.y + (int)blockIdx.y;
        const int cb = (job.C + 63) / 64, nse = job.dw1 ? cb * job.R : 0;
        if (id < nse) {
            se_fc_bwd_w_body(job.pooled, job.ws, job.N, job.C, job.R, job.dw1, job.db1, job.dw2, job.db2, job.accumulate, id % cb, id / cb, red);
            return;
        }
        id -= nse;
        const int n1 = sum1.n > 0 ? sum1.gx * sum1.gofs[sum1.n] : 0;
        if (id < n1) { sum_multi_body(sum1, 1, id % sum1.gx, id / sum1.gx); return; }
        id -= n1;
        if (sum2.n > 0 && id < sum2.gx * sum2.n) sum_multi_body(sum2, 2, id % sum2.gx, id / sum2.gx);
        return;
    }
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int c0 = (blockIdx.y * cm.cvb + vl) * V;
    float sc[V], sh[V], mean[V], rstd[V];
    load_f32<V>(bnstate + c0, sc);
    load_f32<V>(bnstate + C + c0, sh);
    load_f32<V>(bnstate + 2 * C + c0, mean);
    load_f32<V>(bnstate + 3 * C + c0, rstd);
    float acc[2 * V];
#pragma unroll
    for (int j = 0; j < 2 * V; ++j) acc[j] = 0.f;
    if (active) {
        const long step = (long)main_x * cm.rpb;
        for (long r = (long)blockIdx.x * cm.rpb + rl; r < rows; r += step) {
            float dv[V], yv[V], gt[V], dp[V];
            Vec<T>::load(y + r * C + c0, yv);
            if constexpr (MODE != 2) Vec<T>::load(D + r * C + c0, dv);
            if constexpr (MODE != 0) {
                const long n = r / HW;
                load_f32<V>(dpool + n * C + c0, dp);
                if constexpr (MODE == 1) load_f32<V>(gate + n * C + c0, gt);
            }
            float agr[V];                                      // act'(scale * y + shift), evaluated in pairs
#pragma unroll
            for (int j = 0; j < V; j += 2) {
                const dfd_f2 z2 = __builtin_elementwise_fma((dfd_f2){sc[j], sc[j + 1]}, (dfd_f2){yv[j], yv[j + 1]}, (dfd_f2){sh[j], sh[j + 1]});
                const dfd_f2 g2 = act_grad2<ACT>(z2);
                agr[j] = g2.x; agr[j + 1] = g2.y;
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float da;
                if constexpr (MODE == 0) da = dv[j];
                else if constexpr (MODE == 1) da = fmaf(dv[j], gt[j], dp[j] * invHW);
                else da = dp[j] * invHW;
                const float d = round_to<T>(da * agr[j]);
                dv[j] = d;
                acc[j] += d;
                acc[V + j] += d * (yv[j] - mean[j]) * rstd[j];
            }
            Vec<T>::store(dz + r * C + c0, dv);
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

// ------------------------------------------------------------------ pool kernels
// grid = (nvc, N, splits): a workgroup owns the rows [z*rows_per .. ) of image n for its channel
// chunk.  Large images are split over several workgroups (one per (chunk, image) leaves most of
// the chip idle and each lane with two loads in flight); k_pool_sum then adds the partial vectors
// in split order.  (A last-arrival reduction inside this kernel was measured 5x slower: the
// device-scope fences it needs write back / invalidate L2 on this multi-XCD part.)
// BWD == false: pooled[n,c] = (1/HW) sum act(scale*y+shift)
// BWD == true : dgate[n,c]  = sum D * act(scale*y+shift)
template <typename T, int ACT, bool BWD>
__global__ void __launch_bounds__(DFD_THREADS)
k_pool(const T* __restrict__ D, const T* __restrict__ y, const float* __restrict__ bnstate,
       float* __restrict__ out, int HW, int C, float mul, ChanMap cm, int rows_per, float* __restrict__ parts) {
    constexpr int V = Vec<T>::N;
    __shared__ float red[DFD_THREADS * V];
    const int t = threadIdx.x, vl = t % cm.cvb, rl = t / cm.cvb;
    const bool active = rl < cm.rpb;
    const int c0 = (blockIdx.x * cm.cvb + vl) * V;
    const long n = blockIdx.y;
    const int splits = gridDim.z, N = gridDim.y;
    const int r_beg = blockIdx.z * rows_per;
    int r_end = r_beg + rows_per;
    if (r_end > HW) r_end = HW;
    float sc[V], sh[V];
    load_f32<V>(bnstate + c0, sc);
    load_f32<V>(bnstate + C + c0, sh);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (active) {
        const T* yb = y + n * HW * (long)C + c0;
        const T* db = BWD ? D + n * HW * (long)C + c0 : nullptr;
        int r = r_beg + rl;
        // four rows in flight per lane
        for (; r + 3 * cm.rpb < r_end; r += 4 * cm.rpb) {
            float yv[4][V], dv[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Vec<T>::load(yb + (long)(r + u * cm.rpb) * C, yv[u]);
                if constexpr (BWD) Vec<T>::load(db + (long)(r + u * cm.rpb) * C, dv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const float a0 = round_to<T>(act_fwd<ACT>(fmaf(sc[j], yv[u][j], sh[j])));
                    if constexpr (BWD) acc[j] += a0 * dv[u][j];
                    else acc[j] += a0;
                }
        }
        for (; r < r_end; r += cm.rpb) {
            float y0[V], d0[V];
            Vec<T>::load(yb + (long)r * C, y0);
            if constexpr (BWD) Vec<T>::load(db + (long)r * C, d0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float a0 = round_to<T>(act_fwd<ACT>(fmaf(sc[j], y0[j], sh[j])));
                if constexpr (BWD) acc[j] += a0 * d0[j];
                else acc[j] += a0;
            }
        }
    }
    reduce_rowlanes<V>(acc, red, cm.cvb, cm.rpb, vl, rl, active);
    if (splits == 1) {
        if (rl == 0) {
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] *= mul;
            store_f32<V>(out + n * C + c0, acc);
        }
        return;
    }
    if (rl == 0) store_f32<V>(parts + ((long)blockIdx.z * N + n) * C + c0, acc);
}
__global__ void k_pool_sum(const float* __restrict__ parts, int splits, long NC, float mul, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += parts[(long)z * NC + i];
    out[i] = s * mul;
}

// ------------------------------------------------------------------ finalize kernels
// workgroup = 8 channels x 32 partial-row lanes; each lane sums every 32nd partial row in
// f32, the 32 lane sums are combined in a fixed order in double.
#define FIN_CH 8
#define FIN_LANES 32
__device__ __forceinline__ void fin_reduce(const float* __restrict__ partials, int nparts, int C, int c, int q,
                                           double* sm, double& s0, double& s1) {
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        // eight rows per trip, loaded before any of them is added: the loop is a chain of
        // memory round trips, not of additions (order of the additions is fixed)
        int p = q;
        for (; p + 7 * FIN_LANES < nparts; p += 8 * FIN_LANES) {
            float u[8], v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float* row = partials + (long)(p + i * FIN_LANES) * 2 * C;
                u[i] = row[c];
                v[i] = row[C + c];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { a0 += u[i]; a1 += v[i]; }
        }
        for (; p < nparts; p += FIN_LANES) {
            a0 += partials[(long)p * 2 * C + c];
            a1 += partials[(long)p * 2 * C + C + c];
        }
    }
    // the 32 lane sums of a channel: butterfly over the eight lanes of each wave (lanes FIN_CH apart), then the four
    // wave totals in wave order — a fixed tree in double instead of 64 dependent LDS reads on the q == 0 lanes
    const int cl = threadIdx.x & (FIN_CH - 1), wave = threadIdx.x >> 6;
    double v0 = (double)a0, v1 = (double)a1;
#pragma unroll
    for (int o = FIN_CH; o < 64; o <<= 1) {
        v0 += __shfl_xor(v0, o);
        v1 += __shfl_xor(v1, o);
    }
    if ((threadIdx.x & 63) < FIN_CH) {
        sm[(wave * FIN_CH + cl) * 2] = v0;
        sm[(wave * FIN_CH + cl) * 2 + 1] = v1;
    }
    __syncthreads();
    s0 = 0.0; s1 = 0.0;
    if (q == 0) {
#pragma unroll
        for (int i = 0; i < FIN_CH * FIN_LANES / 64; ++i) { s0 += sm[(i * FIN_CH + cl) * 2]; s1 += sm[(i * FIN_CH + cl) * 2 + 1]; }
    }
}

__global__ void __launch_bounds__(FIN_CH * FIN_LANES)
k_bn_finalize(const float* __restrict__ partials, int nparts, int C, double count,
              const float* __restrict__ gamma, const float* __restrict__ beta,
              const float* __restrict__ conv_bias, const float* __restrict__ ls,
              float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
              float* __restrict__ bnstate) {
    __shared__ double sm[FIN_CH * FIN_LANES * 2];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
    double s, ss;
    // the per-channel parameters are requested before the reduction, not after it: these kernels are a chain of
    // memory round trips (~5 us each, ~100 of them per step), and this takes one link out
    const bool mine = c < C && q == 0;
    const float g = (mine && gamma) ? gamma[c] : 1.f, b = (mine && beta) ? beta[c] : 0.f;
    const float l = (mine && ls) ? ls[c] : 1.f;             // LayerScale folded into the affine map
    const float cb = (mine && rmean && conv_bias) ? conv_bias[c] : 0.f;
    const float rm0 = (mine && rmean) ? rmean[c] : 0.f, rv0 = (mine && rmean) ? rvar[c] : 0.f;
    fin_reduce(partials, nparts, C, c, q, sm, s, ss);
    if (mine) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float scale = g * rstd;
        bnstate[c] = l * scale;
        bnstate[C + c] = l * (b - (float)mean * scale);
        bnstate[2 * C + c] = (float)mean;
        bnstate[3 * C + c] = rstd;
        if (rmean) {
            // the convolution's bias never enters the kernels: a constant per channel cancels in
            // batch-statistics BatchNorm and only shifts the running mean
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            const float mb = (float)mean + cb;
            rmean[c] = (1.f - momentum) * rm0 + momentum * mb;
            rvar[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
        }
    }
}

__global__ void k_bn_eval_coeffs(const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ conv_bias, const float* __restrict__ ls,
                                 const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                 float* __restrict__ bnstate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.0f / sqrtf(rvar[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float m = rmean[c] - (conv_bias ? conv_bias[c] : 0.f);      // BN(y + bias) == BN'(y) with mean - bias
    const float l = ls ? ls[c] : 1.f;
    const float scale = g * rstd;
    bnstate[c] = l * scale;
    bnstate[C + c] = l * (b - m * scale);
    bnstate[2 * C + c] = m;
    bnstate[3 * C + c] = rstd;
}

__global__ void __launch_bounds__(FIN_CH * FIN_LANES)
k_bn_bwd_finalize(const float* __restrict__ partials, int nparts, int C, double count,
                  const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ ls,
                  const float* __restrict__ bnstate, int train,
                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dls, float* __restrict__ dbias,
                  int accumulate, float* __restrict__ coef) {
    __shared__ double sm[FIN_CH * FIN_LANES * 2];
    const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
    double s1, s2;
    const bool mine = c < C && q == 0;                       // parameters first, reduction second (see k_bn_finalize)
    const double l = (mine && ls) ? (double)ls[c] : 1.0;
    const double g0 = (mine && gamma) ? (double)gamma[c] : 1.0;
    const double b0 = (mine && beta) ? (double)beta[c] : 0.0;
    const double mean = mine ? (double)bnstate[2 * C + c] : 0.0, rstd = mine ? (double)bnstate[3 * C + c] : 1.0;
    const float og = (mine && accumulate && dgamma) ? dgamma[c] : 0.f, ob = (mine && accumulate && dbeta) ? dbeta[c] : 0.f;
    const float ol = (mine && accumulate && dls) ? dls[c] : 0.f, obi = (mine && accumulate && dbias) ? dbias[c] : 0.f;
    fin_reduce(partials, nparts, C, c, q, sm, s1, s2);
    if (mine) {
        // s1 = sum dz, s2 = sum dz*xhat with dz = d loss / d (ls * BN(y)); BN(y) = gamma*xhat + beta
        if (dgamma) dgamma[c] = og + (float)(l * s2);
        if (dbeta) dbeta[c] = ob + (float)(l * s1);
        if (dls) dls[c] = ol + (float)(g0 * s2 + b0 * s1);
        const double g = g0 * l;
        const double a = g * rstd;
        // the producing convolution's bias: d/d bias = sum dy, exactly 0 through batch statistics
        if (dbias) dbias[c] = obi + (train ? 0.f : (float)(a * s1));
        double b = 0.0, cc = 0.0;
        if (train) {
            b = -g * rstd * rstd * s2 / count;
            cc = -g * rstd * s1 / count + g * rstd * rstd * mean * s2 / count;
        }
        coef[c] = (float)a;
        coef[C + c] = (float)b;
        coef[2 * C + c] = (float)cc;
    }
}

// ------------------------------------------------------------------ host side
static inline int grid_for(long items, int per_block, int cap) {
    long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
static inline bool shape_ok(int dtype, int N, int HW, int C) {
    if (dtype != DFD_F32 && dtype != DFD_BF16) return false;
    if (N <= 0 || HW <= 0 || C <= 0) return false;
    return C % 8 == 0;
}


template <typename T>
static int bn_act_apply_t(const void* y, const float* bnstate, int act, const void* residual, const float* rs,
                          void* out, int N, int HW, int C, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    const int CV = C / V;
    const long total = (long)N * HW * CV, vpi = (long)HW * CV;
    const int grid = grid_for(total, DFD_THREADS * 4, 16384);
#define LAUNCH_APPLY(RES, RS) \
    hipLaunchKernelGGL((k_bn_act_apply<T, ACT, RES, RS>), dim3(grid), dim3(DFD_THREADS), 0, st, (const T*)y, bnstate, \
                       (const T*)residual, rs, (T*)out, total, CV, C, vpi)
    DISPATCH_ACT(act, {
        if (residual && rs) LAUNCH_APPLY(true, true);
        else if (residual) LAUNCH_APPLY(true, false);
        else if (rs) LAUNCH_APPLY(false, true);
        else LAUNCH_APPLY(false, false);
    });
#undef LAUNCH_APPLY
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_bn_act_apply(int dtype, const void* y, const float* bnstate, int act, const void* residual,
                                const float* row_scale, void* out, int N, int HW, int C, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !y || !bnstate || !out) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16 ? bn_act_apply_t<bf16>(y, bnstate, act, residual, row_scale, out, N, HW, C, st)
                             : bn_act_apply_t<float>(y, bnstate, act, residual, row_scale, out, N, HW, C, st);
}

template <typename T>
static int scale_rows_t(const void* x, const float* rs, void* out, int N, int HW, int C, hipStream_t st) {
    constexpr int V = Vec<T>::N;
    const long total = (long)N * HW * (C / V), vpi = (long)HW * (C / V);
    const int grid = grid_for(total, DFD_THREADS * 4, 16384);
    hipLaunchKernelGGL((k_scale_rows<T>), dim3(grid), dim3(DFD_THREADS), 0, st, (const T*)x, rs, (T*)out, total, vpi);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_scale_rows(int dtype, const void* x, const float* row_scale, void* out, int N, int HW, int C,
                              dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !x || !row_scale || !out) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16 ? scale_rows_t<bf16>(x, row_scale, out, N, HW, C, st)
                             : scale_rows_t<float>(x, row_scale, out, N, HW, C, st);
}

static inline int pick_parts(long rows, int rpb, int pcap) {
    int cap = pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS;
    // at least ~4 row iterations per workgroup, at most `cap` workgroups
    long want = (rows + (long)rpb * 4 - 1) / ((long)rpb * 4);
    if (want < 1) want = 1;
    if (want > cap) want = cap;
    return (int)want;
}

template <typename T>
static int bn_bwd_reduce_t(const void* g, const void* y, const float* bnstate, const float* rs, int N, int HW, int C,
                           float* partials, int pcap, int* nparts, hipStream_t st) {
    const ChanMap cm = make_chanmap(C, Vec<T>::N);
    const long rows = (long)N * HW;
    const int P = pick_parts(rows, cm.rpb, pcap);
    *nparts = P;
    dim3 grid(P, cm.nvc);
#define LAUNCH_BBR(RS, HY) hipLaunchKernelGGL((k_bn_bwd_reduce<T, RS, HY>), grid, dim3(DFD_THREADS), 0, st, (const T*)g, (const T*)y, bnstate, rs, \
                                             rows, HW, C, cm, partials)
    if (rs) { if (y) LAUNCH_BBR(true, true); else LAUNCH_BBR(true, false); }
    else { if (y) LAUNCH_BBR(false, true); else LAUNCH_BBR(false, false); }
#undef LAUNCH_BBR
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_bwd_reduce(int dtype, const void* g, const void* y, const float* bnstate, const float* row_scale,
                                 int N, int HW, int C, float* partials, int pcap, int* nparts, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !g || !y || !bnstate || !partials || !nparts || pcap < 1) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16 ? bn_bwd_reduce_t<bf16>(g, y, bnstate, row_scale, N, HW, C, partials, pcap, nparts, st)
                             : bn_bwd_reduce_t<float>(g, y, bnstate, row_scale, N, HW, C, partials, pcap, nparts, st);
}
// Bias gradient of a layer without statistics (Linear + bias): dbias[c] (+)= sum over rows of g[row][c] (* row_scale[n]).
// One reduction launch into partial rows [P][C] in `ws`, summed in row order by the weight gradients' summation (inside
// dfd_sum_batch_begin/_end: with the block's other sums) — the BatchNorm-shaped pair it replaces (dfd_bn_bwd_reduce reading
// y to multiply it by zero + a finalize launch per layer) cost two dependent launches where the second one was ~7 us of
// pure latency (68 of them per FasterViT-0 step).
extern "C" size_t dfd_bias_grad_ws(int N, int HW, int C) {
    if (N < 1 || HW < 1 || C < 1) return 0;
    return (size_t)(DFD_MAX_PARTIALS + DFD_MAX_PARTIALS / 32 + 2) * C * 4;
}
extern "C" int dfd_bias_grad(int dtype, const void* g, const float* row_scale, int N, int HW, int C, float* dbias, int accumulate,
                             float* ws, size_t ws_bytes, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !g || !dbias || !ws) return DFD_EINVAL;
    if (ws_bytes < dfd_bias_grad_ws(N, HW, C)) return DFD_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int P = 0;
    const int rc = dtype == DFD_BF16 ? bn_bwd_reduce_t<bf16>(g, nullptr, nullptr, row_scale, N, HW, C, ws, DFD_MAX_PARTIALS, &P, st)
                                     : bn_bwd_reduce_t<float>(g, nullptr, nullptr, row_scale, N, HW, C, ws, DFD_MAX_PARTIALS, &P, st);
    if (rc != DFD_OK) return rc;
    return dfd_launch_sum_partials(ws, P, (long)C, dbias, accumulate, st);
}

template <typename T>
static int act_bn_bwd_t(const void* D, const void* y, const float* gate, const float* dpool, const float* bnstate,
                        int act, void* dz, int N, int HW, int C, float* partials, int pcap, int* nparts,
                        hipStream_t st, const SeWgradJob* se = nullptr) {
    const ChanMap cm = make_chanmap(C, Vec<T>::N);
    const long rows = (long)N * HW;
    const int P = pick_parts(rows, cm.rpb, pcap);
    *nparts = P;
    SeWgradJob job{nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
    SumJobs sum1, sum2;
    dfd_sum_passengers_take(st, &sum1, &sum2);           // slab sums waiting for a launch on this stream (usually none)
    long passengers = (sum1.n > 0 ? (long)sum1.gx * sum1.gofs[sum1.n] : 0) + (sum2.n > 0 ? (long)sum2.gx * sum2.n : 0);
    if (se) {                                            // passenger workgroups behind the P x nvc main ones
        job = *se;
        passengers += (long)((job.C + 63) / 64) * job.R;
    }
    const int extra = (int)((passengers + cm.nvc - 1) / cm.nvc);
    dim3 grid(P + extra, cm.nvc);
    const float invHW = 1.0f / (float)HW;
    int mode;
    if (D && gate && dpool) mode = 1;
    else if (D && !gate && !dpool) mode = 0;
    else if (!D && !gate && dpool) mode = 2;
    else return DFD_EINVAL;
#define LAUNCH_ABB(MODE) \
    hipLaunchKernelGGL((k_act_bn_bwd<T, ACT, MODE>), grid, dim3(DFD_THREADS), 0, st, (const T*)D, (const T*)y, gate, dpool, \
                       bnstate, (T*)dz, rows, HW, C, invHW, cm, partials, P, job, sum1, sum2)
    DISPATCH_ACT(act, {
        if (mode == 0) LAUNCH_ABB(0);
        else if (mode == 1) LAUNCH_ABB(1);
        else LAUNCH_ABB(2);
    });
#undef LAUNCH_ABB
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_act_bn_bwd(int dtype, const void* D, const void* y, const float* gate, const float* dpool,
                              const float* bnstate, int act, void* dz, int N, int HW, int C, float* partials, int pcap,
                              int* nparts, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !y || !bnstate || !dz || !partials || !nparts || pcap < 1) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16
               ? act_bn_bwd_t<bf16>(D, y, gate, dpool, bnstate, act, dz, N, HW, C, partials, pcap, nparts, st)
               : act_bn_bwd_t<float>(D, y, gate, dpool, bnstate, act, dz, N, HW, C, partials, pcap, nparts, st);
}

// HW rows per split so that the launch has about 2048 workgroups; one split for small images
static void pool_plan(int N, int HW, int C, int vec, int* splits, int* rows_per) {
    const ChanMap cm = make_chanmap(C, vec);
    long base = (long)cm.nvc * N;
    int s = (int)(2048 / (base > 0 ? base : 1));
    const int max_s = HW / (cm.rpb * 8);                // at least 8 rows per lane
    if (s > max_s) s = max_s;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    int rp = (HW + s - 1) / s;
    s = (HW + rp - 1) / rp;
    *splits = s;
    *rows_per = rp;
}
static size_t pool_ws_bytes(int N, int HW, int C, int vec) {
    int s, rp;
    pool_plan(N, HW, C, vec, &s, &rp);
    if (s <= 1) return 0;
    return (size_t)s * N * C * sizeof(float);
}
// dfd_act_bn_bwd with the block's squeeze-excite FC weight gradients riding along (include/dfd_hip.h)
extern "C" int dfd_act_bn_bwd_se(int dtype, const void* D, const void* y, const float* gate, const float* dpool,
                                 const float* bnstate, int act, void* dz, int N, int HW, int C, float* partials, int pcap,
                                 int* nparts, const float* se_pooled, const float* se_ws, int R, float* dw1, float* db1, float* dw2,
                                 float* db2, int accumulate, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !y || !bnstate || !dz || !partials || !nparts || pcap < 1) return DFD_EINVAL;
    if (!se_pooled || !se_ws || !dw1 || !dw2 || R < 1) return DFD_EINVAL;
    const SeWgradJob job{se_pooled, se_ws, N, C, R, dw1, db1, dw2, db2, accumulate};
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16
               ? act_bn_bwd_t<bf16>(D, y, gate, dpool, bnstate, act, dz, N, HW, C, partials, pcap, nparts, st, &job)
               : act_bn_bwd_t<float>(D, y, gate, dpool, bnstate, act, dz, N, HW, C, partials, pcap, nparts, st, &job);
}

extern "C" size_t dfd_pool_ws(int dtype, int N, int HW, int C) {
    if (!shape_ok(dtype, N, HW, C)) return 0;
    return pool_ws_bytes(N, HW, C, dtype == DFD_BF16 ? Vec<bf16>::N : Vec<float>::N);
}

// defer != nullptr: the caller's next kernel adds the split vectors itself (dfd_se_fwd / dfd_se_bwd): no k_pool_sum
// launch; defer[0] = splits (1: `out` is final), mul_out = the factor still to apply to the sum
template <typename T, bool BWD>
static int pool_t(const void* D, const void* y, const float* bnstate, int act, float* out, int N, int HW, int C,
                  void* ws, size_t ws_bytes, hipStream_t st, int* defer = nullptr, float* mul_out = nullptr) {
    const ChanMap cm = make_chanmap(C, Vec<T>::N);
    int splits = 1, rows_per = HW;
    if (ws) {
        pool_plan(N, HW, C, Vec<T>::N, &splits, &rows_per);
        if (splits > 1 && pool_ws_bytes(N, HW, C, Vec<T>::N) > ws_bytes) return DFD_EWORKSPACE;
    }
    float* parts = splits > 1 ? (float*)ws : nullptr;
    dim3 grid(cm.nvc, N, splits);
    const float mul = BWD ? 1.0f : 1.0f / (float)HW;
    DISPATCH_ACT(act, {
        hipLaunchKernelGGL((k_pool<T, ACT, BWD>), grid, dim3(DFD_THREADS), 0, st, (const T*)D, (const T*)y, bnstate, out, HW,
                           C, mul, cm, rows_per, parts);
    });
    if (defer) {
        *defer = splits;
        *mul_out = mul;
    } else if (splits > 1) {
        const long NC = (long)N * C;
        hipLaunchKernelGGL(k_pool_sum, dim3((unsigned)((NC + 255) / 256)), dim3(256), 0, st, parts, splits, NC, mul, out);
    }
    return DFD_CHECK_LAUNCH();
}
int dfd_pool_launch(int dtype, bool bwd, const void* D, const void* y, const float* bnstate, int act, float* out, int N,
                    int HW, int C, void* ws, size_t ws_bytes, hipStream_t st, int* splits, float* mul) {
    if (!shape_ok(dtype, N, HW, C) || !y || !bnstate || !out || (bwd && !D)) return DFD_EINVAL;
    if (dtype == DFD_BF16)
        return bwd ? pool_t<bf16, true>(D, y, bnstate, act, out, N, HW, C, ws, ws_bytes, st, splits, mul)
                   : pool_t<bf16, false>(nullptr, y, bnstate, act, out, N, HW, C, ws, ws_bytes, st, splits, mul);
    return bwd ? pool_t<float, true>(D, y, bnstate, act, out, N, HW, C, ws, ws_bytes, st, splits, mul)
               : pool_t<float, false>(nullptr, y, bnstate, act, out, N, HW, C, ws, ws_bytes, st, splits, mul);
}
extern "C" int dfd_pool_act(int dtype, const void* y, const float* bnstate, int act, float* pooled, int N, int HW,
                            int C, void* ws, size_t ws_bytes, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !y || !bnstate || !pooled) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16 ? pool_t<bf16, false>(nullptr, y, bnstate, act, pooled, N, HW, C, ws, ws_bytes, st)
                             : pool_t<float, false>(nullptr, y, bnstate, act, pooled, N, HW, C, ws, ws_bytes, st);
}
extern "C" int dfd_pool_bwd_reduce(int dtype, const void* D, const void* y, const float* bnstate, int act,
                                   float* dgate, int N, int HW, int C, void* ws, size_t ws_bytes, dfd_stream stream) {
    if (!shape_ok(dtype, N, HW, C) || !D || !y || !bnstate || !dgate) return DFD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    return dtype == DFD_BF16 ? pool_t<bf16, true>(D, y, bnstate, act, dgate, N, HW, C, ws, ws_bytes, st)
                             : pool_t<float, true>(D, y, bnstate, act, dgate, N, HW, C, ws, ws_bytes, st);
}

extern "C" int dfd_bn_finalize_ex(const float* partials, int nparts, int C, double count, const float* gamma,
                                  const float* beta, const float* conv_bias, const float* ls_gamma, float* running_mean,
                                  float* running_var, float momentum, float eps, float* bnstate, dfd_stream stream) {
    if (!partials || nparts < 1 || C < 1 || count <= 0 || !bnstate) return DFD_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return DFD_EINVAL;
    const int threads = FIN_CH * FIN_LANES, grid = (C + FIN_CH - 1) / FIN_CH;
    hipLaunchKernelGGL(k_bn_finalize, dim3(grid), dim3(threads), 0, (hipStream_t)stream, partials, nparts, C, count, gamma,
                       beta, conv_bias, ls_gamma, running_mean, running_var, momentum, eps, bnstate);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_finalize(const float* partials, int nparts, int C, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* bnstate, dfd_stream stream) {
    return dfd_bn_finalize_ex(partials, nparts, C, count, gamma, beta, nullptr, nullptr, running_mean, running_var, momentum,
                              eps, bnstate, stream);
}
extern "C" int dfd_bn_eval_coeffs_ex(const float* gamma, const float* beta, const float* conv_bias, const float* ls_gamma,
                                     const float* running_mean, const float* running_var, float eps, int C, float* bnstate,
                                     dfd_stream stream) {
    if (!running_mean || !running_var || C < 1 || !bnstate) return DFD_EINVAL;
    const int threads = 256, grid = (C + threads - 1) / threads;
    hipLaunchKernelGGL(k_bn_eval_coeffs, dim3(grid), dim3(threads), 0, (hipStream_t)stream, gamma, beta, conv_bias, ls_gamma,
                       running_mean, running_var, eps, C, bnstate);
    return DFD_CHECK_LAUNCH();
}
// every BatchNorm of a network in eval mode in one (or two) launches: jobs travel as kernel arguments
#define BN_EVAL_JOBS_PER_LAUNCH 40
struct BnEvalJobs { dfd_bn_eval_job j[BN_EVAL_JOBS_PER_LAUNCH]; };
__global__ void k_bn_eval_coeffs_multi(BnEvalJobs J) {
    const dfd_bn_eval_job& b = J.j[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= b.C) return;
    const float rstd = 1.0f / sqrtf(b.running_var[c] + b.eps);
    const float g = b.gamma ? b.gamma[c] : 1.f, be = b.beta ? b.beta[c] : 0.f;
    const float m = b.running_mean[c] - (b.conv_bias ? b.conv_bias[c] : 0.f);
    const float l = b.ls ? b.ls[c] : 1.f;
    const float scale = g * rstd;
    b.bnstate[c] = l * scale;
    b.bnstate[b.C + c] = l * (be - m * scale);
    b.bnstate[2 * b.C + c] = m;
    b.bnstate[3 * b.C + c] = rstd;
}
extern "C" int dfd_bn_eval_coeffs_multi(const dfd_bn_eval_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i)
        if (!jobs[i].running_mean || !jobs[i].running_var || !jobs[i].bnstate || jobs[i].C < 1) return DFD_EINVAL;
    for (int base = 0; base < njobs; base += BN_EVAL_JOBS_PER_LAUNCH) {
        BnEvalJobs J;
        const int cnt = njobs - base < BN_EVAL_JOBS_PER_LAUNCH ? njobs - base : BN_EVAL_JOBS_PER_LAUNCH;
        int maxC = 0;
        for (int i = 0; i < cnt; ++i) { J.j[i] = jobs[base + i]; if (J.j[i].C > maxC) maxC = J.j[i].C; }
        hipLaunchKernelGGL(k_bn_eval_coeffs_multi, dim3((maxC + 255) / 256, cnt), dim3(256), 0, (hipStream_t)stream, J);
    }
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, int C, float* bnstate, dfd_stream stream) {
    return dfd_bn_eval_coeffs_ex(gamma, beta, nullptr, nullptr, running_mean, running_var, eps, C, bnstate, stream);
}
extern "C" int dfd_bn_bwd_finalize_ex(const float* partials, int nparts, int C, double count, const float* gamma,
                                      const float* beta, const float* ls_gamma, const float* bnstate, int train,
                                      float* dgamma, float* dbeta, float* dls, float* dbias, int accumulate, float* coef,
                                      dfd_stream stream) {
    if (!partials || nparts < 1 || C < 1 || count <= 0 || !bnstate || !coef) return DFD_EINVAL;
    if (dls && (!ls_gamma || !beta)) return DFD_EINVAL;
    const int threads = FIN_CH * FIN_LANES, grid = (C + FIN_CH - 1) / FIN_CH;
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(grid), dim3(threads), 0, (hipStream_t)stream, partials, nparts, C, count,
                       gamma, beta, ls_gamma, bnstate, train, dgamma, dbeta, dls, dbias, accumulate, coef);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_bn_bwd_finalize(const float* partials, int nparts, int C, double count, const float* gamma,
                                   const float* bnstate, int train, float* dgamma, float* dbeta, int accumulate,
                                   float* coef, dfd_stream stream) {
    return dfd_bn_bwd_finalize_ex(partials, nparts, C, count, gamma, nullptr, nullptr, bnstate, train, dgamma, dbeta, nullptr,
                                  nullptr, accumulate, coef, stream);
}

extern "C" int dfd_version(void) { return 135; }   // see include/dfd_hip.h
