// dfd_dwconv.hip — fixed-order reduction of per-workgroup partial slabs (used by every
// weight-gradient kernel).  The depthwise kernels live in dfd_dwfwd.hip / dfd_dwbwd.hip.
#include "dfd_common.h"

// out[i] (+)= sum_p partials[p][i], in a fixed order.  Two stages when there are many
// partial rows: groups of SUM_GROUP rows are summed by independent workgroups into the
// rows that FOLLOW the slab in the workspace ([P .. P + ceil(P/SUM_GROUP))), then those.
#define SUM_GROUP 32
__global__ void k_sum_partials(const float* __restrict__ partials, int P, long L, float* __restrict__ out,
                               long out_stride, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const int p0 = blockIdx.y * SUM_GROUP;
    const int p1 = (p0 + SUM_GROUP < P) ? p0 + SUM_GROUP : P;
    float s = 0.f;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {          // eight independent loads in flight, added in row order
        float u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = partials[(long)(p + k) * L + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += u[k];
    }
    for (; p < p1; ++p) s += partials[(long)p * L + i];
    float* o = out + (long)blockIdx.y * out_stride + i;
    *o = (accumulate ? *o : 0.f) + s;
}
// ---- batched form: between dfd_sum_batch_begin() and dfd_sum_batch_end() on one host thread, the final summation of
// every weight-gradient kernel launched from that thread is recorded instead of launched, and the batch is added up by
// ONE pair of launches (stage 1 over all (job, group) pairs, stage 2 over all jobs) — same grouping, same order, same
// bits as the unbatched form, 2 launches instead of up to 2 per weight gradient (they are ~5 us each, launch-floor bound).
#define SUM_MAX_JOBS 8
struct SumJobs {
    const float* parts[SUM_MAX_JOBS];
    float* out[SUM_MAX_JOBS];
    long L[SUM_MAX_JOBS];
    int P[SUM_MAX_JOBS], acc[SUM_MAX_JOBS];
    int gofs[SUM_MAX_JOBS + 1];             // first stage-1 group of each job (jobs with P <= SUM_GROUP have none)
    int n;
};
__device__ __forceinline__ float sum_rows_ordered(const float* __restrict__ base, int p0, int p1, long L, long i) {
    float s = 0.f;
    int p = p0;
    for (; p + 8 <= p1; p += 8) {
        float u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = base[(long)(p + k) * L + i];
#pragma unroll
        for (int k = 0; k < 8; ++k) s += u[k];
    }
    for (; p < p1; ++p) s += base[(long)p * L + i];
    return s;
}
__global__ void k_sum_multi(SumJobs J, int stage) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (stage == 1) {
        int j = 0;
        while (j + 1 < J.n && (int)blockIdx.y >= J.gofs[j + 1]) ++j;
        if (i >= J.L[j]) return;
        const int g = blockIdx.y - J.gofs[j], p0 = g * SUM_GROUP;
        const int p1 = (p0 + SUM_GROUP < J.P[j]) ? p0 + SUM_GROUP : J.P[j];
        float* mid = const_cast<float*>(J.parts[j]) + (long)J.P[j] * J.L[j];
        mid[(long)g * J.L[j] + i] = sum_rows_ordered(J.parts[j], p0, p1, J.L[j], i);
    } else {
        const int j = blockIdx.y;
        if (i >= J.L[j]) return;
        const int P = J.P[j];
        float s;
        if (P > SUM_GROUP) s = sum_rows_ordered(J.parts[j] + (long)P * J.L[j], 0, (P + SUM_GROUP - 1) / SUM_GROUP, J.L[j], i);
        else s = sum_rows_ordered(J.parts[j], 0, P, J.L[j], i);
        float* o = J.out[j] + i;
        *o = (J.acc[j] ? *o : 0.f) + s;
    }
}
static thread_local struct { bool on; SumJobs j; hipStream_t st; } tl_batch = {false, {}, nullptr};

static int sum_batch_flush() {
    SumJobs& J = tl_batch.j;
    if (J.n == 0) return DFD_OK;
    long maxL = 0;
    for (int k = 0; k < J.n; ++k) if (J.L[k] > maxL) maxL = J.L[k];
    const unsigned gx = (unsigned)((maxL + 255) / 256);
    if (J.gofs[J.n] > 0) hipLaunchKernelGGL(k_sum_multi, dim3(gx, J.gofs[J.n]), dim3(256), 0, tl_batch.st, J, 1);
    hipLaunchKernelGGL(k_sum_multi, dim3(gx, J.n), dim3(256), 0, tl_batch.st, J, 2);
    J.n = 0;
    J.gofs[0] = 0;
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_sum_batch_begin(void) {
    if (tl_batch.on) return DFD_EINVAL;
    tl_batch.on = true;
    tl_batch.j.n = 0;
    tl_batch.j.gofs[0] = 0;
    return DFD_OK;
}
extern "C" int dfd_sum_batch_end(void) {
    if (!tl_batch.on) return DFD_EINVAL;
    tl_batch.on = false;
    return sum_batch_flush();
}

// `partials` must have room for P + ceil(P / SUM_GROUP) rows of L floats.
// Inside a batch only slabs up to 24 MB wait for the batch's end (they are launch-bound: one pair of launches for all of them); a slab
// of tens of megabytes (FasterViT's level-2/3 linears: ~30 MB each) is summed at once, while it still sits in the last-level
// cache — deferred, four of them were read back from HBM (1.15 ms per FasterViT-0 step, 0.6 ms of it saved here).
#ifndef SUM_DEFER_MAX_BYTES
#define SUM_DEFER_MAX_BYTES (24l << 20)
#endif
int dfd_launch_sum_partials(float* partials, int P, long L, float* out, int accumulate, hipStream_t st, bool deferrable) {
    if (tl_batch.on && deferrable && (long)P * L * 4 <= SUM_DEFER_MAX_BYTES) {
        SumJobs& J = tl_batch.j;
        if (J.n == SUM_MAX_JOBS || (J.n > 0 && st != tl_batch.st)) {
            const int rc = sum_batch_flush();
            if (rc != DFD_OK) return rc;
        }
        tl_batch.st = st;
        const int k = J.n++;
        J.parts[k] = partials; J.out[k] = out; J.L[k] = L; J.P[k] = P; J.acc[k] = accumulate;
        J.gofs[k + 1] = J.gofs[k] + (P > SUM_GROUP ? (P + SUM_GROUP - 1) / SUM_GROUP : 0);
        return DFD_OK;
    }
    const int threads = 256;
    const unsigned gx = (unsigned)((L + threads - 1) / threads);
    if (P > SUM_GROUP) {
        const int G = (P + SUM_GROUP - 1) / SUM_GROUP;          // <= 32 for P <= 1024
        float* mid = partials + (long)P * L;
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, G), dim3(threads), 0, st, partials, P, L, mid, L, 0);
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, mid, G, L, out, 0, accumulate);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, partials, P, L, out, 0, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}
