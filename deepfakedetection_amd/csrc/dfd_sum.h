// dfd_sum.h — the batched fixed-order summation of weight-gradient partial slabs (dfd_dwconv.hip) as device code that other
// kernels can run as passenger workgroups (dfd_rowpass.hip: k_act_bn_bwd): the sums are read by the optimizer only, so they do not
// have to be launches of their own on the backward pass's dependency chain.
#pragma once
#include "dfd_common.h"

#define SUM_GROUP 32
#define SUM_MAX_JOBS 8
struct SumJobs {
    const float* parts[SUM_MAX_JOBS];
    float* out[SUM_MAX_JOBS];
    long L[SUM_MAX_JOBS];
    int P[SUM_MAX_JOBS], acc[SUM_MAX_JOBS];
    int gofs[SUM_MAX_JOBS + 1];             // first stage-1 group of each job (jobs with P <= SUM_GROUP have none)
    int n;
    int gx;                                 // 256-thread blocks along L (of the longest job)
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
// block (bx, by) of stage 1 (grid gx x gofs[n]: groups of SUM_GROUP rows -> the rows behind the slab) or stage 2 (grid gx x n)
__device__ __forceinline__ void sum_multi_body(const SumJobs& J, int stage, int bx, int by) {
    const long i = (long)bx * 256 + threadIdx.x;
    if (stage == 1) {
        int j = 0;
        while (j + 1 < J.n && by >= J.gofs[j + 1]) ++j;
        if (i >= J.L[j]) return;
        const int g = by - J.gofs[j], p0 = g * SUM_GROUP;
        const int p1 = (p0 + SUM_GROUP < J.P[j]) ? p0 + SUM_GROUP : J.P[j];
        float* mid = const_cast<float*>(J.parts[j]) + (long)J.P[j] * J.L[j];
        mid[(long)g * J.L[j] + i] = sum_rows_ordered(J.parts[j], p0, p1, J.L[j], i);
    } else {
        const int j = by;
        if (i >= J.L[j]) return;
        const int P = J.P[j];
        float s;
        if (P > SUM_GROUP) s = sum_rows_ordered(J.parts[j] + (long)P * J.L[j], 0, (P + SUM_GROUP - 1) / SUM_GROUP, J.L[j], i);
        else s = sum_rows_ordered(J.parts[j], 0, P, J.L[j], i);
        float* o = J.out[j] + i;
        *o = (J.acc[j] ? *o : 0.f) + s;
    }
}
// host side (dfd_dwconv.hip): batches handed over by dfd_sum_batch_end_deferred wait here for a launch on their stream that can carry
// them; `take` moves the stage-1 batch on to stage 2 and hands out both for this launch (n = 0: nothing)
void dfd_sum_passengers_take(hipStream_t st, SumJobs* stage1, SumJobs* stage2);
