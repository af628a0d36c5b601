// dfd_dwconv.hip — fixed-order reduction of per-workgroup partial slabs (used by every
// weight-gradient kernel).  The depthwise kernels live in dfd_dwfwd.hip / dfd_dwbwd.hip.
#include "dfd_common.h"
#include "dfd_sum.h"
#include <mutex>

// out[i] (+)= sum_p partials[p][i], in a fixed order.  Two stages when there are many
// partial rows: groups of SUM_GROUP rows are summed by independent workgroups into the
// rows that FOLLOW the slab in the workspace ([P .. P + ceil(P/SUM_GROUP))), then those.
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
#ifndef SUM_1PASS_MAX_GROUPS
#define SUM_1PASS_MAX_GROUPS 8       // (16 = up to 512 rows in 64-byte pieces per group and row: FasterViT-0 19.73 -> 19.82 ms, S1 +-0; same-box A/B)
#endif
// Both stages in ONE launch for SUM_GROUP < P <= SUM_1PASS_MAX_GROUPS * SUM_GROUP (an immediate sum sits on the dependency chain, and its second stage is a
// ~5 us launch that adds 2..8 rows): thread (g, i) adds group g's rows of output i exactly as stage 1 does, the groups meet in LDS and the
// g == 0 threads add them in group order exactly as stage 2 does — same grouping, same order, same bits; the slab is not written.
// G2 = groups rounded up to a power of two; 256 / G2 outputs per workgroup, consecutive lanes on consecutive outputs.
__global__ void __launch_bounds__(256) k_sum_partials_1pass(const float* __restrict__ partials, int P, long L, float* __restrict__ out,
                                                             int accumulate, int g2_log2) {
    __shared__ float sm[256];
    const int opb = 256 >> g2_log2;
    const int g = threadIdx.x >> (8 - g2_log2), ii = threadIdx.x & (opb - 1);
    const long i = (long)blockIdx.x * opb + ii;
    const int G = (P + SUM_GROUP - 1) / SUM_GROUP;
    float s = 0.f;
    if (i < L && g < G) {
        const int p0 = g * SUM_GROUP;
        s = sum_rows_ordered(partials, p0, (p0 + SUM_GROUP < P) ? p0 + SUM_GROUP : P, L, i);
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < L) {
        float t = 0.f;
        for (int k = 0; k < G; ++k) t += sm[k * opb + ii];
        float* o = out + i;
        *o = (accumulate ? *o : 0.f) + t;
    }
}
// ---- batched form: between dfd_sum_batch_begin() and dfd_sum_batch_end() on one host thread, the final summation of
// every weight-gradient kernel launched from that thread is recorded instead of launched, and the batch is added up by
// ONE pair of launches (stage 1 over all (job, group) pairs, stage 2 over all jobs) — same grouping, same order, same
// bits as the unbatched form, 2 launches instead of up to 2 per weight gradient (they are ~5 us each, launch-floor bound).
__global__ void k_sum_multi(SumJobs J, int stage) { sum_multi_body(J, stage, blockIdx.x, blockIdx.y); }
static thread_local struct { bool on; SumJobs j; hipStream_t st; } tl_batch = {false, {}, nullptr};

static void sum_jobs_finish(SumJobs& J) {
    long maxL = 0;
    for (int k = 0; k < J.n; ++k) if (J.L[k] > maxL) maxL = J.L[k];
    J.gx = (int)((maxL + 255) / 256);
}
static void sum_jobs_launch(const SumJobs& J, int stage, hipStream_t st) {
    if (J.n == 0) return;
    if (stage == 1) { if (J.gofs[J.n] > 0) hipLaunchKernelGGL(k_sum_multi, dim3(J.gx, J.gofs[J.n]), dim3(256), 0, st, J, 1); }
    else hipLaunchKernelGGL(k_sum_multi, dim3(J.gx, J.n), dim3(256), 0, st, J, 2);
}
static int sum_batch_flush() {
    SumJobs& J = tl_batch.j;
    if (J.n == 0) return DFD_OK;
    sum_jobs_finish(J);
    sum_jobs_launch(J, 1, tl_batch.st);
    sum_jobs_launch(J, 2, tl_batch.st);
    J.n = 0;
    J.gofs[0] = 0;
    return DFD_CHECK_LAUNCH();
}

// ---- passengers.  A batch handed over by dfd_sum_batch_end_deferred() is not launched: its two stages ride along as extra
// workgroups of the next two launches on its stream that can carry them (dfd_act_bn_bwd: one per network block), and
// dfd_sum_passengers_flush() launches whatever is still waiting (the end of the backward pass).  The state is process-wide and
// keyed by the stream (the batch is closed on autograd's thread, the flush comes from the thread that called backward()).
static std::mutex g_pass_mu;
static struct { SumJobs j; hipStream_t st; } g_pass[2];          // [0]: stage 1 pending, [1]: stage 2 pending (n = 0: free)

void dfd_sum_passengers_take(hipStream_t st, SumJobs* stage1, SumJobs* stage2) {
    stage1->n = 0;
    stage2->n = 0;
    std::lock_guard<std::mutex> lock(g_pass_mu);
    if (g_pass[1].j.n > 0 && g_pass[1].st == st) { *stage2 = g_pass[1].j; g_pass[1].j.n = 0; }
    if (g_pass[0].j.n > 0 && g_pass[0].st == st && g_pass[1].j.n == 0) {
        *stage1 = g_pass[0].j;
        g_pass[1] = g_pass[0];
        g_pass[0].j.n = 0;
    }
}
extern "C" int dfd_sum_passengers_flush(dfd_stream stream) {
    hipStream_t st = (hipStream_t)stream;
    std::lock_guard<std::mutex> lock(g_pass_mu);
    if (g_pass[1].j.n > 0 && g_pass[1].st == st) { sum_jobs_launch(g_pass[1].j, 2, st); g_pass[1].j.n = 0; }
    if (g_pass[0].j.n > 0 && g_pass[0].st == st) {
        sum_jobs_launch(g_pass[0].j, 1, st);
        sum_jobs_launch(g_pass[0].j, 2, st);
        g_pass[0].j.n = 0;
    }
    return DFD_CHECK_LAUNCH();
}
// forget whatever still waits (a backward pass that ended in an exception never reached its flush: the jobs point at memory the next
// pass may no longer own)
extern "C" int dfd_sum_passengers_discard(void) {
    std::lock_guard<std::mutex> lock(g_pass_mu);
    g_pass[0].j.n = 0;
    g_pass[1].j.n = 0;
    return DFD_OK;
}
extern "C" int dfd_sum_batch_end_deferred(void) {
    if (!tl_batch.on) return DFD_EINVAL;
    tl_batch.on = false;
    SumJobs& J = tl_batch.j;
    if (J.n == 0) return DFD_OK;
    sum_jobs_finish(J);
    hipStream_t st = tl_batch.st;
    std::lock_guard<std::mutex> lock(g_pass_mu);
    // the stage-1 slot must be free: a batch nobody carried since the last hand-over is launched now (any order is fine: batches
    // write disjoint outputs)
    if (g_pass[0].j.n > 0) {
        if (g_pass[1].j.n > 0) { sum_jobs_launch(g_pass[1].j, 2, g_pass[1].st); g_pass[1].j.n = 0; }
        sum_jobs_launch(g_pass[0].j, 1, g_pass[0].st);
        sum_jobs_launch(g_pass[0].j, 2, g_pass[0].st);
        g_pass[0].j.n = 0;
    }
    if (J.gofs[J.n] == 0) {                              // no job has more than SUM_GROUP rows: stage 2 only
        if (g_pass[1].j.n > 0) { sum_jobs_launch(g_pass[1].j, 2, g_pass[1].st); g_pass[1].j.n = 0; }
        g_pass[1].j = J;
        g_pass[1].st = st;
    } else {
        g_pass[0].j = J;
        g_pass[0].st = st;
    }
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
    if (P > SUM_GROUP && P <= SUM_1PASS_MAX_GROUPS * SUM_GROUP) {
        const int G = (P + SUM_GROUP - 1) / SUM_GROUP;
        const int g2_log2 = G <= 2 ? 1 : (G <= 4 ? 2 : (G <= 8 ? 3 : 4));
        const long opb = 256 >> g2_log2;
        hipLaunchKernelGGL(k_sum_partials_1pass, dim3((unsigned)((L + opb - 1) / opb)), dim3(256), 0, st, partials, P, L, out, accumulate, g2_log2);
    } else if (P > SUM_GROUP) {
        const int G = (P + SUM_GROUP - 1) / SUM_GROUP;          // <= 32 for P <= 1024
        float* mid = partials + (long)P * L;
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, G), dim3(threads), 0, st, partials, P, L, mid, L, 0);
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, mid, G, L, out, 0, accumulate);
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(gx, 1), dim3(threads), 0, st, partials, P, L, out, 0, accumulate);
    }
    return DFD_CHECK_LAUNCH();
}
