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
// `partials` must have room for P + ceil(P / SUM_GROUP) rows of L floats.
int dfd_launch_sum_partials(float* partials, int P, long L, float* out, int accumulate, hipStream_t st) {
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

