// dfd_se.h — the squeeze-excite FC weight gradients as a device function, shared by its own kernel (dfd_misc.hip) and by the row-pass
// backward kernel that can carry it as extra workgroups (dfd_rowpass.hip, dfd_act_bn_bwd_se): the result is read by AdamW only, so it
// does not have to be a launch of its own on the block's dependency chain (16 launches of ~13 us per EfficientNet-B0 step).
#pragma once
#include "dfd_common.h"

__device__ __forceinline__ float se_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block (bx, by) of a (ceil(C / 64), R) grid of 256 threads; red: >= 3 * 4 * 64 floats of LDS.  ws = [g: N x C | dh: N x R | h: N x R]
// (dfd_se_bwd's workspace).  thread (c, nl) sums every 4th image, the four image lanes are combined through LDS in a fixed order.
__device__ __forceinline__ void se_fc_bwd_w_body(const float* __restrict__ pooled, const float* __restrict__ ws, int N, int C, int R,
                                                 float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                 float* __restrict__ db2, int accumulate, int bx, int by, float* red) {
    const float* ws_g = ws;
    const float* ws_dh = ws + (long)N * C;
    const float* ws_h = ws_dh + (long)N * R;
    const int t = threadIdx.x, cl = t & 63, nl = t >> 6;
    const int c = bx * 64 + cl, r = by;
    float a1 = 0.f, a2 = 0.f, sb2 = 0.f;
    if (c < C) {
        int n = nl;
        for (; n + 12 < N; n += 16) {                           // four images in flight per lane
            float g[4], pc[4], h[4], dh[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long m = n + 4 * u;
                g[u] = ws_g[m * C + c]; pc[u] = pooled[m * C + c]; h[u] = ws_h[m * R + r]; dh[u] = ws_dh[m * R + r];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a2 = fmaf(g[u], h[u], a2); a1 = fmaf(dh[u], pc[u], a1); sb2 += g[u]; }
        }
        for (; n < N; n += 4) {
            const float g = ws_g[(long)n * C + c], pc = pooled[(long)n * C + c];
            a2 = fmaf(g, ws_h[(long)n * R + r], a2);
            a1 = fmaf(ws_dh[(long)n * R + r], pc, a1);
            sb2 += g;
        }
    }
    red[(0 * 4 + nl) * 64 + cl] = a1; red[(1 * 4 + nl) * 64 + cl] = a2; red[(2 * 4 + nl) * 64 + cl] = sb2;
    __syncthreads();
    if (nl == 0 && c < C) {
        a1 = red[(0 * 4 + 0) * 64 + cl] + red[(0 * 4 + 1) * 64 + cl] + red[(0 * 4 + 2) * 64 + cl] + red[(0 * 4 + 3) * 64 + cl];
        a2 = red[(1 * 4 + 0) * 64 + cl] + red[(1 * 4 + 1) * 64 + cl] + red[(1 * 4 + 2) * 64 + cl] + red[(1 * 4 + 3) * 64 + cl];
        float* p2 = dw2 + (long)c * R + r;
        float* p1 = dw1 + (long)r * C + c;
        *p2 = (accumulate ? *p2 : 0.f) + a2;
        *p1 = (accumulate ? *p1 : 0.f) + a1;
        if (r == 0 && db2) {
            sb2 = red[(2 * 4 + 0) * 64 + cl] + red[(2 * 4 + 1) * 64 + cl] + red[(2 * 4 + 2) * 64 + cl] + red[(2 * 4 + 3) * 64 + cl];
            db2[c] = (accumulate ? db2[c] : 0.f) + sb2;
        }
    }
    if (bx == 0 && db1) {
        __syncthreads();
        float s = 0.f;
        for (int n = t; n < N; n += DFD_THREADS) s += ws_dh[(long)n * R + r];
        s = se_wave_sum(s);
        if ((t & 63) == 0) red[t >> 6] = s;
        __syncthreads();
        if (t == 0) db1[r] = (accumulate ? db1[r] : 0.f) + red[0] + red[1] + red[2] + red[3];
    }
}
