// dfd_coord.hip — every coordinate MLP of a FasterViT level in ONE launch.
//
// fastervit's PosEmbMLPSwinv1D (token position embedding) and PosEmbMLPSwinv2D (relative-position attention bias) are
//      table[t][:] = W2 . relu(W0 . coords[t] + b0)          coords [T][2] constant, W0 [Hd][2], b0 [Hd], W2 [D][Hd], Hd = 512
// per attention block (reference call sites: the third-party module's forward / backward at trainers/fastervit.py:271, :274).
// Their inputs are parameters and constant coordinate tables only — nothing depends on the activations — so all of them can
// run before the first block (forward) and after the last one (backward).  Launched per layer they were ~120 forward and ~160
// backward launches of 5-15 us for FasterViT-0 (a fifth of the step's launches); here a level's jobs travel by value in the
// kernel arguments (hipGraph-safe, no table upload) and take 2 + 2 launches.
//
//   k_coord_fwd_multi      one workgroup per (8 rows, job): h[8][512] in LDS, one wave per output column d (lanes stride k)
//   k_relpos_fwd_multi     bias[h][ng+i][ng+j] = 16 sigmoid(table[idx[i*nl+j]][h]) for every attention layer (dfd_relpos_bias_fwd)
//   k_relpos_bwd_multi     dtable[t][h] = 16 s (1-s) sum_{idx==t} dbias[h][..]   — one wave per (t, h), fixed order
//   k_coord_bwd_multi      one workgroup per (64-wide hidden chunk, job): h chunk recomputed into LDS, dtable staged through LDS
//                          16 / 32 columns at a time, then
//                            dw2[d][k] = sum_t dtable[t][d] h[t][k]
//                            dh[t][k]  = sum_d dtable[t][d] w2[d][k];  dw0[k][:] = sum_t [h>0] dh coords[t][:];  db0[k] likewise
//                          every sum in a fixed order (4 row groups combined through LDS in group order): reproducible.
// All f32.  Latency-bound by construction (a few MFLOP per job); what matters is that they are 4 launches, off the blocks' path.
#include "dfd_common.h"

#define CM_JOBS 24             // jobs per launch (24 x 88 B of kernel arguments; a level of FasterViT has 20-24 jobs)
#define CM_MAX_T 176           // rows of the largest table (13 x 13 = 169 relative coordinates)
struct CmJobs { dfd_cmlp_job j[CM_JOBS]; };
struct RpJobs { dfd_relpos_job j[CM_JOBS]; };

#define CM_ROWS 8               // table rows per workgroup of the forward kernel: W2 is read once per 8 rows
#define CM_DCH 64               // output columns per workgroup
// grid.x = row chunk * dchunks + column chunk
__global__ void __launch_bounds__(256)
k_coord_fwd_multi(CmJobs jobs, int dchunks) {
    const dfd_cmlp_job& jb = jobs.j[blockIdx.y];
    const int t0 = (blockIdx.x / dchunks) * CM_ROWS, d0 = (blockIdx.x % dchunks) * CM_DCH;
    if (t0 >= jb.T || d0 >= jb.D) return;
    __shared__ __attribute__((aligned(16))) float h[CM_ROWS][1024];
    for (int i = threadIdx.x; i < CM_ROWS * jb.Hd; i += 256) {
        const int r = i / jb.Hd, k = i - r * jb.Hd;
        const int t = t0 + r < jb.T ? t0 + r : jb.T - 1;
        const float v = fmaf(jb.coords[2 * t], jb.w0[2 * k], fmaf(jb.coords[2 * t + 1], jb.w0[2 * k + 1], jb.b0[k]));
        h[r][k] = v > 0.f ? v : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int dend = d0 + CM_DCH < jb.D ? d0 + CM_DCH : jb.D;
    for (int d = d0 + wave; d < dend; d += 4) {
        const float* wr = jb.w2 + (long)d * jb.Hd;
        float s[CM_ROWS];
#pragma unroll
        for (int r = 0; r < CM_ROWS; ++r) s[r] = 0.f;
        for (int k = lane * 4; k < jb.Hd; k += 256) {
            const float4 w = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
            for (int r = 0; r < CM_ROWS; ++r) {
                const float4 hv = *reinterpret_cast<const float4*>(&h[r][k]);
                s[r] = fmaf(hv.x, w.x, s[r]); s[r] = fmaf(hv.y, w.y, s[r]); s[r] = fmaf(hv.z, w.z, s[r]); s[r] = fmaf(hv.w, w.w, s[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < CM_ROWS; ++r) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s[r] += __shfl_xor(s[r], o);
            if (lane == 0 && t0 + r < jb.T) jb.table[(long)(t0 + r) * jb.D + d] = s[r];
        }
    }
}

__global__ void __launch_bounds__(256)
k_relpos_fwd_multi(RpJobs jobs) {
    const dfd_relpos_job& jb = jobs.j[blockIdx.y];
    const int S = jb.n_local + jb.n_global;
    const long total = (long)jb.H * S * S;
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % S), i = (int)((e / S) % S), h = (int)(e / ((long)S * S));
    float v = 0.f;
    if (i >= jb.n_global && j >= jb.n_global)
        v = 16.0f * sigmoid_f(jb.table[(long)jb.idx[(i - jb.n_global) * jb.n_local + (j - jb.n_global)] * jb.H + h]);
    jb.full[e] = v;
}

__global__ void __launch_bounds__(256)
k_relpos_bwd_multi(RpJobs jobs) {
    const dfd_relpos_job& jb = jobs.j[blockIdx.y];
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= jb.T * jb.H) return;
    const int t = wave / jb.H, h = wave - t * jb.H;
    const int S = jb.n_local + jb.n_global, nl = jb.n_local, ng = jb.n_global;
    float s = 0.f;
    for (int l = lane; l < nl * nl; l += 64) {
        if (jb.idx[l] != t) continue;
        const int i = l / nl, j = l - i * nl;
        s += jb.dfull[((long)h * S + ng + i) * S + ng + j];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        const float sg = sigmoid_f(jb.table[(long)t * jb.H + h]);
        jb.dtable[(long)t * jb.H + h] = s * 16.0f * sg * (1.0f - sg);
    }
}

// rows per row group kept in registers: RPG = 16 (T <= 64: position tables) or 44 (T <= 176: relative-coordinate tables)
template <int RPG>
__device__ __forceinline__ void coord_bwd_body(const dfd_cmlp_job& jb, float* hs, float* dts, float* red) {
    const int k0 = blockIdx.x * 64;
    const int kk = threadIdx.x & 63, grp = threadIdx.x >> 6, k = k0 + kk;
    const int T = jb.T, D = jb.D, Hd = jb.Hd;
    const bool kok = k < Hd;
    const int DC = T > 88 ? 16 : 32;              // columns of dtable staged per pass: T * DC floats fit `dts`
    const float w0x = kok ? jb.w0[2 * k] : 0.f, w0y = kok ? jb.w0[2 * k + 1] : 0.f, bb = kok ? jb.b0[k] : 0.f;
    for (int t = grp; t < T; t += 4) {
        const float v = fmaf(jb.coords[2 * t], w0x, fmaf(jb.coords[2 * t + 1], w0y, bb));
        hs[t * 64 + kk] = (kok && v > 0.f) ? v : 0.f;
    }
    float dh[RPG];
#pragma unroll
    for (int i = 0; i < RPG; ++i) dh[i] = 0.f;
    const bool want_dh = jb.dw0 || jb.db0;
    // the operands of pass p + 1 (this lane's W2 column piece, this thread's share of the dtable chunk) are requested before
    // the arithmetic of pass p: with < 1 workgroup per CU nothing else hides the two dependent memory round trips per pass
    constexpr int NST = (CM_MAX_T * 16 + 255) / 256;                   // dtable elements per thread and pass (<= 11)
    float w[32], nw[32], st[NST];
    auto fetch = [&](int d0) {
        const int dc = D - d0 < DC ? D - d0 : DC;
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = threadIdx.x + u * 256;
            const int t = i / DC, dd = i - t * DC;
            st[u] = (i < T * DC && dd < dc) ? jb.dtable[(long)t * D + d0 + dd] : 0.f;
        }
        if (want_dh) {
#pragma unroll
            for (int dd = 0; dd < 32; ++dd) nw[dd] = (kok && dd < dc) ? jb.w2[(long)(d0 + dd) * Hd + k] : 0.f;
        }
    };
    fetch(0);
    for (int d0 = 0; d0 < D; d0 += DC) {
        const int dc = D - d0 < DC ? D - d0 : DC;
        __syncthreads();                                              // previous pass done with dts (first pass: hs complete)
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = threadIdx.x + u * 256;
            if (i < T * DC) dts[i] = st[u];
        }
#pragma unroll
        for (int dd = 0; dd < 32; ++dd) w[dd] = nw[dd];
        __syncthreads();
        if (d0 + DC < D) fetch(d0 + DC);
        // dw2[d][k] = sum_t dtable[t][d] * h[t][k]   (ascending t)
        if (jb.dw2 && kok) {
            for (int dd = grp; dd < dc; dd += 4) {
                float s = 0.f;
                for (int t = 0; t < T; ++t) s = fmaf(dts[t * DC + dd], hs[t * 64 + kk], s);
                jb.dw2[(long)(d0 + dd) * Hd + k] = s;
            }
        }
        if (!want_dh) continue;
        // dh[t][k] += sum_{d in pass} dtable[t][d] * w2[d][k] for the rows of this group (t = grp, grp + 4, ...), d ascending
#pragma unroll
        for (int i = 0; i < RPG; ++i) {
            const int t = grp + 4 * i;
            if (t < T) {
                const float* row = dts + t * DC;
                float a = dh[i];
                if (DC == 32) {
#pragma unroll
                    for (int dd = 0; dd < 32; dd += 4) {
                        const float4 q = *reinterpret_cast<const float4*>(row + dd);
                        a = fmaf(q.x, w[dd], a); a = fmaf(q.y, w[dd + 1], a); a = fmaf(q.z, w[dd + 2], a); a = fmaf(q.w, w[dd + 3], a);
                    }
                } else {
#pragma unroll
                    for (int dd = 0; dd < 16; dd += 4) {
                        const float4 q = *reinterpret_cast<const float4*>(row + dd);
                        a = fmaf(q.x, w[dd], a); a = fmaf(q.y, w[dd + 1], a); a = fmaf(q.z, w[dd + 2], a); a = fmaf(q.w, w[dd + 3], a);
                    }
                }
                dh[i] = a;
            }
        }
    }
    if (!want_dh) return;
    float ax = 0.f, ay = 0.f, ab = 0.f;
#pragma unroll
    for (int i = 0; i < RPG; ++i) {
        const int t = grp + 4 * i;
        if (t < T && hs[t * 64 + kk] > 0.f) {
            ax = fmaf(dh[i], jb.coords[2 * t], ax);
            ay = fmaf(dh[i], jb.coords[2 * t + 1], ay);
            ab += dh[i];
        }
    }
    red[(0 * 4 + grp) * 64 + kk] = ax;
    red[(1 * 4 + grp) * 64 + kk] = ay;
    red[(2 * 4 + grp) * 64 + kk] = ab;
    __syncthreads();
    if (grp == 0 && kok) {
        float sx = 0.f, sy = 0.f, sb = 0.f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) { sx += red[(0 * 4 + gq) * 64 + kk]; sy += red[(1 * 4 + gq) * 64 + kk]; sb += red[(2 * 4 + gq) * 64 + kk]; }
        if (jb.dw0) { jb.dw0[2 * k] = sx; jb.dw0[2 * k + 1] = sy; }
        if (jb.db0) jb.db0[k] = sb;
    }
}

__global__ void __launch_bounds__(256)
k_coord_bwd_multi(CmJobs jobs) {
    const dfd_cmlp_job& jb = jobs.j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= jb.Hd) return;
    __shared__ __attribute__((aligned(16))) float hs[CM_MAX_T * 64];     // h[t][kk] of this hidden chunk
    __shared__ __attribute__((aligned(16))) float dts[CM_MAX_T * 16];    // dtable[t][d0 .. d0 + DC)
    __shared__ float red[3 * 4 * 64];                                    // dw0x, dw0y, db0 of the 4 row groups
    if (jb.T <= 64) coord_bwd_body<16>(jb, hs, dts, red);
    else coord_bwd_body<(CM_MAX_T + 3) / 4>(jb, hs, dts, red);
}

static bool cm_ok(const dfd_cmlp_job& j, bool bwd) {
    if (!j.coords || !j.w0 || !j.b0 || !j.w2 || j.T < 1 || j.T > CM_MAX_T || j.D < 1 || j.Hd < 4 || j.Hd > 1024 || (j.Hd & 3)) return false;
    return bwd ? (j.dtable && (j.dw0 || j.db0 || j.dw2)) : (j.table != nullptr);
}

extern "C" int dfd_coord_mlp_fwd_multi(const dfd_cmlp_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) if (!cm_ok(jobs[i], false)) return DFD_EINVAL;
    for (int base = 0; base < njobs; base += CM_JOBS) {
        CmJobs pj;
        const int cnt = njobs - base < CM_JOBS ? njobs - base : CM_JOBS;
        int most = 0, mostd = 0;
        for (int i = 0; i < cnt; ++i) {
            pj.j[i] = jobs[base + i];
            if (pj.j[i].T > most) most = pj.j[i].T;
            if (pj.j[i].D > mostd) mostd = pj.j[i].D;
        }
        const int dchunks = (mostd + CM_DCH - 1) / CM_DCH;
        hipLaunchKernelGGL(k_coord_fwd_multi, dim3(((most + CM_ROWS - 1) / CM_ROWS) * dchunks, cnt), dim3(256), 0, (hipStream_t)stream, pj, dchunks);
    }
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_coord_mlp_bwd_multi(const dfd_cmlp_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) if (!cm_ok(jobs[i], true)) return DFD_EINVAL;
    for (int base = 0; base < njobs; base += CM_JOBS) {
        CmJobs pj;
        const int cnt = njobs - base < CM_JOBS ? njobs - base : CM_JOBS;
        int most = 0;
        for (int i = 0; i < cnt; ++i) { pj.j[i] = jobs[base + i]; if (pj.j[i].Hd > most) most = pj.j[i].Hd; }
        hipLaunchKernelGGL(k_coord_bwd_multi, dim3((most + 63) / 64, cnt), dim3(256), 0, (hipStream_t)stream, pj);
    }
    return DFD_CHECK_LAUNCH();
}

static bool rp_ok(const dfd_relpos_job& j, bool bwd) {
    if (!j.table || !j.idx || j.H < 1 || j.T < 1 || j.n_local < 1 || j.n_global < 0) return false;
    return bwd ? (j.dfull && j.dtable) : (j.full != nullptr);
}

extern "C" int dfd_relpos_bias_fwd_multi(const dfd_relpos_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) if (!rp_ok(jobs[i], false)) return DFD_EINVAL;
    for (int base = 0; base < njobs; base += CM_JOBS) {
        RpJobs pj;
        const int cnt = njobs - base < CM_JOBS ? njobs - base : CM_JOBS;
        long most = 0;
        for (int i = 0; i < cnt; ++i) {
            pj.j[i] = jobs[base + i];
            const long S = pj.j[i].n_local + pj.j[i].n_global, tot = (long)pj.j[i].H * S * S;
            if (tot > most) most = tot;
        }
        hipLaunchKernelGGL(k_relpos_fwd_multi, dim3((unsigned)((most + 255) / 256), cnt), dim3(256), 0, (hipStream_t)stream, pj);
    }
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_relpos_bias_bwd_multi(const dfd_relpos_job* jobs, int njobs, dfd_stream stream) {
    if (!jobs || njobs < 1) return DFD_EINVAL;
    for (int i = 0; i < njobs; ++i) if (!rp_ok(jobs[i], true)) return DFD_EINVAL;
    for (int base = 0; base < njobs; base += CM_JOBS) {
        RpJobs pj;
        const int cnt = njobs - base < CM_JOBS ? njobs - base : CM_JOBS;
        long most = 0;
        for (int i = 0; i < cnt; ++i) {
            pj.j[i] = jobs[base + i];
            const long waves = (long)pj.j[i].T * pj.j[i].H;
            if (waves > most) most = waves;
        }
        hipLaunchKernelGGL(k_relpos_bwd_multi, dim3((unsigned)((most + 3) / 4), cnt), dim3(256), 0, (hipStream_t)stream, pj);
    }
    return DFD_CHECK_LAUNCH();
}
