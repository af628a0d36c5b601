// dfd_attn.hip — fused window attention on the bf16 matrix cores (FasterViT WindowAttention, head_dim 32, <= 64 tokens:
// 49 + 4 carrier tokens in level 2, 49 in level 3, 16 carrier tokens; BASELINE config 5 "hierarchical attention on CDNA4
// ... MFMA"; the arithmetic of the third-party module's forward at trainers/fastervit.py:271 and backward at :274).
//
//      S = scale * q k^T + bias[h]        P = softmax_k(S)        o = P v
//
// ONE WAVE per (window, head); S and P never leave the register file.  Replaces bgemm(q k^T) -> softmax -> bgemm(P v) of
// dfd_vit.hip (and their four backward GEMMs), which moved S and P through HBM as f32 [windows][heads][T][T] five times.
//
// Orientation (cdna_hip_programming.md "an accumulator tile as the next MFMA's operand"): every product is formed so that the
// index the NEXT product sums over sits in the accumulator's register (row) index:
//   forward   S^T = K Q^T            (A = k fragment, B = q fragment: both 16-byte global loads, d contiguous)
//             o^T = V^T P^T          (B = P^T accumulators re-packed to bf16: keys are rows; A = V^T by ds_read_tr16_b64)
//   backward  S^T, dP^T = V dO^T  -> dS^T -> dq^T = K^T dS^T          (A = K^T transposed read)
//             S, dP = dO V^T      -> P, dS -> dv^T = dO^T P, dk^T = Q^T dS   (A = dO^T, Q^T transposed reads)
// The k order of the re-packed operand is permuted — slot (g, j) of k-step s is token 32 s + 16 (j >> 2) + 4 g + (j & 3) — and
// the transposed reads fetch exactly those tokens: two ds_read_tr16_b64 blocks (4 tokens x 16 columns each) per fragment.
// Transposed outputs put 4 consecutive d of one token in a lane's 4 registers: 8-byte stores.
// Padding: T < 16 NT tokens — loads are clamped to row T-1 (finite values), padded keys get P = 0, padded queries get
// P = dS = 0 and are not stored.  softmax statistics in f32; P and dS rounded to bf16 for their products (f32 accumulate).
// Backward recomputes P from (q, k, bias, L) with L = row max + log(row sum) saved by the forward: nothing T x T is saved.
// Bias gradient dbias[h] = sum over windows of dS: the 4 waves of a workgroup (4 windows, same head) add their dS through LDS
// in wave order and the workgroup writes ONE partial row [H][T][T]; dfd_sum_rows adds the rows in a fixed order.
// Roofline: HBM — forward reads qkv once and writes o (+ L); backward reads qkv, dO, L and writes dqkv (+ partial rows).
#include "dfd_common.h"
#include "dfd_pw.h"

#define AT_HD 32                 // head dimension (all FasterViT variants)
#define AT_LDS_ROW 72            // bytes per token row of a staged [token][32] bf16 tile (64 + 8: 8-byte aligned, rows 18 banks apart)

typedef __attribute__((address_space(3))) short4_t lds_short4;

__device__ __forceinline__ bf16x8_t as_bf(const uint4& q) { return __builtin_bit_cast(bf16x8_t, q); }
__device__ __forceinline__ f32x4_t mma32(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// 8 f32 -> 8 bf16 (RNE), in k-slot order j = 0..7
__device__ __forceinline__ bf16x8_t pack8(const f32x4_t& lo, const f32x4_t& hi) {
    uint4 q;
    q.x = pack_bf2(lo[0], lo[1]); q.y = pack_bf2(lo[2], lo[3]);
    q.z = pack_bf2(hi[0], hi[1]); q.w = pack_bf2(hi[2], hi[3]);
    return as_bf(q);
}
// stage rows [0, 16 NT) of a [T][AT_HD] bf16 matrix (row pitch `pitch` elements) into a wave-private LDS tile; rows >= T are
// zero.  lane l writes 16-byte piece (l & 3) of rows (l >> 2) + 16 i as two 8-byte stores
template <int NT>
__device__ __forceinline__ void stage_tile(unsigned char* tile, const unsigned short* __restrict__ src, long pitch, int T, int lane) {
    uint4 r[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int row = 16 * i + (lane >> 2);
        r[i] = *reinterpret_cast<const uint4*>(src + (long)(row < T ? row : T - 1) * pitch + (lane & 3) * 8);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int row = 16 * i + (lane >> 2);
        uint4 v = row < T ? r[i] : make_uint4(0, 0, 0, 0);
        uint2* d = reinterpret_cast<uint2*>(tile + row * AT_LDS_ROW + (lane & 3) * 16);
        d[0] = make_uint2(v.x, v.y);
        d[1] = make_uint2(v.z, v.w);
    }
}
// A-operand fragment of X^T (row = d = 16 dt + (lane & 15), k-slot (g, j) = token 32 s + 16 (j >> 2) + 4 g + (j & 3)) from a
// staged [token][32] tile: two transposed 4-token x 16-column blocks.  EXEC must be all ones (every lane supplies an address).
__device__ __forceinline__ bf16x8_t tr_frag(const unsigned char* tile, int dt, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const unsigned char* a = tile + (32 * s + 4 * g + q) * AT_LDS_ROW + (16 * dt + 4 * p) * 2;
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(a));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(a + 16 * AT_LDS_ROW));
    return __builtin_bit_cast(bf16x8_t, (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
__device__ __forceinline__ float xor_max(float v, int lane_mask) { return __builtin_fmaxf(v, __shfl_xor(v, lane_mask)); }
__device__ __forceinline__ float xor_add(float v, int lane_mask) { return v + __shfl_xor(v, lane_mask); }

// ---------------------------------------------------------------------------------------------------------------- forward
// grid: ceil(n / 4) * H workgroups of 4 waves; workgroup (wg % H = head, wg / H = group of 4 windows)
template <int NT>
__global__ void __launch_bounds__(256)
k_wattn_fwd(const unsigned short* __restrict__ qkv, const float* __restrict__ bias, unsigned short* __restrict__ out, float* __restrict__ Lout,
            int n, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * AT_HD, TP = T | 1;
    float* bl = reinterpret_cast<float*>(smem);                               // [T][TP] bias of this head
    unsigned char* tiles = smem + (((size_t)T * TP * 4 + 15) & ~(size_t)15);  // [4 waves][16 NT][AT_LDS_ROW]
    const int h = blockIdx.x % H, grp = blockIdx.x / H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if (bias) {
        for (int i = threadIdx.x; i < T * T; i += 256) { const int qy = i / T; bl[qy * TP + (i - qy * T)] = bias[(long)h * T * T + i]; }
    }
    int b = grp * 4 + wave;
    const bool live = b < n;
    if (!live) b = n - 1;                                                     // idle waves redo the last window, store nothing
    const unsigned short* base = qkv + (long)b * T * 3 * C + h * AT_HD;
    unsigned char* vt = tiles + wave * (16 * NT * AT_LDS_ROW);
    stage_tile<NT>(vt, base + 2 * C, 3l * C, T, lane);
    bf16x8_t kf[NT], qf[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int row = 16 * t + c < T ? 16 * t + c : T - 1;
        kf[t] = as_bf(*reinterpret_cast<const uint4*>(base + C + (long)row * 3 * C + 8 * g));
        qf[t] = as_bf(*reinterpret_cast<const uint4*>(base + (long)row * 3 * C + 8 * g));
    }
    __syncthreads();                                                          // bias table and the V tile are in LDS
    f32x4_t acc[2][NT];                                                       // o^T[d = 16 dt + 4 g + r][query = 16 qt + c]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) acc[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t vfr[2][NT / 2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) vfr[dt][s] = tr_frag(vt, dt, s, lane);
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
        const int query = 16 * qt + c;
        const bool qok = query < T;
        f32x4_t st[NT];                                                       // S^T[key = 16 kt + 4 g + r][query]
        float m = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            st[kt] = mma32(kf[kt], qf[qt], f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                float x = st[kt][r] * scale;
                if (bias && qok && key < T) x += bl[query * TP + key];
                x = key < T ? x : -3.0e38f;
                st[kt][r] = x;
                m = __builtin_fmaxf(m, x);
            }
        }
        m = xor_max(xor_max(m, 16), 32);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (16 * kt + 4 * g + r) < T ? __expf(st[kt][r] - m) : 0.f;
                st[kt][r] = p;
                sum += p;
            }
        sum = xor_add(xor_add(sum, 16), 32);
        const float inv = __builtin_amdgcn_rcpf(sum);
        if (Lout && live && qok && g == 0) Lout[((long)b * H + h) * T + query] = m + __logf(sum);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kt][r] *= inv;
#pragma unroll
        for (int s = 0; s < NT / 2; ++s) {
            const bf16x8_t pf = pack8(st[2 * s], st[2 * s + 1]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) acc[dt][qt] = mma32(vfr[dt][s], pf, acc[dt][qt]);
        }
    }
    if (live) {
#pragma unroll
        for (int qt = 0; qt < NT; ++qt) {
            const int query = 16 * qt + c;
            if (query >= T) continue;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const f32x4_t o = acc[dt][qt];
                *reinterpret_cast<uint2*>(out + ((long)b * T + query) * C + h * AT_HD + 16 * dt + 4 * g) =
                    make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------------- backward
template <int NT>
__global__ void __launch_bounds__(256, 2)
k_wattn_bwd(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dout, const float* __restrict__ Lin,
            const float* __restrict__ bias, unsigned short* __restrict__ dqkv, float* __restrict__ dbias_parts, int n, int T, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int C = H * AT_HD, TP = T | 1;
    constexpr int TILE = 16 * NT * AT_LDS_ROW;
    float* bl = reinterpret_cast<float*>(smem);                               // [T][TP]
    const size_t off_tiles = ((size_t)T * TP * 4 + 15) & ~(size_t)15;
    unsigned char* tiles = smem + off_tiles;                                  // [4 waves][region: 3 tiles q, k, dO | later dS]
    const int need = (T * T * 4 + 15) & ~15;
    const int wreg = 3 * TILE > need ? 3 * TILE : need;                       // bytes per wave region
    float* dl = reinterpret_cast<float*>(tiles + 4 * (size_t)wreg);           // [4 waves][16 NT] row sums D
    const int h = blockIdx.x % H, grp = blockIdx.x / H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if (bias) {
        for (int i = threadIdx.x; i < T * T; i += 256) { const int qy = i / T; bl[qy * TP + (i - qy * T)] = bias[(long)h * T * T + i]; }
    }
    int b = grp * 4 + wave;
    const bool live = b < n;
    if (!live) b = n - 1;
    const unsigned short* base = qkv + (long)b * T * 3 * C + h * AT_HD;
    const unsigned short* dob = dout + (long)b * T * C + h * AT_HD;
    unsigned char* tq = tiles + (size_t)wave * wreg;
    unsigned char* tk = tq + TILE;
    unsigned char* td = tk + TILE;
    stage_tile<NT>(tq, base, 3l * C, T, lane);
    stage_tile<NT>(tk, base + C, 3l * C, T, lane);
    stage_tile<NT>(td, dob, (long)C, T, lane);
    __syncthreads();
    const float* Lrow = Lin + ((long)b * H + h) * T;
    // row fragment t of a [T][32] matrix in global memory: lane (row 16 t + c, d = 8 g ..)
    auto rowfrag = [&](const unsigned short* m, long pitch, int t) -> bf16x8_t {
        const int row = 16 * t + c < T ? 16 * t + c : T - 1;
        return as_bf(*reinterpret_cast<const uint4*>(m + (long)row * pitch + 8 * g));
    };

    // ---- phase A: query on the lanes, one 16-query tile per trip.  P^T, dP^T -> D, dS^T -> dq^T = K^T dS^T
    {
        bf16x8_t kf[NT], vf[NT], ktr[2][NT / 2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { kf[t] = rowfrag(base + C, 3l * C, t); vf[t] = rowfrag(base + 2 * C, 3l * C, t); }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < NT / 2; ++s) ktr[dt][s] = tr_frag(tk, dt, s, lane);
#pragma unroll 1
        for (int qt = 0; qt < NT; ++qt) {
            const int query = 16 * qt + c;
            const bool qok = query < T;
            const bf16x8_t qf = rowfrag(base, 3l * C, qt), df = rowfrag(dob, (long)C, qt);
            const float Lq = Lrow[qok ? query : T - 1];
            f32x4_t pt[NT], dpt[NT];
            float dsum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                pt[kt] = mma32(kf[kt], qf, f32x4_t{0.f, 0.f, 0.f, 0.f});
                dpt[kt] = mma32(vf[kt], df, f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = 16 * kt + 4 * g + r;
                    float x = pt[kt][r] * scale;
                    if (bias && qok && key < T) x += bl[query * TP + key];
                    const float p = (qok && key < T) ? __expf(x - Lq) : 0.f;
                    pt[kt][r] = p;
                    dsum = fmaf(p, dpt[kt][r], dsum);
                }
            }
            dsum = xor_add(xor_add(dsum, 16), 32);                            // D[query] = sum_key P dP
            if (g == 0) dl[wave * 16 * NT + query] = dsum;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) pt[kt][r] = pt[kt][r] * (dpt[kt][r] - dsum) * scale;      // scale * dS^T
            f32x4_t dq[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < NT / 2; ++s) {
                const bf16x8_t dsf = pack8(pt[2 * s], pt[2 * s + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq[dt] = mma32(ktr[dt][s], dsf, dq[dt]);
            }
            if (live && qok) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    *reinterpret_cast<uint2*>(dqkv + ((long)b * T + query) * 3 * C + h * AT_HD + 16 * dt + 4 * g) =
                        make_uint2(pack_bf2(dq[dt][0], dq[dt][1]), pack_bf2(dq[dt][2], dq[dt][3]));
            }
        }
    }
    // (D went through LDS inside this wave: LDS operations of one wave complete in order)

    // ---- phase B: key on the lanes, one 16-key tile per trip.  P, dP -> dS;  dv^T = dO^T P,  dk^T = Q^T dS
    // raw dS[query = 16 qt + 4 g + r][key = 16 kt + c] (the bias gradient) goes to this wave's OWN region, whose staged tiles
    // are dead once the transposed fragments below are in registers
    float* myds = reinterpret_cast<float*>(tq);
    {
        bf16x8_t qf[NT], df[NT], qtr[2][NT / 2], dtr[2][NT / 2];
#pragma unroll
        for (int t = 0; t < NT; ++t) { qf[t] = rowfrag(base, 3l * C, t); df[t] = rowfrag(dob, (long)C, t); }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int s = 0; s < NT / 2; ++s) { qtr[dt][s] = tr_frag(tq, dt, s, lane); dtr[dt][s] = tr_frag(td, dt, s, lane); }
        float Lr[NT][4], Dr[NT][4];
#pragma unroll
        for (int qt = 0; qt < NT; ++qt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int query = 16 * qt + 4 * g + r;
                Lr[qt][r] = Lrow[query < T ? query : T - 1];
                Dr[qt][r] = dl[wave * 16 * NT + query];
            }
#pragma unroll 1
        for (int kt = 0; kt < NT; ++kt) {
            const int key = 16 * kt + c;
            const bool kok = key < T;
            const bf16x8_t kf = rowfrag(base + C, 3l * C, kt), vf = rowfrag(base + 2 * C, 3l * C, kt);
            f32x4_t p[NT], ds[NT];
#pragma unroll
            for (int qt = 0; qt < NT; ++qt) {
                p[qt] = mma32(qf[qt], kf, f32x4_t{0.f, 0.f, 0.f, 0.f});              // S[query rows][key lanes]
                const f32x4_t dp = mma32(df[qt], vf, f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int query = 16 * qt + 4 * g + r;
                    float x = p[qt][r] * scale;
                    if (bias && kok && query < T) x += bl[query * TP + key];
                    const float pv = (kok && query < T) ? __expf(x - Lr[qt][r]) : 0.f;
                    p[qt][r] = pv;
                    const float d = pv * (dp[r] - Dr[qt][r]);
                    if (dbias_parts && kok && query < T) myds[query * T + key] = live ? d : 0.f;
                    ds[qt][r] = d * scale;
                }
            }
            f32x4_t dk[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}}, dv[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int s = 0; s < NT / 2; ++s) {
                const bf16x8_t pf = pack8(p[2 * s], p[2 * s + 1]);
                const bf16x8_t dsf = pack8(ds[2 * s], ds[2 * s + 1]);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = mma32(dtr[dt][s], pf, dv[dt]);
                    dk[dt] = mma32(qtr[dt][s], dsf, dk[dt]);
                }
            }
            if (live && kok) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    unsigned short* row = dqkv + ((long)b * T + key) * 3 * C + h * AT_HD + 16 * dt + 4 * g;
                    *reinterpret_cast<uint2*>(row + C) = make_uint2(pack_bf2(dk[dt][0], dk[dt][1]), pack_bf2(dk[dt][2], dk[dt][3]));
                    *reinterpret_cast<uint2*>(row + 2 * C) = make_uint2(pack_bf2(dv[dt][0], dv[dt][1]), pack_bf2(dv[dt][2], dv[dt][3]));
                }
            }
        }
    }
    if (!dbias_parts) return;
    // ---- bias gradient: the 4 windows of this workgroup added in wave order (fixed order: reproducible)
    __syncthreads();
    float* prow = dbias_parts + ((long)grp * H + h) * T * T;
    const float* r0 = reinterpret_cast<const float*>(tiles);
    const int ws = wreg >> 2;                                                 // floats between the waves' regions
    for (int i = threadIdx.x; i < T * T; i += 256) {
        float v = r0[i];
        v += r0[ws + i];
        v += r0[2 * ws + i];
        v += r0[3 * ws + i];
        prow[i] = v;
    }
}

static size_t wattn_lds(int T, int NT, bool bwd) {
    const size_t bias = ((size_t)T * (T | 1) * 4 + 15) & ~(size_t)15;
    const size_t tile = (size_t)16 * NT * AT_LDS_ROW;
    if (!bwd) return bias + 4 * tile;
    const size_t need = ((size_t)T * T * 4 + 15) & ~(size_t)15;
    const size_t wreg = 3 * tile > need ? 3 * tile : need;
    return bias + 4 * wreg + (size_t)4 * 16 * NT * 4;
}
static bool wattn_ok(int n, int T, int H, int hd) { return n >= 1 && T >= 1 && T <= 64 && H >= 1 && hd == AT_HD; }

/* rows of [H][T][T] the backward writes into dbias_parts */
extern "C" int dfd_wattn_parts(int n) { return (n + 3) / 4; }

extern "C" int dfd_wattn_fwd(const void* qkv, const float* bias, void* out, float* L, int n, int T, int H, int hd, float scale,
                             dfd_stream stream) {
    if (!qkv || !out) return DFD_EINVAL;
    if (!wattn_ok(n, T, H, hd)) return DFD_EUNSUPPORTED;
    const dim3 grid((unsigned)(((n + 3) / 4) * H));
    hipStream_t st = (hipStream_t)stream;
    if (T <= 32)
        hipLaunchKernelGGL((k_wattn_fwd<2>), grid, dim3(256), wattn_lds(T, 2, false), st, (const unsigned short*)qkv, bias,
                           (unsigned short*)out, L, n, T, H, scale);
    else
        hipLaunchKernelGGL((k_wattn_fwd<4>), grid, dim3(256), wattn_lds(T, 4, false), st, (const unsigned short*)qkv, bias,
                           (unsigned short*)out, L, n, T, H, scale);
    return DFD_CHECK_LAUNCH();
}

extern "C" int dfd_wattn_bwd(const void* qkv, const void* dout, const float* L, const float* bias, void* dqkv, float* dbias_parts,
                             int n, int T, int H, int hd, float scale, dfd_stream stream) {
    if (!qkv || !dout || !L || !dqkv) return DFD_EINVAL;
    if (!wattn_ok(n, T, H, hd)) return DFD_EUNSUPPORTED;
    const dim3 grid((unsigned)(((n + 3) / 4) * H));
    hipStream_t st = (hipStream_t)stream;
    if (T <= 32)
        hipLaunchKernelGGL((k_wattn_bwd<2>), grid, dim3(256), wattn_lds(T, 2, true), st, (const unsigned short*)qkv,
                           (const unsigned short*)dout, L, bias, (unsigned short*)dqkv, dbias_parts, n, T, H, scale);
    else
        hipLaunchKernelGGL((k_wattn_bwd<4>), grid, dim3(256), wattn_lds(T, 4, true), st, (const unsigned short*)qkv,
                           (const unsigned short*)dout, L, bias, (unsigned short*)dqkv, dbias_parts, n, T, H, scale);
    return DFD_CHECK_LAUNCH();
}

// =====================================================================================================================
// The attention GEMMs of EfficientFormerV2's Attention2d (timm; forward at trainers/efficientformer_v2.py:244) on the matrix
// cores.  Talking heads mix the 8 heads of a (query, key) pair before and after the softmax, so S / P / T2 stay f32 tensors
// [B][H][Nq][Nk] that the softmax kernels of dfd_vit.hip read and write; what moves here are the six batched products around
// them, which ran on dfd_bgemm (one 256-thread workgroup per (image, head): stage both operands element-wise into f32 LDS,
// 4 x 4 register tiles on the vector unit, scalar stores: 19-50 us per call for 0.2-0.7 GFLOP).  ONE WAVE per (image, head),
// <= 64 tokens on either side, fragments straight from global memory where the reduction index is contiguous, the other
// operand staged once as a [token][32] bf16 tile and read transposed (the helpers of the window-attention kernels above):
//   dfd_attn_scores   out[b][h][i][j] = alpha * sum_d x[b][i][h D + d] * y[b][j][h D + d] (+ bias[h][i][j])      S, dT2
//   dfd_attn_apply    out[b][i][h D + d] = alpha * sum_t f[b][h][i][t] * x[b][t][h D + d]  (f_trans: f[b][h][t][i])   O, dQ | dV, dK
// The f32 operand of dfd_attn_apply is rounded to bf16 for its product (f32 accumulation), as P and dS are in the window
// attention; everything else is the arithmetic of the bgemm path.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage rows [r0, r0 + 64) of a [T][.] bf16 matrix, columns [0, dvalid) of a 32-column chunk, into a wave-private [64][32] tile;
// rows >= T and columns >= dvalid are zero (never read from memory: the last chunk of the last head may end the tensor)
__device__ __forceinline__ void stage_tile_w(unsigned char* tile, const unsigned short* __restrict__ src, long pitch, int r0, int T,
                                             int dvalid, int lane) {
    uint4 r[4];
    const bool cok = (lane & 3) * 8 < dvalid;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 16 * i + (lane >> 2);
        r[i] = *reinterpret_cast<const uint4*>(src + (long)(row < T ? row : T - 1) * pitch + (cok ? (lane & 3) * 8 : 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lrow = 16 * i + (lane >> 2);
        const uint4 v = (r0 + lrow < T && cok) ? r[i] : make_uint4(0, 0, 0, 0);
        uint2* d = reinterpret_cast<uint2*>(tile + lrow * AT_LDS_ROW + (lane & 3) * 16);
        d[0] = make_uint2(v.x, v.y);
        d[1] = make_uint2(v.z, v.w);
    }
}

// one wave per (image, head, 64-row block of x, 64-row block of y); D % 8 == 0 (a partial last 32-wide step is zero-filled)
__global__ void __launch_bounds__(256)
k_attn_scores(const unsigned short* __restrict__ x, const unsigned short* __restrict__ y, float* __restrict__ out,
              const float* __restrict__ bias, float alpha, int n, int H, int Tx, int Ty, int D, int XB, int YB) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)n * H * XB * YB) return;
    const int yblk = (int)(item % YB), xblk = (int)((item / YB) % XB);
    const long bh = item / ((long)XB * YB);
    const int b = (int)(bh / H), h = (int)(bh - (long)b * H);
    const long pitch = (long)H * D;
    const int x0 = 64 * xblk, y0 = 64 * yblk;
    const unsigned short* xb = x + (long)b * Tx * pitch + h * D;
    const unsigned short* yb = y + (long)b * Ty * pitch + h * D;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < D; k0 += 32) {
        const bool kok = k0 + 8 * g < D;                                   // (lanes past D feed zeros; their address is clamped)
        const int ko = kok ? k0 + 8 * g : 0;
        bf16x8_t xf[4], yf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int rx = x0 + 16 * t + c < Tx ? x0 + 16 * t + c : Tx - 1, ry = y0 + 16 * t + c < Ty ? y0 + 16 * t + c : Ty - 1;
            const uint4 qx = *reinterpret_cast<const uint4*>(xb + rx * pitch + ko), qy = *reinterpret_cast<const uint4*>(yb + ry * pitch + ko);
            xf[t] = as_bf(kok ? qx : make_uint4(0, 0, 0, 0));
            yf[t] = as_bf(kok ? qy : make_uint4(0, 0, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mma32(xf[i], yf[j], acc[i][j]);       // [x row 16 i + 4 g + r][y row 16 j + c]
    }
    // straight from the accumulators: a store instruction covers 4 rows x 16 consecutive columns.  (Going through an LDS patch for
    // row-contiguous stores and bias reads measured SLOWER — 19.8 -> 22.9 us at 49 x 49 x 32: the index arithmetic and the round trip
    // cost more than the 64-byte segments do.)
    float* ob = out + bh * Tx * Ty;
    const float* bb = bias ? bias + (long)h * Tx * Ty : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = x0 + 16 * i + 4 * g + r;
            if (row >= Tx) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = y0 + 16 * j + c;
                if (col >= Ty) continue;
                float v = alpha * acc[i][j][r];
                if (bb) v += bb[row * Ty + col];
                ob[row * Ty + col] = v;
            }
        }
}

// one wave per (image, head, 64-row block of the output); the contraction index is walked in blocks of 64 tokens.
// TRANS = false: f[b][h][i][t] (To x Tc);  true: f[b][h][t][i] (Tc x To).
// DC > 0: accumulators of all DC 32-column chunks of D live across the token blocks (Tc > 64);  DC = 0: one token block (Tc <= 64) — the
// chunks are walked one after the other with 8 accumulator tiles (25 / 17 us against 26 / 25 for the general form at 49 x 49 x 128)
template <bool TRANS, int DC>
__global__ void __launch_bounds__(256)
k_attn_apply(const float* __restrict__ f, const unsigned short* __restrict__ x, unsigned short* __restrict__ out, float alpha,
             int n, int H, int To, int Tc, int D, int OB) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    long item = (long)blockIdx.x * 4 + wave;
    const bool live = item < (long)n * H * OB;
    if (!live) item = (long)n * H * OB - 1;              // (idle waves redo the last item, store nothing: transposed reads need EXEC all ones)
    const int oblk = (int)(item % OB);
    const long bh = item / OB;
    const int b = (int)(bh / H), h = (int)(bh - (long)b * H);
    const long pitch = (long)H * D;
    const int o0 = 64 * oblk;
    unsigned char* tile = smem + wave * (64 * AT_LDS_ROW);
    const float* fb = f + bh * To * Tc;
    const unsigned short* xb = x + (long)b * Tc * pitch + h * D;
    unsigned short* ob = out + (long)b * To * pitch + h * D;
    // the f32 operand as bf16 B fragments, straight from global memory: lane (output row o0 + 16 it + c), k-slot (g, j) of step s <->
    // token t0 + 32 s + 16 (j >> 2) + 4 g + (j & 3); 64 independent loads per lane (staging the block through LDS for row-contiguous
    // loads measured slower: 22.9 -> 32 us at 49 x 49 x 128)
    auto load_ff = [&](int t0, bf16x8_t (&ff)[4][2]) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = o0 + 16 * it + c;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f32x4_t lo, hi;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ta = t0 + 32 * s + 4 * g + u, tb = ta + 16;
                    const bool oka = i < To && ta < Tc, okb = i < To && tb < Tc;
                    const float va = fb[oka ? (TRANS ? (long)ta * To + i : (long)i * Tc + ta) : 0];
                    const float vb = fb[okb ? (TRANS ? (long)tb * To + i : (long)i * Tc + tb) : 0];
                    lo[u] = oka ? va : 0.f;
                    hi[u] = okb ? vb : 0.f;
                }
                ff[it][s] = pack8(lo, hi);
            }
        }
    };
    auto chunk_mma = [&](int t0, int dc, const bf16x8_t (&ff)[4][2], f32x4_t (&acc)[2][4]) {
        stage_tile_w(tile, xb + 32 * dc, pitch, t0, Tc, D - 32 * dc, lane);
        wave_sync_lds();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8_t xt = tr_frag(tile, dt, s, lane);
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[dt][it] = mma32(xt, ff[it][s], acc[dt][it]);       // [d 16 dt + 4 g + r][row 16 it + c]
            }
        wave_sync_lds();                                                  // every lane's transposed reads are done before the tile is restaged
    };
    auto chunk_store = [&](int dc, const f32x4_t (&acc)[2][4]) {
        if (!live) return;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = o0 + 16 * it + c;
            if (row >= To) continue;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                if (32 * dc + 16 * dt + 4 * g >= D) continue;
                const f32x4_t o = acc[dt][it];
                *reinterpret_cast<uint2*>(ob + row * pitch + 32 * dc + 16 * dt + 4 * g) =
                    make_uint2(pack_bf2(alpha * o[0], alpha * o[1]), pack_bf2(alpha * o[2], alpha * o[3]));
            }
        }
    };
    if constexpr (DC == 0) {
        bf16x8_t ff[4][2];
        load_ff(0, ff);
        const int ndc = (D + 31) >> 5;
#pragma unroll 1
        for (int dc = 0; dc < ndc; ++dc) {
            f32x4_t acc[2][4];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[dt][it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            chunk_mma(0, dc, ff, acc);
            chunk_store(dc, acc);
        }
    } else {
        f32x4_t acc[DC][2][4];
#pragma unroll
        for (int dc = 0; dc < DC; ++dc)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[dc][dt][it] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int t0 = 0; t0 < Tc; t0 += 64) {
            bf16x8_t ff[4][2];
            load_ff(t0, ff);
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) chunk_mma(t0, dc, ff, acc[dc]);
        }
#pragma unroll
        for (int dc = 0; dc < DC; ++dc) chunk_store(dc, acc[dc]);
    }
}

static bool attn2_ok(int n, int H, int Ta, int Tb, int D) { return n >= 1 && H >= 1 && Ta >= 1 && Ta <= 256 && Tb >= 1 && Tb <= 256 && D >= 8 && D <= 128 && D % 8 == 0; }

extern "C" int dfd_attn_scores(const void* x, const void* y, float* out, const float* bias, float alpha, int n, int H, int Tx, int Ty,
                               int D, dfd_stream stream) {
    if (!x || !y || !out) return DFD_EINVAL;
    if (!attn2_ok(n, H, Tx, Ty, D)) return DFD_EUNSUPPORTED;
    const int XB = (Tx + 63) / 64, YB = (Ty + 63) / 64;
    const dim3 grid((unsigned)(((long)n * H * XB * YB + 3) / 4));
    hipLaunchKernelGGL(k_attn_scores, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x, (const unsigned short*)y, out, bias,
                       alpha, n, H, Tx, Ty, D, XB, YB);
    return DFD_CHECK_LAUNCH();
}
extern "C" int dfd_attn_apply(const float* f, int f_trans, const void* x, void* out, float alpha, int n, int H, int To, int Tc, int D,
                              dfd_stream stream) {
    if (!f || !x || !out) return DFD_EINVAL;
    if (!attn2_ok(n, H, To, Tc, D)) return DFD_EUNSUPPORTED;
    const int OB = (To + 63) / 64;
    const dim3 grid((unsigned)(((long)n * H * OB + 3) / 4));
    const size_t lds = 4 * 64 * AT_LDS_ROW;
    hipStream_t st = (hipStream_t)stream;
#define AP_GO(TR, DCV) hipLaunchKernelGGL((k_attn_apply<TR, DCV>), grid, dim3(256), lds, st, f, (const unsigned short*)x, (unsigned short*)out, alpha, n, H, To, Tc, D, OB)
    const int dc = Tc <= 64 ? 0 : (D + 31) / 32;         // one token block: the chunk-by-chunk form
    if (f_trans) { if (dc == 0) AP_GO(true, 0); else if (dc == 1) AP_GO(true, 1); else if (dc == 2) AP_GO(true, 2); else if (dc == 3) AP_GO(true, 3); else AP_GO(true, 4); }
    else { if (dc == 0) AP_GO(false, 0); else if (dc == 1) AP_GO(false, 1); else if (dc == 2) AP_GO(false, 2); else if (dc == 3) AP_GO(false, 3); else AP_GO(false, 4); }
#undef AP_GO
    return DFD_CHECK_LAUNCH();
}
