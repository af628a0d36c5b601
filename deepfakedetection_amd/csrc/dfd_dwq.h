// dfd_dwq.h — shared pieces of the quad-based depthwise kernels (forward, data gradient).
#pragma once
#include "dfd_common.h"

typedef float f2 __attribute__((ext_vector_type(2)));

// workgroups a depthwise launch aims for (all channel chunks together); 1024 are co-resident at 4 per CU.  The training kernels
// (forward, data gradient, weight gradient) read theirs from dfd_tune keys 8-10 (default 1024, dfd_dwmm.hip); this constant is what
// the eval-form forward and the fused backward (off by default) still use.
#ifndef DFD_DW_GRID
#define DFD_DW_GRID 2048
#endif

// 16-byte vectors a lane requests before it touches the first one while staging a tile (a tile is at most ~10
// vectors per lane and tensor).  Measured per layer on MI355X (scripts/bench_layers.py): the 3x3 kernels gain up
// to 30 % from 6-8 in flight (fewer exposed memory round trips per work item); the 5x5 kernels sit at the
// 128-VGPR cap of four workgroups per CU and lose to the spills, so they keep 4.
template <int K, int S> struct StageDepth {
    static constexpr int X = (K == 3 && S == 1) ? 8 : 4;     // one tensor (input tile)
    static constexpr int DY = (K == 3) ? 6 : 4;              // two tensors (dz and y of the BN-backward map)
};

struct DwQGeom {
    int N, H, W, C, Ho, Wo, pt, pl;
    int CV, cvb_log2;
    int TH, QW, NQ;            // tile = TH rows x QW quads (4*QW columns), NQ = TH*QW
    unsigned qw_magic;         // q / QW == (q * qw_magic) >> 20
    int tiles_y, tiles_x, nwork;
    int IH, IW;
    unsigned iw_magic;
    int remap;                 // XCD-aware workgroup order (dwq_block)
};

// XCD-aware workgroup mapping for the (channel chunk, work slot) grid.  Hardware deals consecutive
// linear workgroup ids round-robin over the 8 XCDs, each with its own L2.  The chunk-workgroups of
// one spatial tile read interleaved 64..256-byte pieces of the same cache lines (pixel stride is not
// a multiple of 128 B), so they must share an L2: measured on block 2 (C = 144) the naive order
// fetched 3.05x the input bytes from HBM and wrote mostly 32-byte fragments.  Logical id =
// (xcd-major) so that ids that differ only in the chunk index land on one XCD, back to back.
// Measured per kernel and layer (EfficientNet-B0, batch 256): a clear win for the weight gradient
// (every layer whose chunks share lines) and the stride-1 data gradient (block 2: 377 -> 273 us),
// neutral-to-worse elsewhere, so the host enables it per launch (`remap`).
__device__ __forceinline__ void dwq_block(int& chunk, int& slot, int remap) {
    if (!remap) { chunk = blockIdx.x; slot = blockIdx.y; return; }
    const int nx = gridDim.x, total = nx * gridDim.y;
    const int lin = blockIdx.y * nx + blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, s = lin >> 3;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + s;
    slot = logical / nx;
    chunk = logical - slot * nx;
}

template <typename T> struct V2 { static constexpr int N = Vec<T>::N / 2; };

// A lane's V consecutive floats of a per-channel LDS table (tap weights, BN coefficients) as 16-byte reads.  Read as
// float2 pairs the compiler emits ds_read2_b64 (8 LDS cycles per wave instruction, 32-bank modulus: the 32-byte lane
// stride is a 4-way conflict); ds_read_b128 takes 4 cycles and is conflict-free at this stride for <= 8 lanes per pixel.
template <int N2>
__device__ __forceinline__ void lds_row(const float* p, f2 (&v)[N2]) {
    static_assert(N2 == 2 || N2 == 4, "V is 4 (f32) or 8 (bf16)");
    const float4 a = *reinterpret_cast<const float4*>(p);
    v[0] = (f2){a.x, a.y};
    v[1] = (f2){a.z, a.w};
    if constexpr (N2 == 4) {
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[2] = (f2){b.x, b.y};
        v[3] = (f2){b.z, b.w};
    }
}

__device__ __forceinline__ void unpack2(const uint4& q, f2 (&v)[4]) {       // 8 x bf16
    v[0] = (f2){__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u)};
    v[1] = (f2){__uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
    v[2] = (f2){__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u)};
    v[3] = (f2){__uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u)};
}
__device__ __forceinline__ void unpack2(const uint4& q, f2 (&v)[2]) {       // 4 x f32
    v[0] = (f2){__uint_as_float(q.x), __uint_as_float(q.y)};
    v[1] = (f2){__uint_as_float(q.z), __uint_as_float(q.w)};
}
__device__ __forceinline__ uint4 pack2(const f2 (&v)[4]) {
    return make_uint4(pack_bf2(v[0].x, v[0].y), pack_bf2(v[1].x, v[1].y), pack_bf2(v[2].x, v[2].y), pack_bf2(v[3].x, v[3].y));
}
__device__ __forceinline__ uint4 pack2(const f2 (&v)[2]) {
    return make_uint4(__float_as_uint(v[0].x), __float_as_uint(v[0].y), __float_as_uint(v[1].x), __float_as_uint(v[1].y));
}
template <typename T> __device__ __forceinline__ f2 round2(f2 v) {
    if constexpr (sizeof(T) == 2) return (f2){bf2f(f2bf(v.x)), bf2f(f2bf(v.y))};
    else return v;
}


// tile geometry chosen by a small cost model; centre_is_input selects the data-gradient form
// extra_lds: fixed bytes; extra_centre: bytes per centre pixel per channel vector (second tile);
// lane_div: pixel lanes are shared by this many roles (weight gradient: K kernel rows)
// halo_tiles: staged tiles of the halo extent that live in LDS at once (the fused backward stages dy AND the activated input)
bool dfd_dwq_geom(const dfd_dwconv_shape* s, int vec, int max_cvb, bool centre_is_input, size_t extra_lds,
                  int extra_centre, int lane_div, DwQGeom* g, int* tile_bytes, int halo_tiles = 1, long lds_budget = 36 * 1024);

// Occupancy class of a vector-unit depthwise launch (op: 0 forward, 1 data gradient, 2 weight gradient): FOUR workgroups per CU with
// tile + tables within 39 KB and a grid of the launch's dfd_tune target (1024), or THREE per CU within 48 KB and 3/4 of the target (768).
// The wide class buys larger tiles (less halo; for 5x5 layers with 16-vector chunks it is what lets a whole 7x7 picture + its 15 KB tap
// table be ONE tile) for a quarter of the resident waves.  Measured per EfficientNet-B0 layer at batch 256 (scripts/dw_ab.py):
// it wins on the 5x5 stride-1 layers (data gradient 7x7 C1152 82 -> 49 us, 28x28 C240 212 -> 174, 14x14 C480 / C672 86 -> 72 / 122 -> 111;
// forward 28x28 153 -> 140, 7x7 51 -> 43; weight gradient 14x14 80 -> 71 / 104 -> 101) and on the 3x3 stride-2 data gradient (389 -> 379,
// 63 -> 55), and loses on the other 3x3 and stride-2 launches (56x56 C144 forward 174 -> 189, weight gradient 232 -> 250; 14 -> 7 5x5
// forward 52 -> 63), so the rule below names those classes.  dfd_tune key 13: -1 this rule, 0 never wide, 1 always wide (A/B runs).
struct DwqOcc { long lds_budget; int grid; };
DwqOcc dwq_occupancy(int op, const dfd_dwconv_shape* s);

// stage the input tile: tile[pix][vl] = rnd(act(scale*x+shift)) or x, zero outside the image
template <typename T, int ACT, bool PRO, int U = 4>
__device__ __forceinline__ void stage_q(uint4* __restrict__ tile, const T* __restrict__ src, const f2 (&sc)[V2<T>::N],
                                        const f2 (&sh)[V2<T>::N], long img_base, int SH, int SW, int C, int c0, bool cvalid,
                                        int gy0, int gx0, int IH, int IW, unsigned magic, int cvb_log2) {
    constexpr int N2 = V2<T>::N;
    const int total = (IH * IW) << cvb_log2;
    for (int base = threadIdx.x; base < total; base += DFD_THREADS * U) {
        uint4 raw[U];
        bool inb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * DFD_THREADS;
            const int pix = idx >> cvb_log2;
            const int iy = (int)(((unsigned)pix * magic) >> 20);
            const int ix = pix - iy * IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            inb[u] = cvalid && idx < total && (unsigned)gy < (unsigned)SH && (unsigned)gx < (unsigned)SW;
            if (inb[u]) raw[u] = *reinterpret_cast<const uint4*>(src + img_base + ((long)gy * SW + gx) * C + c0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * DFD_THREADS;
            if (idx >= total) continue;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (inb[u]) {
                if constexpr (!PRO) {
                    q = raw[u];
                } else {
                    f2 v[N2];
                    unpack2(raw[u], v);
#pragma unroll
                    for (int j = 0; j < N2; ++j) {
                        const f2 z = __builtin_elementwise_fma(sc[j], v[j], sh[j]);
                        if constexpr (ACT == DFD_ACT_SILU) {
                            const f2 e = (f2){__expf(-z.x), __expf(-z.y)};
                            const f2 d = e + (f2){1.f, 1.f};
                            v[j] = z * (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                        } else {
                            v[j] = act_fwd2<ACT>(z);
                        }
                    }
                    q = pack2(v);
                }
            }
            tile[idx] = q;
        }
    }
}


// stage dy = ka*dz + kb*y + kc (or dz as is) for rows gy0.., cols gx0.. of the [SH][SW] dy image
template <typename T, bool COEF, int U = 4>
__device__ __forceinline__ void stage_dy(uint4* __restrict__ tile, const T* __restrict__ dz, const T* __restrict__ yraw,
                                         const float* __restrict__ cf, int cvbV, int vl, long img_base, int SH, int SW, int C,
                                         int c0, bool cvalid, int gy0, int gx0, int IH, int IW, unsigned magic, int cvb_log2) {
    constexpr int V = Vec<T>::N, N2 = V / 2;
    f2 ka[N2], kb[N2], kc[N2];
    if constexpr (COEF) {
        lds_row<N2>(cf + vl * V, ka);
        lds_row<N2>(cf + cvbV + vl * V, kb);
        lds_row<N2>(cf + 2 * cvbV + vl * V, kc);
    }
    const int total = (IH * IW) << cvb_log2;
    for (int base = threadIdx.x; base < total; base += DFD_THREADS * U) {
        uint4 r1[U], r2[U];
        bool inb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * DFD_THREADS;
            const int pix = idx >> cvb_log2;
            const int iy = (int)(((unsigned)pix * magic) >> 20);
            const int ix = pix - iy * IW;
            const int gy = gy0 + iy, gx = gx0 + ix;
            inb[u] = cvalid && idx < total && (unsigned)gy < (unsigned)SH && (unsigned)gx < (unsigned)SW;
            if (inb[u]) {
                const long off = img_base + ((long)gy * SW + gx) * C + c0;
                r1[u] = *reinterpret_cast<const uint4*>(dz + off);
                if constexpr (COEF) r2[u] = *reinterpret_cast<const uint4*>(yraw + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * DFD_THREADS;
            if (idx >= total) continue;
            uint4 q = make_uint4(0, 0, 0, 0);
            if (inb[u]) {
                if constexpr (!COEF) {
                    q = r1[u];
                } else {
                    f2 a[N2], b[N2];
                    unpack2(r1[u], a);
                    unpack2(r2[u], b);
#pragma unroll
                    for (int j = 0; j < N2; ++j)
                        a[j] = __builtin_elementwise_fma(ka[j], a[j], __builtin_elementwise_fma(kb[j], b[j], kc[j]));
                    q = pack2(a);
                }
            }
            tile[idx] = q;
        }
    }
}


#define DISPATCH_KS(KV, SV, ...)                                                        \
    if (KV == 3 && SV == 1) { constexpr int K = 3, S = 1; __VA_ARGS__; }                \
    else if (KV == 3 && SV == 2) { constexpr int K = 3, S = 2; __VA_ARGS__; }           \
    else if (KV == 5 && SV == 1) { constexpr int K = 5, S = 1; __VA_ARGS__; }           \
    else if (KV == 5 && SV == 2) { constexpr int K = 5, S = 2; __VA_ARGS__; }           \
    else return DFD_EUNSUPPORTED;
#define DISPATCH_ACT_DW(ACTV, ...)                                                   \
    switch (ACTV) {                                                                  \
        case DFD_ACT_NONE: { constexpr int ACT = DFD_ACT_NONE; __VA_ARGS__; } break; \
        case DFD_ACT_SILU: { constexpr int ACT = DFD_ACT_SILU; __VA_ARGS__; } break; \
        case DFD_ACT_RELU: { constexpr int ACT = DFD_ACT_RELU; __VA_ARGS__; } break; \
        case DFD_ACT_GELU: { constexpr int ACT = DFD_ACT_GELU; __VA_ARGS__; } break; \
        default: return DFD_EUNSUPPORTED;                                            \
    }
