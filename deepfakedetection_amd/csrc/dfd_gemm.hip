// dfd_gemm.hip — plain bf16 NT product out[M][N] = a[M][K] * w[N][K]^T for MANY rows and a deep K (FasterViT's Linear layers:
// 54 k x 256..1024 x 256..1024, forward and data gradient), the shape class where the 128 x 128 four-wave kernel of
// dfd_pwconv.hip stops at half of hipBLASLt's rate: its barrier-per-stage loop of 64 x 64-per-wave tiles costs 45 us on
// 54 k x 1024 x 256 with every global load removed, against 11 us of MFMA time.
//
// Structure (cdna_hip_programming.md 5, "glds vs register staging": 256^2 tile, one workgroup per CU, two LDS buffers, BK = 64):
//   * 256 x 256 output tile, 512 threads = 8 waves as 4 (rows) x 2 (columns): 64 x 128 outputs per wave = 32 accumulator
//     tiles; per 32-deep k sub-step 4 + 8 fragment reads feed 32 MFMAs (0.375 reads per MFMA against 0.5);
//   * operands go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`): no staging registers, no ds_write pass.  The LDS
//     image of a stage is [rows][64 k] in 128-byte rows with the 16-byte chunks XOR-swizzled by the row (conflict-free
//     fragment reads); a DMA instruction's destination is lane-linear (1 KB = 8 rows), so the swizzle is applied to the
//     per-lane SOURCE address;
//   * per stage: wait for the stage's DMA, barrier, all fragment reads, request the next stage into the other buffer, MFMAs;
//   * epilogue through an LDS overlay in whole 16-byte row pieces (8-byte stores from the accumulators measured slower).
// Rows / columns past the operands are clamped at the source and masked at the store.  Anything else (prologues, statistics,
// residuals, f32, K % 64, narrow N, too few tiles to fill the chip) stays with k_pw_nt.
#include "dfd_common.h"
#include "dfd_pw.h"

#ifndef GD_ROWS128
#define GD_ROWS128 1            // 128-row tiles where 256-row tiles would leave CUs idle (A/B switch)
#endif
#define GD_BN 256
#define GD_BK 64
#define GD_B_BYTES (GD_BN * 128)
#define GD_OROW (GD_BN * 2 + 16)
// WM = waves along the rows (4: 256-row tiles, 512 threads; 2: 128-row tiles, 256 threads — when 256-row tiles would not fill the chip)
template <int WM> struct GdGeom {
    static constexpr int BM = 64 * WM, NW = 2 * WM, THREADS = 64 * NW;
    static constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + GD_B_BYTES;
    static constexpr int LDS = (2 * STAGE) > (BM * GD_OROW) ? (2 * STAGE) : (BM * GD_OROW);
    static constexpr int JA = BM * 8 / 64 / NW, JB = GD_BN * 8 / 64 / NW;        // DMA instructions per wave and stage
};

typedef __attribute__((address_space(3))) void gd_lds_void;
typedef __attribute__((address_space(1))) const void gd_glb_void;

// EACT < 0: out = the product.  EACT = DFD_ACT_NONE / DFD_ACT_GELU: the BN-apply pass of the engine fused into the store
// loop — out = act(scale[n] * y + shift[n]) [* row_scale[row / HW]] [+ residual], with y the product ROUNDED to bf16 first (it
// sits in the LDS overlay as bf16), i.e. the same arithmetic on the same values as dfd_bn_act_apply on a stored y: identical
// bits; yraw (optional) also receives y itself (the layers whose backward needs the pre-activation).
struct GdEpi {
    const float* st;            // [2][N]: scale, shift
    const unsigned short* res;  // [M][N] or null
    const float* rs;            // per image or null
    unsigned short* yraw;       // [M][N] or null
    int HW;
};
template <int WM, int EACT>
__global__ void __launch_bounds__(GdGeom<WM>::THREADS, 1)
k_gemm_nt_dma(const unsigned short* __restrict__ a, const unsigned short* __restrict__ w, unsigned short* __restrict__ out, int M, int K,
              int N, int m_tiles, int n_tiles, GdEpi ep) {
    using G = GdGeom<WM>;
    constexpr int GD_BM = G::BM, GD_A_BYTES = G::A_BYTES, GD_STAGE = G::STAGE, GD_THREADS = G::THREADS, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware order: the column tiles of one row tile run back to back on one XCD (they share the A rows in its L2)
    const int total = m_tiles * n_tiles;
    int lin = blockIdx.x;
    {
        const int q = total >> 3, r = total & 7, xcd = lin & 7, slot = lin >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int mt = lin / n_tiles, nt = lin - mt * n_tiles;
    const int m0 = mt * GD_BM, n0 = nt * GD_BN;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fk = lane >> 4;
    const int nk = K / GD_BK;

    // DMA sources: instruction j of this wave fills LDS units [(j * 8 + wave) * 64, +64) of the A (B) image; unit u is row
    // u >> 3, slot u & 7, and holds source chunk (u & 7) ^ (row & 7) of that row
    const unsigned short* srcA[G::JA];
    const unsigned short* srcB[G::JB];
#pragma unroll
    for (int j = 0; j < G::JA; ++j) {
        const int u = (j * NW + wave) * 64 + lane, r = u >> 3, c = (u & 7) ^ (r & 7);
        srcA[j] = a + (long)(m0 + r < M ? m0 + r : M - 1) * K + c * 8;
    }
#pragma unroll
    for (int j = 0; j < G::JB; ++j) {
        const int u = (j * NW + wave) * 64 + lane, r = u >> 3, c = (u & 7) ^ (r & 7);
        srcB[j] = w + (long)(n0 + r < N ? n0 + r : N - 1) * K + c * 8;
    }
    auto request = [&](int kt, int buf) {
        unsigned char* sa = smem + buf * GD_STAGE;
        unsigned char* sb = sa + GD_A_BYTES;
#pragma unroll
        for (int j = 0; j < G::JA; ++j)
            __builtin_amdgcn_global_load_lds((gd_glb_void*)(srcA[j] + kt * GD_BK), (gd_lds_void*)(sa + (j * NW + wave) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < G::JB; ++j)
            __builtin_amdgcn_global_load_lds((gd_glb_void*)(srcB[j] + kt * GD_BK), (gd_lds_void*)(sb + (j * NW + wave) * 1024), 16, 0, 0);
    };

    f32x4_t acc[8][4];                                   // [n fragment][m fragment]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Per stage: wait for its DMA, barrier, read ALL of the stage's fragments (24 x 16 bytes per lane), THEN request the next
    // stage, then the 64 MFMAs.  The compiler orders every ds_read behind a pending LDS-DMA with s_waitcnt vmcnt(0) (it cannot
    // tell the buffers apart): requested before the fragment reads, the next stage's DMA was waited for on the spot and
    // overlapped nothing inside the wave; requested after them it drains underneath the MFMAs.
    request(0, 0);
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): this wave's DMA of stage kt has landed
        __syncthreads();                                 // ... everybody's has, and the other buffer is no longer read
        const unsigned char* ab = smem + buf * GD_STAGE;
        const unsigned char* bb = ab + GD_A_BYTES;
        uint4 fa[2][4], fw[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + fk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = wm * 64 + j * 16 + frow;
                fa[ks][j] = *reinterpret_cast<const uint4*>(ab + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = wn * 128 + i * 16 + frow;
                fw[ks][i] = *reinterpret_cast<const uint4*>(bb + r * 128 + ((c ^ (r & 7)) << 4));
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): the fragments are in registers
        if (kt + 1 < nk) request(kt + 1, buf ^ 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw[ks][i]),
                                                                        __builtin_bit_cast(bf16x8_t, fa[ks][j]), acc[i][j], 0, 0, 0);
        buf ^= 1;
    }
    // ---- accumulators -> LDS overlay [m][n] -> 16-byte row-major stores
    __syncthreads();                                     // every wave is done with the last stage's buffers
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = wm * 64 + j * 16 + frow, n = wn * 128 + i * 16 + fk * 4;
            *reinterpret_cast<uint2*>(smem + m * GD_OROW + n * 2) = make_uint2(pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3]));
        }
    __syncthreads();
#pragma unroll 4
    for (int idx = t; idx < GD_BM * (GD_BN / 8); idx += GD_THREADS) {
        const int r = idx >> 5, ch = idx & 31;           // 32 sixteen-byte pieces per row
        const int m = m0 + r, n = n0 + ch * 8;
        if (m < M && n < N) {
            uint4 q = *reinterpret_cast<const uint4*>(smem + r * GD_OROW + ch * 16);
            if constexpr (EACT >= 0) {
                if (ep.yraw) *reinterpret_cast<uint4*>(ep.yraw + (long)m * N + n) = q;
                float v[8], sc[8], sh[8];
                Vec<bf16>::unpack(q, v);
                load_f32<8>(ep.st + n, sc);
                load_f32<8>(ep.st + N + n, sh);
                bn_act_array<EACT, 8>(v, sc, sh);
                if (ep.rs) {
                    const float rr = ep.rs[m / ep.HW];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= rr;
                }
                if (ep.res) {
                    float qq[8];
                    Vec<bf16>::unpack(*reinterpret_cast<const uint4*>(ep.res + (long)m * N + n), qq);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += qq[j];
                }
                q = Vec<bf16>::pack(v);
            }
            *reinterpret_cast<uint4*>(out + (long)m * N + n) = q;
        }
    }
}

// DFD_EUNSUPPORTED: not this kernel's shape (the caller runs k_pw_nt)
template <int WM, int EACT>
static int gemm_nt_dma_launch(const void* a, const void* w, void* out, int M, int K, int N, int m_tiles, int n_tiles, const GdEpi& ep,
                              hipStream_t st) {
    using G = GdGeom<WM>;
    struct GdTag;
    dfd_allow_lds_once<GdTag>(k_gemm_nt_dma<WM, EACT>, G::LDS);
    hipLaunchKernelGGL((k_gemm_nt_dma<WM, EACT>), dim3((unsigned)(m_tiles * n_tiles)), dim3(G::THREADS), G::LDS, st, (const unsigned short*)a,
                       (const unsigned short*)w, (unsigned short*)out, M, K, N, m_tiles, n_tiles, ep);
    return DFD_CHECK_LAUNCH();
}
static bool gemm_nt_dma_plan(int M, int K, int N, int* wm, int* m_tiles, int* n_tiles) {
    if (K % GD_BK || K < 2 * GD_BK || N % 8 || N < GD_BN || M < 256) return false;
    *n_tiles = (N + GD_BN - 1) / GD_BN;
    const int mt256 = (M + 255) / 256, mt128 = (M + 127) / 128;
    if ((long)mt256 * *n_tiles >= 160) { *wm = 4; *m_tiles = mt256; return true; }
    if (GD_ROWS128 && (long)mt128 * *n_tiles >= 160) { *wm = 2; *m_tiles = mt128; return true; }
    return false;                                        // fewer tiles than CUs: the smaller tiles of k_pw_nt fill the chip better
}
int dfd_gemm_nt_dma(const void* a, const void* w, void* out, int M, int K, int N, hipStream_t st) {
    int wm, mt, nt;
    if (!gemm_nt_dma_plan(M, K, N, &wm, &mt, &nt)) return DFD_EUNSUPPORTED;
    const GdEpi ep{nullptr, nullptr, nullptr, nullptr, 1};
    return wm == 4 ? gemm_nt_dma_launch<4, -1>(a, w, out, M, K, N, mt, nt, ep, st) : gemm_nt_dma_launch<2, -1>(a, w, out, M, K, N, mt, nt, ep, st);
}
// which tile form serves a plain bf16 product (tests assert the path they name instead of trusting a silent fallback)
extern "C" int dfd_gemm_plan(int M, int K, int N) {
    int wm, mt, nt;
    return gemm_nt_dma_plan(M, K, N, &wm, &mt, &nt) ? 64 * wm : 0;
}
// out = act(scale * (a w^T) + shift) [* row_scale] [+ residual] (+ the raw product in yraw): include/dfd_hip.h, dfd_gemm_bias_act
extern "C" int dfd_gemm_bias_act(int dtype, const void* a, const void* w_nk, int M, int K, int N, const float* state, int act,
                              const void* residual, const float* row_scale, int HW, void* yraw, void* out, dfd_stream stream) {
    if (!a || !w_nk || !out || !state || M < 1 || K < 1 || N < 1 || (row_scale && HW < 1)) return DFD_EINVAL;
    if (dtype != DFD_BF16 || !(act == DFD_ACT_NONE || act == DFD_ACT_GELU)) return DFD_EUNSUPPORTED;
    int wm, mt, nt;
    if (!gemm_nt_dma_plan(M, K, N, &wm, &mt, &nt)) return DFD_EUNSUPPORTED;
    const GdEpi ep{state, (const unsigned short*)residual, row_scale, (unsigned short*)yraw, row_scale ? HW : 1};
    hipStream_t st = (hipStream_t)stream;
    if (act == DFD_ACT_GELU)
        return wm == 4 ? gemm_nt_dma_launch<4, DFD_ACT_GELU>(a, w_nk, out, M, K, N, mt, nt, ep, st)
                       : gemm_nt_dma_launch<2, DFD_ACT_GELU>(a, w_nk, out, M, K, N, mt, nt, ep, st);
    return wm == 4 ? gemm_nt_dma_launch<4, DFD_ACT_NONE>(a, w_nk, out, M, K, N, mt, nt, ep, st)
                   : gemm_nt_dma_launch<2, DFD_ACT_NONE>(a, w_nk, out, M, K, N, mt, nt, ep, st);
}
