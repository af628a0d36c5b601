// dfd_pwtnw.hip — 1x1 convolution weight gradient (TN form) for the large-M layers (bf16):
//
//     out[ia][jb] = sum_m A(a)[m][ia] * B(b)[m][jb]        a: narrow (<= 48 channels), b: wide (<= 144)
//
// EfficientNet blocks 0-3 (3.2M .. 200K rows, 16..40 x 32..144 channels).  The whole output fits a
// wave's accumulators, so — as in the NT kernel of dfd_pwntw.hip — one wave is one independent stream:
//   * a wave walks its own 32-row steps; per step it loads 32 rows of both operands (16-byte row
//     chunks, everything of the NEXT step requested before the current step's MFMAs), applies the
//     prologues (BN-backward affine map on either operand, BN + SiLU + SE gate on the wide one) and
//     writes the rows into a wave-private LDS tile in the swizzled layout of the tiled TN kernel;
//   * fragments are read back with ds_read_b64_tr_b16 (hardware transpose: 8 consecutive rows of one
//     channel per lane) and fed to v_mfma_f32_16x16x32_bf16; the wide operand goes through its tile
//     in column groups of 128 channels;
//   * no workgroup barrier in the main loop; at the end the 4 waves add their accumulators in LDS in
//     wave order (fixed order: reproducible) and the workgroup writes ONE partial slab, summed over
//     workgroups by the same two-stage k_sum_partials as the tiled kernel.
// Either operand may be the first GEMM operand: `swap` writes the slab transposed.
//
// DG (round 4): the EXPAND layer's backward in one pass.  Its weight gradient dW = d^T x and its data gradient dx = d W (+ skip)
// both start from d = a * dz + b * y + c, the BN-backward map of two [M][Cmid] tensors — the widest tensors of the network —
// and as two kernels each of them read that pair (blocks 1-3: 2 x 1.1 GB per step).  Here the wave that has just written its rows
// of d into its private tile for the transposed weight-gradient reads also multiplies them (plain 16-byte reads of the same tile,
// rows as the MFMA's N index) with the layer's [Cin][Cmid] weight, resident in LDS, and stores its rows of dx as 8-byte pieces
// (a lane's accumulator = 4 consecutive input channels of one row; 16 rows x Cin channels are contiguous in memory).  Same
// d values, same MFMA instruction and K order as dfd_pwntw.hip's data-gradient kernel: identical bits.
#include "dfd_pw.h"

// rows per wave step: 32 (v_mfma_f32_16x16x32_bf16) or, for the widest outputs where the accumulators
// leave fewer registers for in-flight rows, 16 (v_mfma_f32_16x16x16_bf16)
#define TNW_MIN_ROWS 16

// PA: prologue of the narrow operand (NONE / AFFINE2); PB: of the wide one (NONE / AFFINE2 / BN_ACT_GATE)
// DG: 0 weight gradient only; 1 also dx[M][Na] = B(b) w^T (w_kn = [Na][Nb] in bf16); 2 the same plus a residual added to dx
// NBT > 9 (the fused backward of 192- / 240-wide layers): one workgroup per CU — a wave's accumulators alone are 144-180 registers,
// at two waves per SIMD the kernel spilled 230-554 — and a wave's 16 rows in flight are ~16 KB, so four waves still keep a CU streaming
template <int ROWS, int NA, int NBT, int PA, int PB, int ACT, int DG = 0>
__global__ void __launch_bounds__(DFD_THREADS, NBT > 9 ? 1 : 2)
k_pw_tnw(const bf16* __restrict__ a, ProArgs pa, int Na, const bf16* __restrict__ b, ProArgs pb, int Nb, int M,
         int rows_per_block, int swap, float* __restrict__ ws, const bf16* __restrict__ w_kn = nullptr,
         const bf16* __restrict__ res = nullptr, bf16* __restrict__ dx = nullptr, int wstride = 0, int woff = 0) {
    constexpr int ITA = (ROWS * NA * 2 + 63) / 64;          // 16-byte items per lane and step
    constexpr int ITB = (ROWS * NBT * 2 + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned char* at = smem + wave * (2 * ROWS * 256);     // this wave's A tile, then its B tile
    unsigned char* bt = at + ROWS * 256;
    float* cl = reinterpret_cast<float*>(smem + 4 * 2 * ROWS * 256);      // coefficients: A 3*Na, B 3*Nb
    const bf16* a2 = reinterpret_cast<const bf16*>(pa.a2);
    const bf16* b2 = reinterpret_cast<const bf16*>(pb.a2);
    const int ca = Na >> 3, cb = Nb >> 3;                       // 16-byte chunks per row
    // PB = BN_ACT_GATE: this wave's copy of the squeeze-excite gate rows of images gimg and gimg + 1 (a step of ROWS <= HW
    // rows touches no other).  Read from global memory inside the prologue, every item waited on vmcnt(0) for its own gate
    // values — in-order completion also drains the next step's row loads — i.e. one memory round trip per item.
    float* gc = cl + 3 * (Na + Nb) + wave * 2 * Nb;
    int gimg = -1;

    for (int i = t; i < 3 * Na; i += DFD_THREADS) cl[i] = (PA != DFD_PRO_NONE) ? pa.coef[i] : 0.f;
    for (int i = t; i < 3 * Nb; i += DFD_THREADS) {
        float v = 0.f;
        if (PB == DFD_PRO_AFFINE2) v = pb.coef[i];
        else if (PB != DFD_PRO_NONE && i < 2 * Nb) v = pb.coef[i];
        cl[3 * Na + i] = v;
    }
    // chunks past the operand's width are never written: clear the tiles once so that partly valid
    // 16-wide MFMA tiles read zeros there
    for (int i = t; i < 4 * 2 * ROWS * 256 / 16; i += DFD_THREADS) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    if constexpr (DG) {
        // the layer's weight as the data gradient's A operand: NA * 16 rows (input channels) x Nb (zero padded to the row stride),
        // row stride = 64-byte k-steps + 16 so that the 16 rows of a fragment read land on different bank groups
        const int chunks = wstride >> 4;
        for (int i = t; i < NA * 16 * chunks; i += DFD_THREADS) {
            const int r = i / chunks, c = i - r * chunks;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (r < Na && c < cb) v = *reinterpret_cast<const uint4*>(w_kn + (long)r * Nb + c * 8);
            *reinterpret_cast<uint4*>(smem + woff + r * wstride + c * 16) = v;
        }
    }
    __syncthreads();

    // item -> (row, chunk) of this lane, fixed for the whole kernel
    int ra[ITA], cha[ITA], rb[ITB], chb[ITB];
#pragma unroll
    for (int i = 0; i < ITA; ++i) {
        const int idx = lane + 64 * i;
        const bool ok = idx < ROWS * ca;
        ra[i] = ok ? idx / ca : -1;
        cha[i] = ok ? idx - ra[i] * ca : 0;
    }
#pragma unroll
    for (int i = 0; i < ITB; ++i) {
        const int idx = lane + 64 * i;
        const bool ok = idx < ROWS * cb;
        rb[i] = ok ? idx / cb : -1;
        chb[i] = ok ? idx - rb[i] * cb : 0;
    }

    f32x4_t acc[NA][NBT];
#pragma unroll
    for (int x = 0; x < NA; ++x)
#pragma unroll
        for (int y = 0; y < NBT; ++y) acc[x][y] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int na = (Na + 15) >> 4, nbt = (Nb + 15) >> 4;        // valid 16-wide tiles
    constexpr int RT = ROWS / 16;                               // 16-row tiles of a step (the data gradient's N tiles)
    f32x4_t accd[DG ? NA : 1][RT];

    const int rows_per_wave = rows_per_block / 4;
    const int mbeg = blockIdx.x * rows_per_block + wave * rows_per_wave;
    int mend = mbeg + rows_per_wave;
    if (mend > M) mend = M;

    // One register set for the rows in flight (the accumulators take the rest of the 128-register budget).  An item's
    // registers are refilled with the NEXT step's row as soon as this step's prologue has consumed them (s_store_*), so
    // that every load has the rest of the prologue arithmetic and the MFMA phase to land — requested in one batch after
    // the stores, the loads had only the MFMA phase (a few hundred cycles against ~2 us of latency).
    uint4 va[ITA], va2[ITA], vb[ITB], vb2[ITB];
    // Loads are UNCONDITIONAL (row and chunk clamped into the operand; what an absent item or a row past the wave's
    // range loads is never stored): with a load inside a divergent branch the compiler cannot count the loads in flight
    // and falls back to s_waitcnt vmcnt(0) in front of every item, which serialises one memory round trip per item.
    auto load_a = [&](int i, int m0) {
        int row = m0 + (ra[i] < 0 ? 0 : ra[i]);
        row = row < M ? row : M - 1;
        const long off = (long)row * Na + cha[i] * 8;
        va[i] = *reinterpret_cast<const uint4*>(a + off);
        if constexpr (PA == DFD_PRO_AFFINE2) va2[i] = *reinterpret_cast<const uint4*>(a2 + off);
    };
    auto load_b = [&](int i, int m0) {
        int row = m0 + (rb[i] < 0 ? 0 : rb[i]);
        row = row < M ? row : M - 1;
        const long off = (long)row * Nb + chb[i] * 8;
        vb[i] = *reinterpret_cast<const uint4*>(b + off);
        if constexpr (PB == DFD_PRO_AFFINE2) vb2[i] = *reinterpret_cast<const uint4*>(b2 + off);
    };
    auto g_load = [&](int m0) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) load_a(i, m0);
#pragma unroll
        for (int i = 0; i < ITB; ++i) load_b(i, m0);
    };
    // one operand item through its prologue
    auto pro_item = [&](auto mode_tag, uint4 q, uint4 q2, const float* cf, int N, int ch, int m, const ProArgs& pr) -> uint4 {
        constexpr int MODE = decltype(mode_tag)::value;
        if constexpr (MODE == DFD_PRO_NONE) {
            return q;
        } else {
            float v[8], c0[8], c1[8];
            q_to_f(q, v);
            load_f32<8>(cf + ch * 8, c0);
            load_f32<8>(cf + N + ch * 8, c1);
            if constexpr (MODE == DFD_PRO_AFFINE2) {
                float v2[8], c2[8];
                q_to_f(q2, v2);
                load_f32<8>(cf + 2 * N + ch * 8, c2);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(c0[j], v[j], fmaf(c1[j], v2[j], c2[j]));
            } else if constexpr (MODE == DFD_PRO_BN_ACT) {
                // BN + activation without a squeeze-excite gate (EfficientFormerV2 ConvMlp: GELU(BN(mid conv)))
                bn_act_array<ACT, 8>(v, c0, c1);
            } else {
                float gt[8];
                load_f32<8>(gc + (pro_image(pr, m) - gimg) * N + ch * 8, gt);
                bn_act_array<ACT, 8>(v, c0, c1);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = round_to<bf16>(v[j]) * gt[j];
            }
            return f_to_q(v);
        }
    };
    // registers -> wave-private tiles (a whole; b one column group of 128 channels = 16 chunks)
    auto s_store_a = [&](int m0) {
#pragma unroll
        for (int i = 0; i < ITA; ++i) {
            if (ra[i] >= 0) {                                   // arithmetic and LDS only inside the branch
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m0 + ra[i] < mend) v = pro_item(std::integral_constant<int, PA>{}, va[i], va2[i], cl, Na, cha[i], m0 + ra[i], pa);
                *reinterpret_cast<uint4*>(at + tn_off_bf16(ra[i], cha[i])) = v;
            }
            load_a(i, m0 + ROWS);
        }
    };
    // refill: this call is the last one that consumes B items of the step (the second column group when there are two)
    auto s_store_b = [&](int m0, int grp, auto refill_tag) {
        constexpr bool refill = decltype(refill_tag)::value;
#pragma unroll
        for (int i = 0; i < ITB; ++i) {
            if (rb[i] >= 0 && (chb[i] >> 4) == grp) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m0 + rb[i] < mend)
                    v = pro_item(std::integral_constant<int, PB>{}, vb[i], vb2[i], cl + 3 * Na, Nb, chb[i], m0 + rb[i], pb);
                *reinterpret_cast<uint4*>(bt + tn_off_bf16(rb[i], chb[i] & 15)) = v;
            }
            if constexpr (refill) load_b(i, m0 + ROWS);
        }
    };
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // transposed fragment of 16 channels starting at chunk `ch0` of a tile: 8 consecutive rows per lane
    const int g = lane >> 4, li = lane & 15, qrow = li >> 2, pcol = li & 3;
    auto frag = [&](const unsigned char* tile, int ch0) -> short8_t {
        const int ch = ch0 + (pcol >> 1), sub = (pcol & 1) << 3;
        if constexpr (ROWS == 32) {
            const int m_lo = 8 * g + qrow, m_hi = m_lo + 4;
            const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(tile + tn_off_bf16(m_lo, ch) + sub));
            const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(tile + tn_off_bf16(m_hi, ch) + sub));
            return (short8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
            const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4_t*)(tile + tn_off_bf16(4 * g + qrow, ch) + sub));
            return (short8_t){lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
        }
    };
    auto mma = [&](const short8_t& fx, const short8_t& fy, f32x4_t c) -> f32x4_t {
        if constexpr (ROWS == 32) {
            return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fx), __builtin_bit_cast(bf16x8_t, fy), c, 0, 0, 0);
        } else {
            const short4_t x4 = (short4_t){fx[0], fx[1], fx[2], fx[3]}, y4 = (short4_t){fy[0], fy[1], fy[2], fy[3]};
            return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x4, y4, c, 0, 0, 0);
        }
    };

    constexpr int NGRP = NBT > 8 ? 2 : 1;               // column groups of 128 channels the B tile is walked in
    if (mbeg < mend) g_load(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += ROWS) {
        short8_t fa[NA];
        if constexpr (PB == DFD_PRO_BN_ACT_GATE) {
            const int img0 = pro_image(pb, m0);                 // wave-uniform
            if (img0 != gimg) {
                gimg = img0;
                for (int i = lane; i < 2 * Nb; i += 64) {
                    const int im = img0 + (i >= Nb ? 1 : 0);
                    gc[i] = ((long)im * pb.HW < M) ? pb.gate[(long)im * Nb + (i >= Nb ? i - Nb : i)] : 0.f;
                }
                wave_sync();
            }
        }
        if constexpr (DG) {
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
                for (int j = 0; j < RT; ++j) accd[x][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        s_store_a(m0);
        s_store_b(m0, 0, std::integral_constant<bool, NGRP == 1>{});
        wave_sync();
#pragma unroll
        for (int x = 0; x < NA; ++x)
            if (x < na) fa[x] = frag(at, 2 * x);
#pragma unroll
        for (int grp = 0; grp < NGRP; ++grp) {
            if (grp > 0) {
                wave_sync();                            // group grp-1's fragment reads are done
                s_store_b(m0, grp, std::true_type{});   // (NGRP == 2: the last group)
                wave_sync();
            }
#pragma unroll
            for (int y = 0; y < NBT; ++y) {
                if ((y >> 3) != grp || y >= nbt) continue;
                const short8_t fb = frag(bt, 2 * (y & 7));
#pragma unroll
                for (int x = 0; x < NA; ++x)
                    if (x < na)
                        acc[x][y] = mma(fa[x], fb, acc[x][y]);
            }
            if constexpr (DG) {
                // data gradient over this group's channels: k-steps of 32 channels = 4 chunks; a lane's B fragment is chunk
                // 4 ks + (lane >> 4) of row (lane & 15) of the tile, as stored (no transpose), zero past the operand's width
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                                            // a group = 128 channels = 4 k-steps
                    if (grp * 8 + 2 * ks >= NBT || grp * 8 + 2 * ks >= nbt) continue;
                    const int chl = 4 * ks + g, chg = grp * 16 + chl;                       // chunk inside the tile / the operand
                    uint4 fd[RT];
#pragma unroll
                    for (int j = 0; j < RT; ++j) {
                        fd[j] = *reinterpret_cast<const uint4*>(bt + tn_off_bf16(j * 16 + li, chl));
                        if (chg >= cb) fd[j] = make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int x = 0; x < NA; ++x) {
                        if (x >= na) continue;
                        const uint4 fw = *reinterpret_cast<const uint4*>(smem + woff + (x * 16 + li) * wstride + chg * 16);
#pragma unroll
                        for (int j = 0; j < RT; ++j)
                            accd[x][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fw), __builtin_bit_cast(bf16x8_t, fd[j]),
                                                                               accd[x][j], 0, 0, 0);
                    }
                }
            }
        }
        if constexpr (DG) {
            // dx rows of this step: lane (row li of tile j, input channels 16 x + 4 g .. + 3); rounded to bf16 as the NT kernel rounds
            // (through its LDS patch) before the skip connection's gradient is added, then rounded again
#pragma unroll
            for (int j = 0; j < RT; ++j) {
                const int m = m0 + j * 16 + li;
#pragma unroll
                for (int x = 0; x < NA; ++x) {
                    const int ci = x * 16 + 4 * g;
                    if (x < na && m < mend && ci < Na) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = round_to<bf16>(accd[x][j][r]);
                        if constexpr (DG == 2) {
                            const uint2 rq = *reinterpret_cast<const uint2*>(res + (long)m * Na + ci);
                            v[0] = round_to<bf16>(v[0] + __uint_as_float(rq.x << 16)); v[1] = round_to<bf16>(v[1] + __uint_as_float(rq.x & 0xffff0000u));
                            v[2] = round_to<bf16>(v[2] + __uint_as_float(rq.y << 16)); v[3] = round_to<bf16>(v[3] + __uint_as_float(rq.y & 0xffff0000u));
                        }
                        *reinterpret_cast<uint2*>(dx + (long)m * Na + ci) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                }
            }
        }
        wave_sync();                                    // reads done before the next step's stores
    }

    // ---- add the four waves' accumulators in LDS in wave order, write one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);        // [NA*NBT][256] floats <= 27 KB
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
                for (int y = 0; y < NBT; ++y) {
                    if (x >= na || y >= nbt) continue;
                    float* r = red + ((x * NBT + y) * 64 + lane) * 4;
                    float4 cur = w == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(r);
                    cur.x += acc[x][y][0]; cur.y += acc[x][y][1]; cur.z += acc[x][y][2]; cur.w += acc[x][y][3];
                    *reinterpret_cast<float4*>(r) = cur;
                }
        }
        __syncthreads();
    }
    // D layout: col (lane & 15) = wide index within the tile, rows 4*(lane>>4)+r = narrow index
    float* o = ws + (long)blockIdx.x * Na * Nb;
    for (int e = t; e < na * nbt * 256; e += DFD_THREADS) {
        const int tile = e >> 8, l = (e >> 2) & 63, r = e & 3;
        const int x = tile / nbt, y = tile - x * nbt;
        const int ia = x * 16 + (l >> 4) * 4 + r, jb = y * 16 + (l & 15);
        if (ia < Na && jb < Nb) {
            const float v = red[((x * NBT + y) * 64 + l) * 4 + r];
            if (swap) o[(long)jb * Na + ia] = v; else o[(long)ia * Nb + jb] = v;
        }
    }
}

// ===========================================================================
// host dispatch
// ===========================================================================
static bool tnw_pro_ok(int mode_narrow, int mode_wide) {
    return (mode_narrow == DFD_PRO_NONE || mode_narrow == DFD_PRO_AFFINE2) &&
           (mode_wide == DFD_PRO_NONE || mode_wide == DFD_PRO_AFFINE2 || mode_wide == DFD_PRO_BN_ACT_GATE ||
            mode_wide == DFD_PRO_BN_ACT);
}

template <int ROWS, int NA, int NBT>
static int tnw_launch(const void* a, const dfd_prologue* pro_a, int Na, const void* b, const dfd_prologue* pro_b, int Nb,
                      int M, int swap, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    const ProArgs pa = pro_args(pro_a), pb = pro_args(pro_b);
    const int ma = pro_a ? pro_a->mode : DFD_PRO_NONE, mb = pro_b ? pro_b->mode : DFD_PRO_NONE;
    const int act = ((mb == DFD_PRO_BN_ACT_GATE || mb == DFD_PRO_BN_ACT) && pro_b) ? pro_b->act : DFD_ACT_NONE;
    // 2 workgroups per CU (64 KB of tiles each); every wave should see several 32-row steps
    int nblocks = 512;
    int rpb = (M + nblocks - 1) / nblocks;
    rpb = (rpb + 4 * ROWS - 1) / (4 * ROWS) * (4 * ROWS);
    nblocks = (M + rpb - 1) / rpb;
    if ((size_t)(nblocks + nblocks / 32 + 2) * Na * Nb * 4 > ws_bytes) return DFD_EUNSUPPORTED;
    const bool gated = mb == DFD_PRO_BN_ACT_GATE;
    if (gated && pb.HW < ROWS) return DFD_EUNSUPPORTED;           // a step must not span more than two images (gate cache)
    const int lds = 4 * 2 * ROWS * 256 + 3 * (Na + Nb) * 4 + (gated ? 4 * 2 * Nb * 4 : 0);
#define LAUNCH_TNW(PAV, PBV, ACTV)                                                                                     \
    hipLaunchKernelGGL((k_pw_tnw<ROWS, NA, NBT, PAV, PBV, ACTV>), dim3(nblocks), dim3(DFD_THREADS), lds, st, (const bf16*)a, pa, \
                       Na, (const bf16*)b, pb, Nb, M, rpb, swap, ws)
    if (mb == DFD_PRO_BN_ACT_GATE) {
        if (act != DFD_ACT_SILU) return DFD_EUNSUPPORTED;
        if (ma == DFD_PRO_AFFINE2) LAUNCH_TNW(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT_GATE, DFD_ACT_SILU);
        else LAUNCH_TNW(DFD_PRO_NONE, DFD_PRO_BN_ACT_GATE, DFD_ACT_SILU);
    } else if (mb == DFD_PRO_BN_ACT) {
        if (ma != DFD_PRO_AFFINE2) return DFD_EUNSUPPORTED;          // the engine's use: dW = (BN-backward of g) x act(BN(y))
        if (act == DFD_ACT_GELU) LAUNCH_TNW(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT, DFD_ACT_GELU);
        else if (act == DFD_ACT_SILU) LAUNCH_TNW(DFD_PRO_AFFINE2, DFD_PRO_BN_ACT, DFD_ACT_SILU);
        else return DFD_EUNSUPPORTED;
    } else if (mb == DFD_PRO_AFFINE2) {
        if (ma == DFD_PRO_AFFINE2) return DFD_EUNSUPPORTED;
        LAUNCH_TNW(DFD_PRO_NONE, DFD_PRO_AFFINE2, DFD_ACT_NONE);
    } else {
        if (ma == DFD_PRO_AFFINE2) LAUNCH_TNW(DFD_PRO_AFFINE2, DFD_PRO_NONE, DFD_ACT_NONE);
        else LAUNCH_TNW(DFD_PRO_NONE, DFD_PRO_NONE, DFD_ACT_NONE);
    }
#undef LAUNCH_TNW
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, nblocks, (long)Na * Nb, dw, accumulate, st);
}

// Expand layer backward in one pass (kernel comment, DG): x [M][Cin] narrow and plain, (dz, y) [M][Cm] wide with the BN-backward map;
// dw [Cm][Cin] (the layer's weight layout), dx [M][Cin].  DFD_EUNSUPPORTED: shapes the wave-autonomous kernel does not hold.
template <int ROWS, int NA, int NBT>
static int tnw_dg_launch(const void* x, int Cin, const void* dz, const dfd_prologue* pro_d, int Cm, const void* w_kn, const void* residual,
                         int M, void* dx, float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    const ProArgs pa = pro_args(nullptr), pb = pro_args(pro_d);
    int nblocks = NBT > 9 ? 256 : 512;                          // one / two workgroups per CU
    int rpb = (M + nblocks - 1) / nblocks;
    rpb = (rpb + 4 * ROWS - 1) / (4 * ROWS) * (4 * ROWS);
    nblocks = (M + rpb - 1) / rpb;
    if ((size_t)(nblocks + nblocks / 32 + 2) * Cin * Cm * 4 > ws_bytes) return DFD_EUNSUPPORTED;
    const int wstride = (Cm + 31) / 32 * 64 + 16;
    int woff = 4 * 2 * ROWS * 256 + 3 * (Cin + Cm) * 4;
    woff = (woff + 15) / 16 * 16;
    const int lds = woff + NA * 16 * wstride;
    if (lds > 80 * 1024) return DFD_EUNSUPPORTED;               // two workgroups per CU
    struct TnwDgTag;
    dfd_allow_lds_once<TnwDgTag>(k_pw_tnw<ROWS, NA, NBT, DFD_PRO_NONE, DFD_PRO_AFFINE2, DFD_ACT_NONE, 2>, 80 * 1024);
    dfd_allow_lds_once<TnwDgTag>(k_pw_tnw<ROWS, NA, NBT, DFD_PRO_NONE, DFD_PRO_AFFINE2, DFD_ACT_NONE, 1>, 80 * 1024);
    if (residual)
        hipLaunchKernelGGL((k_pw_tnw<ROWS, NA, NBT, DFD_PRO_NONE, DFD_PRO_AFFINE2, DFD_ACT_NONE, 2>), dim3(nblocks), dim3(DFD_THREADS), lds, st,
                           (const bf16*)x, pa, Cin, (const bf16*)dz, pb, Cm, M, rpb, 1, ws, (const bf16*)w_kn, (const bf16*)residual, (bf16*)dx,
                           wstride, woff);
    else
        hipLaunchKernelGGL((k_pw_tnw<ROWS, NA, NBT, DFD_PRO_NONE, DFD_PRO_AFFINE2, DFD_ACT_NONE, 1>), dim3(nblocks), dim3(DFD_THREADS), lds, st,
                           (const bf16*)x, pa, Cin, (const bf16*)dz, pb, Cm, M, rpb, 1, ws, (const bf16*)w_kn, (const bf16*)nullptr, (bf16*)dx,
                           wstride, woff);
    if (hipGetLastError() != hipSuccess) return DFD_ELAUNCH;
    return dfd_launch_sum_partials(ws, nblocks, (long)Cin * Cm, dw, accumulate, st);
}
extern "C" int dfd_pwconv_bwd_fused(int dtype, const void* dz, const void* y, const float* coef, const void* x, const void* w_kn,
                                    const void* residual, int M, int Cm, int Cin, void* dx, float* dw, int accumulate, float* ws,
                                    size_t ws_bytes, dfd_stream stream) {
    if (!dz || !y || !coef || !x || !w_kn || !dx || !dw || !ws || M < 1 || Cm < 8 || Cin < 8 || Cm % 8 || Cin % 8) return DFD_EINVAL;
    if (dtype != DFD_BF16 || M < 2048 * 32 * 3 || Cin > 48 || Cin > Cm || !(Cm <= 128 || (Cm <= 240 && Cm % 16 == 0))) return DFD_EUNSUPPORTED;
    if (Cin > 32 && Cm <= 144) return DFD_EUNSUPPORTED;           // (no such layer: not instantiated)
    dfd_prologue pro{};
    pro.mode = DFD_PRO_AFFINE2; pro.act = DFD_ACT_NONE; pro.HW = 1; pro.a2 = y; pro.coef = coef; pro.gate = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (Cm <= 96) return tnw_dg_launch<32, 2, 6>(x, Cin, dz, &pro, Cm, w_kn, residual, M, dx, dw, accumulate, ws, ws_bytes, st);
    if (Cm <= 144) return tnw_dg_launch<16, 2, 9>(x, Cin, dz, &pro, Cm, w_kn, residual, M, dx, dw, accumulate, ws, ws_bytes, st);
    if (Cm <= 192) return tnw_dg_launch<16, 3, 12>(x, Cin, dz, &pro, Cm, w_kn, residual, M, dx, dw, accumulate, ws, ws_bytes, st);
    return tnw_dg_launch<16, 3, 15>(x, Cin, dz, &pro, Cm, w_kn, residual, M, dx, dw, accumulate, ws, ws_bytes, st);
}

int dfd_pw_tnw(const void* p, const dfd_prologue* pro_p, int Ni, const void* q, const dfd_prologue* pro_q, int Nj, int M,
               float* dw, int accumulate, float* ws, size_t ws_bytes, hipStream_t st) {
    // large M only: each of 2048 waves needs a few 32-row steps for the pipeline to matter
    if (M < 2048 * 32 * 3) return DFD_EUNSUPPORTED;
    const int mp = pro_p ? pro_p->mode : DFD_PRO_NONE, mq = pro_q ? pro_q->mode : DFD_PRO_NONE;
    // narrow operand <= 48 channels, wide one <= 144; dw is [Ni][Nj]
    const auto wide_ok = [](int n) { return n <= 128 || (n <= 144 && n % 16 == 0); };
#define TNW_PICK(NARROW, WIDE, ...)                                             \
    do {                                                                       \
        if ((NARROW) <= 32 && (WIDE) <= 96) return tnw_launch<32, 2, 6>(__VA_ARGS__); \
        if ((NARROW) <= 32) return tnw_launch<16, 2, 9>(__VA_ARGS__);               \
        return DFD_EUNSUPPORTED;    /* 3 x 9 accumulator tiles spill at 2 waves per SIMD: tiled kernel */ \
    } while (0)
    if (Ni <= 48 && wide_ok(Nj) && Ni <= Nj && tnw_pro_ok(mp, mq))
        TNW_PICK(Ni, Nj, p, pro_p, Ni, q, pro_q, Nj, M, 0, dw, accumulate, ws, ws_bytes, st);
    if (Nj <= 48 && wide_ok(Ni) && tnw_pro_ok(mq, mp))
        TNW_PICK(Nj, Ni, q, pro_q, Nj, p, pro_p, Ni, M, 1, dw, accumulate, ws, ws_bytes, st);
#undef TNW_PICK
    return DFD_EUNSUPPORTED;
}
