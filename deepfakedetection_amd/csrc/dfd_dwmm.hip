// dfd_dwmm.hip — depthwise k x k convolution, forward, on the matrix cores (bf16; design notes in dfd_dwm.h).
//
// One 512-thread workgroup per CU, persistent over (image block, spatial tile) work items of ONE 64-channel chunk (4 groups of
// 16 channels; the last chunk of a layer may hold 1-3).  The stages of an item:
//     L  global loads of the item's input window into registers         (in flight for a whole iteration)
//     A  BN + activation of the producer on those registers -> LDS image (VALU; planar by group, zero outside the picture)
//     T  taps: per run of 16 output pixels (K*K + 1) / 2 MFMAs, each fed by ONE ds_read_b128; the group's diagonal weight
//        fragments stay in registers for the whole kernel; results rounded to bf16, counted into the BatchNorm sums, parked in LDS
//     S  output tile -> global as 16-byte pieces (up to 128 contiguous bytes per pixel)
// are software-pipelined over the items with ONE barrier per item — iteration i runs S(i-1), T(i), A(i+1), L(i+2); images and
// output tiles are double-buffered — and the two halves of the workgroup take {S, T} and {A, L} in opposite order, so that on every
// SIMD one wave is in its matrix phase while its partner is in its vector phase (MI355X_MICROARCH.md, "two waves per SIMD", item 9).
// S sits in front of T because vmcnt counts loads and stores together: the wait in front of A also waits for every store issued
// before it, and with the taps in between those have long landed.
//
// What the first two versions of this kernel taught: with stage -> barrier -> taps -> barrier -> store per tile the phases
// add up (loads 95 + activation 23 + taps 70 + stores 60 = 248 of 256 us on block 0) — nothing overlaps at 3 workgroups per CU;
// and a pipelined loop is instruction-issue bound the moment its stages are full of per-element branches (an EMPTY iteration cost
// 2.8 us).  So: everything a thread needs per stage (which window element it loads, where that goes in LDS, which output
// pixels it stores, the run bases of its wave) is fixed by the tile geometry and computed ONCE; the stages are straight-line code
// with lane masks (out-of-picture loads read a valid dummy address and are masked to zero), uniform slot counts, and a per-item
// `clean` flag (the tile lies inside the picture) that drops the bounds arithmetic of interior tiles.
#include "dfd_dwm.h"
#include <climits>

typedef float dwm_f4 __attribute__((ext_vector_type(4)));
typedef __bf16 dwm_bf8 __attribute__((ext_vector_type(8)));

static int g_tune[DFD_TUNE_COUNT] = {/*DW_MFMA*/ 1, /*DW_LDS_KB*/ 156, /*DW_GRID*/ 256, /*DEBUG*/ 0, /*PW_NTD*/ 1, /*NTD_NS*/ 0, /*NTD_MAXN*/ 0, /*NTD_MINT*/ 16,
                                      /*DWQ_GRID_FWD*/ 1024, /*DWQ_GRID_BWD*/ 1024, /*DWQ_GRID_WGRAD*/ 1024, /*DWQ_GRID_MIN*/ 32, /*TN_WGS*/ 512, /*DWQ_WIDE*/ -1};
int dfd_tune_get(int key) { return (key >= 0 && key < DFD_TUNE_COUNT) ? g_tune[key] : 0; }
extern "C" int dfd_tune(int key, int value) {
    if (key < 0 || key >= DFD_TUNE_COUNT) return DFD_EINVAL;
    g_tune[key] = value;
    return DFD_OK;
}

#define DWP_THREADS 512
#define DWP_MAXV 6          // staged 16-byte vectors per thread and item
#define DWP_MAXS 4          // stored 16-byte vectors per thread and item
#define DWP_MAXR 8         // runs per wave and item

template <int K, int S, int ACT, bool PRO, bool STATS, bool WHOLE>
__global__ void __launch_bounds__(DWP_THREADS, 2)
k_dw_fwd_mp(const unsigned short* __restrict__ x, const float* __restrict__ bnstate, const float* __restrict__ w,
            unsigned short* __restrict__ y, DwMGeom g, float* __restrict__ partials, int dbg) {
    constexpr int KK = K * K, NPAIR = (KK + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* image0 = smem;
    unsigned char* out0 = smem + 2 * g.in_bytes;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int n16 = lane & 15, kg = lane >> 4;
    int bx, by;
    dwq_block(bx, by, g.remap);
    const int c0 = bx * 64;
    const int left = (g.C >> 4) - 4 * bx;
    const int G = left < 4 ? left : 4;                        // 16-channel groups of this chunk
    const int nvl = G == 1 ? 1 : (G == 2 ? 2 : 3);            // log2 of the 16-byte vectors per pixel that are staged
    const int v = t & ((1 << nvl) - 1);
    const bool cvalid = c0 + v * 8 < g.C;
    const int rsl = G == 1 ? 3 : (G == 2 ? 2 : 1);            // the runs of a group are shared by 2^rsl of the 8 waves
    const int mg = wave >> rsl, rs = wave & ((1 << rsl) - 1), RS = 1 << rsl;
    const bool mact = mg < G;
    const bool first_half = wave < 4;

    f2 sc[4], sh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = (PRO && cvalid) ? *reinterpret_cast<const f2*>(bnstate + c0 + v * 8 + 2 * j) : (f2){1.f, 1.f};
        sh[j] = (PRO && cvalid) ? *reinterpret_cast<const f2*>(bnstate + g.C + c0 + v * 8 + 2 * j) : (f2){0.f, 0.f};
    }
    uint4 wf[NPAIR];
    {
        const float* wc = w + (long)(c0 + mg * 16 + n16) * KK;
        dwm_weight_frags<NPAIR, KK>(wf, lane, mact, [&](int tt) { return wc[tt]; });
    }
    int po[NPAIR];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        int tt = 2 * pr + (kg >> 1);
        if (tt > KK - 1) tt = KK - 1;
        const int i = tt / K, j = tt - i * K;
        po[pr] = (dwm_tapoff<S>(g, i, j) << 5) + ((kg & 1) << 4);
    }
    // ---- staging slots of this thread: window element -> (global offset, LDS offset, position for the bounds check).  The last slot
    // of threads past the end duplicates an earlier element (same value to the same place: harmless), so slot counts are uniform.
    const int npx = WHOLE ? g.NI * g.H * g.W : g.IHW;
    const int stotal = npx << nvl;
    const int nslots = (stotal + DWP_THREADS - 1) / DWP_THREADS;
    const int vofs = (((v >> 1) * g.plane) << 5) + ((v & 1) << 4);
    int goff[DWP_MAXV], loff[DWP_MAXV], spos[DWP_MAXV];
#pragma unroll
    for (int u = 0; u < DWP_MAXV; ++u) {
        int idx = t + u * DWP_THREADS;
        if (idx >= stotal) idx = stotal > DWP_THREADS ? idx - stotal : idx % stotal;     // (stotal is a multiple of the vectors per pixel: v is kept)
        if (idx >= stotal) idx = v;
        const int pix = idx >> nvl;
        int img, iy, ix;
        if constexpr (WHOLE) {
            img = dwm_div(pix, g.hw_magic);
            const int rem = pix - img * g.H * g.W;
            const int gy = dwm_div(rem, g.w_magic), gx = rem - gy * g.W;
            iy = gy + g.pt; ix = gx + g.pl;
            goff[u] = ((img * g.H + gy) * g.W + gx) * g.C + v * 8;
            // the image slot (the element exists if n0 + slot < N); rows / columns no tap reaches and lanes past the channels never do
            spos[u] = (iy < g.IH && ix < g.IW && cvalid) ? img : (1 << 28);
            if (spos[u] != img) { iy = 0; ix = 0; }
        } else {
            img = 0;
            iy = dwm_div(pix, g.iw_magic);
            ix = pix - iy * g.IW;
            goff[u] = (iy * g.W + ix) * g.C + v * 8;                                       // relative to the window origin (may lie outside)
            spos[u] = cvalid ? ((iy << 16) | ix) : (0x4000 << 16);                         // (row 16384 is never inside a picture)
        }
        loff[u] = (dwm_lpix<S>(g, img, iy, ix) << 5) + vofs;
    }
    // ---- store slots: output pixel of the tile -> (global offset, position); threads past the end are masked (opos < 0)
    const int ototal = g.NPV << nvl;
    const int nstore = (ototal + DWP_THREADS - 1) / DWP_THREADS;
    int soff[DWP_MAXS], opos[DWP_MAXS], ooff[DWP_MAXS];
#pragma unroll
    for (int u = 0; u < DWP_MAXS; ++u) {
        const int idx = t + u * DWP_THREADS;
        const int p = idx < ototal ? (idx >> nvl) : 0;
        const int img = dwm_div(p, g.thw_magic);
        const int rem = p - img * g.THW;
        const int qy = dwm_div(rem, g.tw_magic), qx = rem - qy * g.TW;
        soff[u] = ((img * g.Ho + qy) * g.Wo + qx) * g.C + v * 8;
        ooff[u] = p * g.opitch + v * 16;
        opos[u] = (idx < ototal && cvalid) ? ((img << 20) | (qy << 10) | qx) : (1 << 30);
    }
    // ---- runs of this wave
    int lb[DWP_MAXR], rpos[DWP_MAXR];
    const int nrw = mact ? (g.R - rs + RS - 1) >> rsl : 0;
#pragma unroll
    for (int j = 0; j < DWP_MAXR; ++j) {
        const int p = (rs + j * RS) * 16 + n16;
        const bool pv = p < g.NPV;
        const int pc = pv ? p : 0;
        const int img = dwm_div(pc, g.thw_magic);
        const int rem = pc - img * g.THW;
        const int qy = dwm_div(rem, g.tw_magic), qx = rem - qy * g.TW;
        lb[j] = (mg * g.plane + dwm_base<S>(g, img, qy, qx)) << 5;
        rpos[j] = pv ? ((img << 20) | (qy << 10) | qx) : (1 << 30);                      // (image slot 1024 never exists)
    }
    const int owave = mg * 32 + kg * 8 + n16 * g.opitch;      // this lane's 8 bytes of an output pixel, relative to its run
    if constexpr (WHOLE) {  // the zero border of both images, once; staging then only writes real pixels
        for (int i = t * 16; i < 2 * g.in_bytes; i += DWP_THREADS * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

    const int tiles = g.tiles_y * g.tiles_x;
    const int cnt = by < g.nwork ? (g.nwork - by + (int)gridDim.y - 1) / (int)gridDim.y : 0;
    uint4 raw[DWP_MAXV];
    unsigned inb_mask = 0;                                    // which of raw[] hold picture elements (of the item loaded last)

    // item origins advance by gridDim.y work items per step: mixed-radix increments instead of divisions
    struct Org { int ib, ty, tx; };
    Org dstep;
    {
        const int gyv = (int)gridDim.y;
        dstep.ib = gyv / tiles;
        const int r = gyv - dstep.ib * tiles;
        dstep.ty = r / g.tiles_x;
        dstep.tx = r - dstep.ty * g.tiles_x;
    }
    auto next = [&](Org o) {
        o.tx += dstep.tx;
        if (o.tx >= g.tiles_x) { o.tx -= g.tiles_x; ++o.ty; }
        o.ty += dstep.ty;
        if (o.ty >= g.tiles_y) { o.ty -= g.tiles_y; ++o.ib; }
        o.ib += dstep.ib;
        return o;
    };
    auto L = [&](Org o) {                                      // issue the loads of an item
        const int n0 = o.ib * g.NI, oy0 = o.ty * g.TH, ox0 = o.tx * g.TW;
        const int iy0 = oy0 * S - g.pt, ix0 = ox0 * S - g.pl;
        const unsigned short* base = WHOLE ? x + (long)n0 * g.H * g.W * g.C + c0
                                           : x + (((long)n0 * g.H + iy0) * g.W + ix0) * g.C + c0;
        const long dummy = x - base;                           // out of the picture: a valid address (masked to zero in A)
        inb_mask = 0;
        if (dbg & 8) return;
        // straight-line code: integer masks and selects, no per-element branches (the first version of this loop compiled to 35
        // branches and 140 scalar-register spills for six loads)
#pragma unroll
        for (int u = 0; u < DWP_MAXV; ++u) {
            if (u >= nslots) break;
            unsigned ok;
            if constexpr (WHOLE) ok = (unsigned)(n0 + spos[u]) < (unsigned)g.N ? 1u : 0u;
            else ok = ((unsigned)(iy0 + (spos[u] >> 16)) < (unsigned)g.H ? 1u : 0u) & ((unsigned)(ix0 + (spos[u] & 0xffff)) < (unsigned)g.W ? 1u : 0u);
            const long off = ok ? (long)goff[u] : dummy;
            raw[u] = *reinterpret_cast<const uint4*>(base + off);
            inb_mask |= ok << u;
        }
    };
    auto A = [&](int par) {                                    // registers -> activated LDS image
        unsigned char* img = image0 + par * g.in_bytes;
#pragma unroll
        for (int u = 0; u < DWP_MAXV; ++u) {
            if (u >= nslots) break;
            uint4 q = raw[u];
            if (PRO && !(dbg & 1)) {
                f2 a[4];
                unpack2(q, a);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f2 z = __builtin_elementwise_fma(sc[j], a[j], sh[j]);
                    if constexpr (ACT == DFD_ACT_SILU) {
                        const f2 e = (f2){__expf(-z.x), __expf(-z.y)};
                        const f2 d = e + (f2){1.f, 1.f};
                        a[j] = z * (f2){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                    } else {
                        a[j] = act_fwd2<ACT>(z);
                    }
                }
                q = pack2(a);
            }
            const unsigned m = 0u - ((inb_mask >> u) & 1u);     // all ones inside the picture
            if constexpr (WHOLE) {
                if (m) *reinterpret_cast<uint4*>(img + loff[u]) = q;                       // the border keeps its zeros
            } else {
                *reinterpret_cast<uint4*>(img + loff[u]) = make_uint4(q.x & m, q.y & m, q.z & m, q.w & m);
            }
        }
    };
    auto T = [&](Org o, int par) {                             // taps of an item
        if (dbg & 2) return;
        const int n0 = o.ib * g.NI, oy0 = o.ty * g.TH, ox0 = o.tx * g.TW;
        const unsigned char* img = image0 + par * g.in_bytes;
        unsigned char* ot = out0 + par * g.out_bytes + owave;
#pragma unroll
        for (int j = 0; j < DWP_MAXR; ++j) {
            if (j >= nrw) break;
            const unsigned char* a0 = img + lb[j];
            dwm_f4 acc = (dwm_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const uint4 b = *reinterpret_cast<const uint4*>(a0 + po[pr]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dwm_bf8, wf[pr]), __builtin_bit_cast(dwm_bf8, b), acc, 0, 0, 0);
            }
            const unsigned q0 = pack_bf2(acc[0], acc[1]), q1 = pack_bf2(acc[2], acc[3]);
            if constexpr (STATS) {
                const int rp = rpos[j];
                const unsigned ov = ((unsigned)(n0 + (rp >> 20)) < (unsigned)g.N ? 1u : 0u) & (oy0 + ((rp >> 10) & 1023) < g.Ho ? 1u : 0u) &
                                    (ox0 + (rp & 1023) < g.Wo ? 1u : 0u);
                const float m = ov ? 1.f : 0.f;
                const float r0 = __uint_as_float(q0 << 16) * m, r1 = __uint_as_float(q0 & 0xffff0000u) * m;
                const float r2 = __uint_as_float(q1 << 16) * m, r3 = __uint_as_float(q1 & 0xffff0000u) * m;
                s1[0] += r0; s1[1] += r1; s1[2] += r2; s1[3] += r3;
                s2[0] = fmaf(r0, r0, s2[0]); s2[1] = fmaf(r1, r1, s2[1]); s2[2] = fmaf(r2, r2, s2[2]); s2[3] = fmaf(r3, r3, s2[3]);
            }
            *reinterpret_cast<uint2*>(ot + ((rs + j * RS) << 4) * g.opitch) = make_uint2(q0, q1);
        }
    };
    auto St = [&](Org o, int par) {                            // output tile of an item -> global
        if (dbg & 4) return;
        const int n0 = o.ib * g.NI, oy0 = o.ty * g.TH, ox0 = o.tx * g.TW;
        const unsigned char* ot = out0 + par * g.out_bytes;
        unsigned short* base = y + (((long)n0 * g.Ho + oy0) * g.Wo + ox0) * g.C + c0;
#pragma unroll
        for (int u = 0; u < DWP_MAXS; ++u) {
            if (u >= nstore) break;
            const int op = opos[u];
            const unsigned ok = ((unsigned)(n0 + (op >> 20)) < (unsigned)g.N ? 1u : 0u) & (oy0 + ((op >> 10) & 1023) < g.Ho ? 1u : 0u) &
                                (ox0 + (op & 1023) < g.Wo ? 1u : 0u);
            if (ok) *reinterpret_cast<uint4*>(base + soff[u]) = *reinterpret_cast<const uint4*>(ot + ooff[u]);
        }
    };

    if (cnt > 0) {
        Org o_prev, o_cur, o_n2;
        o_cur.ib = by / tiles;
        {
            const int r = by - o_cur.ib * tiles;
            o_cur.ty = r / g.tiles_x;
            o_cur.tx = r - o_cur.ty * g.tiles_x;
        }
        o_prev = o_cur;
        Org o_n1 = next(o_cur);
        o_n2 = next(o_n1);
        L(o_cur);
        A(0);
        if (cnt > 1) L(o_n1);
        int par = 0;
        for (int i = 0; i <= cnt; ++i) {
            __syncthreads();
            // two phases; the halves of the workgroup take them in opposite order (one copy of each stage in the code)
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                if ((ph == 0) == first_half) {
                    if (i >= 1) St(o_prev, par ^ 1);
                    if (i < cnt) T(o_cur, par);
                } else {
                    if (i + 1 < cnt) A(par ^ 1);
                    if (i + 2 < cnt) L(o_n2);
                }
            }
            o_prev = o_cur; o_cur = o_n1; o_n1 = o_n2; o_n2 = next(o_n2);
            par ^= 1;
        }
    }
    if constexpr (STATS) {
        // lanes of one 16-lane row hold the same 4 channels for different pixels: butterfly over the row (fixed order), then the
        // waves that shared a group are added in wave order
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                s1[i] += __shfl_xor(s1[i], m, 64);
                s2[i] += __shfl_xor(s2[i], m, 64);
            }
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);          // [wave][2][16]
        if (n16 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                red[wave * 32 + kg * 4 + i] = s1[i];
                red[wave * 32 + 16 + kg * 4 + i] = s2[i];
            }
        }
        __syncthreads();
        if (t < 64 * 2) {                                     // thread = (which, group, channel)
            const int which = t >> 6, gq = (t >> 4) & 3, ch = t & 15;
            if (gq < G) {
                float s = 0.f;
                for (int k = 0; k < RS; ++k) s += red[((gq << rsl) + k) * 32 + which * 16 + ch];
                partials[(long)by * 2 * g.C + which * g.C + c0 + gq * 16 + ch] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// host: tile geometry
// ---------------------------------------------------------------------------
static unsigned magic20(int d) { return ((1u << 20) + (unsigned)d - 1u) / (unsigned)d; }
static bool magic_ok(unsigned magic, int d, int limit) {
    for (unsigned v = 0; v < (unsigned)limit; ++v)
        if (((v * magic) >> 20) != v / (unsigned)d) return false;
    return true;
}

// pro: the staging phase applies an activation (its cost per staged vector); extra_lds: bytes beside the tiles
bool dfd_dwm_geom(const dfd_dwconv_shape* s, bool centre_is_input, bool pro, int npair, int extra_lds, DwMGeom* g) {
    if (!s || s->N <= 0 || s->H <= 0 || s->W <= 0 || s->Ho <= 0 || s->Wo <= 0 || s->C <= 0 || s->C % 16) return false;
    if (!(s->k == 3 || s->k == 5) || !(s->stride == 1 || s->stride == 2)) return false;
    if (s->pad_top < 0 || s->pad_left < 0 || s->pad_top >= s->k || s->pad_left >= s->k) return false;
    if ((s->Ho - 1) * s->stride - s->pad_top > s->H - 1 || (s->Wo - 1) * s->stride - s->pad_left > s->W - 1) return false;
    if (centre_is_input) return false;                       // (the data gradient has its own planner)
    const int K = s->k, S = s->stride;
    g->N = s->N; g->H = s->H; g->W = s->W; g->C = s->C; g->Ho = s->Ho; g->Wo = s->Wo; g->pt = s->pad_top; g->pl = s->pad_left;
    const long budget = (long)dfd_tune_get(DFD_TUNE_DW_LDS_KB) * 1024 - extra_lds;
    const int CH = s->Ho, CW = s->Wo;
    const int gmax = s->C >= 64 ? 4 : s->C / 16;                       // groups of the widest chunk
    const int nv = gmax == 1 ? 2 : (gmax == 2 ? 4 : 8);                // 16-byte vectors per pixel
    const int rsplit = gmax == 1 ? 8 : (gmax == 2 ? 4 : 2);            // waves that share a group's runs
    const int opitch = gmax * 32 + 16;                                 // 16-byte aligned, not a multiple of 128
    auto ext = [&](int centre) { return (centre - 1) * S + K; };
    auto pitch = [&](int IW) { return S == 1 ? IW : (IW + 1) / 2; };
    auto plane_of = [&](int ni, int IH, int IW) {
        int px = ni * (S == 1 ? IH : 2 * IH) * pitch(IW);
        while ((px & 3) != 1) ++px;
        return px;
    };
    double best = 1e300;
    int bTH = 0, bTW = 0, bNI = 1;
    for (int TW = 1; TW <= CW; ++TW) {
        if (TW != CW && TW % 4) continue;                    // full width, or multiples of 4
        for (int TH = 1; TH <= CH; ++TH) {
            const int IH = ext(TH), IW = ext(TW);
            const bool whole = TH == CH && TW == CW;
            const int nimax = whole ? 16 : 1;
            for (int NI = 1; NI <= nimax && NI <= s->N; ++NI) {
                const long npv = (long)NI * TH * TW;
                const long staged = whole ? (long)NI * s->H * s->W : (long)IH * IW;
                if (npv > 1000 || (long)NI * IH * IW >= DWM_MAXPIX || staged >= DWM_MAXPIX || TH >= 1024 || TW >= 1024) break;
                if (staged * nv > (long)DWP_THREADS * DWP_MAXV || npv * nv > (long)DWP_THREADS * DWP_MAXS) break;
                const int R = (int)((npv + 15) / 16);
                if ((R + rsplit - 1) / rsplit > DWP_MAXR) break;
                const long lds = 2 * ((long)gmax * plane_of(NI, IH, IW) * 32 + (long)R * 16 * opitch);
                if (lds > budget) break;
                const long tiles = (long)((CH + TH - 1) / TH) * ((CW + TW - 1) / TW) * ((s->N + NI - 1) / NI);
                // cycles of one item on a CU: the three pipes run beside each other, the slowest one sets the pace
                const double stage = (double)((staged * nv + DWP_THREADS - 1) / DWP_THREADS) * (pro ? 330.0 : 90.0) * 2.0;
                const double taps = (double)((R + rsplit - 1) / rsplit) * (npair * 16.0 + 110.0) * 2.0;
                const double hbm = (double)(staged + npv) * nv * 16.0 / 9.0;
                double pace = stage + taps * 0.5;            // (the taps' own vector instructions compete with the staging)
                if (taps > pace) pace = taps;
                if (hbm > pace) pace = hbm;
                const double cost = (double)tiles * (pace + 1500.0);
                if (cost < best) { best = cost; bTH = TH; bTW = TW; bNI = NI; }
            }
        }
    }
    if (!bTH) return false;
    g->NI = bNI; g->TH = bTH; g->TW = bTW; g->THW = bTH * bTW; g->NPV = bNI * g->THW; g->R = (g->NPV + 15) / 16;
    g->IH = ext(bTH); g->IW = ext(bTW); g->IHW = g->IH * g->IW;
    g->P = pitch(g->IW);
    g->IMGP = (S == 1 ? g->IH : 2 * g->IH) * g->P;
    g->plane = plane_of(bNI, g->IH, g->IW);
    g->opitch = opitch;
    g->tiles_y = (CH + bTH - 1) / bTH; g->tiles_x = (CW + bTW - 1) / bTW;
    const long nwork = (long)g->tiles_y * g->tiles_x * ((s->N + bNI - 1) / bNI);
    if (nwork > INT_MAX / 2) return false;
    g->nwork = (int)nwork;
    g->whole = (bTH == CH && bTW == CW) ? 1 : 0;
    g->tw_magic = magic20(g->TW); g->thw_magic = magic20(g->THW); g->iw_magic = magic20(g->IW); g->ihw_magic = magic20(g->IHW);
    g->w_magic = magic20(g->W); g->hw_magic = magic20(g->H * g->W);
    if (!magic_ok(g->tw_magic, g->TW, g->THW) || !magic_ok(g->thw_magic, g->THW, g->R * 16) || !magic_ok(g->iw_magic, g->IW, g->IHW) ||
        !magic_ok(g->ihw_magic, g->IHW, bNI * g->IHW))
        return false;
    if (g->whole && (!magic_ok(g->w_magic, g->W, g->H * g->W) || !magic_ok(g->hw_magic, g->H * g->W, bNI * g->H * g->W))) return false;
    g->remap = 0;
    g->in_bytes = gmax * g->plane * 32;
    g->out_bytes = g->R * 16 * opitch;
    return true;
}

// DFD_EUNSUPPORTED: not this kernel's case (the caller runs the vector-unit kernel)
int dfd_dw_fwd_mm(const void* x, const float* in_bnstate, int in_act, const float* w, void* y, const dfd_dwconv_shape* s,
                  float* partials, int pcap, int* nparts, hipStream_t st) {
    const int mode = dfd_tune_get(DFD_TUNE_DW_MFMA);
    if (!(mode & 1)) return DFD_EUNSUPPORTED;
    DwMGeom g;
    const bool pro = in_bnstate != nullptr;
    const int npair = (s->k * s->k + 1) / 2;
    if (!dfd_dwm_geom(s, false, pro && in_act != DFD_ACT_NONE, npair, 0, &g)) return DFD_EUNSUPPORTED;
    // Measured per EfficientNet-B0 layer at batch 256 (scripts/dw_ab.py): this kernel wins where a tile is several whole
    // pictures and the tap loop is long (5x5 on 7x7 maps: 44 against 58 us) and loses on the large maps (block 1: 435 against 292 us),
    // where its per-item instruction overhead — not the taps — sets the pace.  Bit 3 of the knob serves every shape (tests, A/B runs).
    if (!(mode & 8) && !(g.whole && s->k == 5 && s->stride == 1 && s->H * s->W <= 64)) return DFD_EUNSUPPORTED;
    const int nchunks = (s->C + 63) / 64;
    g.remap = (nchunks > 1 && g.tiles_y * g.tiles_x >= 2 && (s->C * 2) % 128 != 0) ? 1 : 0;
    const bool stats = partials != nullptr;
    const int cap = stats ? (pcap < DFD_MAX_PARTIALS ? pcap : DFD_MAX_PARTIALS) : DFD_MAX_PARTIALS;
    int gy = dfd_tune_get(DFD_TUNE_DW_GRID) / nchunks;
    if (gy < 1) gy = 1;
    if (gy > cap) gy = cap;
    if (gy > g.nwork) gy = g.nwork;
    if (stats) *nparts = gy;
    size_t lds = 2 * ((size_t)g.in_bytes + g.out_bytes);
    if (lds < 1024) lds = 1024;
    dim3 grid(nchunks, gy);
#define LAUNCH_MM(PRO, STATS)                                                                                                     \
    do {                                                                                                                          \
        struct DwmTag;                                                                                                            \
        if (g.whole) {                                                                                                            \
            auto kern = k_dw_fwd_mp<K, S, ACT, PRO, STATS, true>;                                                                \
            dfd_allow_lds_once<DwmTag>(kern, 160 * 1024);                                                                         \
            hipLaunchKernelGGL(kern, grid, dim3(DWP_THREADS), lds, st, (const unsigned short*)x, in_bnstate, w, (unsigned short*)y, \
                               g, partials, dfd_tune_get(DFD_TUNE_DEBUG));                                                        \
        } else {                                                                                                                  \
            auto kern = k_dw_fwd_mp<K, S, ACT, PRO, STATS, false>;                                                               \
            dfd_allow_lds_once<DwmTag>(kern, 160 * 1024);                                                                         \
            hipLaunchKernelGGL(kern, grid, dim3(DWP_THREADS), lds, st, (const unsigned short*)x, in_bnstate, w, (unsigned short*)y, \
                               g, partials, dfd_tune_get(DFD_TUNE_DEBUG));                                                        \
        }                                                                                                                         \
    } while (0)
    DISPATCH_KS(s->k, s->stride, {
        if (pro) {
            DISPATCH_ACT_DW(in_act, { if (stats) LAUNCH_MM(true, true); else LAUNCH_MM(true, false); });
        } else {
            constexpr int ACT = DFD_ACT_NONE;
            if (stats) LAUNCH_MM(false, true); else LAUNCH_MM(false, false);
        }
    });
#undef LAUNCH_MM
    return DFD_CHECK_LAUNCH();
}

// diagnostics / tests: the tile plan of the forward kernel for a shape
extern "C" int dfd_dw_mm_plan(const dfd_dwconv_shape* s, int pro, int* out) {
    DwMGeom g;
    if (!s || !out) return DFD_EINVAL;
    if (!dfd_dwm_geom(s, false, pro != 0, (s->k * s->k + 1) / 2, 0, &g)) return DFD_EUNSUPPORTED;
    const int v[12] = {g.NI, g.TH, g.TW, g.R, g.IH, g.IW, g.P, g.plane, g.nwork, g.whole, g.in_bytes, g.out_bytes};
    for (int i = 0; i < 12; ++i) out[i] = v[i];
    return DFD_OK;
}
