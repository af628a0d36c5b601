// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3) on gfx950: which (lane, byte) of the A / B operand registers is
// which (row|col, k), which lane's scale byte applies to which elements, and where D lands.  One wave, exact data.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/mx_layout_probe.hip -o scripts/probes/mx_layout_probe && ./mx_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// out[probe][lane][4]
__global__ void k_probe(const unsigned char* A, const unsigned char* B, const int* SA, const int* SB, float* out, int nprobe) {
    const int lane = threadIdx.x;
    for (int p = 0; p < nprobe; ++p) {
        const int* a = reinterpret_cast<const int*>(A + ((size_t)p * 64 + lane) * 32);
        const int* b = reinterpret_cast<const int*>(B + ((size_t)p * 64 + lane) * 32);
        i32x8_t av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = a[i]; bv[i] = b[i]; }
        f32x4_t c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, SA[p * 64 + lane], 0, SB[p * 64 + lane]);
        for (int i = 0; i < 4; ++i) out[((size_t)p * 64 + lane) * 4 + i] = c[i];
    }
}

static const unsigned char ONE = 0x38;   // e4m3fn 1.0: exponent field 7, mantissa 0

int main() {
    // probe set 1: A one-hot at (lane La in {0,16,32,48} U {1}, byte) ; B all ones ; scales 1  -> row of A positions
    // probe set 2: A one-hot over row-0 candidates, B one-hot over col-0 candidates -> k match
    // probe set 3: scale probes
    std::vector<unsigned char> A, B;
    std::vector<int> SA, SB;
    struct Desc { int kind, x, y; };
    std::vector<Desc> desc;
    auto add = [&](int kind, int x, int y) {
        desc.push_back({kind, x, y});
        A.resize(A.size() + 64 * 32, 0); B.resize(B.size() + 64 * 32, 0);
        SA.resize(SA.size() + 64, 127); SB.resize(SB.size() + 64, 127);
        return desc.size() - 1;
    };
    // set 1: every (lane, byte) of A one-hot, B all ones
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 32; b += 1) {
        size_t p = add(1, l, b);
        A[(p * 64 + l) * 32 + b] = ONE;
        memset(&B[p * 64 * 32], ONE, 64 * 32);
    }
    // set 4: every (lane, byte) of B one-hot, A all ones
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 32; b += 1) {
        size_t p = add(4, l, b);
        B[(p * 64 + l) * 32 + b] = ONE;
        memset(&A[p * 64 * 32], ONE, 64 * 32);
    }
    // set 2: A one-hot at lanes {0,16,32,48} x 32 bytes; B one-hot at lanes {0,16,32,48} x 32 bytes
    for (int la = 0; la < 4; ++la) for (int ba = 0; ba < 32; ++ba)
        for (int lb = 0; lb < 4; ++lb) for (int bb = 0; bb < 32; ++bb) {
            size_t p = add(2, la * 32 + ba, lb * 32 + bb);
            A[(p * 64 + la * 16) * 32 + ba] = ONE;
            B[(p * 64 + lb * 16) * 32 + bb] = ONE;
        }
    // set 3: A one-hot at lanes {0,16,32,48} x 32 bytes (row 0 presumably), B all ones, scale_a of lane ls = 128 (x2)
    for (int la = 0; la < 4; ++la) for (int ba = 0; ba < 32; ba += 1)
        for (int ls = 0; ls < 64; ++ls) {
            size_t p = add(3, la * 32 + ba, ls);
            A[(p * 64 + la * 16) * 32 + ba] = ONE;
            memset(&B[p * 64 * 32], ONE, 64 * 32);
            SA[p * 64 + ls] = 128;
        }
    // set 5: same for scale_b with B one-hot
    for (int lb = 0; lb < 4; ++lb) for (int bb = 0; bb < 32; bb += 1)
        for (int ls = 0; ls < 64; ++ls) {
            size_t p = add(5, lb * 32 + bb, ls);
            B[(p * 64 + lb * 16) * 32 + bb] = ONE;
            memset(&A[p * 64 * 32], ONE, 64 * 32);
            SB[p * 64 + ls] = 128;
        }
    const int np = (int)desc.size();
    unsigned char *dA, *dB; int *dSA, *dSB; float* dO;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dSA, SA.size() * 4); hipMalloc(&dSB, SB.size() * 4);
    hipMalloc(&dO, (size_t)np * 64 * 4 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dSA, SA.data(), SA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dSB, SB.data(), SB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dO, np);
    std::vector<float> O((size_t)np * 64 * 4);
    hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    // D element (lane, reg) -> assume standard 16x16 layout: col = lane & 15, row = (lane >> 4) * 4 + reg; verify via sets 1/4
    auto nz = [&](int p, int& cnt, int& first_lane, int& first_reg, float& val) {
        cnt = 0; first_lane = first_reg = -1; val = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            float v = O[((size_t)p * 64 + l) * 4 + r];
            if (v != 0.f) { if (!cnt) { first_lane = l; first_reg = r; val = v; } ++cnt; }
        }
    };
    printf("# set1: A one-hot (lane, byte), B ones: nonzero D elements (count, and the (lane,reg) set summarised)\n");
    for (int p = 0; p < np; ++p) if (desc[p].kind == 1 && (desc[p].y == 0 || desc[p].y == 31)) {
        int cnt, fl, fr; float v; nz(p, cnt, fl, fr, v);
        // collect distinct (lane>>4, reg) and whether all 16 low-lane values present
        int rows_mask = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (O[((size_t)p * 64 + l) * 4 + r] != 0.f) rows_mask |= 1 << ((l >> 4) * 4 + r);
        printf("A lane %2d byte %2d -> cnt %2d rows_mask %04x (row by std layout) val %g\n", desc[p].x, desc[p].y, cnt, rows_mask, v);
    }
    printf("# set4: B one-hot (lane, byte), A ones\n");
    for (int p = 0; p < np; ++p) if (desc[p].kind == 4 && (desc[p].y == 0 || desc[p].y == 31)) {
        int cnt, fl, fr; float v; nz(p, cnt, fl, fr, v);
        int cols_mask = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (O[((size_t)p * 64 + l) * 4 + r] != 0.f) cols_mask |= 1 << (l & 15);
        printf("B lane %2d byte %2d -> cnt %2d cols_mask %04x val %g\n", desc[p].x, desc[p].y, cnt, cols_mask, v);
    }
    printf("# set2: k match: A pos (lane16idx*32+byte) -> matching B pos\n");
    {
        std::vector<int> match(128, -1), nmatch(128, 0);
        for (int p = 0; p < np; ++p) if (desc[p].kind == 2) {
            int cnt, fl, fr; float v; nz(p, cnt, fl, fr, v);
            if (cnt) { match[desc[p].x] = desc[p].y; nmatch[desc[p].x]++; }
        }
        int identical = 0;
        for (int i = 0; i < 128; ++i) identical += match[i] == i && nmatch[i] == 1;
        printf("identical positions: %d / 128\n", identical);
        for (int i = 0; i < 128; ++i) if (!(match[i] == i && nmatch[i] == 1)) printf("  A pos %3d (lane %2d byte %2d) -> B pos %3d (n=%d)\n", i, (i / 32) * 16, i % 32, match[i], nmatch[i]);
    }
    printf("# set3: scale_a of lane ls doubles A element at pos: list per A pos the lanes ls that double it\n");
    for (int kind : {3, 5}) {
        printf("## kind %d (%s)\n", kind, kind == 3 ? "scale_a / A" : "scale_b / B");
        for (int pos = 0; pos < 128; pos += 1) {
            if (!(pos % 32 == 0 || pos % 32 == 15 || pos % 32 == 16 || pos % 32 == 31)) continue;
            printf("pos %3d (lane %2d byte %2d): doubled by scale lanes:", pos, (pos / 32) * 16, pos % 32);
            for (int p = 0; p < np; ++p) if (desc[p].kind == kind && desc[p].x == pos) {
                int cnt, fl, fr; float v; nz(p, cnt, fl, fr, v);
                if (v == 2.f) printf(" %d", desc[p].y);
                else if (v != 1.f) printf(" [%d:%g]", desc[p].y, v);
            }
            printf("\n");
        }
    }
    return 0;
}
