// dfd_augment.hip — RandomRotation + ColorJitter of the reference's default 224-pixel training pipeline on the device
// (/root/reference/trainers/efficientnet.py:173-181: transforms.RandomRotation(10), transforms.ColorJitter(0.2, 0.2, 0.2, 0.05)),
// between the resize / crop kernel (dfd_resize.hip) and the flip / to-float / normalise / erase kernel (dfd_image_prep).
//
// BYTE work, restated from Pillow (which torchvision calls for PIL images) so that the result equals the CPU pipeline of data.py bit
// for bit — oracle/image_ref.py is the same arithmetic in numpy, pinned against Pillow itself:
//   rotate     Image.rotate(angle, NEAREST, expand=False) = Geometry.c affine_fixed: 16.16 fixed point, the six coefficients are
//              formed on the host in double precision exactly as Image.rotate forms them (data.rotate_plan); fill 0;
//   blends     ImageEnhance.{Brightness, Contrast, Color}.enhance(f) = Blend.c: (int)a + alpha * ((int)b - (int)a) evaluated in f32
//              (a multiply and an add, no fused multiply-add: built with -ffp-contract=off), truncated for 0 <= f <= 1, clipped
//              otherwise; Contrast's degenerate is the ROUNDED MEAN of convert("L") of the image AS IT IS when the operation runs
//              (after the operations that precede it in the drawn order), an integer reduction over the picture;
//   hue        convert("HSV") -> (h + delta) mod 256 -> convert("RGB"): Convert.c's float / double mix, spelled out.
// One 1024-thread workgroup per picture, the picture resident in LDS (H * W * 3 <= 156 KB: up to 228 x 228) for all of it: one
// read and one write of the batch for rotation + four colour operations + the reduction.
#include "dfd_common.h"

#define AUG_THREADS 1024
#define AUG_MAX_BYTES (156 * 1024)

__device__ __forceinline__ unsigned char aug_blend(int d, int v, float alpha, bool interp) {
    const float temp = (float)d + alpha * (float)(v - d);
    if (interp) return (unsigned char)(int)temp;
    if (temp <= 0.f) return 0;
    if (temp >= 255.f) return 255;
    return (unsigned char)(int)temp;
}
__device__ __forceinline__ int aug_l(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int aug_clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

__device__ __forceinline__ void aug_rgb2hsv(int r, int g, int b, int& uh, int& us, int& uv) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    uv = maxc;
    if (minc == maxc) { uh = 0; us = 0; return; }
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = aug_clip8((int)((double)h * 255.0));
    us = aug_clip8((int)((double)s * 255.0));
}
__device__ __forceinline__ void aug_hsv2rgb(int h, int s, int v, int& r, int& g, int& b) {
    if (s == 0) { r = g = b = v; return; }
    const double hd = (double)(float)h * 6.0 / 255.0;
    const int i = (int)floor(hd);
    const double f = (double)(float)(hd - (double)(float)i);
    const double fs = (double)(float)((double)(float)s / 255.0);
    const double vd = (double)(float)v;
    const int p = aug_clip8((int)floor(vd * (1.0 - fs) + 0.5));
    const int q = aug_clip8((int)floor(vd * (1.0 - fs * f) + 0.5));
    const int t = aug_clip8((int)floor(vd * (1.0 - fs * (1.0 - f)) + 0.5));
    switch (i % 6) {
        case 0: r = v; g = t; b = p; break;
        case 1: r = q; g = v; b = p; break;
        case 2: r = p; g = v; b = t; break;
        case 3: r = p; g = q; b = v; break;
        case 4: r = t; g = p; b = v; break;
        default: r = v; g = p; b = q; break;
    }
}

__global__ void __launch_bounds__(AUG_THREADS)
k_augment_u8(const unsigned char* __restrict__ src, const dfd_augment_job* __restrict__ jobs, unsigned char* __restrict__ dst, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) unsigned char img[];
    __shared__ int red[AUG_THREADS / 64];
    __shared__ int mean_sh;
    const int n = blockIdx.x, t = threadIdx.x;
    const dfd_augment_job jb = jobs[n];
    const int npx = H * W;
    const unsigned char* in = src + (long)n * npx * 3;
    // ---- rotation (or copy): gather from global memory into the LDS picture
    for (int p = t; p < npx; p += AUG_THREADS) {
        const int y = p / W, x = p - y * W;
        int sx = x, sy = y;
        bool ok = true;
        if (jb.mode == 1) {
            const long long xx = (long long)jb.a[2] + (long long)y * jb.a[1] + (long long)x * jb.a[0];
            const long long yy = (long long)jb.a[5] + (long long)y * jb.a[4] + (long long)x * jb.a[3];
            sx = (int)(xx >> 16); sy = (int)(yy >> 16);
            ok = sx >= 0 && sx < W && sy >= 0 && sy < H;
        } else if (jb.mode == 2) { sx = W - 1 - x; sy = H - 1 - y; }
        else if (jb.mode == 3) { sy = x; sx = W - 1 - y; }
        else if (jb.mode == 4) { sy = H - 1 - x; sx = y; }
        unsigned char r = 0, g = 0, b = 0;
        if (ok) { const unsigned char* q = in + ((long)sy * W + sx) * 3; r = q[0]; g = q[1]; b = q[2]; }
        img[3 * p] = r; img[3 * p + 1] = g; img[3 * p + 2] = b;
    }
    __syncthreads();
    // ---- the colour operations in the drawn order
    for (int slot = 0; slot < 4; ++slot) {
        const int op = jb.order[slot];
        if (op < 0 || op > 3 || !((jb.enable >> op) & 1)) continue;
        if (op == 0) {
            const float a = jb.fb;
            const bool interp = a >= 0.f && a <= 1.f;
            for (int i = t; i < npx * 3; i += AUG_THREADS) img[i] = aug_blend(0, img[i], a, interp);
        } else if (op == 1) {
            int part = 0;
            for (int p = t; p < npx; p += AUG_THREADS) part += aug_l(img[3 * p], img[3 * p + 1], img[3 * p + 2]);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
            if ((t & 63) == 0) red[t >> 6] = part;
            __syncthreads();
            if (t == 0) {
                long long sum = 0;
                for (int k = 0; k < AUG_THREADS / 64; ++k) sum += red[k];
                mean_sh = (int)((double)sum / (double)npx + 0.5);
            }
            __syncthreads();
            const int mean = mean_sh;
            const float a = jb.fc;
            const bool interp = a >= 0.f && a <= 1.f;
            for (int i = t; i < npx * 3; i += AUG_THREADS) img[i] = aug_blend(mean, img[i], a, interp);
        } else if (op == 2) {
            const float a = jb.fs;
            const bool interp = a >= 0.f && a <= 1.f;
            for (int p = t; p < npx; p += AUG_THREADS) {
                const int r = img[3 * p], g = img[3 * p + 1], b = img[3 * p + 2];
                const int l = aug_l(r, g, b);
                img[3 * p] = aug_blend(l, r, a, interp); img[3 * p + 1] = aug_blend(l, g, a, interp); img[3 * p + 2] = aug_blend(l, b, a, interp);
            }
        } else {
            for (int p = t; p < npx; p += AUG_THREADS) {
                int h, s, v, r, g, b;
                aug_rgb2hsv(img[3 * p], img[3 * p + 1], img[3 * p + 2], h, s, v);
                h = (h + jb.dh) & 255;
                aug_hsv2rgb(h, s, v, r, g, b);
                img[3 * p] = (unsigned char)r; img[3 * p + 1] = (unsigned char)g; img[3 * p + 2] = (unsigned char)b;
            }
        }
        __syncthreads();
    }
    unsigned char* out = dst + (long)n * npx * 3;
    for (int i = t; i < npx * 3; i += AUG_THREADS) out[i] = img[i];
}

extern "C" int dfd_augment_u8(const unsigned char* src, const dfd_augment_job* jobs_dev, unsigned char* dst, int N, int H, int W,
                              dfd_stream stream) {
    if (!src || !jobs_dev || !dst || N < 1 || H < 1 || W < 1 || src == dst) return DFD_EINVAL;
    const long bytes = (long)H * W * 3;
    if (bytes > AUG_MAX_BYTES || H >= 32768 || W >= 32768) return DFD_EUNSUPPORTED;        // the picture must fit one CU's LDS
    struct AugTag;
    dfd_allow_lds_once<AugTag>(k_augment_u8, AUG_MAX_BYTES);
    hipLaunchKernelGGL(k_augment_u8, dim3(N), dim3(AUG_THREADS), (size_t)((bytes + 15) / 16 * 16), (hipStream_t)stream, src, jobs_dev, dst, H, W);
    return DFD_CHECK_LAUNCH();
}
