// dfd_resize.hip — Resize / CenterCrop / RandomResizedCrop of decoded uint8 RGB images on the device
// (reference pipeline trainers/efficientnet.py:111-234, orchestrator.py:316-347: torchvision's
// Resize + CenterCrop / RandomResizedCrop on PIL images with bilinear interpolation).
//
// Bit-exact restatement of Pillow's ImagingResample for 8-bit images with the BILINEAR filter (Resample.c; pinned by
// tests/test_ops_gpu.py against Pillow itself on the CPU):
//   * per axis: scale = in / out, filterscale = max(scale, 1), support = 1 * filterscale (anti-aliased when shrinking),
//     for every output index xx: center = in0 + (xx + 0.5) * scale, xmin = max(0, (int)(center - support + 0.5)),
//     xmax = min(in, (int)(center + support + 0.5)) - xmin, w[x] = tri((x + xmin - center + 0.5) / filterscale),
//     normalised by their sum — all in DOUBLE as Pillow does, then fixed point: k = (int)(+-0.5 + w * 2^22);
//   * two passes with an 8-bit intermediate: horizontal first, out = clip8((2^21 + sum pix * k) >> 22), then vertical on
//     the rounded intermediate.  A thread computes its output pixel directly: for every source row of its vertical window it
//     forms the rounded horizontal value and feeds it to the vertical sum, so the result equals the two-pass one bit for bit.
// One image = one descriptor: the resize acts on the box (bx, by, bw, bh) of the source (`img.crop(box).resize((rw, rh))`,
// box = whole image for Resize) and the output is the OH x OW window at (cx, cy) of the resized image (CenterCrop; pixels
// outside the resized image are 0, which is CenterCrop's black padding of too-small images).
// Images arrive as tightly packed HWC uint8 in ONE flat buffer (variable sizes, no padding to a common canvas); output is
// uint8 [N][OH][OW][3], what dfd_image_prep (flip, to-float, normalise, erase) takes.
// Roofline: HBM-trivial (a 256-image batch of 500x375 photos is 144 MB in, 38 MB out); the kernel is integer VALU work,
// ~ (2 scale + 1)^2 taps per output pixel and channel.
#include "dfd_common.h"

#define RS_MAXTAPS 96        // taps per axis: ceil(support) * 2 + 1 with support = scale -> shrink factors up to ~47

struct RsAxis {
    int mn, cnt;
    int k[RS_MAXTAPS];
};

// coefficients of output index `xx` along one axis, as Pillow's precompute_coeffs + normalize_coeffs_8bpc
__device__ __forceinline__ void rs_coeffs(int in_size, double in0, double in1, int out_size, int xx, int& mn, int& cnt, int* k) {
    const double scale = (in1 - in0) / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = in0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > RS_MAXTAPS) xmax = RS_MAXTAPS;     // host rejects such shapes; keeps the loop bounded
    double w[RS_MAXTAPS];
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double v = (x + xmin - center + 0.5) * ss;
        if (v < 0.0) v = -v;
        const double f = v < 1.0 ? 1.0 - v : 0.0;
        w[x] = f;
        ww += f;
    }
    for (int x = 0; x < xmax; ++x) {
        double kv = w[x];
        if (ww != 0.0) kv /= ww;
        k[x] = kv < 0 ? (int)(-0.5 + kv * (double)(1 << 22)) : (int)(0.5 + kv * (double)(1 << 22));
    }
    mn = xmin;
    cnt = xmax;
}
__device__ __forceinline__ int clip8(int v) {
    // Pillow: clip8(in) = clip8_lookups[in >> PRECISION_BITS]: saturate to 0..255
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one workgroup = one output row segment of one image: threads share the vertical coefficients (same oy), each owns an ox
__global__ void __launch_bounds__(256)
k_resize_crop(const unsigned char* __restrict__ src, const dfd_resize_job* __restrict__ jobs, unsigned char* __restrict__ dst,
              int OH, int OW) {
    const int n = blockIdx.z, oy = blockIdx.y;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    const dfd_resize_job jb = jobs[n];
    __shared__ RsAxis ky;
    const int ry = oy + jb.cy;                    // row / column in the resized image
    const bool row_in = ry >= 0 && ry < jb.rh;
    if (threadIdx.x == 0 && row_in) rs_coeffs(jb.bh, 0.0, (double)jb.bh, jb.rh, ry, ky.mn, ky.cnt, ky.k);
    __syncthreads();
    if (ox >= OW) return;
    unsigned char* o = dst + (((long)n * OH + oy) * OW + ox) * 3;
    const int rx = ox + jb.cx;
    if (!row_in || rx < 0 || rx >= jb.rw) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
    int xmn, xcnt, kx[RS_MAXTAPS];
    rs_coeffs(jb.bw, 0.0, (double)jb.bw, jb.rw, rx, xmn, xcnt, kx);
    int acc0 = 1 << 21, acc1 = 1 << 21, acc2 = 1 << 21;
    for (int y = 0; y < ky.cnt; ++y) {
        const unsigned char* row = src + jb.offset + ((long)(jb.by + ky.mn + y) * jb.W + jb.bx + xmn) * 3;
        int h0 = 1 << 21, h1 = 1 << 21, h2 = 1 << 21;
        for (int x = 0; x < xcnt; ++x) {
            const int kk = kx[x];
            h0 += (int)row[3 * x] * kk;
            h1 += (int)row[3 * x + 1] * kk;
            h2 += (int)row[3 * x + 2] * kk;
        }
        const int kv = ky.k[y];
        acc0 += clip8(h0) * kv;
        acc1 += clip8(h1) * kv;
        acc2 += clip8(h2) * kv;
    }
    o[0] = (unsigned char)clip8(acc0);
    o[1] = (unsigned char)clip8(acc1);
    o[2] = (unsigned char)clip8(acc2);
}

extern "C" int dfd_resize_crop_u8(const unsigned char* src, const dfd_resize_job* jobs_dev, unsigned char* dst, int N, int OH, int OW,
                                  int max_shrink, dfd_stream stream) {
    if (!src || !jobs_dev || !dst || N < 1 || OH < 1 || OW < 1) return DFD_EINVAL;
    // taps per axis = ceil(scale) * 2 + 1 must fit RS_MAXTAPS; the caller passes the largest shrink factor of the batch
    if (max_shrink < 1 || 2 * (max_shrink + 1) + 1 > RS_MAXTAPS) return DFD_EUNSUPPORTED;
    const dim3 grid((OW + 255) / 256, OH, N);
    hipLaunchKernelGGL(k_resize_crop, grid, dim3(256), 0, (hipStream_t)stream, src, jobs_dev, dst, OH, OW);
    return DFD_CHECK_LAUNCH();
}
