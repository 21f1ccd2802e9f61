// Row N2 of SURVEY.md section 8f: the resize step of the reference's input transform on the device.
//
// The reference resizes PIL images with T.Resize(img_size, BICUBIC) (strhub/data/module.py:77; read.py:41-43), i.e.
// Pillow's ImagingResample for 8-bit images (third-party, not vendored; restated in oracle/resize_oracle.py and pinned
// against Pillow's own outputs in tests/golden/resize_pillow.npz).  Bit-exactness needs the same arithmetic:
//   * tap weights in IEEE double WITHOUT fused multiply-add (explicit __dmul_rn / __dadd_rn: hipcc contracts a * b + c
//     by default, Pillow's x86-64 build does not), normalised, then 22-bit fixed point, round half away from zero;
//   * horizontal pass first, its result clipped and stored as uint8, then the vertical pass; a pass whose size does not
//     change is skipped;  pixel = clip8((2^21 + sum in * k) >> 22) in 32-bit integers.
// One workgroup per image: the two weight tables are built in LDS (one thread per output column / row), then each thread
// produces output pixels by evaluating, for every vertical tap, the horizontal sum it needs (no intermediate image: the
// crops are small and the taps are few, so recomputing the horizontal pass per output row is cheaper than a round trip).
#pragma once
#include "common.h"

namespace pq {

struct ImageDesc { const unsigned char* data; int height, width; long long row_stride; };   // RGB, HWC, uint8 (mirrors parseq_image_desc)

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ double rs_bicubic(double x) {
    const double a = -0.5;
    x = fabs(x);
    if (x < 1.0) return __dadd_rn(__dmul_rn(__dmul_rn(__dadd_rn(__dmul_rn(a + 2.0, x), -(a + 3.0)), x), x), 1.0);   // ((a+2)x - (a+3)) x x + 1
    if (x < 2.0) return __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(x, -5.0), x), 8.0), x), -4.0), a);   // (((x-5)x + 8)x - 4) a
    return 0.0;
}

// weights of output index xx of one pass into kk[0 .. n), returns (first tap, n)
__device__ __forceinline__ void rs_coeffs(int in_size, int out_size, int xx, int* kk, int* first, int* count) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = __dmul_rn(2.0, filterscale);
    const double ss = 1.0 / filterscale;
    const double center = __dmul_rn((double)xx + 0.5, scale);
    int xmin = (int)(__dadd_rn(__dadd_rn(center, -support), 0.5));
    if (xmin < 0) xmin = 0;
    int xmax = (int)(__dadd_rn(__dadd_rn(center, support), 0.5));
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww = __dadd_rn(ww, rs_bicubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + xmin), -center), 0.5), ss)));
    for (int x = 0; x < xmax; ++x) {
        double w = rs_bicubic(__dmul_rn(__dadd_rn(__dadd_rn((double)(x + xmin), -center), 0.5), ss));
        if (ww != 0.0) w = __ddiv_rn(w, ww);
        const double f = __dmul_rn(w, (double)(1 << RS_PRECISION_BITS));
        kk[x] = w < 0.0 ? (int)(__dadd_rn(-0.5, f)) : (int)(__dadd_rn(0.5, f));
    }
    *first = xmin; *count = xmax;
}

__device__ __forceinline__ int rs_clip8(int v) {
    v >>= RS_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// out: uint8 [B][3][out_h][out_w].  ksh / ksv: LDS row pitch (max taps over the batch) of the horizontal / vertical table.
static __global__ __launch_bounds__(256)
void resize_bicubic_kernel(const ImageDesc* __restrict__ images, int out_h, int out_w, int ksh, int ksv,
                           unsigned char* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rs[];
    int* kh = reinterpret_cast<int*>(smem_rs);            // [out_w][ksh]
    int* kv = kh + out_w * ksh;                            // [out_h][ksv]
    int* bh = kv + out_h * ksv;                            // [out_w][2]
    int* bv = bh + 2 * out_w;                              // [out_h][2]
    const ImageDesc im = images[blockIdx.x];
    const int H = im.height, W = im.width;
    const bool pass_h = W != out_w, pass_v = H != out_h;
    for (int i = threadIdx.x; i < out_w + out_h; i += blockDim.x) {
        if (i < out_w) { if (pass_h) rs_coeffs(W, out_w, i, kh + i * ksh, bh + 2 * i, bh + 2 * i + 1); }
        else { const int y = i - out_w; if (pass_v) rs_coeffs(H, out_h, y, kv + y * ksv, bv + 2 * y, bv + 2 * y + 1); }
    }
    __syncthreads();
    const int half = 1 << (RS_PRECISION_BITS - 1);
    unsigned char* dst = out + (size_t)blockIdx.x * 3 * out_h * out_w;
    for (int p = threadIdx.x; p < out_h * out_w; p += blockDim.x) {
        const int yy = p / out_w, xx = p - yy * out_w;
        const int x0 = pass_h ? bh[2 * xx] : xx, nx = pass_h ? bh[2 * xx + 1] : 1;
        const int y0 = pass_v ? bv[2 * yy] : yy, ny = pass_v ? bv[2 * yy + 1] : 1;
        int acc[3] = {half, half, half};
        for (int iy = 0; iy < ny; ++iy) {
            const unsigned char* row = im.data + (size_t)(y0 + iy) * im.row_stride + (size_t)x0 * 3;
            int t[3];
            if (pass_h) {
                int s0 = half, s1 = half, s2 = half;
                const int* k = kh + xx * ksh;
                for (int ix = 0; ix < nx; ++ix) {
                    const int w = k[ix];
                    s0 += (int)row[3 * ix] * w; s1 += (int)row[3 * ix + 1] * w; s2 += (int)row[3 * ix + 2] * w;
                }
                t[0] = rs_clip8(s0); t[1] = rs_clip8(s1); t[2] = rs_clip8(s2);
            } else {
                t[0] = row[0]; t[1] = row[1]; t[2] = row[2];
            }
            if (pass_v) {
                const int w = kv[yy * ksv + iy];
                acc[0] += t[0] * w; acc[1] += t[1] * w; acc[2] += t[2] * w;
            } else {
                acc[0] = t[0]; acc[1] = t[1]; acc[2] = t[2];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[((size_t)c * out_h + yy) * out_w + xx] = (unsigned char)(pass_v ? rs_clip8(acc[c]) : acc[c]);
    }
}

}  // namespace pq
