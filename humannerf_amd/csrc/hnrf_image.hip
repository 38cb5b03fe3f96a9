// Image pre-processing in front of the path: what Dataset.load_image
// (core/data/human_nerf/train.py:351-417, freeview.py:137-166) does to the two decoded PNGs of a frame through
// OpenCV -- cv2.undistort (train.py:366-371), the float64 composite over the background colour (train.py:406) and
// cv2.resize with INTER_LANCZOS4 for the image / INTER_LINEAR for the mask (train.py:408-417).  Every prepared
// ZJU-MoCap directory has `distortions` and every 387 / wild yaml sets resize_img_scale 0.5, so this runs per frame.
//
// The numpy statement of the same functions is humannerf_amd/imageproc.py (its header lists what of OpenCV is
// restated and what stays unpinned: cv2 is not importable).  This file follows it operation for operation, in
// float64 with contraction OFF (the Makefile compiles it with -ffp-contract=off), so the two routes give the same
// bits: the 8-bit undistortion exactly, the resized float32 pixels exactly as long as libm's sin / cos are out of
// the picture -- the per-axis coefficient tables are built once on the host and handed in.
//
// All three kernels are gather-type, HBM/L2-bound and tiny against the path: a 1024x1024x3 undistortion reads and
// writes 3 MB; a training item's 6 x 32 x 32 targets take 6144 x 64 taps out of L2.  One lane per output pixel.
#include "hnrf_common.h"

namespace hnrf {

// ---- cv2.undistort: per-stripe inverse camera, Brown-Conrady forward model, CV_16SC2 fixed-point map, bilinear
// remap with the exact 1/32-fraction weights, BORDER_CONSTANT 0.  cam = fx fy u0 v0 k1 k2 p1 p2 k3 (float64).
__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t* __restrict__ src, int H, int W, int C,
                                                        const double* __restrict__ cam, const double* __restrict__ ir,
                                                        int stripe_rows, uint8_t* __restrict__ dst) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const int yrow = p / W, jcol = p - yrow * W;
    const double* r = ir + (size_t)(yrow / stripe_rows) * 9;
    const double i = (double)(yrow % stripe_rows), j = (double)jcol;
    const double fx = cam[0], fy = cam[1], u0 = cam[2], v0 = cam[3];
    const double k1 = cam[4], k2 = cam[5], p1 = cam[6], p2 = cam[7], k3 = cam[8];
    const double w = 1.0 / (j * r[6] + (i * r[7] + r[8]));
    const double x = (j * r[0] + (i * r[1] + r[2])) * w;
    const double y = (j * r[3] + (i * r[4] + r[5])) * w;
    const double x2 = x * x, y2 = y * y;
    const double r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / 1.0;
    const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
    const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
    const double u = fx * xd + u0, v = fy * yd + v0;
    const double lim = 2147483647.0;
    const long long iu = (long long)rint(fmin(fmax(u * 32.0, -lim - 1.0), lim));   // cvRound: nearest, ties to even
    const long long iv = (long long)rint(fmin(fmax(v * 32.0, -lim - 1.0), lim));
    const int sx = (int)(iu >> 5), sy = (int)(iv >> 5);
    const int ax = (int)(iu & 31), ay = (int)(iv & 31);
    const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const int64_t b00 = ((int64_t)sy * W + sx) * C;
    for (int c = 0; c < C; ++c) {
        const int v00 = (x0 && y0) ? src[b00 + c] : 0;
        const int v01 = (x1 && y0) ? src[b00 + C + c] : 0;
        const int v10 = (x0 && y1) ? src[b00 + (int64_t)W * C + c] : 0;
        const int v11 = (x1 && y1) ? src[b00 + (int64_t)W * C + C + c] : 0;
        dst[(int64_t)p * C + c] = (uint8_t)((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + 512) >> 10);
    }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- composite over the background (+ cv2.resize INTER_LANCZOS4) for n windows of ph x pw destination pixels.
// resize == 0: destination = source grid, out = float32((a o + (1 - a) bg) / 255).
// resize == 1: horizontal 8-tap pass then vertical 8-tap pass over the float64 composite, each a left-to-right sum
//              with float32 weights (resize.cpp HResizeLanczos4 / VResizeLanczos4 for CV_64F), taps clamped.
__global__ __launch_bounds__(256) void composite_windows_kernel(
    const uint8_t* __restrict__ orig, const uint8_t* __restrict__ alpha, int Hs, int Ws, const float* __restrict__ bg,
    int resize, const int* __restrict__ xofs, const float* __restrict__ xw, const int* __restrict__ yofs,
    const float* __restrict__ yw, const int* __restrict__ win_xy, int n_win, int ph, int pw, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n_win * ph * pw) return;
    const int wi = (int)(t / ((int64_t)ph * pw));
    const int rem = (int)(t - (int64_t)wi * ph * pw);
    const int dy = win_xy[2 * wi + 1] + rem / pw, dx = win_xy[2 * wi] + rem % pw;
    const double bgc[3] = {(double)bg[0], (double)bg[1], (double)bg[2]};
    double res[3];
    if (!resize) {
        const int64_t b = ((int64_t)dy * Ws + dx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double a = (double)alpha[b + c] / 255.0;
            res[c] = a * (double)orig[b + c] + (1.0 - a) * bgc[c];
        }
    } else {
        const int sx0 = xofs[dx] - 3, sy0 = yofs[dy] - 3;
        float wx[8], wy[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { wx[k] = xw[dx * 8 + k]; wy[k] = yw[dy * 8 + k]; }
        res[0] = res[1] = res[2] = 0.0;
        for (int ky = 0; ky < 8; ++ky) {
            const int sy = clampi(sy0 + ky, 0, Hs - 1);
            double row[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) {
                const int sx = clampi(sx0 + kx, 0, Ws - 1);
                const int64_t b = ((int64_t)sy * Ws + sx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const double a = (double)alpha[b + c] / 255.0;
                    const double v = a * (double)orig[b + c] + (1.0 - a) * bgc[c];
                    row[c] = row[c] + v * (double)wx[kx];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) res[c] = ky == 0 ? row[c] * (double)wy[0] : res[c] + row[c] * (double)wy[ky];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[t * 3 + c] = (float)(res[c] / 255.0);
}

// ---- cv2.resize(mask / 255, INTER_LINEAR), channel `ch` of a 3-channel uint8 mask -> float32 [Hd, Wd].
// mode 1: the two-tap passes of HResizeLinear / VResizeLinear; mode 2: the 2x2 box mean hal::resize substitutes at
// scale exactly 1/2 (INTER_AREA fast path: ((s00 + s01) + s10) + s11) * 0.25).
__global__ __launch_bounds__(256) void resize_mask_kernel(const uint8_t* __restrict__ alpha, int Hs, int Ws, int ch,
                                                          int mode, const int* __restrict__ xofs,
                                                          const float* __restrict__ xw, const int* __restrict__ yofs,
                                                          const float* __restrict__ yw, int Hd, int Wd,
                                                          float* __restrict__ out) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Hd * Wd) return;
    const int dy = p / Wd, dx = p - dy * Wd;
    double res;
    if (mode == 2) {
        const int64_t b = ((int64_t)(2 * dy) * Ws + 2 * dx) * 3 + ch;
        const double s00 = (double)alpha[b] / 255.0, s01 = (double)alpha[b + 3] / 255.0;
        const double s10 = (double)alpha[b + (int64_t)Ws * 3] / 255.0, s11 = (double)alpha[b + (int64_t)Ws * 3 + 3] / 255.0;
        res = (((s00 + s01) + s10) + s11) * 0.25;
    } else {
        const int sx = xofs[dx], sy = yofs[dy];
        const double a0 = (double)xw[dx * 2], a1 = (double)xw[dx * 2 + 1];
        const double b0 = (double)yw[dy * 2], b1 = (double)yw[dy * 2 + 1];
        double rows[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int y = clampi(sy + k, 0, Hs - 1);
            const double l = (double)alpha[((int64_t)y * Ws + clampi(sx, 0, Ws - 1)) * 3 + ch] / 255.0;
            const double r = (double)alpha[((int64_t)y * Ws + clampi(sx + 1, 0, Ws - 1)) * 3 + ch] / 255.0;
            rows[k] = l * a0 + r * a1;
        }
        res = rows[0] * b0 + rows[1] * b1;
    }
    out[p] = (float)res;
}

}  // namespace hnrf

using namespace hnrf;

extern "C" int hnrf_undistort_image(const uint8_t* src, int H, int W, int C, const double* cam, const double* ir,
                                    int n_stripes, int stripe_rows, uint8_t* dst, void* stream) {
    HNRF_REQUIRE(src && cam && ir && dst, HNRF_E_ARG, "hnrf_undistort_image: null pointer");
    HNRF_REQUIRE(H > 0 && W > 0 && C > 0 && C <= 4 && (int64_t)H * W * C < 2147483647LL, HNRF_E_ARG,
                 "hnrf_undistort_image: bad image %dx%dx%d", H, W, C);
    HNRF_REQUIRE(stripe_rows > 0 && (int64_t)n_stripes * stripe_rows >= H, HNRF_E_ARG,
                 "hnrf_undistort_image: %d stripes of %d rows do not cover %d rows", n_stripes, stripe_rows, H);
    HNRF_REQUIRE(src != dst, HNRF_E_ARG, "hnrf_undistort_image: in-place call");
    hipLaunchKernelGGL(undistort_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, H, W, C, cam, ir,
                       stripe_rows, dst);
    return check_launch("hnrf_undistort_image");
}

extern "C" int hnrf_composite_windows(const uint8_t* orig, const uint8_t* alpha, int Hs, int Ws, const float* bgcolor,
                                      int resize, const int* xofs, const float* xw, const int* yofs, const float* yw,
                                      int Hd, int Wd, const int* win_xy, int n_win, int ph, int pw, float* out,
                                      void* stream) {
    HNRF_REQUIRE(orig && alpha && bgcolor && win_xy && out, HNRF_E_ARG, "hnrf_composite_windows: null pointer");
    HNRF_REQUIRE(Hs > 0 && Ws > 0 && (int64_t)Hs * Ws * 3 < 2147483647LL && Hd > 0 && Wd > 0, HNRF_E_ARG,
                 "hnrf_composite_windows: bad image size");
    HNRF_REQUIRE(resize == 0 || resize == 1, HNRF_E_ARG, "hnrf_composite_windows: resize must be 0 or 1");
    HNRF_REQUIRE(resize == 0 ? (Hd == Hs && Wd == Ws) : (xofs && xw && yofs && yw), HNRF_E_ARG,
                 "hnrf_composite_windows: %s", resize ? "coefficient tables missing" : "sizes differ without resize");
    HNRF_REQUIRE(n_win > 0 && ph > 0 && pw > 0 && ph <= Hd && pw <= Wd, HNRF_E_ARG, "hnrf_composite_windows: bad windows");
    const int64_t n = (int64_t)n_win * ph * pw;
    hipLaunchKernelGGL(composite_windows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, orig,
                       alpha, Hs, Ws, bgcolor, resize, xofs, xw, yofs, yw, win_xy, n_win, ph, pw, out);
    return check_launch("hnrf_composite_windows");
}

extern "C" int hnrf_resize_mask(const uint8_t* alpha, int Hs, int Ws, int channel, int mode, const int* xofs,
                                const float* xw, const int* yofs, const float* yw, int Hd, int Wd, float* out,
                                void* stream) {
    HNRF_REQUIRE(alpha && out, HNRF_E_ARG, "hnrf_resize_mask: null pointer");
    HNRF_REQUIRE(Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && (int64_t)Hs * Ws * 3 < 2147483647LL && channel >= 0 && channel < 3,
                 HNRF_E_ARG, "hnrf_resize_mask: bad size / channel");
    HNRF_REQUIRE(mode == 1 || mode == 2, HNRF_E_ARG, "hnrf_resize_mask: mode must be 1 (two-tap) or 2 (2x2 box)");
    HNRF_REQUIRE(mode == 2 ? (2 * Hd <= Hs && 2 * Wd <= Ws) : (xofs && xw && yofs && yw), HNRF_E_ARG,
                 "hnrf_resize_mask: %s", mode == 2 ? "2x2 box needs 2*dst <= src" : "coefficient tables missing");
    hipLaunchKernelGGL(resize_mask_kernel, dim3((Hd * Wd + 255) / 256), dim3(256), 0, (hipStream_t)stream, alpha, Hs, Ws,
                       channel, mode, xofs, xw, yofs, yw, Hd, Wd, out);
    return check_launch("hnrf_resize_mask");
}

// ---- fold of a transposed 3-D convolution (kernel 4, stride 2, padding 1): the weight-volume decoder's layers
// (core/utils/network_util.py:12-50: ConvTranspose3d stack of MotionWeightVolumeDecoder).  The GEMM col[i, (co, k)] =
// sum_ci x[ci, i] W[ci, (co, k)] (one library GEMM on the weight's native layout, humannerf_amd/network.py) leaves, per
// input voxel i = (d, h, w), the 4x4x4 block it adds to the output at (2d-1+kd, 2h-1+kh, 2w-1+kw); this kernel gathers,
// per OUTPUT voxel, the 8 blocks that reach it (2 per axis) and adds the bias.  It is the adjoint of the pad + unfold
// the backward GEMMs read through.  MIOpen's batch-1 transposed convolutions picked anything from a GEMM + col2im
// (0.15 ms) to a grouped-convolution kernel (1.3 ms per training step) depending on the box's find results.
namespace hnrf {
__global__ __launch_bounds__(256) void deconv_fold_kernel(const float* __restrict__ col, const float* __restrict__ bias,
                                                          int cout, int D, int H, int W, float* __restrict__ out) {
    const int64_t n = (int64_t)cout * 8 * D * H * W;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int OW = 2 * W, OH = 2 * H, OD = 2 * D;
    const int ow = (int)(t % OW), oh = (int)((t / OW) % OH), od = (int)((t / ((int64_t)OW * OH)) % OD);
    const int co = (int)(t / ((int64_t)OW * OH * OD));
    // per axis: output o gets taps (k, i) with 2 i - 1 + k = o: o even -> (1, o/2), (3, o/2 - 1); o odd -> (2, (o-1)/2), (0, (o+1)/2)
    int kd[2], id[2], kh[2], ih[2], kw[2], iw[2];
    kd[0] = (od & 1) ? 2 : 1; id[0] = od >> 1; kd[1] = (od & 1) ? 0 : 3; id[1] = (od & 1) ? (od >> 1) + 1 : (od >> 1) - 1;
    kh[0] = (oh & 1) ? 2 : 1; ih[0] = oh >> 1; kh[1] = (oh & 1) ? 0 : 3; ih[1] = (oh & 1) ? (oh >> 1) + 1 : (oh >> 1) - 1;
    kw[0] = (ow & 1) ? 2 : 1; iw[0] = ow >> 1; kw[1] = (ow & 1) ? 0 : 3; iw[1] = (ow & 1) ? (ow >> 1) + 1 : (ow >> 1) - 1;
    float acc = bias ? bias[co] : 0.f;
    const int64_t rs = (int64_t)cout * 64;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if (id[a] < 0 || id[a] >= D) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (ih[b] < 0 || ih[b] >= H) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (iw[c] < 0 || iw[c] >= W) continue;
                const int64_t i = ((int64_t)id[a] * H + ih[b]) * W + iw[c];
                acc += col[i * rs + (int64_t)co * 64 + kd[a] * 16 + kh[b] * 4 + kw[c]];
            }
        }
    }
    out[t] = acc;
}
}  // namespace hnrf

extern "C" int hnrf_deconv_fold(const float* col, const float* bias, int cout, int D, int H, int W, float* out, void* stream) {
    HNRF_REQUIRE(col && out, HNRF_E_ARG, "hnrf_deconv_fold: null pointer");
    HNRF_REQUIRE(cout > 0 && D > 0 && H > 0 && W > 0 && (int64_t)cout * 8 * D * H * W < (1LL << 40), HNRF_E_ARG,
                 "hnrf_deconv_fold: bad dimensions");
    const int64_t n = (int64_t)cout * 8 * D * H * W;
    hipLaunchKernelGGL(deconv_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, col, bias, cout,
                       D, H, W, out);
    return check_launch("hnrf_deconv_fold");
}
