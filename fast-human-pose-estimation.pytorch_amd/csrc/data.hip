// Device data pipeline of the training loop (/root/reference/lib/dataset/JointsDataset.py:113-198 __getitem__ and :233-289
// generate_target): what the reference does per sample in DataLoader workers (cv2 + numpy) as two batch kernels, so that a
// loader only has to hand over decoded 8-bit images, the augmentation parameters it drew and the joint annotations.
//   render_targets_kernel  generate_target: integer placement of the Gaussian patch (bit-exact index work; the patch values
//                          come from the caller, computed with the reference's float32 numpy expression)
//   warp_affine_kernel     cv2.warpAffine(INTER_LINEAR, zero border) in OpenCV's fixed-point arithmetic -> ToTensor ->
//                          Normalize, written straight into the NCHW fp32 batch the stem kernel reads
#include <algorithm>

#include "common.h"

namespace {

// numpy's int(): truncation toward zero of a float64
__device__ __forceinline__ int trunc_i(double v) { return (int)v; }

__global__ __launch_bounds__(256) void render_targets_kernel(const fpd_targets_t a) {
    const int bj = blockIdx.x;
    const double* jt = a.joints + (size_t)bj * 3;
    float* tg = a.target + (size_t)bj * a.H * a.W;
    const int tmp = (a.patch - 1) / 2;                                  // sigma * 3
    // JointsDataset.py:254-255 (float64 division and sum, then int())
    const int mu_x = trunc_i(__dadd_rn(__ddiv_rn(jt[0], a.stride_x), 0.5));
    const int mu_y = trunc_i(__dadd_rn(__ddiv_rn(jt[1], a.stride_y), 0.5));
    const int ulx = mu_x - tmp, uly = mu_y - tmp, brx = mu_x + tmp + 1, bry = mu_y + tmp + 1;
    float w = a.vis[bj];
    const bool outside = ulx >= a.W || uly >= a.H || brx < 0 || bry < 0;
    if (outside) w = 0.f;                                              // :259-263
    if (threadIdx.x == 0) a.weight[bj] = w;
    const bool paste = !outside && w > 0.5f;                           // :279-282
    const int x0 = max(0, ulx), x1 = min(brx, a.W), y0 = max(0, uly), y1 = min(bry, a.H);
    for (int p = threadIdx.x; p < a.H * a.W; p += blockDim.x) {
        const int y = p / a.W, x = p - y * a.W;
        float v = 0.f;
        if (paste && x >= x0 && x < x1 && y >= y0 && y < y1) v = a.g[(y - uly) * a.patch + (x - ulx)];
        tg[p] = v;
    }
}

// cv::saturate_cast<int>(double) = cvRound = round half to even
__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }
__device__ __forceinline__ int sat_short(int v) { return min(max(v, -32768), 32767); }

// One thread per destination pixel, three channels.  OpenCV (imgwarp.cpp WarpAffineInvoker + remapBilinear, 8-bit):
//   AB_BITS 10, INTER_BITS 5, round_delta = 1024/32/2 = 16
//   X = (cvRound((m1*y + m2)*1024) + 16 + cvRound(m0*x*1024)) >> 5     (1/32-pixel source coordinate), Y likewise
//   sx = X >> 5, fx = X & 31; weights (32-fx)(32-fy)*32 ... as shorts (32768 saturates to 32767), sum of the four
//   products + 2^14 >> 15, saturated to 8 bit; taps outside the image contribute the border value 0.
__global__ __launch_bounds__(256) void warp_affine_kernel(const fpd_warp_t a) {
    const int b = blockIdx.y;
    const fpd_warp_src_t s = a.src[b];
    const int HW = a.H * a.W;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        const int y = p / a.W, x = p - y * a.W;
        const int X0 = cv_round(__dmul_rn(__dadd_rn(__dmul_rn(s.minv[1], (double)y), s.minv[2]), 1024.0)) + 16;
        const int Y0 = cv_round(__dmul_rn(__dadd_rn(__dmul_rn(s.minv[4], (double)y), s.minv[5]), 1024.0)) + 16;
        const int ad = cv_round(__dmul_rn(__dmul_rn(s.minv[0], (double)x), 1024.0));
        const int bd = cv_round(__dmul_rn(__dmul_rn(s.minv[3], (double)x), 1024.0));
        const int X = (X0 + ad) >> 5, Y = (Y0 + bd) >> 5;
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
        const int fx = X & 31, fy = Y & 31;
        int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        if (w00 > 32767) w00 = 32767;                                   // saturate_cast<short>; fx = fy = 0: the result is
                                                                        // the source pixel either way
        const bool in_x0 = (unsigned)sx < (unsigned)s.w, in_x1 = (unsigned)(sx + 1) < (unsigned)s.w;
        const bool in_y0 = (unsigned)sy < (unsigned)s.h, in_y1 = (unsigned)(sy + 1) < (unsigned)s.h;
        const uint8_t* r0 = s.img + (int64_t)sy * s.row_bytes;
        const uint8_t* r1 = r0 + s.row_bytes;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p00 = (in_y0 && in_x0) ? r0[3 * sx + c] : 0;
            const int p01 = (in_y0 && in_x1) ? r0[3 * (sx + 1) + c] : 0;
            const int p10 = (in_y1 && in_x0) ? r1[3 * sx + c] : 0;
            const int p11 = (in_y1 && in_x1) ? r1[3 * (sx + 1) + c] : 0;
            int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
            v = min(max(v, 0), 255);
            // ToTensor: uint8 -> float32 / 255; Normalize: (t - mean) / std  (torch fp32 ops, in this order)
            const float t = __fdiv_rn((float)v, 255.f);
            a.out[((size_t)b * 3 + c) * HW + p] = __fdiv_rn(__fsub_rn(t, a.mean[c]), a.std[c]);
        }
    }
}

}  // namespace

int fpd_render_targets_launch(const fpd_targets_t& a, hipStream_t st) {
    FPD_LAUNCH(render_targets_kernel, dim3(a.B * a.J), dim3(256), 0, st, a);
    return 0;
}

int fpd_warp_affine_launch(const fpd_warp_t& a, hipStream_t st) {
    const int bx = std::min(cdiv(a.H * a.W, 256), 256);
    FPD_LAUNCH(warp_affine_kernel, dim3(bx, a.B), dim3(256), 0, st, a);
    return 0;
}
