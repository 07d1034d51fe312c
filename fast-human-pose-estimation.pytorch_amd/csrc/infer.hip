// Inference post-processing of the validate / flip-test path on the device
// (/root/reference/lib/core/function.py:189-332 `validate`; the reference round-trips every tensor through numpy):
//   flip_w_kernel       input[:, :, :, ::-1]                                     (function.py:217-221: np.flip(input, 3))
//   flip_merge_kernel   flip_back (utils/transforms.py:15-29: reverse the width axis, swap the left/right joint
//                       channels) + the one-pixel shift of the flipped map (function.py:233-236, TEST.SHIFT_HEATMAP)
//                       + the average (output + output_flipped) * 0.5 (function.py:238), one pass
//   final_preds_kernel  get_final_preds (core/inference.py:49-79): heat-map arg-max (first maximum wins, coordinates
//                       zeroed where the maximum is not positive), the quarter-pixel shift towards the higher neighbour
//                       (TEST.POST_PROCESS) and the affine map back to image coordinates (transform_preds,
//                       utils/transforms.py:50-55: float64 [x, y, 1] . trans^T, result stored as float32)
// All three are index / byte work or exact fp32 arithmetic in the reference's own operation order: results are
// bit-identical to the reference functions (tests/golden/infer_ref.npz).
#include "argmax.h"
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void flip_w_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int W) {
    const int64_t total = rows * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        y[i] = x[r * W + (W - 1 - c)];
    }
}

// a, b: [N,J,H,W] fp32 (the module API's outputs for the image and for the flipped image); src[j] = channel of b that
// lands in channel j after flip_back's sequential pair swaps.
__global__ __launch_bounds__(256) void flip_merge_kernel(const fpd_flipmerge_t p) {
    const int64_t total = (int64_t)p.N * p.J * p.H * p.W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.W);
        const int64_t row = i / p.W;                         // (n*J + j)*H + y
        const int y = (int)(row % p.H);
        const int64_t nj = row / p.H;
        const int j = (int)(nj % p.J);
        const int64_t n = nj / p.J;
        // flipped-back column x holds b's column W-1-x; the shift copies column x-1 into x for x >= 1 (column 0 stays)
        const int xs = (p.shift && x >= 1) ? x - 1 : x;
        const float f = p.b[((n * p.J + p.src[j]) * p.H + y) * p.W + (p.W - 1 - xs)];
        p.y[i] = p.a != nullptr ? (p.a[i] + f) * 0.5f : f;      // a == NULL: flip_back (+ shift) only
    }
}

__global__ __launch_bounds__(256) void final_preds_kernel(const fpd_finalpreds_t p) {
    __shared__ ArgMax s[4];
    const int n = blockIdx.x / p.J;
    const int HW = p.H * p.W;
    const float* hm = p.hm + (size_t)blockIdx.x * HW;
    ArgMax m = {-3.4e38f, 0x7fffffff};
    for (int q = threadIdx.x; q < HW; q += blockDim.x) {
        ArgMax t = {hm[q], q};
        m = better(m, t);
    }
    m = block_argmax(m, s);
    if (threadIdx.x != 0) return;
    float cx = m.v > 0.f ? (float)(m.i % p.W) : 0.f, cy = m.v > 0.f ? (float)(m.i / p.W) : 0.f;
    if (p.post_process) {                                   // inference.py:57-70
        const int px = (int)floorf(cx + 0.5f), py = (int)floorf(cy + 0.5f);
        if (1 < px && px < p.W - 1 && 1 < py && py < p.H - 1) {
            const float dx = hm[py * p.W + px + 1] - hm[py * p.W + px - 1];
            const float dy = hm[(py + 1) * p.W + px] - hm[(py - 1) * p.W + px];
            cx += (dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f));
            cy += (dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f));
        }
    }
    p.coords[2 * blockIdx.x] = cx;
    p.coords[2 * blockIdx.x + 1] = cy;
    p.maxvals[blockIdx.x] = m.v;
    if (p.trans != nullptr && p.preds != nullptr) {         // transforms.py:50-55,99-102 in float64
        const double* t = p.trans + 6 * n;
        // numpy.dot of a [2,3] matrix with a 3-vector: row . vector accumulated left to right
        const double X = (double)cx, Y = (double)cy;
        p.preds[2 * blockIdx.x] = (float)(t[0] * X + t[1] * Y + t[2] * 1.0);
        p.preds[2 * blockIdx.x + 1] = (float)(t[3] * X + t[4] * Y + t[5] * 1.0);
    }
}

}  // namespace

int fpd_flip_w_launch(const float* x, float* y, int64_t rows, int W, hipStream_t st) {
    const int64_t total = rows * W;
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    FPD_LAUNCH(flip_w_kernel, dim3(blocks), dim3(256), 0, st, x, y, rows, W);
    return 0;
}

int fpd_flip_merge_launch(const fpd_flipmerge_t& p, hipStream_t st) {
    const int64_t total = (int64_t)p.N * p.J * p.H * p.W;
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    FPD_LAUNCH(flip_merge_kernel, dim3(blocks), dim3(256), 0, st, p);
    return 0;
}

int fpd_final_preds_launch(const fpd_finalpreds_t& p, hipStream_t st) {
    FPD_LAUNCH(final_preds_kernel, dim3(p.N * p.J), dim3(256), 0, st, p);
    return 0;
}
