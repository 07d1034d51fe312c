// Per-step training metric on the device: heat-map arg-max + PCK@thr of the last student map against the target
// (/root/reference/lib/core/inference.py:18-46 get_max_preds, lib/core/evaluate.py:16-71 calc_dists/dist_acc/accuracy;
// the reference copies both tensors to the host and runs numpy every iteration, function.py:154-155).
//   pck_kernel    one block per (sample, joint): arg-max of the NHWC prediction and of the NCHW target (first maximum
//                 wins, like numpy.argmax), normalised distance, -> counts[joint] += {hit, valid}
//   pck_finish    one block: per-joint accuracy = hits/valid (joints with no valid sample are skipped), average over the
//                 remaining joints -> log[slot] = {avg_acc, cnt, pose, kd} (fp64, the reference's arithmetic type; pose/kd
//                 copied from the fused loss kernel's accumulators so that EVERY iteration feeds the loss meters like
//                 function.py:150-152 without a host sync); counts are zeroed for the next iteration
// Index work is bit-exact with the reference, quirk included: evaluate.py:55 builds the normaliser as [h, w]/10 and
// applies it to (x, y), i.e. x is divided by h/10 and y by w/10 (identical for square maps, different at 64x48).
#include "argmax.h"
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pck_kernel(const fpd_pck_t a) {
    __shared__ ArgMax s[4];
    const int b = blockIdx.x / a.J, j = blockIdx.x - b * a.J;
    const int HW = a.H * a.W;
    const T* out = reinterpret_cast<const T*>(a.out) + (size_t)b * HW * a.J + j;      // [B][H][W][J]
    const float* tg = a.target + ((size_t)b * a.J + j) * HW;                           // [B][J][H][W]
    ArgMax mp = {-3.4e38f, 0x7fffffff}, mg = {-3.4e38f, 0x7fffffff};
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        ArgMax t = {DT<T>::ld(out + (size_t)p * a.J), p};
        mp = better(mp, t);
        ArgMax g = {tg[p], p};
        mg = better(mg, g);
    }
    mp = block_argmax(mp, s);
    mg = block_argmax(mg, s);
    if (threadIdx.x == 0) {
        // get_max_preds: coordinates are zeroed where the maximum is not positive
        const float px = mp.v > 0.f ? (float)(mp.i % a.W) : 0.f, py = mp.v > 0.f ? (float)(mp.i / a.W) : 0.f;
        const float gx = mg.v > 0.f ? (float)(mg.i % a.W) : 0.f, gy = mg.v > 0.f ? (float)(mg.i / a.W) : 0.f;
        if (gx > 1.f && gy > 1.f) {                     // calc_dists: only targets away from the top-left corner count
            // evaluate.py:23-26,55: float32 coordinates / float64 normaliser [h, w]/10 -> float64 norm of the difference
            const double nx = (double)a.H / 10.0, ny = (double)a.W / 10.0;
            const double dx = (double)px / nx - (double)gx / nx, dy = (double)py / ny - (double)gy / ny;
            const double d = sqrt(dx * dx + dy * dy);
            atomicAdd(a.counts + 2 * j + 1, 1.f);
            if (d < (double)a.thr) atomicAdd(a.counts + 2 * j, 1.f);
        }
    }
}

__global__ void pck_finish_kernel(const fpd_pck_t a) {
    if (threadIdx.x != 0) return;
    double sum = 0.0;                                   // evaluate.py:33-39,62-68 in its own (float64) arithmetic
    int cnt = 0;
    for (int j = 0; j < a.J; ++j) {
        const float hit = a.counts[2 * j], valid = a.counts[2 * j + 1];
        if (valid > 0.f) { sum += (double)hit * 1.0 / (double)valid; ++cnt; }
        a.counts[2 * j] = 0.f;
        a.counts[2 * j + 1] = 0.f;
    }
    const long long k = *a.cursor;
    double* dst = a.log + 4 * (k % a.log_slots);
    dst[0] = cnt ? sum / cnt : 0.0;
    dst[1] = (double)cnt;
    dst[2] = a.losses ? a.losses[0] : 0.0;
    dst[3] = a.losses ? a.losses[1] : 0.0;
    *a.cursor = k + 1;
}

}  // namespace

int fpd_pck_launch(const fpd_pck_t& a, hipStream_t st) {
    if (a.dtype == FPD_BF16) FPD_LAUNCH(pck_kernel<bf16_t>, dim3(a.B * a.J), dim3(256), 0, st, a);
    else FPD_LAUNCH(pck_kernel<float>, dim3(a.B * a.J), dim3(256), 0, st, a);
    FPD_LAUNCH(pck_finish_kernel, dim3(1), dim3(64), 0, st, a);
    return 0;
}
