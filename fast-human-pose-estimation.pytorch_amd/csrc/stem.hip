// Stem convolution Conv2d(3, K, kernel 7, stride 2, pad 3) on the fp32 NCHW input image
// (/root/reference/lib/models/hourglass.py:116,172) and its weight gradient.  Cin = 3 is no MFMA shape,
// so these are direct kernels: an 8x16 output tile per step, the 21x37x3 input patch and the operand
// tile staged in LDS, outputs written NHWC.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TH = 8, TW = 16, TP = TH * TW;          // output tile
constexpr int PH = 2 * TH + 5, PW = 2 * TW + 5;       // input patch rows/cols
constexpr int NTAP = 147, KMAX = 64;

__device__ __forceinline__ void load_patch(float* patch, const float* x, int n, int H, int W, int py0, int px0, int tid) {
    for (int i = tid; i < 3 * PH * PW; i += 256) {
        const int c = i / (PH * PW), rem = i - c * (PH * PW);
        const int pr = rem / PW, pc = rem - pr * PW;
        const int ih = 2 * py0 - 3 + pr, iw = 2 * px0 - 3 + pc;
        float v = 0.f;
        if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[((size_t)(n * 3 + c) * H + ih) * W + iw];
        patch[i] = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const fpd_stem_t a) {
    __shared__ float patch[3 * PH * PW];
    __shared__ __attribute__((aligned(16))) float buf[NTAP * KMAX];  // weights [tap][64], later the output tile
    const int tid = threadIdx.x;
    const int K = a.K, P = a.P, Q = a.Q;
    const int tiles_x = cdiv_dev(Q, TW), tiles_y = cdiv_dev(P, TH);
    int b = blockIdx.x;
    const int tx0 = (b % tiles_x) * TW; b /= tiles_x;
    const int ty0 = (b % tiles_y) * TH; b /= tiles_y;
    const int n = b;

    for (int i = tid; i < NTAP * KMAX; i += 256) {
        const int tap = i / KMAX, k = i - tap * KMAX;
        buf[i] = (k < K) ? a.w[(size_t)k * NTAP + tap] : 0.f;
    }
    load_patch(patch, a.x, n, a.H, a.W, ty0, tx0, tid);
    __syncthreads();

    const int pix = tid & 127, half = tid >> 7;
    const int ty = pix / TW, tx = pix - ty * TW;
    const int kh = (K + 1) / 2;            // channels per half (K <= 64 -> kh <= 32)
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    const int pbase = (2 * ty) * PW + 2 * tx;
    for (int tap = 0; tap < NTAP; ++tap) {
        const int c = tap % 3, s = (tap / 3) % 7, r = tap / 21;
        const float xv = patch[c * PH * PW + r * PW + s + pbase];
        const float* wrow = buf + tap * KMAX + half * kh;
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < kh) acc[j] = fmaf(xv, wrow[j], acc[j]);
    }
    __syncthreads();   // everyone is done with the weights; reuse buf as the [128][K+1] output tile
    const int LDO = K + 1;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int k = half * kh + j;
        if (j < kh && k < K) buf[pix * LDO + k] = DT<T>::rnd(acc[j] + a.bias[k]);
    }
    __syncthreads();
    T* y = reinterpret_cast<T*>(a.y);
    for (int i = tid; i < TP * K; i += 256) {
        const int p = i / K, k = i - p * K;
        const int py = ty0 + p / TW, px = tx0 + p % TW;
        if (py < P && px < Q) DT<T>::st(y + ((size_t)(n * P + py) * Q + px) * K + k, buf[p * LDO + k]);
    }
    if (a.out_stats != nullptr && tid < K) {
        double s1 = 0.0, s2 = 0.0;
        for (int p = 0; p < TP; ++p) {
            const int py = ty0 + p / TW, px = tx0 + p % TW;
            if (py < P && px < Q) { const double v = (double)buf[p * LDO + tid]; s1 += v; s2 += v * v; }
        }
        stat_atomic_add(a.out_stats, K, 0, tid, s1);
        stat_atomic_add(a.out_stats, K, 1, tid, s2);
    }
}

// dw[k][tap] += sum_pixels dy[pix][k] * x[patch(pix, tap)] ; dbias[k] += sum dy.  Each block walks a strided
// set of output tiles and flushes once with fp32 atomics.
template <typename T>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const fpd_stem_t a, const int ntiles) {
    __shared__ float patch[3 * PH * PW];
    __shared__ float dys[TP * KMAX];
    const int tid = threadIdx.x;
    const int K = a.K, P = a.P, Q = a.Q;
    const int tiles_x = cdiv_dev(Q, TW), tiles_y = cdiv_dev(P, TH);
    const int TG = 256 / K;                // tap groups (K in {16,32,64} -> 16,8,4)
    const int k = tid % K, tg = tid / K;
    const bool active = tg < TG;
    constexpr int MAXNT = 37;              // ceil(147/4)
    float acc[MAXNT];
    int off[MAXNT];
#pragma unroll
    for (int i = 0; i < MAXNT; ++i) {
        acc[i] = 0.f;
        const int tap = tg + i * TG;
        const int c = tap % 3, s = (tap / 3) % 7, r = tap / 21;
        off[i] = (active && tap < NTAP) ? c * PH * PW + r * PW + s : -1;
    }
    float bsum = 0.f;
    const T* dy = reinterpret_cast<const T*>(a.dy);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b = tile;
        const int tx0 = (b % tiles_x) * TW; b /= tiles_x;
        const int ty0 = (b % tiles_y) * TH; b /= tiles_y;
        const int n = b;
        __syncthreads();
        load_patch(patch, a.x, n, a.H, a.W, ty0, tx0, tid);
        for (int i = tid; i < TP * K; i += 256) {
            const int p = i / K, kk = i - p * K;
            const int py = ty0 + p / TW, px = tx0 + p % TW;
            dys[i] = (py < P && px < Q) ? DT<T>::ld(dy + ((size_t)(n * P + py) * Q + px) * K + kk) : 0.f;
        }
        __syncthreads();
        if (active) {
            for (int p = 0; p < TP; ++p) {
                const float g = dys[p * K + k];
                const int pbase = (2 * (p / TW)) * PW + 2 * (p % TW);
                if (tg == 0) bsum += g;
#pragma unroll
                for (int i = 0; i < MAXNT; ++i)
                    if (off[i] >= 0) acc[i] = fmaf(g, patch[off[i] + pbase], acc[i]);
            }
        }
    }
    if (active) {
        // no atomics: persistent block b -> slab b (a.partial), or a single block adding into dw directly
        float* slab = a.partial != nullptr ? a.partial + (size_t)blockIdx.x * a.partial_stride : nullptr;
#pragma unroll
        for (int i = 0; i < MAXNT; ++i) {
            const int tap = tg + i * TG;
            if (off[i] >= 0) {
                if (slab != nullptr) slab[(size_t)k * NTAP + tap] = acc[i];
                else a.dw[(size_t)k * NTAP + tap] += acc[i];
            }
        }
        if (tg == 0 && a.dbias != nullptr) {
            if (slab != nullptr) slab[(size_t)K * NTAP + k] = bsum;
            else a.dbias[k] += bsum;
        }
    }
}

}  // namespace

int fpd_stem_forward_launch(const fpd_stem_t& a, hipStream_t st) {
    if (a.K > KMAX || a.K < 1) return fpd_fail(-3, "stem: K=%d unsupported (1..%d)", a.K, KMAX);
    const int tiles = a.N * cdiv(a.P, TH) * cdiv(a.Q, TW);
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((stem_fwd_kernel<bf16_t>), dim3(tiles), dim3(256), 0, st, a);
    else
        FPD_LAUNCH((stem_fwd_kernel<float>), dim3(tiles), dim3(256), 0, st, a);
    return 0;
}

int fpd_stem_wgrad_partials(const fpd_stem_t& a) {
    if (a.K > KMAX || 256 % a.K != 0) return 0;
    return std::min(a.N * cdiv(a.P, TH) * cdiv(a.Q, TW), 1024);
}

int fpd_stem_wgrad_launch(const fpd_stem_t& a, hipStream_t st) {
    if (a.K > KMAX || 256 % a.K != 0) return fpd_fail(-3, "stem wgrad: K=%d must divide 256 and be <= %d", a.K, KMAX);
    const int tiles = a.N * cdiv(a.P, TH) * cdiv(a.Q, TW);
    const int grid = a.partial != nullptr ? std::min(tiles, 1024) : 1;
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((stem_wgrad_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a, tiles);
    else
        FPD_LAUNCH((stem_wgrad_kernel<float>), dim3(grid), dim3(256), 0, st, a, tiles);
    return 0;
}
