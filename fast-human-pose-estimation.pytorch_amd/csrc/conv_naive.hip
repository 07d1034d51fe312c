// Direct (one thread per output) HIP convolution kernels with the same fused prologue/epilogue contract
// as conv_mfma.hip.  They serve as the on-device cross-check of the MFMA kernels, and as the HIP path for
// shapes the MFMA tiling does not cover (channel counts that are not a multiple of 16, e.g. J = 17).
#include <algorithm>

#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void conv_naive_kernel(const fpd_conv_t a) {
    __shared__ float s_scale[FPD_MAXC], s_shift[FPD_MAXC];
    __shared__ float s_epi[4][FPD_MAXC];
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, S = a.S, P = a.P, Q = a.Q;
    const int M = a.N * P * Q;
    bn_fill(a.bn, C, (double)a.N * H * W, s_scale, s_shift);
    if (a.epi == FPD_EPI_BNRELU_BWD) {
        for (int k = threadIdx.x; k < K; k += blockDim.x)
            bn_coef(a.epi_bn, k, K, (double)M, s_epi[0][k], s_epi[1][k], s_epi[2][k], s_epi[3][k]);
    }
    __syncthreads();
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* w = reinterpret_cast<const T*>(a.w);
    const T* res = reinterpret_cast<const T*>(a.residual);
    const T* ex = reinterpret_cast<const T*>(a.epi_x);
    T* y = reinterpret_cast<T*>(a.y);
    const size_t total = (size_t)M * K;
    // statistics: when the grid stride is a multiple of K a thread sees ONE output channel in all its iterations, so it keeps
    // its two sums in registers (fp64) and the block adds them per channel at the end (LDS, then one atomic pair per channel
    // per block) -- per-element atomics otherwise (tiny / odd K only)
    const bool want_stats = a.epi == FPD_EPI_BNRELU_BWD || a.out_stats != nullptr;
    const bool fast = want_stats && ((size_t)gridDim.x * blockDim.x) % (size_t)K == 0 && K <= FPD_MAXC;
    __shared__ long long s_acc[4 * FPD_MAXC];          // [2 sums][2 limbs][K] exact accumulator (common.h)
    if (fast) {
        for (int k = threadIdx.x; k < 4 * K; k += blockDim.x) s_acc[k] = 0;
        __syncthreads();
    }
    double t1 = 0.0, t2 = 0.0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / K), k = (int)(idx - (size_t)m * K);
        const int n = m / (P * Q), rem = m - n * (P * Q);
        const int p = rem / Q, q = rem - p * Q;
        float acc = 0.f;
        for (int r = 0; r < R; ++r) {
            const int ih = p * a.stride - a.pad + r;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int s = 0; s < S; ++s) {
                const int iw = q * a.stride - a.pad + s;
                if ((unsigned)iw >= (unsigned)W) continue;
                const T* xp = x + (size_t)((n * H + ih) * W + iw) * C;
                const T* wp = w + (size_t)((k * R + r) * S + s) * C;
                for (int c = 0; c < C; ++c) {
                    float xv = DT<T>::ld(xp + c);
                    if (a.bn.mode != FPD_BN_NONE) xv = DT<T>::rnd(bn_act(xv, s_scale[c], s_shift[c], a.bn.relu));
                    acc = fmaf(xv, DT<T>::ld(wp + c), acc);
                }
            }
        }
        float v = acc + (a.bias ? a.bias[k] : 0.f);
        if (res) v += DT<T>::ld(res + idx);
        if (a.epi == FPD_EPI_BNRELU_BWD) {
            const float xv = DT<T>::ld(ex + idx);
            const float z = fmaf(xv, s_epi[0][k], s_epi[1][k]);
            v = (!a.epi_bn.relu || z > 0.f) ? v : 0.f;
            const float vr = DT<T>::rnd(v);
            const double u2 = (double)(vr * ((xv - s_epi[2][k]) * s_epi[3][k]));
            if (fast) { t1 += (double)vr; t2 += u2; }
            else {
                stat_atomic_add(a.epi_stats, K, 0, k, (double)vr);
                stat_atomic_add(a.epi_stats, K, 1, k, u2);
            }
        } else if (a.out_stats) {
            const float vr = DT<T>::rnd(v);
            if (fast) { t1 += (double)vr; t2 += (double)vr * (double)vr; }
            else {
                stat_atomic_add(a.out_stats, K, 0, k, (double)vr);
                stat_atomic_add(a.out_stats, K, 1, k, (double)(vr * vr));
            }
        }
        DT<T>::st(y + idx, v);
    }
    if (fast) {
        const int k = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % (size_t)K);
        stat_lds_add(s_acc, K, 0, k, t1);
        stat_lds_add(s_acc, K, 1, k, t2);
        __syncthreads();
        fpd_stat_t* dst = a.epi == FPD_EPI_BNRELU_BWD ? a.epi_stats : a.out_stats;
        for (int c = threadIdx.x; c < K; c += blockDim.x) stat_flush_lds(dst, s_acc, K, c);
    }
}

// one thread per weight element and pixel chunk (blockIdx.y): serial loop over the chunk's pixels, one plain store per
// thread at the end (to its chunk's slab, or += into dw when the launch has a single chunk).  Odd shapes: channel counts that are not multiples
// of 16 -- the 3-channel first convolution of HRNet (pose_hrnet.py:279), J = 17 heads, the scaled-down test networks.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_naive_kernel(const fpd_wgrad_t a) {
    __shared__ float s_scale[FPD_MAXC], s_shift[FPD_MAXC];
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, S = a.S, P = a.P, Q = a.Q;
    const int M = a.N * P * Q;
    bn_fill(a.bn, C, (double)a.N * H * W, s_scale, s_shift);
    __syncthreads();
    const T* x = reinterpret_cast<const T*>(a.x);
    const T* dy = reinterpret_cast<const T*>(a.dy);
    const int total = K * R * S * C;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % C, s = (idx / C) % S, r = (idx / (C * S)) % R, k = idx / (C * S * R);
    float acc = 0.f, bsum = 0.f;
    const int per = (M + (int)gridDim.y - 1) / (int)gridDim.y;
    const int m_begin = (int)blockIdx.y * per, m_end = min(M, m_begin + per);
    for (int m = m_begin; m < m_end; ++m) {
        const float g = DT<T>::ld(dy + (size_t)m * K + k);
        bsum += g;
        const int n = m / (P * Q), rem = m - n * (P * Q);
        const int p = rem / Q, q = rem - p * Q;
        const int ih = p * a.stride - a.pad + r, iw = q * a.stride - a.pad + s;
        if ((unsigned)ih >= (unsigned)H || (unsigned)iw >= (unsigned)W) continue;
        float xv = DT<T>::ld(x + (size_t)((n * H + ih) * W + iw) * C + c);
        if (a.bn.mode != FPD_BN_NONE) xv = DT<T>::rnd(bn_act(xv, s_scale[c], s_shift[c], a.bn.relu));
        acc = fmaf(g, xv, acc);
    }
    // no atomics: pixel chunk blockIdx.y stores to slab blockIdx.y (a.partial, summed by fpd_wgrad_reduce() in a fixed
    // order); without slabs the launch has one chunk and adds into dw directly
    if (a.partial != nullptr) {
        float* slab = a.partial + (size_t)blockIdx.y * a.partial_stride;
        slab[idx] = acc;
        if (a.dbias && c == 0 && r == 0 && s == 0) slab[total + k] = bsum;
    } else {
        a.dw[idx] += acc;
        if (a.dbias && c == 0 && r == 0 && s == 0) a.dbias[k] += bsum;
    }
}

}  // namespace

int fpd_conv_naive_launch(const fpd_conv_t& a, hipStream_t st) {
    if (a.C > FPD_MAXC || a.K > FPD_MAXC) return fpd_fail(-3, "conv: channel count above %d", FPD_MAXC);
    const size_t total = (size_t)a.N * a.P * a.Q * a.K;
    const bool stats = a.out_stats != nullptr || a.epi == FPD_EPI_BNRELU_BWD;
    const int grid = (int)std::min<size_t>((total + 255) / 256, stats ? 2048 : 65536);    // statistics: few, long-lived blocks
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((conv_naive_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a);
    else
        FPD_LAUNCH((conv_naive_kernel<float>), dim3(grid), dim3(256), 0, st, a);
    return 0;
}

static int wgrad_naive_chunks(const fpd_wgrad_t& a) {
    const int M = a.N * a.P * a.Q;
    return std::max(1, std::min(1024, M / 512));                      // >= 512 pixels per thread
}
int fpd_wgrad_naive_partials(const fpd_wgrad_t& a) { return a.C > FPD_MAXC ? 0 : wgrad_naive_chunks(a); }

int fpd_wgrad_naive_launch(const fpd_wgrad_t& a, hipStream_t st) {
    if (a.C > FPD_MAXC) return fpd_fail(-3, "wgrad: channel count above %d", FPD_MAXC);
    const int total = a.K * a.R * a.S * a.C;
    const int chunks = a.partial != nullptr ? wgrad_naive_chunks(a) : 1;
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((wgrad_naive_kernel<bf16_t>), dim3(cdiv(total, 256), chunks), dim3(256), 0, st, a);
    else
        FPD_LAUNCH((wgrad_naive_kernel<float>), dim3(cdiv(total, 256), chunks), dim3(256), 0, st, a);
    return 0;
}
