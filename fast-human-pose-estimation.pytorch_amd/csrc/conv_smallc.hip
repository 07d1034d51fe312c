// conv_smallc: direct convolution for the few layers whose INPUT channel count is tiny and not a multiple of 16 -- HRNet's
// first stem convolution (3x3 stride 2, 3 -> 64 channels, pose_hrnet.py:279) and the data gradient of its final layer
// (1x1, J = 17 -> 32/48 channels, pose_hrnet.py:320-326) -- which the MFMA kernels do not tile and the one-thread-per-output
// cross-check kernel (conv_naive.hip) runs at ~1 ms per launch (r02_hrnet_per_shape.csv: 3.3 ms of a 29 ms step).
//   one thread = one output pixel; its R*S*C inputs sit in registers, the weights of a 64-channel output chunk in LDS as
//   fp32 [64][NTP] read as broadcast 16-byte vectors; eight output channels at a time (8 accumulators -> one 16-byte store:
//   a pixel's channels are contiguous in NHWC); batch statistics of the rounded outputs by a butterfly that halves the
//   values per lane for three steps and reduces the last value over the remaining lane bits, fp64 from the wave total on.
// Contract of fpd_conv_t restricted to: no BN prologue, plain epilogue, C*R*S <= 64, K a multiple of 8.
#include <algorithm>

#include "common.h"

namespace {

template <typename T, int R, int C>
__global__ __launch_bounds__(256) void conv_smallc_kernel(const fpd_conv_t a) {
    constexpr int NT = R * R * C, NTP = (NT + 3) & ~3, VEC = DT<T>::VEC;
    __shared__ __attribute__((aligned(16))) float s_w[64 * NTP];
    __shared__ long long s_acc[4 * 64];            // [2 sums][2 limbs][64] exact accumulator (common.h)
    const int tid = threadIdx.x, lane = tid & 63;
    const int H = a.H, W = a.W, K = a.K, P = a.P, Q = a.Q;
    const int M = a.N * P * Q;
    const int m = blockIdx.x * 256 + tid;
    const bool live = m < M;
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
    const T* res = reinterpret_cast<const T*>(a.residual);
    T* __restrict__ y = reinterpret_cast<T*>(a.y);

    float in[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) in[t] = 0.f;
    if (live) {
        const int n = m / (P * Q), rem = m - n * (P * Q);
        const int p = rem / Q, q = rem - p * Q;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int ih = p * a.stride - a.pad + r;
#pragma unroll
            for (int s = 0; s < R; ++s) {
                const int iw = q * a.stride - a.pad + s;
                if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) {
                    const T* xp = x + (size_t)((n * H + ih) * W + iw) * C;
#pragma unroll
                    for (int c = 0; c < C; ++c) in[(r * R + s) * C + c] = DT<T>::ld(xp + c);
                }
            }
        }
    }
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int kn = min(64, K - k0);
        __syncthreads();                                   // previous chunk's weights / statistics consumed
        for (int i = tid; i < 64 * NTP; i += 256) {
            const int k = i / NTP, t = i - k * NTP;
            s_w[i] = (k < kn && t < NT) ? DT<T>::ld(w + (size_t)(k0 + k) * NT + t) : 0.f;
        }
        if (tid < 256) s_acc[tid] = 0;
        __syncthreads();
        // eight output channels at a time (one 16-byte vector of bf16): 8 accumulators, their weights as broadcast vectors
#pragma unroll 1
        for (int g = 0; g < 8; ++g) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = 0.f;
                const float* wk = s_w + (g * 8 + j) * NTP;
#pragma unroll
                for (int t4 = 0; t4 < NTP / 4; ++t4) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wk + t4 * 4);       // same address in every lane: broadcast
                    v = fmaf(in[t4 * 4 + 0], wv[0], v);
                    v = fmaf(in[t4 * 4 + 1], wv[1], v);
                    v = fmaf(in[t4 * 4 + 2], wv[2], v);
                    v = fmaf(in[t4 * 4 + 3], wv[3], v);
                }
                acc[j] = v;
            }
            const int kb = g * 8;                          // first channel of the group within the chunk
            const bool vok = live && kb < kn;              // K % 8 == 0: a group is entirely inside or outside
            // bias, residual, one rounding, 16-byte stores (a pixel's channels are contiguous)
            if (res != nullptr && vok) {
                float rr[8];
                if constexpr (VEC == 8) {
                    DT<T>::unpack(*reinterpret_cast<const uint4*>(res + (size_t)m * K + k0 + kb), rr);
                } else {
                    DT<T>::unpack(*reinterpret_cast<const uint4*>(res + (size_t)m * K + k0 + kb), rr);
                    DT<T>::unpack(*reinterpret_cast<const uint4*>(res + (size_t)m * K + k0 + kb + 4), rr + 4);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += rr[j];
            }
            if (a.bias != nullptr && kb < kn) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += a.bias[k0 + kb + j];
            }
            if constexpr (VEC == 8) {
                const uint4 pk = DT<T>::pack(acc);
                if (vok) *reinterpret_cast<uint4*>(y + (size_t)m * K + k0 + kb) = pk;
                DT<T>::unpack(pk, acc);                    // the statistics are those of the STORED values
            } else {
                if (vok) {
                    *reinterpret_cast<uint4*>(y + (size_t)m * K + k0 + kb) = DT<T>::pack(acc);
                    *reinterpret_cast<uint4*>(y + (size_t)m * K + k0 + kb + 4) = DT<T>::pack(acc + 4);
                }
            }
            if (a.out_stats != nullptr) {
                // per-channel sums over the 64 pixels of the wave: three halving steps (8 -> 1 value per lane, the channel
                // is then (lane >> 3) & 7), three plain steps over the remaining lane bits
                float s1[8], s2[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1[j] = vok ? acc[j] : 0.f; s2[j] = s1[j] * s1[j]; }
#pragma unroll
                for (int h = 4; h >= 1; h >>= 1) {
                    const int bit = h * 8;                 // lane bits 5, 4, 3
                    const bool up = (lane & bit) != 0;
#pragma unroll
                    for (int j = 0; j < h; ++j) {
                        const float k1 = up ? s1[j + h] : s1[j], g1 = up ? s1[j] : s1[j + h];
                        const float k2 = up ? s2[j + h] : s2[j], g2 = up ? s2[j] : s2[j + h];
                        s1[j] = k1 + __shfl_xor(g1, bit, 64);
                        s2[j] = k2 + __shfl_xor(g2, bit, 64);
                    }
                }
#pragma unroll
                for (int o = 4; o >= 1; o >>= 1) {
                    s1[0] += __shfl_xor(s1[0], o, 64);
                    s2[0] += __shfl_xor(s2[0], o, 64);
                }
                if ((lane & 7) == 0) {
                    const int ch = kb + ((lane >> 3) & 7);
                    stat_lds_add(s_acc, 64, 0, ch, (double)s1[0]);
                    stat_lds_add(s_acc, 64, 1, ch, (double)s2[0]);
                }
            }
        }
        if (a.out_stats != nullptr) {
            __syncthreads();
            if (tid < kn) {
                unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.out_stats) + (size_t)stats_replica() * 4 * K + k0 + tid;
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicAdd(dst + q * K, (unsigned long long)s_acc[q * 64 + tid]);
            }
        }
    }
}

template <typename T, int R, int C>
int launch_smallc(const fpd_conv_t& a, hipStream_t st) {
    const int M = a.N * a.P * a.Q;
    FPD_LAUNCH((conv_smallc_kernel<T, R, C>), dim3(cdiv(M, 256)), dim3(256), 0, st, a);
    return 0;
}

template <typename T>
int dispatch_smallc(const fpd_conv_t& a, hipStream_t st) {
    if (a.R == 3 && a.C == 3) return launch_smallc<T, 3, 3>(a, st);
    if (a.R == 1 && a.C == 17) return launch_smallc<T, 1, 17>(a, st);
    if (a.R == 1 && a.C == 5) return launch_smallc<T, 1, 5>(a, st);       // the scaled-down test networks' joint count
    return 1;
}

// ---- weight gradient of a convolution with a tiny reduction footprint per pixel (R*S*C <= 32): HRNet's first stem
// convolution (3x3 / stride 2, C = 3, K = 64: pose_hrnet.py:279), whose weight gradient is the LAST launch of the backward
// and therefore fully exposed in front of Adam (1.19 ms through the one-thread-per-weight kernel).
//   dw[k][r][s][c] = sum_m dy[m][k] * x[m*stride - pad + (r, s)][c]
// A 256-thread block walks 128-pixel tiles persistently: the dy tile goes to LDS as fp32 [128][64], the im2col rows of the
// tile as fp32 [128][32] (zero where the tap falls outside the image); thread (k = tid % 64, tg = tid / 64) owns the eight
// im2col columns 8 tg .. 8 tg + 7 of output channel k: per pixel one conflict-free read of dy and two 16-byte reads of
// im2col values whose address is uniform over the wave (broadcast).  One slab per block at the end, no atomics.
constexpr int WS_TP = 128, WS_COLS = 32, WS_KB = 64;

template <typename T>
__global__ __launch_bounds__(256) void wgrad_smallc_kernel(const fpd_wgrad_t a, const int ntiles) {
    constexpr int VEC = DT<T>::VEC;
    __shared__ __attribute__((aligned(16))) float s_dy[WS_TP * WS_KB];
    __shared__ __attribute__((aligned(16))) float s_col[WS_TP * WS_COLS];
    __shared__ int s_pix[WS_TP * 3];                    // image, first input row, first input column of every tile pixel
    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, S = a.S, P = a.P, Q = a.Q;
    const int M = a.N * P * Q, RSC = R * S * C, PQ = P * Q;
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);
    const int k = tid & (WS_KB - 1), tg = tid >> 6;
    // im2col column t -> (r, s, c), fixed per gather slot of this thread: slot i handles element tid + 256 i of [128][32]
    int g_r[16], g_s[16], g_c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = (tid + 256 * i) & (WS_COLS - 1);
        const int rs = t / C;
        g_c[i] = t - rs * C; g_r[i] = rs / S; g_s[i] = rs - (rs / S) * S;
        if (t >= RSC) g_r[i] = -(1 << 20);              // padding columns stay zero
    }
    float acc[8], bsum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const int kvec = K / VEC;                           // 16-byte vectors per dy pixel (K % VEC == 0, K <= 64)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * WS_TP;
        __syncthreads();                                // previous tile consumed
        if (tid < WS_TP) {
            const int m = m0 + tid;
            const int n = m / PQ, rem = m - n * PQ;
            const int pp = rem / Q, q = rem - pp * Q;
            s_pix[tid * 3] = m < M ? n : -1;
            s_pix[tid * 3 + 1] = pp * a.stride - a.pad;
            s_pix[tid * 3 + 2] = q * a.stride - a.pad;
        }
        for (int v = tid; v < WS_TP * (WS_KB / VEC); v += 256) {       // dy tile, zero beyond K / M
            const int p = v / (WS_KB / VEC), cv = v - p * (WS_KB / VEC);
            float f[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) f[e] = 0.f;
            if (m0 + p < M && cv < kvec) {
                if (VEC == 8) DT<T>::unpack(*reinterpret_cast<const uint4*>(dy + (size_t)(m0 + p) * K + cv * VEC), f);
                else {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(dy + (size_t)(m0 + p) * K + cv * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = t4[e];
                }
            }
#pragma unroll
            for (int e = 0; e < VEC; e += 4)
                *reinterpret_cast<f32x4*>(s_dy + p * WS_KB + cv * VEC + e) = f32x4{f[e], f[e + 1], f[e + 2], f[e + 3]};
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {                  // im2col gather
            const int e = tid + 256 * i, p = e >> 5;
            const int n = s_pix[p * 3], ih = s_pix[p * 3 + 1] + g_r[i], iw = s_pix[p * 3 + 2] + g_s[i];
            float v = 0.f;
            if (n >= 0 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
                v = DT<T>::ld(x + ((size_t)(n * H + ih) * W + iw) * C + g_c[i]);
            s_col[e] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < WS_TP; ++p) {
            const float g = s_dy[p * WS_KB + k];
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(s_col + p * WS_COLS + tg * 8);
            const f32x4 c1 = *reinterpret_cast<const f32x4*>(s_col + p * WS_COLS + tg * 8 + 4);
            bsum += g;
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] = fmaf(g, c0[j], acc[j]); acc[4 + j] = fmaf(g, c1[j], acc[4 + j]); }
        }
    }
    // flush: slab blockIdx.x (layout of dw, bias sums behind it) or, single block without slabs, straight into dw
    if (k < K) {
        float* dst = a.partial != nullptr ? a.partial + (size_t)blockIdx.x * a.partial_stride : a.dw;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = tg * 8 + j;
            if (t < RSC) {
                if (a.partial != nullptr) dst[k * RSC + t] = acc[j];
                else dst[k * RSC + t] += acc[j];
            }
        }
        if (a.dbias != nullptr && tg == 0) {
            if (a.partial != nullptr) dst[(size_t)K * RSC + k] = bsum;
            else a.dbias[k] += bsum;
        }
    }
}

static bool wgrad_smallc_domain(const fpd_wgrad_t& a) {
    if (a.bn.mode != FPD_BN_NONE || a.R * a.S * a.C > WS_COLS || a.K > WS_KB) return false;
    return a.K % (a.dtype == FPD_BF16 ? 8 : 4) == 0;
}
static int wgrad_smallc_blocks(const fpd_wgrad_t& a) {
    return std::max(1, std::min(512, cdiv(a.N * a.P * a.Q, WS_TP)));
}

}  // namespace

// slabs a launch writes (0 = outside the domain); 1 = fpd_wgrad_smallc_launch declined
int fpd_wgrad_smallc_partials(const fpd_wgrad_t& a) { return wgrad_smallc_domain(a) ? wgrad_smallc_blocks(a) : 0; }

int fpd_wgrad_smallc_launch(const fpd_wgrad_t& a, hipStream_t st) {
    if (!wgrad_smallc_domain(a)) return 1;
    const int ntiles = cdiv(a.N * a.P * a.Q, WS_TP);
    const int blocks = a.partial != nullptr ? wgrad_smallc_blocks(a) : 1;     // no slabs: one block owns every sum (deterministic, slow)
    if (a.dtype == FPD_BF16) FPD_LAUNCH((wgrad_smallc_kernel<bf16_t>), dim3(blocks), dim3(256), 0, st, a, ntiles);
    else FPD_LAUNCH((wgrad_smallc_kernel<float>), dim3(blocks), dim3(256), 0, st, a, ntiles);
    return 0;
}

// 1 = outside this kernel's domain (the caller falls through to the direct cross-check kernel)
int fpd_conv_smallc_launch(const fpd_conv_t& a, hipStream_t st) {
    if (a.R != a.S || a.bn.mode != FPD_BN_NONE || a.epi != FPD_EPI_PLAIN) return 1;
    if (a.K % 8 != 0 || a.K > FPD_MAXC) return 1;          // the kernel works on groups of eight output channels
    return a.dtype == FPD_BF16 ? dispatch_smallc<bf16_t>(a, st) : dispatch_smallc<float>(a, st);
}
