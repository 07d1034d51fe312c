// Fused JointsMSELoss (pose + distillation, all stacks, forward + gradient in one pass), flat Adam,
// weight working-copy preparation, BN running-statistics update, casts and layout changes.
// Replaces /root/reference/lib/core/loss.py:21-39 as called from lib/core/function.py:128-134 (2*S
// criterion calls + autograd), torch.optim.Adam (lib/utils/utils.py:69-73) and the running-stat side
// effect of nn.BatchNorm2d(momentum=0.1) (lib/models/hourglass.py:10).
#include <algorithm>

#include "common.h"

namespace {

// Block = 64 consecutive pixels of one image x all J joints.  The fp32 target arrives in the loader's
// NCHW layout; it is read coalesced along hw and turned through LDS so the NHWC maps are read/written
// with consecutive lanes on consecutive joints.
template <typename T>
__global__ __launch_bounds__(256) void loss_kernel(const fpd_loss_t a) {
    constexpr int PT = 64;
    __shared__ float s_tg[PT * 33];       // [pixel][joint] with J <= 32, padded rows
    __shared__ double s_acc[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = a.J, HW = a.H * a.W;
    const int tiles_per_img = (HW + PT - 1) / PT;
    const int b = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - b * tiles_per_img) * PT;
    const int LDJ = J + 1;
    if (a.target_nchw) {
        for (int i = tid; i < J * PT; i += 256) {
            const int j = i / PT, p = i - j * PT;
            s_tg[p * LDJ + j] = (p0 + p < HW) ? a.target[((size_t)b * J + j) * HW + p0 + p] : 0.f;
        }
    } else {
        for (int i = tid; i < J * PT; i += 256) {
            const int p = i / J, j = i - p * J;
            s_tg[p * LDJ + j] = (p0 + p < HW) ? a.target[((size_t)b * HW + p0 + p) * J + j] : 0.f;
        }
    }
    __syncthreads();
    const double cnt = (double)a.B * J * HW;
    const float gs = a.grad_scale / (float)cnt;
    const T* tch = reinterpret_cast<const T*>(a.teacher);
    float pose = 0.f, kd = 0.f;
    for (int i = tid; i < J * PT; i += 256) {
        const int p = i / J, j = i - p * J;
        if (p0 + p >= HW) continue;
        const size_t off = ((size_t)b * HW + p0 + p) * J + j;
        const float wgt = a.weight[b * J + j], w2 = wgt * wgt;
        const float wk = a.weight_kd ? a.weight_kd[b * J + j] : wgt, w2k = wk * wk;
        const float g = s_tg[p * LDJ + j];
        const float t = DT<T>::ld(tch + off);
        for (int s = 0; s < a.S; ++s) {
            const float pv = DT<T>::ld(reinterpret_cast<const T*>(a.out[s]) + off);
            const float dg = pv - g, dt = pv - t;
            pose += w2 * dg * dg;
            kd += w2k * dt * dt;
            if (a.dout[s] != nullptr)
                DT<T>::st(reinterpret_cast<T*>(a.dout[s]) + off, gs * ((1.f - a.alpha) * w2 * dg + a.alpha * w2k * dt));
        }
    }
    const double dp = wave_sum_d((double)pose), dk = wave_sum_d((double)kd);
    if (lane == 0) { s_acc[0][wave] = dp; s_acc[1][wave] = dk; }
    __syncthreads();
    if (tid == 0) {
        // Order-independent sum over the blocks (bit-repeatable losses): every contribution is rounded to a multiple of
        // 2^-44 (5.7e-14), so the fp64 additions are EXACT while the loss stays below 2^9 -- whatever order they arrive in.
        const double sc = 0.5 / cnt, q = 17592186044416.0;     // 2^44
        const double vp = sc * (s_acc[0][0] + s_acc[0][1] + s_acc[0][2] + s_acc[0][3]);
        const double vk = sc * (s_acc[1][0] + s_acc[1][1] + s_acc[1][2] + s_acc[1][3]);
        atomicAdd(a.losses + 0, rint(vp * q) / q);
        atomicAdd(a.losses + 1, rint(vk * q) / q);
    }
}

// Vectorised variant (J a multiple of the 16-byte vector: 8 joints in bf16, 4 in fp32): block = 128 consecutive pixels of one
// image; a thread owns ONE 16-byte vector of joints of one pixel for all S stacks -- one vector load per map, one vector store
// per gradient (the element-per-thread kernel above moves 2-byte words: 65 us for the 2 M elements of the benchmark, on the
// student's critical chain).  Same arithmetic per element, same per-block sums (fixed order), same grid-aligned loss terms.
// Round 5: a block walks `tpb` tiles of its image and adds ONE pair of loss terms at the end.  With a block per tile the 2 x 1 024
// fp64 atomics of the benchmark shape all hit the same two addresses and serialise in the L2 at ~20 ns each: 36 us for 42 MB of
// traffic, between the student's forward and backward where nothing else of the chain can run (r05 trace; requesting all maps of a
// pixel up front changed nothing).
template <typename T>
__global__ __launch_bounds__(256) void loss_vec_kernel(const fpd_loss_t a, const int tpb) {
    constexpr int VEC = DT<T>::VEC, PT = 128;
    __shared__ float s_tg[PT * 33];       // [pixel][joint], J <= 32, padded rows
    __shared__ double s_acc[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int J = a.J, HW = a.H * a.W;
    const int tiles_per_img = (HW + PT - 1) / PT, bpi = (tiles_per_img + tpb - 1) / tpb;       // blocks per image
    const int b = blockIdx.x / bpi, tile0 = (blockIdx.x - b * bpi) * tpb;
    const int LDJ = J + 1;
    const double cnt = (double)a.B * J * HW;
    const float gs = a.grad_scale / (float)cnt;
    const T* tch = reinterpret_cast<const T*>(a.teacher);
    const int VPP = J / VEC;                     // vectors per pixel
    float pose = 0.f, kd = 0.f;
    for (int tt = tile0; tt < min(tile0 + tpb, tiles_per_img); ++tt) {
    const int p0 = tt * PT;
    __syncthreads();                             // the previous tile's targets have been read
    if (a.target_nchw) {
        for (int i = tid; i < J * PT; i += 256) {
            const int j = i / PT, p = i - j * PT;
            s_tg[p * LDJ + j] = (p0 + p < HW) ? a.target[((size_t)b * J + j) * HW + p0 + p] : 0.f;
        }
    } else {
        for (int i = tid; i < J * PT; i += 256) {
            const int p = i / J, j = i - p * J;
            s_tg[p * LDJ + j] = (p0 + p < HW) ? a.target[((size_t)b * HW + p0 + p) * J + j] : 0.f;
        }
    }
    __syncthreads();
    for (int v = tid; v < PT * VPP; v += 256) {
        const int p = v / VPP, j0 = (v - p * VPP) * VEC;
        if (p0 + p >= HW) continue;
        const size_t off = ((size_t)b * HW + p0 + p) * J + j0;
        float w2[VEC], w2k[VEC], g[VEC], t[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float wgt = a.weight[b * J + j0 + e];
            const float wk = a.weight_kd ? a.weight_kd[b * J + j0 + e] : wgt;
            w2[e] = wgt * wgt; w2k[e] = wk * wk;
            g[e] = s_tg[p * LDJ + j0 + e];
        }
        DT<T>::unpack(*reinterpret_cast<const uint4*>(tch + off), t);
        for (int s = 0; s < a.S; ++s) {
            float pv[VEC], d[VEC];
            DT<T>::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(a.out[s]) + off), pv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float dg = pv[e] - g[e], dt = pv[e] - t[e];
                pose += w2[e] * dg * dg;
                kd += w2k[e] * dt * dt;
                d[e] = gs * ((1.f - a.alpha) * w2[e] * dg + a.alpha * w2k[e] * dt);
            }
            if (a.dout[s] != nullptr) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.dout[s]) + off) = DT<T>::pack(d);
        }
    }
    }                                            // tiles of this block
    const double dp = wave_sum_d((double)pose), dk = wave_sum_d((double)kd);
    if (lane == 0) { s_acc[0][wave] = dp; s_acc[1][wave] = dk; }
    __syncthreads();
    if (tid == 0) {
        const double sc = 0.5 / cnt, q = 17592186044416.0;     // 2^44 (see loss_kernel)
        const double vp = sc * (s_acc[0][0] + s_acc[0][1] + s_acc[0][2] + s_acc[0][3]);
        const double vk = sc * (s_acc[1][0] + s_acc[1][1] + s_acc[1][2] + s_acc[1][3]);
        atomicAdd(a.losses + 0, rint(vp * q) / q);
        atomicAdd(a.losses + 1, rint(vk * q) / q);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(const fpd_adam_t a) {
    float bc1 = a.bias_corr1, bc2 = a.bias_corr2, lr = a.lr;
    if (a.step_dev != nullptr) {
        const double t = (double)(*a.step_dev + 1);
        bc1 = (float)(1.0 - pow((double)a.beta1, t));
        bc2 = (float)(1.0 - pow((double)a.beta2, t));
    }
    if (a.lr_dev != nullptr) lr = *a.lr_dev;
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    bf16_t* lp = reinterpret_cast<bf16_t*>(a.param_lp);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = a.grad[i] * a.grad_scale;
        const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
        const float p = a.param[i] - step_size * (m / denom);
        a.m[i] = m; a.v[i] = v; a.param[i] = p;
        if (lp) lp[i] = f2bf(p);
    }
}
__global__ void adam_tick_kernel(int64_t* step) { *step += 1; }

// Working copies of the convolution weights, refreshed once per step from the fp32 masters ([K][R][S][C]): w_fwd = the same layout in
// the storage type, w_bwd = the flipped, IO-swapped [C][R][S][K] copy the data gradient multiplies with.  Round 5: 32 x 32 (k, c)
// tiles of one tap go through LDS, so that BOTH sides are coalesced -- the element-wise form (four integer divisions per element and a
// 2-byte store with a stride of R S K elements for every element of w_bwd) took 39 us per step for 3.3 M parameters, in front of
// the student's forward.
template <typename T>
__global__ __launch_bounds__(256) void wprep_kernel(const fpd_wprep_entry_t* table) {
    __shared__ float tile[32][33];
    const fpd_wprep_entry_t e = table[blockIdx.y];
    const int K = e.K, RS = e.R * e.S, C = e.C;
    T* wf = reinterpret_cast<T*>(e.w_fwd);
    T* wb = reinterpret_cast<T*>(e.w_bwd);
    const int kt = (K + 31) >> 5, ct = (C + 31) >> 5, ntiles = RS * kt * ct;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int rs = t / (kt * ct), rem = t - rs * (kt * ct), k0 = (rem / ct) * 32, c0 = (rem - (rem / ct) * ct) * 32;
        __syncthreads();                                              // the previous tile has been written out
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + ty + 8 * j, c = c0 + tx;
            float v = 0.f;
            if (k < K && c < C) {
                v = e.w[((size_t)k * RS + rs) * C + c];
                if (wf) DT<T>::st(wf + ((size_t)k * RS + rs) * C + c, v);
            }
            tile[ty + 8 * j][tx] = v;
        }
        if (wb) {
            __syncthreads();
            const int rsf = RS - 1 - rs;                              // (R-1-r, S-1-s) of a square filter = the reversed flat tap index
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + ty + 8 * j, k = k0 + tx;
                if (k < K && c < C) DT<T>::st(wb + ((size_t)c * RS + rsf) * K + k, tile[tx][ty + 8 * j]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void bnupd_kernel(const fpd_bnupd_entry_t* table) {
    const fpd_bnupd_entry_t e = table[blockIdx.x];
    for (int c = threadIdx.x; c < e.C; c += blockDim.x) {
        const double mean = stats_sum(e.stats, e.C, 0, c) / e.count;
        double var = stats_sum(e.stats, e.C, 1, c) / e.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double unb = e.count > 1.0 ? var * e.count / (e.count - 1.0) : var;
        e.running_mean[c] = (float)((1.0 - e.momentum) * (double)e.running_mean[c] + e.momentum * mean);
        e.running_var[c] = (float)((1.0 - e.momentum) * (double)e.running_var[c] + e.momentum * unb);
    }
    if (threadIdx.x == 0 && e.num_batches_tracked != nullptr) *e.num_batches_tracked += 1;
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* s, TD* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        DT<TD>::st(d + i, DT<TS>::ld(s + i));
}

// out[n][h][w][c] = in[n][c][h][w]   (thread per output element; small tensors only)
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* dst, int N, int C, int H, int W) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const int hw = (int)(pix % ((int64_t)H * W)), n = (int)(pix / ((int64_t)H * W));
        DT<T>::st(dst + i, src[((int64_t)n * C + c) * H * W + hw]);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, float* dst, int N, int C, int H, int W) {
    const int64_t total = (int64_t)N * C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % ((int64_t)H * W));
        const int64_t nc = i / ((int64_t)H * W);
        const int c = (int)(nc % C), n = (int)(nc / C);
        dst[i] = DT<T>::ld(src + ((int64_t)n * H * W + hw) * C + c);
    }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

}  // namespace

int fpd_loss_launch(const fpd_loss_t& a, hipStream_t st) {
    if (a.J > 32 || a.S > FPD_MAX_STACKS || a.S < 1) return fpd_fail(-3, "loss: J=%d (<=32), S=%d (1..%d)", a.J, a.S, FPD_MAX_STACKS);
    const int vec = a.dtype == FPD_BF16 ? 8 : 4;
    bool aligned = a.J % vec == 0 && ((uintptr_t)a.teacher & 15) == 0;
    for (int s = 0; s < a.S; ++s) aligned = aligned && ((uintptr_t)a.out[s] & 15) == 0 && ((uintptr_t)a.dout[s] & 15) == 0;
    if (aligned) {                               // 16-byte vectors of joints (the benchmark's J = 16)
        // <= 256 blocks (= 512 same-address atomics), tiles of one image per block
        const int tpi = cdiv(a.H * a.W, 128);
        int tpb = 1;
        while (a.B * cdiv(tpi, tpb) > 256 && tpb < tpi) ++tpb;
        const int vt = a.B * cdiv(tpi, tpb);
        if (a.dtype == FPD_BF16)
            FPD_LAUNCH((loss_vec_kernel<bf16_t>), dim3(vt), dim3(256), 0, st, a, tpb);
        else
            FPD_LAUNCH((loss_vec_kernel<float>), dim3(vt), dim3(256), 0, st, a, tpb);
        return 0;
    }
    const int tiles = a.B * cdiv(a.H * a.W, 64);
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((loss_kernel<bf16_t>), dim3(tiles), dim3(256), 0, st, a);
    else
        FPD_LAUNCH((loss_kernel<float>), dim3(tiles), dim3(256), 0, st, a);
    return 0;
}

int fpd_adam_launch(const fpd_adam_t& a, hipStream_t st) {
    FPD_LAUNCH(adam_kernel, dim3(grid_for(a.n)), dim3(256), 0, st, a);
    if (a.step_dev != nullptr) FPD_LAUNCH(adam_tick_kernel, dim3(1), dim3(1), 0, st, a.step_dev);
    return 0;
}

int fpd_weight_prep_launch(const fpd_wprep_entry_t* table, int n, int64_t max_elems, int dtype, hipStream_t st) {
    if (n <= 0) return 0;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((max_elems + 1023) / 1024, 64)), (unsigned)n);     // 32 x 32 tiles, grid-stride
    if (dtype == FPD_BF16)
        FPD_LAUNCH((wprep_kernel<bf16_t>), grid, dim3(256), 0, st, table);
    else
        FPD_LAUNCH((wprep_kernel<float>), grid, dim3(256), 0, st, table);
    return 0;
}

int fpd_bn_update_running_launch(const fpd_bnupd_entry_t* table, int n, hipStream_t st) {
    if (n <= 0) return 0;
    FPD_LAUNCH(bnupd_kernel, dim3(n), dim3(256), 0, st, table);
    return 0;
}

int fpd_cast_launch(const void* src, void* dst, int64_t n, int sd, int dd, hipStream_t st) {
    const int g = grid_for(n);
    if (sd == FPD_F32 && dd == FPD_BF16)
        FPD_LAUNCH((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n);
    else if (sd == FPD_BF16 && dd == FPD_F32)
        FPD_LAUNCH((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n);
    else if (sd == FPD_F32 && dd == FPD_F32)
        FPD_LAUNCH((cast_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, n);
    else
        FPD_LAUNCH((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n);
    return 0;
}

int fpd_nchw_to_nhwc_launch(const float* src, void* dst, int N, int C, int H, int W, int dtype, hipStream_t st) {
    const int g = grid_for((int64_t)N * C * H * W);
    if (dtype == FPD_BF16)
        FPD_LAUNCH((nchw_to_nhwc_kernel<bf16_t>), dim3(g), dim3(256), 0, st, src, (bf16_t*)dst, N, C, H, W);
    else
        FPD_LAUNCH((nchw_to_nhwc_kernel<float>), dim3(g), dim3(256), 0, st, src, (float*)dst, N, C, H, W);
    return 0;
}

int fpd_nhwc_to_nchw_launch(const void* src, float* dst, int N, int C, int H, int W, int dtype, hipStream_t st) {
    const int g = grid_for((int64_t)N * C * H * W);
    if (dtype == FPD_BF16)
        FPD_LAUNCH((nhwc_to_nchw_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, dst, N, C, H, W);
    else
        FPD_LAUNCH((nhwc_to_nchw_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, dst, N, C, H, W);
    return 0;
}
