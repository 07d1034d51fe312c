// HBM-bound NHWC kernels of the hourglass graph and its backward: BN+ReLU, 2x2 max-pool, nearest x2
// up-add, their gradients, BatchNorm-backward apply.  16-byte vectors along C, one vector per thread per
// step, grid-stride over pixels; producers of a tensor that feeds a train-mode BN accumulate its
// per-channel {sum, sum^2} (fp64 in registers -> exact integer limbs in LDS -> one limb pair per sum, channel and block).
// Replaces F.max_pool2d / nn.Upsample / nn.BatchNorm2d / nn.ReLU and their autograd in
// /root/reference/lib/models/hourglass.py:80-92,172-177.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void ldv(const T* p, float* f) {
    DT<T>::unpack(*reinterpret_cast<const uint4*>(p), f);
}
template <typename T>
__device__ __forceinline__ void stv(T* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = DT<T>::pack(f);
}

// One thread owns channel-vector `cv` (VEC channels) and walks pixels pl, pl+stride, ...
// (bx, gx): this block's index and the number of blocks working on `a` (grid-stride over pixels)
// NA ("no alias"): no input tensor of the op is its output tensor (decided on the host by comparing the pointers): the
// inputs and the output are then __restrict__, which is what lets the loads of the next pixel pass the stores of this one
// (two pixels in flight per thread).  An op that works IN PLACE (`add` / `x` may be `y`: include/fpd_amd.h) takes the
// instantiation without the qualifiers -- reading an object through one restrict pointer and writing it through another is
// undefined behaviour, however the code object of the day happens to schedule it.
template <typename T, bool NA> struct EwPtr { typedef const T* in; typedef T* out; };
template <typename T> struct EwPtr<T, true> { typedef const T* __restrict__ in; typedef T* __restrict__ out; };

template <typename T, int OP, bool NA>
__device__ __forceinline__ void ew_body(const fpd_ew_t& a, const int bx, const int gx) {
    constexpr int VEC = DT<T>::VEC;
    __shared__ float s_t0[FPD_MAXC], s_t1[FPD_MAXC], s_t2[FPD_MAXC], s_t3[FPD_MAXC];
    __shared__ float s_is[FPD_MAXC];
    __shared__ long long s_sum[4 * FPD_MAXC];   // [2 sums][2 limbs][C]: exact block accumulator of the statistics (common.h)
    const int tid = threadIdx.x;
    const int C = a.C, H = a.H, W = a.W, N = a.N;
    const int VP = C / VEC;              // vectors per pixel
    const int PB = 256 / VP;             // pixels per block step (VP <= 128 guaranteed by host)
    const int cv = (tid % VP) * VEC;
    const int pl = tid / VP;
    const bool active = pl < PB;
    constexpr bool HALF_OUT = (OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_MAXPOOL_BWD || OP == FPD_EW_SUMPOOL);
    const int OH = HALF_OUT ? H / 2 : H, OW = HALF_OUT ? W / 2 : W;   // iteration space
    const int npix = N * OH * OW;
    constexpr bool STATS = (OP == FPD_EW_BNRELU_FWD || OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_UPADD_FWD);
    constexpr bool BSTATS = (OP == FPD_EW_BNRELU_BWD_R);

    // per-channel tables
    if (OP == FPD_EW_BNRELU_FWD || OP == FPD_EW_BNRELU_BWD_R || OP == FPD_EW_BN_BWD_APPLY) {
        const double cnt = (double)N * H * W;
        for (int c = tid; c < C; c += 256) {
            float sc, sh, mu, is;
            bn_coef(a.bn, c, C, cnt, sc, sh, mu, is);
            if (OP == FPD_EW_BN_BWD_APPLY) {
                s_t0[c] = a.bn.gamma[c] * is;                 // gamma * invstd
                const double b1 = stats_sum(a.bstats, C, 0, c), b2 = stats_sum(a.bstats, C, 1, c);
                s_t1[c] = (float)(b1 / cnt);                  // mean(dz)
                s_t2[c] = (float)(b2 / cnt);                  // mean(dz * xhat)
                s_t3[c] = mu;
                s_is[c] = is;
                if (bx == 0) {   // gradients of the BN affine parameters fall out of the two sums
                    if (a.dgamma) a.dgamma[c] = (float)b2;
                    if (a.dbeta) a.dbeta[c] = (float)b1;
                }
            } else {
                s_t0[c] = sc; s_t1[c] = sh; s_t2[c] = mu; s_t3[c] = is;
            }
        }
    }
    if (STATS || BSTATS)
        for (int c = tid; c < 4 * C; c += 256) s_sum[c] = 0;
    __syncthreads();

    double acc1[VEC], acc2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc1[j] = 0.0; acc2[j] = 0.0; }

    const typename EwPtr<T, NA>::in x = reinterpret_cast<const T*>(a.x);
    const typename EwPtr<T, NA>::in x2 = reinterpret_cast<const T*>(a.x2);
    const typename EwPtr<T, NA>::in dy = reinterpret_cast<const T*>(a.dy);
    const typename EwPtr<T, NA>::in add = reinterpret_cast<const T*>(a.add);
    const typename EwPtr<T, NA>::out y = reinterpret_cast<T*>(a.y);
    const bool do_stats = STATS ? (a.out_stats != nullptr) : BSTATS;

    if (active) {
        for (int pix = bx * PB + pl; pix < npix; pix += gx * PB) {
            float o[VEC];
            if (OP == FPD_EW_BNRELU_FWD) {
                float v[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = bn_act(v[j], s_t0[cv + j], s_t1[cv + j], a.bn.relu);
                stv<T>(y + (size_t)pix * C + cv, o);
            } else if (OP == FPD_EW_BNRELU_BWD_R) {
                float v[VEC], g[VEC], mk[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
                ldv<T>(dy + (size_t)pix * C + cv, g);
                // mk = what the ReLU mask is taken from (one uniform branch per pixel; per element it compiled to a branch each)
                if (x2 != nullptr) {
                    ldv<T>(x2 + (size_t)pix * C + cv, mk);                    // forward output of a relu(sum) op
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) mk[j] = a.bn.relu ? fmaf(v[j], s_t0[cv + j], s_t1[cv + j]) : 1.f;
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    o[j] = mk[j] > 0.f ? g[j] : 0.f;
                    const double r = (double)DT<T>::rnd(o[j]);
                    acc1[j] += r;
                    acc2[j] += r * (double)((v[j] - s_t2[cv + j]) * s_t3[cv + j]);
                }
                if (y != nullptr) stv<T>(y + (size_t)pix * C + cv, o);     // y NULL: the two sums only
            } else if (OP == FPD_EW_RELU_MASK) {
                float v[VEC], g[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
                ldv<T>(dy + (size_t)pix * C + cv, g);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = v[j] > 0.f ? g[j] : 0.f;
                stv<T>(y + (size_t)pix * C + cv, o);
            } else if (OP == FPD_EW_DILATE2) {
                const int n = pix / (H * W), rem = pix - n * (H * W);
                const int oy = rem / W, ox = rem - oy * W;
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = 0.f;
                if (((oy | ox) & 1) == 0) ldv<T>(x + ((size_t)(n * (H / 2) + oy / 2) * (W / 2) + ox / 2) * C + cv, o);
                stv<T>(y + (size_t)pix * C + cv, o);
            } else if (OP == FPD_EW_BN_BWD_APPLY) {
                float v[VEC], g[VEC], ad[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
                ldv<T>(dy + (size_t)pix * C + cv, g);
                if (add) ldv<T>(add + (size_t)pix * C + cv, ad);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int c = cv + j;
                    const float xhat = (v[j] - s_t3[c]) * s_is[c];
                    o[j] = s_t0[c] * (g[j] - s_t1[c] - xhat * s_t2[c]) + (add ? ad[j] : 0.f);
                }
                stv<T>(y + (size_t)pix * C + cv, o);
            } else if (OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_MAXPOOL_BWD || OP == FPD_EW_SUMPOOL) {
                const int n = pix / (OH * OW), rem = pix - n * (OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                const size_t i00 = ((size_t)(n * H + 2 * oy) * W + 2 * ox) * C + cv;
                const size_t i01 = i00 + C, i10 = i00 + (size_t)W * C, i11 = i10 + C;
                float v0[VEC], v1[VEC], v2[VEC], v3[VEC];
                ldv<T>(x + i00, v0); ldv<T>(x + i01, v1); ldv<T>(x + i10, v2); ldv<T>(x + i11, v3);
                if (OP == FPD_EW_MAXPOOL_FWD) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] = fmaxf(fmaxf(v0[j], v1[j]), fmaxf(v2[j], v3[j]));
                    stv<T>(y + (size_t)pix * C + cv, o);
                } else if (OP == FPD_EW_SUMPOOL) {
                    float ad[VEC];
                    if (add) ldv<T>(add + (size_t)pix * C + cv, ad);
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] = (v0[j] + v1[j]) + (v2[j] + v3[j]) + (add ? ad[j] : 0.f);
                    stv<T>(y + (size_t)pix * C + cv, o);
                } else {  // MAXPOOL_BWD: route dy to the first maximum in scan order (torch max_pool2d semantics)
                    float g[VEC], a0[VEC], a1[VEC], a2[VEC], a3[VEC];
                    ldv<T>(dy + (size_t)pix * C + cv, g);
                    if (add) { ldv<T>(add + i00, a0); ldv<T>(add + i01, a1); ldv<T>(add + i10, a2); ldv<T>(add + i11, a3); }
                    float o0[VEC], o1[VEC], o2[VEC], o3[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        int arg = 0; float best = v0[j];
                        if (v1[j] > best) { best = v1[j]; arg = 1; }
                        if (v2[j] > best) { best = v2[j]; arg = 2; }
                        if (v3[j] > best) { best = v3[j]; arg = 3; }
                        o0[j] = (arg == 0 ? g[j] : 0.f) + (add ? a0[j] : 0.f);
                        o1[j] = (arg == 1 ? g[j] : 0.f) + (add ? a1[j] : 0.f);
                        o2[j] = (arg == 2 ? g[j] : 0.f) + (add ? a2[j] : 0.f);
                        o3[j] = (arg == 3 ? g[j] : 0.f) + (add ? a3[j] : 0.f);
                    }
                    stv<T>(y + i00, o0); stv<T>(y + i01, o1); stv<T>(y + i10, o2); stv<T>(y + i11, o3);
                }
            } else if (OP == FPD_EW_UPADD_FWD) {
                const int n = pix / (H * W), rem = pix - n * (H * W);
                const int oy = rem / W, ox = rem - oy * W;
                float v[VEC], u[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
                ldv<T>(x2 + ((size_t)(n * (H / 2) + oy / 2) * (W / 2) + ox / 2) * C + cv, u);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = v[j] + u[j];
                stv<T>(y + (size_t)pix * C + cv, o);
            } else {  // ADD
                float v[VEC], u[VEC];
                ldv<T>(x + (size_t)pix * C + cv, v);
                ldv<T>(x2 + (size_t)pix * C + cv, u);
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = v[j] + u[j];
                stv<T>(y + (size_t)pix * C + cv, o);
            }
            if (STATS && do_stats) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const double r = (double)DT<T>::rnd(o[j]); acc1[j] += r; acc2[j] += r * r; }
            }
        }
    }
    if (do_stats) {
        if (active) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { stat_lds_add(s_sum, C, 0, cv + j, acc1[j]); stat_lds_add(s_sum, C, 1, cv + j, acc2[j]); }
        }
        __syncthreads();
        fpd_stat_t* dst = STATS ? a.out_stats : a.bstats;
        for (int c = tid; c < C; c += 256) stat_flush_lds(dst, s_sum, C, c);
    }
}

template <typename T, int OP, bool NA>
__global__ __launch_bounds__(256) void ew_kernel(const fpd_ew_t a) { ew_body<T, OP, NA>(a, blockIdx.x, gridDim.x); }

// true if no input of the op is its output (exact pointer equality: the graph's in-place ops alias whole tensors)
static bool ew_no_alias(const fpd_ew_t& a) {
#ifdef FPD_EW_ASSUME_NOALIAS      // probe build only (A/B of what the qualifiers are worth): the round-5 behaviour, undefined for in-place ops
    return true;
#endif
    return a.y == nullptr || (a.y != a.x && a.y != a.x2 && a.y != a.dy && a.y != a.add);
}

// two independent ops of the same kind in one launch (the BN-backward applies of the paired bottleneck chains)
template <typename T, int OP, bool NA>
__global__ __launch_bounds__(256) void ew_pair_kernel(const fpd_ew_t a, const fpd_ew_t b, const int ga) {
    const int gb = (int)gridDim.x - ga;                       // b (half resolution) first: no tail of small blocks
    if ((int)blockIdx.x < gb) ew_body<T, OP, NA>(b, blockIdx.x, gb);
    else ew_body<T, OP, NA>(a, (int)blockIdx.x - gb, ga);
}

// grid cap of the ops that end in statistics atomics (FPD_EW_STATS_BLOCKS)
static int ew_stats_blocks() {
    static int v = 0;
    if (!v) {
        const char* e = getenv("FPD_EW_STATS_BLOCKS");
        v = e ? atoi(e) : 512;
        if (v < 1) v = 1;
    }
    return v;
}

template <typename T, int OP>
int launch_ew(const fpd_ew_t& a, hipStream_t st) {
    constexpr int VEC = DT<T>::VEC;
    const int VP = a.C / VEC, PB = 256 / VP;
    const bool half = (OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_MAXPOOL_BWD || OP == FPD_EW_SUMPOOL);
    const int npix = a.N * (half ? a.H / 2 : a.H) * (half ? a.W / 2 : a.W);
    // grid-stride kernels: ops that end with per-channel statistics atomics get few, long-lived blocks -- every block
    // adds into the SAME [2][C] buffer and same-address device atomics serialise (~12 ns each)
    const bool stats = (OP == FPD_EW_BNRELU_FWD || OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_UPADD_FWD) ? a.out_stats != nullptr
                                                                                                   : OP == FPD_EW_BNRELU_BWD_R;
    const int grid = std::max(1, std::min(cdiv(npix, PB), stats ? ew_stats_blocks() : 2048));
    if (ew_no_alias(a)) FPD_LAUNCH((ew_kernel<T, OP, true>), dim3(grid), dim3(256), 0, st, a);
    else FPD_LAUNCH((ew_kernel<T, OP, false>), dim3(grid), dim3(256), 0, st, a);
    return 0;
}

template <typename T, int OP>
int ew_grid(const fpd_ew_t& a) {
    constexpr int VEC = DT<T>::VEC;
    const int VP = a.C / VEC, PB = 256 / VP;
    const bool half = (OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_MAXPOOL_BWD || OP == FPD_EW_SUMPOOL);
    const int npix = a.N * (half ? a.H / 2 : a.H) * (half ? a.W / 2 : a.W);
    const bool stats = (OP == FPD_EW_BNRELU_FWD || OP == FPD_EW_MAXPOOL_FWD || OP == FPD_EW_UPADD_FWD) ? a.out_stats != nullptr
                                                                                                   : OP == FPD_EW_BNRELU_BWD_R;
    return std::max(1, std::min(cdiv(npix, PB), stats ? ew_stats_blocks() : 2048));
}

template <typename T>
int launch_ew_pair(const fpd_ew_t& a, const fpd_ew_t& b, hipStream_t st) {
    if (a.op != FPD_EW_BN_BWD_APPLY) return 1;                 // the only pairing the graph builder emits
    const int ga = ew_grid<T, FPD_EW_BN_BWD_APPLY>(a), gb = ew_grid<T, FPD_EW_BN_BWD_APPLY>(b);
    if (ew_no_alias(a) && ew_no_alias(b)) FPD_LAUNCH((ew_pair_kernel<T, FPD_EW_BN_BWD_APPLY, true>), dim3(ga + gb), dim3(256), 0, st, a, b, ga);
    else FPD_LAUNCH((ew_pair_kernel<T, FPD_EW_BN_BWD_APPLY, false>), dim3(ga + gb), dim3(256), 0, st, a, b, ga);
    return 0;
}

// y = relu?(sum_j bn_j(up_j(x_j))), see fpd_affsum_t.  Same thread layout as ew_body: a thread owns one channel vector and
// walks output pixels grid-stride; per-term scale/shift tables in LDS.
template <typename T>
__global__ __launch_bounds__(256) void affsum_kernel(const fpd_affsum_t a) {
    constexpr int VEC = DT<T>::VEC;
    __shared__ float s_sc[FPD_AFFSUM_MAX][FPD_MAXC], s_sh[FPD_AFFSUM_MAX][FPD_MAXC];
    const int tid = threadIdx.x;
    const int C = a.C, H = a.H, W = a.W;
    for (int j = 0; j < a.nterms; ++j) {
        if (a.t[j].bn.mode == FPD_BN_NONE) continue;
        const int up = a.t[j].up;
        const double cnt = (double)a.N * (H / up) * (W / up);
        for (int c = tid; c < C; c += 256) {
            float sc, sh, mu, is;
            bn_coef(a.t[j].bn, c, C, cnt, sc, sh, mu, is);
            s_sc[j][c] = sc;
            s_sh[j][c] = sh;
        }
    }
    __syncthreads();
    const int VP = C / VEC, PB = 256 / VP;
    const int cv = (tid % VP) * VEC, pl = tid / VP;
    if (pl >= PB) return;
    const int npix = a.N * H * W;
    T* y = reinterpret_cast<T*>(a.y);
    for (int pix = blockIdx.x * PB + pl; pix < npix; pix += gridDim.x * PB) {
        const int n = pix / (H * W), rem = pix - n * (H * W);
        const int oy = rem / W, ox = rem - oy * W;
        float acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        for (int j = 0; j < a.nterms; ++j) {
            const int up = a.t[j].up;
            const T* x = reinterpret_cast<const T*>(a.t[j].x);
            float v[VEC];
            ldv<T>(x + ((size_t)(n * (H / up) + oy / up) * (W / up) + ox / up) * C + cv, v);
            if (a.t[j].bn.mode != FPD_BN_NONE) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] = bn_act(v[e], s_sc[j][cv + e], s_sh[j][cv + e], a.t[j].bn.relu);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += v[e];
        }
        if (a.relu) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        stv<T>(y + (size_t)pix * C + cv, acc);
    }
}

template <typename T>
int dispatch_ew(const fpd_ew_t& a, hipStream_t st) {
    switch (a.op) {
        case FPD_EW_BNRELU_FWD: return launch_ew<T, FPD_EW_BNRELU_FWD>(a, st);
        case FPD_EW_BNRELU_BWD_R: return launch_ew<T, FPD_EW_BNRELU_BWD_R>(a, st);
        case FPD_EW_BN_BWD_APPLY: return launch_ew<T, FPD_EW_BN_BWD_APPLY>(a, st);
        case FPD_EW_MAXPOOL_FWD: return launch_ew<T, FPD_EW_MAXPOOL_FWD>(a, st);
        case FPD_EW_MAXPOOL_BWD: return launch_ew<T, FPD_EW_MAXPOOL_BWD>(a, st);
        case FPD_EW_UPADD_FWD: return launch_ew<T, FPD_EW_UPADD_FWD>(a, st);
        case FPD_EW_SUMPOOL: return launch_ew<T, FPD_EW_SUMPOOL>(a, st);
        case FPD_EW_ADD: return launch_ew<T, FPD_EW_ADD>(a, st);
        case FPD_EW_RELU_MASK: return launch_ew<T, FPD_EW_RELU_MASK>(a, st);
        case FPD_EW_DILATE2: return launch_ew<T, FPD_EW_DILATE2>(a, st);
    }
    return fpd_fail(-2, "elementwise: unknown op %d", a.op);
}

}  // namespace

static int ew_check(const fpd_ew_t& a) {
    const int vec = (a.dtype == FPD_BF16) ? 8 : 4;
    if (a.C % vec != 0 || a.C > FPD_MAXC || a.C / vec > 128)
        return fpd_fail(-3, "elementwise: C=%d must be a multiple of %d and <= %d", a.C, vec, FPD_MAXC);
    return 0;
}

// 0 = launched as one kernel, 1 = not pairable (caller launches them one by one), <0 error
int fpd_elementwise_pair_launch(const fpd_ew_t& a, const fpd_ew_t& b, hipStream_t st) {
    if (a.op != b.op || a.dtype != b.dtype) return 1;
    int rc = ew_check(a);
    if (rc) return rc;
    rc = ew_check(b);
    if (rc) return rc;
    return a.dtype == FPD_BF16 ? launch_ew_pair<bf16_t>(a, b, st) : launch_ew_pair<float>(a, b, st);
}

int fpd_elementwise_launch(const fpd_ew_t& a, hipStream_t st) {
    const int vec = (a.dtype == FPD_BF16) ? 8 : 4;
    if (a.C % vec != 0 || a.C > FPD_MAXC || a.C / vec > 128)
        return fpd_fail(-3, "elementwise: C=%d must be a multiple of %d and <= %d", a.C, vec, FPD_MAXC);
    const bool pooled = (a.op == FPD_EW_MAXPOOL_FWD || a.op == FPD_EW_MAXPOOL_BWD || a.op == FPD_EW_SUMPOOL ||
                         a.op == FPD_EW_UPADD_FWD || a.op == FPD_EW_DILATE2);
    if (pooled && ((a.H & 1) || (a.W & 1))) return fpd_fail(-3, "elementwise: 2x2 ops need even H,W (got %dx%d)", a.H, a.W);
    return a.dtype == FPD_BF16 ? dispatch_ew<bf16_t>(a, st) : dispatch_ew<float>(a, st);
}

int fpd_affsum_launch(const fpd_affsum_t& a, hipStream_t st) {
    const int vec = (a.dtype == FPD_BF16) ? 8 : 4;
    if (a.C % vec != 0 || a.C > FPD_MAXC || a.C / vec > 128)
        return fpd_fail(-3, "affsum: C=%d must be a multiple of %d and <= %d", a.C, vec, FPD_MAXC);
    const int PB = 256 / (a.C / vec);
    const int grid = std::max(1, std::min(cdiv(a.N * a.H * a.W, PB), 2048));
    if (a.dtype == FPD_BF16) FPD_LAUNCH((affsum_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a);
    else FPD_LAUNCH((affsum_kernel<float>), dim3(grid), dim3(256), 0, st, a);
    return 0;
}
