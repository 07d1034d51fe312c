// Implicit-GEMM convolution on the gfx950 matrix cores (NHWC activations, K,R,S,C weights).
//
//   conv_mfma_kernel  : y[m][k] = sum_{r,s,c} a(x[pix(m,r,s)][c]) * w[k][r][s][c]  (+bias, +residual)
//                       a() = fused BatchNorm+ReLU of the producer tensor (train or eval statistics);
//                       epilogue optionally accumulates the batch statistics of y (next train-mode BN)
//                       or, for data gradients, applies the ReLU mask of the tensor the gradient flows
//                       into and accumulates the two BatchNorm-backward sums.
//   wgrad_mfma_kernel : dw[k][r][s][c] += sum_m dy[m][k] * a(x[pix(m,r,s)][c]) , split over pixel chunks.
//
// Replaces the cuDNN/ATen arithmetic behind nn.Conv2d / nn.BatchNorm2d / nn.ReLU in
// /root/reference/lib/models/hourglass.py:32-52,170-192 and their autograd.
//
// Tiling: 256 threads = 4 wave64; block tile = 128 output pixels x (32*TN) output channels; each wave
// owns 32 pixels x 32*TN channels as TN accumulators of v_mfma_f32_32x32x16_bf16 (bf16 storage) or
// v_mfma_f32_32x32x2_f32 (fp32 storage: exact fp32, used for the 1e-4 parity build).  The K loop walks
// (filter tap, BK-channel chunk); global loads of tile t+1 are issued before the MFMAs of tile t and the
// BN+ReLU transform runs on the way from registers to LDS.
#include <algorithm>

#include "common.h"
#include "conv_epilogue.h"

namespace {

template <typename T>
struct MmaTile;

// bf16: fragment = 8 consecutive k per lane; lanes 0-31 take k 0..7, lanes 32-63 take k 8..15 of each 16-slab.
template <>
struct MmaTile<bf16_t> {
    template <int TN, int BK, int LD>
    static __device__ __forceinline__ void run(const bf16_t* sA, const bf16_t* sB, int arow, int lane, f32x16* acc) {
        const int koff = 8 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sA + arow * LD + kk * 16 + koff);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(sB + (tn * 32 + (lane & 31)) * LD + kk * 16 + koff);
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tn], 0, 0, 0);
            }
        }
    }
};

// fp32: v_mfma_f32_32x32x2_f32 takes ONE k per lane (lanes 0-31: slot 0, lanes 32-63: slot 1).  The k order
// inside a tile is free as long as A and B agree, so slot 0 walks columns [0,BK/2) and slot 1 walks
// [BK/2,BK): each lane then reads 4 consecutive floats (one ds_read_b128) per 4 MFMAs.
template <>
struct MmaTile<float> {
    template <int TN, int BK, int LD>
    static __device__ __forceinline__ void run(const float* sA, const float* sB, int arow, int lane, f32x16* acc) {
        const int koff = (lane >> 5) * (BK / 2);
#pragma unroll
        for (int t4 = 0; t4 < BK / 8; ++t4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sA + arow * LD + koff + t4 * 4);
            f32x4 b[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                b[tn] = *reinterpret_cast<const f32x4*>(sB + (tn * 32 + (lane & 31)) * LD + koff + t4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[tn][j], acc[tn], 0, 0, 0);
            }
        }
    }
};

template <typename T, int TN, int BK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const fpd_conv_t a) {
    constexpr int VEC = DT<T>::VEC;
    constexpr int BM = 128, BNT = 32 * TN;
    constexpr int LD = BK + 16 / (int)sizeof(T);  // +16 bytes: conflict-free ds_read_b128 over 16 rows
    constexpr int VPR = BK / VEC;                  // 16-byte vectors per tile row
    constexpr int NVA = BM * VPR / 256;
    constexpr int NVB_TOT = BNT * VPR;
    constexpr int NVB = (NVB_TOT + 255) / 256;
    static_assert(NVA >= 1, "tile too small");

    __shared__ __attribute__((aligned(16))) T sA[BM * LD];
    __shared__ __attribute__((aligned(16))) T sB[BNT * LD];
    __shared__ float s_scale[FPD_MAXC], s_shift[FPD_MAXC];
    __shared__ float s_epi[4][BNT];
    __shared__ double s_red[4][BNT][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, S = a.S, P = a.P, Q = a.Q;
    const int M = a.N * P * Q;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BNT;
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ w = reinterpret_cast<const T*>(a.w);

    bn_fill(a.bn, C, (double)a.N * H * W, s_scale, s_shift);
    conv_epi_tables<BNT>(a, n0, M, &s_epi[0][0]);

    // ---- per-thread staging coordinates (fixed over the K loop) ----
    const int cvA = (tid % VPR) * VEC;
    int a_base[NVA], a_ih0[NVA], a_iw0[NVA], a_row[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int row = tid / VPR + i * (256 / VPR);
        a_row[i] = row;
        const int m = m0 + row;
        if (m < M) {
            const int n = m / (P * Q), rem = m - n * (P * Q);
            const int p = rem / Q, q = rem - p * Q;
            a_base[i] = n * H * W;
            a_ih0[i] = p * a.stride - a.pad;
            a_iw0[i] = q * a.stride - a.pad;
        } else {
            a_base[i] = 0; a_ih0[i] = -(1 << 28); a_iw0[i] = -(1 << 28);
        }
    }
    int b_row[NVB], b_col[NVB];
    bool b_ok[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int v = tid + i * 256;
        b_row[i] = v / VPR;
        b_col[i] = (v % VPR) * VEC;
        b_ok[i] = (v < NVB_TOT) && (n0 + b_row[i] < K);
    }

    uint4 ra[NVA], rb[NVB];
    unsigned amask = 0;
    auto load_tile = [&](int r, int s, int c0) {
        amask = 0;
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int ih = a_ih0[i] + r, iw = a_iw0[i] + s;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (ok) {
                ra[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(a_base[i] + ih * W + iw) * C + c0 + cvA));
                amask |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            rb[i] = make_uint4(0, 0, 0, 0);
            if (b_ok[i])
                rb[i] = *reinterpret_cast<const uint4*>(
                    w + ((size_t)((n0 + b_row[i]) * R + r) * S + s) * C + c0 + b_col[i]);
        }
    };
    auto store_tile = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            uint4 v = ra[i];
            if (a.bn.mode != FPD_BN_NONE) {
                float f[VEC];
                DT<T>::unpack(v, f);
                const bool ok = (amask >> i) & 1u;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int c = c0 + cvA + j;
                    f[j] = ok ? bn_act(f[j], s_scale[c], s_shift[c], a.bn.relu) : 0.f;
                }
                v = DT<T>::pack(f);
            }
            *reinterpret_cast<uint4*>(sA + a_row[i] * LD + cvA) = v;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i)
            if (tid + i * 256 < NVB_TOT) *reinterpret_cast<uint4*>(sB + b_row[i] * LD + b_col[i]) = rb[i];
    };

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    const int nk = R * S * (C / BK);
    int r = 0, s = 0, c0 = 0;
    load_tile(0, 0, 0);
    const int arow = wave * 32 + (lane & 31);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // LDS free (previous MFMA phase done); also publishes the BN tables on kt == 0
        store_tile(c0);
        __syncthreads();
        c0 += BK;
        if (c0 >= C) { c0 = 0; if (++s >= S) { s = 0; ++r; } }
        if (kt + 1 < nk) load_tile(r, s, c0);
        MmaTile<T>::template run<TN, BK, LD>(sA, sB, arow, lane, acc);
    }

    // ---- epilogue: bias, residual, (ReLU-mask + BN-backward sums | batch statistics), store ----
    __syncthreads();   // all MFMA reads of sA/sB are done: s_red may reuse nothing here, it has its own storage
    conv_epilogue<T, TN>(a, acc, m0 + wave * 32, n0, M, &s_epi[0][0], &s_red[0][0][0]);
}

// -------------------------------------------------------------------------------------------
// weight gradient
// -------------------------------------------------------------------------------------------
template <typename T>
struct WgTile;

// fp32: LDS tiles [pixel][channel]; the 32x32x2 MFMA takes one k (pixel) per lane so no transpose is needed.
template <>
struct WgTile<float> {
    static constexpr int BKP = 16;            // pixels per step
    static constexpr int LDS_ELEMS = BKP * 128;
    template <bool PRO>
    static __device__ __forceinline__ void stage(float* s, const float* g, int tid, const int* pix_off, const bool* pix_ok,
                                                 int ch0, int chn, const float* s_scale, const float* s_shift, int relu,
                                                 int cbase) {
        // 16 pixels x 32 float4 per row = 512 vectors, 2 per thread
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = tid + i * 256;
            const int row = v >> 5, c4 = (v & 31) * 4;
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            if (pix_ok[row] && ch0 + c4 < chn) {
                const uint4 raw = *reinterpret_cast<const uint4*>(g + (size_t)pix_off[row] + ch0 + c4);
                DT<float>::unpack(raw, f);
                if (PRO) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = bn_act(f[j], s_scale[cbase + c4 + j], s_shift[cbase + c4 + j], relu);
                }
            }
            *reinterpret_cast<uint4*>(s + row * 128 + c4) = DT<float>::pack(f);
        }
    }
    static __device__ __forceinline__ void mma(const float* sdy, const float* sa, int ti, int tj, int lane, f32x16& acc) {
#pragma unroll
        for (int t = 0; t < BKP / 2; ++t) {
            const int kk = 2 * t + (lane >> 5);
            const float av = sdy[kk * 128 + ti * 32 + (lane & 31)];
            const float bv = sa[kk * 128 + tj * 32 + (lane & 31)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    static __device__ __forceinline__ float colsum(const float* sdy, int t) {
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < BKP; ++kk) v += sdy[kk * 128 + t];
        return v;
    }
};

// bf16: the 32x32x16 MFMA wants 8 consecutive k (pixels) per lane, so tiles are stored transposed
// [channel][pixel]; a thread packs the same channel of two neighbouring pixels into one dword.
template <>
struct WgTile<bf16_t> {
    static constexpr int BKP = 32;
    static constexpr int LDT = 40;            // 32 pixels + 16 bytes pad
    static constexpr int LDS_ELEMS = 128 * LDT;
    template <bool PRO>
    static __device__ __forceinline__ void stage(bf16_t* s, const bf16_t* g, int tid, const int* pix_off, const bool* pix_ok,
                                                 int ch0, int chn, const float* s_scale, const float* s_shift, int relu,
                                                 int cbase) {
        const int pp = tid & 15, cv = (tid >> 4) * 8;  // pixel pair, channel vector
        float f0[8], f1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { f0[j] = 0.f; f1[j] = 0.f; }
        const bool chok = ch0 + cv < chn;
        if (chok && pix_ok[2 * pp]) {
            const uint4 raw = *reinterpret_cast<const uint4*>(g + (size_t)pix_off[2 * pp] + ch0 + cv);
            DT<bf16_t>::unpack(raw, f0);
            if (PRO) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f0[j] = bn_act(f0[j], s_scale[cbase + cv + j], s_shift[cbase + cv + j], relu);
            }
        }
        if (chok && pix_ok[2 * pp + 1]) {
            const uint4 raw = *reinterpret_cast<const uint4*>(g + (size_t)pix_off[2 * pp + 1] + ch0 + cv);
            DT<bf16_t>::unpack(raw, f1);
            if (PRO) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f1[j] = bn_act(f1[j], s_scale[cbase + cv + j], s_shift[cbase + cv + j], relu);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t word = f2bf_pk(f0[j], f1[j]);
            *reinterpret_cast<uint32_t*>(s + (cv + j) * LDT + 2 * pp) = word;
        }
    }
    static __device__ __forceinline__ void mma(const bf16_t* sdy, const bf16_t* sa, int ti, int tj, int lane, f32x16& acc) {
        const int koff = 8 * (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < BKP / 16; ++kk) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(sdy + (ti * 32 + (lane & 31)) * LDT + kk * 16 + koff);
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(sa + (tj * 32 + (lane & 31)) * LDT + kk * 16 + koff);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
    }
    static __device__ __forceinline__ float colsum(const bf16_t* sdy, int t) {
        float v = 0.f;
#pragma unroll
        for (int kk = 0; kk < BKP; ++kk) v += bf2f(sdy[t * LDT + kk]);
        return v;
    }
};

// grid: x = pixel chunk, y = filter tap (r*S+s), z = ktile*ctiles + ctile (128x128 tiles of the K x C plane)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const fpd_wgrad_t a, const int pch, const int ctiles) {
    using WT = WgTile<T>;
    constexpr int BKP = WT::BKP;
    __shared__ __attribute__((aligned(16))) T s_dy[WT::LDS_ELEMS];
    __shared__ __attribute__((aligned(16))) T s_a[WT::LDS_ELEMS];
    __shared__ float s_scale[FPD_MAXC], s_shift[FPD_MAXC];
    __shared__ int s_poff_dy[BKP], s_poff_x[BKP];
    __shared__ bool s_pok_dy[BKP], s_pok_x[BKP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, C = a.C, K = a.K, S = a.S, P = a.P, Q = a.Q;
    const int M = a.N * P * Q;
    const int tap = blockIdx.y, r = tap / S, s = tap - r * S;
    const int kt = blockIdx.z / ctiles, ct = blockIdx.z - kt * ctiles;
    const int k0 = kt * 128, c0 = ct * 128;
    const int kn = min(128, K - k0), cn = min(128, C - c0);   // valid channels of this tile
    const int kt32 = (kn + 31) / 32, ct32 = (cn + 31) / 32;
    const int ntile = kt32 * ct32;
    const int mbeg = blockIdx.x * pch, mend = min(M, mbeg + pch);
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);

    bn_fill(a.bn, C, (double)a.N * H * W, s_scale, s_shift);

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (a.dbias != nullptr) && tap == 0 && ct == 0;

    for (int mb = mbeg; mb < mend; mb += BKP) {
        __syncthreads();  // previous step's MFMAs are done with LDS (and BN tables are visible)
        if (tid < BKP) {
            const int m = mb + tid;
            bool okd = m < mend, okx = false;
            int offx = 0;
            if (okd) {
                const int n = m / (P * Q), rem = m - n * (P * Q);
                const int p = rem / Q, q = rem - p * Q;
                const int ih = p * a.stride - a.pad + r, iw = q * a.stride - a.pad + s;
                okx = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
                offx = ((n * H + ih) * W + iw) * C;
            }
            s_pok_dy[tid] = okd; s_poff_dy[tid] = m * K;
            s_pok_x[tid] = okx;  s_poff_x[tid] = offx;
        }
        __syncthreads();
        WT::template stage<false>(s_dy, dy, tid, s_poff_dy, s_pok_dy, k0, K, nullptr, nullptr, 0, 0);
        if (a.bn.mode != FPD_BN_NONE)
            WT::template stage<true>(s_a, x, tid, s_poff_x, s_pok_x, c0, C, s_scale, s_shift, a.bn.relu, c0);
        else
            WT::template stage<false>(s_a, x, tid, s_poff_x, s_pok_x, c0, C, nullptr, nullptr, 0, 0);
        __syncthreads();
        if (do_bias && tid < kn) bsum += WT::colsum(s_dy, tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = wave + 4 * q;
            if (t < ntile) {
                const int ti = t / ct32, tj = t - ti * ct32;
                WT::mma(s_dy, s_a, ti, tj, lane, acc[q]);
            }
        }
    }

    const int R = a.R;
    // no atomics: pixel chunk blockIdx.x stores its accumulator to slab blockIdx.x (a.partial), or -- without slabs the
    // launch has a single chunk -- adds it straight into dw
    float* slab = a.partial != nullptr ? a.partial + (size_t)blockIdx.x * a.partial_stride : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = wave + 4 * q;
        if (t < ntile) {
            const int ti = t / ct32, tj = t - ti * ct32;
            const int c = c0 + tj * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + ti * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
                if (k < K && c < C) {
                    const size_t idx = ((size_t)(k * R + r) * S + s) * C + c;
                    if (slab != nullptr) slab[idx] = acc[q][i];      // summed in a fixed order by fpd_wgrad_reduce()
                    else a.dw[idx] += acc[q][i];                     // no slabs: ONE pixel chunk per launch owns the element
                }
            }
        }
    }
    if (do_bias && tid < kn) {
        if (slab != nullptr) slab[(size_t)K * R * S * C + k0 + tid] = bsum;
        else a.dbias[k0 + tid] += bsum;
    }
}

template <typename T, int TN, int BK>
int launch_conv_t(const fpd_conv_t& a, hipStream_t st) {
    const int M = a.N * a.P * a.Q;
    dim3 grid(cdiv(M, 128), cdiv(a.K, 32 * TN));
    FPD_LAUNCH((conv_mfma_kernel<T, TN, BK>), grid, dim3(256), 0, st, a);
    return 0;
}

template <typename T, int BK>
int launch_conv_tn(const fpd_conv_t& a, hipStream_t st) {
    if (a.K > 64) return launch_conv_t<T, 4, BK>(a, st);
    if (a.K > 32) return launch_conv_t<T, 2, BK>(a, st);
    return launch_conv_t<T, 1, BK>(a, st);
}

}  // namespace

// returns 1 if the shape is not supported by the MFMA path (caller falls back to the direct HIP kernel)
int fpd_conv_mfma_launch(const fpd_conv_t& a, hipStream_t st) {
    if (a.C % 16 != 0 || a.C > FPD_MAXC) return 1;
    if (a.epi == FPD_EPI_BNRELU_BWD && a.K > FPD_MAXC) return 1;
    if (a.dtype == FPD_BF16) {
        if (a.C % 64 == 0) return launch_conv_tn<bf16_t, 64>(a, st);
        if (a.C % 32 == 0) return launch_conv_tn<bf16_t, 32>(a, st);
        return launch_conv_tn<bf16_t, 16>(a, st);
    }
    if (a.C % 32 == 0) return launch_conv_tn<float, 32>(a, st);
    return launch_conv_tn<float, 16>(a, st);
}

// pixel chunks (= slabs) of the generic weight-gradient kernel and the pixels per chunk; 0 chunks = outside its domain
static int wgrad_mfma_chunks(const fpd_wgrad_t& a, int& pch) {
    if (a.C % 16 != 0 || a.K % 16 != 0 || a.C > FPD_MAXC) return 0;
    const int M = a.N * a.P * a.Q;
    const int taps = a.R * a.S, ktiles = cdiv(a.K, 128), ctiles = cdiv(a.C, 128);
    const int bkp = (a.dtype == FPD_BF16) ? 32 : 16;
    int chunks = cdiv(768, taps * ktiles * ctiles);
    chunks = std::max(1, std::min(chunks, cdiv(M, 4 * bkp)));
    pch = cdiv(cdiv(M, chunks), bkp) * bkp;
    return cdiv(M, pch);
}
int fpd_wgrad_mfma_partials(const fpd_wgrad_t& a) {
    int pch;
    return wgrad_mfma_chunks(a, pch);
}

int fpd_wgrad_mfma_launch(const fpd_wgrad_t& a, hipStream_t st) {
    int pch = 0;
    int chunks = wgrad_mfma_chunks(a, pch);
    if (chunks == 0) return 1;
    const int taps = a.R * a.S, ktiles = cdiv(a.K, 128), ctiles = cdiv(a.C, 128);
    if (a.partial == nullptr) {              // no slabs: one chunk owns every element and adds into dw directly
        const int bkp = (a.dtype == FPD_BF16) ? 32 : 16;
        chunks = 1;
        pch = cdiv(a.N * a.P * a.Q, bkp) * bkp;
    }
    dim3 grid(chunks, taps, ktiles * ctiles);
    if (a.dtype == FPD_BF16)
        FPD_LAUNCH((wgrad_mfma_kernel<bf16_t>), grid, dim3(256), 0, st, a, pch, ctiles);
    else
        FPD_LAUNCH((wgrad_mfma_kernel<float>), grid, dim3(256), 0, st, a, pch, ctiles);
    return 0;
}
