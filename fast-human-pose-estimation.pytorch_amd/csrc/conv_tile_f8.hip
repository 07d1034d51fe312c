// conv_tile_f8: the halo-tile implicit-GEMM convolution of conv_tile.hip on the CDNA4 fp8 matrix pipe
// (v_mfma_f32_32x32x16_fp8_fp8, OCP e4m3fn operands, fp32 accumulation) -- BASELINE configs[4]: "HRNet-W32 student
// fp8 weights (CDNA4 fp8 MFMA)".  Forward of stride-1 1x1 / 3x3 "same" convolutions (pose_hrnet.py:28-98 blocks, the
// 1x1 fuse convolutions :199-211, final_layer :320-326 when K % 8 == 0); the data / weight gradients stay on the bf16
// kernels (straight-through: gradients of the bf16 function at the fp8-rounded forward point).
//   weights      e4m3 with ONE fp32 scale per output channel (amax / 448), written by wquant_kernel from the fp32 master
//                weights once per step; the scale multiplies the fp32 accumulator before the epilogue
//   activations  stay bf16 in HBM; the BN+ReLU prologue result is rounded to e4m3 (saturating at +-448, unit scale:
//                post-BN activations are O(1)) ON THE WAY INTO LDS, so the halo tile and both weight buffers take half
//                the LDS bytes and half the LDS read bandwidth of the bf16 kernel
// Same tile geometry and the same epilogue (bias / residual / batch statistics) as conv_tile.hip; LDS rows are BK
// bytes of channels + 16 bytes pad.  Specification: oracle/fp8_ref.py (fake quantisation in torch).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv_epilogue.h"

namespace {

typedef unsigned char f8_t;

__device__ __forceinline__ float clamp448(float v) { return __builtin_amdgcn_fmed3f(v, 448.f, -448.f); }
// 8 floats -> 8 e4m3 bytes (round to nearest even, saturating)
__device__ __forceinline__ uint2 pack_f8x8(const float* f) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[0]), clamp448(f[1]), lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[2]), clamp448(f[3]), lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[4]), clamp448(f[5]), hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(f[6]), clamp448(f[7]), hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }
struct TileGeo8 { int nrows; unsigned mW, mWV; };
__device__ __forceinline__ int qdiv(int v, unsigned magic) { return (int)__umulhi((unsigned)v, magic); }
static inline unsigned magic_of(int d) { return (unsigned)((0x100000000ull / (unsigned long long)d) + 1ull); }

// fp32 master weights [K][RSC] -> e4m3 [K][RSC] + scale[K]: one block per (convolution, output channel)
__global__ __launch_bounds__(256) void wquant_kernel(const fpd_wquant_entry_t* table) {
    __shared__ float s_max[4];
    const fpd_wquant_entry_t e = table[blockIdx.y];
    for (int k = blockIdx.x; k < e.K; k += gridDim.x) {
        const float* w = e.w + (size_t)k * e.RSC;
        float m = 0.f;
        for (int i = threadIdx.x; i < e.RSC; i += blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
        __syncthreads();
        m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        __syncthreads();
        const float scale = m > 0.f ? m / 448.f : 1.f;
        if (threadIdx.x == 0) e.scale[k] = scale;
        f8_t* q = reinterpret_cast<f8_t*>(e.w8) + (size_t)k * e.RSC;
        for (int i = threadIdx.x; i < e.RSC; i += blockDim.x) {
            const float v = clamp448(w[i] / scale);
            q[i] = (f8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, v, 0, false) & 0xff);
        }
    }
}

template <int TN, int BK>
__global__ __launch_bounds__(256, 2) void conv_tile_f8_kernel(const fpd_conv_t a, const f8_t* __restrict__ w8,
                                                              const float* __restrict__ wscale, const TileGeo8 geo) {
    constexpr int VEC = 8;                       // bf16 elements per 16-byte global vector of x
    constexpr int BNT = 32 * TN;
    constexpr int LDB = BK + 16;                 // LDS row pitch in bytes (= fp8 elements)
    constexpr int VPR = BK / VEC, LOG_VPR = ilog2(VPR);
    constexpr int VPW = BK / 16;                 // 16-byte vectors per weight row of a chunk
    constexpr int NVB_TOT = BNT * VPW, NVB = (NVB_TOT + 255) / 256;
    constexpr int NVH = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, pad = a.pad;
    const int M = a.N * H * W, GR = a.N * H;
    const int nrows = geo.nrows, hrows = nrows + R - 1, WP = W + R - 1;
    const int TPX = nrows * W;
    const int zero_px = hrows * WP;
    const int HP = zero_px + 3;
    const int m0 = blockIdx.x * TPX, n0 = blockIdx.y * BNT;
    const int g0 = blockIdx.x * nrows;

    float* s_scale = reinterpret_cast<float*>(smem);
    float* s_shift = s_scale + C;
    float* s_epi = s_shift + C;
    f8_t* sH = reinterpret_cast<f8_t*>(s_epi + 4 * BNT);
    f8_t* sB = sH + ((HP * LDB + 15) & ~15);
    float* stage = reinterpret_cast<float*>(sH);
    double* s_red = reinterpret_cast<double*>(sH);
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);

    const int ml = wave * 32 + (lane & 31);
    const int ti = qdiv(ml, geo.mW), tj = ml - ti * W;
    int ab0, ab1, ab2;
    {
        const int g = g0 + ti;
        const int p = g % H;
        const bool live = g < GR && ml < TPX;
        if (R == 3) {
            ab0 = (live && p - 1 >= 0) ? ((ti + 0) * WP + tj) * LDB : zero_px * LDB;
            ab1 = live ? ((ti + 1) * WP + tj) * LDB : zero_px * LDB;
            ab2 = (live && p + 1 < H) ? ((ti + 2) * WP + tj) * LDB : zero_px * LDB;
        } else {
            ab0 = ab1 = ab2 = live ? (ti * WP + tj) * LDB : zero_px * LDB;
        }
    }

    const int nvtot = hrows * W * VPR;
    const int WV = W * VPR;
    uint4 rh[NVH];
    unsigned hmask = 0;
    auto halo_load = [&](int c0) {
        hmask = 0;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int v = tid + i * 256;
            rh[i] = make_uint4(0, 0, 0, 0);
            if (v < nvtot) {
                const int hr = qdiv(v, geo.mWV);
                const int rem = v - hr * WV;
                const int j = rem >> LOG_VPR, cv = (rem & (VPR - 1)) * VEC;
                const int g = g0 - pad + hr;
                if ((unsigned)g < (unsigned)GR) {
                    rh[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(g * W + j) * C + c0 + cv));
                    hmask |= 1u << i;
                }
            }
        }
    };
    const int cvh = (tid & (VPR - 1)) * VEC;
    const float relu_lo = a.bn.relu ? 0.f : -3.4e38f;
    auto halo_store = [&](int c0) {
        float psc[VEC], psh[VEC];
        if (a.bn.mode != FPD_BN_NONE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { psc[e] = s_scale[c0 + cvh + e]; psh[e] = s_shift[c0 + cvh + e]; }
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int v = tid + i * 256;
            if (v < nvtot) {
                const int hr = qdiv(v, geo.mWV);
                const int j = (v - hr * WV) >> LOG_VPR;
                float f[VEC];
                DT<bf16_t>::unpack(rh[i], f);
                if (a.bn.mode != FPD_BN_NONE) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], psc[e], psh[e]), relu_lo);
                }
                uint2 val = pack_f8x8(f);
                if (!((hmask >> i) & 1u)) val = make_uint2(0, 0);       // rows outside the tensor stay exactly zero
                *reinterpret_cast<uint2*>(sH + (hr * WP + j + pad) * LDB + cvh) = val;
            }
        }
    };
    int b_loff[NVB];
    int b_goff[NVB];
    bool b_ok[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int v = tid + i * 256;
        const int row = v / VPW, col = (v - row * VPW) * 16;
        b_loff[i] = row * LDB + col;
        b_ok[i] = (v < NVB_TOT) && (n0 + row < K);
        b_goff[i] = (n0 + row) * (R * R) * C + col;
    }
    uint4 rb[NVB];
    const int RS = R * R;
    auto b_load = [&](int tap, int c0) {
        const int toff = tap * C + c0;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            rb[i] = make_uint4(0, 0, 0, 0);
            if (b_ok[i]) rb[i] = *reinterpret_cast<const uint4*>(w8 + (b_goff[i] + toff));
        }
    };
    auto b_store = [&](int buf) {
        f8_t* dst = sB + buf * BNT * LDB;
#pragma unroll
        for (int i = 0; i < NVB; ++i)
            if (tid + i * 256 < NVB_TOT) *reinterpret_cast<uint4*>(dst + b_loff[i]) = rb[i];
    };
    auto tap_mma = [&](const f8_t* arow, const f8_t* wt, f32x16* acc) {
        const int koff = 8 * (lane >> 5);
        const f8_t* brow = wt + (lane & 31) * LDB + koff;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const long av = *reinterpret_cast<const long*>(arow + kk * 16 + koff);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const long bv = *reinterpret_cast<const long*>(brow + tn * 32 * LDB + kk * 16);
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(av, bv, acc[tn], 0, 0, 0);
            }
        }
    };

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    const int nchunk = C / BK;
    halo_load(0);
    b_load(0, 0);
    {   // zero border columns of every halo row + the three zero pixels (8-byte vectors over the BK payload bytes)
        const uint2 z = make_uint2(0, 0);
        constexpr int ZV = BK / 8;
        const int nb = (R == 3) ? 2 * hrows : 0;
        for (int v = tid; v < (nb + 3) * ZV; v += 256) {
            const int pz = v / ZV, cv = (v - pz * ZV) * 8;
            const int px = pz < nb ? ((pz >> 1) * WP + ((pz & 1) ? WP - 1 : 0)) : zero_px + (pz - nb);
            *reinterpret_cast<uint2*>(sH + px * LDB + cv) = z;
        }
    }
    bn_fill(a.bn, C, (double)M, s_scale, s_shift);
    conv_epi_tables<BNT>(a, n0, M, s_epi);

    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = ch * BK;
        const bool more = ch + 1 < nchunk;
        __syncthreads();
        halo_store(c0);
        b_store(0);
        __syncthreads();
        if (more) halo_load(c0 + BK);
        if (RS > 1) b_load(1, c0); else if (more) b_load(0, c0 + BK);
        int r = 0, s = 0;
        for (int tap = 0; tap < RS; ++tap) {
            const int ab = (r == 0) ? ab0 : ((r == 1) ? ab1 : ab2);
            tap_mma(sH + ab + s * LDB, sB + (tap & 1) * BNT * LDB, acc);
            if (tap + 1 < RS) {
                b_store((tap + 1) & 1);
                if (tap + 2 < RS) b_load(tap + 2, c0); else if (more) b_load(0, c0 + BK);
                __syncthreads();
            }
            if (++s == R) { s = 0; ++r; }
        }
    }
    // per-output-channel weight scale: a lane's accumulator column is channel n0 + tn*32 + (lane & 31)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int k = n0 + tn * 32 + (lane & 31);
        const float sc = k < K ? wscale[k] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] *= sc;
    }
    const int Mlim = min(M, m0 + TPX);
    conv_epilogue_vec<bf16_t, TN>(a, acc, m0, n0, Mlim, s_epi, stage, s_red);   // starts with a barrier
}

constexpr size_t LDS_MAX = 160 * 1024;
static int tile_rows(int W) { return std::max(1, 128 / W); }

template <int TN, int BK>
int launch_f8(const fpd_conv_t& a, const void* w8, const float* wscale, hipStream_t st) {
    constexpr int LDB = BK + 16;
    const int hrows = tile_rows(a.W) + a.R - 1, WP = a.W + a.R - 1;
    const size_t tile = (((size_t)(hrows * WP + 3) * LDB + 15) & ~(size_t)15) + (size_t)2 * 32 * TN * LDB;
    const size_t epi = std::max((size_t)128 * (32 * TN + 4) * sizeof(float), (size_t)4 * 32 * TN * 2 * sizeof(double));
    const size_t lds = (size_t)(2 * a.C + 4 * 32 * TN) * sizeof(float) + std::max(tile, epi);
    if (lds > LDS_MAX) return 1;
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&conv_tile_f8_kernel<TN, BK>), lds)) return rc_;
    TileGeo8 g;
    g.nrows = tile_rows(a.W);
    g.mW = magic_of(a.W);
    g.mWV = magic_of(a.W * (BK / 8));
    dim3 grid(cdiv(a.N * a.H, g.nrows), cdiv(a.K, 32 * TN));
    FPD_LAUNCH((conv_tile_f8_kernel<TN, BK>), grid, dim3(256), lds, st, a, reinterpret_cast<const f8_t*>(w8), wscale, g);
    return 0;
}

template <int BK>
int launch_f8_tn(const fpd_conv_t& a, const void* w8, const float* wscale, hipStream_t st) {
    const int mt = cdiv(a.N * a.H, tile_rows(a.W));
    int tn = a.K > 64 ? 4 : (a.K > 32 ? 2 : 1);
    while (tn > 1 && mt * cdiv(a.K, 32 * tn) < 128) tn >>= 1;
    if (tn == 4) return launch_f8<4, BK>(a, w8, wscale, st);
    if (tn == 2) return launch_f8<2, BK>(a, w8, wscale, st);
    return launch_f8<1, BK>(a, w8, wscale, st);
}

}  // namespace

// shapes the fp8 halo-tile kernel covers (a subset of conv_tile.hip's domain): bf16 storage, stride-1 "same" 1x1 / 3x3,
// rows of at most 128 pixels, C a multiple of 32, K a multiple of 8 (vector epilogue), plain epilogue
bool fpd_conv_f8_domain(const fpd_conv_t& a) {
    if (a.dtype != FPD_BF16 || a.epi != FPD_EPI_PLAIN) return false;
    if (a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return false;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < 2) return false;
    if (a.C % 32 != 0 || a.C > FPD_MAXC || a.K % 8 != 0) return false;
    const int bk = (a.C % 64 == 0) ? 64 : 32;
    return (tile_rows(a.W) + a.R - 1) * a.W * (bk / 8) <= 2048;       // the halo fits the 8 staging vectors per thread
}

// 1 = outside the domain
int fpd_conv_tile_f8_launch(const fpd_conv_t& a, const void* w8, const float* wscale, hipStream_t st) {
    if (!fpd_conv_f8_domain(a)) return 1;
    if (a.C % 64 == 0) return launch_f8_tn<64>(a, w8, wscale, st);
    return launch_f8_tn<32>(a, w8, wscale, st);
}

int fpd_weight_quant_f8_launch(const fpd_wquant_entry_t* table, int n, hipStream_t st) {
    if (n <= 0) return 0;
    FPD_LAUNCH(wquant_kernel, dim3(64, n), dim3(256), 0, st, table);
    return 0;
}
