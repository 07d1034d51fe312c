// wgrad_tile: weight gradient of a stride-1 1x1 / 3x3 "same" convolution,
//     dw[k][r][s][c] += sum_m dy[m][k] * a(x)[m + (r-1, s-1)][c] ,   dbias[k] += sum_m dy[m][k]
// with BOTH operands kept pixel-major in LDS exactly as they sit in HBM (NHWC): per 128-pixel tile the block stages
// the input halo (BatchNorm+ReLU applied once per element) and the dy tile with 16-byte copies, and every filter tap
// reads its MFMA operands from the same halo at a shifted pixel offset.  The pixel (reduction) axis is turned into
// the MFMA K axis by the gfx950 transposing LDS read ds_read_b64_tr_b16 (bf16) -- within a 16-lane group lane s
// fetches 8 bytes of row s/4 and receives one column of the 4x16 block -- so no transposing stores are needed; the
// fp32 MFMA takes one k per lane and reads the pixel-major tile directly.
// A 512-thread block owns a 64x64 (bf16) / 32x32 (fp32) block of the K x C plane for ALL taps (8 waves, <= 5
// accumulator tiles each, so two waves fit per SIMD and hide each other's staging / LDS latency),
// walks pixel tiles persistently with register prefetch of the next tile, and flushes once with fp32 atomics.
// Replaces autograd of nn.Conv2d (weight/bias gradient) in /root/reference/lib/models/hourglass.py:20,23,27.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "mfma_frag.h"

namespace {

template <typename T>
struct WFrag;

template <>
struct WFrag<bf16_t> {
    static constexpr int KSTEP = 16;     // pixels per MFMA
    typedef bf16x8 frag_t;
    // tile: pixel-major [pixel][LD]; returns the 32(ch) x 16(pixel) operand: lane L -> channel ch0 + (L&31), pixels
    // pix0 + 8*(L>>5) .. +7
    static __device__ __forceinline__ frag_t load(const bf16_t* tile, int LD, int pix0, int ch0, int lane) {
        return tr_frag_bf16(tile, LD, pix0, ch0, lane);
    }
    static __device__ __forceinline__ void mma(const frag_t& a, const frag_t& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
};

template <>
struct WFrag<float> {
    static constexpr int KSTEP = 2;
    typedef float frag_t;
    static __device__ __forceinline__ frag_t load(const float* tile, int LD, int pix0, int ch0, int lane) {
        return tile[(pix0 + (lane >> 5)) * LD + ch0 + (lane & 31)];
    }
    static __device__ __forceinline__ void mma(const frag_t& a, const frag_t& b, f32x16& acc) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
};

template <typename T, int R>
__global__ __launch_bounds__(512) void wgrad_tile_kernel(const fpd_wgrad_t a, const int logW, const int ctiles,
                                                         const int mtiles) {
    using WF = WFrag<T>;
    constexpr int VEC = DT<T>::VEC;
    constexpr int CW = 128 / (int)sizeof(T);          // channels per tile row: 64 (bf16) / 32 (fp32) = 128 bytes
    constexpr int LD = CW + 16 / (int)sizeof(T);
    constexpr int KC1 = CW / 32, KC = KC1 * KC1;      // (ki,ci) sub-tiles of 32x32: 4 / 1
    constexpr int KSTEP = WF::KSTEP;
    constexpr int BLK = 512, NW = BLK / 64;           // 8 waves: two per SIMD, sharing one staged tile
    constexpr int NVH = 2048 / BLK, NVD = 128 * 8 / BLK;   // halo / dy 16-byte vectors per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, C = a.C, K = a.K, pad = a.pad;
    const int GR = a.N * H;
    const int nrows = 128 >> logW, hrows = nrows + R - 1, WP = W + R - 1;
    const int HP = hrows * WP;
    const int kt = blockIdx.y / ctiles, ct = blockIdx.y - kt * ctiles;
    const int k0 = kt * CW, c0 = ct * CW;
    const int kn = min(CW, K - k0), cn = min(CW, C - c0);     // valid channels (multiples of VEC)
    const int vpr_c = cn / VEC, vpr_k = kn / VEC;             // vectors per pixel actually present

    float* s_scale = reinterpret_cast<float*>(smem);
    float* s_shift = s_scale + CW;
    T* sH = reinterpret_cast<T*>(s_shift + CW);
    T* sD = sH + HP * LD;
    T* sZ = sD + 128 * LD;                              // 16 all-zero pixels: operand of taps whose row is outside the image
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);

    // zero the whole tile region once: border columns, unused channel columns and ragged rows then stay zero
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        const int nv = (HP + 128 + 16) * LD / VEC;
        for (int v = tid; v < nv; v += BLK) *reinterpret_cast<uint4*>(sH + v * VEC) = z;
    }
    if (a.bn.mode != FPD_BN_NONE) {
        for (int c = tid; c < cn; c += BLK) {
            float sc, sh, mu, is;
            bn_coef(a.bn, c0 + c, C, (double)a.N * H * W, sc, sh, mu, is);
            s_scale[c] = sc; s_shift[c] = sh;
        }
    }

    // ---- this wave's accumulator slots: u = wave + 4*i -> (tap, ki, ci) ----
    constexpr int RS = R * R, NT = RS * KC;
    constexpr int NS = (NT + NW - 1) / NW;    // accumulator slots per wave: 5 (3x3 bf16), 2 (3x3 fp32), 1 (1x1)
    const int kc = wave % KC;                 // fixed per wave (KC in {1,4})
    const int ki = kc / KC1, ci = kc % KC1;
    int tap_off[NS], slot_r[NS];
    bool slot_on[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int u = wave + NW * i;
        slot_on[i] = u < NT;
        const int tap = slot_on[i] ? u / KC : 0;      // idle slots compute on tap 0 and are never flushed
        const int r = tap / R, s = tap - r * R;
        slot_r[i] = r;
        tap_off[i] = (r * WP + s) * LD;
    }
    f32x16 acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (a.dbias != nullptr) && ct == 0;
    const int bch = tid % CW, bpart = tid / CW;        // bias partial sums: channel, pixel part (BLK/CW parts)
    constexpr int BPARTS = BLK / CW, BPIX = 128 / BPARTS;

    // ---- staging: everything that does not depend on the pixel tile is computed once per thread ----
    const int nvh = (hrows << logW) * vpr_c;          // halo vectors of one tile
    const int nvd = 128 * vpr_k;
    int h_goff[NVH], h_loff[NVH], h_row[NVH];         // global element offset (tile 0), LDS element offset, halo row
    int d_goff[NVD], d_loff[NVD];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int v = tid + i * BLK;
        const int px = v / vpr_c, cv = (v - px * vpr_c) * VEC;
        const int hr = px >> logW, j = px & (W - 1);
        h_row[i] = (v < nvh) ? hr - pad : -(1 << 28);             // flattened row relative to the tile's first row
        h_goff[i] = ((hr - pad) * W + j) * C + c0 + cv;
        h_loff[i] = (hr * WP + j + pad) * LD + cv;
    }
#pragma unroll
    for (int i = 0; i < NVD; ++i) {
        const int v = tid + i * BLK;
        const int px = v / vpr_k, cv = (v - px * vpr_k) * VEC;
        d_goff[i] = (v < nvd) ? px * K + k0 + cv : -1;
        d_loff[i] = px * LD + cv;
    }
    // a thread always stages the same VEC channels (BLK % vpr_c == 0): keep their BN coefficients in registers
    float psc[VEC], psh[VEC];
    {
        const int cvh = ((tid % vpr_c) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { psc[e] = 1.f; psh[e] = 0.f; }
        if (a.bn.mode != FPD_BN_NONE) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < VEC; ++e) { psc[e] = s_scale[cvh + e]; psh[e] = s_shift[cvh + e]; }
        }
    }
    uint4 rh[NVH], rd[NVD];
    unsigned hmask = 0;
    auto loads = [&](int tile) {
        const int g0 = tile * nrows;
        const size_t xbase = (size_t)g0 * W * C;
        const size_t dbase = (size_t)g0 * W * K;
        const int mleft = (GR - g0) * W;                          // pixels left from the tile start
        hmask = 0;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            rh[i] = make_uint4(0, 0, 0, 0);
            if ((unsigned)(g0 + h_row[i]) < (unsigned)GR) {
                rh[i] = *reinterpret_cast<const uint4*>(x + xbase + h_goff[i]);
                hmask |= 1u << i;
            }
        }
#pragma unroll
        for (int i = 0; i < NVD; ++i) {
            rd[i] = make_uint4(0, 0, 0, 0);
            if (d_goff[i] >= 0 && d_goff[i] < mleft * K) rd[i] = *reinterpret_cast<const uint4*>(dy + dbase + d_goff[i]);
        }
    };
    auto stores = [&]() {
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (h_row[i] > -(1 << 27)) {
                uint4 val = rh[i];
                if (a.bn.mode != FPD_BN_NONE) {
                    float f[VEC];
                    DT<T>::unpack(val, f);
                    const bool ok = (hmask >> i) & 1u;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float t = bn_act(f[e], psc[e], psh[e], a.bn.relu);
                        f[e] = ok ? t : 0.f;
                    }
                    val = DT<T>::pack(f);
                }
                *reinterpret_cast<uint4*>(sH + h_loff[i]) = val;
            }
        }
#pragma unroll
        for (int i = 0; i < NVD; ++i)
            if (d_goff[i] >= 0) *reinterpret_cast<uint4*>(sD + d_loff[i]) = rd[i];
    };

    int tile = blockIdx.x;
    if (tile < mtiles) loads(tile);
    for (; tile < mtiles; tile += gridDim.x) {
        __syncthreads();                    // previous tile consumed (first time: zero fill + tables visible)
        stores();
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < mtiles) loads(next);     // in flight during this tile's MFMAs
        const int g0 = tile * nrows;
        if (do_bias) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll 4
            for (int p = 0; p < BPIX; p += 2) {
                s0 += DT<T>::ld(sD + (bpart * BPIX + p) * LD + bch);
                s1 += DT<T>::ld(sD + (bpart * BPIX + p + 1) * LD + bch);
            }
            bsum += s0 + s1;
        }
#pragma unroll 1
        for (int pix0 = 0; pix0 < 128; pix0 += KSTEP) {
            const int ti = pix0 >> logW, j0 = pix0 & (W - 1);
            const int g = g0 + ti;
            if (g >= GR) break;             // ragged last tile (wave-uniform)
            const int p = g % H;
            const typename WF::frag_t af = WF::load(sD, LD, pix0, ki * 32, lane);
            const T* hbase = sH + (ti * WP + j0) * LD;
            typename WF::frag_t bf[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {                  // branch-free: rows outside the image read the zero block
                const bool inside = (unsigned)(p + slot_r[i] - pad) < (unsigned)H;   // uniform for the k-step
                bf[i] = WF::load(inside ? hbase + tap_off[i] : sZ, LD, 0, ci * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) WF::mma(af, bf[i], acc[i]);
        }
    }

    // ---- flush ----
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (slot_on[i]) {
            const int u = wave + NW * i;
            const int tap = u / KC;
            const int c = c0 + ci * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + ki * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (k < K && c < C) atomicAdd(a.dw + ((size_t)k * RS + tap) * C + c, acc[i][e]);
            }
        }
    }
    if (do_bias && bch < kn) atomicAdd(a.dbias + k0 + bch, bsum);
}

template <typename T, int R>
int launch_wt(const fpd_wgrad_t& a, int logW, hipStream_t st) {
    constexpr int CW = 128 / (int)sizeof(T);
    constexpr int LD = CW + 16 / (int)sizeof(T);
    const int nrows = 128 >> logW, hrows = nrows + a.R - 1, WP = a.W + a.R - 1;
    const size_t lds = 2 * CW * sizeof(float) + (size_t)(hrows * WP + 128 + 16) * LD * sizeof(T);
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tile_kernel<T, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fpd_fail(-100 - (int)e, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        configured = lds;
    }
    const int mtiles = cdiv(a.N * a.H * a.W, 128);
    const int ktiles = cdiv(a.K, CW), ctiles = cdiv(a.C, CW);
    const int gy = ktiles * ctiles;
    int target = 128;   // persistent blocks: each flushes its whole accumulator with device-scope atomics once
    if (const char* e = getenv("FPD_WGRAD_BLOCKS")) target = atoi(e);
    const int gx = std::max(1, std::min(mtiles, cdiv(target, gy)));
    hipLaunchKernelGGL((wgrad_tile_kernel<T, R>), dim3(gx, gy), dim3(512), lds, st, a, logW, ctiles, mtiles);
    return 0;
}

}  // namespace

// returns 1 when the shape is outside this kernel's domain
int fpd_wgrad_tile_launch(const fpd_wgrad_t& a, hipStream_t st) {
    if (a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return 1;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < 16 || (a.W & (a.W - 1)) != 0) return 1;
    if (a.C % 16 != 0 || a.K % 16 != 0) return 1;
    int logW = 0;
    while ((1 << logW) < a.W) ++logW;
    const int hrows = (128 >> logW) + a.R - 1;
    const int vec = a.dtype == FPD_BF16 ? 8 : 4, cw = a.dtype == FPD_BF16 ? 64 : 32;
    if (hrows * a.W * (std::min(a.C, cw) / vec) > 2048) return 1;
    if (a.R == 3) return a.dtype == FPD_BF16 ? launch_wt<bf16_t, 3>(a, logW, st) : launch_wt<float, 3>(a, logW, st);
    return a.dtype == FPD_BF16 ? launch_wt<bf16_t, 1>(a, logW, st) : launch_wt<float, 1>(a, logW, st);
}
