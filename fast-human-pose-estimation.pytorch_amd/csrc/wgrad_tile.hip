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
// walks pixel tiles persistently with register prefetch of the next tile, and flushes once: to a private slab that
// fpd_wgrad_reduce() sums in a fixed order, or (no slabs given: one block per tile) straight into dw -- no atomics.
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

// H4 (bf16, 3x3, W a multiple of 4 but not of 16: HRNet's 24- / 12-wide maps): a 16-pixel k-step may straddle image rows,
// so every 4-pixel group of the transposing read (the granularity at which a lane addresses rows) gets its own halo base,
// row-validity test and liveness instead of one per k-step.
// SMALL (bf16, 3x3, C <= 32 and K <= 32: HRNet-W32's first branch): only the (0, 0) 32x32 sub-tile of the block's K x C plane
// is live, so the nine taps -- not 9 x 4 sub-tiles, three quarters of them zero -- are spread over the waves.
template <typename T, int R, int TP, bool H4, bool SMALL = false>
__global__ __launch_bounds__(512) void wgrad_tile_kernel(const fpd_wgrad_t a, const int nrows, const unsigned mW, const int ctiles,
                                                         const int mtiles) {
    using WF = WFrag<T>;
    constexpr int VEC = DT<T>::VEC;
    constexpr int CW = 128 / (int)sizeof(T);          // channels per tile row: 64 (bf16) / 32 (fp32) = 128 bytes
    constexpr int LD = CW + 16 / (int)sizeof(T);
    constexpr int KC1 = SMALL ? 1 : CW / 32, KC = KC1 * KC1;      // (ki,ci) sub-tiles of 32x32: 4 (bf16) / 1 (fp32, SMALL)
    constexpr int KSTEP = WF::KSTEP;
    constexpr int BLK = 512, NW = BLK / 64;           // 8 waves: two per SIMD, sharing one staged tile
    constexpr int NVH = 2048 / BLK, NVD = TP * 8 / BLK;    // halo / dy 16-byte vectors per thread (TP pixels per tile)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, W = a.W, C = a.C, K = a.K, pad = a.pad;
    const int GR = a.N * H;
    const unsigned mH = (unsigned)((0x100000000ull / (unsigned long long)H) + 1ull);   // g / H by multiply-high (H4 only)
    // a tile = nrows whole image rows = TPX <= TP pixels (W a power of two: TPX == TP; HRNet's 96- / 48-wide maps: 96 of 128)
    const int hrows = nrows + R - 1, WP = W + R - 1;
    const int HP = hrows * WP, TPX = nrows * W;
    const int kt = blockIdx.y / ctiles, ct = blockIdx.y - kt * ctiles;
    const int k0 = kt * CW, c0 = ct * CW;
    const int kn = min(CW, K - k0), cn = min(CW, C - c0);     // valid channels (multiples of VEC)
    const int vpr_c = cn / VEC, vpr_k = kn / VEC;             // vectors per pixel actually present

    float* s_scale = reinterpret_cast<float*>(smem);
    float* s_shift = s_scale + CW;
    T* sH = reinterpret_cast<T*>(s_shift + CW);
    T* sD = sH + HP * LD;
    T* sZ = sD + TP * LD;                              // 16 all-zero pixels: operand of taps whose row is outside the image
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ dy = reinterpret_cast<const T*>(a.dy);

    // zero the whole tile region once: border columns, unused channel columns and ragged rows then stay zero
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        const int nv = (HP + TP + 16) * LD / VEC;
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
    constexpr bool KSPLIT = (NT <= NW / 2);   // 1x1: fewer tiles than waves -> the two wave groups split the pixel steps
    const int kgrp = KSPLIT ? wave / (NW / 2) : 0;
    const int kc = wave % KC;                 // fixed per wave (KC in {1,4})
    const int ki = kc / KC1, ci = kc % KC1;
    int tap_off[NS], slot_r[NS];
    bool slot_on[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int u = (KSPLIT ? wave % (NW / 2) : wave) + NW * i;
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
    constexpr int BPARTS = BLK / CW, BPIX = TP / BPARTS;

    // ---- staging: everything that does not depend on the pixel tile is computed once per thread ----
    const int nvh = hrows * W * vpr_c;                // halo vectors of one tile
    const int nvd = TPX * vpr_k;
    int h_goff[NVH], h_loff[NVH], h_row[NVH];         // global element offset (tile 0), LDS element offset, halo row
    int d_goff[NVD], d_loff[NVD];
#pragma unroll
    for (int i = 0; i < NVH; ++i) {
        const int v = tid + i * BLK;
        const int px = v / vpr_c, cv = (v - px * vpr_c) * VEC;
        const int hr = px / W, j = px - hr * W;
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
        // one k-step = KSTEP pixels: operand fetch (transposing LDS reads) then NS MFMAs.  With many slots (3x3) the
        // MFMAs of a step cover the next step's LDS latency; with one slot (1x1) several steps are unrolled instead.
        auto kstep = [&](int pix0) {
            const int ti = (int)__umulhi((unsigned)pix0, mW), j0 = pix0 - ti * W;      // k-steps never straddle rows (KSTEP | W)
            const int g = g0 + ti;
            const int p = g % H;
            const bool live = g < GR;                       // ragged last tile: rows past the tensor contribute zeros
            const typename WF::frag_t af = WF::load(live ? sD : sZ, LD, live ? pix0 : 0, ki * 32, lane);
            const T* hbase = sH + (ti * WP + j0) * LD;
            typename WF::frag_t bf[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {                  // branch-free: rows outside the image read the zero block
                const bool inside = (unsigned)(p + slot_r[i] - pad) < (unsigned)H;   // uniform for the k-step
                bf[i] = WF::load(inside ? hbase + tap_off[i] : sZ, LD, 0, ci * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) WF::mma(af, bf[i], acc[i]);
        };
        // H4: per-lane 4-pixel groups q = 0 (rows +0..3 of this lane's half), 1 (rows +4..7)
        auto kstep4 = [&](int pix0) {
            if constexpr (H4) {
                const int s16 = lane & 15, gq = (lane >> 4) & 1, half = lane >> 5;
                const int inrow = (s16 >> 2) * LD + 16 * gq + 4 * (s16 & 3);        // this lane's row / column inside a 4 x 32 group
                const bf16_t* ap[2];
                int hb[2], prow[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int px = pix0 + 8 * half + 4 * q;
                    const int ti = (int)__umulhi((unsigned)px, mW), j0 = px - ti * W;
                    const int g = g0 + ti;
                    const int p = g - (int)__umulhi((unsigned)g, mH) * H;
                    const bool live = g < GR && px < TPX;
                    ap[q] = (live ? sD + px * LD : sZ) + inrow + ki * 32;
                    hb[q] = (ti * WP + j0) * LD + inrow + ci * 32;
                    prow[q] = live ? p : -(1 << 20);
                }
                union { struct { s16x4 a, b; } h; bf16x8 f; } ua;
                ua.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap[0]));
                ua.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(ap[1]));
                bf16x8 bf[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    union { struct { s16x4 a, b; } h; bf16x8 f; } ub;
                    const bool in0 = (unsigned)(prow[0] + slot_r[i] - pad) < (unsigned)H;
                    const bool in1 = (unsigned)(prow[1] + slot_r[i] - pad) < (unsigned)H;
                    const bf16_t* b0 = in0 ? sH + hb[0] + tap_off[i] : sZ + inrow;
                    const bf16_t* b1 = in1 ? sH + hb[1] + tap_off[i] : sZ + inrow;
                    ub.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b0));
                    ub.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(b1));
                    bf[i] = ub.f;
                }
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.f, bf[i], acc[i], 0, 0, 0);
            }
        };
        if (H4) {
#pragma unroll 1
            for (int pix0 = 0; pix0 < TPX; pix0 += KSTEP) kstep4(pix0);
        } else if (NS == 1) {
#pragma unroll 4
            for (int pix0 = (KSPLIT ? kgrp * KSTEP : 0); pix0 < TPX; pix0 += (KSPLIT ? 2 : 1) * KSTEP) kstep(pix0);
        } else {
#pragma unroll 1
            for (int pix0 = 0; pix0 < TPX; pix0 += KSTEP) kstep(pix0);
        }
    }

    // ---- flush ----
    // No atomics anywhere: with slabs (a.partial) every persistent block stores its accumulator to its own slab and
    // fpd_wgrad_reduce() adds the slabs in a fixed order; without slabs the launch has ONE block per (k, c) tile
    // (gridDim.x == 1, see fpd_wgrad_tile_launch), which owns its dw / dbias elements and adds to them directly.
    // Either way a weight gradient is a fixed-order sum: identical bytes run to run.
    const bool slabs = a.partial != nullptr;
    if (KSPLIT && !slabs) {
        // the two wave groups hold partial sums of the SAME elements: group 1 hands its accumulator over through LDS
        float* s_x = reinterpret_cast<float*>(sH);       // >= (NW/2) * 64 * 16 floats: the tile region is free now
        __syncthreads();
        if (kgrp == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s_x[((wave - NW / 2) * 16 + e) * 64 + lane] = acc[0][e];
        }
        __syncthreads();
        if (kgrp == 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][e] += s_x[(wave * 16 + e) * 64 + lane];
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        if (slot_on[i] && (slabs || kgrp == 0)) {
            const int u = (KSPLIT ? wave % (NW / 2) : wave) + NW * i;
            const int tap = u / KC;
            const int c = c0 + ci * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + ki * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (k < K && c < C) {
                    const size_t idx = ((size_t)k * RS + tap) * C + c;
                    if (slabs) {
                        // one slab per persistent block (two when the wave groups split the pixel steps)
                        float* slab = a.partial + (size_t)(blockIdx.x * (KSPLIT ? 2 : 1) + kgrp) * a.partial_stride;
                        slab[idx] = acc[i][e];
                    } else {
                        a.dw[idx] += acc[i][e];
                    }
                }
            }
        }
    }
    if (do_bias) {
        // block-level reduction of the per-thread partial sums in a FIXED order (parts 0 .. BPARTS-1 per channel)
        float* s_bias = reinterpret_cast<float*>(sD);       // [BPARTS][CW]
        __syncthreads();
        s_bias[bpart * CW + bch] = bsum;
        __syncthreads();
        if (tid < kn) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < BPARTS; ++q) tot += s_bias[q * CW + tid];
            if (slabs) {                         // bias partials live behind the weight slab: [K*R*S*C .. +K)
                const size_t boff = (size_t)K * RS * C + k0 + tid;
                a.partial[(size_t)(blockIdx.x * (KSPLIT ? 2 : 1)) * a.partial_stride + boff] = tot;
                if (KSPLIT) a.partial[(size_t)(blockIdx.x * 2 + 1) * a.partial_stride + boff] = 0.f;
            } else {
                a.dbias[k0 + tid] += tot;
            }
        }
    }
}

struct WtGrid { int nrows, tp, gx, gy, ctiles, mtiles, slabs; bool h4; };

// single source of truth for the launch geometry (also tells the caller how many partial slabs will be written)
bool wt_grid(const fpd_wgrad_t& a, WtGrid& g) {
    if (a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return false;
    // pixels per MFMA k-step: 16 (bf16) / 2 (fp32).  A k-step must stay inside one image row -- except in the bf16 H4 variant
    // (3x3, W % 4 == 0), which resolves rows per 4-pixel group, and for bf16 1x1 (no halo: the tile is flat, ragged rows are
    // zero on the dy side)
    const int kstep = a.dtype == FPD_BF16 ? 16 : 2;
    const bool row_aligned = a.W % kstep == 0;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < (row_aligned ? 16 : 4)) return false;
    if (!row_aligned && !(a.dtype == FPD_BF16 && a.W % 4 == 0)) return false;
    g.h4 = !row_aligned && a.R == 3;
    if (a.C % 16 != 0 || a.K % 16 != 0) return false;
    const int vec = a.dtype == FPD_BF16 ? 8 : 4, cw = a.dtype == FPD_BF16 ? 64 : 32;
    // 1x1: no halo, HBM-bound -> 256-pixel tiles double the bytes in flight per block and halve the barriers
    g.tp = (a.R == 1 && a.N * a.H * a.W >= 256 * 256 && 256 % a.W == 0) ? 256 : 128;
    g.nrows = g.tp / a.W;
    if ((g.nrows + a.R - 1) * a.W * (std::min(a.C, cw) / vec) > 2048) return false;
    g.mtiles = cdiv(a.N * a.H, g.nrows);
    g.ctiles = cdiv(a.C, cw);
    g.gy = cdiv(a.K, cw) * g.ctiles;
    // persistent blocks, each flushing its whole accumulator once: 3x3 = 147 KB per block -> few blocks;
    // 1x1 = 16 KB per block and HBM-bound -> enough blocks to keep every CU streaming
    int target = (a.R == 1) ? 256 : 128;
    // C, K <= 32 (the SMALL variant: one live 32x32 sub-tile, nine taps): a slab is 37 KB, not 147 KB, and the launch is bound by
    // the per-tile staging of its 128 blocks -- the hourglass' 3x3 32->32 at 128x128 is one of the last launches of the backward
    // (158 us, exposed in front of Adam): four times the blocks
    if (a.R == 3 && a.dtype == FPD_BF16 && a.C <= 32 && a.K <= 32) target = std::max(128, std::min(512, g.mtiles / 4));      // (>= 4 tiles per block: HRNet's 64x48 maps keep 192)
    // (the FPD_WGRAD_BLOCKS / _1 / _3 knobs of rounds 2-5 are gone: 256 blocks for 1x1, 128 for 3x3 measured best)
    g.gx = std::max(1, std::min(g.mtiles, cdiv(target, g.gy)));
    // 1x1 on the small maps: a block per 128-pixel tile meant 128 slabs of 32 KB for an 8 192-pixel problem (34 MB per gradient
    // class and step, r05 slab inventory) at the launch-latency floor either way: at least four tiles per block
    if (a.R == 1 && g.mtiles <= 256) g.gx = std::max(1, std::min(g.gx, cdiv(g.mtiles, 4)));
    const int nt = a.R * a.R * (cw / 32) * (cw / 32);
    g.slabs = g.gx * (nt <= 4 ? 2 : 1);          // KSPLIT kernels write one slab per wave group
    return true;
}

template <typename T, int R, int TP, bool H4 = false, bool SMALL = false>
int launch_wt(const fpd_wgrad_t& a, const WtGrid& g, hipStream_t st) {
    constexpr int CW = 128 / (int)sizeof(T);
    constexpr int LD = CW + 16 / (int)sizeof(T);
    const int hrows = g.nrows + a.R - 1, WP = a.W + a.R - 1;
    const size_t lds = 2 * CW * sizeof(float) + (size_t)(hrows * WP + TP + 16) * LD * sizeof(T);
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&wgrad_tile_kernel<T, R, TP, H4, SMALL>), lds)) return rc_;
    FPD_LAUNCH((wgrad_tile_kernel<T, R, TP, H4, SMALL>), dim3(g.gx, g.gy), dim3(512), lds, st, a, g.nrows,
                       (unsigned)((0x100000000ull / (unsigned long long)a.W) + 1ull), g.ctiles, g.mtiles);
    return 0;
}

// dw[i] += sum_b partial[b*stride + i]: second stage of the weight-gradient reduction for a table of convolutions
__global__ __launch_bounds__(256) void wreduce_kernel(const fpd_wreduce_entry_t* table) {
    const fpd_wreduce_entry_t e = table[blockIdx.y];
    const int64_t n4 = e.n >> 2;                 // 16-byte vectors; the (rare) tail is handled below
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(e.dw + 4 * i);
        const float* p = e.partial + 4 * i;
        // The slabs are ADDED one after the other in index order (the result must not depend on the launch geometry of this
        // kernel), but REQUESTED sixteen at a time: with four in flight a thread paid count / 4 memory latencies in a row
        // (128 slabs: 29 us for a few hundred KB, r04 trace).
        int b = 0;
        for (; b + 16 <= e.count; b += 16) {
            f32x4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(b + u) * e.stride);
#pragma unroll
            for (int u = 0; u < 16; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
        }
        for (; b + 4 <= e.count; b += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (size_t)(b + u) * e.stride);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
        }
        for (; b < e.count; ++b) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)b * e.stride);
            acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
        }
        *reinterpret_cast<f32x4*>(e.dw + 4 * i) = acc;
    }
    // tail (n not a multiple of 4: e.g. the 17 biases of an HRNet head)
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < (e.n & 3)) {
        const int64_t i = 4 * n4 + threadIdx.x;
        float acc = e.dw[i];
        for (int b = 0; b < e.count; ++b) acc += e.partial[(size_t)b * e.stride + i];
        e.dw[i] = acc;
    }
}

}  // namespace

// returns 1 when the shape is outside this kernel's domain
int fpd_wgrad3_launch(const fpd_wgrad_t& a, hipStream_t st);      // wgrad3.hip: the student's 3x3 64 -> 64 with slabs
int fpd_wgrad3_partials(const fpd_wgrad_t& a);

int fpd_wgrad_tile_launch(const fpd_wgrad_t& a, hipStream_t st) {
    if (const int rc3 = fpd_wgrad3_launch(a, st); rc3 != 1) return rc3;
    WtGrid g;
    if (!wt_grid(a, g)) return 1;
    if (a.partial != nullptr && a.partial_stride < (int64_t)a.K * a.R * a.S * a.C + a.K)
        return fpd_fail(-2, "wgrad: partial_stride %lld smaller than weight + bias", (long long)a.partial_stride);
    if (a.partial == nullptr) g.gx = 1;      // no slabs: one block per (k, c) tile adds straight into dw (deterministic, slow)
    if (g.h4) return launch_wt<bf16_t, 3, 128, true>(a, g, st);
    if (a.R == 3 && a.dtype == FPD_BF16 && a.C <= 32 && a.K <= 32) return launch_wt<bf16_t, 3, 128, false, true>(a, g, st);
    if (a.R == 3) return a.dtype == FPD_BF16 ? launch_wt<bf16_t, 3, 128>(a, g, st) : launch_wt<float, 3, 128>(a, g, st);
    if (g.tp == 256) return a.dtype == FPD_BF16 ? launch_wt<bf16_t, 1, 256>(a, g, st) : launch_wt<float, 1, 256>(a, g, st);
    return a.dtype == FPD_BF16 ? launch_wt<bf16_t, 1, 128>(a, g, st) : launch_wt<float, 1, 128>(a, g, st);
}

int fpd_wgrad_tile_partials(const fpd_wgrad_t& a) {
    if (const int n3 = fpd_wgrad3_partials(a)) return n3;
    WtGrid g;
    return wt_grid(a, g) ? g.slabs : 0;
}

int fpd_wreduce_launch(const fpd_wreduce_entry_t* table, int n, int64_t max_elems, hipStream_t st) {
    if (n <= 0) return 0;
    dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>((max_elems / 4 + 255) / 256, 64)), (unsigned)n);
    FPD_LAUNCH(wreduce_kernel, grid, dim3(256), 0, st, table);
    return 0;
}
