// conv_pp: PERSISTENT "ping-pong" convolution for the big feature maps of the student chain (bf16, stride-1 1x1 / 3x3
// "same" convolutions with C, K <= 128: every train-mode convolution and data gradient of the hourglass at 64x64 and
// 128x128, where a launch has >= 512 pixel tiles).  Same contract as conv_tile_kernel (fpd_conv_t): BatchNorm+ReLU
// prologue on the operand, bias / residual, batch statistics of the result or the BN-backward epilogue.
//
// Why a second kernel: conv_tile runs load -> stage -> MFMA -> epilogue serially inside every block and re-fetches the
// weight tiles per tap and block, so at 64x64 a convolution takes 24-34 us against a 7-17 us HBM bound (DESIGN.md
// section 5, r02 ablation).  Here
//   * a block is resident for the whole launch (<= 1 per CU, 8 wave64) and owns a contiguous range of pixel tiles;
//   * ALL weights of the convolution (<= 72 KB: 3x3 64->64) are staged into LDS ONCE per block, XOR-swizzled
//     (16-byte chunk c of row n sits at c ^ sw(n)) so that the unpadded rows are read without bank conflicts;
//   * the 8 waves form two groups of 4 that work on alternating tiles half a period apart ("ping-pong"): while one
//     group issues the MFMAs of its tile (phase X), the other one runs the epilogue of its previous tile and stages its
//     next one (phase Y) -- matrix pipe beside VALU / LDS stores / global stores on every SIMD; the next tile's operand
//     rows and the epilogue's residual / epi_x vectors are requested at the start of X and consumed in Y (a full phase
//     of latency cover, loads survive the barriers);
//   * the MFMAs run transposed (first operand = weights): a lane owns 4 consecutive channels of one pixel, the
//     accumulators go through a wave-private fp32 LDS staging area and leave as 16-byte vectors in full 128-byte lines;
//   * statistics are accumulated per thread over all tiles of the block and flushed once per block;
//   * a block computes a slab of <= 64 output channels (K = 128: two slabs = two blocks per tile range, placed on the same
//     XCD so that the second one finds the operand rows in that L2): 32 accumulator registers per lane, and everything a
//     thread keeps across the phases (operand prefetch 32, residual / epi_x prefetch 16 + 16, statistics 24) fits the
//     256-register budget of two waves per SIMD without spills.
// Every phase is split in two halves by a block-wide barrier (X1 | X2 beside Y1 | Y2): Y1's staging area aliases the
// group's operand image, which Y2 then overwrites with the next tile.
//
// Replaces the same reference calls as conv_tile (nn.Conv2d + BatchNorm2d + ReLU, /root/reference/lib/models/hourglass.py:18-52).
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int pp_ilog2(int v) { return v <= 1 ? 0 : 1 + pp_ilog2(v / 2); }

struct PPGeo {
    int nrows;                // image rows per tile (tile = nrows * W <= 128 pixels)
    int ntiles;               // pixel tiles of this convolution
    int nblk;                 // persistent blocks working on it
    int region;               // bytes of one group's LDS region (operand image / epilogue staging), multiple of 16
    unsigned mW, mWV, mH;     // multiply-high reciprocals of W, W * (C / 8), H
};
__device__ __forceinline__ int pp_qdiv(int v, unsigned magic) { return (int)__umulhi((unsigned)v, magic); }
static inline unsigned pp_magic(int d) { return (unsigned)((0x100000000ull / (unsigned long long)d) + 1ull); }

// launder a value: everything derived from it is recomputed where it is used instead of being hoisted out of the phase loop
// and kept in registers for the whole kernel (the register budget is what bounds this kernel, not a few VALU operations)
__device__ __forceinline__ int pp_fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

#ifdef FPD_PP_TIMING      // probe build only (tools/probes): cycle stamps of two blocks at the phase boundaries, printed by the kernel
#define PP_STAMP() do { if ((tid & 255) == 0 && s_ns < 60) s_stamp[grp * 60 + s_ns++] = clock64(); } while (0)
#else
#define PP_STAMP() do { } while (0)
#endif

template <int R, int C, int TN, bool BWD>
__device__ __forceinline__ void conv_pp_body(const fpd_conv_t& a, const PPGeo geo, const int bi, const int n0) {
    constexpr int RS = R * R, KP = 32 * TN, CPR = C / 8, LOG_CPR = pp_ilog2(CPR), KS = C / 16;
    constexpr int LDA = C * 2 + 16;                       // operand-image pixel pitch in BYTES (16 B pad: conflict-free b128 reads)
    constexpr int CVN = KP / 8, RPI = 64 / CVN, NIT = 32 / RPI;     // 8-channel chunks, rows per read-back step, steps
    constexpr int LDST = KP * 4 + 16;                     // staging pitch in bytes (fp32)
    constexpr int NVH = 8;                                // operand vectors per thread and tile (host guarantees the fit)
    constexpr int pad = (R - 1) / 2;
    constexpr int RPB = CPR >= 16 ? 1 : 16 / CPR;         // weight rows per 256-byte bank row
    constexpr int NWV = (KP * RS * CPR + 511) / 512;      // weight vectors per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);        // 0 / 1: wave-uniform
    const int tg = tid & 255, wq = (tid >> 6) & 3;
    const int H = a.H, W = a.W, K = a.K;
    const int M = a.N * H * W, GR = a.N * H;
    const int nrows = geo.nrows, hrows = nrows + R - 1, WP = W + R - 1;
    const int TPX = nrows * W;
    const int zero_px = hrows * WP;
#ifdef FPD_PP_TIMING
    long long* s_stamp = reinterpret_cast<long long*>(smem + (2 * C + 5 * KP) * 4 + RS * KP * C * 2 + 2 * geo.region);
    int s_ns = 0;
    if (tid < 120) s_stamp[tid] = 0;
    __syncthreads();
#endif
    PP_STAMP();

    float* s_scale = reinterpret_cast<float*>(smem);      // [C]
    float* s_shift = s_scale + C;                         // [C]
    float* s_epi = s_shift + C;                           // [4][KP] (BNRELU_BWD)
    float* s_bias = s_epi + 4 * KP;                       // [KP]
    unsigned char* sW = reinterpret_cast<unsigned char*>(s_bias + KP);
    unsigned char* sG = sW + RS * KP * C * 2 + grp * geo.region;     // this group's region
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(a.y);
    const bf16_t* res = reinterpret_cast<const bf16_t*>(a.residual);
    const bf16_t* ex = reinterpret_cast<const bf16_t*>(a.epi_x);
    const bool want_stats = BWD || (a.out_stats != nullptr);

    // ---- this block's tiles: a contiguous range; group g takes every other one ----
    const int t_beg = (int)((long long)bi * geo.ntiles / geo.nblk);
    const int t_end = (int)((long long)(bi + 1) * geo.ntiles / geo.nblk);
    const int nt = t_end - t_beg;
    const int n_g = (nt + 1 - grp) >> 1;
    const int nint = max(2 * ((nt + 1) >> 1), (nt >> 1) > 0 ? 2 * (nt >> 1) + 1 : 0);

    // ---- operand staging (one group = 256 threads; a thread always stages the same 8 channels: 256 % CPR == 0) ----
    const int WV = W * CPR, nvtot = hrows * WV;
    uint4 rh[NVH];
    unsigned hmask = 0;
    auto halo_load = [&](int tile) {
        const int g0 = tile * nrows;
        const int tgl = pp_fresh(tg);
        const int cve = (tgl & (CPR - 1)) * 8;
        hmask = 0;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int v = tgl + i * 256;
            rh[i] = make_uint4(0, 0, 0, 0);
            if (v < nvtot) {
                const int hr = pp_qdiv(v, geo.mWV);
                const int j = (v - hr * WV) >> LOG_CPR;
                const int g = g0 - pad + hr;
                if ((unsigned)g < (unsigned)GR) {
                    rh[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(g * W + j) * C + cve));
                    hmask |= 1u << i;
                }
            }
        }
    };
    const float relu_lo = a.bn.relu ? 0.f : -3.4e38f;
    auto halo_store = [&]() {
        const int tgl = pp_fresh(tg);
        const int cvb = (tgl & (CPR - 1)) * 16;
        // the zero border columns and zero pixels first (the epilogue staging of the previous tile overwrote them)
        {
            const uint4 z = make_uint4(0, 0, 0, 0);
            const int nb = (R == 3) ? 2 * hrows : 0;
            for (int v = tgl; v < (nb + 3) * CPR; v += 256) {
                const int pz = v >> LOG_CPR, cv = (v & (CPR - 1)) * 16;
                const int px = pz < nb ? ((pz >> 1) * WP + ((pz & 1) ? WP - 1 : 0)) : zero_px + (pz - nb);
                *reinterpret_cast<uint4*>(sG + px * LDA + cv) = z;
            }
        }
        f32x4 sc0, sc1, sh0, sh1;
        if (a.bn.mode != FPD_BN_NONE) {
            sc0 = *reinterpret_cast<const f32x4*>(s_scale + (cvb >> 1));
            sc1 = *reinterpret_cast<const f32x4*>(s_scale + (cvb >> 1) + 4);
            sh0 = *reinterpret_cast<const f32x4*>(s_shift + (cvb >> 1));
            sh1 = *reinterpret_cast<const f32x4*>(s_shift + (cvb >> 1) + 4);
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int v = tgl + i * 256;
            if (v < nvtot) {
                const int hr = pp_qdiv(v, geo.mWV);
                const int j = (v - hr * WV) >> LOG_CPR;
                uint4 val = rh[i];
                if (a.bn.mode != FPD_BN_NONE) {
                    float f[8];
                    DT<bf16_t>::unpack(val, f);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        f[e] = fmaxf(fmaf(f[e], sc0[e], sh0[e]), relu_lo);
                        f[4 + e] = fmaxf(fmaf(f[4 + e], sc1[e], sh1[e]), relu_lo);
                    }
                    val = DT<bf16_t>::pack(f);
                    if (!((hmask >> i) & 1u)) val = make_uint4(0, 0, 0, 0);      // rows outside the tensor stay exactly zero
                }
                *reinterpret_cast<uint4*>(sG + (hr * WP + j + pad) * LDA + cvb) = val;
            }
        }
    };

    // ---- prologue: first operand rows, then the weights, requested before anything else ----
    if (n_g > 0) halo_load(t_beg + grp);
    {
        uint4 rw[NWV];
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            const int k = v / (RS * CPR), rem = v - k * (RS * CPR);
            rw[i] = make_uint4(0, 0, 0, 0);
            if (v < KP * RS * CPR && n0 + k < K) rw[i] = *reinterpret_cast<const uint4*>(w + ((size_t)(n0 + k) * RS * C + rem * 8));
        }
        bn_fill(a.bn, C, (double)M, s_scale, s_shift);
        conv_epi_tables<KP>(a, n0, M, s_epi);
        for (int t = tid; t < KP; t += 512) s_bias[t] = (a.bias != nullptr && n0 + t < K) ? a.bias[n0 + t] : 0.f;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            if (v < KP * RS * CPR) {
                const int k = v / (RS * CPR), rem = v - k * (RS * CPR);
                const int tap = rem >> LOG_CPR, ch = rem & (CPR - 1);
                const int sw = (k / RPB) & (CPR - 1);
                *reinterpret_cast<uint4*>(sW + ((tap * KP + k) * CPR + (ch ^ sw)) * 16) = rw[i];
            }
        }
    }
    PP_STAMP();
    __syncthreads();                                      // tables + weights visible
    PP_STAMP();
    if (n_g > 0) halo_store();
    PP_STAMP();

    // ---- per-lane MFMA addressing ----
    const int ml = wq * 32 + l31;                         // pixel of the tile this lane feeds (second MFMA operand)
    const int ti = pp_qdiv(ml, geo.mW), tj = ml - ti * W;
    int ab[3];
    auto tile_addr = [&](int tile) {
        const int g = tile * nrows + ti;
        const int p = g - pp_qdiv(g, geo.mH) * H;
        const bool live = g < GR && ml < TPX;
        const int zp = zero_px * LDA;
        if (R == 3) {
            ab[0] = (live && p >= 1) ? ((ti + 0) * WP + tj) * LDA : zp;
            ab[1] = live ? ((ti + 1) * WP + tj) * LDA : zp;
            ab[2] = (live && p + 1 < H) ? ((ti + 2) * WP + tj) * LDA : zp;
        } else {
            ab[0] = ab[1] = ab[2] = live ? (ti * WP + tj) * LDA : zp;
        }
    };

    f32x16 acc[TN];
    // steps [lo, hi) of the tap x k-step sequence of one tile.  Weight fragment of (tap, tn, kk): row tn*32 + l31 of tile `tap`,
    // 16-byte chunk (2 kk + hh) ^ sw(row); sw(row) is the same for rows l31 + 32 tn.
    auto mma_steps = [&](const int lo, const int hi) {
        const int l31f = pp_fresh(l31);
        const unsigned char* wrow = sW + l31f * (C * 2);
        const int wsw = (l31f / RPB) & (CPR - 1);
#pragma unroll
        for (int st = 0; st < RS * KS; ++st) {
            if (st < lo || st >= hi) continue;
            const int tap = st / KS, kk = st - tap * KS;
            const int r = tap / R, s = tap - r * R;
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sG + ab[r] + s * LDA + kk * 32 + hh * 16);
            const int wo = ((2 * kk + hh) ^ wsw) * 16;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + (tap * KP + tn * 32) * (C * 2) + wo);
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[tn], 0, 0, 0);
            }
        }
    };
    constexpr int NSTEP = RS * KS, HALF = (NSTEP + 1) / 2;

    // ---- epilogue state: a lane reads back the 8-channel chunk cvl of pixel rows row0, row0 + RPI, ... of its wave's 32 pixels ----
    uint4 rres[NIT], rex[BWD ? NIT : 1];
    float f1[8], f2[8], cshift[8];
    int nrow = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { f1[e] = 0.f; f2[e] = 0.f; cshift[e] = 0.f; }
    const float relu_gate = a.epi_bn.relu ? 0.f : -3.4e38f;
    auto request = [&](int tile) {                        // residual / epi_x vectors of the tile whose MFMAs start now
        const int m0 = tile * TPX;
        const int ln = pp_fresh(lane);
        const int cvl = ln % CVN, row0 = ln / CVN;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int px = wq * 32 + row0 + RPI * it;
            const int m = m0 + px, k0 = n0 + cvl * 8;
            rres[it] = make_uint4(0, 0, 0, 0);
            if (BWD) rex[it] = make_uint4(0, 0, 0, 0);
            if (px < TPX && m < M && k0 < K) {
                const size_t off = (size_t)m * K + k0;
                if (res != nullptr) rres[it] = *reinterpret_cast<const uint4*>(res + off);
                if (BWD) rex[it] = *reinterpret_cast<const uint4*>(ex + off);
            }
        }
    };
    auto epilogue = [&](int tile) {
        const int m0 = tile * TPX;
        const int ln = pp_fresh(lane);
        const int cvl = ln % CVN, row0 = ln / CVN;
        unsigned char* stg = sG + wq * (32 * LDST);       // wave-private [32 px][LDST]
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[tn][4 * j + e];
                *reinterpret_cast<f32x4*>(stg + l31 * LDST + (tn * 32 + 8 * j + 4 * hh) * 4) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's staging stores are in the LDS
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_bias + cvl * 8);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_bias + cvl * 8 + 4);
        float esc[8], esh[8], emu[8], eis[8];
        if (BWD) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(s_epi + cvl * 8 + 4 * q);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(s_epi + KP + cvl * 8 + 4 * q);
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(s_epi + 2 * KP + cvl * 8 + 4 * q);
                const f32x4 t3 = *reinterpret_cast<const f32x4*>(s_epi + 3 * KP + cvl * 8 + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) { esc[4 * q + e] = t0[e]; esh[4 * q + e] = t1[e]; emu[4 * q + e] = t2[e]; eis[4 * q + e] = t3[e]; }
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = row0 + RPI * it;
            const int px = wq * 32 + row;
            const int m = m0 + px, k0 = n0 + cvl * 8;
            if (px < TPX && m < M && k0 < K) {
                float v[8];
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(stg + row * LDST + cvl * 32);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(stg + row * LDST + cvl * 32 + 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = t0[e]; v[4 + e] = t1[e]; }
                const size_t off = (size_t)m * K + k0;
                if (res != nullptr) {
                    float r8[8];
                    DT<bf16_t>::unpack(rres[it], r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += r8[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                if (BWD) {
                    float xv[8], vr[8];
                    DT<bf16_t>::unpack(rex[it], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float z = fmaf(xv[e], esc[e], esh[e]);
                        v[e] = (z > relu_gate) ? v[e] : 0.f;
                    }
                    const uint4 pk = DT<bf16_t>::pack(v);
                    DT<bf16_t>::unpack(pk, vr);                      // the stored (rounded) gradient is what gets summed
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        f1[e] += vr[e];
                        f2[e] = fmaf(vr[e], (xv[e] - emu[e]) * eis[e], f2[e]);
                    }
                    *reinterpret_cast<uint4*>(y + off) = pk;
                } else {
                    const uint4 pk = DT<bf16_t>::pack(v);
                    if (want_stats) {
                        float vr[8];
                        DT<bf16_t>::unpack(pk, vr);
                        if (nrow == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) cshift[e] = vr[e];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float d = vr[e] - cshift[e];
                            f1[e] += d;
                            f2[e] = fmaf(d, d, f2[e]);
                        }
                        ++nrow;
                    }
                    *reinterpret_cast<uint4*>(y + off) = pk;
                }
            }
        }
    };

    // =========================== the ping-pong loop ===========================
    // phase k: group g runs X (MFMAs of its tile i = (k - g) / 2) when k + g is even, Y (epilogue of tile i = (k - g - 1) / 2,
    // staging of tile i + 1) when it is odd.  Every wave passes exactly two barriers per phase.
    __syncthreads();                                      // first operand images visible
    PP_STAMP();
    for (int k = 0; k < nint; ++k) {
        const bool xrole = ((k + grp) & 1) == 0;
        const int i = xrole ? (k - grp) >> 1 : (k - grp - 1) >> 1;
        const bool on = i >= 0 && i < n_g && (k - grp) >= 0;
        const int tile = t_beg + grp + 2 * i;
        if (xrole) {
            if (on) {
                request(tile);
                if (i + 1 < n_g) halo_load(tile + 2);     // in flight during both halves of X
                tile_addr(tile);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;
                mma_steps(0, HALF);
            }
        } else if (on) {
            epilogue(tile);
        }
        PP_STAMP();
        __syncthreads();
        PP_STAMP();
        if (xrole) {
            if (on) mma_steps(HALF, NSTEP);
        } else if (on && i + 1 < n_g) {
            halo_store();
        }
        PP_STAMP();
        __syncthreads();
        PP_STAMP();
    }

    // ---- statistics: one flush per block ----
    if (want_stats) {
        double* s_red = reinterpret_cast<double*>(sW + RS * KP * C * 2);      // [8 waves][KP][2] (both regions are free now)
        const int wave = tid >> 6;
        const int cvl = lane % CVN;
        float cs[8];
        float nr = (float)nrow;
        // shifted fp32 sums are re-based to a shift that is common to the lanes of a chunk (the first lane's), combined in
        // fp32 across those lanes, and un-shifted once, in fp64 (same scheme as conv_epilogue_vec)
        const float nfirst = __shfl(nr, cvl, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] = (BWD || nfirst == 0.f) ? 0.f : __shfl(cshift[e], cvl, 64);
            if (!BWD) {
                const float d = cshift[e] - cs[e];
                f2[e] = f2[e] + 2.f * d * f1[e] + nr * d * d;
                f1[e] = f1[e] + nr * d;
            }
        }
#pragma unroll
        for (int o = CVN; o < 64; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f1[e] += __shfl_xor(f1[e], o, 64);
                f2[e] += __shfl_xor(f2[e], o, 64);
            }
            nr += __shfl_xor(nr, o, 64);
        }
        if (lane < CVN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const double c = (double)cs[e], n = (double)nr;
                const double s1 = (double)f1[e] + n * c;
                const double s2 = BWD ? (double)f2[e] : (double)f2[e] + 2.0 * c * (double)f1[e] + n * c * c;
                s_red[(wave * KP + cvl * 8 + e) * 2 + 0] = s1;
                s_red[(wave * KP + cvl * 8 + e) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        double* st = (BWD ? a.epi_stats : a.out_stats) + (size_t)stats_replica() * 2 * K;
        for (int t = tid; t < KP; t += 512) {
            if (n0 + t < K) {
                double u1 = 0.0, u2 = 0.0;
#pragma unroll
                for (int wv = 0; wv < 8; ++wv) { u1 += s_red[(wv * KP + t) * 2]; u2 += s_red[(wv * KP + t) * 2 + 1]; }
                atomicAdd(st + n0 + t, u1);
                atomicAdd(st + K + n0 + t, u2);
            }
        }
    }
#ifdef FPD_PP_TIMING
    PP_STAMP();
    __syncthreads();
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) {
        // entry | tables+weights issued/stored | barrier | first image stored | barrier | per phase: half 1, barrier, half 2, barrier | flush
        printf("conv_pp R=%d C=%d K=%d blk %d nt=%d nint=%d  g0:", R, C, K, (int)blockIdx.x, nt, nint);
        for (int q = 1; q < 60 && s_stamp[q] != 0; ++q) printf(" %lld", s_stamp[q] - s_stamp[0]);
        printf("\n   g1:");
        for (int q = 0; q < 60 && s_stamp[60 + q] != 0; ++q) printf(" %lld", s_stamp[60 + q] - s_stamp[0]);
        printf("\n");
    }
#endif
}

// One or two INDEPENDENT convolutions of the same template configuration (the up-/low-branch Bottleneck convolutions of an
// hourglass level) in one launch.  A convolution owns nblk tile ranges x ks channel slabs = nblk * ks blocks ("units"); the
// units of `b` are spread evenly over the grid (Bresenham), so that whatever subset of the grid is resident first serves both
// in proportion.  gb.nblk == 0: single convolution.  Unit u -> (range, slab): the slabs of one range are 8 block ids apart,
// i.e. on the same XCD (its L2 serves the operand rows to the second slab).
template <int R, int C, int TN, bool BWD>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const fpd_conv_t a, const fpd_conv_t b, const PPGeo ga, const PPGeo gb,
                                                         const int ks) {
    const int bid = blockIdx.x, n = gridDim.x, nb = gb.nblk * ks;
    const int fb0 = (int)((long long)bid * nb / n), fb1 = (int)((long long)(bid + 1) * nb / n);
    const bool isb = fb1 > fb0;
    const int u = isb ? fb0 : bid - fb0;
    const int nr = isb ? gb.nblk : ga.nblk;
    int range, slab;
    if (ks == 2 && (nr & 7) == 0) { slab = (u >> 3) & 1; range = (u & 7) + 8 * (u >> 4); }
    else { slab = u % ks; range = u / ks; }
    if (isb) conv_pp_body<R, C, TN, BWD>(b, gb, range, slab * 64);
    else conv_pp_body<R, C, TN, BWD>(a, ga, range, slab * 64);
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr size_t PP_LDS_MAX = 160 * 1024;

// FPD_CONV_PP: 0 = never, 1 = launches with >= 512 tiles (default), 2 = whenever the shape is in the domain;
// FPD_CONV_PP_BLOCKS: persistent grid (default: one block per compute unit).  Both can be changed at run time through
// fpd_set_option("conv_pp" / "conv_pp_blocks", v) (tests drive small shapes through the kernel that way).
static int g_pp_mode = -1, g_pp_blocks = -1;
static int pp_mode() {
    if (g_pp_mode < 0) { const char* e = getenv("FPD_CONV_PP"); g_pp_mode = e ? atoi(e) : 0; }
    return g_pp_mode;
}
static int pp_blocks() {
    if (g_pp_blocks < 0) { const char* e = getenv("FPD_CONV_PP_BLOCKS"); g_pp_blocks = e ? atoi(e) : 256; }
    return g_pp_blocks < 2 ? 2 : g_pp_blocks;
}

static bool pp_domain(const fpd_conv_t& a) {
    if (a.dtype != FPD_BF16 || a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return false;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < 2) return false;
    if (a.C != 16 && a.C != 32 && a.C != 64 && a.C != 128) return false;
    if (a.K > 128 || a.K % 8 != 0) return false;
    if (a.R == 3 && a.C > 64) return false;                                // all nine weight tiles of a slab must fit the LDS
    const int nrows = std::max(1, 128 / a.W);
    if ((nrows + a.R - 1) * a.W * (a.C / 8) > 2048) return false;          // 8 operand vectors per thread
    return true;
}
static int pp_tiles(const fpd_conv_t& a) { return cdiv(a.N * a.H, std::max(1, 128 / a.W)); }

template <int C, int TN>
static PPGeo pp_geo(const fpd_conv_t& a) {
    constexpr int LDST = 32 * TN * 4 + 16, LDA = C * 2 + 16;
    PPGeo g;
    g.nrows = std::max(1, 128 / a.W);
    g.ntiles = pp_tiles(a);
    g.nblk = 0;
    const int hrows = g.nrows + a.R - 1, WP = a.W + a.R - 1;
    g.region = std::max((hrows * WP + 3) * LDA, 4 * 32 * LDST);
    g.mW = pp_magic(a.W);
    g.mWV = pp_magic(a.W * (C / 8));
    g.mH = pp_magic(a.H);
    return g;
}

template <int R, int C, int TN, bool BWD>
static int pp_launch_t(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    PPGeo ga = pp_geo<C, TN>(a), gb = ga;
    fpd_conv_t bb = a;
    gb.nblk = 0;
    int total = ga.ntiles;
    if (b != nullptr) {
        gb = pp_geo<C, TN>(*b);
        bb = *b;
        total += gb.ntiles;
    }
    const int region = std::max(ga.region, b ? gb.region : 0);
    ga.region = gb.region = region;
    size_t lds = (size_t)(2 * C + 5 * 32 * TN) * sizeof(float) + (size_t)R * R * 32 * TN * C * 2 + 2 * (size_t)region;
#ifdef FPD_PP_TIMING
    lds += 1024;
#endif
    if (lds > PP_LDS_MAX) return 1;
    const int ks = cdiv(a.K, 64);                          // channel slabs of <= 64 (K = 128: two blocks per tile range)
    // tile ranges: every block should own at least two tiles (one per group); ks blocks per range
    int ranges = std::max(1, std::min(pp_blocks() / ks, total / 2));
    if (b != nullptr) {
        if (ranges < 2) return 1;
        gb.nblk = std::max(1, std::min(ranges - 1, (int)((long long)ranges * gb.ntiles / total)));
    }
    ga.nblk = ranges - gb.nblk;
    if (ks == 2 && ga.nblk >= 8) ga.nblk &= ~7;            // (the XCD pairing of the slabs wants multiples of 8)
    if (ks == 2 && gb.nblk >= 8) gb.nblk &= ~7;
    const int grid = (ga.nblk + gb.nblk) * ks;
    static size_t configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pp_kernel<R, C, TN, BWD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fpd_fail(-100 - (int)e, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        configured = lds;
    }
    hipLaunchKernelGGL((conv_pp_kernel<R, C, TN, BWD>), dim3(grid), dim3(512), lds, st, a, bb, ga, gb, ks);
    return 0;
}

template <int R, int C>
static int pp_launch_c(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD;
    if (a.K > 32) return bwd ? pp_launch_t<R, C, 2, true>(a, b, st) : pp_launch_t<R, C, 2, false>(a, b, st);
    return bwd ? pp_launch_t<R, C, 1, true>(a, b, st) : pp_launch_t<R, C, 1, false>(a, b, st);
}

static int pp_launch(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    if (a.R == 3) {
        if (a.C == 64) return pp_launch_c<3, 64>(a, b, st);
        if (a.C == 32) return pp_launch_c<3, 32>(a, b, st);
        return pp_launch_c<3, 16>(a, b, st);
    }
    switch (a.C) {
        case 16: return pp_launch_c<1, 16>(a, b, st);
        case 32: return pp_launch_c<1, 32>(a, b, st);
        case 64: return pp_launch_c<1, 64>(a, b, st);
        default: return pp_launch_c<1, 128>(a, b, st);
    }
}

}  // namespace

int fpd_conv_pp_option(int which, int value) {     // which: 0 = mode, 1 = blocks; returns the previous value
    int& g = which == 0 ? g_pp_mode : g_pp_blocks;
    const int prev = which == 0 ? pp_mode() : pp_blocks();
    g = value;
    return prev;
}

// 0 = launched, 1 = outside this kernel's domain (the caller tries conv_tile next), < 0 error
int fpd_conv_pp_launch(const fpd_conv_t& a, hipStream_t st) {
    const int mode = pp_mode();
    if (mode == 0 || !pp_domain(a)) return 1;
    if (mode == 1 && pp_tiles(a) < 512) return 1;
    return pp_launch(a, nullptr, st);
}

// two independent convolutions of equal channel shapes in one launch; 1 = not pairable here
int fpd_conv_pp_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    const int mode = pp_mode();
    if (mode == 0 || !pp_domain(a) || !pp_domain(b)) return 1;
    if (a.K != b.K || a.C != b.C || a.R != b.R || a.epi != b.epi) return 1;
    if (mode == 1 && pp_tiles(a) + pp_tiles(b) < 512) return 1;
    return pp_launch(a, &b, st);
}
