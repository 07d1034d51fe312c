// conv_c1: STREAMING 1x1 convolution for the big maps of the student chain (bf16, C, K in {32, 64, 128}, N*H*W a multiple of
// 32): forward (BatchNorm+ReLU prologue, bias / residual, batch statistics of the result) and the BatchNorm-backward data
// gradient (ReLU mask of epi_x + the two sums), optionally evaluating a folded BN-backward apply on its operand (fold_x) and
// forming the forward convolution's weight / bias gradient in the same launch (wg_partial).  Same contract as conv_pp /
// conv_tile (fpd_conv_t); round 6.
//
// Why a third kernel.  Round 5 found conv_pp's tile loops bound by instruction ISSUE: 400-1 200 vector instructions per wave
// and 128-pixel tile around 8-16 MFMAs, three block barriers per tile (nine serial phases in the data gradients, one block
// per CU), every element passing through the LDS twice as fp32.  A 1x1 convolution has no halo, so nothing has to be shared
// between waves at all:
//   * a WAVE owns a 32-pixel tile outright -- all K output channels of it.  It brings the tile as whole 1 KB lines (lane l
//     always holds the same 8 channels, so BN scale / shift, the coefficients of a folded apply, the epilogue tables and the
//     statistics accumulators are registers, loaded once per block), applies the prologue, writes the operand into its
//     PRIVATE LDS tile and reads its MFMA fragments from there; the weights sit in the LDS once per block (XOR-swizzled);
//   * the transposed MFMA (weights first) leaves a lane with 4 consecutive channels of one pixel: + residual + bias in fp32,
//     ONE rounding, an 8-byte LDS store into the wave's bf16 output tile -- which is read back as whole 16-byte vectors of 8
//     channels for everything that is per channel (statistics; ReLU mask of the data gradient, which commutes with the
//     rounding; the forward operand a(u) of the fused weight gradient) and leaves as 1 KB global stores;
//   * no barrier in the tile loop (a wave's LDS operations execute in order): eight waves of a block drift apart and fill
//     each other's latencies.  Only the fused weight gradient exchanges data: after a ROUND (8 waves x 32 pixels) a wave
//     multiplies ITS 32x32 tile of dW over the 256 pixels of all eight tiles (dy and a(u) through transposing LDS reads:
//     two barriers per round), so no cross-wave reduction is ever needed and a block writes one slab at the end.
// Instructions per wave and 256-pixel round (isa_report): see DESIGN.md section 5.
//
// Replaces the same reference calls as conv_pp (nn.Conv2d 1x1 + BatchNorm2d + ReLU and their autograd,
// /root/reference/lib/models/hourglass.py:18-52, 134-137).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma_frag.h"

namespace {

constexpr int C1_NW = 8;                                 // waves per block = tiles per round

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 c1_unpack(unsigned w) {
    f32x2 r = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    return r;
}
__device__ __forceinline__ unsigned c1_pack(f32x2 v) { return f2bf_pk(v[0], v[1]); }
// max of the two signed 16-bit halves with `lo`: lo = 0 is ReLU on a packed bf16 pair, lo = -32768 the identity (conv_pp.hip)
__device__ __forceinline__ unsigned c1_floor(unsigned w, short lo) {
    s16x2 a = *reinterpret_cast<const s16x2*>(&w);
    const s16x2 b = {lo, lo};
    a = __builtin_elementwise_max(a, b);
    return *reinterpret_cast<const unsigned*>(&a);
}

// C, K: channels in / out of THIS launch.  BWD: BNRELU_BWD epilogue (no bias, no residual, no prologue BN).  FOLD (BWD only):
// the operand is a folded BN-backward apply (fold_x).  WG (BWD only): also the weight / bias gradient of the forward
// convolution this launch is the data gradient of.
template <int C, int K, bool BWD, bool WG>
struct C1Geo {
    static constexpr int CV = C / 8, KV = K / 8;         // 16-byte chunks per operand / output pixel
    static constexpr int NV = C / 16, NK = K / 16;       // operand / output vectors per lane and tile (32 px * CV / 64)
    static constexpr int KT = K / 32, KS = C / 16;       // 32-channel output tiles, 16-channel reduction steps
    static constexpr int PXA = C * 2 + 16, PXO = K * 2 + 16;     // pixel pitch of the operand / output tile (bytes)
    static constexpr int TILE_A = 32 * PXA, TILE_O = 32 * PXO;
    // (only the fused weight gradient reads the operand tile after the products: without it the output tile takes its place)
    static constexpr int WAVE_LDS = WG ? TILE_A + TILE_O : (TILE_A > TILE_O ? TILE_A : TILE_O);
    static constexpr int FLUSH = C1_NW * 64 * 16 * 4 + C1_NW * K * 4 + 8 * 2 * K * 8;      // statistics records + common shifts + partial sums
    static constexpr int REGION = C1_NW * WAVE_LDS > FLUSH ? C1_NW * WAVE_LDS : FLUSH;
    static constexpr int TABLES = (2 * C + (BWD ? 3 * C + 4 * K : 0) + K) * 4;
    static constexpr int LDS = TABLES + K * C * 2 + REGION;
    static constexpr int NT = (C / 32) * (K / 32);       // 32x32 tiles of dW
    static constexpr int NTW = NT / C1_NW;               // ... per wave
    static_assert(!WG || (BWD && NT % C1_NW == 0 && NTW >= 1), "fused weight gradient: every wave owns whole tiles of dW");
    static_assert(64 % CV == 0 && 64 % KV == 0, "a lane keeps its channel chunk");
};

// the compiler may not move LDS accesses across the phases of a tile: the phases of ONE wave communicate through its private
// tiles without a barrier (a wave's LDS operations execute in order), with accesses of different vector types
// (and no arithmetic either: hipcc fills the shadow of the MFMAs with the NEXT phase's unpacking of every vector in flight,
//  which costs more registers than the kernel has; the other wave of the SIMD is what runs beside the matrix pipe)
#define C1_PHASE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

#ifdef FPD_C1_TIMING      // probe build only: cycle stamps of waves 0 and 7 of two blocks at the phase boundaries, printed by the kernel
#define C1_STAMP() do { if (lane == 0 && (wave == 0 || wave == 7) && c1_ns < 40) c1_stamp[(wave ? 40 : 0) + c1_ns++] = clock64(); } while (0)
#else
#define C1_STAMP() do { } while (0)
#endif

template <int C, int K, bool BWD, bool FOLD, bool WG, bool RES>
__device__ __forceinline__ void c1_body(const fpd_conv_t& a, const int bi, const int nblk) {
    using G = C1Geo<C, K, BWD, WG>;
    constexpr int CV = G::CV, KV = G::KV, NV = G::NV, NK = G::NK, KT = G::KT, KS = G::KS, PXA = G::PXA, PXO = G::PXO;
    constexpr int RPB = CV >= 16 ? 1 : 16 / CV;           // weight rows per 256-byte bank row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int M = a.N * a.H * a.W;
    const int ntile = M >> 5;                             // (M % 32 == 0: checked by the host)
    const int nround = (ntile + C1_NW - 1) / C1_NW;
    const int r_beg = fpd_cut(bi, nround, nblk), r_end = fpd_cut(bi + 1, nround, nblk);

    float* s_scale = reinterpret_cast<float*>(smem);      // [C] prologue BN
    float* s_shift = s_scale + C;                         // [C]
    float* s_fold = s_shift + C;                          // BWD: [3][C] coefficients of a folded BN-backward apply
    float* s_epi = s_fold + (BWD ? 3 * C : 0);            // BWD: [4][K] scale, shift, mean, invstd of epi_bn
    float* s_bias = s_epi + (BWD ? 4 * K : 0);            // [K]
    unsigned char* sW = reinterpret_cast<unsigned char*>(s_bias + K);     // [K][C] bf16, 16-byte chunk c of row n at c ^ sw(n)
    unsigned char* sT = sW + K * C * 2;                   // the waves' tiles
    unsigned char* tA = sT + wave * G::WAVE_LDS;          // operand tile [32 px][PXA]
    unsigned char* tO = WG ? tA + G::TILE_A : tA;         // output tile [32 px][PXO] (no fused weight gradient: the operand tile is dead by then)

    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    bf16_t* y = reinterpret_cast<bf16_t*>(a.y);           // (the residual may alias it)
    const bf16_t* res = reinterpret_cast<const bf16_t*>(a.residual);
    const bf16_t* ex = reinterpret_cast<const bf16_t*>(a.epi_x);
    // (a launch whose OTHER convolution folds: u is loaded from the operand itself and ignored -- no conditional loads)
    const bf16_t* __restrict__ fx = reinterpret_cast<const bf16_t*>(a.fold_x != nullptr ? a.fold_x : a.x);
    const bool fold = FOLD && a.fold_x != nullptr;        // (FOLD: some convolution of the launch folds; this one may not)
    bf16_t* fo = fold ? reinterpret_cast<bf16_t*>(a.fold_out) : nullptr;
    const bool has_bn = !BWD && a.bn.mode != FPD_BN_NONE;
    constexpr bool has_res = RES;                         // (forward only; a template parameter: no conditional loads)
    const bool want_stats = BWD || a.out_stats != nullptr;
    const bool wg = WG && a.wg_partial != nullptr;
    const bool wg_bias = wg && a.wg_bias;
#ifdef FPD_C1_TIMING
    long long* c1_stamp = reinterpret_cast<long long*>(smem + G::LDS);      // [2][40] behind everything (the probe build asks for 1 KB more)
    int c1_ns = 0;
#endif
    C1_STAMP();

    // ---- prologue: the table chains (loads -> fp64 -> LDS) on different waves, requested before the long loads ----
    BnRaw braw;
    float bias_raw;                                      // (set in the r_bias branch only: a default written here is sunk by hipcc
                                                          //  behind the other branches' loads, where it needs vmcnt(0) -- see bn_request)
    const int te = tid - 128, tb = tid - 256;
    const bool r_bn = has_bn && tid < C;
    const bool r_fold = fold && tid < C;
    const bool r_epi = BWD && te >= 0 && te < K;
    const bool r_bias = tb >= 0 && tb < K;
    StatRaw fs1, fs2;
    if (r_bn) bn_request(a.bn, tid, C, braw);
    else if (r_fold) {
        bn_request(a.fold_bn, tid, C, braw);
        stat_request(a.fold_stats, C, 0, tid, fs1);
        stat_request(a.fold_stats, C, 1, tid, fs2);
    }
    else if (r_epi) bn_request(a.epi_bn, te, K, braw);
    else if (r_bias) { bias_raw = 0.f; if (a.bias != nullptr) bias_raw = a.bias[tb]; }
    __builtin_amdgcn_sched_barrier(0);

    // ---- the tile's vectors: vector i of a lane is 16 bytes at (tile base) + (i * 64 + lane) * 16 -- whole 1 KB lines ----
    uint4 rx[NV];                                         // operand (fold: the masked gradient g)
    uint4 ru[FOLD ? NV : 1];                              // fold: the BN input u
    uint4 rr[NK];                                         // forward: residual; backward: epi_x
    auto load_x = [&](int t) {
        const uint4* px = reinterpret_cast<const uint4*>(x) + ((size_t)t * 32 * CV + lane);
#pragma unroll
        for (int i = 0; i < NV; ++i) rx[i] = px[i * 64];
        if constexpr (FOLD) {
            const uint4* pu = reinterpret_cast<const uint4*>(fx) + ((size_t)t * 32 * CV + lane);
#pragma unroll
            for (int i = 0; i < NV; ++i) ru[i] = pu[i * 64];
        }
    };
    auto load_r = [&](int t) {                            // (callers test has_res in forward mode)
        const uint4* pr = reinterpret_cast<const uint4*>(BWD ? ex : res) + ((size_t)t * 32 * KV + lane);
#pragma unroll
        for (int i = 0; i < NK; ++i) rr[i] = pr[i * 64];
    };
    int tile = r_beg * C1_NW + wave;
    {
        // requested in the order they are needed: tables (above), weights -- the first barrier waits for both --, then the
        // first tile (100+ KB per block: at the chip's ~12 bytes per cycle and CU it takes 6-11 k cycles to arrive; stamps)
        constexpr int NWV = (K * CV + 511) / 512;         // weight vectors per thread
        uint4 rw[NWV];
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            rw[i] = make_uint4(0, 0, 0, 0);
            if (v < K * CV) rw[i] = *reinterpret_cast<const uint4*>(w + (size_t)v * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            // (unconditionally, a wave without a tile re-reads the last one: behind a branch hipcc waits for EVERY load in
            //  flight -- vmcnt(0) -- before the weights go to the LDS, i.e. the first barrier waited for the whole first tile)
            const int t0 = min(tile, ntile - 1);
            load_x(t0);
            if constexpr (BWD || has_res) load_r(t0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (r_bn) {
            float sc, sh, mu, is;
            bn_resolve(braw, (double)M, sc, sh, mu, is);
            s_scale[tid] = sc;
            s_shift[tid] = sh;
        } else if (r_fold) {
            // dy = gamma*is*(g - m1 - xhat*m2), xhat = (u - mu)*is  ==  A g + B u + D   (coefficients formed in fp64, as conv_pp)
            const double s1 = stat_resolve(braw.s1), s2 = stat_resolve(braw.s2), b1 = stat_resolve(fs1), b2 = stat_resolve(fs2);
            const double cnt = (double)M, mu = s1 / cnt;
            double var = s2 / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)braw.eps), gi = (double)braw.g * is;
            const double m1 = b1 / cnt, m2 = b2 / cnt;
            s_fold[tid] = (float)gi;
            s_fold[C + tid] = (float)(-gi * is * m2);
            s_fold[2 * C + tid] = (float)(gi * (mu * is * m2 - m1));
            if (bi == 0) {                                // the affine parameters' gradients fall out of the two sums
                if (a.fold_dgamma != nullptr) a.fold_dgamma[tid] = (float)b2;
                if (a.fold_dbeta != nullptr) a.fold_dbeta[tid] = (float)b1;
            }
        } else if (r_epi) {
            float sc, sh, mu, is;
            bn_resolve(braw, (double)M, sc, sh, mu, is);
            s_epi[te] = sc; s_epi[K + te] = sh; s_epi[2 * K + te] = mu; s_epi[3 * K + te] = is;
        } else if (r_bias) {
            s_bias[tb] = bias_raw;
        }
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            if (v < K * CV) {
                const int k = v / CV, ch = v % CV;
                const int sw = (k / RPB) & (CV - 1);
                *reinterpret_cast<uint4*>(sW + (k * CV + (ch ^ sw)) * 16) = rw[i];
            }
        }
    }
    C1_STAMP();
    __syncthreads();                                      // tables + weights visible
    C1_STAMP();

    // ---- per-lane constants: the 8 channels of this lane's operand / output chunk ----
    const int cch = lane % CV, kch = lane % KV;           // chunk of the operand / output pixel this lane always holds
    const int pxa0 = lane / CV, pxo0 = lane / KV;         // its pixel in vector 0 (vector i: + i * 64 / CV)
    f32x2 p_sc[4], p_sh[4];                               // forward BN scale / shift, as pairs
    if (has_bn) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p_sc[e] = *reinterpret_cast<const f32x2*>(s_scale + cch * 8 + 2 * e);
            p_sh[e] = *reinterpret_cast<const f32x2*>(s_shift + cch * 8 + 2 * e);
        }
    }
    // (the data gradients re-read their per-channel tables from the LDS in the phase that uses them: the registers are
    //  needed for the accumulators and the vectors in flight)
    const short relu_floor = a.bn.relu ? (short)0 : (short)-32768;
    const short epi_floor = a.epi_bn.relu ? (short)0 : (short)-32768;
    const float relu_gate = a.epi_bn.relu ? 0.f : -3.4e38f;
    f32x2 F1[4], F2[4], CS[4];                            // statistics partials / common shift of this lane's 8 output channels
#pragma unroll
    for (int e = 0; e < 4; ++e) { F1[e] = f32x2{0.f, 0.f}; F2[e] = f32x2{0.f, 0.f}; CS[e] = f32x2{0.f, 0.f}; }
    const int wsw = (l31 / RPB) & (CV - 1);               // swizzle of this lane's weight rows (row = kt * 32 + l31)
    f32x16 wacc[WG ? G::NTW : 1], bacc[WG ? G::NTW : 1];  // WG: this wave's tiles of dW / of the bias gradient
#pragma unroll
    for (int j = 0; j < (WG ? G::NTW : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { wacc[j][e] = 0.f; bacc[j][e] = 0.f; }

    // ---- 1. operand: prologue on the way into the wave's tile ----
    auto stage = [&](const int t, auto modec) {
        constexpr int MODE = decltype(modec)::value;       // 0 = raw operand, 1 = BatchNorm(+ReLU) prologue, 2 = folded BN-backward apply
        f32x2 f_a[4], f_b[4], f_d[4];                     // fold coefficients of this lane's 8 channels
        if constexpr (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f_a[e] = *reinterpret_cast<const f32x2*>(s_fold + cch * 8 + 2 * e);
                f_b[e] = *reinterpret_cast<const f32x2*>(s_fold + C + cch * 8 + 2 * e);
                f_d[e] = *reinterpret_cast<const f32x2*>(s_fold + 2 * C + cch * 8 + 2 * e);
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            uint4 val = rx[i];
            if constexpr (MODE == 2) {
                const uint4 uu = ru[i];
                const unsigned gw[4] = {val.x, val.y, val.z, val.w}, uw[4] = {uu.x, uu.y, uu.z, uu.w};
                unsigned ow[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 tt = __builtin_elementwise_fma(f_b[e], c1_unpack(uw[e]), f_d[e]);
                    ow[e] = c1_pack(__builtin_elementwise_fma(f_a[e], c1_unpack(gw[e]), tt));
                }
                val = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            } else if constexpr (MODE == 1) {
                val.x = c1_floor(c1_pack(__builtin_elementwise_fma(c1_unpack(val.x), p_sc[0], p_sh[0])), relu_floor);
                val.y = c1_floor(c1_pack(__builtin_elementwise_fma(c1_unpack(val.y), p_sc[1], p_sh[1])), relu_floor);
                val.z = c1_floor(c1_pack(__builtin_elementwise_fma(c1_unpack(val.z), p_sc[2], p_sh[2])), relu_floor);
                val.w = c1_floor(c1_pack(__builtin_elementwise_fma(c1_unpack(val.w), p_sc[3], p_sh[3])), relu_floor);
            }
            rx[i] = val;
            *reinterpret_cast<uint4*>(tA + (pxa0 + i * (64 / CV)) * PXA + cch * 16) = val;
        }
        if (MODE == 2 && fo != nullptr) {                 // the evaluated operand has other readers: written out once
            uint4* po = reinterpret_cast<uint4*>(fo) + ((size_t)t * 32 * CV + lane);
#pragma unroll
            for (int i = 0; i < NV; ++i) po[i * 64] = rx[i];
        }
    };
    // ---- 3. + residual + bias in fp32, ONE rounding, into the wave's bf16 output tile (lane: pixel l31, channels
    //         kt * 32 + 8 q + 4 hh .. + 3) ----
    auto finish = [&](const f32x16& accv, const int kt, auto resc) {
        constexpr bool HASRES = decltype(resc)::value;
        {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned char* p = tO + l31 * PXO + (kt * 32 + 8 * q + 4 * hh) * 2;
                f32x2 v0 = {accv[4 * q], accv[4 * q + 1]}, v1 = {accv[4 * q + 2], accv[4 * q + 3]};
                if constexpr (!BWD) {
                    if constexpr (HASRES) {
                        const uint2 r2 = *reinterpret_cast<const uint2*>(p);
                        v0 += c1_unpack(r2.x);
                        v1 += c1_unpack(r2.y);
                    }
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + kt * 32 + 8 * q + 4 * hh);
                    v0 += f32x2{b4[0], b4[1]};
                    v1 += f32x2{b4[2], b4[3]};
                }
                *reinterpret_cast<uint2*>(p) = make_uint2(c1_pack(v0), c1_pack(v1));
            }
        }
    };
    // ---- 4. per channel: whole vectors of 8 channels again (vector i: pixel pxo0 + i * 64 / KV, chunk kch) ----
    auto perchan = [&](const int t, const bool first, auto statc) {
        constexpr bool STATS = decltype(statc)::value;
        uint4* py = reinterpret_cast<uint4*>(y) + ((size_t)t * 32 * KV + lane);
        f32x2 e_sc[4], e_sh[4], e_mu[4], e_is[4];         // BWD: epilogue BN of this lane's 8 output channels
        if constexpr (BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                e_sc[e] = *reinterpret_cast<const f32x2*>(s_epi + kch * 8 + 2 * e);
                e_sh[e] = *reinterpret_cast<const f32x2*>(s_epi + K + kch * 8 + 2 * e);
                e_mu[e] = *reinterpret_cast<const f32x2*>(s_epi + 2 * K + kch * 8 + 2 * e);
                e_is[e] = *reinterpret_cast<const f32x2*>(s_epi + 3 * K + kch * 8 + 2 * e);
            }
        }
        if (!BWD && STATS && first) {
            // common shift of the wave's shifted sums: the value of its first pixel (every lane of a chunk reads it)
            const uint4 c4 = *reinterpret_cast<const uint4*>(tO + kch * 16);
            CS[0] = c1_unpack(c4.x); CS[1] = c1_unpack(c4.y); CS[2] = c1_unpack(c4.z); CS[3] = c1_unpack(c4.w);
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            unsigned char* p = tO + (pxo0 + i * (64 / KV)) * PXO + kch * 16;
            const uint4 o4 = *reinterpret_cast<const uint4*>(p);
            unsigned ow[4] = {o4.x, o4.y, o4.z, o4.w};
            if constexpr (BWD) {
                const unsigned xw[4] = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
                unsigned aw[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 xv = c1_unpack(xw[e]);
                    const f32x2 z = __builtin_elementwise_fma(xv, e_sc[e], e_sh[e]);
                    // ReLU mask of the forward tensor (it commutes with the rounding), then the two BatchNorm-backward sums of
                    // the stored (rounded) gradient
                    const unsigned keep = (z[0] > relu_gate ? 0x0000ffffu : 0u) | (z[1] > relu_gate ? 0xffff0000u : 0u);
                    ow[e] &= keep;
                    const f32x2 g = c1_unpack(ow[e]);
                    F1[e] += g;
                    F2[e] = __builtin_elementwise_fma(g, (xv - e_mu[e]) * e_is[e], F2[e]);
                    if constexpr (WG) aw[e] = c1_floor(c1_pack(z), epi_floor);     // the forward operand as the forward convolution staged it
                }
                if constexpr (WG) *reinterpret_cast<uint4*>(p) = make_uint4(aw[0], aw[1], aw[2], aw[3]);
            } else if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 d = c1_unpack(ow[e]) - CS[e];
                    F1[e] += d;
                    F2[e] = __builtin_elementwise_fma(d, d, F2[e]);
                }
            }
            py[i * 64] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if constexpr (BWD || STATS) {
                // one vector at a time, its sums formed HERE: hipcc otherwise sinks the statistics arithmetic of all vectors
                // behind the loop and keeps every unpacked operand alive until then (more registers than the kernel has)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(F1[e]), "+v"(F2[e]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- 5. WG: this wave's tiles of dW over the 256 pixels of the round (dy^T and a(u) through transposing reads).  A full
    //         round is unrolled with the fragments of tile s + 1 requested before the products of tile s (stamps of the
    //         rolled loop: 5-6.7 k cycles per round for 16-32 MFMAs -- every pair of tiles waited for its LDS reads) ----
    auto wg_frags = [&](const int s, bf16x8 (&af)[2][WG ? G::NTW : 1], bf16x8 (&bfr)[2][WG ? G::NTW : 1]) {
        const bf16_t* dyT = reinterpret_cast<const bf16_t*>(sT + s * G::WAVE_LDS);
        const bf16_t* aT = reinterpret_cast<const bf16_t*>(sT + s * G::WAVE_LDS + G::TILE_A);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < (WG ? G::NTW : 1); ++j) {
                const int tw = wave + C1_NW * j, ci = tw / KT, ko = tw % KT;
                af[h][j] = tr_frag_bf16(dyT, PXA / 2, h * 16, ci * 32, lane);
                bfr[h][j] = tr_frag_bf16(aT, PXO / 2, h * 16, ko * 32, lane);
            }
        }
    };
    auto wg_round = [&](const int round, auto biasc) {
        constexpr bool BIAS = decltype(biasc)::value;      // this wave owns tiles with ko == 0: the bias gradient rides along
        const bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};
        bf16x8 af[2][2][WG ? G::NTW : 1], bfr[2][2][WG ? G::NTW : 1];      // [buffer][pixel half][tile]
        auto mult = [&](const int b) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < (WG ? G::NTW : 1); ++j) {
                    wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[b][h][j], bfr[b][h][j], wacc[j], 0, 0, 0);
                    if constexpr (BIAS) bacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[b][h][j], ones, bacc[j], 0, 0, 0);
                }
            }
        };
        if (round * C1_NW + C1_NW <= ntile) {
            wg_frags(0, af[0], bfr[0]);
#pragma unroll
            for (int s = 0; s < C1_NW; ++s) {
                if (s + 1 < C1_NW) wg_frags(s + 1, af[(s + 1) & 1], bfr[(s + 1) & 1]);
                mult(s & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll 1
            for (int s = 0; s < C1_NW; ++s) {
                if (round * C1_NW + s < ntile) {
                    wg_frags(s, af[0], bfr[0]);
                    mult(0);
                }
            }
        }
    };

    // =========================== the round loop ===========================
    bool first = true;
    for (int round = r_beg; round < r_end; ++round) {
        tile = round * C1_NW + wave;
        const bool live = tile < ntile;                   // (wave-uniform)
        const bool pf = round + 1 < r_end && tile + C1_NW < ntile;      // a next tile to prefetch
        if (live) {
            if (FOLD && fold) { if constexpr (FOLD) stage(tile, std::integral_constant<int, 2>{}); }
            else if (has_bn) { if constexpr (!BWD) stage(tile, std::integral_constant<int, 1>{}); }
            else stage(tile, std::integral_constant<int, 0>{});
            if (pf) load_x(tile + C1_NW);                 // the next tile's vectors take the registers over
            C1_PHASE();
            C1_STAMP();
            // ---- 2. the products: D[k][px] = sum_c W[k][c] * opnd[px][c]; a lane gets 4 consecutive channels of pixel l31 ----
            bf16x8 bf[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) bf[kk] = *reinterpret_cast<const bf16x8*>(tA + l31 * PXA + kk * 32 + hh * 16);
            if constexpr (has_res) {                       // the residual goes through the output tile (after the fragments are read)
                C1_PHASE();
#pragma unroll
                for (int i = 0; i < NK; ++i)
                    *reinterpret_cast<uint4*>(tO + (pxo0 + i * (64 / KV)) * PXO + kch * 16) = make_uint4(rr[i].x, rr[i].y, rr[i].z, rr[i].w);
                C1_PHASE();
            }
            // two 32-channel tiles of the output at a time: the accumulators of one pair are finished (step 3) while the next
            // pair multiplies
#pragma unroll
            for (int k0 = 0; k0 < KT; k0 += 2) {
                f32x16 acc0, acc1;
#pragma unroll
                for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
                const unsigned char* wrow0 = sW + (k0 * 32 + l31) * (C * 2);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const bf16x8 wf0 = *reinterpret_cast<const bf16x8*>(wrow0 + ((2 * kk + hh) ^ wsw) * 16);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf0, bf[kk], acc0, 0, 0, 0);
                    if (k0 + 1 < KT) {
                        const bf16x8 wf1 = *reinterpret_cast<const bf16x8*>(wrow0 + 32 * (C * 2) + ((2 * kk + hh) ^ wsw) * 16);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf1, bf[kk], acc1, 0, 0, 0);
                    }
                }
                finish(acc0, k0, std::integral_constant<bool, has_res>{});
                if (k0 + 1 < KT) finish(acc1, k0 + 1, std::integral_constant<bool, has_res>{});
            }
            if constexpr (!BWD && has_res) { if (pf) load_r(tile + C1_NW); }
            C1_PHASE();
            C1_STAMP();
            if (want_stats) perchan(tile, first, std::true_type{}); else perchan(tile, first, std::false_type{});
            if (BWD && pf) load_r(tile + C1_NW);
            C1_PHASE();
            C1_STAMP();
            first = false;
        }
        if (WG) {
            __syncthreads();                              // the eight tiles of the round: dy and a(u) complete
            if (wg) {
                // (whether this wave also multiplies for the bias gradient is decided once: a branch per MFMA otherwise)
                if (wg_bias && (wave % KT) == 0) wg_round(round, std::true_type{}); else wg_round(round, std::false_type{});
            }
            __syncthreads();                              // tiles free for the next round
            C1_STAMP();
        }
    }
    C1_STAMP();

    // ---- WG: this block's slab -- every wave stores the tiles of dW it owns (fpd_wgrad_reduce adds the slabs in order) ----
    if (WG && wg) {
        float* slab = a.wg_partial + (size_t)bi * a.wg_stride;
#pragma unroll
        for (int j = 0; j < (WG ? G::NTW : 1); ++j) {
            const int tw = wave + C1_NW * j, ci = tw / KT, ko = tw % KT;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = ci * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                slab[(size_t)row * K + ko * 32 + l31] = wacc[j][e];
                if (wg_bias && ko == 0 && l31 == 0) slab[(size_t)C * K + row] = bacc[j][e];
            }
        }
    }

    // ---- statistics: one flush per block.  A lane holds fp32 partial sums of 8 channels over its pixels (forward: shifted by
    //      the wave's common shift; backward {sum dz, sum dz * xhat}); fixed-order fp64 sums per channel, exact limbs out ----
    if (want_stats) {
        __syncthreads();                                  // every wave is done with its tiles
        float* rec = reinterpret_cast<float*>(sT);        // [8 waves][64 lanes][16]
        float* shf = rec + C1_NW * 64 * 16;               // [8 waves][K]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<f32x2*>(rec + (wave * 64 + lane) * 16 + 2 * e) = F1[e];
            *reinterpret_cast<f32x2*>(rec + (wave * 64 + lane) * 16 + 8 + 2 * e) = F2[e];
        }
        if (!BWD && lane < KV) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(shf + wave * K + lane * 8 + 2 * e) = CS[e];
        }
        __syncthreads();
        fpd_stat_t* st = BWD ? a.epi_stats : a.out_stats;
        // stage A (all threads): thread (part, which, ch) sums the records of NPW waves in a fixed order in fp64 and un-shifts
        // them; stage B: the parts in order, one exact pair of limbs per (sum, channel) out.  (One thread per sum over all
        // eight waves took 6.4 k cycles at the end of every block: stamps, round 6.)
        constexpr int NPART = 512 / (2 * K) > C1_NW ? C1_NW : 512 / (2 * K);      // 2 (K = 128), 4 (K = 64), 8 (K = 32)
        constexpr int NPW = C1_NW / NPART;
        double* s_part = reinterpret_cast<double*>(shf + C1_NW * K);             // [NPART][2 K]
        {
            const int part = tid / (2 * K), rem = tid % (2 * K);
            const int ch = rem % K, which = rem / K;      // which: 0 = first sum, 1 = second
            const int chunk = ch >> 3, e = ch & 7;
            if (part < NPART) {
                double tot = 0.0;
#pragma unroll
                for (int q = 0; q < NPW; ++q) {
                    const int wv = part * NPW + q;
                    double t1 = 0.0, t2 = 0.0;
#pragma unroll
                    for (int jj = 0; jj < 64 / KV; ++jj) {
                        const float* rp = rec + (wv * 64 + chunk + KV * jj) * 16;
                        t1 += (double)rp[e];
                        t2 += (double)rp[8 + e];
                    }
                    if (BWD) tot += which ? t2 : t1;
                    else {
                        // tiles wave wv has processed: rounds r of this block with r * 8 + wv < ntile
                        const int last = (ntile - 1 - wv) >= 0 ? (ntile - 1 - wv) / C1_NW : -1;
                        const int hi = min(r_end - 1, last);
                        const int nt = hi >= r_beg ? hi - r_beg + 1 : 0;
                        const double c = (double)shf[wv * K + ch], nn = 32.0 * nt;
                        tot += which ? (t2 + 2.0 * c * t1 + nn * c * c) : (t1 + nn * c);
                    }
                }
                s_part[part * 2 * K + rem] = tot;
            }
        }
        __syncthreads();
        if (tid < 2 * K) {
            double tot = 0.0;
#pragma unroll
            for (int q = 0; q < NPART; ++q) tot += s_part[q * 2 * K + tid];
            stat_atomic_add(st, K, tid / K, tid % K, tot);
        }
    }
#ifdef FPD_C1_TIMING
    C1_STAMP();
    if (lane == 0 && (wave == 0 || wave == 7)) c1_stamp[(wave ? 40 : 0) + 39] = c1_ns;
    __syncthreads();
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 1)) {
        // entry | tables + weights stored | barrier | per tile: staged (+ next requested), products + finish, per-channel + stores (WG: + round) | loop end | flush
        // (one printf per row: concurrent printfs interleave)
        for (int wv = 0; wv < 2; ++wv) {
            const long long* sp = c1_stamp + wv * 40;
            const long long t0 = c1_stamp[0];
            long long d[16];
            for (int q = 0; q < 16; ++q) d[q] = q < (int)sp[39] ? sp[q] - t0 : -1;
            printf("conv_c1 C=%d K=%d bwd %d fold %d wg %d blk %d wave %d rounds %d: %lld %lld %lld | %lld %lld %lld %lld | %lld %lld %lld %lld | %lld %lld %lld %lld %lld\n",
                   C, K, (int)BWD, (int)FOLD, (int)WG, (int)blockIdx.x, wv * 7, r_end - r_beg,
                   d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], d[12], d[13], d[14], d[15]);
        }
    }
#endif
}

// One or two INDEPENDENT convolutions of the same template configuration in one launch (the up- / low-branch Bottleneck
// convolutions of an hourglass level): blocks [0, nblk[0]) work on descriptor 0, the rest on descriptor 1, spread evenly
// over the grid (Bresenham) so that whatever part of the grid is resident first serves both in proportion.
struct C1Args { fpd_conv_t c[2]; int nblk[2]; };

template <int C, int K, bool BWD, bool FOLD, bool WG, bool RES>
__global__ __launch_bounds__(512, 2) void c1_kernel(const C1Args p) {
    const int bid = blockIdx.x, n = gridDim.x, nb = p.nblk[1];
    const int fb0 = fpd_cut(bid, nb, n), fb1 = fpd_cut(bid + 1, nb, n);
    const int isb = fb1 > fb0 ? 1 : 0;
    const int u = isb ? fb0 : bid - fb0;
    c1_body<C, K, BWD, FOLD, WG, RES>(p.c[isb], u, p.nblk[isb]);
}

// ---------------------------------------------------------------------------------------------------------------------
// FPD_C1: 0 = never, 1 = launches of >= FPD_C1_MIN_PX pixels (default), 2 = whenever the shape is in the domain (tests:
// fpd_set_option("conv_c1", v)); FPD_C1_BLOCKS: persistent blocks of a launch (default 192 of the 256 CUs, see c1_blocks()).
int g_c1_mode = -1, g_c1_blocks = -1;
std::atomic<int> g_c1_launches{0};                       // launches this kernel has served (tests: "conv_c1_launches")
int c1_mode() {
    if (g_c1_mode < 0) { const char* e = getenv("FPD_C1"); g_c1_mode = e ? atoi(e) : 1; }
    return g_c1_mode;
}
int c1_blocks() {
    // (192, not 256: a resident block owns most of its CU's LDS, and the frozen teacher's fused Bottlenecks -- 128 blocks of 151 KB on
    //  another stream -- need compute units of their own; r06 sweep inside the step, one box: 160 / 192 / 224 / 256 blocks ->
    //  9.21 / 9.15 / 9.17-9.45 / 9.25 ms)
    if (g_c1_blocks < 0) { const char* e = getenv("FPD_C1_BLOCKS"); g_c1_blocks = e ? atoi(e) : 192; }
    return g_c1_blocks < 1 ? 1 : g_c1_blocks;
}
// Smallest launch (pixels) the kernel takes in mode 1, forward launches and data gradients apart.  Kernels alone on the small maps
// (profiles/r06_c1_small_maps.txt): the forward 128 -> 64 wins at every size (7.5 vs 8.9-9.2 us down to 4x4 = 512 pixels), the data
// gradients with their fused weight gradient lose 1 us to conv_tile's plain data gradient below 32x32 -- but take the weight gradient
// off the lane.  Inside the step (one box, three interleaved runs each): forward / backward 2048 / 2048 -> 8.763, 8.823, 8.764 ms;
// 512 / 2048 -> 8.762, 8.760, 8.747; 512 / 8192 -> 8.755, 8.750, 8.768; 512 / 16384 -> 8.796, 8.763, 8.807; the earlier sweep of
// both together: 32768 / 8192 / 2048 / 512 -> 9.669 / 9.60 / 9.574 / 9.581.
int c1_min_px() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FPD_C1_MIN_PX"); v = e ? atoi(e) : 512; }
    return v;
}
int c1_min_px_bwd() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FPD_C1_MIN_PX_BWD"); v = e ? atoi(e) : 2048; }
    return v;
}
int c1_fuse_wgrad() { return 1; }

bool c1_chan(int c) { return c == 32 || c == 64 || c == 128; }
bool c1_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool c1_domain(const fpd_conv_t& a) {
    if (a.dtype != FPD_BF16 || a.R != 1 || a.S != 1 || a.stride != 1 || a.pad != 0 || a.P != a.H || a.Q != a.W) return false;
    // (16 -> 128: the inter-stack score_ convolution and the data gradient of the score convolution, hourglass.py:136-137, one k-step)
    if (!(c1_chan(a.C) || (a.C == 16 && a.K == 128)) || !c1_chan(a.K)) return false;
    const long long M = (long long)a.N * a.H * a.W;
    if (M % 32 != 0 || M > (1ll << 30)) return false;
    if (!c1_aligned(a.x) || !c1_aligned(a.y) || !c1_aligned(a.w) || !c1_aligned(a.residual) || !c1_aligned(a.epi_x) ||
        !c1_aligned(a.fold_x) || !c1_aligned(a.fold_out)) return false;
    if (a.epi == FPD_EPI_BNRELU_BWD) {
        // the data gradients of the hot path: no bias, no accumulate source, no prologue BN; dW tiles for every wave
        if (a.bias != nullptr || a.residual != nullptr || a.bn.mode != FPD_BN_NONE) return false;
        if (a.C == 16) { if (a.fold_x != nullptr) return false; }      // (no dW tiles: its weight gradient stays a separate launch)
        // 128 <-> 64 with the fused weight gradient; 128 x 128 (the dy + a(u) tiles of 8 waves do not fit the LDS) as a plain data
        // gradient, its weight gradient a launch of its own on the lane
        else if ((a.C / 32) * (a.K / 32) != C1_NW && !(a.C == 128 && a.K == 128 && a.fold_x == nullptr)) return false;
    } else {
        if (a.epi != FPD_EPI_PLAIN || a.fold_x != nullptr || a.wg_partial != nullptr) return false;
        if (a.y == a.x) return false;
    }
    return true;
}
bool c1_wg_shape(const fpd_conv_t& a) {
    return c1_fuse_wgrad() != 0 && c1_domain(a) && a.epi == FPD_EPI_BNRELU_BWD && (a.C / 32) * (a.K / 32) == C1_NW;
}
int c1_rounds(const fpd_conv_t& a) { return cdiv(a.N * a.H * a.W / 32, C1_NW); }

bool c1_takes(const fpd_conv_t& a, const fpd_conv_t* b) {
    const int mode = c1_mode();
    if (mode == 0 || !c1_domain(a)) return false;
    long long px = (long long)a.N * a.H * a.W;
    if (b != nullptr) {
        if (!c1_domain(*b) || a.K != b->K || a.C != b->C || a.epi != b->epi) return false;
        if ((a.residual != nullptr) != (b->residual != nullptr)) return false;      // (the residual is a template parameter)
        px += (long long)b->N * b->H * b->W;
    }
    return mode != 1 || px >= (a.epi == FPD_EPI_BNRELU_BWD ? c1_min_px_bwd() : c1_min_px());
}

struct C1Plan { int na, nb, grid; bool wg; };
bool c1_plan(const fpd_conv_t& a, const fpd_conv_t* b, bool want_wg, C1Plan& pl) {
    const int ra = c1_rounds(a), rb = b ? c1_rounds(*b) : 0;
    pl.wg = want_wg && c1_wg_shape(a) && (b == nullptr || c1_wg_shape(*b));
    // blocks under the cap, balanced: every block runs the same number of rounds (512 rounds under a cap of 192 -> 171 blocks of 3,
    // not 192 blocks of which two thirds run 3 and the rest 2)
    int total = std::max(1, std::min(c1_blocks(), ra + rb));
    total = cdiv(ra + rb, cdiv(ra + rb, total));
    pl.nb = 0;
    if (b != nullptr) {
        if (total < 2) return false;
        pl.nb = std::max(1, std::min(total - 1, (int)((long long)total * rb / (ra + rb))));
    }
    pl.na = total - pl.nb;
    pl.grid = total;
    return true;
}

template <int C, int K, bool BWD, bool FOLD, bool WG, bool RES>
int c1_launch_t(const fpd_conv_t& a, const fpd_conv_t* b, const C1Plan& pl, hipStream_t st) {
    static LdsAttr configured;
#ifdef FPD_C1_TIMING
    constexpr size_t lds = C1Geo<C, K, BWD, WG>::LDS + 1024;
#else
    constexpr size_t lds = C1Geo<C, K, BWD, WG>::LDS;
#endif
    static_assert(lds <= 160 * 1024, "conv_c1: LDS budget");
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&c1_kernel<C, K, BWD, FOLD, WG, RES>), lds)) return rc_;
    C1Args args;
    args.c[0] = a; args.c[1] = b ? *b : a; args.nblk[0] = pl.na; args.nblk[1] = pl.nb;
    FPD_LAUNCH((c1_kernel<C, K, BWD, FOLD, WG, RES>), dim3(pl.grid), dim3(512), lds, st, args);
    g_c1_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

template <int C, int K>
int c1_launch_ck(const fpd_conv_t& a, const fpd_conv_t* b, const C1Plan& pl, hipStream_t st) {
    if (a.epi == FPD_EPI_BNRELU_BWD) {
        if constexpr ((C / 32) * (K / 32) == C1_NW) {
            if (a.fold_x != nullptr || (b != nullptr && b->fold_x != nullptr))
                return pl.wg ? c1_launch_t<C, K, true, true, true, false>(a, b, pl, st) : c1_launch_t<C, K, true, true, false, false>(a, b, pl, st);
            return pl.wg ? c1_launch_t<C, K, true, false, true, false>(a, b, pl, st) : c1_launch_t<C, K, true, false, false, false>(a, b, pl, st);
        } else if constexpr (C == 16 || (C == 128 && K == 128)) {
            return c1_launch_t<C, K, true, false, false, false>(a, b, pl, st);
        } else {
            return 1;
        }
    }
    if (a.residual != nullptr) return c1_launch_t<C, K, false, false, false, true>(a, b, pl, st);
    return c1_launch_t<C, K, false, false, false, false>(a, b, pl, st);
}

template <int C>
int c1_launch_c(const fpd_conv_t& a, const fpd_conv_t* b, const C1Plan& pl, hipStream_t st) {
    switch (a.K) {
        case 32: return c1_launch_ck<C, 32>(a, b, pl, st);
        case 64: return c1_launch_ck<C, 64>(a, b, pl, st);
        default: return c1_launch_ck<C, 128>(a, b, pl, st);
    }
}

int c1_launch(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    const bool want_wg = a.wg_partial != nullptr || (b != nullptr && b->wg_partial != nullptr);
    C1Plan pl;
    if (!c1_plan(a, b, want_wg, pl)) return 1;
    if (want_wg) {
        if (!pl.wg) return fpd_fail(-2, "conv: a fused weight gradient was requested for a launch fpd_conv_fused_wgrad_partials() reports 0 for");
        const fpd_conv_t* cs[2] = {&a, b};
        const int nblk[2] = {pl.na, pl.nb};
        for (int i = 0; i < 2; ++i) {
            if (cs[i] == nullptr || cs[i]->wg_partial == nullptr) continue;
            if (cs[i]->wg_stride < (int64_t)cs[i]->C * cs[i]->K + cs[i]->C)
                return fpd_fail(-2, "conv: wg_stride %lld smaller than weight + bias", (long long)cs[i]->wg_stride);
            if (cs[i]->wg_count != nblk[i])
                return fpd_fail(-2, "conv: the launch writes %d weight-gradient slabs but the caller sized its workspace for %d "
                                    "(fpd_conv_fused_wgrad_partials: has a conv_c1 option changed since?)", nblk[i], cs[i]->wg_count);
        }
        if ((a.wg_partial == nullptr) != (b != nullptr && b->wg_partial == nullptr) && b != nullptr)
            return fpd_fail(-2, "conv_pair: both or neither convolution of a pair take a fused weight gradient");
    }
    switch (a.C) {
        case 16: return c1_launch_ck<16, 128>(a, b, pl, st);
        case 32: return c1_launch_c<32>(a, b, pl, st);
        case 64: return c1_launch_c<64>(a, b, pl, st);
        default: return c1_launch_c<128>(a, b, pl, st);
    }
}

}  // namespace

int fpd_conv_c1_option(int which, int value) {      // which: 0 = mode, 1 = blocks (returns the previous value), 2 = launches served so far
    if (which == 2) return g_c1_launches.load(std::memory_order_relaxed);
    int& g = which == 0 ? g_c1_mode : g_c1_blocks;
    const int prev = which == 0 ? c1_mode() : c1_blocks();
    g = value;
    return prev;
}

// 0 = launched, 1 = outside this kernel's domain (the caller tries conv_pp next), < 0 error
int fpd_conv_c1_launch(const fpd_conv_t& a, hipStream_t st) {
    if (!c1_takes(a, nullptr)) return 1;
    return c1_launch(a, nullptr, st);
}
int fpd_conv_c1_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    if (!c1_takes(a, &b)) return 1;
    return c1_launch(a, &b, st);
}
// 1 if the launch (pair) is served by this kernel as a BNRELU_BWD data gradient: a folded BN-backward apply is a run-time flag
int fpd_conv_c1_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b) {
    if (a.epi != FPD_EPI_BNRELU_BWD || (b != nullptr && b->epi != FPD_EPI_BNRELU_BWD)) return 0;
    C1Plan pl;
    return (c1_takes(a, b) && c1_plan(a, b, false, pl)) ? 1 : 0;
}
// -1 = this kernel does not take the launch (ask the next kernel); else the slabs of the fused weight gradient (0: served
// here, but without the fusion)
int fpd_conv_c1_wgrad_partials(const fpd_conv_t& a) {
    C1Plan pl;
    if (!c1_takes(a, nullptr) || !c1_plan(a, nullptr, true, pl)) return -1;
    return pl.wg ? pl.na : 0;
}
int fpd_conv_c1_pair_wgrad_partials(const fpd_conv_t& a, const fpd_conv_t& b, int* na, int* nb) {
    C1Plan pl;
    *na = *nb = 0;
    if (!c1_takes(a, &b) || !c1_plan(a, &b, true, pl)) return -1;
    if (pl.wg) { *na = pl.na; *nb = pl.nb; }
    return 0;
}
