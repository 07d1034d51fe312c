// conv_pp: PERSISTENT convolution for the big feature maps of the student chain (bf16, stride-1 1x1 / 3x3 "same"
// convolutions with C, K <= 128: the train-mode convolutions and data gradients of the hourglass at 64x64 and 128x128,
// where a launch has >= 512 pixel tiles).  Same contract as conv_tile_kernel (fpd_conv_t): BatchNorm+ReLU prologue on
// the operand, bias / residual, batch statistics of the result or the BN-backward epilogue.
//
// Why a second kernel: conv_tile re-fetches the weight tiles per tap and block, flushes statistics per 128-pixel block
// and stages its epilogue with 4-byte LDS stores, so at 64x64 a convolution takes 24-34 us against a 7-17 us HBM bound
// (DESIGN.md section 5, r02 ablation: the epilogue alone is 12-19 us).  Here
//   * a block (8 wave64) is resident for the whole launch and walks a contiguous range of pixel tiles; up to two blocks
//     share a CU (<= 128 registers per lane, LDS permitting), so one block's barriers / memory waits are the other's
//     issue slots;
//   * ALL weights of its <= 64-channel output slab (<= 72 KB: 3x3 64->64) are staged into LDS ONCE per block,
//     XOR-swizzled (16-byte chunk c of row n sits at c ^ sw(n)): unpadded rows, conflict-free b128 fragment reads;
//   * the next tile's operand rows are requested as soon as the current tile is staged (in flight during its MFMAs
//     and epilogue; global loads survive the barriers), all loads are issued branch-free (clamped addresses, validity
//     kept as a bit mask);
//   * the MFMAs run transposed (first operand = weights): a lane owns 4 consecutive channels of one pixel, so the
//     accumulators go to a WAVE-PRIVATE fp32 LDS staging tile as 16-byte stores and leave as 16-byte vectors of
//     8 channels (+ residual + bias, one rounding -- the same rounding points as conv_tile);
//   * statistics are accumulated per thread over all tiles of the block and flushed once per block.
// An 8-wave block computes 128 pixels x 64 channels per tile (4 pixel groups x 2 channel halves), or 256 pixels x 32
// channels when K <= 32.  K = 128: two slabs = two blocks per tile range, 8 block ids apart (same XCD: its L2 serves
// the operand rows to the second slab).  A first version split the 8 waves into two groups working half a period
// apart (MFMA of one beside the epilogue / staging of the other): correct, but the VALU-heavy half ran at 1 wave per
// SIMD and bounded the tile (profiles/r03_conv_pp_stamps.txt); all waves on every phase + two blocks per CU is faster.
//
// Replaces the same reference calls as conv_tile (nn.Conv2d + BatchNorm2d + ReLU, /root/reference/lib/models/hourglass.py:18-52).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"
#include "mfma_frag.h"

namespace {

constexpr int pp_ilog2(int v) { return v <= 1 ? 0 : 1 + pp_ilog2(v / 2); }

struct PPGeo {
    int nrows;                // image rows per tile
    int ntiles;               // pixel tiles of this convolution
    int nblk;                 // tile ranges (persistent blocks per channel slab) working on it
    int region;               // bytes of the LDS region holding the operand image / the epilogue staging tiles, multiple of 16
    unsigned mW, mWV, mH;     // multiply-high reciprocals of W, W * (C / 8), H
};
__device__ __forceinline__ int pp_qdiv(int v, unsigned magic) { return (int)__umulhi((unsigned)v, magic); }
static inline unsigned pp_magic(int d) { return (unsigned)((0x100000000ull / (unsigned long long)d) + 1ull); }

// launder a value: everything derived from it is recomputed where it is used instead of being hoisted out of the tile loop
// and kept in registers for the whole kernel (the register budget bounds this kernel, not a few VALU operations)
__device__ __forceinline__ int pp_fresh(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// Two bf16 of a dword as two floats and back: the forward arithmetic below is written on PAIRS (v_pk_fma_f32 / v_pk_add_f32 /
// v_pk_mul_f32, v_pk_max_i16 on the packed result) -- the same operation per element in the same order as the scalar form, so the
// same bits, at fewer vector instructions (r04 ISA count of the 1x1 128->64 forward tile loop: 513 -> 406 per tile and wave,
// against 8 MFMAs).  Measured (r04, interleaved A/B + per-shape trace A/B): 2-4 % on the 1x1 forward launches in isolation, nothing
// on the step -- the tile loop is bound by its barriers and LDS round trips, not by VALU issue.  The BN-backward epilogue stays
// scalar: its packed form was 2-7 % slower on the +wgrad +fold kernels (more registers, longer dependent chains).
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 bf2_unpack(unsigned w) {
    f32x2 r = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    return r;
}
__device__ __forceinline__ unsigned bf2_pack(f32x2 v) { return f2bf_pk(v[0], v[1]); }
// max of the two signed 16-bit halves with `lo` in both: lo = 0 is ReLU on a packed bf16 pair (a negative bf16 is a negative
// int16; rounding is monotonic and keeps the sign, so relu(round(t)) == round(relu(t))), lo = -32768 is the identity.
// Not the bits of the scalar fmaxf path in two corners: -0.0 becomes +0.0 (equal as an MFMA operand and in
// every sum), and a NaN survives with its sign bit clear / becomes 0 with it set, where fmaxf(NaN, 0) = 0 -- a NaN activation
// means the step has diverged either way; the cross-kernel bit tests use finite inputs.
__device__ __forceinline__ unsigned bf2_floor(unsigned w, short lo) {
    s16x2 a = *reinterpret_cast<const s16x2*>(&w);
    const s16x2 b = {lo, lo};
    a = __builtin_elementwise_max(a, b);
    return *reinterpret_cast<const unsigned*>(&a);
}

#ifdef FPD_PP_TIMING      // probe build only (tools/probes): cycle stamps of two blocks at the phase boundaries, printed by the kernel
#define PP_STAMP() do { if (tid == 0 && s_ns < 100) s_stamp[s_ns++] = clock64(); } while (0)
#else
#define PP_STAMP() do { } while (0)
#endif

// KH = 2: tile = 128 pixels x 64 channels (wave = pixel group w % 4, channel half w / 4); KH = 1: 256 pixels x 32 channels
// WG (BWD, 1x1 only): the launch also forms the weight / bias gradient of the forward convolution it is the data gradient of
// (fpd_conv_t.wg_partial): dW[k][c] = sum_p dy[p][k] * a(u)[p][c] from the operand image (dy, pixel-major in the LDS anyway) and
// the forward operand a(u) = relu?(bn(u)) the epilogue evaluates for its ReLU mask -- both through transposing LDS reads.
template <int R, int C, int KH, bool BWD, bool WG>
__device__ __forceinline__ void conv_pp_body(const fpd_conv_t& a, const PPGeo geo, const int bi, const int n0) {
    constexpr int RS = R * R, KP = 32 * KH, PXW = 8 / KH, CPR = C / 8, LOG_CPR = pp_ilog2(CPR), KS = C / 16;
    constexpr int LDA = C * 2 + 16;                       // operand-image pixel pitch in BYTES (16 B pad: conflict-free b128 reads)
    constexpr int LDST = 32 * 4 + 16;                     // pitch of a wave's [32 px][32 ch] fp32 staging tile
    constexpr int NVH = 4;                                // operand vectors per thread and tile (host guarantees the fit)
    constexpr int pad = (R - 1) / 2;
    constexpr int RPB = CPR >= 16 ? 1 : 16 / CPR;         // weight rows per 256-byte bank row
    constexpr int NWV = (KP * RS * CPR + 511) / 512;      // weight vectors per thread
    constexpr int LDAT = KP * 2 + 16;                     // WG: pitch of the forward-operand tile [px][KP] bf16
    constexpr int NTILE = WG ? (C / 32) * KH : 1;         // WG: 32x32 tiles of dW this block owns (k tiles x c tiles)
    constexpr int NPART = WG ? 8 / NTILE : 1;             // WG: waves per tile, splitting the pixel steps
    static_assert(!WG || (BWD && R == 1 && C >= 32 && NTILE * NPART == 8), "fused weight gradient: 1x1 data gradients only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave % PXW, hc = wave / PXW;           // pixel group / channel half of this wave
    const int H = a.H, W = a.W, K = a.K;
    const int M = a.N * H * W, GR = a.N * H;
    const int nrows = geo.nrows, hrows = nrows + R - 1, WP = W + R - 1;
    const int TPX = nrows * W;                            // pixels per tile (<= 32 * PXW)
    const int zero_px = hrows * WP;

    float* s_scale = reinterpret_cast<float*>(smem);      // [C]
    float* s_shift = s_scale + C;                         // [C]
    float* s_epi = s_shift + C;                           // [4][KP] (BNRELU_BWD)
    float* s_bias = s_epi + 4 * KP;                       // [KP]
    float* s_fold = s_bias + KP;                          // BWD: [3][C] coefficients of a folded BN-backward apply (fold_x)
    unsigned char* sW = reinterpret_cast<unsigned char*>(s_fold + (BWD ? 3 * C : 0));
    unsigned char* sG = sW + RS * KP * C * 2;             // operand image; the waves' epilogue staging tiles alias it
    unsigned char* sS = WG ? sG + geo.region : sG;        // ... unless the weight gradient needs the image after the epilogue
    unsigned char* sAT = sS + 8 * 32 * LDST;              // WG: a(u) tile
    const bool wg = WG && a.wg_partial != nullptr;
#ifdef FPD_PP_TIMING
    long long* s_stamp = reinterpret_cast<long long*>(sG + geo.region + (WG ? 8 * 32 * LDST + 32 * PXW * LDAT : 0));
    int s_ns = 0;
    if (tid < 100) s_stamp[tid] = 0;
    __syncthreads();
#endif
    PP_STAMP();
    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(a.y);
    const bf16_t* res = reinterpret_cast<const bf16_t*>(a.residual);
    const bf16_t* ex = reinterpret_cast<const bf16_t*>(a.epi_x);
    const bool want_stats = BWD || (a.out_stats != nullptr);
    // FOLDED BN-BACKWARD APPLY (fpd_conv_t.fold_x): the operand proper is dy = A_c g + B_c u + D_c, g = x (the masked
    // gradient), u = fold_x; evaluated on the way into the operand image, rounded once like the stand-alone apply's output
    const bool fold = BWD && a.fold_x != nullptr;
    const bf16_t* __restrict__ fx = reinterpret_cast<const bf16_t*>(a.fold_x);
    bf16_t* fo = (fold && n0 == 0) ? reinterpret_cast<bf16_t*>(a.fold_out) : nullptr;     // written once: by the first K slab

    // ---- this block's tiles: a contiguous range ----
    const int t_beg = fpd_cut(bi, geo.ntiles, geo.nblk);
    const int t_end = fpd_cut(bi + 1, geo.ntiles, geo.nblk);

    // ---- operand staging: vector v = tid + 512 i of the tile's hrows x W x CPR operand vectors (a thread always stages the
    //      same 8 channels: 512 % CPR == 0).  Loads are unconditional (rows outside the tensor: clamped address, cleared bit).
    const int WV = W * CPR, nvtot = hrows * WV;
    const int nvh = (nvtot + 511) >> 9;                   // uniform trip count
    uint4 rh[NVH];
    uint4 ru[BWD ? NVH : 1];                              // fold: the same vectors of u
    unsigned hmask = 0;
    auto halo_load = [&](int tile) {
        const int g0 = tile * nrows - pad;
        const int tl = pp_fresh(tid);
        const int cve = (tl & (CPR - 1)) * 8;
        hmask = 0;
        if constexpr (R == 1) {
            // no halo: a tile is TPX consecutive pixels of the NHWC tensor, vector v is pixel v / CPR of the tile -- no row / column
            // arithmetic at all (this loop is VALU-bound: 513 vector instructions per tile and wave before, r04 ISA count)
            const int m0t = tile * TPX;
#pragma unroll
            for (int i = 0; i < NVH; ++i) {
                if (i < nvh) {
                    const int v = min(tl + i * 512, nvtot - 1);
                    const int m = m0t + (v >> LOG_CPR);
                    const size_t off = (size_t)min(m, M - 1) * C + cve;
                    rh[i] = *reinterpret_cast<const uint4*>(x + off);
                    if (BWD && fold) ru[i] = *reinterpret_cast<const uint4*>(fx + off);
                    hmask |= (m < M ? 1u : 0u) << i;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (i < nvh) {
                const int v = min(tl + i * 512, nvtot - 1);
                const int hr = pp_qdiv(v, geo.mWV);
                const int j = (v - hr * WV) >> LOG_CPR;
                const int g = g0 + hr;
                const int gc = min(max(g, 0), GR - 1);
                rh[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(gc * W + j) * C + cve));
                if (BWD && fold) ru[i] = *reinterpret_cast<const uint4*>(fx + ((size_t)(gc * W + j) * C + cve));
                hmask |= (g == gc ? 1u : 0u) << i;
            }
        }
    };
    const short relu_floor = a.bn.relu ? (short)0 : (short)-32768;       // see bf2_floor
    float bs[8];                                          // WG: this thread's share of sum_pixels dy[., 8 channels]
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = 0.f;
    const bool wg_bias = wg && a.wg_bias && n0 == 0;
    auto halo_store = [&](const int tile) {
        const int tl = pp_fresh(tid);
        const int cvb = (tl & (CPR - 1)) * 16;
        const int g0s = tile * nrows - pad;
        // zero border columns and zero pixels first (the epilogue staging of the previous tile overwrote them)
        {
            const uint4 z = make_uint4(0, 0, 0, 0);
            const int nb = (R == 3) ? 2 * hrows : 0;
            for (int v = tl; v < (nb + 3) * CPR; v += 512) {
                const int pz = v >> LOG_CPR, cv = (v & (CPR - 1)) * 16;
                const int px = pz < nb ? ((pz >> 1) * WP + ((pz & 1) ? WP - 1 : 0)) : zero_px + (pz - nb);
                *reinterpret_cast<uint4*>(sG + px * LDA + cv) = z;
            }
        }
        f32x4 fa0, fa1, fb0, fb1, fd0, fd1;
        if (BWD && fold) {
            const float* t = s_fold + (cvb >> 1);
            fa0 = *reinterpret_cast<const f32x4*>(t); fa1 = *reinterpret_cast<const f32x4*>(t + 4);
            fb0 = *reinterpret_cast<const f32x4*>(t + C); fb1 = *reinterpret_cast<const f32x4*>(t + C + 4);
            fd0 = *reinterpret_cast<const f32x4*>(t + 2 * C); fd1 = *reinterpret_cast<const f32x4*>(t + 2 * C + 4);
        }
        f32x4 sc0, sc1, sh0, sh1;
        const bool has_bn = a.bn.mode != FPD_BN_NONE;
        if (has_bn) {
            sc0 = *reinterpret_cast<const f32x4*>(s_scale + (cvb >> 1));
            sc1 = *reinterpret_cast<const f32x4*>(s_scale + (cvb >> 1) + 4);
            sh0 = *reinterpret_cast<const f32x4*>(s_shift + (cvb >> 1));
            sh1 = *reinterpret_cast<const f32x4*>(s_shift + (cvb >> 1) + 4);
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (i < nvh) {
                const int v = tl + i * 512;
                const int vc = min(v, nvtot - 1);
                int hr, j;
                if constexpr (R == 1) { hr = 0; j = vc >> LOG_CPR; }        // (pixel of the tile: rows are contiguous without a halo)
                else { hr = pp_qdiv(vc, geo.mWV); j = (vc - hr * WV) >> LOG_CPR; }
                uint4 val = rh[i];
                if (BWD && fold) {
                    float g[8], u[8];
                    DT<bf16_t>::unpack(val, g);
                    DT<bf16_t>::unpack(ru[i], u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        g[e] = fmaf(fa0[e], g[e], fmaf(fb0[e], u[e], fd0[e]));
                        g[4 + e] = fmaf(fa1[e], g[4 + e], fmaf(fb1[e], u[4 + e], fd1[e]));
                    }
                    val = DT<bf16_t>::pack(g);
                    const bool in = (hmask >> i) & 1u;    // rows outside the tensor stay exactly zero (D_c is not)
                    val.x = in ? val.x : 0u; val.y = in ? val.y : 0u; val.z = in ? val.z : 0u; val.w = in ? val.w : 0u;
                    // the evaluated operand is written out once for its other consumers (a separate weight-gradient launch):
                    // rows of this tile proper only -- halo rows belong to the neighbouring tiles
                    if (fo != nullptr && in && v < nvtot && hr >= pad && hr < pad + nrows)
                        *reinterpret_cast<uint4*>(fo + ((size_t)((g0s + hr) * W + j) * C + (cvb >> 1))) = val;     // (R == 1: g0s * W + j = the pixel)
                }
                if (WG && wg_bias) {
                    float f[8];
                    DT<bf16_t>::unpack(val, f);
                    const float cnt = (v < nvtot && ((hmask >> i) & 1u)) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs[e] = fmaf(f[e], cnt, bs[e]);
                }
                if (has_bn) {
                    // a = relu?(x * scale + shift), rounded once: fma on pairs, ReLU on the packed result
                    const f32x2 s0 = {sc0[0], sc0[1]}, s1 = {sc0[2], sc0[3]}, s2 = {sc1[0], sc1[1]}, s3 = {sc1[2], sc1[3]};
                    const f32x2 h0 = {sh0[0], sh0[1]}, h1 = {sh0[2], sh0[3]}, h2 = {sh1[0], sh1[1]}, h3 = {sh1[2], sh1[3]};
                    val.x = bf2_floor(bf2_pack(__builtin_elementwise_fma(bf2_unpack(val.x), s0, h0)), relu_floor);
                    val.y = bf2_floor(bf2_pack(__builtin_elementwise_fma(bf2_unpack(val.y), s1, h1)), relu_floor);
                    val.z = bf2_floor(bf2_pack(__builtin_elementwise_fma(bf2_unpack(val.z), s2, h2)), relu_floor);
                    val.w = bf2_floor(bf2_pack(__builtin_elementwise_fma(bf2_unpack(val.w), s3, h3)), relu_floor);
                    if constexpr (R != 1) {               // halo rows outside the tensor stay exactly zero (a 1x1 convolution has
                        const bool in = (hmask >> i) & 1u;    // no halo: pixels past the tensor are dead lanes that read the zero pixel)
                        val.x = in ? val.x : 0u; val.y = in ? val.y : 0u; val.z = in ? val.z : 0u; val.w = in ? val.w : 0u;
                    }
                }
                // threads past the last vector of a ragged tile write to the spare 16 bytes behind the zero pixels
                const int dst = v < nvtot ? (R == 1 ? j : hr * WP + j + pad) * LDA + cvb : (zero_px + 3) * LDA;
                *reinterpret_cast<uint4*>(sG + dst) = val;
            }
        }
    };

    // ---- prologue.  Three independent load -> fp64 arithmetic -> table chains (operand BN, epilogue BN of a data gradient,
    //      bias) go to different waves and are REQUESTED first; then the first operand rows and the weights (all hands);
    //      only then is anything waited for -- the chains run concurrently with each other and under the long loads instead
    //      of one after the other in waves 0-1 (measured: 6.7 k -> cycles from entry to the first barrier).
    BnRaw braw;
    float bias_raw;                                      // (set in the r_bias branch only: a default written here is sunk by hipcc
                                                          //  behind the other branches' loads, where it needs vmcnt(0) -- see bn_request)
    const int te = tid - 128, tb = tid - 256;
    const bool r_bn = a.bn.mode != FPD_BN_NONE && tid < C;
    const bool r_epi = a.epi == FPD_EPI_BNRELU_BWD && te >= 0 && te < KP;
    const bool r_bias = tb >= 0 && tb < KP;
    const bool r_fold = fold && tid < C;
    StatRaw fs1, fs2;
    if (r_bn) bn_request(a.bn, tid, C, braw);
    else if (r_fold) {
        bn_request(a.fold_bn, tid, C, braw);
        stat_request(a.fold_stats, C, 0, tid, fs1);
        stat_request(a.fold_stats, C, 1, tid, fs2);
    }
    else if (r_epi && n0 + te < K) bn_request(a.epi_bn, n0 + te, K, braw);
    else if (r_bias) { bias_raw = 0.f; if (a.bias != nullptr && n0 + tb < K) bias_raw = a.bias[n0 + tb]; }
    __builtin_amdgcn_sched_barrier(0);
    if (t_beg < t_end) halo_load(t_beg);
    PP_STAMP();
    {
        uint4 rw[NWV];
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            const int k = v / (RS * CPR), rem = v - k * (RS * CPR);
            rw[i] = make_uint4(0, 0, 0, 0);
            if (v < KP * RS * CPR && n0 + k < K) rw[i] = *reinterpret_cast<const uint4*>(w + ((size_t)(n0 + k) * RS * C + rem * 8));
        }
        __builtin_amdgcn_sched_barrier(0);
        PP_STAMP();
        if (r_bn) {
            float sc, sh, mu, is;
            bn_resolve(braw, (double)M, sc, sh, mu, is);
            s_scale[tid] = sc;
            s_shift[tid] = sh;
        } else if (r_fold) {
            // dy = gamma*is*(g - m1 - xhat*m2), xhat = (u - mu)*is  ==  A g + B u + D   (coefficients formed in fp64)
            const double s1 = stat_resolve(braw.s1), s2 = stat_resolve(braw.s2), b1 = stat_resolve(fs1), b2 = stat_resolve(fs2);
            const double cnt = (double)M, mu = s1 / cnt;
            double var = s2 / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)braw.eps), gi = (double)braw.g * is;
            const double m1 = b1 / cnt, m2 = b2 / cnt;
            s_fold[tid] = (float)gi;
            s_fold[C + tid] = (float)(-gi * is * m2);
            s_fold[2 * C + tid] = (float)(gi * (mu * is * m2 - m1));
            if (bi == 0 && n0 == 0) {                     // the affine parameters' gradients fall out of the two sums
                if (a.fold_dgamma != nullptr) a.fold_dgamma[tid] = (float)b2;
                if (a.fold_dbeta != nullptr) a.fold_dbeta[tid] = (float)b1;
            }
        } else if (r_epi) {
            float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
            if (n0 + te < K) bn_resolve(braw, (double)M, sc, sh, mu, is);
            s_epi[te] = sc; s_epi[KP + te] = sh; s_epi[2 * KP + te] = mu; s_epi[3 * KP + te] = is;
        } else if (r_bias) {
            s_bias[tb] = bias_raw;
        }
        PP_STAMP();
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

    // ---- per-lane MFMA addressing ----
    const int ml = wq * 32 + l31;                         // pixel of the tile this lane feeds (second MFMA operand)
    const int ti = pp_qdiv(ml, geo.mW), tj = ml - ti * W;
    int ab[3];
    auto tile_addr = [&](int tile) {
        if constexpr (R == 1) {                           // pixel ml of the tile sits at image pixel ml
            const bool live1 = tile * TPX + ml < M && ml < TPX;
            ab[0] = ab[1] = ab[2] = live1 ? ml * LDA : zero_px * LDA;
            return;
        }
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

    // ---- epilogue state: a lane reads back the 8-channel chunk cv4 of rows r16, r16 + 16 of its wave's 32 px x 32 ch block ----
    uint4 rres[2], rex[BWD ? 2 : 1];
    f32x2 F1[4], F2[4], CS[4];                            // statistics partials / common shift of this lane's 8 channels, as pairs
    int nrow = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { F1[e] = f32x2{0.f, 0.f}; F2[e] = f32x2{0.f, 0.f}; CS[e] = f32x2{0.f, 0.f}; }
    const float relu_gate = a.epi_bn.relu ? 0.f : -3.4e38f;
    const int kw0 = n0 + hc * 32;                         // first channel of this wave's block
    auto request = [&](int tile) {                        // residual / epi_x vectors of `tile`
        const int ln = pp_fresh(lane);
        const int cv4 = ln & 3, r16 = ln >> 2;
        const int k0 = min(kw0 + cv4 * 8, K - 8);         // (clamped: channel chunks past K are never stored)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = min(tile * TPX + wq * 32 + r16 + 16 * it, M - 1);
            const size_t off = (size_t)m * K + k0;
            if (res != nullptr) rres[it] = *reinterpret_cast<const uint4*>(res + off);
            if (BWD) rex[it] = *reinterpret_cast<const uint4*>(ex + off);
        }
    };
    f32x16 acc;
    f32x16 wacc;                                          // WG: this wave's 32x32 tile of dW (its share of the pixel steps)
#pragma unroll
    for (int e = 0; e < 16; ++e) wacc[e] = 0.f;
    auto epilogue = [&](int tile, auto firstc) {
        constexpr bool first = decltype(firstc)::value;    // the block's first tile (peeled: it alone picks the shift of the sums)
        const int ln = pp_fresh(lane);
        const int cv4 = ln & 3, r16 = ln >> 2;
        unsigned char* stg = sS + wave * (32 * LDST);     // wave-private [32 px][LDST]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[4 * j + e];
            *reinterpret_cast<f32x4*>(stg + l31 * LDST + (8 * j + 4 * hh) * 4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's staging stores are in the LDS
        const int kl = hc * 32 + cv4 * 8;                 // channel chunk inside the slab
        const int k0 = n0 + kl;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = r16 + 16 * it;
            const int px = wq * 32 + row;
            const int m = tile * TPX + px;
            const bool ok = px < TPX && m < M && k0 < K;
            const float live = ok ? 1.f : 0.f;
            const f32x2 live2 = {live, live};
            const unsigned rr[4] = {rres[it].x, rres[it].y, rres[it].z, rres[it].w};
            const unsigned xr[4] = {rex[BWD ? it : 0].x, rex[BWD ? it : 0].y, rex[BWD ? it : 0].z, rex[BWD ? it : 0].w};
            unsigned pw[4];
            unsigned aw[4] = {0u, 0u, 0u, 0u};            // WG: bf16 a(u) of this row's 8 channels
            // 4 channels at a time, as two pairs (packed fp32 arithmetic: the same operation per element in the same order as
            // the scalar form); the per-channel tables are re-read from the LDS (laundered address) instead of living in
            // registers across the tile loop: v = acc + residual + bias, rounded once
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int klq = pp_fresh(kl) + 4 * q;
                const int p0 = 2 * q, p1 = 2 * q + 1;
                const f32x4 t = *reinterpret_cast<const f32x4*>(stg + row * LDST + cv4 * 32 + 16 * q);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(s_bias + klq);
                if constexpr (BWD) {
                    // (scalar form: the packed form of this branch measured 2-7 % SLOWER on the +wgrad +fold kernels, r04 trace A/B)
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = t[e];
                    if (res != nullptr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const unsigned wd = rr[2 * q + (e >> 1)];
                            v[e] += __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16));
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bq[e];
                    // ReLU mask of the forward tensor + the two BatchNorm-backward sums
                    const float* te = s_epi + klq;
                    const f32x4 esc = *reinterpret_cast<const f32x4*>(te);
                    const f32x4 esh = *reinterpret_cast<const f32x4*>(te + KP);
                    const f32x4 emu = *reinterpret_cast<const f32x4*>(te + 2 * KP);
                    const f32x4 eis = *reinterpret_cast<const f32x4*>(te + 3 * KP);
                    float xv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned wd = xr[2 * q + (e >> 1)];
                        xv[e] = __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16));
                        const float z = fmaf(xv[e], esc[e], esh[e]);
                        v[e] = (z > relu_gate) ? v[e] : 0.f;
                    }
                    if (WG) {
                        // the forward operand exactly as the forward convolution staged it: bf16(max(fma(u, scale, shift), lo))
                        float az[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) az[e] = fmaxf(fmaf(xv[e], esc[e], esh[e]), relu_gate);
                        aw[p0] = ok ? f2bf_pk(az[0], az[1]) : 0u;
                        aw[p1] = ok ? f2bf_pk(az[2], az[3]) : 0u;
                    }
                    pw[p0] = f2bf_pk(v[0], v[1]);
                    pw[p1] = f2bf_pk(v[2], v[3]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {                    // the stored (rounded) gradient is what gets summed
                        const unsigned wd = pw[2 * q + (e >> 1)];
                        const float g = __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16)) * live;
                        F1[2 * q + (e >> 1)][e & 1] += g;
                        F2[2 * q + (e >> 1)][e & 1] = fmaf(g, (xv[e] - emu[e]) * eis[e], F2[2 * q + (e >> 1)][e & 1]);
                    }
                } else {
                    f32x2 v0 = {t[0], t[1]}, v1 = {t[2], t[3]};
                    if (res != nullptr) {
                        v0 += bf2_unpack(rr[p0]);
                        v1 += bf2_unpack(rr[p1]);
                    }
                    v0 += f32x2{bq[0], bq[1]};
                    v1 += f32x2{bq[2], bq[3]};
                    pw[p0] = bf2_pack(v0);
                    pw[p1] = bf2_pack(v1);
                    if (want_stats) {
                        if constexpr (first) if (it == 0) {
                            // the shift of the shifted sums must be COMMON to the 16 lanes that own a channel chunk: the rounded
                            // value of the wave's first pixel row (lanes r16 == 0 hold it), handed over through the LDS
                            float* sh = reinterpret_cast<float*>(stg) + cv4 * 8 + 4 * q;      // (row 0 of the staging tile was read above)
                            if (r16 == 0) {
                                const f32x2 a0 = bf2_unpack(pw[p0]), a1 = bf2_unpack(pw[p1]);
                                f32x4 c4 = {ok ? a0[0] : 0.f, ok ? a0[1] : 0.f, ok ? a1[0] : 0.f, ok ? a1[1] : 0.f};
                                *reinterpret_cast<f32x4*>(sh) = c4;
                            }
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            const f32x4 c4 = *reinterpret_cast<const f32x4*>(sh);
                            CS[p0] = f32x2{c4[0], c4[1]};
                            CS[p1] = f32x2{c4[2], c4[3]};
                        }
                        const f32x2 d0 = (bf2_unpack(pw[p0]) - CS[p0]) * live2, d1 = (bf2_unpack(pw[p1]) - CS[p1]) * live2;
                        F1[p0] += d0;
                        F1[p1] += d1;
                        F2[p0] = __builtin_elementwise_fma(d0, d0, F2[p0]);
                        F2[p1] = __builtin_elementwise_fma(d1, d1, F2[p1]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);        // (the two halves one after the other: fewer live temporaries)
            }
            if (want_stats && ok) ++nrow;
            if (ok) *reinterpret_cast<uint4*>(y + ((size_t)m * K + k0)) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            if (WG) *reinterpret_cast<uint4*>(sAT + px * LDAT + kl * 2) = make_uint4(aw[0], aw[1], aw[2], aw[3]);
            __builtin_amdgcn_sched_barrier(0);            // one row at a time: interleaving both rows doubles the live temporaries
        }
    };

    // =========================== the tile loop ===========================
    // Weight fragment of (tap, kk): row hc*32 + l31 of tile `tap`, 16-byte chunk (2 kk + hh) ^ sw(row); sw(l31 + 32) == sw(l31).
    auto do_tile = [&](int tile, auto firstc) {
        halo_store(tile);                                 // BN+ReLU / folded BN-backward apply on the way into the operand image
        if (tile + 1 < t_end) halo_load(tile + 1);        // in flight during this tile's MFMAs and epilogue
        if (res != nullptr || BWD) request(tile);
        tile_addr(tile);
        PP_STAMP();
        __syncthreads();                                  // operand image complete
        PP_STAMP();
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        if (hc < KH) {
            const int l31f = pp_fresh(l31);
            const unsigned char* wrow = sW + (hc * 32 + l31f) * (C * 2);
            const int wsw = (l31f / RPB) & (CPR - 1);
#pragma unroll
            for (int st = 0; st < RS * KS; ++st) {
                const int tap = st / KS, kk = st - tap * KS;
                const int r = tap / R, s = tap - r * R;
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sG + ab[r] + s * LDA + kk * 32 + hh * 16);
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + tap * KP * (C * 2) + ((2 * kk + hh) ^ wsw) * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
            }
        }
        PP_STAMP();
        if (!WG) __syncthreads();                         // every wave is done reading the image: staging tiles may overwrite it
        PP_STAMP();
        epilogue(tile, firstc);
        PP_STAMP();
        __syncthreads();                                  // staging tiles read back: the next operand image may overwrite them
        PP_STAMP();                                       // (WG: and the a(u) tile is complete)
        if (WG) {
            if (wg) {
                const int tw = wave % NTILE, part = wave / NTILE;
                const int tkf = tw / KH, tcf = tw % KH;
                const int nks = TPX >> 4;                 // pixel steps of 16 (TPX % 16 == 0: checked by the host)
                // compile-time trip count (the tile has at most 32 * PXW pixels) so that the transposing reads of the later
                // steps are in flight under the MFMAs of the earlier ones: a rolled loop pays one LDS latency per step
                // (stamps: 1.66 k cycles for 8 steps)
                constexpr int NKW = (2 * PXW + NPART - 1) / NPART;
#pragma unroll
                for (int i = 0; i < NKW; ++i) {
                    const int ks = part + i * NPART;
                    if (ks < nks) {
                        const bf16x8 af = tr_frag_bf16(reinterpret_cast<const bf16_t*>(sG), LDA / 2, ks * 16, tkf * 32, lane);
                        const bf16x8 bf = tr_frag_bf16(reinterpret_cast<const bf16_t*>(sAT), LDAT / 2, ks * 16, tcf * 32, lane);
                        wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, wacc, 0, 0, 0);
                    }
                }
            }
            PP_STAMP();
            __syncthreads();                              // image + a(u) tile free for the next tile
        }
    };
    // the first tile is peeled when the epilogue accumulates forward statistics: `first` as a run-time flag cost 40 selects per tile
    int tile0 = t_beg;
    if constexpr (!BWD) {
        if (want_stats && tile0 < t_end) do_tile(tile0++, std::true_type{});
    }
    for (int tile = tile0; tile < t_end; ++tile) do_tile(tile, std::false_type{});

    // ---- WG: weight / bias gradient partial sums of this block -> its slab ----
    if (WG && wg) {
        const int tw = wave % NTILE, part = wave / NTILE;
        const int tkf = tw / KH, tcf = tw % KH;
        float* s_w = reinterpret_cast<float*>(sS);        // [8 waves][16][64] (the staging tiles are free)
        if (NPART > 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s_w[(wave * 16 + e) * 64 + lane] = wacc[e];
            __syncthreads();
            if (part == 0) {
#pragma unroll
                for (int q = 1; q < NPART; ++q)
#pragma unroll
                    for (int e = 0; e < 16; ++e) wacc[e] += s_w[((tw + q * NTILE) * 16 + e) * 64 + lane];
            }
        }
        float* slab = a.wg_partial + (size_t)bi * a.wg_stride;
        if (part == 0) {
            const int cf = n0 + tcf * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kf = tkf * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (cf < K) slab[(size_t)kf * K + cf] = wacc[e];
            }
        }
        if (wg_bias) {
            // threads t, t + CPR, ... hold partial sums of the same 8 channels: fixed-order sum through the LDS
            __syncthreads();
            float* s_b = reinterpret_cast<float*>(sS);    // [512][8]
#pragma unroll
            for (int e = 0; e < 8; ++e) s_b[tid * 8 + e] = bs[e];
            __syncthreads();
            if (tid < C) {
                const int ch = tid >> 3, e = tid & 7;
                float tot = 0.f;
                for (int t = ch; t < 512; t += CPR) tot += s_b[t * 8 + e];
                slab[(size_t)C * K + tid] = tot;
            }
        }
        __syncthreads();                                  // staging region free for the statistics flush
    }

    // ---- statistics: one flush per block ----
    // Each lane holds fp32 partial sums {sum (v - c), sum (v - c)^2} (forward: c = the wave's common shift of that channel;
    // backward: {sum dz, sum dz * xhat}, c = 0) of 8 channels over its rows.  The 16 lanes of a channel chunk are combined by
    // a transposition through the wave's own staging tile (48 LDS reads per lane; a shuffle tree is 77 dependent
    // ds_bpermute round trips, 3 us at the end of every block), un-shifted once in fp64, then the waves of a channel half
    // are added in a fixed order and ONE exact pair of sums (integer limbs, common.h) per channel leaves the block.
    if (want_stats) {
        float* rec = reinterpret_cast<float*>(sS + wave * (32 * LDST));      // [64 lanes][17]: f1[8] f2[8] nrow
#pragma unroll
        for (int e = 0; e < 8; ++e) { rec[lane * 17 + e] = F1[e >> 1][e & 1]; rec[lane * 17 + 8 + e] = F2[e >> 1][e & 1]; }
        rec[lane * 17 + 16] = (float)nrow;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int cv4 = (lane >> 3) & 3, ce = lane & 7;                     // lanes 0..31: channel 8 cv4 + ce of the wave's 32
        float t1 = 0.f, t2 = 0.f, tn = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* src = rec + (r * 4 + cv4) * 17;
            t1 += src[ce]; t2 += src[8 + ce]; tn += src[16];
        }
        // common shift of this channel (every lane of the chunk holds the same one; lane cv4 is one of them)
        float csv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) csv[e] = BWD ? 0.f : CS[e >> 1][e & 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // all reads of `rec` done before it is reused
        float* shf = rec;                                                    // [4 chunks][8]
        if ((lane >> 2) == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) shf[(lane & 3) * 8 + e] = csv[e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const double c = (double)shf[cv4 * 8 + ce], n = (double)tn;
        const double s1 = (double)t1 + n * c;
        const double s2 = BWD ? (double)t2 : (double)t2 + 2.0 * c * (double)t1 + n * c * c;
        __syncthreads();                                                     // every wave is done with its staging tile
        double* s_red = reinterpret_cast<double*>(sS);                       // [8 waves][32][2]
        if (lane < 32) {
            s_red[(wave * 32 + lane) * 2 + 0] = s1;
            s_red[(wave * 32 + lane) * 2 + 1] = s2;
        }
        __syncthreads();
        fpd_stat_t* st = BWD ? a.epi_stats : a.out_stats;
        for (int t = tid; t < KP; t += 512) {
            if (n0 + t < K) {
                const int h2 = t >> 5, c32 = t & 31;      // the PXW waves of channel half h2 hold partial sums of channel t
                double u1 = 0.0, u2 = 0.0;
#pragma unroll
                for (int q = 0; q < PXW; ++q) {
                    u1 += s_red[((h2 * PXW + q) * 32 + c32) * 2];
                    u2 += s_red[((h2 * PXW + q) * 32 + c32) * 2 + 1];
                }
                stat_atomic_add(st, K, 0, n0 + t, u1);
                stat_atomic_add(st, K, 1, n0 + t, u2);
            }
        }
    }
#ifdef FPD_PP_TIMING
    PP_STAMP();
    __syncthreads();
    if (tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 1)) {
        // operand rows requested | weights requested | tables stored | weights stored | barrier | per tile: staged+requests issued, barrier, MFMAs, barrier, epilogue, barrier | flush
        printf("conv_pp R=%d C=%d K=%d blk %d tiles %d:", R, C, K, (int)blockIdx.x, t_end - t_beg);
        for (int q = 1; q < 100 && s_stamp[q] != 0; ++q) printf(" %lld", s_stamp[q] - s_stamp[0]);
        printf("\n");
    }
#endif
}

// One or two INDEPENDENT convolutions of the same template configuration (the up-/low-branch Bottleneck convolutions of an
// hourglass level) in one launch.  A convolution owns nblk tile ranges x ks channel slabs = nblk * ks blocks ("units"); the
// units of `b` are spread evenly over the grid (Bresenham), so that whatever subset of the grid is resident first serves both
// in proportion.  gb.nblk == 0: single convolution.  Unit u -> (range, slab): the slabs of one range are 8 block ids apart,
// i.e. on the same XCD.
struct PPArgs { fpd_conv_t c[2]; PPGeo g[2]; int ks; };

// Forward kernels fit 128 registers per lane (two blocks per CU); the BN-backward epilogue (epi_x prefetch, mask, two more sums)
// does not without spilling the operand prefetch to scratch, so those kernels take the 256-register budget and one block per CU.
template <int R, int C, int KH, bool BWD, bool WG>
__global__ __launch_bounds__(512, BWD ? 2 : 4) void conv_pp_kernel(const PPArgs p) {
    const int bid = blockIdx.x, n = gridDim.x, ks = p.ks, nb = p.g[1].nblk * ks;
    const int fb0 = fpd_cut(bid, nb, n), fb1 = fpd_cut(bid + 1, nb, n);
    const int isb = fb1 > fb0 ? 1 : 0;                    // (the descriptor is indexed, not branched on: ONE copy of the body)
    const int u = isb ? fb0 : bid - fb0;
    const int nr = p.g[isb].nblk;
    int range, slab;
    if (ks == 2 && (nr & 7) == 0) { slab = (u >> 3) & 1; range = (u & 7) + 8 * (u >> 4); }
    else { slab = u % ks; range = u / ks; }
    conv_pp_body<R, C, KH, BWD, WG>(p.c[isb], p.g[isb], range, slab * 64);
}

// ---------------------------------------------------------------------------------------------------------------------
constexpr size_t PP_LDS_MAX = 160 * 1024;

// FPD_CONV_PP: 0 = never, 1 = launches with >= FPD_CONV_PP_MIN_TILES (256) pixel tiles (default), 2 = whenever the shape is in
// the domain; FPD_CONV_PP_BLOCKS: persistent blocks per occupancy slot (default 256 (128 until the teacher wait moved in front of the loss): the grid is that times the blocks a CU can
// hold, at most 2 -- r03 sweep inside the pipelined step, one box: 64/96/128/256 x 2 -> 12.38/11.41/11.00/11.01 ms, 128/192/256 x 1
// -> 11.35/11.12/11.03; threshold 512/256/128 tiles -> 11.00/10.79/10.78; conv_tile only: 11.50).  mode / blocks can be changed
// at run time through fpd_set_option("conv_pp" / "conv_pp_blocks", v) (tests drive small shapes through the kernel that way).
static int g_pp_mode = -1, g_pp_blocks = -1;
static int pp_mode() {
    if (g_pp_mode < 0) { const char* e = getenv("FPD_CONV_PP"); g_pp_mode = e ? atoi(e) : 1; }
    return g_pp_mode;
}
static int pp_blocks() {
    if (g_pp_blocks < 0) { const char* e = getenv("FPD_CONV_PP_BLOCKS"); g_pp_blocks = e ? atoi(e) : 256; }
    return g_pp_blocks < 1 ? 1 : g_pp_blocks;
}
static int pp_min_tiles() {  // FPD_CONV_PP_MIN_TILES: smallest launch (pixel tiles) the kernel takes in mode 1
    static int v = -1;
    if (v < 0) { const char* e = getenv("FPD_CONV_PP_MIN_TILES"); v = e ? atoi(e) : 256; }
    return v;
}
static int pp_occ_cap() { return 2; }        // resident blocks per CU the forward grids are sized for (FPD_CONV_PP_OCC of rounds 3-5)
static int pp_blocks_bwd() { return pp_blocks(); }      // data-gradient kernels: one block per CU, same count (128 / 192 / 256 / 384 -> 10.33 / 10.22 / 10.08 / 10.42 ms, round 4)
static int pp_fuse_wgrad() { return 1; }      // 1x1 data gradients also form their weight gradient

static int pp_tile_px(const fpd_conv_t& a) { return a.K > 32 ? 128 : 256; }
static int pp_nrows(const fpd_conv_t& a) { return std::max(1, pp_tile_px(a) / a.W); }

static bool pp_domain(const fpd_conv_t& a) {
    if (a.dtype != FPD_BF16 || a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return false;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < 2) return false;
    if (a.C != 16 && a.C != 32 && a.C != 64 && a.C != 128) return false;
    if (a.K > 128 || a.K % 8 != 0) return false;
    if (a.R == 3 && a.C > 64) return false;                                // all nine weight tiles of a slab must fit the LDS
    if ((pp_nrows(a) + a.R - 1) * a.W * (a.C / 8) > 2048) return false;    // 4 operand vectors per thread
    return true;
}
static int pp_tiles(const fpd_conv_t& a) { return cdiv(a.N * a.H, pp_nrows(a)); }
// shapes whose data-gradient launch can carry the forward convolution's weight gradient (conv_pp_body<.., WG = true>)
static int pp_wg_kmax() { return 128; }       // widest data gradient (output channels) that also forms the weight gradient
static int pp_wg_ckmax() { return 128 * 128; }      // largest weight matrix (C x K elements) formed inside a data-gradient launch
static bool pp_wg_shape(const fpd_conv_t& a) {
    return pp_fuse_wgrad() != 0 && pp_domain(a) && a.epi == FPD_EPI_BNRELU_BWD && a.R == 1 && a.C >= 32 && a.K <= pp_wg_kmax() && a.C * a.K <= pp_wg_ckmax() &&
           (pp_nrows(a) * a.W) % 16 == 0;
}

static PPGeo pp_geo(const fpd_conv_t& a) {
    const int LDST = 32 * 4 + 16, LDA = a.C * 2 + 16;
    PPGeo g;
    g.nrows = pp_nrows(a);
    g.ntiles = pp_tiles(a);
    g.nblk = 0;
    const int hrows = g.nrows + a.R - 1, WP = a.W + a.R - 1;
    g.region = std::max((hrows * WP + 4) * LDA, 8 * 32 * LDST);
    g.mW = pp_magic(a.W);
    g.mWV = pp_magic(a.W * (a.C / 8));
    g.mH = pp_magic(a.H);
    return g;
}

// launch geometry of one convolution (b == nullptr) or a pair: ONE place decides it, for the launch and for the slab-count query
struct PPPlan { PPGeo ga, gb; int ks, grid; size_t lds; bool wg; };
static bool pp_plan(const fpd_conv_t& a, const fpd_conv_t* b, bool want_wg, PPPlan& pl) {
    const int KH = a.K > 32 ? 2 : 1;
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD;
    pl.ga = pp_geo(a);
    pl.gb = pl.ga;
    pl.gb.nblk = 0;
    int total = pl.ga.ntiles;
    if (b != nullptr) {
        pl.gb = pp_geo(*b);
        total += pl.gb.ntiles;
    }
    pl.wg = want_wg && pp_wg_shape(a) && (b == nullptr || pp_wg_shape(*b));
    int region = std::max(pl.ga.region, b ? pl.gb.region : 0);
    if (pl.wg) {                                           // the image is not aliased: only the image counts
        const int LDA = a.C * 2 + 16;
        auto img = [&](const fpd_conv_t& c, const PPGeo& g) { return (g.nrows * c.W + 4) * LDA; };
        region = std::max(img(a, pl.ga), b ? img(*b, pl.gb) : 0);
    }
    pl.ga.region = pl.gb.region = region;
    pl.lds = (size_t)(2 * a.C + 5 * 32 * KH + (bwd ? 3 * a.C : 0)) * sizeof(float) + (size_t)a.R * a.R * 32 * KH * a.C * 2 + (size_t)region;
    if (pl.wg) pl.lds += (size_t)8 * 32 * (32 * 4 + 16) + (size_t)(256 / KH) * (32 * KH * 2 + 16);
#ifdef FPD_PP_TIMING
    pl.lds += 1024;
#endif
    if (pl.lds > PP_LDS_MAX) return false;
    const int occ = bwd ? 1 : std::max(1, std::min(pp_occ_cap(), (int)(PP_LDS_MAX / pl.lds)));
    pl.ks = cdiv(a.K, 64);                                 // channel slabs of <= 64 (K = 128: two blocks per tile range)
    // tile ranges: ks blocks per range, every block should own at least two tiles
    int ranges = std::max(1, std::min((bwd ? pp_blocks_bwd() : pp_blocks()) * occ / pl.ks, total / 2));
    if (b != nullptr) {
        if (ranges < 2) return false;
        pl.gb.nblk = std::max(1, std::min(ranges - 1, (int)((long long)ranges * pl.gb.ntiles / total)));
    }
    pl.ga.nblk = ranges - pl.gb.nblk;
    if (pl.ks == 2 && pl.ga.nblk >= 8) pl.ga.nblk &= ~7;   // (the XCD pairing of the slabs wants multiples of 8)
    if (pl.ks == 2 && pl.gb.nblk >= 8) pl.gb.nblk &= ~7;
    pl.grid = (pl.ga.nblk + pl.gb.nblk) * pl.ks;
    return true;
}

template <int R, int C, int KH, bool BWD, bool WG>
static int pp_launch_t(const fpd_conv_t& a, const fpd_conv_t* b, const PPPlan& pl, hipStream_t st) {
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&conv_pp_kernel<R, C, KH, BWD, WG>), pl.lds)) return rc_;
    PPArgs args;
    args.c[0] = a; args.c[1] = b ? *b : a; args.g[0] = pl.ga; args.g[1] = pl.gb; args.ks = pl.ks;
    FPD_LAUNCH((conv_pp_kernel<R, C, KH, BWD, WG>), dim3(pl.grid), dim3(512), pl.lds, st, args);
    return 0;
}

template <int R, int C>
static int pp_launch_c(const fpd_conv_t& a, const fpd_conv_t* b, const PPPlan& pl, hipStream_t st) {
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD;
    if constexpr (R == 1 && C >= 32) {
        if (pl.wg) return a.K > 32 ? pp_launch_t<R, C, 2, true, true>(a, b, pl, st) : pp_launch_t<R, C, 1, true, true>(a, b, pl, st);
    }
    if (a.K > 32) return bwd ? pp_launch_t<R, C, 2, true, false>(a, b, pl, st) : pp_launch_t<R, C, 2, false, false>(a, b, pl, st);
    return bwd ? pp_launch_t<R, C, 1, true, false>(a, b, pl, st) : pp_launch_t<R, C, 1, false, false>(a, b, pl, st);
}

static int pp_launch(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    const bool want_wg = a.wg_partial != nullptr || (b != nullptr && b->wg_partial != nullptr);
    PPPlan pl;
    if (!pp_plan(a, b, want_wg, pl)) return 1;
    if (want_wg) {
        if (!pl.wg) return fpd_fail(-2, "conv: a fused weight gradient was requested for a launch fpd_conv_fused_wgrad_partials() reports 0 for");
        const fpd_conv_t* cs[2] = {&a, b};
        const int nblk[2] = {pl.ga.nblk, pl.gb.nblk};
        for (int i = 0; i < 2; ++i) {
            if (cs[i] == nullptr || cs[i]->wg_partial == nullptr) continue;
            if (cs[i]->wg_stride < (int64_t)cs[i]->C * cs[i]->K + cs[i]->C)
                return fpd_fail(-2, "conv: wg_stride %lld smaller than weight + bias", (long long)cs[i]->wg_stride);
            // the slab count was asked for when the workspace was sized; a geometry that has changed since (conv_pp_blocks) would
            // write past the workspace or leave slabs unwritten
            if (cs[i]->wg_count != nblk[i])
                return fpd_fail(-2, "conv: the launch writes %d weight-gradient slabs but the caller sized its workspace for %d "
                                    "(fpd_conv_fused_wgrad_partials: has a conv_pp option changed since?)", nblk[i], cs[i]->wg_count);
        }
    }
    if (a.R == 3) {
        if (a.C == 64) return pp_launch_c<3, 64>(a, b, pl, st);
        if (a.C == 32) return pp_launch_c<3, 32>(a, b, pl, st);
        return pp_launch_c<3, 16>(a, b, pl, st);
    }
    switch (a.C) {
        case 16: return pp_launch_c<1, 16>(a, b, pl, st);
        case 32: return pp_launch_c<1, 32>(a, b, pl, st);
        case 64: return pp_launch_c<1, 64>(a, b, pl, st);
        default: return pp_launch_c<1, 128>(a, b, pl, st);
    }
}

static bool pp_takes(const fpd_conv_t& a, const fpd_conv_t* b) {
    const int mode = pp_mode();
    if (mode == 0 || !pp_domain(a)) return false;
    int tiles = pp_tiles(a);
    if (b != nullptr) {
        if (!pp_domain(*b) || a.K != b->K || a.C != b->C || a.R != b->R || a.epi != b->epi) return false;
        tiles += pp_tiles(*b);
    }
    return mode != 1 || tiles >= pp_min_tiles();
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
    if (!pp_takes(a, nullptr)) return 1;
    return pp_launch(a, nullptr, st);
}

// two independent convolutions of equal channel shapes in one launch; 1 = not pairable here
int fpd_conv_pp_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    if (!pp_takes(a, &b)) return 1;
    return pp_launch(a, &b, st);
}

// slabs of the fused weight gradient (fpd_conv_t.wg_partial) for a single launch / the two halves of a pair launch; 0 = the
// launch would not run here, or not with the fusion
// 1 if the launch (pair) is served by this kernel as a BNRELU_BWD data gradient without a prologue BN: the configuration in which
// a folded BN-backward apply (fpd_conv_t.fold_x) is evaluated on the way into the operand image
int fpd_conv_pp_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b) {
    if (a.epi != FPD_EPI_BNRELU_BWD || a.bn.mode != FPD_BN_NONE) return 0;
    if (b != nullptr && (b->epi != FPD_EPI_BNRELU_BWD || b->bn.mode != FPD_BN_NONE)) return 0;
    if (!pp_takes(a, b)) return 0;
    // exactly what pp_launch() will decide: the launch may still be declined by its geometry (LDS, a pair with < 2 ranges),
    // with and without the fused weight gradient
    PPPlan pl;
    if (!pp_plan(a, b, false, pl)) return 0;
    if (pp_wg_shape(a) && (b == nullptr || pp_wg_shape(*b)) && !pp_plan(a, b, true, pl)) return 0;
    return 1;
}

int fpd_conv_pp_wgrad_partials(const fpd_conv_t& a) {
    PPPlan pl;
    if (!pp_takes(a, nullptr) || !pp_plan(a, nullptr, true, pl) || !pl.wg) return 0;
    return pl.ga.nblk;
}
int fpd_conv_pp_pair_wgrad_partials(const fpd_conv_t& a, const fpd_conv_t& b, int* na, int* nb) {
    PPPlan pl;
    *na = *nb = 0;
    if (!pp_takes(a, &b) || !pp_plan(a, &b, true, pl) || !pl.wg) return 0;
    *na = pl.ga.nblk;
    *nb = pl.gb.nblk;
    return 0;
}
