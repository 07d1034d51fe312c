// conv_c3: the student's 3x3 "same" convolution at C = K = 64 (conv2 of every Bottleneck; bf16, stride 1, 16..64-wide maps whose
// rows tile 256-pixel strips): forward (BatchNorm+ReLU prologue, bias, batch statistics of the result) and the
// BatchNorm-backward data gradient (ReLU mask of epi_x + the two sums), optionally evaluating a folded BN-backward apply on
// its operand (fold_x) and writing the evaluated operand out once for the separate weight-gradient launch (fold_out).
// Same contract as conv_pp / conv_tile (fpd_conv_t); round 6, after the r06 what-if put the >= 32-high 3x3 convolutions at 1.4 ms
// of the 9.2 ms step -- the largest item of the student chain once the 1x1 convolutions had moved to conv_c1.
//
// The epilogue is conv_c1's: the transposed MFMA leaves a lane with 4 consecutive channels of one pixel, ONE rounding into a
// wave-private bf16 output tile, read back as whole vectors of 8 channels (a lane always holds the same channels: epilogue
// tables, mask, sums in registers) and stored as 1 KB lines -- no fp32 round trip through the LDS, no barrier around it.
// What a 3x3 filter adds is the halo, i.e. an operand image the waves SHARE:
//   * a block (8 waves) walks 256-pixel STRIPS (256 / W whole image rows); the strip's rows + one above and below (a contiguous
//     run of the NHWC tensor: whole 1 KB lines, branch-free, rows outside the image become zeros) are brought by all hands,
//     the prologue (BN+ReLU or the folded apply; the thread's 8 channels are fixed) applied once per element on the way
//     into an UNPADDED, XOR-swizzled LDS image with zero border columns; the next strip is requested as soon as the registers
//     are free (in flight during the products and the epilogue);
//   * all nine weight taps (72 KB) sit in the LDS once per block; a wave multiplies ITS 32 pixels x 64 channels: per tap and
//     16-channel step one operand fragment (address = a per-lane constant of the tap ^ the step) and two weight fragments
//     (per-lane constants + immediates) for two MFMAs;
//   * two barriers per 256 pixels (image complete / image free) where conv_pp has three per 128.
// LDS: 2.5 KB tables + 72 KB weights + 50.7 KB image (W = 64) + 8 x 4 KB output tiles = 157 KB: one block per CU.
//
// Replaces the same reference calls as conv_pp (nn.Conv2d 3x3 + BatchNorm2d + ReLU and their autograd,
// /root/reference/lib/models/hourglass.py:22-23, 32-52).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr int C3_NW = 8, C3_C = 64, C3_STRIP = 256;       // waves per block, channels in = out, pixels per strip
constexpr int C3_CV = C3_C / 8;                           // 16-byte chunks per pixel
constexpr int C3_NVH = 6;                                 // operand vectors per thread and strip, at most ((256 + 2 W) * 8 / 512 at W = 64)
constexpr int C3_NK = C3_C / 16;                          // output vectors per lane and tile
constexpr int C3_WBYTES = 9 * C3_C * C3_C * 2;            // all nine taps
constexpr int C3_TILE_O = 32 * C3_C * 2;                  // a wave's output tile [32 px][128 B], 16-byte chunk c of pixel p at c ^ sw(p)
constexpr int C3_TABLES = (2 * C3_C + 3 * C3_C + 4 * C3_C + C3_C) * 4;
constexpr int C3_FLUSH = C3_NW * 64 * 16 * 4 + C3_NW * C3_C * 4 + 8 * 2 * C3_C * 8;

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x2 c3_unpack(unsigned w) {
    f32x2 r = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    return r;
}
__device__ __forceinline__ unsigned c3_pack(f32x2 v) { return f2bf_pk(v[0], v[1]); }
__device__ __forceinline__ unsigned c3_floor(unsigned w, short lo) {      // lo = 0: ReLU on a packed bf16 pair, -32768: identity (conv_pp.hip)
    s16x2 a = *reinterpret_cast<const s16x2*>(&w);
    const s16x2 b = {lo, lo};
    a = __builtin_elementwise_max(a, b);
    return *reinterpret_cast<const unsigned*>(&a);
}
// swizzle of the 16-byte chunks of pixel p (128-byte pixels, two per 256-byte bank row): 16 consecutive pixels put one chunk
// column on 16 different 16-byte slots -- conflict-free b128 reads without padding
__device__ __forceinline__ int c3_sw(int p) { return (p >> 1) & 7; }

#define C3_PHASE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

struct C3Geo { int lgW, rs, nstrip, nblk, img_bytes; };   // log2 W, image rows per strip, strips of the convolution, blocks working on it

template <bool BWD, bool FOLD>
__device__ __forceinline__ void c3_body(const fpd_conv_t& a, const C3Geo g, const int bi) {
    constexpr int C = C3_C, K = C3_C, CV = C3_CV, NK = C3_NK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, H = a.H, lgW = g.lgW, RS = g.rs, WP = W + 2;
    const int M = a.N * H * W;
    const int s_beg = fpd_cut(bi, g.nstrip, g.nblk), s_end = fpd_cut(bi + 1, g.nstrip, g.nblk);
    const int spi = H / RS;                               // strips per image

    float* s_scale = reinterpret_cast<float*>(smem);      // [C] prologue BN
    float* s_shift = s_scale + C;
    float* s_fold = s_shift + C;                          // [3][C] coefficients of a folded BN-backward apply
    float* s_epi = s_fold + 3 * C;                        // [4][K] scale, shift, mean, invstd of epi_bn
    float* s_bias = s_epi + 4 * K;                        // [K]
    unsigned char* sW = reinterpret_cast<unsigned char*>(s_bias + K);     // [9][K][C] bf16: row tap * 64 + k, chunk c at c ^ ((k >> 1) & 7)
    unsigned char* sI = sW + C3_WBYTES;                   // operand image [(RS + 2) * WP px][128 B], swizzled; border columns zero
    unsigned char* sO = sI + g.img_bytes;                 // the waves' output tiles
    unsigned char* tO = sO + wave * C3_TILE_O;

    const bf16_t* __restrict__ x = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* __restrict__ w = reinterpret_cast<const bf16_t*>(a.w);
    bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(a.y);
    const bf16_t* __restrict__ ex = reinterpret_cast<const bf16_t*>(a.epi_x);
    const bf16_t* __restrict__ fx = reinterpret_cast<const bf16_t*>(a.fold_x != nullptr ? a.fold_x : a.x);
    const bool fold = FOLD && a.fold_x != nullptr;        // (FOLD: some convolution of the launch folds; this one may not)
    bf16_t* fo = fold ? reinterpret_cast<bf16_t*>(a.fold_out) : nullptr;
    const bool has_bn = !BWD && a.bn.mode != FPD_BN_NONE;
    const bool want_stats = BWD || a.out_stats != nullptr;

    // ---- prologue: table chains requested first, then the weights, then the first strip (conv_c1.hip) ----
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

    // ---- the strip's operand vectors: vector v = tid + 512 i of its (RS + 2) rows x W pixels x 8 chunks is 16 bytes at
    //      (first pixel of the row above the strip) * 128 + v * 16 -- one contiguous run of the tensor.  A thread always holds
    //      chunk tid & 7.  Rows outside the image: clamped address, the value is replaced by zeros when it is staged. ----
    const int nvtot = (C3_STRIP + 2 * W) * CV;
    const int nvh = (nvtot + 511) >> 9;                   // uniform trip count (6 / 5 / 5 at W = 64 / 32 / 16)
    uint4 rx[C3_NVH];
    uint4 ru[FOLD ? C3_NVH : 1];
    uint4 rr[NK];                                         // BWD: epi_x of the wave's tile
    const long long xlast = (long long)M * CV - 1;        // last vector of the tensor
    auto load_x = [&](const int strip) {
        const int n = strip / spi, y0 = (strip - n * spi) * RS;
        const long long v0 = ((long long)(n * H + y0 - 1) << lgW) * CV + tid;      // (may be negative / past the end: clamped)
#pragma unroll
        for (int i = 0; i < C3_NVH; ++i) {
            if (i < nvh) {
                long long v = v0 + i * 512;
                v = v < 0 ? 0 : (v > xlast ? xlast : v);
                rx[i] = reinterpret_cast<const uint4*>(x)[v];
                if constexpr (FOLD) ru[i] = reinterpret_cast<const uint4*>(fx)[v];
            }
        }
    };
    auto load_r = [&](const int strip) {
        const uint4* pr = reinterpret_cast<const uint4*>(ex) + ((size_t)(strip * C3_NW + wave) * 32 * CV + lane);
#pragma unroll
        for (int i = 0; i < NK; ++i) rr[i] = pr[i * 64];
    };
    {
        constexpr int NWV = 9 * K * CV / 512;             // 9 weight vectors per thread
        u32x4 rw[NWV];                                    // (a native vector type: an array of HIP's uint4 structs copied whole goes through scratch)
#pragma unroll
        for (int i = 0; i < NWV; ++i) rw[i] = reinterpret_cast<const u32x4*>(w)[tid + i * 512];
        __builtin_amdgcn_sched_barrier(0);
        {
            const int st0 = min(s_beg, g.nstrip - 1);     // (unconditionally: counted waits in front of the first barrier)
            load_x(st0);
            if constexpr (BWD) load_r(st0);
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
            if (bi == 0) {
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
        // weights: global [k][tap][c] -> LDS row tap * 64 + k
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int v = tid + i * 512;
            const int k = v / (9 * CV), rem = v - k * (9 * CV);
            const int tap = rem >> 3, ch = rem & 7;
            *reinterpret_cast<u32x4*>(sW + ((tap * K + k) * CV + (ch ^ ((k >> 1) & 7))) * 16) = rw[i];
        }
        // zero border columns of the image (never written again)
        for (int r = tid; r < (RS + 2) * 2 * CV; r += 512) {
            const int row = r / (2 * CV), side = (r / CV) & 1, ch = r & 7;
            const int p = row * WP + (side ? WP - 1 : 0);
            *reinterpret_cast<uint4*>(sI + p * 128 + ch * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();                                      // tables + weights visible

    // ---- per-lane constants ----
    const int cch = tid & 7;                              // the 16-byte chunk (8 channels) this thread stages
    const int kch = lane & 7, pxo0 = lane >> 3;           // output vectors: chunk, first pixel (vector i: + 8 i)
    f32x2 p_sc[4], p_sh[4];
    if (has_bn) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p_sc[e] = *reinterpret_cast<const f32x2*>(s_scale + cch * 8 + 2 * e);
            p_sh[e] = *reinterpret_cast<const f32x2*>(s_shift + cch * 8 + 2 * e);
        }
    }
    const short relu_floor = a.bn.relu ? (short)0 : (short)-32768;
    const float relu_gate = a.epi_bn.relu ? 0.f : -3.4e38f;
    f32x2 F1[4], F2[4], CS[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { F1[e] = f32x2{0.f, 0.f}; F2[e] = f32x2{0.f, 0.f}; CS[e] = f32x2{0.f, 0.f}; }
    // operand fragment of tap (r, s), step kk: pixel (ry + r) * WP + x + s of the image, chunk (2 kk + hh) ^ sw(pixel)
    //   = pre[tap] ^ (kk * 32), pre = pixel * 128 | ((hh ^ sw) * 16)   (the image is 128-byte aligned in the LDS)
    const int q = wave * 32 + l31, ry = q >> lgW, xx = q & (W - 1);
    int pre[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int p = (ry + t / 3) * WP + xx + t % 3;
        pre[t] = p * 128 + ((hh ^ c3_sw(p)) * 16);
    }
    // weight fragment of (tap, kt, kk): row tap * 64 + kt * 32 + l31, chunk (2 kk + hh) ^ ((l31 >> 1) & 7): a per-lane constant
    // per kk + the immediate (tap * 64 + kt * 32) * 128
    int wpre[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wpre[kk] = l31 * 128 + (((2 * kk + hh) ^ ((l31 >> 1) & 7)) * 16);
    const int osw = c3_sw(l31);                           // swizzle of this lane's pixel row in the output tile

    // ---- 1. the strip's operand: prologue on the way into the image ----
    auto stage = [&](const int strip, auto modec) {
        constexpr int MODE = decltype(modec)::value;       // 0 = raw operand, 1 = BatchNorm(+ReLU) prologue, 2 = folded BN-backward apply
        f32x2 f_a[4], f_b[4], f_d[4];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f_a[e] = *reinterpret_cast<const f32x2*>(s_fold + cch * 8 + 2 * e);
                f_b[e] = *reinterpret_cast<const f32x2*>(s_fold + C + cch * 8 + 2 * e);
                f_d[e] = *reinterpret_cast<const f32x2*>(s_fold + 2 * C + cch * 8 + 2 * e);
            }
        }
        const int n = strip / spi, y0 = (strip - n * spi) * RS;
        const long long v0 = ((long long)(n * H + y0 - 1) << lgW) * CV;
#pragma unroll
        for (int i = 0; i < C3_NVH; ++i) {
            const int v = tid + i * 512;
            if (v < nvtot) {
                const int pxl = v >> 3, row = pxl >> lgW, col = pxl & (W - 1);
                const int gy = y0 - 1 + row;
                const bool in = gy >= 0 && gy < H;
                uint4 val = rx[i];
                if constexpr (MODE == 2) {
                    const uint4 uu = ru[i];
                    const unsigned gw[4] = {val.x, val.y, val.z, val.w}, uw[4] = {uu.x, uu.y, uu.z, uu.w};
                    unsigned ow[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f32x2 tt = __builtin_elementwise_fma(f_b[e], c3_unpack(uw[e]), f_d[e]);
                        ow[e] = c3_pack(__builtin_elementwise_fma(f_a[e], c3_unpack(gw[e]), tt));
                    }
                    val = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                    // the evaluated operand is written out once (rows of the strip proper: halo rows belong to the neighbours)
                    if (fo != nullptr && row >= 1 && row <= RS) reinterpret_cast<uint4*>(fo)[v0 + v] = val;
                } else if constexpr (MODE == 1) {
                    val.x = c3_floor(c3_pack(__builtin_elementwise_fma(c3_unpack(val.x), p_sc[0], p_sh[0])), relu_floor);
                    val.y = c3_floor(c3_pack(__builtin_elementwise_fma(c3_unpack(val.y), p_sc[1], p_sh[1])), relu_floor);
                    val.z = c3_floor(c3_pack(__builtin_elementwise_fma(c3_unpack(val.z), p_sc[2], p_sh[2])), relu_floor);
                    val.w = c3_floor(c3_pack(__builtin_elementwise_fma(c3_unpack(val.w), p_sc[3], p_sh[3])), relu_floor);
                }
                // rows outside the image are exactly zero (the filter pads the ACTIVATION)
                val.x = in ? val.x : 0u; val.y = in ? val.y : 0u; val.z = in ? val.z : 0u; val.w = in ? val.w : 0u;
                const int p = row * WP + col + 1;
                *reinterpret_cast<uint4*>(sI + p * 128 + ((cch ^ c3_sw(p)) * 16)) = val;
            }
        }
    };
    // ---- 3. (+ bias) ONE rounding into the wave's bf16 output tile: lane = pixel l31, channels kt * 32 + 8 q + 4 hh .. + 3 ----
    auto finish = [&](const f32x16& accv, const int kt) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            unsigned char* p = tO + l31 * 128 + (((kt * 4 + qq) ^ osw) * 16) + hh * 8;
            f32x2 v0 = {accv[4 * qq], accv[4 * qq + 1]}, v1 = {accv[4 * qq + 2], accv[4 * qq + 3]};
            if constexpr (!BWD) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_bias + kt * 32 + 8 * qq + 4 * hh);
                v0 += f32x2{b4[0], b4[1]};
                v1 += f32x2{b4[2], b4[3]};
            }
            *reinterpret_cast<uint2*>(p) = make_uint2(c3_pack(v0), c3_pack(v1));
        }
    };
    // ---- 4. per channel: whole vectors of 8 channels (vector i: pixel pxo0 + 8 i of the tile, chunk kch) ----
    auto perchan = [&](const int tile, const bool first, auto statc) {
        constexpr bool STATS = decltype(statc)::value;
        uint4* py = reinterpret_cast<uint4*>(y) + ((size_t)tile * 32 * CV + lane);
        f32x2 e_sc[4], e_sh[4], e_mu[4], e_is[4];
        if constexpr (BWD) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                e_sc[e] = *reinterpret_cast<const f32x2*>(s_epi + kch * 8 + 2 * e);
                e_sh[e] = *reinterpret_cast<const f32x2*>(s_epi + K + kch * 8 + 2 * e);
                e_mu[e] = *reinterpret_cast<const f32x2*>(s_epi + 2 * K + kch * 8 + 2 * e);
                e_is[e] = *reinterpret_cast<const f32x2*>(s_epi + 3 * K + kch * 8 + 2 * e);
            }
        }
        if (!BWD && STATS && first) {                      // common shift of the wave's shifted sums: its first pixel (sw(0) = 0)
            const uint4 c4 = *reinterpret_cast<const uint4*>(tO + kch * 16);
            CS[0] = c3_unpack(c4.x); CS[1] = c3_unpack(c4.y); CS[2] = c3_unpack(c4.z); CS[3] = c3_unpack(c4.w);
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) {
            const int px = pxo0 + 8 * i;
            const uint4 o4 = *reinterpret_cast<const uint4*>(tO + px * 128 + ((kch ^ c3_sw(px)) * 16));
            unsigned ow[4] = {o4.x, o4.y, o4.z, o4.w};
            if constexpr (BWD) {
                const unsigned xw[4] = {rr[i].x, rr[i].y, rr[i].z, rr[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 xv = c3_unpack(xw[e]);
                    const f32x2 z = __builtin_elementwise_fma(xv, e_sc[e], e_sh[e]);
                    const unsigned keep = (z[0] > relu_gate ? 0x0000ffffu : 0u) | (z[1] > relu_gate ? 0xffff0000u : 0u);
                    ow[e] &= keep;                         // (the ReLU mask commutes with the rounding)
                    const f32x2 gq = c3_unpack(ow[e]);
                    F1[e] += gq;
                    F2[e] = __builtin_elementwise_fma(gq, (xv - e_mu[e]) * e_is[e], F2[e]);
                }
            } else if constexpr (STATS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x2 d = c3_unpack(ow[e]) - CS[e];
                    F1[e] += d;
                    F2[e] = __builtin_elementwise_fma(d, d, F2[e]);
                }
            }
            py[i * 64] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
            if constexpr (BWD || STATS) {                  // (conv_c1.hip: the sums are formed here, one vector at a time)
#pragma unroll
                for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(F1[e]), "+v"(F2[e]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // =========================== the strip loop ===========================
    bool first = true;
    for (int strip = s_beg; strip < s_end; ++strip) {
        const bool pf = strip + 1 < s_end;
        if (FOLD && fold) { if constexpr (FOLD) stage(strip, std::integral_constant<int, 2>{}); }
        else if (has_bn) { if constexpr (!BWD) stage(strip, std::integral_constant<int, 1>{}); }
        else stage(strip, std::integral_constant<int, 0>{});
        if (pf) load_x(strip + 1);                        // in flight during the products and the epilogue
        __syncthreads();                                  // image complete
        // ---- 2. the products: 9 taps x 4 steps, one operand fragment and two weight fragments for two MFMAs ----
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(sI + (pre[t] ^ (kk * 32)));
                const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(sW + wpre[kk] + (t * 64) * 128);
                const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(sW + wpre[kk] + (t * 64 + 32) * 128);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf, acc1, 0, 0, 0);
            }
        }
        __syncthreads();                                  // every wave is done with the image: the next strip may be staged
        finish(acc0, 0);
        finish(acc1, 1);
        C3_PHASE();
        const int tile = strip * C3_NW + wave;
        if (want_stats) perchan(tile, first, std::true_type{}); else perchan(tile, first, std::false_type{});
        if (BWD && pf) load_r(strip + 1);
        C3_PHASE();
        first = false;
    }

    // ---- statistics: one flush per block (conv_c1.hip) ----
    if (want_stats) {
        __syncthreads();
        float* rec = reinterpret_cast<float*>(sI);        // [8 waves][64 lanes][16]
        float* shf = rec + C3_NW * 64 * 16;               // [8 waves][K]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            *reinterpret_cast<f32x2*>(rec + (wave * 64 + lane) * 16 + 2 * e) = F1[e];
            *reinterpret_cast<f32x2*>(rec + (wave * 64 + lane) * 16 + 8 + 2 * e) = F2[e];
        }
        if (!BWD && lane < CV) {
#pragma unroll
            for (int e = 0; e < 4; ++e) *reinterpret_cast<f32x2*>(shf + wave * K + lane * 8 + 2 * e) = CS[e];
        }
        __syncthreads();
        fpd_stat_t* st = BWD ? a.epi_stats : a.out_stats;
        constexpr int NPART = 4, NPW = C3_NW / NPART;     // 512 threads = 4 parts x 2 sums x 64 channels
        double* s_part = reinterpret_cast<double*>(shf + C3_NW * K);
        {
            const int part = tid / (2 * K), rem = tid % (2 * K);
            const int ch = rem % K, which = rem / K;
            const int chunk = ch >> 3, e = ch & 7;
            double tot = 0.0;
#pragma unroll
            for (int qv = 0; qv < NPW; ++qv) {
                const int wv = part * NPW + qv;
                double t1 = 0.0, t2 = 0.0;
#pragma unroll
                for (int jj = 0; jj < 64 / CV; ++jj) {
                    const float* rp = rec + (wv * 64 + chunk + CV * jj) * 16;
                    t1 += (double)rp[e];
                    t2 += (double)rp[8 + e];
                }
                if (BWD) tot += which ? t2 : t1;
                else {
                    const double c = (double)shf[wv * K + ch], nn = 32.0 * (s_end - s_beg);      // every wave has a tile in every strip
                    tot += which ? (t2 + 2.0 * c * t1 + nn * c * c) : (t1 + nn * c);
                }
            }
            s_part[part * 2 * K + rem] = tot;
        }
        __syncthreads();
        if (tid < 2 * K) {
            double tot = 0.0;
#pragma unroll
            for (int qv = 0; qv < NPART; ++qv) tot += s_part[qv * 2 * K + tid];
            stat_atomic_add(st, K, tid / K, tid % K, tot);
        }
    }
}

// One or two INDEPENDENT convolutions (the up- / low-branch Bottleneck convolutions of an hourglass level) in one launch: the
// blocks of descriptor 1 are spread evenly over the grid (Bresenham), as in conv_c1 / conv_pp.
struct C3Args { fpd_conv_t c[2]; C3Geo g[2]; };

template <bool BWD, bool FOLD>
__global__ __launch_bounds__(512, 2) void c3_kernel(const C3Args p) {
    const int bid = blockIdx.x, n = gridDim.x, nb = p.g[1].nblk;
    const int fb0 = fpd_cut(bid, nb, n), fb1 = fpd_cut(bid + 1, nb, n);
    const int isb = fb1 > fb0 ? 1 : 0;
    const int u = isb ? fb0 : bid - fb0;
    c3_body<BWD, FOLD>(p.c[isb], p.g[isb], u);
}

// ---------------------------------------------------------------------------------------------------------------------
// FPD_C3: 0 = never, 1 = launches of >= FPD_C3_MIN_PX pixels (default), 2 = whenever the shape is in the domain (tests:
// fpd_set_option("conv_c3", v)); FPD_C3_BLOCKS: persistent blocks of a launch (default 160 of the 256 CUs, see c3_blocks()).
int g_c3_mode = -1, g_c3_blocks = -1;
std::atomic<int> g_c3_launches{0};
int c3_mode() {
    if (g_c3_mode < 0) { const char* e = getenv("FPD_C3"); g_c3_mode = e ? atoi(e) : 1; }
    return g_c3_mode;
}
int c3_blocks() {
    // (160, not 256: the block owns its CU's LDS (157 KB) and the frozen teacher's fused Bottlenecks on the other stream need compute
    //  units of their own; r06 sweep inside the step, one box: 128 / 160 / 192 / 224 / 256 blocks -> 9.02 / 9.07 / 9.08 / 9.21 / 9.25 ms)
    if (g_c3_blocks < 0) { const char* e = getenv("FPD_C3_BLOCKS"); g_c3_blocks = e ? atoi(e) : 160; }
    return g_c3_blocks < 1 ? 1 : g_c3_blocks;
}
int c3_min_px() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("FPD_C3_MIN_PX"); v = e ? atoi(e) : 32768; }
    return v;
}
bool c3_aligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool c3_domain(const fpd_conv_t& a) {
    if (a.dtype != FPD_BF16 || a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.P != a.H || a.Q != a.W) return false;
    if (a.C != C3_C || a.K != C3_C) return false;
    if (a.W < 16 || a.W > 64 || (a.W & (a.W - 1)) != 0) return false;      // (128-wide rows: the image of a strip does not fit beside the weights)
    const int rs = C3_STRIP / a.W;
    if (a.H % rs != 0) return false;                      // whole strips inside one image
    if ((long long)a.N * a.H * a.W > (1ll << 28)) return false;
    if (!c3_aligned(a.x) || !c3_aligned(a.y) || !c3_aligned(a.w) || !c3_aligned(a.epi_x) || !c3_aligned(a.fold_x) || !c3_aligned(a.fold_out))
        return false;
    if (a.residual != nullptr || a.wg_partial != nullptr || a.y == a.x) return false;
    if (a.epi == FPD_EPI_BNRELU_BWD) {
        if (a.bias != nullptr || a.bn.mode != FPD_BN_NONE) return false;
    } else {
        if (a.epi != FPD_EPI_PLAIN || a.fold_x != nullptr) return false;
    }
    return true;
}
bool c3_takes(const fpd_conv_t& a, const fpd_conv_t* b) {
    const int mode = c3_mode();
    if (mode == 0 || !c3_domain(a)) return false;
    long long px = (long long)a.N * a.H * a.W;
    if (b != nullptr) {
        if (!c3_domain(*b) || a.epi != b->epi) return false;
        px += (long long)b->N * b->H * b->W;
    }
    return mode != 1 || px >= c3_min_px();
}
C3Geo c3_geo(const fpd_conv_t& a) {
    C3Geo g;
    g.lgW = 0;
    while ((1 << g.lgW) < a.W) ++g.lgW;
    g.rs = C3_STRIP / a.W;
    g.nstrip = a.N * a.H / g.rs;
    g.nblk = 0;
    g.img_bytes = (g.rs + 2) * (a.W + 2) * 128;
    return g;
}
struct C3Plan { C3Geo ga, gb; int grid; size_t lds; };
bool c3_plan(const fpd_conv_t& a, const fpd_conv_t* b, C3Plan& pl) {
    pl.ga = c3_geo(a);
    pl.gb = b ? c3_geo(*b) : pl.ga;
    pl.gb.nblk = 0;
    const int sa = pl.ga.nstrip, sb = b ? pl.gb.nstrip : 0;
    // blocks under the cap, balanced: every block walks the same number of strips (conv_c1.hip)
    int total = std::max(1, std::min(c3_blocks(), sa + sb));
    total = cdiv(sa + sb, cdiv(sa + sb, total));
    if (b != nullptr) {
        if (total < 2) return false;
        pl.gb.nblk = std::max(1, std::min(total - 1, (int)((long long)total * sb / (sa + sb))));
    }
    pl.ga.nblk = total - pl.gb.nblk;
    pl.grid = total;
    const int img = std::max(pl.ga.img_bytes, b ? pl.gb.img_bytes : 0);
    pl.ga.img_bytes = pl.gb.img_bytes = img;
    pl.lds = (size_t)C3_TABLES + C3_WBYTES + std::max(img + C3_NW * C3_TILE_O, C3_FLUSH);
    return pl.lds <= 160 * 1024;
}

template <bool BWD, bool FOLD>
int c3_launch_t(const fpd_conv_t& a, const fpd_conv_t* b, const C3Plan& pl, hipStream_t st) {
    static LdsAttr configured;
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&c3_kernel<BWD, FOLD>), pl.lds)) return rc_;
    C3Args args;
    args.c[0] = a; args.c[1] = b ? *b : a; args.g[0] = pl.ga; args.g[1] = pl.gb;
    FPD_LAUNCH((c3_kernel<BWD, FOLD>), dim3(pl.grid), dim3(512), pl.lds, st, args);
    g_c3_launches.fetch_add(1, std::memory_order_relaxed);
    return 0;
}
int c3_launch(const fpd_conv_t& a, const fpd_conv_t* b, hipStream_t st) {
    C3Plan pl;
    if (!c3_plan(a, b, pl)) return 1;
    if (a.epi == FPD_EPI_BNRELU_BWD) {
        if (a.fold_x != nullptr || (b != nullptr && b->fold_x != nullptr)) return c3_launch_t<true, true>(a, b, pl, st);
        return c3_launch_t<true, false>(a, b, pl, st);
    }
    return c3_launch_t<false, false>(a, b, pl, st);
}

}  // namespace

int fpd_conv_c3_option(int which, int value) {      // which: 0 = mode, 1 = blocks (returns the previous value), 2 = launches served so far
    if (which == 2) return g_c3_launches.load(std::memory_order_relaxed);
    int& g = which == 0 ? g_c3_mode : g_c3_blocks;
    const int prev = which == 0 ? c3_mode() : c3_blocks();
    g = value;
    return prev;
}
// 0 = launched, 1 = outside this kernel's domain (the caller tries conv_pp next), < 0 error
int fpd_conv_c3_launch(const fpd_conv_t& a, hipStream_t st) {
    if (!c3_takes(a, nullptr)) return 1;
    return c3_launch(a, nullptr, st);
}
int fpd_conv_c3_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    if (!c3_takes(a, &b)) return 1;
    return c3_launch(a, &b, st);
}
// 1 if the launch (pair) is served by this kernel as a BNRELU_BWD data gradient (a folded BN-backward apply is then evaluated on
// its operand; the weight gradient of a 3x3 convolution stays a separate launch that reads fold_out)
int fpd_conv_c3_fold_ok(const fpd_conv_t& a, const fpd_conv_t* b) {
    if (a.epi != FPD_EPI_BNRELU_BWD || (b != nullptr && b->epi != FPD_EPI_BNRELU_BWD)) return 0;
    C3Plan pl;
    return (c3_takes(a, b) && c3_plan(a, b, pl)) ? 1 : 0;
}
// 1 if this kernel takes the launch (so that no other kernel's fused weight gradient may be planned for it)
int fpd_conv_c3_takes(const fpd_conv_t& a, const fpd_conv_t* b) {
    C3Plan pl;
    return (c3_takes(a, b) && c3_plan(a, b, pl)) ? 1 : 0;
}
