// stem_s2d: Conv2d(3, K, 7, stride 2, pad 3) (/root/reference/lib/models/hourglass.py:116,172) as a STRIDE-1 4x4 convolution over
// the space-to-depth image, on the matrix cores (bf16 build, forward; round 5).
//
// stem_fwd_mfma (stem_mfma.hip) builds an explicit im2col tile per 128 output pixels: every block converts and scatters the whole
// fp32 weight array, stages a 66 KB fp32 patch, expands it element by element into A[pixel][160 taps] and only then issues its
// 10-20 MFMAs -- 130 KB of LDS, one 4-wave block per CU, 170 us (K = 64) / 105 us (K = 32) for a layer whose HBM roofline is
// 15 us; the student's launch sits on the critical chain.  Here the stride disappears instead: with u = 2p - 3 + r written as
// u + 4 = 2 (p + kr) + dy, the 7 taps of a row are 4 whole steps kr of a half-resolution grid times 2 phases dy (one of the 8 slots
// has no tap: zero weight), likewise for columns.  So
//     y[p][q][k] = sum_{kr, ks < 4} sum_{ch < 16} S[p + kr][q + ks][ch] * W'[k][kr][ks][ch]
// with the space-to-depth pixel S[R][Cc][(dy*3 + c)*2 + dx] = x[c][2R + dy - 4][2Cc + dx - 4] (12 channels, padded to 16, zero
// outside the image) -- a 4x4 "same-ish" convolution with 16 channels whose operand fragment is ONE 16-byte LDS read per tap: no
// im2col.  The S rows live in a ring in LDS (bf16, 48-byte pixels: conflict-free 16-byte reads); a persistent block owns a
// contiguous range of 128-pixel tiles, converts the weights once, and a tile brings only its new input rows (requested one tile
// ahead in registers).  K = 256 products per output instead of 147 (zero weights in the unused slots): the matrix pipe has the
// room.  The MFMAs are "transposed" (weights first: a lane ends up with 16 channels of ONE pixel), so the epilogue needs no LDS pass:
// bias, rounding and 8-byte stores straight from the accumulators, and the statistics of y (for the train-mode bn1) are kept per
// lane over all tiles of the block -- shifted by the channel's bias, a common shift -- and leave the block ONCE, as exact limbs
// (r05 stamps of the first version, which went through the shared LDS-staged epilogue with a statistics flush per tile: 6 k of
// a tile's 10 k cycles were epilogue, and the block's 30 k-cycle prologue was a chain of dependent weight loads).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int S2_PIXB = 48;                 // bytes of an LDS pixel: 16 bf16 + 16 bytes of padding
constexpr int S2_WROW = 16 * 16 * 2 + 16;   // bytes of a weight row [16 taps][16 ch] bf16 + padding
constexpr int S2_NV = 4;                    // float2 vectors per thread of a tile's new rows (rows * 6 * (Q + 3) <= 1024)

// Probe build only (-DS2_TIMING): cycle stamps of thread 0 of block 0, printed by the kernel
#ifdef S2_TIMING
__shared__ long long s2_stamp[48];
__shared__ int s2_ns;
#define S2_STAMP() do { if (threadIdx.x == 0 && blockIdx.x == 0 && s2_ns < 48) s2_stamp[s2_ns++] = clock64(); } while (0)
#else
#define S2_STAMP() do { } while (0)
#endif

template <int TN>
__global__ __launch_bounds__(256) void stem_s2d_fwd_kernel(const fpd_stem_t a, const int logQ, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int K = a.K, P = a.P, Q = a.Q, H = a.H, W = a.W;
    const int rows = 128 >> logQ, QP = Q + 3, RING = 2 * rows + 3;
    const unsigned RSs = (unsigned)QP * S2_PIXB, RINGB = (unsigned)RING * RSs;
    const int M = a.N * P * Q;
    unsigned char* sS = smem;                                        // [RING][QP][48 B]
    unsigned char* sW = sS + RINGB;                                  // [32 TN][S2_WROW]
    const float* __restrict__ x = a.x;
    auto wrap = [&](unsigned v) { return min(v, v - RINGB); };

#ifdef S2_TIMING
    if (threadIdx.x == 0) s2_ns = 0;
    __syncthreads();
#endif
    S2_STAMP();
    // this block's contiguous tile range
    const int t_beg = fpd_cut((int)blockIdx.x, ntiles, (int)gridDim.x), t_end = fpd_cut((int)blockIdx.x + 1, ntiles, (int)gridDim.x);
    if (t_beg >= t_end) return;

    // ---- once per block: weights -> bf16 [k][tap = kr*4 + ks][ch = (dy*3 + c)*2 + dx], zero where the slot has no tap ----
    // (thread = element e of a 256-element row [tap][ch]; rows k = 0 .. 32 TN - 1: all requests of a thread go out before the
    //  first conversion -- as a load -> store loop this prologue was 32 TN exposed memory latencies, 30 k cycles at K = 32)
    {
        const int e = tid, tap = e >> 4, ch = e & 15;
        const int kr = tap >> 2, ks = tap & 3, dx = ch & 1, dc = ch >> 1, dy = dc >= 3 ? 1 : 0, cch = dc - 3 * dy;
        const int r = 2 * kr + dy - 1, s_ = 2 * ks + dx - 1;
        const bool tap_ok = ch < 12 && r >= 0 && s_ >= 0;
        const int woff = tap_ok ? (r * 7 + s_) * 3 + cch : 0;
        float wv[32 * TN];
#pragma unroll
        for (int k = 0; k < 32 * TN; ++k) wv[k] = a.w[(unsigned)(min(k, K - 1) * 147 + woff)];      // unconditional (clamped): no branch, no wait per request
#pragma unroll
        for (int k = 0; k < 32 * TN; ++k) *reinterpret_cast<bf16_t*>(sW + k * S2_WROW + tap * 32 + ch * 2) = f2bf((tap_ok && k < K) ? wv[k] : 0.f);
    }
    // the padding channels 12..15 of every ring pixel are never written again: zero (a zero weight times stale bits could be NaN)
    for (int i = tid; i < RING * QP; i += 256) *reinterpret_cast<uint2*>(sS + (unsigned)i * S2_PIXB + 24) = make_uint2(0u, 0u);

    // ---- staging of S rows: vector = (row, dc = dy*3 + c, column cc): the two input pixels (dx = 0, 1) of one (row, phase, channel)
    auto s_row_store = [&](int n, int R, int dc, int cc, float2 v) {
        const unsigned slot = (unsigned)(R % RING);
        *reinterpret_cast<unsigned*>(sS + slot * RSs + (unsigned)cc * S2_PIXB + dc * 4) = f2bf_pk(v.x, v.y);
    };
    auto s_load = [&](int n, int R, int dc, int cc) {
        const int dy = dc >= 3 ? 1 : 0, c = dc - 3 * dy;
        const int ih = 2 * R + dy - 4, iw = 2 * cc - 4;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        float2 v = *reinterpret_cast<const float2*>(x + ((size_t)(n * 3 + c) * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 2));   // clamped: branch-free
        if (!ok) v = make_float2(0.f, 0.f);
        return v;
    };
    // full (re)load of rows [R0, R1) of image n: the first tile of the block and of every image
    // (requests in batches of 8 before their stores: a load -> store loop exposed one memory latency per vector)
    auto full_load = [&](int n, int R0, int R1) {
        const int per_row = 6 * QP, total = (R1 - R0) * per_row;
        for (int base = 0; base < total; base += 8 * 256) {
            float2 v[8];
            int rl[8], dcv[8], ccv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = min(base + tid + i * 256, total - 1);
                rl[i] = idx / per_row;
                const int rem = idx - rl[i] * per_row;
                dcv[i] = rem / QP; ccv[i] = rem - dcv[i] * QP;
                v[i] = s_load(n, R0 + rl[i], dcv[i], ccv[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (base + tid + i * 256 < total) s_row_store(n, R0 + rl[i], dcv[i], ccv[i], v[i]);
        }
    };
    // the `rows` new rows of a tile that continues the previous one: a thread's S2_NV vectors, decoded once
    int v_rl[S2_NV], v_dc[S2_NV], v_cc[S2_NV];
    const int per_row = 6 * QP, nnew = rows * per_row;
#pragma unroll
    for (int i = 0; i < S2_NV; ++i) {
        const int idx = min(tid + i * 256, nnew - 1);
        v_rl[i] = idx / per_row;
        const int rem = idx - v_rl[i] * per_row;
        v_dc[i] = rem / QP;
        v_cc[i] = rem - v_dc[i] * QP;
    }
    float2 pv0, pv1, pv2, pv3;
#define S2_PREF(i, PV, n_, R0_) PV = (tid + i * 256 < nnew) ? s_load(n_, (R0_) + v_rl[i], v_dc[i], v_cc[i]) : make_float2(0.f, 0.f);
#define S2_PUT(i, PV, n_, R0_) if (tid + i * 256 < nnew) s_row_store(n_, (R0_) + v_rl[i], v_dc[i], v_cc[i], PV);

    // per-lane MFMA addressing that does not depend on the tile
    const int ml = wave * 32 + l31, ti = ml >> logQ, tj = ml & (Q - 1);
    const unsigned a_lane = (unsigned)tj * S2_PIXB + (unsigned)hh * 16;
    const unsigned char* wrow = sW + l31 * S2_WROW + hh * 16;

    // epilogue state: accumulator element 4 j + i of tile tn is channel 32 tn + 8 j + 4 hh + i of pixel ml
    bf16_t* __restrict__ y = reinterpret_cast<bf16_t*>(a.y);
    const bool want_stats = a.out_stats != nullptr;
    f32x4 bias4[TN][4], f1[TN][4], f2[TN][4];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 32 * tn + 8 * j + 4 * hh;
            bias4[tn][j] = (k < K) ? f32x4{a.bias[k], a.bias[k + 1], a.bias[k + 2], a.bias[k + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};      // K % 8 == 0
            f1[tn][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            f2[tn][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

    int ring_n = -1, ring_hi = 0;      // rows [.., ring_hi) of image ring_n are in the ring (as far as the current tile needs them)
    bool pref = false;                 // pv0..3 hold the new rows of the coming tile
    for (int t = t_beg; t < t_end; ++t) {
        const int g0 = t * rows, n = g0 / P, p0 = g0 - n * P;             // P % rows == 0: a tile lies in one image
        S2_STAMP();                                                        // tile start
        __syncthreads();                                                   // previous tile done with the LDS (first time: weights)
        if (pref && n == ring_n && p0 + 3 == ring_hi) {
            S2_PUT(0, pv0, n, ring_hi) S2_PUT(1, pv1, n, ring_hi) S2_PUT(2, pv2, n, ring_hi) S2_PUT(3, pv3, n, ring_hi)
        } else {
            full_load(n, p0, p0 + rows + 3);
        }
        ring_n = n; ring_hi = p0 + rows + 3;
        S2_STAMP();                                                        // rows stored
        __syncthreads();
        // request the next tile's new rows (same image, next rows) while this one is multiplied
        pref = false;
        if (t + 1 < t_end && p0 + rows < P) {
            S2_PREF(0, pv0, n, ring_hi) S2_PREF(1, pv1, n, ring_hi) S2_PREF(2, pv2, n, ring_hi) S2_PREF(3, pv3, n, ring_hi)
            pref = true;
        }
        S2_STAMP();                                                        // next rows requested
        // ---- 16 taps x TN MFMAs: operand = 16 bytes of pixel (p0 + ti + kr, tj + ks) ----
        f32x16 acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;
        const unsigned r0 = (unsigned)((p0 + ti) % RING) * RSs;
#pragma unroll
        for (int kr = 0; kr < 4; ++kr) {
            const unsigned char* arow = sS + wrap(r0 + (unsigned)kr * RSs) + a_lane;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + ks * S2_PIXB);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wrow + tn * 32 * S2_WROW + (kr * 4 + ks) * 32);
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[tn], 0, 0, 0);      // weights first: rows = channels
                }
            }
        }
        S2_STAMP();                                                        // MFMAs issued
        // ---- epilogue from the accumulators: + bias, one rounding, 8-byte stores; statistics of the ROUNDED values, shifted by
        //      the channel's bias (common to every lane and block; the convolution of a normalised image is centred on it) ----
        bf16_t* yrow = y + (size_t)(t * 128 + ml) * K + 4 * hh;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (32 * tn + 8 * j < K) {
                    const f32x4 b = bias4[tn][j];
                    const float v0 = acc[tn][4 * j] + b[0], v1 = acc[tn][4 * j + 1] + b[1], v2 = acc[tn][4 * j + 2] + b[2], v3 = acc[tn][4 * j + 3] + b[3];
                    const unsigned p0 = f2bf_pk(v0, v1), p1 = f2bf_pk(v2, v3);
                    *reinterpret_cast<uint2*>(yrow + 32 * tn + 8 * j) = make_uint2(p0, p1);
                    if (want_stats) {
                        const f32x4 d = {__uint_as_float(p0 << 16) - b[0], __uint_as_float(p0 & 0xffff0000u) - b[1],
                                         __uint_as_float(p1 << 16) - b[2], __uint_as_float(p1 & 0xffff0000u) - b[3]};
                        f1[tn][j] += d;
                        f2[tn][j] = __builtin_elementwise_fma(d, d, f2[tn][j]);
                    }
                }
            }
    }
    // ---- statistics: one flush per block.  The 32 pixel lanes of a (wave, channel) are added in lane order through an LDS
    //      transposition (a shuffle tree is 160 dependent cross-lane operations per lane: 10 k cycles at the end of every block,
    //      r05 stamps), the four waves in wave order, un-shifted in fp64, and ONE exact pair of sums per channel leaves the block ----
    if (want_stats) {
        float* s_t = reinterpret_cast<float*>(smem);            // [4 waves][32 TN channels][2 sums][33]: the ring / weights are dead
        __syncthreads();
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 32 * tn + 8 * j + 4 * hh + i;
                    s_t[((wave * 32 * TN + k) * 2 + 0) * 33 + l31] = f1[tn][j][i];
                    s_t[((wave * 32 * TN + k) * 2 + 1) * 33 + l31] = f2[tn][j][i];
                }
        __syncthreads();
        double* s_red2 = reinterpret_cast<double*>(s_t + 4 * 32 * TN * 2 * 33);      // [4 * 32 TN * 2] doubles behind the transposition
        for (int task = tid; task < 4 * 32 * TN * 2; task += 256) {
            float tot = 0.f;
#pragma unroll 8
            for (int l = 0; l < 32; ++l) tot += s_t[task * 33 + l];
            s_red2[task] = (double)tot;
        }
        __syncthreads();
        if (tid < K) {
            double u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { u1 += s_red2[(w * 32 * TN + tid) * 2]; u2 += s_red2[(w * 32 * TN + tid) * 2 + 1]; }
            const double b = (double)a.bias[tid], cnt = (double)(t_end - t_beg) * 128.0;
            stat_atomic_add(a.out_stats, K, 0, tid, u1 + cnt * b);
            stat_atomic_add(a.out_stats, K, 1, tid, u2 + 2.0 * b * u1 + cnt * b * b);
        }
    }
    S2_STAMP();
#ifdef S2_TIMING
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        printf("stem_s2d<%d> stamps (cycles since entry), %d tiles:", TN, t_end - t_beg);
        for (int i = 1; i < s2_ns; ++i) printf(" %lld", s2_stamp[i] - s2_stamp[0]);
        printf("\n");
    }
#endif
#undef S2_PREF
#undef S2_PUT
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem in the same space-to-depth form (K <= 32: the hourglass student; round 5):
//     dW'[k][kr][ks][ch] = sum_pixels dy[p][q][k] * S[p + kr][q + ks][ch],      dw[k][r][s][c] = dW'[k][kr][ks][(dy*3 + c)*2 + dx]
// for r = 2 kr + dy - 1 >= 0, s = 2 ks + dx - 1 >= 0; dbias[k] = sum dy.  It is the LAST weight gradient of a backward -- its
// operand is the last tensor the chain produces -- so its duration is exposed in front of Adam (81 us as an im2col GEMM,
// stem_wgrad_mfma).  Both operands stay pixel-major in LDS (the S ring of the forward kernel, the dy tile as it sits in HBM) and
// reach the MFMA through transposing reads; an operand's 32 columns are two taps of 16 channels, (kr, ks) and (kr, ks + 2), which
// are the SAME pixels shifted by two -- a per-lane address offset --, and the pair (ks + 1, ks + 3) is their funnel shift by one
// pixel: 3 reads + 4 v_perm per tap row for 2 MFMAs.  4 waves split the eight 16-pixel k-steps of a tile, 8 accumulator tiles each;
// fixed-order cross-wave sum through LDS at the end, one slab per block.
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef short s16x4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4s* lds_s16x4s_ptr;
__device__ __forceinline__ u32x2s s2_tr(const unsigned char* p) {
    return __builtin_bit_cast(u32x2s, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4s_ptr)(p)));
}

__global__ __launch_bounds__(256) void stem_s2d_wgrad_kernel(const fpd_stem_t a, const int logQ, const int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = a.K, P = a.P, Q = a.Q, H = a.H, W = a.W;       // K <= 32
    const int rows = 128 >> logQ, QP = Q + 3, RING = 2 * rows + 3;
    const unsigned RSs = (unsigned)QP * S2_PIXB, RINGB = (unsigned)RING * RSs;
    unsigned char* sS = smem;                                        // [RING][QP][48 B]
    unsigned char* sD = sS + RINGB;                                  // dy tile [128][32] bf16 (64-byte rows); the ring's over-reads end here
    const float* __restrict__ x = a.x;
    const bf16_t* __restrict__ dyp = reinterpret_cast<const bf16_t*>(a.dy);
    auto wrap = [&](unsigned v) { return min(v, v - RINGB); };
    const int t_beg = fpd_cut((int)blockIdx.x, ntiles, (int)gridDim.x), t_end = fpd_cut((int)blockIdx.x + 1, ntiles, (int)gridDim.x);

    for (int i = tid; i < RING * QP; i += 256) *reinterpret_cast<uint2*>(sS + (unsigned)i * S2_PIXB + 24) = make_uint2(0u, 0u);
    for (int i = tid; i < 128 * 4; i += 256) *reinterpret_cast<uint4*>(sD + i * 16) = make_uint4(0, 0, 0, 0);      // columns >= K stay zero

    auto s_row_store = [&](int R, int dc, int cc, float2 v) {
        const unsigned slot = (unsigned)(R % RING);
        *reinterpret_cast<unsigned*>(sS + slot * RSs + (unsigned)cc * S2_PIXB + dc * 4) = f2bf_pk(v.x, v.y);
    };
    auto s_load = [&](int n, int R, int dc, int cc) {
        const int dy = dc >= 3 ? 1 : 0, c = dc - 3 * dy;
        const int ih = 2 * R + dy - 4, iw = 2 * cc - 4;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        float2 v = *reinterpret_cast<const float2*>(x + ((size_t)(n * 3 + c) * H + min(max(ih, 0), H - 1)) * W + min(max(iw, 0), W - 2));
        if (!ok) v = make_float2(0.f, 0.f);
        return v;
    };
    auto full_load = [&](int n, int R0, int R1) {
        const int per_row = 6 * QP, total = (R1 - R0) * per_row;
        for (int base = 0; base < total; base += 8 * 256) {
            float2 v[8];
            int rl[8], dcv[8], ccv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = min(base + tid + i * 256, total - 1);
                rl[i] = idx / per_row;
                const int rem = idx - rl[i] * per_row;
                dcv[i] = rem / QP; ccv[i] = rem - dcv[i] * QP;
                v[i] = s_load(n, R0 + rl[i], dcv[i], ccv[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (base + tid + i * 256 < total) s_row_store(R0 + rl[i], dcv[i], ccv[i], v[i]);
        }
    };
    int v_rl[S2_NV], v_dc[S2_NV], v_cc[S2_NV];
    const int per_row = 6 * QP, nnew = rows * per_row;
#pragma unroll
    for (int i = 0; i < S2_NV; ++i) {
        const int idx = min(tid + i * 256, nnew - 1);
        v_rl[i] = idx / per_row;
        const int rem = idx - v_rl[i] * per_row;
        v_dc[i] = rem / QP;
        v_cc[i] = rem - v_dc[i] * QP;
    }
    float2 pv0, pv1, pv2, pv3;
#define S2_PREF(i, PV, n_, R0_) PV = s_load(n_, (R0_) + v_rl[i], v_dc[i], v_cc[i]);
#define S2_PUT(i, PV, R0_) if (tid + i * 256 < nnew) s_row_store((R0_) + v_rl[i], v_dc[i], v_cc[i], PV);
    // dy tile: 128 pixels x K channels = 128 x (K / 8) 16-byte vectors, <= 2 per thread; LDS rows of 64 bytes
    const int vpr = K >> 3;
    const int dpx0 = tid / vpr, dcv0 = (tid - dpx0 * vpr) * 8, dpx1 = (tid + 256) / vpr, dcv1 = (tid + 256 - dpx1 * vpr) * 8;
    const bool dok0 = tid < 128 * vpr, dok1 = tid + 256 < 128 * vpr;
    uint4 rd0 = make_uint4(0, 0, 0, 0), rd1 = rd0;
    auto d_load = [&](int t) {
        const bf16_t* base = dyp + (size_t)t * 128 * K;
        if (dok0) rd0 = *reinterpret_cast<const uint4*>(base + dpx0 * K + dcv0);
        if (dok1) rd1 = *reinterpret_cast<const uint4*>(base + dpx1 * K + dcv1);
    };
    auto d_store = [&]() {
        if (dok0) *reinterpret_cast<uint4*>(sD + dpx0 * 64 + dcv0 * 2) = rd0;
        if (dok1) *reinterpret_cast<uint4*>(sD + dpx1 * 64 + dcv1 * 2) = rd1;
    };

    f32x16 acc[8];                                   // [kr][pair]: columns 0..15 = tap (kr, pair), 16..31 = tap (kr, pair + 2)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
    const unsigned bsel = (((lane & 15) == 0 && ((lane >> 4) & 1) == 0) || ((lane & 15) == 1 && ((lane >> 4) & 1) == 1)) ? 0x3f803f80u : 0u;
    const u32x4s bones = {bsel, bsel, bsel, bsel};
    const bool do_bias = a.dbias != nullptr;
    // lane geometry of a transposing fragment read (mfma_frag.h): 16-lane group g, lane s: row 8 (g >> 1) + (s >> 2), column half g & 1
    const int g = lane >> 4, s16 = lane & 15;
    const unsigned fr_row = (unsigned)(8 * (g >> 1) + (s16 >> 2));
    const unsigned a_off = fr_row * 64 + (unsigned)(16 * (g & 1) + 4 * (s16 & 3)) * 2;                       // dy tile: 32 columns = channels
    const unsigned b_off = (fr_row + 2 * (g & 1)) * S2_PIXB + (unsigned)(4 * (s16 & 3)) * 2;                // S ring: column half = pixel shift + 2

    int ring_n = -1, ring_hi = 0;
    bool pref = false;
    if (t_beg < t_end) d_load(t_beg);
    for (int t = t_beg; t < t_end; ++t) {
        const int g0 = t * rows, n = g0 / P, p0 = g0 - n * P;
        __syncthreads();                                                   // previous tile's fragment reads are done
        if (pref && n == ring_n && p0 + 3 == ring_hi) {
            S2_PUT(0, pv0, ring_hi) S2_PUT(1, pv1, ring_hi) S2_PUT(2, pv2, ring_hi) S2_PUT(3, pv3, ring_hi)
        } else {
            full_load(n, p0, p0 + rows + 3);
        }
        d_store();
        ring_n = n; ring_hi = p0 + rows + 3;
        __syncthreads();
        pref = false;
        if (t + 1 < t_end) {
            d_load(t + 1);
            if (p0 + rows < P) {
                S2_PREF(0, pv0, n, ring_hi) S2_PREF(1, pv1, n, ring_hi) S2_PREF(2, pv2, n, ring_hi) S2_PREF(3, pv3, n, ring_hi)
                pref = true;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int pix0 = (wave + 4 * h) * 16;                          // this wave's k-step: 16 pixels of one output row
            const int ti = pix0 >> logQ, tj0 = pix0 & (Q - 1);
            union { struct { u32x2s a, b; } hh; bf16x8 f; } ua;
            ua.hh.a = s2_tr(sD + (unsigned)pix0 * 64 + a_off);
            ua.hh.b = s2_tr(sD + (unsigned)pix0 * 64 + a_off + 4 * 64);
            const bf16x8 af = ua.f;
            const unsigned r0 = (unsigned)((p0 + ti) % RING) * RSs;
            const unsigned col = (unsigned)tj0 * S2_PIXB + b_off;
#pragma unroll
            for (int kr = 0; kr < 4; ++kr) {
                const unsigned char* base = sS + wrap(r0 + (unsigned)kr * RSs) + col;
                const u32x2s lo = s2_tr(base), hi = s2_tr(base + 4 * S2_PIXB), nx = s2_tr(base + 8 * S2_PIXB);
                const u32x4s b02 = {lo[0], lo[1], hi[0], hi[1]};
                const u32x4s b13 = {__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(hi[0], lo[1], 16),
                                    __builtin_amdgcn_alignbit(hi[1], hi[0], 16), __builtin_amdgcn_alignbit(nx[0], hi[1], 16)};
                acc[2 * kr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b02), acc[2 * kr], 0, 0, 0);
                acc[2 * kr + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b13), acc[2 * kr + 1], 0, 0, 0);
            }
            if (do_bias) accb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8, bones), accb, 0, 0, 0);
        }
    }
#undef S2_PREF
#undef S2_PUT

    // ---- flush: the four waves added in wave order, one tap row (two accumulator tiles) per pass, scattered into the dw layout ----
    float* slab = a.partial != nullptr ? a.partial + (size_t)blockIdx.x * a.partial_stride : nullptr;
    float* s_red = reinterpret_cast<float*>(smem);                      // [4 waves][2 tiles][16][64]
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
        __syncthreads();
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int e = 0; e < 16; ++e) s_red[((wave * 2 + t2) * 16 + e) * 64 + lane] = acc[2 * kr + t2][e];
        __syncthreads();
        for (int id = tid; id < 2 * 16 * 64; id += 256) {
            const int t2 = id >> 10, e = (id >> 6) & 15, l = id & 63;
            float sum = s_red[((0 * 2 + t2) * 16 + e) * 64 + l];
#pragma unroll
            for (int w = 1; w < 4; ++w) sum += s_red[((w * 2 + t2) * 16 + e) * 64 + l];
            const int k = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
            const int ks = t2 + 2 * (col >> 4), ch = col & 15, dx = ch & 1, dc = ch >> 1, dyy = dc >= 3 ? 1 : 0, c = dc - 3 * dyy;
            const int r = 2 * kr + dyy - 1, sx = 2 * ks + dx - 1;
            if (k < K && ch < 12 && r >= 0 && sx >= 0) {
                const size_t idx = ((size_t)(k * 7 + r) * 7 + sx) * 3 + c;
                if (slab != nullptr) slab[idx] = sum; else a.dw[idx] += sum;
            }
        }
    }
    if (do_bias) {
        __syncthreads();
        if ((lane & 15) < 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s_red[wave * 32 + 16 * (lane & 15) + 4 * (lane >> 4) + e] = accb[e];
        }
        __syncthreads();
        if (tid < K) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) tot += s_red[w * 32 + tid];
            if (slab != nullptr) slab[(size_t)K * 147 + tid] = tot; else a.dbias[tid] += tot;
        }
    }
}

}  // namespace

// return 1 = not applicable (the caller falls through to stem_fwd_mfma / the direct kernels)
int fpd_stem_forward_s2d_launch(const fpd_stem_t& a, hipStream_t st) {
    if (a.dtype != FPD_BF16 || a.K % 8 != 0 || a.K > 64) return 1;
    if (a.Q > 128 || a.Q < 16 || (a.Q & (a.Q - 1)) != 0 || a.H != 2 * a.P || a.W != 2 * a.Q) return 1;
    int logQ = 0;
    while ((1 << logQ) < a.Q) ++logQ;
    const int rows = 128 >> logQ;
    if (a.P % rows != 0) return 1;
    const int TN = a.K > 32 ? 2 : 1;
    const int ring = 2 * rows + 3;
    const size_t lds = std::max((size_t)ring * (a.Q + 3) * S2_PIXB + (size_t)32 * TN * S2_WROW,
                                (size_t)4 * 32 * TN * 2 * (33 * sizeof(float) + sizeof(double)));      // tiles | block-end statistics transposition
    const int tiles = a.N * a.P * a.Q / 128;
    static const int cap = 0;        // (0 = the per-K defaults below; the FPD_STEM_BLOCKS knob of round 5 is gone)
    // K = 32: two blocks per CU (35.6 us at 512 blocks, 47.6 at 256, 44.1 at 1024); K = 64: 62.2 us at 256, 71.8 at 512 (r05, one box)
    const int blocks = std::max(1, std::min(tiles, cap > 0 ? cap : (TN == 1 ? 512 : 256)));
    static LdsAttr cfg1, cfg2;
    if (TN == 1) {
        if (int rc_ = cfg1.ensure(reinterpret_cast<const void*>(&stem_s2d_fwd_kernel<1>), lds)) return rc_;
        FPD_LAUNCH((stem_s2d_fwd_kernel<1>), dim3(blocks), dim3(256), lds, st, a, logQ, tiles);
    } else {
        if (int rc_ = cfg2.ensure(reinterpret_cast<const void*>(&stem_s2d_fwd_kernel<2>), lds)) return rc_;
        FPD_LAUNCH((stem_s2d_fwd_kernel<2>), dim3(blocks), dim3(256), lds, st, a, logQ, tiles);
    }
    return 0;
}

static bool s2d_wgrad_ok(const fpd_stem_t& a, int& logQ, int& blocks) {
    if (a.dtype != FPD_BF16 || a.K % 8 != 0 || a.K > 32) return false;
    if (a.Q > 128 || a.Q < 16 || (a.Q & (a.Q - 1)) != 0 || a.H != 2 * a.P || a.W != 2 * a.Q) return false;
    logQ = 0;
    while ((1 << logQ) < a.Q) ++logQ;
    if (a.P % (128 >> logQ) != 0) return false;
    const int tiles = a.N * a.P * a.Q / 128;
    static const int cap = 256;      // (= its slab count: 41.6 us; 128 / 512 blocks 67.4 / 52.5 us, round 5)
    blocks = std::max(1, std::min(tiles, cap));
    return true;
}

int fpd_stem_wgrad_s2d_partials(const fpd_stem_t& a) {
    int logQ, blocks;
    return s2d_wgrad_ok(a, logQ, blocks) ? blocks : 0;
}

// return 1 = not applicable
int fpd_stem_wgrad_s2d_launch(const fpd_stem_t& a, hipStream_t st) {
    int logQ, blocks;
    if (!s2d_wgrad_ok(a, logQ, blocks)) return 1;
    if (a.partial == nullptr) blocks = 1;                  // no slabs: one block adds straight into dw (small problems / tests)
    const int rows = 128 >> logQ, ring = 2 * rows + 3;
    const size_t lds = std::max((size_t)ring * (a.Q + 3) * S2_PIXB + (size_t)128 * 64 + 1024, (size_t)4 * 2 * 16 * 64 * sizeof(float));
    const int tiles = a.N * a.P * a.Q / 128;
    static LdsAttr cfg;
    if (int rc_ = cfg.ensure(reinterpret_cast<const void*>(&stem_s2d_wgrad_kernel), lds)) return rc_;
    FPD_LAUNCH(stem_s2d_wgrad_kernel, dim3(blocks), dim3(256), lds, st, a, logQ, tiles);
    return 0;
}
