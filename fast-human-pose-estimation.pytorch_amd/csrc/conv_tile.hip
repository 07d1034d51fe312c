// conv_tile: stride-1 1x1 / 3x3 "same" convolution as an implicit GEMM whose A operand is a halo tile staged ONCE
// per channel chunk in LDS (BatchNorm+ReLU applied once per element on the way in) and then read by all R*S filter
// taps at shifted LDS addresses; the weight tile of the next tap is prefetched into registers while the MFMAs of
// the current tap run (two LDS weight buffers, one barrier per tap), and the next chunk's halo is in flight in
// registers during all taps of the current chunk.
//
// Tile: nrows whole image rows = nrows*W <= 128 consecutive output pixels (W a power of two: exactly 128; HRNet's 96 / 48 /
// 24 / 12 / 6 wide maps: 96..126, the remaining MFMA rows idle) x 32*TN output channels, 4 wave64, each wave 32 pixels x
// 32*TN channels.  LDS rows are 128 B of channels + 16 B pad (bf16: 64 ch, fp32: 32 ch).
// Image-border rows are handled without branches in the MFMA loop: a lane whose tap row falls outside its image
// reads its A fragment from a block of zero pixels; left/right borders are explicit zero columns in the halo.
//
// Same contract as conv_mfma_kernel (see conv_mfma.hip / include/fpd_amd.h); replaces the same reference calls.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv_epilogue.h"

namespace {

template <typename T>
struct TapMma;
template <>
struct TapMma<bf16_t> {
    template <int TN, int BK, int LD>
    static __device__ __forceinline__ void run(const bf16_t* arow, const bf16_t* sB, int lane, f32x16* acc) {
        const int koff = 8 * (lane >> 5);
        const bf16_t* brow = sB + (lane & 31) * LD + koff;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + kk * 16 + koff);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(brow + tn * 32 * LD + kk * 16);
                acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[tn], 0, 0, 0);
            }
        }
    }
};
template <>
struct TapMma<float> {
    template <int TN, int BK, int LD>
    static __device__ __forceinline__ void run(const float* arow, const float* sB, int lane, f32x16* acc) {
        const int koff = (lane >> 5) * (BK / 2);
        const float* brow = sB + (lane & 31) * LD + koff;
#pragma unroll
        for (int t4 = 0; t4 < BK / 8; ++t4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(arow + koff + t4 * 4);
            f32x4 b[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = *reinterpret_cast<const f32x4*>(brow + tn * 32 * LD + t4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[tn][j], acc[tn], 0, 0, 0);
        }
    }
};

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

// tile geometry of one convolution: rows per tile and the multiply-high reciprocals of W and W*VPR (exact quotients for the
// small indices divided here: v * d < 2^32)
struct TileGeo { int nrows; unsigned mW, mWV; };
__device__ __forceinline__ int qdiv(int v, unsigned magic) { return (int)__umulhi((unsigned)v, magic); }
static inline unsigned magic_of(int d) { return (unsigned)((0x100000000ull / (unsigned long long)d) + 1ull); }

// One 128-pixel x 32*TN-channel tile of convolution `a`; (bx, by) = tile coordinates.  Shared by the single-conv
// kernel and the pair kernel (two independent convolutions in one launch).
// ALLW (3x3, one channel chunk, grids that do not fill the chip): the weights of ALL nine taps are staged up front, so the
// MFMA loop runs its 9 taps back to back behind ONE barrier instead of one barrier (and one exposed weight-load latency) per
// tap -- these launches are latency-bound, LDS capacity is not a constraint for them.
template <typename T, int TN, int BK, bool ALLW = false, bool FOLD = false>
__device__ __forceinline__ void conv_tile_body(const fpd_conv_t& a, const TileGeo geo, const int bx, const int by) {
    constexpr int VEC = DT<T>::VEC;
    constexpr int BNT = 32 * TN;
    constexpr int LD = BK + 16 / (int)sizeof(T);
    constexpr int VPR = BK / VEC, LOG_VPR = ilog2(VPR);
    constexpr int NVB_TOT = BNT * VPR, NVB = (NVB_TOT + 255) / 256;
    constexpr int NVH = 8;                       // halo vectors per thread (host guarantees the halo fits)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef FPD_TILE_TIMING
    if (tid == 0) fpd_tile_ns = 0;
#endif
    TILE_STAMP();
    const int H = a.H, W = a.W, C = a.C, K = a.K, R = a.R, pad = a.pad;
    const int M = a.N * H * W, GR = a.N * H;     // pixels, flattened (n,h) rows
    const int nrows = geo.nrows, hrows = nrows + R - 1, WP = W + R - 1;
    const int TPX = nrows * W;                   // output pixels of this tile (<= 128)
    const int zero_px = hrows * WP;              // 3 all-zero pixels behind the halo
    const int HP = zero_px + 3;
    const int m0 = bx * TPX, n0 = by * BNT;
    const int g0 = bx * nrows;

    // LDS: [BN tables | epilogue tables] then the tile region (halo + 2 weight buffers), which the epilogue reuses
    float* s_scale = reinterpret_cast<float*>(smem);
    float* s_shift = s_scale + C;
    float* s_epi = s_shift + C;
    // FOLDED BN-BACKWARD APPLY (fpd_conv_t.fold_x, BNRELU_BWD data gradients without a prologue BN): the operand proper is
    // dy = A_c g + B_c u + D_c (g = x, u = fold_x), evaluated on the way into the halo tile and rounded once
    const bool bwd_epi = a.epi == FPD_EPI_BNRELU_BWD;
    const bool fold = FOLD && bwd_epi && a.fold_x != nullptr;      // FOLD: compiled in for TN <= 2 only (register budget)
    float* s_fold = s_epi + 4 * BNT;                      // [3][C] when the epilogue is BNRELU_BWD
    T* sH = reinterpret_cast<T*>(s_fold + (bwd_epi ? 3 * C : 0));
    T* sB = sH + HP * LD;
    float* stage = reinterpret_cast<float*>(sH);         // epilogue staging tile (after the last MFMA)
    double* s_red = reinterpret_cast<double*>(sH);
    const T* __restrict__ x = reinterpret_cast<const T*>(a.x);
    const T* __restrict__ w = reinterpret_cast<const T*>(a.w);
    const T* __restrict__ fx = reinterpret_cast<const T*>(a.fold_x);
    T* fo = (fold && by == 0) ? reinterpret_cast<T*>(a.fold_out) : nullptr;       // written once: by the first channel tile

    // ---- per-lane A addressing: output pixel -> halo row/col; invalid tap rows point at the zero pixels ----
    const int ml = wave * 32 + (lane & 31);
    const int ti = qdiv(ml, geo.mW), tj = ml - ti * W;
    int ab0, ab1, ab2;
    {
        const int g = g0 + ti;
        const int p = g % H;
        const bool live = g < GR && ml < TPX;
        if (R == 3) {
            ab0 = (live && p - 1 >= 0) ? ((ti + 0) * WP + tj) * LD : zero_px * LD;
            ab1 = live ? ((ti + 1) * WP + tj) * LD : zero_px * LD;
            ab2 = (live && p + 1 < H) ? ((ti + 2) * WP + tj) * LD : zero_px * LD;
        } else {
            ab0 = ab1 = ab2 = live ? (ti * WP + tj) * LD : zero_px * LD;
        }
    }

    // ---- staging registers ----
    const int nvtot = hrows * W * VPR;
    const int WV = W * VPR;
    uint4 rh[NVH], ru[FOLD ? NVH : 1];           // ru: the same vectors of u (fold)
    unsigned hmask = 0;
    // Branch-free: every thread issues its requests back to back from clamped addresses (a row outside the tensor re-reads the
    // nearest one, a thread past the last vector re-reads the last one) and the staging pass zeroes / skips them by hmask /
    // v < nvtot.  With the loads under per-vector conditions the compiler funnelled them through one temporary and waited
    // for each pair (FOLD variants: s_waitcnt vmcnt(0) after every second vector = four exposed memory latencies, r04 stamps).
    const T* __restrict__ fxs = (FOLD && fold) ? fx : x;
    const int nvh = (nvtot + 255) >> 8;          // vectors per thread (block-uniform)
    auto halo_load = [&](int c0) {
        hmask = 0;
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            if (i < nvh) {
                const int v = min(tid + i * 256, nvtot - 1);
                const int hr = qdiv(v, geo.mWV);
                const int rem = v - hr * WV;
                const int j = rem >> LOG_VPR, cv = (rem & (VPR - 1)) * VEC;
                const int g = g0 - pad + hr;
                // 32-bit element offset from the (scalar) tensor base: a tensor of this library has < 2^31 elements (checked by the
                // launcher).  As size_t the NVH per-thread addresses were 64-bit register pairs, which pushed the <bf16, 1, *>
                // variants two registers over their 128 (an 8-byte spill reloaded in front of a request, r04 code object).
                const unsigned off = (unsigned)((min(max(g, 0), GR - 1) * W + j) * C + c0 + cv);
                rh[i] = *reinterpret_cast<const uint4*>(x + off);
                if constexpr (FOLD) ru[i] = *reinterpret_cast<const uint4*>(fxs + off);
                hmask |= ((unsigned)g < (unsigned)GR) ? (1u << i) : 0u;
            }
        }
    };
    // a thread always stages the same VEC channels of a chunk (256 % VPR == 0): BN coefficients live in registers
    const int cvh = (tid & (VPR - 1)) * VEC;
    const float relu_lo = a.bn.relu ? 0.f : -3.4e38f;
    auto halo_store = [&](int c0) {
        float psc[VEC], psh[VEC];
        if (a.bn.mode != FPD_BN_NONE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { psc[e] = s_scale[c0 + cvh + e]; psh[e] = s_shift[c0 + cvh + e]; }
        }
        float pfa[VEC], pfb[VEC], pfd[VEC];
        if (FOLD && fold) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                pfa[e] = s_fold[c0 + cvh + e]; pfb[e] = s_fold[C + c0 + cvh + e]; pfd[e] = s_fold[2 * C + c0 + cvh + e];
            }
        }
#pragma unroll
        for (int i = 0; i < NVH; ++i) {
            const int v = tid + i * 256;
            if (v < nvtot) {
                const int hr = qdiv(v, geo.mWV);
                const int j = (v - hr * WV) >> LOG_VPR;
                uint4 val = rh[i];
                if (FOLD && fold) {
                    const bool in = (hmask >> i) & 1u;
                    float g[VEC], u[VEC];
                    DT<T>::unpack(val, g);
                    DT<T>::unpack(in ? ru[i] : make_uint4(0, 0, 0, 0), u);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g[e] = fmaf(pfa[e], g[e], fmaf(pfb[e], u[e], pfd[e]));
                    val = DT<T>::pack(g);
                    if (!in) val = make_uint4(0, 0, 0, 0);                         // rows outside the tensor stay exactly zero
                    // written out once for the operand's other consumer (a separate weight-gradient launch): rows of this tile only
                    if (fo != nullptr && in && hr >= pad && hr < pad + nrows)
                        *reinterpret_cast<uint4*>(fo + ((size_t)((g0 - pad + hr) * W + j) * C + c0 + cvh)) = val;
                }
                if (a.bn.mode != FPD_BN_NONE) {
                    float f[VEC];
                    DT<T>::unpack(val, f);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) f[e] = fmaxf(fmaf(f[e], psc[e], psh[e]), relu_lo);
                    val = DT<T>::pack(f);
                }
                if (!((hmask >> i) & 1u)) val = make_uint4(0, 0, 0, 0);          // rows outside the tensor stay exactly zero
                *reinterpret_cast<uint4*>(sH + (hr * WP + j + pad) * LD + cvh) = val;
            }
        }
    };
    int b_loff[NVB];
    int b_goff[NVB];
    bool b_ok[NVB];
#pragma unroll
    for (int i = 0; i < NVB; ++i) {
        const int v = tid + i * 256;
        const int row = v >> LOG_VPR, col = (v & (VPR - 1)) * VEC;
        b_loff[i] = row * LD + col;
        b_ok[i] = (v < NVB_TOT) && (n0 + row < K);
        b_goff[i] = (n0 + row) * (R * R) * C + col;                  // + tap*C + c0 (wave-uniform) per tile
    }
    uint4 rb[NVB];
    const int RS = R * R;
    auto b_load = [&](int tap, int c0) {
        const int toff = tap * C + c0;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            rb[i] = make_uint4(0, 0, 0, 0);
            if (b_ok[i]) rb[i] = *reinterpret_cast<const uint4*>(w + (b_goff[i] + toff));
        }
    };
    auto b_store = [&](int buf) {
        T* dst = sB + buf * BNT * LD;
#pragma unroll
        for (int i = 0; i < NVB; ++i)
            if (tid + i * 256 < NVB_TOT) *reinterpret_cast<uint4*>(dst + b_loff[i]) = rb[i];
    };

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;

    const int nchunk = C / BK;
    // the first loads of the block go out before anything else (tables, zero fill and their fp64 arithmetic run under them)
    halo_load(0);
    if constexpr (!ALLW) b_load(0, 0);
    constexpr bool W9 = ALLW && NVB == 1;        // 9 vectors per thread (TN = 1); 18 (TN = 2) cost the fourth block per CU
    uint4 rw9[W9 ? 9 : 1][NVB];
    if constexpr (ALLW) {
        // nine weight tiles [BNT][BK] -> LDS buffers 0..8; then the halo; ONE barrier; 9 taps.  W9 (TN = 1: nine vectors per thread): all
        // nine are requested here and stored behind the table arithmetic below -- one memory latency, shared with the tables'
        // own loads (the stamps of r04 showed the three rounds of three as 3 x ~1.7 k cycles in front of everything else on
        // launches that are pure latency); three taps at a time otherwise
        if constexpr (W9) {
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int i = 0; i < NVB; ++i) {
                    rw9[u][i] = make_uint4(0, 0, 0, 0);
                    if (b_ok[i]) rw9[u][i] = *reinterpret_cast<const uint4*>(w + (b_goff[i] + u * C));
                }
        } else {
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                uint4 rw[3][NVB];
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int i = 0; i < NVB; ++i) {
                        rw[u][i] = make_uint4(0, 0, 0, 0);
                        if (b_ok[i]) rw[u][i] = *reinterpret_cast<const uint4*>(w + (b_goff[i] + (t3 * 3 + u) * C));
                    }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int i = 0; i < NVB; ++i)
                        if (tid + i * 256 < NVB_TOT) *reinterpret_cast<uint4*>(sB + (t3 * 3 + u) * BNT * LD + b_loff[i]) = rw[u][i];
            }
        }
    }
    TILE_STAMP();                        // 1: first loads requested
    // ---- one-time LDS initialisation: zero border columns + zero pixels; BN tables ----
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        const int nb = (R == 3) ? 2 * hrows : 0;
        for (int v = tid; v < (nb + 3) * VPR; v += 256) {
            const int pz = v >> LOG_VPR, cv = (v & (VPR - 1)) * VEC;
            const int px = pz < nb ? ((pz >> 1) * WP + ((pz & 1) ? WP - 1 : 0)) : zero_px + (pz - nb);
            *reinterpret_cast<uint4*>(sH + px * LD + cv) = z;
        }
    }
    TILE_STAMP();                        // 2: zero fill
    bn_fill(a.bn, C, (double)M, s_scale, s_shift);
    TILE_STAMP();                        // 3: BN tables
    if (FOLD && fold) {
        // dy = gamma*is*(g - m1 - xhat*m2), xhat = (u - mu)*is  ==  A g + B u + D   (coefficients formed in fp64);
        // the upper half of the block does it while the lower half fills the epilogue tables
        for (int c = tid - 128; c >= 0 && c < C; c += 128) {
            BnRaw r;
            bn_request(a.fold_bn, c, C, r);
            const double b1 = stats_sum(a.fold_stats, C, 0, c), b2 = stats_sum(a.fold_stats, C, 1, c);
            const double s1 = stat_resolve(r.s1), s2 = stat_resolve(r.s2);
            const double cnt = (double)M, mu = s1 / cnt;
            double var = s2 / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
            const double is = 1.0 / sqrt(var + (double)r.eps), gi = (double)r.g * is;
            const double m1 = b1 / cnt, m2 = b2 / cnt;
            s_fold[c] = (float)gi;
            s_fold[C + c] = (float)(-gi * is * m2);
            s_fold[2 * C + c] = (float)(gi * (mu * is * m2 - m1));
            if (bx == 0 && by == 0) {                     // the affine parameters' gradients fall out of the two sums
                if (a.fold_dgamma != nullptr) a.fold_dgamma[c] = (float)b2;
                if (a.fold_dbeta != nullptr) a.fold_dbeta[c] = (float)b1;
            }
        }
    }
    TILE_STAMP();                        // 4: fold tables
    conv_epi_tables<BNT>(a, n0, M, s_epi);
    TILE_STAMP();                        // 5: epilogue tables

    if constexpr (ALLW) {
        if constexpr (W9) {
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int i = 0; i < NVB; ++i)
                    if (tid + i * 256 < NVB_TOT) *reinterpret_cast<uint4*>(sB + u * BNT * LD + b_loff[i]) = rw9[u][i];
        }
        __syncthreads();                 // tables / zero fill visible to halo_store
        halo_store(0);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, sx = tap - 3 * r;
            const int ab = (r == 0) ? ab0 : ((r == 1) ? ab1 : ab2);
            TapMma<T>::template run<TN, BK, LD>(sH + ab + sx * LD, sB + tap * BNT * LD, lane, acc);
        }
    } else {
    for (int ch = 0; ch < nchunk; ++ch) {
        const int c0 = ch * BK;
        const bool more = ch + 1 < nchunk;
        __syncthreads();                 // previous chunk fully consumed (and, first time, tables/zero fill visible)
        TILE_STAMP();                    // per chunk: barrier | staged | barrier | taps
        halo_store(c0);
        b_store(0);
        TILE_STAMP();
        __syncthreads();
        TILE_STAMP();
        if (more) halo_load(c0 + BK);    // in flight during all taps of this chunk
        if (RS > 1) b_load(1, c0); else if (more) b_load(0, c0 + BK);
        int r = 0, s = 0;
        for (int tap = 0; tap < RS; ++tap) {
            const int ab = (r == 0) ? ab0 : ((r == 1) ? ab1 : ab2);
            TapMma<T>::template run<TN, BK, LD>(sH + ab + s * LD, sB + (tap & 1) * BNT * LD, lane, acc);
            if (tap + 1 < RS) {
                b_store((tap + 1) & 1);
                if (tap + 2 < RS) b_load(tap + 2, c0); else if (more) b_load(0, c0 + BK);
                __syncthreads();
            }
            if (++s == R) { s = 0; ++r; }
        }
        TILE_STAMP();
    }
    }
    const int Mlim = min(M, m0 + TPX);   // rows of the 128-row MFMA tile beyond the tile's pixels belong to the next tile
    if (K % VEC == 0) {
        conv_epilogue_vec<T, TN>(a, acc, m0, n0, Mlim, s_epi, stage, s_red);   // starts with a barrier
    } else {
        __syncthreads();                 // every wave is done reading the tile region before s_red (aliased) is written
        conv_epilogue<T, TN>(a, acc, m0 + wave * 32, n0, Mlim, s_epi, s_red);
    }
#ifdef FPD_TILE_TIMING
    TILE_STAMP();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TILE_STAMP();                        // every memory operation of thread 0 has returned
    __syncthreads();
    if (tid == 0 && bx == 0 && by == 0) {
        // entry | loads requested | zero fill | BN tables | fold tables | epi tables | per chunk: barrier, staged, barrier, taps |
        // epilogue: barrier, staged+barrier, rows stored, [stats shuffled+barrier, barrier, atomics issued] | end | memory drained
        printf("conv_tile R=%d C=%d K=%d HxW=%dx%d TN=%d BK=%d allw=%d fold=%d epi=%d bn=%d blocks=%dx%d:", R, C, K, H, W, TN, BK, (int)ALLW, (int)fold,
               a.epi, a.bn.mode, (int)gridDim.x, (int)gridDim.y);
        for (int q = 1; q < fpd_tile_ns; ++q) printf(" %lld", fpd_tile_stamp[q] - fpd_tile_stamp[0]);
        printf("\n");
    }
#endif
}

// Waves per SIMD the register allocation must leave room for (= blocks per CU: a block is one wave per SIMD).  The variants
// that fit 128 registers are held there -- the launches of a thousand blocks (student head 1x1 128->16 at 64x64) lose a quarter
// of their throughput with three blocks per CU instead of four (r04: 13.7 -> 17.8 us when the count crept to 130).
template <int TN, bool ALLW, bool FOLD>
constexpr int tile_waves() { return (!FOLD && (TN == 1 || (TN == 2 && ALLW))) ? 4 : 2; }

template <typename T, int TN, int BK, bool ALLW, bool FOLD = false>
__global__ __launch_bounds__(256, (tile_waves<TN, ALLW, FOLD>())) void conv_tile_kernel(const fpd_conv_t a, const TileGeo geo) {
    conv_tile_body<T, TN, BK, ALLW, FOLD>(a, geo, blockIdx.x, blockIdx.y);
}

// Two INDEPENDENT convolutions with the same tile configuration in one launch: pixel tiles [0, nbx_a) belong to `a`,
// the rest to `b` (block-uniform choice; the descriptors live in kernel-argument memory).  Used for the two parallel
// bottlenecks of an hourglass level (up-branch at full, low-branch at half resolution): one launch latency for both.
template <typename T, int TN, int BK, bool ALLW, bool FOLD = false>
__global__ __launch_bounds__(256, (tile_waves<TN, ALLW, FOLD>())) void conv_tile_pair_kernel(const fpd_conv_t a, const fpd_conv_t b, const TileGeo logWa,
                                                                const TileGeo logWb, const int nbx_a) {
    // `b` (the half-resolution, shorter job) gets the FIRST block indices: its blocks are dispatched up front and the
    // launch ends with a's normal tail instead of a's tail followed by b's
    const int nbx_b = (int)gridDim.x - nbx_a;
    if ((int)blockIdx.x < nbx_b) conv_tile_body<T, TN, BK, ALLW, FOLD>(b, logWb, blockIdx.x, blockIdx.y);
    else conv_tile_body<T, TN, BK, ALLW, FOLD>(a, logWa, (int)blockIdx.x - nbx_b, blockIdx.y);
}

// largest grid (blocks) that uses the all-taps-staged variant (the FPD_CONV_ALLW knob of rounds 4-5: 320 measured best)
static int allw_max_blocks() { return 320; }
template <typename T, int BK>
static bool allw_ok(const fpd_conv_t& a, int blocks) {
    return a.R == 3 && a.C == BK && blocks <= allw_max_blocks();
}

constexpr size_t LDS_MAX = 160 * 1024;
static int tile_rows(int W) { return std::max(1, 128 / W); }
static int tiles_of(const fpd_conv_t& a) { return cdiv(a.N * a.H, tile_rows(a.W)); }
template <typename T, int BK>
static TileGeo make_geo(const fpd_conv_t& a) {
    constexpr int VPR = BK / DT<T>::VEC;
    TileGeo g;
    g.nrows = tile_rows(a.W);
    g.mW = magic_of(a.W);
    g.mWV = magic_of(a.W * VPR);
    return g;
}
template <typename T, int TN, int BK, bool ALLW>
static size_t tile_lds(const fpd_conv_t& c) {
    constexpr int LD = BK + 16 / (int)sizeof(T);
    const int hrows = tile_rows(c.W) + c.R - 1, WP = c.W + c.R - 1;
    return (size_t)(hrows * WP + 3) * LD * sizeof(T) + (size_t)(ALLW ? 9 : 2) * 32 * TN * LD * sizeof(T);
}

template <typename T, int TN, int BK, bool ALLW>
int launch_tile_v(const fpd_conv_t& a, hipStream_t st) {
    const size_t epi = std::max((size_t)128 * (32 * TN + 4) * sizeof(float), (size_t)4 * 32 * TN * 2 * sizeof(double));
    const size_t lds = (size_t)(2 * a.C + 4 * 32 * TN + (a.epi == FPD_EPI_BNRELU_BWD ? 3 * a.C : 0)) * sizeof(float) + std::max(tile_lds<T, TN, BK, ALLW>(a), epi);
    if (lds > LDS_MAX) return 1;
    dim3 grid(tiles_of(a), cdiv(a.K, 32 * TN));
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&conv_tile_kernel<T, TN, BK, ALLW>), lds)) return rc_;
    if constexpr (TN <= 2) {
        if (a.fold_x != nullptr) {
            static LdsAttr configured_f;        // per device, set once (thread-safe: common.h)
            if (int rc_ = configured_f.ensure(reinterpret_cast<const void*>(&conv_tile_kernel<T, TN, BK, ALLW, true>), lds)) return rc_;
            FPD_LAUNCH((conv_tile_kernel<T, TN, BK, ALLW, true>), grid, dim3(256), lds, st, a, make_geo<T, BK>(a));
            return 0;
        }
    }
    if (a.fold_x != nullptr) return fpd_fail(-2, "conv_tile: a folded BN-backward apply is compiled for TN <= 2 only (fpd_conv_fold_supported)");
    FPD_LAUNCH((conv_tile_kernel<T, TN, BK, ALLW>), grid, dim3(256), lds, st, a, make_geo<T, BK>(a));
    return 0;
}
template <typename T, int TN, int BK>
int launch_tile(const fpd_conv_t& a, hipStream_t st) {
    if constexpr (BK == 64 || (BK == 32 && sizeof(T) == 4)) {
        if (allw_ok<T, BK>(a, tiles_of(a) * cdiv(a.K, 32 * TN))) {
            const int rc = launch_tile_v<T, TN, BK, true>(a, st);
            if (rc != 1) return rc;              // 1: the nine weight tiles do not fit the LDS next to the halo
        }
    }
    return launch_tile_v<T, TN, BK, false>(a, st);
}

// 32-channel output tiles per block: 4 / 2 / 1 by K, halved while the launch would have fewer than 128 blocks
static int tile_tn(int K, int mt) {
    int tn = K > 64 ? 4 : (K > 32 ? 2 : 1);
    while (tn > 1 && mt * cdiv(K, 32 * tn) < 128) tn >>= 1;       // small layers: more, shorter blocks
    return tn;
}

template <typename T, int BK>
int launch_tile_tn(const fpd_conv_t& a, hipStream_t st) {
    const int tn = tile_tn(a.K, tiles_of(a));
    if (tn == 4) return launch_tile<T, 4, BK>(a, st);
    if (tn == 2) return launch_tile<T, 2, BK>(a, st);
    return launch_tile<T, 1, BK>(a, st);
}

// shapes the halo-tile kernel covers: stride-1 "same" 1x1 / 3x3, rows of at most 128 pixels, C a multiple of 16
static bool tile_domain(const fpd_conv_t& a) {
    if ((long long)a.N * a.H * a.W * a.C >= (1ll << 31)) return false;        // 32-bit element offsets in the halo requests
    if (a.stride != 1 || a.R != a.S || (a.R != 1 && a.R != 3) || a.pad != (a.R - 1) / 2) return false;
    if (a.P != a.H || a.Q != a.W || a.W > 128 || a.W < 2) return false;
    if (a.C % 16 != 0 || a.C > FPD_MAXC) return false;
    if (a.epi == FPD_EPI_BNRELU_BWD && a.K > FPD_MAXC) return false;
    return true;
}
// the halo of one channel chunk must fit the 8 staging vectors a thread holds
static bool halo_fits(const fpd_conv_t& a, int vpr) { return (tile_rows(a.W) + a.R - 1) * a.W * vpr <= 2048; }
// channel chunk of a single bf16 launch: the widest of 64 / 32 / 16 that divides C and whose halo fits (0: none).  A 3x3
// convolution on 128-wide rows does not fit 64-channel chunks (3 rows x 128 pixels x 8 vectors): it runs on 32-channel
// chunks instead of falling through to the generic kernel (the frozen teacher's 3x3 64->64 at 128x128: 165 us there)
static int tile_bk_bf16(const fpd_conv_t& a) {
    for (int bk = 64; bk >= 16; bk >>= 1)
        if (a.C % bk == 0 && halo_fits(a, bk / 8)) return bk;
    return 0;
}

template <typename T, int TN, int BK, bool ALLW>
int launch_pair_v(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    const size_t epi = std::max((size_t)128 * (32 * TN + 4) * sizeof(float), (size_t)4 * 32 * TN * 2 * sizeof(double));
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD || b.epi == FPD_EPI_BNRELU_BWD;
    const size_t lds = (size_t)((bwd ? 5 : 2) * std::max(a.C, b.C) + 4 * 32 * TN) * sizeof(float) +
                       std::max({tile_lds<T, TN, BK, ALLW>(a), tile_lds<T, TN, BK, ALLW>(b), epi});
    if (lds > LDS_MAX) return 1;
    const int nbx_a = tiles_of(a), nbx_b = tiles_of(b);
    dim3 grid(nbx_a + nbx_b, cdiv(a.K, 32 * TN));
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&conv_tile_pair_kernel<T, TN, BK, ALLW>), lds)) return rc_;
    if constexpr (TN <= 2) {
        if (a.fold_x != nullptr || b.fold_x != nullptr) {
            static LdsAttr configured_f;        // per device, set once (thread-safe: common.h)
            if (int rc_ = configured_f.ensure(reinterpret_cast<const void*>(&conv_tile_pair_kernel<T, TN, BK, ALLW, true>), lds)) return rc_;
            FPD_LAUNCH((conv_tile_pair_kernel<T, TN, BK, ALLW, true>), grid, dim3(256), lds, st, a, b, make_geo<T, BK>(a), make_geo<T, BK>(b), nbx_a);
            return 0;
        }
    }
    if (a.fold_x != nullptr || b.fold_x != nullptr) return fpd_fail(-2, "conv_tile pair: a folded BN-backward apply is compiled for TN <= 2 only");
    FPD_LAUNCH((conv_tile_pair_kernel<T, TN, BK, ALLW>), grid, dim3(256), lds, st, a, b, make_geo<T, BK>(a), make_geo<T, BK>(b), nbx_a);
    return 0;
}
template <typename T, int TN, int BK>
int launch_pair(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    if constexpr (BK == 64 || (BK == 32 && sizeof(T) == 4)) {
        const int blocks = (tiles_of(a) + tiles_of(b)) * cdiv(a.K, 32 * TN);
        if (allw_ok<T, BK>(a, blocks)) {
            const int rc = launch_pair_v<T, TN, BK, true>(a, b, st);
            if (rc != 1) return rc;
        }
    }
    return launch_pair_v<T, TN, BK, false>(a, b, st);
}

template <typename T, int BK>
int launch_pair_tn(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    const int tn = tile_tn(a.K, tiles_of(a) + tiles_of(b));
    if (tn == 4) return launch_pair<T, 4, BK>(a, b, st);
    if (tn == 2) return launch_pair<T, 2, BK>(a, b, st);
    return launch_pair<T, 1, BK>(a, b, st);
}

}  // namespace

// 1 if a BNRELU_BWD data gradient without a prologue BN is served by this kernel WITH a folded BN-backward apply
// (fpd_conv_t.fold_x): the FOLD variants are compiled for TN <= 2 (register budget) and kept to channel counts for which
// the LDS never runs out.
static bool tile_fold_shape(const fpd_conv_t& a) {
    if (a.epi != FPD_EPI_BNRELU_BWD || a.bn.mode != FPD_BN_NONE || !tile_domain(a) || a.C > 128 || a.K > 128) return false;
    if (a.dtype == FPD_BF16) return tile_bk_bf16(a) != 0;
    return halo_fits(a, ((a.C % 32 == 0) ? 32 : 16) / 4);
}
int fpd_conv_tile_fold_ok(const fpd_conv_t& a) { return (tile_fold_shape(a) && tile_tn(a.K, tiles_of(a)) <= 2) ? 1 : 0; }
// -1: this kernel would not pair the two (the caller launches them one by one); else 1 / 0 as above for the pair launch
int fpd_conv_tile_pair_fold_ok(const fpd_conv_t& a, const fpd_conv_t& b) {
    if (!tile_domain(a) || !tile_domain(b) || a.dtype != b.dtype || a.K != b.K || a.C != b.C || a.R != b.R) return -1;
    const int vpr = a.dtype == FPD_BF16 ? ((a.C % 64 == 0) ? 64 : ((a.C % 32 == 0) ? 32 : 16)) / 8 : ((a.C % 32 == 0) ? 32 : 16) / 4;
    if (!halo_fits(a, vpr) || !halo_fits(b, vpr)) return -1;
    return (tile_fold_shape(a) && tile_fold_shape(b) && tile_tn(a.K, tiles_of(a) + tiles_of(b)) <= 2) ? 1 : 0;
}

// Two independent convolutions in one launch; 1 = the pair is outside the domain (caller launches them one by one).
// Requires both in the halo-tile domain with the same dtype, K (n-tiling), R and channel chunking.
int fpd_conv_tile_pair_launch(const fpd_conv_t& a, const fpd_conv_t& b, hipStream_t st) {
    if (!tile_domain(a) || !tile_domain(b)) return 1;
    if (a.dtype != b.dtype || a.K != b.K || a.C != b.C || a.R != b.R) return 1;
    if (a.dtype == FPD_BF16) {
        const int bk = (a.C % 64 == 0) ? 64 : ((a.C % 32 == 0) ? 32 : 16);
        if (!halo_fits(a, bk / 8) || !halo_fits(b, bk / 8)) return 1;
        if (bk == 64) return launch_pair_tn<bf16_t, 64>(a, b, st);
        if (bk == 32) return launch_pair_tn<bf16_t, 32>(a, b, st);
        return launch_pair_tn<bf16_t, 16>(a, b, st);
    }
    const int bk = (a.C % 32 == 0) ? 32 : 16;
    if (!halo_fits(a, bk / 4) || !halo_fits(b, bk / 4)) return 1;
    if (bk == 32) return launch_pair_tn<float, 32>(a, b, st);
    return launch_pair_tn<float, 16>(a, b, st);
}

// returns 1 when the shape is outside this kernel's domain (caller tries the generic MFMA kernel next)
int fpd_conv_tile_launch(const fpd_conv_t& a, hipStream_t st) {
    if (!tile_domain(a)) return 1;
    if (a.dtype == FPD_BF16) {
        const int bk = tile_bk_bf16(a);
        if (bk == 0) return 1;
        if (bk == 64) return launch_tile_tn<bf16_t, 64>(a, st);
        if (bk == 32) return launch_tile_tn<bf16_t, 32>(a, st);
        return launch_tile_tn<bf16_t, 16>(a, st);
    }
    const int bk = (a.C % 32 == 0) ? 32 : 16;
    if (!halo_fits(a, bk / 4)) return 1;
    if (bk == 32) return launch_tile_tn<float, 32>(a, st);
    return launch_tile_tn<float, 16>(a, st);
}
