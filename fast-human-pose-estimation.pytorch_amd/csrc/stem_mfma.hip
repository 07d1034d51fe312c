// bf16 stem on the matrix cores: Conv2d(3, K, 7, stride 2, pad 3) (/root/reference/lib/models/hourglass.py:116,172)
// and its weight gradient as GEMMs over an im2col tile built in LDS.
//   tile = 128 consecutive output pixels (whole output rows, Q a power of two <= 128); the fp32 NCHW input patch
//   ((2*rows+5) x (2Q+5) x 3) is staged in LDS, expanded to A[pixel][tap] (147 taps padded to 160, bf16);
//   forward : y[pixel][k]  = A[pixel][:] . W[k][:]          A-operand rows = pixels (plain ds_read_b128)
//   wgrad   : dW[k][tap]  += dy[pixel][k]^T A[pixel][tap]   both operands through the transposing LDS read
// fp32 builds keep the direct kernels of stem.hip.
#include <algorithm>

#include "common.h"
#include "conv_epilogue.h"
#include "mfma_frag.h"

namespace {

constexpr int NTAP = 147, KP = 160, LDA = KP + 8;     // padded taps; LDS row pitch (336 B)

// patch[c][pr][pc] <- x[n][c][2*prow0 - 3 + pr][-3 + pc]   (zero outside the image); one wave per patch row, lanes
// along the row: coalesced, no per-element division.  All global loads of a wave are issued before the first LDS
// store (a load -> store loop pays one memory latency per iteration: ~25 serial latencies per block, the old 314 us).
template <int RMAX, int CMAX>       // rows per wave (3*prows over 4 waves), 64-lane steps per row (pcols)
__device__ __forceinline__ void stage_patch_t(float* patch, const float* x, int n, int H, int W, int prow0, int prows,
                                              int pcols, int tid, int nthreads) {
    const int lane = tid & 63, wv = tid >> 6, nw = nthreads >> 6;
    float v[RMAX][CMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int rr = wv + i * nw;
        const int c = rr / prows, pr = rr - c * prows;
        const int ih = 2 * prow0 - 3 + pr;
        const bool rok = rr < 3 * prows && (unsigned)ih < (unsigned)H;
        const float* src = x + ((size_t)(n * 3 + (rok ? c : 0)) * H + (rok ? ih : 0)) * W;
#pragma unroll
        for (int j = 0; j < CMAX; ++j) {
            const int iw = lane + 64 * j - 3;
            v[i][j] = (rok && lane + 64 * j < pcols && (unsigned)iw < (unsigned)W) ? src[iw] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
        const int rr = wv + i * nw;
        if (rr < 3 * prows) {
#pragma unroll
            for (int j = 0; j < CMAX; ++j)
                if (lane + 64 * j < pcols) patch[rr * pcols + lane + 64 * j] = v[i][j];
        }
    }
}
__device__ __forceinline__ void stage_patch(float* patch, const float* x, int n, int H, int W, int prow0, int prows,
                                            int pcols, int tid, int nthreads, int logQ) {
    // Q = 128: 21 rows x 261 cols; 64: 27 x 133; 32: 39 x 69; 16: 63 x 37   (256 threads = 4 waves)
    if (logQ == 7) stage_patch_t<6, 5>(patch, x, n, H, W, prow0, prows, pcols, tid, nthreads);
    else if (logQ == 6) stage_patch_t<7, 3>(patch, x, n, H, W, prow0, prows, pcols, tid, nthreads);
    else if (logQ == 5) stage_patch_t<10, 2>(patch, x, n, H, W, prow0, prows, pcols, tid, nthreads);
    else stage_patch_t<16, 1>(patch, x, n, H, W, prow0, prows, pcols, tid, nthreads);
}

// A[pixel][tap] (bf16, two taps per dword) from the patch; tap = (r*7+s)*3+c, taps >= 147 are zero
// s_off[tap] = patch offset of tap (r,s,c) relative to the pixel's top-left input; taps >= 147 point at a zero word
__device__ __forceinline__ void build_tap_table(int* s_off, int prows, int pcols, int zero_idx, int tid, int nthreads) {
    const int per_c = prows * pcols;
    for (int tap = tid; tap < KP; tap += nthreads) {
        const int c = tap % 3, s = (tap / 3) % 7, r = tap / 21;
        s_off[tap] = tap < NTAP ? c * per_c + r * pcols + s : zero_idx;
    }
}
// A[pixel][tap] (bf16, two taps per dword).  Thread owns a fixed tap pair (80 pairs) and walks pixels.
__device__ __forceinline__ void build_im2col(bf16_t* sA, const float* patch, const int* s_off, int logQ, int pcols,
                                             int zero_idx, int tid, int nthreads) {
    const int Q = 1 << logQ;
    // nthreads = 256: pair = tid % 80 would need a division per thread only; use 240 active threads = 3 pixels x 80 pairs
    const int pair = tid % (KP / 2), pl = tid / (KP / 2), np = nthreads / (KP / 2);
    if (pl >= np) return;
    const int o0 = s_off[2 * pair], o1 = s_off[2 * pair + 1];
    const bool z0 = o0 == zero_idx, z1 = o1 == zero_idx;
    for (int pix = pl; pix < 128; pix += np) {
        const int base = (2 * (pix >> logQ)) * pcols + 2 * (pix & (Q - 1));
        const float v0 = patch[z0 ? zero_idx : o0 + base], v1 = patch[z1 ? zero_idx : o1 + base];
        *reinterpret_cast<uint32_t*>(sA + pix * LDA + 2 * pair) = f2bf_pk(v0, v1);
    }
}

template <int TN>
__global__ __launch_bounds__(256) void stem_fwd_mfma_kernel(const fpd_stem_t a, const int logQ) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, P = a.P, Q = a.Q;
    const int rows = 128 >> logQ, prows = 2 * rows + 5, pcols = 2 * Q + 5;
    const int M = a.N * P * Q;
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                 // [128][LDA]  (reused by the epilogue)
    bf16_t* sW = sA + 128 * LDA;                                  // [32*TN][LDA]
    float* patch = reinterpret_cast<float*>(sW + 32 * TN * LDA);  // [3][prows][pcols] + one zero word
    const int zero_idx = 3 * prows * pcols;
    int* s_off = reinterpret_cast<int*>(patch + zero_idx + 4);     // [160]
    if (tid == 0) patch[zero_idx] = 0.f;
    build_tap_table(s_off, prows, pcols, zero_idx, tid, 256);
    for (int i = tid; i < 128 * 4; i += 256) *reinterpret_cast<uint32_t*>(sA + (i >> 2) * LDA + KP + 2 * (i & 3)) = 0u;
    const int m0 = blockIdx.x * 128;
    const int g0 = m0 >> logQ;                                     // flattened (n, output row)
    const int n = g0 / P, prow0 = g0 - n * P;                      // P % rows == 0 (host-checked): one image per tile

    // weights [K][147] fp32 -> bf16 [32*TN][160]: the flat fp32 array is fetched as 16-byte vectors, all of them in
    // flight before the patch loads are issued; converted and scattered to [k][tap] afterwards (K % 4 == 0: host-checked)
    constexpr int NWV = (32 * TN * NTAP / 4 + 255) / 256;
    float4 wv[NWV];
    const int nwvec = K * NTAP / 4;
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        const int idx = tid + i * 256;
        wv[i] = idx < nwvec ? reinterpret_cast<const float4*>(a.w)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int i = tid; i < 32 * TN * 8; i += 256) {                 // zero taps 144..159 of every row (147.. stay zero)
        const int k = i >> 3;
        *reinterpret_cast<uint32_t*>(sW + k * LDA + 144 + 2 * (i & 7)) = 0u;
    }
    for (int i = tid; i < (32 * TN - K) * (KP / 2); i += 256) {   // rows beyond K
        const int k = K + i / (KP / 2), tp = (i % (KP / 2)) * 2;
        *reinterpret_cast<uint32_t*>(sW + k * LDA + tp) = 0u;
    }
    stage_patch(patch, a.x, n, a.H, a.W, prow0, prows, pcols, tid, 256, logQ);
    __syncthreads();                                                // zero fills above are ordered before the scatter
#pragma unroll
    for (int i = 0; i < NWV; ++i) {
        const int idx = tid + i * 256;
        if (idx < nwvec) {
            const float f[4] = {wv[i].x, wv[i].y, wv[i].z, wv[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fl = idx * 4 + e;
                const int k = fl / NTAP, tp = fl - k * NTAP;
                sW[k * LDA + tp] = f2bf(f[e]);
            }
        }
    }
    build_im2col(sA, patch, s_off, logQ, pcols, zero_idx, tid, 256);
    __syncthreads();

    f32x16 acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tn][e] = 0.f;
    const int koff = 8 * (lane >> 5);
    const bf16_t* arow = sA + (wave * 32 + (lane & 31)) * LDA + koff;
    const bf16_t* brow = sW + (lane & 31) * LDA + koff;
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(arow + kk * 16);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(brow + tn * 32 * LDA + kk * 16);
            acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[tn], 0, 0, 0);
        }
    }
    fpd_conv_t c;                      // reuse the conv epilogue: bias, statistics of y, 16-byte NHWC stores
    c.K = K; c.bias = a.bias; c.residual = nullptr; c.y = a.y; c.out_stats = a.out_stats; c.epi = FPD_EPI_PLAIN;
    c.epi_x = nullptr; c.epi_stats = nullptr; c.epi_bn.relu = 0;
    conv_epilogue_vec<bf16_t, TN>(c, acc, m0, 0, M, nullptr, reinterpret_cast<float*>(sA), reinterpret_cast<double*>(sA));
}

// grid-stride over pixel tiles; dW[k][tap] accumulated in registers, one flush per block (to its slab).
// KMAX = 32 (the hg4x128 student: K = 32): the dy tile shrinks to [128][40], which takes the block from 83 KB to 75 KB of LDS --
// TWO blocks per CU instead of one.  The launch is the last weight gradient of a backward, exposed in front of Adam.
template <int KMAX>
__global__ __launch_bounds__(256) void stem_wgrad_mfma_kernel(const fpd_stem_t a, const int logQ, const int mtiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int LDD = KMAX + 8;                                  // dy tile pitch (K <= KMAX)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K, P = a.P, Q = a.Q;
    const int rows = 128 >> logQ, prows = 2 * rows + 5, pcols = 2 * Q + 5;
    bf16_t* sA = reinterpret_cast<bf16_t*>(smem);                 // [128][LDA]
    bf16_t* sD = sA + 128 * LDA;                                  // [128][LDD]
    float* patch = reinterpret_cast<float*>(sD + 128 * LDD);
    const int zero_idx = 3 * prows * pcols;
    int* s_off = reinterpret_cast<int*>(patch + zero_idx + 4);
    if (tid == 0) patch[zero_idx] = 0.f;
    build_tap_table(s_off, prows, pcols, zero_idx, tid, 256);
    const bf16_t* dy = reinterpret_cast<const bf16_t*>(a.dy);
    const int KT = (K + 31) / 32;                                  // 1 or 2 cout tiles
    // wave w owns tap tiles {w, w+4} (5 tiles of 32 taps) x all cout tiles: <= 2 x 2 accumulators
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;
    for (int i = tid; i < 128 * LDD / 8; i += 256) reinterpret_cast<uint4*>(sD)[i] = make_uint4(0, 0, 0, 0);

    for (int tile = blockIdx.x; tile < mtiles; tile += gridDim.x) {
        const int g0 = tile * rows;
        const int n = g0 / P, prow0 = g0 - n * P;
        __syncthreads();
        stage_patch(patch, a.x, n, a.H, a.W, prow0, prows, pcols, tid, 256, logQ);
        const int vpr = K / 8;                                     // 16-byte vectors per dy pixel
        for (int v = tid; v < 128 * vpr; v += 256) {
            const int px = v / vpr, cv = (v - px * vpr) * 8;
            *reinterpret_cast<uint4*>(sD + px * LDD + cv) =
                *reinterpret_cast<const uint4*>(dy + ((size_t)tile * 128 + px) * K + cv);
        }
        __syncthreads();
        build_im2col(sA, patch, s_off, logQ, pcols, zero_idx, tid, 256);
        __syncthreads();
        if (a.dbias != nullptr && tid < K) {
            float s0 = 0.f;
            for (int p = 0; p < 128; ++p) s0 += bf2f(sD[p * LDD + tid]);
            bsum += s0;
        }
#pragma unroll 1
        for (int p0 = 0; p0 < 128; p0 += 16) {
            bf16x8 df[2];
#pragma unroll
            for (int j = 0; j < KMAX / 32; ++j) df[j] = tr_frag_bf16(sD, LDD, p0, j * 32, lane);     // dy^T: rows = couts
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tt = wave + 4 * i;                      // tap tile
                if (tt < KP / 32) {
                    const bf16x8 af = tr_frag_bf16(sA, LDA, p0, tt * 32, lane);              // A^T: cols = taps
#pragma unroll
                    for (int j = 0; j < KMAX / 32; ++j)
                        if (j < KT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[j], af, acc[i][j], 0, 0, 0);
                }
            }
        }
    }
    // no atomics: persistent block b stores its accumulator to slab b (a.partial; fpd_wgrad_reduce() adds the slabs in a
    // fixed order), or -- no slabs: the launch is a single block -- adds it into dw directly
    float* slab = a.partial != nullptr ? a.partial + (size_t)blockIdx.x * a.partial_stride : nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int tt = wave + 4 * i;
        if (tt < KP / 32) {
            const int tap = tt * 32 + (lane & 31);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = j * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                    if (j < KT && k < K && tap < NTAP) {
                        if (slab != nullptr) slab[(size_t)k * NTAP + tap] = acc[i][j][e];
                        else a.dw[(size_t)k * NTAP + tap] += acc[i][j][e];
                    }
                }
        }
    }
    if (a.dbias != nullptr && tid < K) {
        if (slab != nullptr) slab[(size_t)K * NTAP + tid] = bsum;
        else a.dbias[tid] += bsum;
    }
}

size_t patch_bytes(int logQ) {
    const int Q = 1 << logQ, rows = 128 >> logQ;
    return (size_t)3 * (2 * rows + 5) * (2 * Q + 5) * sizeof(float) + 16 + KP * sizeof(int);
}

}  // namespace

static bool stem_mfma_ok(const fpd_stem_t& a, int& logQ) {
    if (a.dtype != FPD_BF16 || a.K % 8 != 0 || a.K > 64) return false;
    if (a.Q > 128 || a.Q < 16 || (a.Q & (a.Q - 1)) != 0 || a.K % 4 != 0) return false;
    logQ = 0;
    while ((1 << logQ) < a.Q) ++logQ;
    if (a.P % (128 >> logQ) != 0) return false;
    return true;
}

// return 1 = not applicable (caller uses the direct kernels)
int fpd_stem_forward_mfma_launch(const fpd_stem_t& a, hipStream_t st) {
    int logQ;
    if (!stem_mfma_ok(a, logQ)) return 1;
    const int TN = a.K > 32 ? 2 : 1;
    const size_t tile = (size_t)(128 + 32 * TN) * LDA * sizeof(bf16_t);
    const size_t lds = std::max(tile, (size_t)128 * (32 * TN + 4) * sizeof(float)) + patch_bytes(logQ);
    const int tiles = a.N * a.P * a.Q / 128;
    static LdsAttr cfg1, cfg2;          // per device, set once (thread-safe: common.h)
    if (TN == 1) {
        if (int rc_ = cfg1.ensure(reinterpret_cast<const void*>(&stem_fwd_mfma_kernel<1>), lds)) return rc_;
        FPD_LAUNCH((stem_fwd_mfma_kernel<1>), dim3(tiles), dim3(256), lds, st, a, logQ);
    } else {
        if (int rc_ = cfg2.ensure(reinterpret_cast<const void*>(&stem_fwd_mfma_kernel<2>), lds)) return rc_;
        FPD_LAUNCH((stem_fwd_mfma_kernel<2>), dim3(tiles), dim3(256), lds, st, a, logQ);
    }
    return 0;
}

int fpd_stem_wgrad_mfma_partials(const fpd_stem_t& a) {
    int logQ;
    if (!stem_mfma_ok(a, logQ)) return 0;
    return std::min(a.N * a.P * a.Q / 128, 512);
}

int fpd_stem_wgrad_mfma_launch(const fpd_stem_t& a, hipStream_t st) {
    int logQ;
    if (!stem_mfma_ok(a, logQ)) return 1;
    const int kmax = a.K <= 32 ? 32 : 64;
    const size_t lds = (size_t)128 * LDA * sizeof(bf16_t) + (size_t)128 * (kmax + 8) * sizeof(bf16_t) + patch_bytes(logQ);
    const int tiles = a.N * a.P * a.Q / 128;
    const dim3 grid(a.partial != nullptr ? std::min(tiles, 512) : 1);
    static LdsAttr cfg32, cfg64;        // per device, set once (thread-safe: common.h)
    if (kmax == 32) {
        if (int rc_ = cfg32.ensure(reinterpret_cast<const void*>(&stem_wgrad_mfma_kernel<32>), lds)) return rc_;
        FPD_LAUNCH((stem_wgrad_mfma_kernel<32>), grid, dim3(256), lds, st, a, logQ, tiles);
    } else {
        if (int rc_ = cfg64.ensure(reinterpret_cast<const void*>(&stem_wgrad_mfma_kernel<64>), lds)) return rc_;
        FPD_LAUNCH((stem_wgrad_mfma_kernel<64>), grid, dim3(256), lds, st, a, logQ, tiles);
    }
    return 0;
}
