// Shared epilogue of the implicit-GEMM conv kernels: each wave holds TN accumulators of a 32x32 MFMA tile
// (rows = 32 consecutive output pixels starting at m_wave, cols = output channels n0 + tn*32 ...).
//   y = acc + bias (+ residual); then either
//     - batch statistics of y for the next train-mode BN            (out_stats)
//     - ReLU mask of epi_x + the two BatchNorm-backward sums        (FPD_EPI_BNRELU_BWD)
// Statistics are accumulated in fp64 from the first add: var = E[x^2]-E[x]^2 must survive |mean| >> std; every sum inside
// the block is formed in a fixed order and the block's contribution leaves as exact integer limbs (common.h).
#pragma once
#include "common.h"

// s_epi: float[4][BNT] {scale, shift, mean, invstd} of epi_bn for this block's channels (BNRELU_BWD only)
// s_red: double[4][BNT][2] scratch (may alias tile buffers: the caller guarantees they are no longer read)
template <typename T, int TN>
__device__ __forceinline__ void conv_epilogue(const fpd_conv_t& a, const f32x16* acc, const int m_wave, const int n0,
                                              const int M, const float* s_epi, double* s_red) {
    constexpr int BNT = 32 * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K;
    T* __restrict__ y = reinterpret_cast<T*>(a.y);
    const T* res = reinterpret_cast<const T*>(a.residual);
    const T* ex = reinterpret_cast<const T*>(a.epi_x);
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD;
    const bool want_stats = (a.out_stats != nullptr) || bwd;
    const int col_l = lane & 31, rhalf = lane >> 5;
    double s1[TN], s2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        s1[tn] = 0.0; s2[tn] = 0.0;
        const int t = tn * 32 + col_l;
        const int k = n0 + t;
        const bool kok = k < K;
        const float bias = (a.bias != nullptr && kok) ? a.bias[k] : 0.f;
        float esc = 0.f, esh = 0.f, emu = 0.f, eis = 0.f;
        if (bwd) { esc = s_epi[t]; esh = s_epi[BNT + t]; emu = s_epi[2 * BNT + t]; eis = s_epi[3 * BNT + t]; }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * rhalf;
            const int m = m_wave + row;
            if (m < M && kok) {
                const size_t off = (size_t)m * K + k;
                float v = acc[tn][i] + bias;
                if (res != nullptr) v += DT<T>::ld(res + off);
                if (bwd) {
                    const float xv = DT<T>::ld(ex + off);
                    const float z = fmaf(xv, esc, esh);
                    v = (!a.epi_bn.relu || z > 0.f) ? v : 0.f;
                    const double vr = (double)DT<T>::rnd(v);
                    s1[tn] += vr;
                    s2[tn] += vr * (double)((xv - emu) * eis);
                } else if (want_stats) {
                    const double vr = (double)DT<T>::rnd(v);
                    s1[tn] += vr;
                    s2[tn] += vr * vr;
                }
                DT<T>::st(y + off, v);
            }
        }
    }
    if (want_stats) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const double t1 = s1[tn] + __shfl_xor(s1[tn], 32, 64);
            const double t2 = s2[tn] + __shfl_xor(s2[tn], 32, 64);
            if (lane < 32) { s_red[(wave * BNT + tn * 32 + lane) * 2 + 0] = t1; s_red[(wave * BNT + tn * 32 + lane) * 2 + 1] = t2; }
        }
        __syncthreads();
        fpd_stat_t* st = bwd ? a.epi_stats : a.out_stats;
        for (int t = tid; t < BNT; t += 256) {
            const int k = n0 + t;
            if (k < K) {
                double u1 = 0.0, u2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { u1 += s_red[(w * BNT + t) * 2]; u2 += s_red[(w * BNT + t) * 2 + 1]; }
                stat_atomic_add(st, K, 0, k, u1);
                stat_atomic_add(st, K, 1, k, u2);
            }
        }
    }
}

// fill s_epi for the BNRELU_BWD epilogue (call before a __syncthreads())
template <int BNT>
__device__ __forceinline__ void conv_epi_tables(const fpd_conv_t& a, const int n0, const int M, float* s_epi) {
    if (a.epi != FPD_EPI_BNRELU_BWD) return;
    for (int t = threadIdx.x; t < BNT; t += 256) {
        const int k = n0 + t;
        float sc = 0.f, sh = 0.f, mu = 0.f, is = 0.f;
        if (k < a.K) bn_coef(a.epi_bn, k, a.K, (double)M, sc, sh, mu, is);
        s_epi[t] = sc; s_epi[BNT + t] = sh; s_epi[2 * BNT + t] = mu; s_epi[3 * BNT + t] = is;
    }
}

// Vectorised epilogue (needs K % VEC == 0): the accumulators go through an fp32 LDS staging tile (all 128 rows
// at once) so that residual / epi_x are read and y is written as 16-byte vectors along the channel axis, each thread
// owning one channel-vector column.  Statistics: a thread sums SHIFTED values (v - c, c = its first value of that
// channel) in fp32 over its <= 16 rows -- no cancellation because c is within a few sigma of the mean -- and converts
// to the global {sum v, sum v^2} in fp64 once (sum v = S1 + n c, sum v^2 = S2 + 2 c S1 + n c^2); everything after
// that (cross-thread, cross-block) is fp64.  The BN-backward sums {dz, dz*xhat} have no such cancellation and are
// accumulated in fp32 per thread, fp64 beyond.
//   stage: >= 128*(32*TN+4) floats, 16-byte aligned; s_red: >= 4*32*TN*2 doubles (may alias stage)
// (r04, measured and dropped: a wave reading back only the rows it staged -- one block barrier less, no measurable gain, and the
// regrouped fp32 partial sums of the statistics move every bit-level regression pin of the repo (the trained-pair pin, the golden
// training curve); a staging tile of its own, i.e. no barrier at all in front of the staging stores -- no gain: the launches this
// would matter for are bound by instruction issue of one wave per SIMD, not by barriers; statistics added per wave instead of
// through the block-level reduction -- slower at every size: 16 limb splits per lane cost ~420 fp64 instructions, and from 128
// blocks on the same-address atomics serialise in the L2.)
template <typename T, int TN>
__device__ __forceinline__ void conv_epilogue_vec(const fpd_conv_t& a, const f32x16* acc, const int m0, const int n0,
                                                  const int M, const float* s_epi, float* stage, double* s_red) {
    constexpr int VEC = DT<T>::VEC;
    constexpr int BNT = 32 * TN, LDST = BNT + 4, CVN = BNT / VEC, RSTEP = 256 / CVN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.K;
    const int cv = tid % CVN, row0 = tid / CVN;
    const int k0 = n0 + cv * VEC;
    const bool kok = k0 < K;
    T* __restrict__ y = reinterpret_cast<T*>(a.y);
    const T* res = reinterpret_cast<const T*>(a.residual);
    const T* ex = reinterpret_cast<const T*>(a.epi_x);
    const bool bwd = a.epi == FPD_EPI_BNRELU_BWD;
    const bool want_stats = (a.out_stats != nullptr) || bwd;
    float bias[VEC], esc[VEC], esh[VEC], emu[VEC], eis[VEC];
    float f1[VEC], f2[VEC], cshift[VEC];
    int nrow = 0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        f1[e] = 0.f; f2[e] = 0.f; cshift[e] = 0.f;
        bias[e] = (a.bias != nullptr && kok) ? a.bias[k0 + e] : 0.f;
        esc[e] = esh[e] = emu[e] = eis[e] = 0.f;
        if (bwd) {
            const int t = cv * VEC + e;
            esc[e] = s_epi[t]; esh[e] = s_epi[BNT + t]; emu[e] = s_epi[2 * BNT + t]; eis[e] = s_epi[3 * BNT + t];
        }
    }
    const float relu_gate = a.epi_bn.relu ? 0.f : -3.4e38f;     // z > gate keeps the gradient
    const int col_l = lane & 31, rhalf = lane >> 5;
    // The rows a thread owns are processed in groups of G: the residual / epi_x vectors of a whole group are requested
    // together (one memory latency per group instead of one per row -- these launches are latency-bound), and the first
    // group's requests go out BEFORE the accumulators travel through the LDS staging tile.
    constexpr int NR = 128 / RSTEP, G = NR < 4 ? NR : 4, NG = NR / G;
    static_assert(NR % G == 0, "row groups");
    uint4 rres[G], rex[G];
    auto request = [&](int g) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int m = m0 + row0 + (g * G + i) * RSTEP;
            rres[i] = make_uint4(0, 0, 0, 0);
            rex[i] = make_uint4(0, 0, 0, 0);
            if (m < M && kok) {
                const size_t off = (size_t)m * K + k0;
                if (res != nullptr) rres[i] = *reinterpret_cast<const uint4*>(res + off);
                if (bwd) rex[i] = *reinterpret_cast<const uint4*>(ex + off);
            }
        }
    };
    request(0);
    __syncthreads();                                       // tile region free: every wave is past its last MFMA LDS read
    TILE_STAMP();
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * rhalf;
            stage[row * LDST + tn * 32 + col_l] = acc[tn][i];
        }
    __syncthreads();
    TILE_STAMP();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g > 0) request(g);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int row = row0 + (g * G + i) * RSTEP;
            const int m = m0 + row;
            if (m < M && kok) {
                float v[VEC];
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + row * LDST + cv * VEC + q * 4);
                    v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
                }
                const size_t off = (size_t)m * K + k0;
                if (res != nullptr) {
                    float r[VEC];
                    DT<T>::unpack(rres[i], r);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v[e] += r[e];
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += bias[e];
                const uint4 packed = DT<T>::pack(v);
                if (bwd) {
                    float xv[VEC], vr[VEC];
                    DT<T>::unpack(rex[i], xv);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float z = fmaf(xv[e], esc[e], esh[e]);
                        v[e] = (z > relu_gate) ? v[e] : 0.f;
                    }
                    const uint4 pk = DT<T>::pack(v);
                    DT<T>::unpack(pk, vr);                  // the stored (rounded) gradient is what gets summed
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        f1[e] += vr[e];
                        f2[e] = fmaf(vr[e], (xv[e] - emu[e]) * eis[e], f2[e]);
                    }
                    *reinterpret_cast<uint4*>(y + off) = pk;
                } else {
                    if (want_stats) {
                        float vr[VEC];
                        DT<T>::unpack(packed, vr);
                        if (nrow == 0) {
#pragma unroll
                            for (int e = 0; e < VEC; ++e) cshift[e] = vr[e];
                        }
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float d = vr[e] - cshift[e];
                            f1[e] += d;
                            f2[e] = fmaf(d, d, f2[e]);
                        }
                        ++nrow;
                    }
                    *reinterpret_cast<uint4*>(y + off) = packed;
                }
            }
        }
    }
    TILE_STAMP();
    if (want_stats) {
        double s1[VEC], s2[VEC];
        if constexpr (sizeof(T) == 2) {
            // bf16 build: the lanes of a wave that own the same channel vector are combined in fp32 first (shuffles of
            // 32-bit values, no per-lane fp64 arithmetic); fp64 starts at the per-wave partial.  The forward statistics use
            // a shift that is COMMON to those lanes -- lane cv's first value, broadcast -- so the shifted fp32 sums of the
            // <= 64 rows a wave covers carry no cancellation; un-shifting happens once, in fp64.
            float cs[VEC];
            float nr = (float)nrow;
#pragma unroll
            for (int e = 0; e < VEC; ++e) cs[e] = bwd ? 0.f : __shfl(cshift[e], lane % CVN, 64);
            if (!bwd) {
                // re-base this lane's sums from its own shift to the common one: sum (v-cs) = f1 + n d, sum (v-cs)^2 =
                // f2 + 2 d f1 + n d^2 with d = c - cs (small: both are values of the same channel)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float d = cshift[e] - cs[e];
                    f2[e] = f2[e] + 2.f * d * f1[e] + nr * d * d;
                    f1[e] = f1[e] + nr * d;
                }
            }
#pragma unroll
            for (int o = CVN; o < 64; o <<= 1) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    f1[e] += __shfl_xor(f1[e], o, 64);
                    f2[e] += __shfl_xor(f2[e], o, 64);
                }
                nr += __shfl_xor(nr, o, 64);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const double c = (double)cs[e], n = (double)nr;
                s1[e] = (double)f1[e] + n * c;
                s2[e] = bwd ? (double)f2[e] : (double)f2[e] + 2.0 * c * (double)f1[e] + n * c * c;
            }
        } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (bwd) {
                s1[e] = (double)f1[e]; s2[e] = (double)f2[e];
            } else {
                const double c = (double)cshift[e], n = (double)nrow;
                s1[e] = (double)f1[e] + n * c;
                s2[e] = (double)f2[e] + 2.0 * c * (double)f1[e] + n * c * c;
            }
#pragma unroll
            for (int o = CVN; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o, 64);
                s2[e] += __shfl_xor(s2[e], o, 64);
            }
        }
        }
        fpd_stat_t* st = bwd ? a.epi_stats : a.out_stats;
        __syncthreads();                                   // staging tile no longer read: s_red may alias it
        TILE_STAMP();
        if (lane < CVN) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s_red[(wave * BNT + cv * VEC + e) * 2 + 0] = s1[e];
                s_red[(wave * BNT + cv * VEC + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        TILE_STAMP();
        for (int t = tid; t < BNT; t += 256) {
            const int k = n0 + t;
            if (k < K) {
                double u1 = 0.0, u2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { u1 += s_red[(w * BNT + t) * 2]; u2 += s_red[(w * BNT + t) * 2 + 1]; }
                stat_atomic_add(st, K, 0, k, u1);
                stat_atomic_add(st, K, 1, k, u2);
            }
        }
        TILE_STAMP();
    }
}
