// head_fused: the inter-stack head of a FROZEN hourglass stack in one launch (hourglass.py:134-137,184-190, eval BN):
//     y     = fc(y0)                          1x1 C->C (+bias)
//     a     = relu(bn(y))                     kept in LDS as a bf16 image, never written to HBM
//     score = score_conv(a)                   1x1 C->J (+bias)        -> HBM (the stack's heat-map) and LDS
//     next  = x + fc_(a) + score_(score)      1x1 C->C, 1x1 J->C (+biases), x = the stack's input     -> HBM
// Four conv launches read/write ~550 MB per 64x64 stack (y, a twice, the residual chain); fused: y0 + x in, next + score
// out (~205 MB).  Same building blocks as bneck_fused.hip: 128-pixel tiles, 8 wave64 (4 pixel groups x 2 channel
// halves), "transposed" MFMAs (weights first) so a lane owns 4 consecutive channels of one pixel, weight tiles [C][64]
// double-buffered in LDS with two register sets in flight, persistent capped grid, wave-private staged epilogue.
// Domain: bf16, C = 256, J = 16 (K of score_ = one MFMA k-step).  Specification: oracle/plan_interp.py run_head.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// folded tables: sc[C], sh'[C] = sh + sc*b_fc | b_score[32] (zero padded) | b_out[C] = b_fc_ + b_score_
template <int C>
__device__ __forceinline__ void head_tables(const fpd_head_t& a, float* out, int tid, int nthreads) {
    for (int c = tid; c < C; c += nthreads) {
        float sc, sh;
        bn_coef_eval(a.bn, c, sc, sh);
        out[c] = sc;
        out[C + c] = fmaf(sc, a.b_fc ? a.b_fc[c] : 0.f, sh);
        out[2 * C + 32 + c] = (a.b_fc2 ? a.b_fc2[c] : 0.f) + (a.b_score2 ? a.b_score2[c] : 0.f);
    }
    for (int c = tid; c < 32; c += nthreads) out[2 * C + c] = (c < a.J && a.b_score) ? a.b_score[c] : 0.f;
}

#ifdef FPD_HEAD_TIMING      // probe build only (tools/probes): cycle stamps of block 0 at the phase boundaries of its LAST tile
#define HSTAMP() do { if (tid == 0 && blockIdx.x == 0 && hn < 24) hst[hn++] = clock64(); } while (0)
#else
#define HSTAMP() do { } while (0)
#endif

template <int C>
__global__ __launch_bounds__(512, 1) void head_eval_kernel(const fpd_head_t a, const int ntiles) {
    constexpr int LDX = 64 + 8;                 // [.][64]-chunk rows (bf16 elements)
    constexpr int LDA = C + 8;                  // activation image rows
    constexpr int LDS_ = 24;                    // score image rows (16 channels + pad)
    constexpr int NCH = C / 64;                 // K chunks of fc / fc_
    constexpr int NSTEP = 2 * NCH + 1;          // fc chunks | fc_ chunks | score_
    constexpr int WV = C * 8 / 512;             // uint4 per thread of a [C][64] weight chunk
    constexpr int CW = 64;                      // channels per wave per epilogue round (a wave owns 128 = 2 rounds)
    constexpr int WTILE = C * LDX;              // one weight buffer (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef FPD_HEAD_TIMING
    long long hst[24]; int hn = 0; int ntl = 0;
#endif
    const int q = wave & 3, hC = wave >> 2;
    const int koff = 8 * (lane >> 5);
    const int M = a.N * a.H * a.W, J = a.J;

    float* s_sc = reinterpret_cast<float*>(smem);
    float* s_sh = s_sc + C;
    float* s_bs = s_sh + C;                      // [32]
    float* s_bo = s_bs + 32;                     // [C]
    bf16_t* sA = reinterpret_cast<bf16_t*>(s_bo + C);        // [128][LDA]; during GEMM1: two [128][LDX] staging buffers
    bf16_t* sS = sA + 128 * LDA;                 // [128][LDS_]
    bf16_t* sW = sS + 128 * LDS_;                // two [C][LDX] weight buffers (GEMM2: [32][LDA] in the second)
    const bf16_t* __restrict__ y0 = reinterpret_cast<const bf16_t*>(a.y0);
    const bf16_t* __restrict__ xres = reinterpret_cast<const bf16_t*>(a.x);
    const bf16_t* __restrict__ wfc = reinterpret_cast<const bf16_t*>(a.w_fc);
    const bf16_t* __restrict__ wsc = reinterpret_cast<const bf16_t*>(a.w_score);
    const bf16_t* __restrict__ wfc2 = reinterpret_cast<const bf16_t*>(a.w_fc2);
    const bf16_t* __restrict__ wsc2 = reinterpret_cast<const bf16_t*>(a.w_score2);
    const bool has_next = a.next != nullptr;

    if (a.folded != nullptr) {
        for (int v = tid; v < (3 * C + 32) / 4; v += 512)
            reinterpret_cast<f32x4*>(s_sc)[v] = reinterpret_cast<const f32x4*>(a.folded)[v];
    } else {
        head_tables<C>(a, s_sc, tid, 512);
    }

    // hoisted addressing
    const int xpx = tid >> 3, xcv = (tid & 7) * 8;
    int wo[WV], wl[WV];
#pragma unroll
    for (int i = 0; i < WV; ++i) {
        const int v = tid + i * 512;
        wo[i] = (v >> 3) * C + (v & 7) * 8;
        wl[i] = (v >> 3) * LDX + (v & 7) * 8;
    }
    u32x4 rx[NCH][2], rb[2][WV];
    int m0 = 0;
    auto a_load = [&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = decltype(kcc)::value;
        static_for<2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int m = m0 + xpx + 64 * i;
            const u32x4 z = {0u, 0u, 0u, 0u};
            rx[kc][i] = z;
            if (m < M) rx[kc][i] = *reinterpret_cast<const u32x4*>(y0 + ((size_t)m * C + kc * 64 + xcv));
        });
    };
    auto a_store = [&](auto kcc) __attribute__((always_inline)) {
        constexpr int kc = decltype(kcc)::value;
        bf16_t* dst = sA + (kc & 1) * 128 * LDX;
        static_for<2>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            *reinterpret_cast<u32x4*>(dst + (xpx + 64 * i) * LDX + xcv) = rx[kc][i];
        });
    };
    // weight step s: 0..NCH-1 = fc chunk s, NCH..2NCH-1 = fc_ chunk, 2NCH = score_ ([C][J], J = 16)
    auto t_load = [&](auto sc_) __attribute__((always_inline)) {
        constexpr int s = decltype(sc_)::value;
        if (s >= NCH && !has_next) return;           // last stack: no fc_ / score_ weights exist
        if constexpr (s < 2 * NCH) {
            const bf16_t* w = s < NCH ? wfc : wfc2;
            constexpr int c0 = (s % NCH) * 64;
            static_for<WV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                rb[s & 1][i] = *reinterpret_cast<const u32x4*>(w + (wo[i] + c0));
            });
        } else {
            rb[s & 1][0] = *reinterpret_cast<const u32x4*>(wsc2 + ((tid >> 1) * 16 + (tid & 1) * 8));
        }
    };
    auto t_store = [&](auto sc_) __attribute__((always_inline)) {
        constexpr int s = decltype(sc_)::value;
        bf16_t* dst = sW + (s & 1) * WTILE;
        if constexpr (s < 2 * NCH) {
            static_for<WV>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                *reinterpret_cast<u32x4*>(dst + wl[i]) = rb[s & 1][i];
            });
        } else {
            *reinterpret_cast<u32x4*>(dst + (tid >> 1) * LDX + (tid & 1) * 8) = rb[s & 1][0];
        }
    };

    constexpr int VW = CW / 8, NIT = 32 * VW / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int t = ((ntiles & 7) == 0) ? (tile & 7) * (ntiles >> 3) + (tile >> 3) : tile;
        m0 = t * 128;
#ifdef FPD_HEAD_TIMING
        hn = 0; ++ntl;
#endif
        HSTAMP();
        static_for<NCH>([&](auto kcc) { a_load(kcc); });
        t_load(std::integral_constant<int, 0>{});
        t_load(std::integral_constant<int, 1>{});
        __syncthreads();                         // tables visible; previous tile's epilogue done with the LDS
        a_store(std::integral_constant<int, 0>{});
        t_store(std::integral_constant<int, 0>{});
        t_load(std::integral_constant<int, 2>{});
        __syncthreads();
        HSTAMP();

        const int ml = q * 32 + (lane & 31);
        f32x16 acc[4];
        u32x4 rres[2][NIT];
        static_for<NSTEP>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if constexpr (s == 0 || s == NCH) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[tn][i] = 0.f;
            }
            if (s < NCH || has_next) {
                const bf16_t* wrow = sW + (s & 1) * WTILE + (hC * 128 + (lane & 31)) * LDX + koff;
                const bf16_t* arow = s < NCH ? sA + (s & 1) * 128 * LDX + ml * LDX + koff
                                   : (s < 2 * NCH ? sA + ml * LDA + (s - NCH) * 64 + koff : sS + ml * LDS_ + koff);
                constexpr int NKK = s < 2 * NCH ? 4 : 1;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 px = *reinterpret_cast<const bf16x8*>(arow + kk * 16);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wrow + tn * 32 * LDX + kk * 16);
                        acc[tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, px, acc[tn], 0, 0, 0);
                    }
                }
            }
            if constexpr (s + 1 < NSTEP) {
                if (s + 1 < NCH || has_next) {
                    if constexpr (s + 1 < NCH) a_store(std::integral_constant<int, (s + 1 < NCH ? s + 1 : 0)>{});
                    t_store(std::integral_constant<int, s + 1>{});
                    if constexpr (s + 3 < NSTEP) t_load(std::integral_constant<int, (s + 3 < NSTEP ? s + 3 : 0)>{});
                }
                if constexpr (s == NSTEP - 3 || s == NSTEP - 2) {       // residual rows under the last steps
                    if (has_next) {
                        constexpr int st = s - (NSTEP - 3);
                        static_for<NIT>([&](auto itc) {
                            constexpr int it = decltype(itc)::value;
                            const int idx = lane + 64 * it;
                            const int px = idx / VW, cv = (idx - px * VW) * 8;
                            const int m = m0 + q * 32 + px;
                            const u32x4 z = {0u, 0u, 0u, 0u};
                            rres[st][it] = z;
                            if (m < M) rres[st][it] = *reinterpret_cast<const u32x4*>(xres + ((size_t)m * C + hC * 128 + st * CW + cv));
                        });
                    }
                }
                __syncthreads();
                HSTAMP();
            }
            if constexpr (s == NCH - 1) {
                // ---- a = relu(bn(fc + b)) -> bf16 image (overwrites the staging buffers: every wave passed the barrier)
                bf16_t* dst = sA + ml * LDA;
#pragma unroll
                for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        const int c = hC * 128 + tn * 32 + 8 * gi + 4 * (lane >> 5);
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(s_sc + c);
                        const f32x4 sh = *reinterpret_cast<const f32x4*>(s_sh + c);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(acc[tn][4 * gi + e], sc[e], sh[e]), 0.f);
                        *reinterpret_cast<uint2*>(dst + c) = make_uint2(f2bf_pk(v[0], v[1]), f2bf_pk(v[2], v[3]));
                    }
                // score weights [J][C] -> rows 0..15 of a [32][LDA] tile in the free weight buffer, rows 16..31 zero
                bf16_t* wS = sW + ((NCH - 1) & 1) * WTILE;
                for (int v = tid; v < 32 * (C / 8); v += 512) {
                    const int row = v / (C / 8), col = (v - row * (C / 8)) * 8;
                    u32x4 w = {0u, 0u, 0u, 0u};
                    if (row < J) w = *reinterpret_cast<const u32x4*>(wsc + ((size_t)row * C + col));
                    *reinterpret_cast<u32x4*>(wS + row * LDA + col) = w;
                }
                __syncthreads();
                // ---- score = W_score . a + b (waves of channel half 0; J <= 16 lives in register quads 0,1) ----
                if (hC == 0) {
                    f32x16 acs;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acs[i] = 0.f;
                    const bf16_t* arow = sA + ml * LDA + koff;
                    const bf16_t* wrow = wS + (lane & 31) * LDA + koff;
#pragma unroll
                    for (int kk = 0; kk < C / 16; ++kk)
                        acs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(wrow + kk * 16),
                                                                      *reinterpret_cast<const bf16x8*>(arow + kk * 16), acs, 0, 0, 0);
                    const int m = m0 + ml;
                    bf16_t* sc_out = reinterpret_cast<bf16_t*>(a.score);
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int c = 8 * gi + 4 * (lane >> 5);
                        const f32x4 b = *reinterpret_cast<const f32x4*>(s_bs + c);
                        const uint2 pk = make_uint2(f2bf_pk(acs[4 * gi] + b[0], acs[4 * gi + 1] + b[1]),
                                                    f2bf_pk(acs[4 * gi + 2] + b[2], acs[4 * gi + 3] + b[3]));
                        *reinterpret_cast<uint2*>(sS + ml * LDS_ + c) = pk;
                        if (m < M) *reinterpret_cast<uint2*>(sc_out + (size_t)m * J + c) = pk;
                    }
                }
                __syncthreads();                 // score image visible; the score-weight tile may be overwritten
                HSTAMP();
            }
        });
        HSTAMP();

        if (has_next) {
            // ---- next = fc_(a) + score_(score) + biases + x ----
            constexpr int LST = CW + 4;
            float* stage = reinterpret_cast<float*>(sW) + wave * 32 * LST;     // wave-private [32 px][CW + 4] fp32
            bf16_t* __restrict__ nxt = reinterpret_cast<bf16_t*>(a.next);
            static_for<2>([&](auto stc) {
                constexpr int st = decltype(stc)::value;
                __syncthreads();
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int gi = 0; gi < 4; ++gi) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[2 * st + tn][4 * gi + e];
                        *reinterpret_cast<f32x4*>(stage + (lane & 31) * LST + tn * 32 + 8 * gi + 4 * (lane >> 5)) = v;
                    }
                __syncthreads();
                static_for<NIT>([&](auto itc) {
                    constexpr int it = decltype(itc)::value;
                    const int idx = lane + 64 * it;
                    const int px = idx / VW, cv = (idx - px * VW) * 8;
                    const int m = m0 + q * 32 + px;
                    if (m < M) {
                        const int c = hC * 128 + st * CW + cv;
                        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + px * LST + cv);
                        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + px * LST + cv + 4);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_bo + c);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_bo + c + 4);
                        float res[8], o[8];
                        const u32x4 r = rres[st][it];
                        DT<bf16_t>::unpack(make_uint4(r[0], r[1], r[2], r[3]), res);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            o[e] = v0[e] + b0[e] + res[e];
                            o[4 + e] = v1[e] + b1[e] + res[4 + e];
                        }
                        *reinterpret_cast<uint4*>(nxt + ((size_t)m * C + c)) = DT<bf16_t>::pack(o);
                    }
                });
            });
        }
        HSTAMP();
    }
#ifdef FPD_HEAD_TIMING
    if (tid == 0 && blockIdx.x == 0) {
        // top | first chunk staged | steps 0..3 (fc) | a image + score | steps 4..7 (fc_) | score_ MFMAs | epilogue
        printf("head tiles %d stamps %d:", ntl, hn);
        for (int i = 1; i < hn; ++i) printf(" %lld", hst[i] - hst[i - 1]);
        printf(" | total %lld\n", hst[hn - 1] - hst[0]);
    }
#endif
}

template <int C>
__global__ void head_fold_kernel(const fpd_head_t a, float* out) { head_tables<C>(a, out, threadIdx.x, blockDim.x); }

static int head_block_cap() {
    static int cap = 0;
    if (!cap) {
        const char* e = getenv("FPD_HEAD_BLOCKS");
        cap = e ? atoi(e) : 160;                 // same argument as the fused Bottleneck: leave CUs to the student chain
        if (cap < 8) cap = 8;
    }
    return cap;
}

}  // namespace

static bool head_in_domain(const fpd_head_t& a) {
    return a.dtype == FPD_BF16 && a.C == 256 && a.J == 16;
}

// 0 = launched, 1 = outside the domain
int fpd_head_fused_launch(const fpd_head_t& a, hipStream_t st) {
    if (!head_in_domain(a)) return 1;
    constexpr int C = 256, LDX = 72, LDA = C + 8;
    const size_t lds = (size_t)(3 * C + 32) * sizeof(float) + (size_t)128 * LDA * 2 + (size_t)128 * 24 * 2 + (size_t)2 * C * LDX * 2;
    static LdsAttr configured;        // per device, set once (thread-safe: common.h)
    if (int rc_ = configured.ensure(reinterpret_cast<const void*>(&head_eval_kernel<C>), lds)) return rc_;
    const int ntiles = cdiv(a.N * a.H * a.W, 128), cap = head_block_cap();
    const int nblk = ntiles <= cap ? ntiles : cdiv(ntiles, cdiv(ntiles, cap));
    FPD_LAUNCH((head_eval_kernel<C>), dim3(nblk), dim3(512), lds, st, a, ntiles);
    return 0;
}

int fpd_head_fold_launch(const fpd_head_t& a, float* out, hipStream_t st) {
    if (a.C != 256) return 1;
    FPD_LAUNCH((head_fold_kernel<256>), dim3(1), dim3(256), 0, st, a, out);
    return 0;
}
