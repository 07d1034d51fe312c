// Shared device helpers for the gfx950 FPD kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "fpd_amd.h"

#define FPD_MAXC 512  // largest channel count a fused BN prologue supports (HRNet-W48 peak is 384)

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

int fpd_fail(int code, const char* fmt, ...);  // sets fpd_last_error(), returns code
#define FPD_CHECK_HIP(expr)                                                          \
    do {                                                                             \
        hipError_t e_ = (expr);                                                      \
        if (e_ != hipSuccess) return fpd_fail(-100 - (int)e_, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define FPD_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return fpd_fail(-2, __VA_ARGS__); \
    } while (0)

extern int g_fpd_backend;

// Every kernel launch of the library goes through FPD_LAUNCH.  With FPD_LAUNCH_LOG=<file> in the environment each launch is
// also appended to that file in host order -- "<kernel expression>\t<grid>\t<block>\t<plan op tag: index, type, shape>" -- so
// that the rows of a rocprofv3 kernel trace (same host order: Dispatch_Id) can be keyed on the tensor shape a launch worked
// on, which the trace itself does not carry (tools/profile_summarize.py).  Off (the default): one predictable branch.
extern bool g_fpd_launch_log;
void fpd_log_launch(const char* kernel, const dim3& grid, const dim3& block);
#define FPD_LAUNCH(kernel, grid, block, lds, stream, ...)                     \
    do {                                                                       \
        if (g_fpd_launch_log) fpd_log_launch(#kernel, grid, block);            \
        hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);     \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel: a per-DEVICE attribute, raised at most once per size and
// device; concurrent callers (one host thread per device) are serialised on the slow path only.
constexpr int FPD_MAX_DEVICES = 64;
struct LdsAttr {
    std::atomic<size_t> done[FPD_MAX_DEVICES];      // static storage: zero-initialised
    std::mutex mu;
    int ensure(const void* fn, size_t lds) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return fpd_fail(-100 - (int)e, "hipGetDevice: %s", hipGetErrorString(e));
        if (dev < 0 || dev >= FPD_MAX_DEVICES) return fpd_fail(-2, "device ordinal %d outside [0,%d)", dev, FPD_MAX_DEVICES);
        if (lds <= done[dev].load(std::memory_order_acquire)) return 0;
        std::lock_guard<std::mutex> lock(mu);
        if (lds <= done[dev].load(std::memory_order_relaxed)) return 0;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fpd_fail(-100 - (int)e, "hipFuncSetAttribute(%zu B LDS): %s", lds, hipGetErrorString(e));
        done[dev].store(lds, std::memory_order_release);
        return 0;
    }
};

// floor(i * n / d), 0 <= i <= d: the first unit of block i's contiguous range when n units are cut into d ranges (every persistent
// kernel starts with two of these).  As a 64-bit quotient it is ~150 dependent scalar instructions -- in front of the first load of
// every block; the products of all shapes but absurd ones fit 32 bits (the branch is uniform).
__device__ __forceinline__ int fpd_cut(int i, int n, int d) {
#ifndef FPD_CUT64      // (probe build: the 64-bit quotient everywhere)
    if (n < (1 << 20) && d < (1 << 11)) return (int)(((unsigned)i * (unsigned)n) / (unsigned)d);
#endif
    return (int)((long long)i * n / d);
}

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// two floats -> packed bf16x2 (lo = a), round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t f2bf_pk(float a, float b) {
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&r);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf_pk(f, 0.f) & 0xffffu); }

// ---- storage-type traits: 16-byte vectors of VEC elements --------------------------------
template <typename T>
struct DT;
template <>
struct DT<float> {
    static constexpr int VEC = 4;
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ void unpack(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y);
        f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <>
struct DT<bf16_t> {
    static constexpr int VEC = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
    static __device__ __forceinline__ void unpack(const uint4& r, float* f) {
        f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
        f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
        f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
        f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ uint4 pack(const float* f) {
        return make_uint4(f2bf_pk(f[0], f[1]), f2bf_pk(f[2], f[3]), f2bf_pk(f[4], f[5]), f2bf_pk(f[6], f[7]));
    }
};

// ---- exact statistics (include/fpd_amd.h, fpd_stat_t): a value travels as two 64-bit integer limbs ----------------
// Integer addition is associative, so the sums do not depend on the order in which blocks (device atomics) or threads
// (LDS atomics) contribute: bit-repeatable statistics at the cost of an fp64 atomic pair (tools/probes/pdl_probe.hip).
// Headroom: the residual a contribution leaves in its lo limb is <= 2^-21, i.e. |lo| <= 2^39 in units of
// 2^-60, so 2^24 = 16.7 M contributions of ANY sign pattern fit the 63 bits of a limb sum (the largest producer of the hot
// path, a capped elementwise grid, adds 512 blocks x 256 threads = 2^17 per channel).  The first encoding of round 4
// (hi in units of 2^-8, |lo| <= 2^51) wrapped after 4 096 worst-case addends.
#define FPD_STAT_HI_SCALE 1048576.0                   // hi limb: units of 2^-20
#define FPD_STAT_LO_SCALE 1152921504606846976.0       // lo limb: units of 2^-60
struct StatLimbs { long long hi, lo; };
__device__ __forceinline__ StatLimbs stat_split(double v) {
    const double lim = 4398046511104.0;               // 2^42: beyond it (or NaN) the network has diverged; stay finite and huge
    v = (v == v) ? fmin(fmax(v, -lim), lim) : lim;
    StatLimbs s;
    s.hi = __double2ll_rn(v * FPD_STAT_HI_SCALE);
    s.lo = __double2ll_rn((v - (double)s.hi * (1.0 / FPD_STAT_HI_SCALE)) * FPD_STAT_LO_SCALE);
    return s;
}
__device__ __forceinline__ double stat_join(long long hi, long long lo) {
    return (double)hi * (1.0 / FPD_STAT_HI_SCALE) + (double)lo * (1.0 / FPD_STAT_LO_SCALE);
}
// replica of a statistics buffer this block adds into
__device__ __forceinline__ int stats_replica() { return (int)((blockIdx.x + blockIdx.y * gridDim.x) % FPD_STATS_REPLICAS); }
// sum `s` (0 / 1) of channel c += v, into the calling block's replica of a buffer over C channels
__device__ __forceinline__ void stat_atomic_add(fpd_stat_t* buf, int C, int s, int c, double v) {
    const StatLimbs l = stat_split(v);
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf) + ((size_t)stats_replica() * 4 + 2 * s) * C + c;
    atomicAdd(p, (unsigned long long)l.hi);
    atomicAdd(p + C, (unsigned long long)l.lo);
}
// the same into an LDS accumulator [2 sums][2 limbs][C] (threads of a block in any order)
__device__ __forceinline__ void stat_lds_add(long long* acc, int C, int s, int c, double v) {
    const StatLimbs l = stat_split(v);
    unsigned long long* p = reinterpret_cast<unsigned long long*>(acc) + (size_t)(2 * s) * C + c;
    atomicAdd(p, (unsigned long long)l.hi);
    atomicAdd(p + C, (unsigned long long)l.lo);
}
// LDS accumulator -> the block's replica (limbs are added as they are: still exact)
__device__ __forceinline__ void stat_flush_lds(fpd_stat_t* buf, const long long* acc, int C, int c) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf) + (size_t)stats_replica() * 4 * C + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p + q * C, (unsigned long long)acc[q * C + c]);
}
// raw limbs of sum s of channel c, all replicas: two halves so that the loads can be issued long before they are needed
struct StatRaw { long long hi[FPD_STATS_REPLICAS], lo[FPD_STATS_REPLICAS]; };
__device__ __forceinline__ void stat_request(const fpd_stat_t* buf, int C, int s, int c, StatRaw& r) {
#pragma unroll
    for (int q = 0; q < FPD_STATS_REPLICAS; ++q) {
        r.hi[q] = buf[((size_t)q * 4 + 2 * s) * C + c];
        r.lo[q] = buf[((size_t)q * 4 + 2 * s + 1) * C + c];
    }
}
__device__ __forceinline__ double stat_resolve(const StatRaw& r) {
    long long hi = 0, lo = 0;
#pragma unroll
    for (int q = 0; q < FPD_STATS_REPLICAS; ++q) { hi += r.hi[q]; lo += r.lo[q]; }
    return stat_join(hi, lo);
}
// sum over replicas of sum s of channel c
__device__ __forceinline__ double stats_sum(const fpd_stat_t* buf, int C, int s, int c) {
    StatRaw r;
    stat_request(buf, C, s, c, r);
    return stat_resolve(r);
}

// ---- BatchNorm coefficients from batch statistics (TRAIN) or running estimates (EVAL) --------
// a = x*scale + shift  ==  gamma*(x-mean)*invstd + beta.  Mean/variance are derived in fp64
// from the exact sums (biased variance, like torch's batch_norm); scale/shift are rounded once.
// The same coefficients in two halves, for prologues that want every load in flight before any of the fp64 arithmetic:
// bn_request() only issues the loads of channel c, bn_resolve() does the arithmetic on what arrived.
struct BnRaw { StatRaw s1, s2; float g, b, eps, rm, rv; int mode; };
__device__ __forceinline__ void bn_request(const fpd_bn_t& bn, int c, int C, BnRaw& r) {
    // Three disjoint paths, every field a path needs assigned exactly ONCE in it and nothing else touched (bn_resolve reads
    // s1 / s2 only in TRAIN mode, rm / rv only otherwise).  The earlier form -- defaults first, loads over them -- let hipcc sink
    // the default writes of rm / rv behind the gamma / beta loads of the TRAIN path: a write to a register with a load possibly
    // in flight needs `s_waitcnt vmcnt(0)`, so the sixteen statistics loads were ISSUED only after gamma and beta had arrived
    // -- one more memory round trip in front of every block of every kernel with a BatchNorm prologue or epilogue (r06, found
    // in conv_c1's assembly: 12.7 k cycles to the first barrier of the 64x64 data gradients).
    r.mode = bn.mode; r.eps = bn.eps;
    if (bn.mode == FPD_BN_TRAIN) {
        r.g = bn.gamma[c];
        r.b = bn.beta[c];
        stat_request(bn.stats, C, 0, c, r.s1);
        stat_request(bn.stats, C, 1, c, r.s2);
    } else if (bn.mode == FPD_BN_NONE) {
        r.g = 1.f; r.b = 0.f; r.rm = 0.f; r.rv = 1.f;
    } else {
        r.g = bn.gamma[c];
        r.b = bn.beta[c];
        r.rm = bn.running_mean[c];
        r.rv = bn.running_var[c];
    }
}
__device__ __forceinline__ void bn_resolve(const BnRaw& r, double count, float& scale, float& shift, float& mean, float& invstd) {
    double m, var;
    if (r.mode == FPD_BN_TRAIN) {
        m = stat_resolve(r.s1) / count;
        var = stat_resolve(r.s2) / count - m * m;
        if (var < 0.0) var = 0.0;
    } else {
        m = (double)r.rm;
        var = (double)r.rv;
    }
    const double is = 1.0 / sqrt(var + (double)r.eps);
    const double g = (double)r.g;
    mean = (float)m;
    invstd = (float)is;
    scale = (float)(g * is);
    shift = (float)((double)r.b - m * g * is);
}
__device__ __forceinline__ void bn_coef(const fpd_bn_t& bn, int c, int C, double count, float& scale,
                                        float& shift, float& mean, float& invstd) {
    BnRaw r;
    bn_request(bn, c, C, r);
    bn_resolve(r, count, scale, shift, mean, invstd);
}

// The same for a FROZEN network's BatchNorm (running estimates only; the fused teacher kernels, whose API admits nothing else).
// Kept free of the mode dispatch on purpose: inlined into bneck_eval_kernel the three-way dispatch of bn_request() was
// miscompiled by hipcc 7.2 (the gamma / beta address registers were left undefined on the EVAL path: memory fault).
__device__ __forceinline__ void bn_coef_eval(const fpd_bn_t& bn, int c, float& scale, float& shift) {
    const double m = (double)bn.running_mean[c], var = (double)bn.running_var[c];
    const double is = 1.0 / sqrt(var + (double)bn.eps);
    const double g = (double)bn.gamma[c];
    scale = (float)(g * is);
    shift = (float)((double)bn.beta[c] - m * g * is);
}

// Fill LDS scale/shift tables for all C channels (call from every thread, then __syncthreads()).
__device__ __forceinline__ void bn_fill(const fpd_bn_t& bn, int C, double count, float* s_scale, float* s_shift) {
    if (bn.mode == FPD_BN_NONE) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float sc, sh, mu, is;
        bn_coef(bn, c, C, count, sc, sh, mu, is);
        s_scale[c] = sc;
        s_shift[c] = sh;
    }
}

__device__ __forceinline__ float bn_act(float x, float scale, float shift, int relu) {
    float v = fmaf(x, scale, shift);
    return (relu && v < 0.f) ? 0.f : v;
}

// wave-level sum over 64 lanes
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Probe build only (tools/probes/tile_timing.sh, -DFPD_TILE_TIMING): cycle stamps of block (0, 0) at the phase boundaries of
// conv_tile_body and its epilogue, printed by the kernel.
#ifdef FPD_TILE_TIMING
__shared__ long long fpd_tile_stamp[32];
__shared__ int fpd_tile_ns;
#define TILE_STAMP() do { if (threadIdx.x == 0 && fpd_tile_ns < 32) fpd_tile_stamp[fpd_tile_ns++] = clock64(); } while (0)
#else
#define TILE_STAMP() do { } while (0)
#endif
