// Shared device helpers for the gfx950 FPD kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ---- BatchNorm coefficients from batch statistics (TRAIN) or running estimates (EVAL) --------
// a = x*scale + shift  ==  gamma*(x-mean)*invstd + beta.  Mean/variance are derived in fp64
// from fp64 sums (biased variance, like torch's batch_norm); scale/shift are rounded once.
__device__ __forceinline__ void bn_coef(const fpd_bn_t& bn, int c, int C, double count, float& scale,
                                        float& shift, float& mean, float& invstd) {
    double m, var;
    if (bn.mode == FPD_BN_TRAIN) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll 4
        for (int r = 0; r < FPD_STATS_REPLICAS; ++r) { s1 += bn.stats[r * 2 * C + c]; s2 += bn.stats[r * 2 * C + C + c]; }
        m = s1 / count;
        var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
    } else {
        m = (double)bn.running_mean[c];
        var = (double)bn.running_var[c];
    }
    const double is = 1.0 / sqrt(var + (double)bn.eps);
    const double g = (double)bn.gamma[c];
    mean = (float)m;
    invstd = (float)is;
    scale = (float)(g * is);
    shift = (float)((double)bn.beta[c] - m * g * is);
}

// The same coefficients in two halves, for prologues that want every load in flight before any of the fp64 arithmetic:
// bn_request() only issues the loads of channel c, bn_resolve() does the arithmetic of bn_coef() on what arrived.
struct BnRaw { double s[2 * FPD_STATS_REPLICAS]; float g, b, eps; int mode; };
__device__ __forceinline__ void bn_request(const fpd_bn_t& bn, int c, int C, BnRaw& r) {
    r.mode = bn.mode; r.eps = bn.eps;
#pragma unroll
    for (int q = 0; q < 2 * FPD_STATS_REPLICAS; ++q) r.s[q] = 0.0;
    r.g = 1.f; r.b = 0.f;
    if (bn.mode == FPD_BN_NONE) return;
    if (bn.mode == FPD_BN_TRAIN) {
#pragma unroll
        for (int q = 0; q < FPD_STATS_REPLICAS; ++q) { r.s[2 * q] = bn.stats[q * 2 * C + c]; r.s[2 * q + 1] = bn.stats[q * 2 * C + C + c]; }
    } else {
        r.s[0] = (double)bn.running_mean[c];
        r.s[1] = (double)bn.running_var[c];
    }
    r.g = bn.gamma[c];
    r.b = bn.beta[c];
}
__device__ __forceinline__ void bn_resolve(const BnRaw& r, double count, float& scale, float& shift, float& mean, float& invstd) {
    double m, var;
    if (r.mode == FPD_BN_TRAIN) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int q = 0; q < FPD_STATS_REPLICAS; ++q) { s1 += r.s[2 * q]; s2 += r.s[2 * q + 1]; }   // same order as bn_coef()
        m = s1 / count;
        var = s2 / count - m * m;
        if (var < 0.0) var = 0.0;
    } else {
        m = r.s[0];
        var = r.s[1];
    }
    const double is = 1.0 / sqrt(var + (double)r.eps);
    const double g = (double)r.g;
    mean = (float)m;
    invstd = (float)is;
    scale = (float)(g * is);
    shift = (float)((double)r.b - m * g * is);
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

// replica of a [R][2][C] statistics buffer this block adds into
__device__ __forceinline__ int stats_replica() { return (int)((blockIdx.x + blockIdx.y * gridDim.x) % FPD_STATS_REPLICAS); }
// sum over replicas of element i of a [R][2][C] buffer
__device__ __forceinline__ double stats_sum(const double* st, int C, int i) {
    double s = 0.0;
#pragma unroll 4
    for (int r = 0; r < FPD_STATS_REPLICAS; ++r) s += st[r * 2 * C + i];
    return s;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
