// Block-wide arg-max with numpy.argmax's tie rule (the FIRST maximum in row-major order wins); shared by the per-step
// training metric (pck.hip) and the inference post-processing (infer.hip).
#pragma once
#include "common.h"

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }
__device__ __forceinline__ ArgMax block_argmax(ArgMax m, ArgMax* s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax t;
        t.v = __shfl_xor(m.v, o, 64);
        t.i = __shfl_xor(m.i, o, 64);
        m = better(m, t);
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    ArgMax r = s[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = better(r, s[w]);
    __syncthreads();
    return r;
}
