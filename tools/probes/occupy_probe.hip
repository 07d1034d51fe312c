// Dev probe: WHAT about a concurrent kernel slows the student chain down?  A shared library with background kernels that take
// compute units the way the fused teacher Bottleneck does (persistent 512-thread blocks that own their CU's LDS) but differ
// in what they do while they hold them:
//   mode 0  spin on the wall clock (no memory traffic, no MFMA)
//   mode 1  stream a small L2-resident buffer (weight-tile traffic of the Bottleneck: L2 -> CU, no HBM)
//   mode 2  stream a large buffer (HBM reads)
//   mode 3  MFMA loop on registers (matrix-pipe / power load, no memory traffic)
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC occupy_probe.hip -o occupy_probe.so     (tools/probes/occupy_probe.py drives it)
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void occupy_kernel(const int mode, const long long ticks, const uint4* __restrict__ buf,
                                                     const long long nvec, float* sink) {
    extern __shared__ unsigned char smem[];
    const long long t0 = wall_clock64();            // 100 MHz
    uint4 acc = make_uint4(0, 0, 0, 0);
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    bf16x8 a;
    for (int i = 0; i < 8; ++i) a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f);
    long long off = (long long)blockIdx.x * 512 + threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        if (mode == 1 || mode == 2) {
#pragma unroll 8
            for (int i = 0; i < 8; ++i) {
                const uint4 v = buf[off % nvec];
                acc.x ^= v.x; acc.y += v.y; acc.z ^= v.z; acc.w += v.w;
                off += (long long)gridDim.x * 512;
            }
        } else if (mode == 3) {
#pragma unroll 16
            for (int i = 0; i < 16; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c, 0, 0, 0);
        }
    }
    if (threadIdx.x == 0) smem[0] = 1;
    if (sink != nullptr && (acc.x + acc.y + acc.z + acc.w == 0x12345u || c[0] == -1.f)) *sink = c[3] + smem[0];
}

extern "C" int occupy_launch(int mode, int blocks, int lds_bytes, long long ticks, const void* buf, long long nvec, void* sink,
                             void* stream) {
    static int configured = 0;
    if (lds_bytes > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
        configured = lds_bytes;
    }
    hipLaunchKernelGGL(occupy_kernel, dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream, mode, ticks,
                       reinterpret_cast<const uint4*>(buf), nvec, reinterpret_cast<float*>(sink));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
