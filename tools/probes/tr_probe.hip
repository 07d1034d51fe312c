// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds ushort value == element index; every lane passes an
// address and we print which 4 elements it receives.   hipcc --offload-arch=gfx950 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0 || mode == 10) addr = l * 8;                                    // lane l -> elements 4l..4l+3 (contiguous 8 B per lane)
    else if (mode == 1) addr = ((l & 15) * 64 + (l >> 4) * 4) * 2;  // row = l&15 (pitch 64 elements), 4-col group = l>>4
    else addr = ((l & 15) * 64 + (l >> 4) * 16) * 2;                // row = l&15, col group of 4 at 16*(l>>4)
    unsigned long long r;
    addr += (unsigned)(size_t)lds;     // LDS base of the array (normally 0)
    if (mode >= 10) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    else asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)((r >> (16 * j)) & 0xffff);
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    const int modes[4] = {10, 0, 1, 2};
    for (int mi = 0; mi < 4; ++mi) {
        const int mode = modes[mi];
        hipMemset(d, 0xff, sizeof(h));
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        hipError_t e3 = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("launch %s sync %s copy %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf(" l%02d:[%4d %4d %4d %4d]", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
