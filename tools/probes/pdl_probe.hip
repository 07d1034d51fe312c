// Dev probe (round 4): can two DEPENDENT kernels of the student chain overlap on gfx950, i.e. can kernel k+1 run its
// prologue (kernarg, weights -> LDS) while kernel k is still working, with the data dependency carried by a device-side
// completion counter instead of the queue's barrier bit ("poor man's programmatic dependent launch")?
//   M0  same stream, plain launches, no counters                      (what the step does today)
//   M3  same stream, plain launches, counters signalled + waited      (cost of the counter protocol alone)
//   M1  same stream, hipExtLaunchKernel(hipExtAnyOrderLaunch), counters
//   M2  two streams ping-pong, plain launches, counters only (no events)
// Every kernel computes y = x + 1 over a buffer (ping-pong), so after N launches every element must equal N: stale reads
// through a non-coherent L2 / scalar cache or a broken ordering show up as wrong values.  Spins are bounded (50 ms) and
// raise an error flag instead of hanging.  Second part: cost of the statistics flush with fp64 atomics vs two 64-bit
// integer limbs per value (order-independent exact accumulation).
// hipcc --offload-arch=gfx950 -O2 pdl_probe.hip -o pdl_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Link { unsigned* wait; unsigned target; unsigned* done; };

__global__ __launch_bounds__(256) void chain_k(const float4* __restrict__ x, float4* __restrict__ y, const long long n4,
                                               const float4* __restrict__ w, const int wn4, const Link lk,
                                               long long* stamps, unsigned* err) {
    extern __shared__ float4 s_w[];
    const int tid = threadIdx.x;
    long long t_entry = 0, t_flag = 0;
    if (tid == 0 && blockIdx.x == 0) t_entry = wall_clock64();
    // prologue that does not depend on the predecessor: "weights" into LDS
    float acc = 0.f;
    for (int i = tid; i < wn4; i += 256) { const float4 v = w[i]; s_w[i] = v; acc += v.x; }
    __syncthreads();
    if (lk.wait != nullptr) {
        if (tid == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(lk.wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < lk.target) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 5000000ll) { atomicOr(err, 1u); break; }      // 50 ms at 100 MHz
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_dcache_inv\n s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (tid == 0 && blockIdx.x == 0) t_flag = wall_clock64();
    const float one = 1.f + 0.f * s_w[(tid * 7) % (wn4 > 0 ? wn4 : 1)].y * (acc == 12345.f ? 1.f : 0.f);
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long b = (long long)blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    for (long long i = b + tid; i < e; i += 256) {
        float4 v = x[i];
        v.x += one; v.y += one; v.z += one; v.w += one;
        y[i] = v;
    }
    if (lk.done != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(lk.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && blockIdx.x == 0 && stamps != nullptr) { stamps[0] = t_entry; stamps[1] = t_flag; stamps[2] = wall_clock64(); }
}

// ---- statistics flush variants ----
// mode 0: fp64 atomicAdd pair per channel; 1: two int64 limbs per value (4 atomics per channel)
__global__ __launch_bounds__(256) void flush_k(double* st, long long* sti, const int C, const int R, const int mode) {
    const int tid = threadIdx.x;
    const int rep = blockIdx.x % R;
    // pretend work: a little ALU so the kernel is not empty
    double v1 = 1.0 + tid * 1e-3 + blockIdx.x * 1e-6, v2 = v1 * v1;
    for (int c = tid; c < C; c += 256) {
        if (mode == 0) {
            atomicAdd(st + (size_t)rep * 2 * C + c, v1);
            atomicAdd(st + (size_t)rep * 2 * C + C + c, v2);
        } else {
            const double vs[2] = {v1, v2};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const double v = vs[s];
                const long long hi = __double2ll_rn(v * 1048576.0);                        // 2^20 (include/fpd_amd.h, ABI version 2)
                const double r = v - (double)hi * (1.0 / 1048576.0);
                const long long lo = __double2ll_rn(r * 1152921504606846976.0);     // 2^60
                unsigned long long* p = reinterpret_cast<unsigned long long*>(sti) + ((size_t)(rep * 2 + s) * 2) * C;
                atomicAdd(p + c, (unsigned long long)hi);
                atomicAdd(p + C + c, (unsigned long long)lo);
            }
        }
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int N = 200;
    const int GRID = argc > 1 ? atoi(argv[1]) : 512;
    const int LDS = 64 * 1024;
    hipStream_t s[2];
    for (auto& q : s) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    hipEvent_t t0, t1, ej;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    unsigned *cnt = nullptr, *err = nullptr;
    long long* stamps = nullptr;
    float4* w = nullptr;
    const int wn4 = 16 * 1024 / 16;            // 16 KB of "weights"
    CK(hipMalloc(&cnt, N * sizeof(unsigned))); CK(hipMalloc(&err, 4)); CK(hipMalloc(&stamps, N * 3 * sizeof(long long)));
    CK(hipMalloc(&w, wn4 * 16)); CK(hipMemset(w, 0, wn4 * 16)); CK(hipMemset(err, 0, 4));
    for (long long bytes : {32ll << 20, 2ll << 20, 256ll << 10}) {
        const long long n4 = bytes / 16;
        float4* buf[2];
        CK(hipMalloc(&buf[0], bytes)); CK(hipMalloc(&buf[1], bytes));
        std::vector<float> host(bytes / 4);
        for (int mode : {0, 3, 1, 2, 0}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(buf[0], 0, bytes)); CK(hipMemset(buf[1], 0, bytes));
                CK(hipMemset(cnt, 0, N * sizeof(unsigned))); CK(hipMemset(stamps, 0, N * 3 * sizeof(long long)));
                CK(hipDeviceSynchronize());
                const double h0 = now();
                CK(hipEventRecord(t0, s[0]));
                if (mode == 2) CK(hipStreamWaitEvent(s[1], t0, 0));
                for (int k = 0; k < N; ++k) {
                    Link lk;
                    lk.wait = (mode != 0 && k > 0) ? cnt + (k - 1) : nullptr;
                    lk.target = (unsigned)GRID;
                    lk.done = mode != 0 ? cnt + k : nullptr;
                    const float4* x = buf[k & 1];
                    float4* y = buf[(k + 1) & 1];
                    long long* stp = stamps + 3 * k;
                    hipStream_t st = mode == 2 ? s[k & 1] : s[0];
                    if (mode == 1) {
                        void* args[] = {(void*)&x, (void*)&y, (void*)&n4, (void*)&w, (void*)&wn4, (void*)&lk, (void*)&stp, (void*)&err};
                        CK(hipExtLaunchKernel(reinterpret_cast<const void*>(&chain_k), dim3(GRID), dim3(256), args, LDS, st, nullptr, nullptr,
                                              hipExtAnyOrderLaunch));
                    } else {
                        hipLaunchKernelGGL(chain_k, dim3(GRID), dim3(256), LDS, st, x, y, n4, (const float4*)w, wn4, lk, stp, err);
                    }
                }
                if (mode == 2) { CK(hipEventRecord(ej, s[1])); CK(hipStreamWaitEvent(s[0], ej, 0)); }
                CK(hipEventRecord(t1, s[0]));
                const double h1 = now();
                CK(hipEventSynchronize(t1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, t0, t1));
                if (rep == 0) continue;
                CK(hipMemcpy(host.data(), buf[N & 1], bytes, hipMemcpyDeviceToHost));
                float mn = 1e30f, mx = -1e30f;
                for (float v : host) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
                unsigned herr = 0;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                std::vector<long long> hs(N * 3);
                CK(hipMemcpy(hs.data(), stamps, N * 3 * sizeof(long long), hipMemcpyDeviceToHost));
                double gap_entry = 0, gap_flag = 0, dur = 0;
                int early = 0;
                for (int k = 1; k < N; ++k) {
                    gap_entry += (double)(hs[3 * k] - hs[3 * (k - 1) + 2]);         // entry of k - end of k-1 (block 0 of each), 10 ns ticks
                    gap_flag += (double)(hs[3 * k + 1] - hs[3 * (k - 1) + 2]);
                    dur += (double)(hs[3 * k + 2] - hs[3 * k]);
                    if (hs[3 * k] < hs[3 * (k - 1) + 2]) ++early;
                }
                printf("bytes %8lld grid %d mode %d: %7.2f us/kernel  host %5.2f us/launch  result [%g, %g] (want %d) err %u | "
                       "entry-prev_end %.2f us, flag-prev_end %.2f us, block0 life %.2f us, entered-before-prev-ended %d/%d\n",
                       bytes, GRID, mode, ms * 1e3 / N, (h1 - h0) * 1e6 / N, mn, mx, N, herr, gap_entry / (N - 1) / 100.0,
                       gap_flag / (N - 1) / 100.0, dur / (N - 1) / 100.0, early, N - 1);
                fflush(stdout);
            }
        }
        CK(hipFree(buf[0])); CK(hipFree(buf[1]));
    }
    // ---- statistics flush: fp64 pair vs two int64 limbs ----
    {
        const int C = 64, R = 4;
        double* st; long long* sti;
        CK(hipMalloc(&st, R * 2 * C * 8)); CK(hipMalloc(&sti, R * 4 * C * 8));
        for (int grid : {512, 1024, 4096})
            for (int mode : {0, 1, 0, 1}) {
                CK(hipMemset(st, 0, R * 2 * C * 8)); CK(hipMemset(sti, 0, R * 4 * C * 8));
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(t0, s[0]));
                for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(flush_k, dim3(grid), dim3(256), 0, s[0], st, sti, C, R, mode);
                CK(hipEventRecord(t1, s[0]));
                CK(hipEventSynchronize(t1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, t0, t1));
                printf("flush grid %4d C %d R %d mode %d (%s): %.2f us/kernel\n", grid, C, R, mode, mode ? "2 x int64 limbs" : "fp64 pair", ms * 1e3 / 200);
            }
    }
    return 0;
}
