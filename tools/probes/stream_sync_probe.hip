// Dev probe: cost of kernel->kernel dependencies on gfx950 -- same stream vs cross-stream (event record + wait),
// eager and as a captured hipGraph.  hipcc --offload-arch=gfx950 -O2 stream_sync_probe.hip -o stream_sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void spin(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (sink && cycles < 0) *sink = 1;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int N = 400;
    const long long cyc = 500;          // wall_clock64 ticks at 100 MHz -> 5 us
    hipStream_t s[8];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(N * 2);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    auto report = [&](const char* name, double host_s, float ms) {
        printf("%-44s %.2f us/kernel (GPU)   host %.2f us/kernel\n", name, ms * 1e3 / N, host_s * 1e6 / N);
    };
    for (int rep = 0; rep < 2; ++rep) {
        float ms; double h;
        // A: one stream
        CK(hipDeviceSynchronize());
        h = now(); CK(hipEventRecord(t0, s[0]));
        for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
        CK(hipEventRecord(t1, s[0])); h = now() - h; CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) report("same stream chain (5 us kernels)", h, ms);
        // B: ping-pong across 2 streams, every kernel waits for the previous one on the other stream
        CK(hipDeviceSynchronize());
        h = now(); CK(hipEventRecord(t0, s[0]));
        for (int i = 0; i < N; ++i) {
            hipStream_t st = s[i & 1];
            if (i) CK(hipStreamWaitEvent(st, ev[i - 1], 0));
            spin<<<1, 64, 0, st>>>(cyc, nullptr);
            CK(hipEventRecord(ev[i], st));
        }
        CK(hipStreamWaitEvent(s[0], ev[N - 1], 0));
        CK(hipEventRecord(t1, s[0])); h = now() - h; CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) report("cross-stream ping-pong (2 streams)", h, ms);
        // C: two independent chains on 2 streams (concurrency check): N/2 kernels each
        CK(hipDeviceSynchronize());
        h = now(); CK(hipEventRecord(t0, s[0]));
        CK(hipStreamWaitEvent(s[1], t0, 0));
        for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s[i & 1]>>>(cyc, nullptr);
        CK(hipEventRecord(ev[0], s[1])); CK(hipStreamWaitEvent(s[0], ev[0], 0));
        CK(hipEventRecord(t1, s[0])); h = now() - h; CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) report("2 independent chains, per kernel of total", h, ms);
        // D: 8 independent chains on 8 streams
        CK(hipDeviceSynchronize());
        h = now(); CK(hipEventRecord(t0, s[0]));
        for (int k = 1; k < 8; ++k) CK(hipStreamWaitEvent(s[k], t0, 0));
        for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s[i & 7]>>>(cyc, nullptr);
        for (int k = 1; k < 8; ++k) { CK(hipEventRecord(ev[k], s[k])); CK(hipStreamWaitEvent(s[0], ev[k], 0)); }
        CK(hipEventRecord(t1, s[0])); h = now() - h; CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) report("8 independent chains, per kernel of total", h, ms);
        // E: fork/join per pair: main K, fork side K || main K, join  (N/2 rounds of: 1 + 2 parallel)
        CK(hipDeviceSynchronize());
        h = now(); CK(hipEventRecord(t0, s[0]));
        for (int i = 0; i < N / 3; ++i) {
            spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
            CK(hipEventRecord(ev[2 * i], s[0])); CK(hipStreamWaitEvent(s[1], ev[2 * i], 0));
            spin<<<1, 64, 0, s[1]>>>(cyc, nullptr);
            spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
            CK(hipEventRecord(ev[2 * i + 1], s[1])); CK(hipStreamWaitEvent(s[0], ev[2 * i + 1], 0));
        }
        CK(hipEventRecord(t1, s[0])); h = now() - h; CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        if (rep) printf("%-44s %.2f us per round (ideal 10 + gaps; serial 15 + gaps)   host %.2f us/round\n", "fork/join rounds (1 + 2 parallel kernels)", ms * 1e3 / (N / 3), h * 1e6 / (N / 3));
    }
    // graphs: capture B (ping-pong) and A (chain) and replay
    for (int variant = 0; variant < 3; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        if (variant == 0) for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
        if (variant == 1) {
            for (int i = 0; i < N; ++i) {
                hipStream_t st = s[i & 1];
                if (i) CK(hipStreamWaitEvent(st, ev[i - 1], 0));
                spin<<<1, 64, 0, st>>>(cyc, nullptr);
                CK(hipEventRecord(ev[i], st));
            }
            CK(hipStreamWaitEvent(s[0], ev[N - 1], 0));
        }
        if (variant == 2) {
            for (int i = 0; i < N / 3; ++i) {
                spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
                CK(hipEventRecord(ev[2 * i], s[0])); CK(hipStreamWaitEvent(s[1], ev[2 * i], 0));
                spin<<<1, 64, 0, s[1]>>>(cyc, nullptr);
                spin<<<1, 64, 0, s[0]>>>(cyc, nullptr);
                CK(hipEventRecord(ev[2 * i + 1], s[1])); CK(hipStreamWaitEvent(s[0], ev[2 * i + 1], 0));
            }
        }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, s[0])); CK(hipGraphLaunch(ge, s[0])); CK(hipEventRecord(t1, s[0]));
            CK(hipEventSynchronize(t1)); CK(hipEventElapsedTime(&ms, t0, t1));
        }
        const char* nm[3] = {"graph: same-stream chain", "graph: ping-pong capture", "graph: fork/join rounds"};
        printf("%-44s %.2f us per %s\n", nm[variant], ms * 1e3 / (variant == 2 ? N / 3 : N), variant == 2 ? "round" : "kernel");
    }
    return 0;
}
