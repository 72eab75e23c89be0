// microbench_issue.hip -- what a dependent chain of small launches pays per vector instruction (development tool).
// Kernel<N>: every lane runs N vector instructions (two independent xor/add chains, so issue is not latency-bound), straight-line
// code (cold instruction cache at every launch) or a loop over a 64-instruction body (warm after the first trip).
// Grid sweep: 1024 waves (one per SIMD on 256 CUs), 2048 (two per SIMD), 4096.  100 launches per hipGraph, dependent (one stream).
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_issue.hip -o gpurun_out/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int N>
__global__ __launch_bounds__(256) void straight(uint32_t *out, uint32_t seed) {
    uint32_t a = threadIdx.x ^ seed, b = blockIdx.x + seed;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(a));
    }
    if ((a ^ b) == 0x12345678u) out[0] = a;  // never true in practice: keeps the chains alive
}

template <int BODY>
__global__ __launch_bounds__(256) void looped(uint32_t *out, uint32_t seed, int trips) {
    uint32_t a = threadIdx.x ^ seed, b = blockIdx.x + seed;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < BODY / 2; ++i) {
            asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(b) : "v"(a));
        }
    }
    if ((a ^ b) == 0x12345678u) out[0] = a;
}

template <typename F>
static float graph_us(F launch, hipStream_t s, int per_graph = 100, int replays = 20) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < per_graph; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return ms * 1e3f / (per_graph * replays);
}

int main() {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    uint32_t *out;
    CK(hipMalloc(&out, 4096));
    printf("waves  kind      N      us/launch\n");
    for (int blocks : {256, 512, 1024, 2048}) {  // x 4 waves per block
        const int waves = blocks * 4;
#define RUN_S(N) printf("%5d  straight %5d  %8.3f\n", waves, N, graph_us([&] { hipLaunchKernelGGL(straight<N>, dim3(blocks), dim3(256), 0, s, out, 1u); }, s));
        RUN_S(2) RUN_S(128) RUN_S(256) RUN_S(512) RUN_S(1024) RUN_S(2048) RUN_S(4096)
#define RUN_L(T) printf("%5d  looped64 %5d  %8.3f\n", waves, 64 * T, graph_us([&] { hipLaunchKernelGGL(looped<64>, dim3(blocks), dim3(256), 0, s, out, 1u, T); }, s));
        RUN_L(2) RUN_L(8) RUN_L(16) RUN_L(32) RUN_L(64)
    }
    return 0;
}
