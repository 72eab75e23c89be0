// microbench_icache.hip -- what does a launch pay for instructions it executes for the first time?  One wave per workgroup runs a straight-line
// body of N dependent vector instructions (v_xor / v_add / v_mul_u24 on one register: 4-8 bytes each) `reps` times and stamps the wall clock
// (s_memrealtime, 10 ns) around every pass: pass 0 fetches its instructions from L2 (the instruction cache is invalidated at every
// dispatch), later passes hit the cache.  Development tool.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_icache.hip -o tools/bin/mb_icache
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define I3 asm volatile("v_xor_b32 %0, %1, %0\n\tv_add_u32 %0, %1, %0\n\tv_lshl_add_u32 %0, %0, 1, %0" : "+v"(v) : "s"(k));
#define R4(x) x x x x
template <int N>
__device__ __forceinline__ uint32_t body(uint32_t v, uint32_t k) {  // N = 192, 768 or 3072 instructions, straight line
    if constexpr (N == 192) { R4(R4(R4(I3))) }
    else if constexpr (N == 768) { R4(R4(R4(R4(I3)))) }
    else { R4(R4(R4(R4(R4(I3))))) }
    return v;
}

template <int N>
__global__ __launch_bounds__(64) void k(uint32_t *out, unsigned long long *t, int reps, uint32_t key) {
    uint32_t v = threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        const unsigned long long t0 = wall_clock64();
        v = body<N>(v, key);
        asm volatile("s_nop 0" ::"v"(v));
        const unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) t[(size_t)blockIdx.x * reps + r] = t1 - t0;
    }
    if (v == 0x12345u) out[0] = v;
}

template <int N>
static void run(int grid, int reps) {
    uint32_t *out; unsigned long long *t;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&t, (size_t)grid * reps * 8));
    for (int launch = 0; launch < 3; ++launch) hipLaunchKernelGGL(k<N>, dim3(grid), dim3(64), 0, 0, out, t, reps, 77u);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)grid * reps);
    CK(hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost));
    printf("N = %5d instructions, %4d workgroups of one wave (third launch of the same kernel):", N, grid);
    for (int r = 0; r < reps; ++r) {
        std::vector<unsigned long long> v;
        for (int b = 0; b < grid; ++b) v.push_back(h[(size_t)b * reps + r]);
        std::sort(v.begin(), v.end());
        printf("  pass %d: min %.2f med %.2f max %.2f us", r, v.front() / 100.0, v[v.size() / 2] / 100.0, v.back() / 100.0);
    }
    printf("\n");
    CK(hipFree(out)); CK(hipFree(t));
}

int main(int argc, char **argv) {
    const int reps = 3;
    for (int grid : {256, 1024, 2048}) {
        run<192>(grid, reps);
        run<768>(grid, reps);
        run<3072>(grid, reps);
    }
    return 0;
}
