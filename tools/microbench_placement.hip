// microbench_placement.hip -- where does the hardware put the workgroups of a launch?  256 threads and 40 KiB of LDS per workgroup (the reset
// kernels' footprint: three fit a CU), 1 280 workgroups; every workgroup records its XCC / SE / CU and start time.  Development tool.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_placement.hip -o tools/bin/mb_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *t, int busy_blocks, int spin) {
    __shared__ uint32_t lds[10240];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.x * 2] = hw;
        out[blockIdx.x * 2 + 1] = xcc;
        t[blockIdx.x] = wall_clock64();
    }
    if ((int)blockIdx.x < busy_blocks) {  // "working" workgroups stay for a while
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    }
    if (lds[(threadIdx.x * 7) & 255] == 9999u) out[0] = 1;
}

int main(int argc, char **argv) {
    const int grid = 1280, busy = argc > 1 ? atoi(argv[1]) : 512;
    uint32_t *out; unsigned long long *t;
    CK(hipMalloc(&out, grid * 8)); CK(hipMalloc(&t, grid * 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, t, busy, 800);  // 8 us
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> h(grid * 2); std::vector<unsigned long long> ht(grid);
    CK(hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), t, grid * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> per_cu;
    for (int b = 0; b < busy; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu]++;
    }
    std::map<int, int> hist;
    for (auto &kv : per_cu) hist[kv.second]++;
    printf("%d busy workgroups on %zu distinct CUs; CUs by number of busy workgroups:", busy, per_cu.size());
    for (auto &kv : hist) printf("  %d x %d", kv.second, kv.first);
    unsigned long long t0 = ~0ull; for (auto v : ht) t0 = v < t0 ? v : t0;
    printf("\nfirst block xcc of blocks 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b + 1] & 0xF);
    printf("\nstart time (us) of block 0, 511, 512, 767, 768, 1023, 1024, 1279: ");
    for (int b : {0, 511, 512, 767, 768, 1023, 1024, 1279}) printf(" %.2f", (double)(ht[b] - t0) / 100.0);
    printf("\n");
    return 0;
}
