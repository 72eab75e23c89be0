// microbench_chain.hip -- where a one-launch-per-step kernel's time goes at 65 536 envs (development tool): in-kernel
// s_memtime stamps around each dependent memory access of the step kernel's chain (action -> gate entry -> row groups -> stores),
// in a hipGraph of dependent launches where every launch reads what the previous one wrote.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_chain.hip -o /tmp/mb_chain
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
    uint4 *state;            // tiles of 64 envs x 8 groups x 16 B
    const int32_t *actions;  // [B]
    const uint2 *gates;      // [170]
    int32_t *depth;
    float *reward;
    uint8_t *done;
    uint64_t *stamps;        // [waves][8] (last launch wins)
    uint32_t B;
    int mode;                // 0: chain as in the step kernel; 1: rows requested up front (all 8 groups), selected after the gate arrives
};

__global__ __launch_bounds__(256) void chain(Args a) {
    const uint32_t env = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
    uint4 all[8];
    if (a.mode == 1) {
#pragma unroll
        for (int g = 0; g < 8; ++g) all[g] = tile[g * 64 + lane];
    }
    const int32_t act = a.actions[env];
    int32_t depth = a.depth[env];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint2 g = a.gates[act];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t2 = __builtin_amdgcn_s_memtime();
    const uint32_t g0 = g.x & 7u, g1 = (g.x >> 3) & 7u;
    uint4 va, vb;
    if (a.mode == 1) {
        va = all[0]; vb = all[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) { if (g0 == (uint32_t)k) va = all[k]; if (g1 == (uint32_t)k) vb = all[k]; }
    } else {
        va = tile[g0 * 64 + lane];
        vb = tile[g1 * 64 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t3 = __builtin_amdgcn_s_memtime();
    va.x ^= vb.y; vb.z ^= va.w; va.y += g.y;
    tile[g0 * 64 + lane] = va;
    if (g1 != g0) tile[g1 * 64 + lane] = vb;
    a.depth[env] = depth - 1;
    a.reward[env] = (float)va.x;
    a.done[env] = (uint8_t)(va.y & 1u);
    const uint64_t t4 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t5 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        uint64_t *s = a.stamps + (uint64_t)(env >> 6) * 8;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5;
    }
}

int main() {
    const uint32_t B = 65536;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    Args a{};
    a.B = B;
    CK(hipMalloc(&a.state, (size_t)B * 128));
    CK(hipMemset(a.state, 1, (size_t)B * 128));
    std::vector<int32_t> acts(B);
    for (auto &x : acts) x = rand() % 170;
    int32_t *dact;
    CK(hipMalloc(&dact, B * 4));
    CK(hipMemcpy(dact, acts.data(), B * 4, hipMemcpyHostToDevice));
    a.actions = dact;
    std::vector<uint2> gates(170);
    for (auto &g : gates) g = make_uint2((uint32_t)rand(), (uint32_t)rand());
    uint2 *dg;
    CK(hipMalloc(&dg, 170 * 8));
    CK(hipMemcpy(dg, gates.data(), 170 * 8, hipMemcpyHostToDevice));
    a.gates = dg;
    CK(hipMalloc(&a.depth, B * 4));
    CK(hipMalloc(&a.reward, B * 4));
    CK(hipMalloc(&a.done, B));
    CK(hipMemset(a.depth, 0, B * 4));
    CK(hipMalloc(&a.stamps, (B / 64) * 8 * 8));
    for (int mode = 0; mode < 2; ++mode) {
        a.mode = mode;
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(chain, dim3(B / 256), dim3(256), 0, s, a);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> st((B / 64) * 8);
        CK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        const char *names[5] = {"action+depth", "gate entry", "row groups", "issue stores", "drain stores"};
        printf("mode %d: %.3f us per launch; per-wave cycles (median / p90 over %u waves), s_memtime ticks:\n", mode, ms * 1e3 / 1000, B / 64);
        for (int k = 0; k < 5; ++k) {
            std::vector<uint64_t> d;
            for (uint32_t w = 0; w < B / 64; ++w) d.push_back(st[w * 8 + k + 1] - st[w * 8 + k]);
            std::sort(d.begin(), d.end());
            printf("   %-14s %6llu / %6llu\n", names[k], (unsigned long long)d[d.size() / 2], (unsigned long long)d[d.size() * 9 / 10]);
        }
        std::vector<uint64_t> t0s, t5s;
        for (uint32_t w = 0; w < B / 64; ++w) { t0s.push_back(st[w * 8]); t5s.push_back(st[w * 8 + 5]); }
        std::sort(t0s.begin(), t0s.end());
        std::sort(t5s.begin(), t5s.end());
        printf("   first wave start -> last wave start %llu ticks; first start -> last end %llu ticks; median lifetime %llu\n",
               (unsigned long long)(t0s.back() - t0s.front()), (unsigned long long)(t5s.back() - t0s.front()), 0ull);
    }
    return 0;
}
