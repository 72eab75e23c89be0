// microbench_layout.hip -- the one-step kernel's memory pattern beyond the Infinity Cache (2^22 envs x 128 B = 512 MiB): which state
// layout moves the fewest DRAM bytes per env-step (development tool).  Every mode does the step kernel's chain (action -> gate entry ->
// the rows of <= 2 qubits -> write back) on a different layout of the same 128 B per env:
//   0  TILE: [tile of 64 envs][group 0..7][lane] uint4 -- two 16-byte gathers / scatters per lane (what kernels_qm.hip does)
//   1  env-major half records: [env][half 0..1][4 x uint4] -- a lane reads the 64-byte half(s) its gate touches, writes them back whole
//   2  env-major records: a lane reads its 128 bytes, writes back the touched 64-byte half(s)
//   3  TILE, but a touched group is written back together with its 64-byte sector mates (4 lanes x 16 B): lanes also store the group(s)
//      their three neighbours touched (they read them first) -- full-sector writes without changing the layout
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_layout.hip -o /tmp/mb_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Args {
    uint4 *state;
    const int32_t *actions;
    const uint2 *gates;
    int32_t *depth;
    uint32_t *bad;
    float *reward;
    uint8_t *done, *success;
    uint32_t B;
};

__device__ inline void finish(const Args &a, uint32_t env, int32_t depth, uint32_t bad, uint32_t v) {
    a.depth[env] = depth - 1;
    if (v & 1u) a.bad[env] = bad ^ v;
    a.reward[env] = (float)(v & 3u);
    a.done[env] = (uint8_t)(v & 1u);
    a.success[env] = (uint8_t)((v >> 1) & 1u);
}

template <int MODE>
__global__ __launch_bounds__(256) void step(Args a) {
    const uint32_t env = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    if (env >= a.B) return;
    const int32_t act = a.actions[env];
    const int32_t depth = a.depth[env];
    const uint32_t bad = a.bad[env];
    const uint2 g = a.gates[act];
    const uint32_t g0 = g.x & 7u, g1 = (g.x >> 3) & 7u;  // the two 16-byte groups the gate touches (equal for one-qubit gates)
    if (MODE == 0) {
        uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
        uint4 va = tile[g0 * 64 + lane], vb = tile[g1 * 64 + lane];
        va.x ^= vb.y; vb.z ^= va.w;
        tile[g0 * 64 + lane] = va;
        if (g1 != g0) tile[g1 * 64 + lane] = vb;
        finish(a, env, depth, bad, va.x ^ vb.z);
    } else if (MODE == 1) {
        uint4 *rec = a.state + (uint64_t)env * 8u;
        const uint32_t h0 = g0 >> 2, h1 = g1 >> 2;
        uint4 p[4], q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) p[k] = rec[h0 * 4 + k];
        if (h1 != h0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = rec[h1 * 4 + k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = p[k];
        }
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((g0 & 3u) == (uint32_t)k) { p[k].x ^= q[(k + 1) & 3].y; acc ^= p[k].x; }
            if ((g1 & 3u) == (uint32_t)k) { q[k].z ^= p[(k + 2) & 3].w; acc ^= q[k].z; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[h0 * 4 + k] = p[k];
        if (h1 != h0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) rec[h1 * 4 + k] = q[k];
        }
        finish(a, env, depth, bad, acc);
    } else if (MODE == 2) {
        uint4 *rec = a.state + (uint64_t)env * 8u;
        uint4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = rec[k];
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (g0 == (uint32_t)k) { r[k].x ^= r[(k + 1) & 7].y; acc ^= r[k].x; }
            if (g1 == (uint32_t)k) { r[k].z ^= r[(k + 2) & 7].w; acc ^= r[k].z; }
        }
        const uint32_t h0 = g0 >> 2, h1 = g1 >> 2;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((uint32_t)(k >> 2) == h0 || (uint32_t)(k >> 2) == h1) rec[k] = r[k];
        finish(a, env, depth, bad, acc);
    } else if (MODE == 4 || MODE == 5) {  // read the whole tile coalesced; write back the touched groups (4) or everything (5)
        uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
        uint4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = tile[k * 64 + lane];
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (g0 == (uint32_t)k) { r[k].x ^= r[(k + 1) & 7].y; acc ^= r[k].x; }
            if (g1 == (uint32_t)k) { r[k].z ^= r[(k + 2) & 7].w; acc ^= r[k].z; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (MODE == 5 || g0 == (uint32_t)k || g1 == (uint32_t)k) tile[k * 64 + lane] = r[k];
        finish(a, env, depth, bad, acc);
    } else if (MODE == 8) {  // TILE32: groups of four qubits, 32 B per lane: [tile][group 0..3][lane][2 x uint4] -- whole 32-byte sectors written
        uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
        const uint32_t h0 = g0 >> 1, h1 = g1 >> 1;
        uint4 a0 = tile[(h0 * 64 + lane) * 2], a1 = tile[(h0 * 64 + lane) * 2 + 1];
        uint4 b0 = a0, b1 = a1;
        if (h1 != h0) { b0 = tile[(h1 * 64 + lane) * 2]; b1 = tile[(h1 * 64 + lane) * 2 + 1]; }
        a0.x ^= b1.y; b1.z ^= a0.w; a1.y ^= b0.x;
        tile[(h0 * 64 + lane) * 2] = a0;
        tile[(h0 * 64 + lane) * 2 + 1] = a1;
        if (h1 != h0) { tile[(h1 * 64 + lane) * 2] = b0; tile[(h1 * 64 + lane) * 2 + 1] = b1; }
        finish(a, env, depth, bad, a0.x ^ b1.z);
    } else if (MODE == 6) {  // mode 0 without any row access: the scalar arrays alone
        finish(a, env, depth, bad, g0 ^ g1);
    } else if (MODE == 7) {  // mode 0, reads only
        uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
        uint4 va = tile[g0 * 64 + lane], vb = tile[g1 * 64 + lane];
        finish(a, env, depth, bad, va.x ^ vb.z);
    } else {
        uint4 *tile = a.state + (uint64_t)(env >> 6) * 512u;
        uint4 va = tile[g0 * 64 + lane], vb = tile[g1 * 64 + lane];
        va.x ^= vb.y; vb.z ^= va.w;
        // which groups do the 4 lanes of this 64-byte sector column touch?  every lane then reads and rewrites all of them
        uint32_t m = (1u << g0) | (1u << g1);
        m |= __shfl_xor((int)m, 1);
        m |= __shfl_xor((int)m, 2);
        uint32_t acc = va.x ^ vb.z;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((m >> k) & 1u) {
                uint4 v = (g0 == (uint32_t)k) ? va : (g1 == (uint32_t)k) ? vb : tile[k * 64 + lane];
                tile[k * 64 + lane] = v;
            }
        }
        finish(a, env, depth, bad, acc);
    }
}

template <int MODE>
static void run(Args a, hipStream_t s, const char *name) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(step<MODE>, dim3(a.B / 256), dim3(256), 0, s, a);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 4; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 64;
    printf("B=%8u mode %d %-44s %9.2f us per launch  %6.0f GB/s at 160 B/env  (frac %.3f)\n", a.B, MODE, name, us, 160.0 * a.B / us / 1e3, 160.0 * a.B / us / 8e6);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
}

int main(int argc, char **argv) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (uint32_t B : {65536u, 1u << 20, 1u << 22}) {
        Args a{};
        a.B = B;
        CK(hipMalloc(&a.state, (size_t)B * 128));
        CK(hipMemset(a.state, 1, (size_t)B * 128));
        std::vector<int32_t> acts(B);
        for (auto &x : acts) x = rand() % 170;
        int32_t *dact;
        CK(hipMalloc(&dact, (size_t)B * 4));
        CK(hipMemcpy(dact, acts.data(), (size_t)B * 4, hipMemcpyHostToDevice));
        a.actions = dact;
        // gate -> groups as on a 16-qubit line with 170 actions: 80 one-qubit gates (one group), 90 two-qubit gates on neighbours
        // (q, q+1): same group for even q, adjacent groups for odd q
        std::vector<uint2> gates(170);
        for (int i = 0; i < 170; ++i) {
            uint32_t g0, g1;
            if (i < 80) { g0 = g1 = (i % 16) / 2; }
            else { const uint32_t q = (i - 80) % 15; g0 = q / 2; g1 = (q + 1) / 2; }
            gates[i] = make_uint2(g0 | (g1 << 3), (uint32_t)rand());
        }
        uint2 *dg;
        CK(hipMalloc(&dg, 170 * 8));
        CK(hipMemcpy(dg, gates.data(), 170 * 8, hipMemcpyHostToDevice));
        a.gates = dg;
        CK(hipMalloc(&a.depth, (size_t)B * 4));
        CK(hipMalloc(&a.bad, (size_t)B * 4));
        CK(hipMalloc(&a.reward, (size_t)B * 4));
        CK(hipMalloc(&a.done, B));
        CK(hipMalloc(&a.success, B));
        CK(hipMemset(a.depth, 0, (size_t)B * 4));
        CK(hipMemset(a.bad, 0, (size_t)B * 4));
        run<0>(a, s, "TILE, 16-byte gathers / scatters");
        run<1>(a, s, "env-major 64-byte halves, whole-half writes");
        run<2>(a, s, "env-major 128-byte read, 64-byte writes");
        run<3>(a, s, "TILE, full 64-byte sector write-back");
        run<4>(a, s, "TILE, whole tile read, touched groups written");
        run<5>(a, s, "TILE, whole tile read and written");
        run<6>(a, s, "scalar arrays only");
        run<7>(a, s, "TILE, 16-byte gathers, no row writes");
        run<8>(a, s, "TILE32: 4-qubit groups, 32 B per lane");
        CK(hipFree(a.state)); CK(hipFree(dact)); CK(hipFree(dg)); CK(hipFree(a.depth)); CK(hipFree(a.bad)); CK(hipFree(a.reward)); CK(hipFree(a.done)); CK(hipFree(a.success));
    }
    return 0;
}
