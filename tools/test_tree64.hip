// development check of scramble_tree64 (device_common.hpp) against a host loop over the same draws
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../qiskit_gym_amd/csrc/device_common.hpp"
using namespace qg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int R = 40;
__global__ __launch_bounds__(256) void k(InitArgs a, uint64_t *out) {
    __shared__ uint64_t prod[4][64];
    uint64_t env = 0, col = 0;
    if (!scramble_tree64<R>(a, 1u, env, col, prod, [](uint32_t s) -> uint64_t { return 1ull << s; })) return;
    for (int i = 0; i < R; ++i) { const uint64_t r = __ballot((col >> i) & 1ull); if (threadIdx.x == 0) out[i] = r; }
}
__global__ __launch_bounds__(256) void kt(InitArgs a, uint32_t count, uint64_t *out) {
    __shared__ uint64_t prod[4][64];
    uint64_t env = 0, col = 0;
    if (!scramble_tree64<R>(a, count, env, col, prod, [](uint32_t s) -> uint64_t { return 1ull << s; })) return;
    for (int i = 0; i < R; ++i) { const uint64_t r = __ballot((col >> i) & 1ull); if (threadIdx.x == 0) out[(size_t)blockIdx.x * R + i] = r; }
}
int main() {
    const uint32_t A = 100;
  int total_bad = 0;
  for (uint32_t n_draws : {1u, 2u, 3u, 4u, 5u, 8u, 64u, 200u}) {
    std::vector<uint32_t> rowops(A);
    uint32_t s = 7;
    auto r = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto &v : rowops) {
        const uint32_t t0 = 1 + r() % 2, t1 = r() % 3, d0 = r() % R, s0 = r() % R, d1 = r() % R, s1 = r() % R;
        v = make_op(t0, d0, s0) | (make_op(t1, d1, s1) << 14);
    }
    if (getenv("CRAFT")) {  // every action the same single row operation: type, dst, src from the environment
        int ty, d, sr; sscanf(getenv("CRAFT"), "%d,%d,%d", &ty, &d, &sr);
        for (auto &v : rowops) v = make_op((uint32_t)ty, (uint32_t)d, (uint32_t)sr);
    }
    uint32_t *d_rowops, *d_list; uint64_t *d_out;
    CK(hipMalloc(&d_rowops, A * 4)); CK(hipMemcpy(d_rowops, rowops.data(), A * 4, hipMemcpyHostToDevice));
    uint32_t zero = 0; CK(hipMalloc(&d_list, 4)); CK(hipMemcpy(d_list, &zero, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, R * 8));
    InitArgs a; memset(&a, 0, sizeof a);
    a.list = d_list; a.rowops = d_rowops; a.n_draws = n_draws; a.num_actions = A; a.seed = 1234; a.env_base = 0; a.B = 4096;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, a, d_out);
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> got(R), want(R);
    CK(hipMemcpy(got.data(), d_out, R * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < R; ++i) want[i] = 1ull << i;
    for (uint32_t t = 0; t < n_draws; ++t) {
        const uint32_t o = rowops[rng_action(1234, 0, t, A)];
        for (int kk = 0; kk < 2; ++kk) {
            const uint32_t op = (o >> (14 * kk)) & 0x3FFFu, type = op >> 12, dst = op & 63u, src = (op >> 6) & 63u;
            if (type == OP_XOR) want[dst] ^= want[src];
            else if (type == OP_SWAP) { uint64_t x = want[dst]; want[dst] = want[src]; want[src] = x; }
        }
    }
    int bad = 0;
    for (int i = 0; i < R; ++i) if (got[i] != want[i]) { if (bad < 8) printf("row %2d got %016llx want %016llx\n", i, (unsigned long long)got[i], (unsigned long long)want[i]); ++bad; }
    printf("n_draws %3u: %d rows differ\n", n_draws, bad);
    total_bad += bad;
  }
    {   // timing: 655 workgroups (1 % of 65 536 envs), 256 gates
        const uint32_t n = 655;
        std::vector<uint32_t> lst(n);
        for (uint32_t i = 0; i < n; ++i) lst[i] = i * 97u;
        uint32_t *d_l; uint64_t *d_o;
        CK(hipMalloc(&d_l, n * 4)); CK(hipMemcpy(d_l, lst.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&d_o, (size_t)n * R * 8));
        InitArgs a; memset(&a, 0, sizeof a);
        uint32_t *d_rowops2; CK(hipMalloc(&d_rowops2, A * 4));
        std::vector<uint32_t> ro(A);
        uint32_t s2 = 9; auto r2 = [&]() { s2 = s2 * 1664525u + 1013904223u; return s2 >> 8; };
        for (auto &v : ro) v = make_op(1 + r2() % 2, r2() % R, r2() % R) | (make_op(r2() % 3, r2() % R, r2() % R) << 14);
        CK(hipMemcpy(d_rowops2, ro.data(), A * 4, hipMemcpyHostToDevice));
        a.list = d_l; a.rowops = d_rowops2; a.n_draws = 256; a.num_actions = A; a.seed = 1; a.B = 65536;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kt, dim3(n), dim3(256), 0, 0, a, n, d_o);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kt, dim3(n), dim3(256), 0, 0, a, n, d_o);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scramble_tree64<%d>, %u envs x 256 gates: %.1f us per launch\n", R, n, ms * 1e3 / 20);
    }
    return total_bad != 0;
}
