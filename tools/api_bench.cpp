// api_bench.cpp -- times qg_vec_rollout_ring through the C ABI without Python/torch (dev tool).
// hipcc -O2 -I include tools/api_bench.cpp -L qiskit_gym_amd/lib -lqgym -Wl,-rpath,$PWD/qiskit_gym_amd/lib -o tools/bin/api_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "qgym.h"
#define CK(x) do { int e = (int)(x); if (e != 0) { printf("%s failed: %d %s\n", #x, e, qg_last_error()); exit(1);} } while (0)
int main(int argc, char **argv) {
    const uint64_t B = argc > 1 ? strtoull(argv[1], 0, 10) : 65536;
    const int N = 16, RING = 16, T = 256;
    std::vector<qg_gate> g;
    for (int k = 0; k < 5; ++k) for (int q = 0; q < N; ++q) g.push_back({k, q, 0});
    for (int k = 5; k < 8; ++k) for (int q = 0; q + 1 < N; ++q) { g.push_back({k, q, q + 1}); g.push_back({k, q + 1, q}); }
    qg_config cfg; qg_config_default(&cfg, QG_CLIFFORD, N);
    cfg.add_inverts = 0; cfg.add_perms = 0; cfg.track_solution = 0; cfg.difficulty = 256;
    qg_vec *v; CK(qg_vec_create(&cfg, g.data(), g.size(), B, 0, &v));
    std::vector<int32_t> h(B * RING);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (int32_t)((i * 2654435761u >> 7) % g.size());
    int32_t *d; CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(qg_vec_reset(v, 1, st));
    CK(qg_vec_rollout_ring(v, d, QG_ACT_I32, T, RING, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        const int R = rep < 2 ? 8 : 64;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < R; ++r) CK(qg_vec_rollout_ring(v, d, QG_ACT_I32, T, RING, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("B=%llu  rep %d (%d replays): %.3f us/step  %.3e env-steps/s\n", (unsigned long long)B, rep, R, ms * 1e3 / (R * T), B * (double)R * T / (ms * 1e-3));
    }
    CK(qg_vec_sync(v, st));
    qg_vec_destroy(v);
    return 0;
}
