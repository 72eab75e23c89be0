// api_bench.cpp -- the scalar drop-in (qg_env_*) driven the way twisterl drives a Box<dyn Env> (rayon workers, per episode
// env.clone(); reset(); loop { observe(); masks(); <policy>; step(a); reward(); is_final() }; rl/configs.py:134-135 num_cores = 32,
// num_episodes = 1024), through the C ABI only -- no Python, no torch, no oracle.  Prints one line per thread count.
//   hipcc -O2 -std=c++17 -I include tools/api_bench.cpp -L qiskit_gym_amd/lib -lqgym -Wl,-rpath,$PWD/qiskit_gym_amd/lib -lpthread -o /tmp/api_bench
//   /tmp/api_bench [episodes per thread = 64] [steps per episode = 16] [qubits = 16]
// `vec` as the first argument times qg_vec_rollout_ring instead (the batched flavour at 65 536 envs), for the comparison INTEGRATION.md quotes.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "qgym.h"
#define CK(x) do { int e_ = (int)(x); if (e_ != 0) { printf("%s failed: %d %s\n", #x, e_, qg_last_error()); exit(1); } } while (0)

static std::vector<qg_gate> line_gateset(int N) {  // envs/synthesis.py:89-103 order on a bidirectional line
    std::vector<qg_gate> g;
    for (int k = 0; k < 5; ++k) for (int q = 0; q < N; ++q) g.push_back({k, q, 0});
    for (int k = 5; k < 8; ++k) for (int q = 0; q + 1 < N; ++q) { g.push_back({k, q, q + 1}); g.push_back({k, q + 1, q}); }
    return g;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int vec_mode(int argc, char **argv) {
    const uint64_t B = argc > 2 ? strtoull(argv[2], 0, 10) : 65536;
    const int N = 16, RING = 16, T = 256;
    std::vector<qg_gate> g = line_gateset(N);
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
    for (int rep = 0; rep < 4; ++rep) {
        const int R = rep < 2 ? 8 : 64;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < R; ++r) CK(qg_vec_rollout_ring(v, d, QG_ACT_I32, T, RING, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("vec B=%llu  rep %d (%d replays): %.3f us/step  %.3e env-steps/s\n", (unsigned long long)B, rep, R, ms * 1e3 / (R * T), B * (double)R * T / (ms * 1e-3));
    }
    CK(qg_vec_sync(v, st));
    qg_vec_destroy(v);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "vec")) return vec_mode(argc, argv);
    const int episodes = argc > 1 ? atoi(argv[1]) : 64, steps = argc > 2 ? atoi(argv[2]) : 16, N = argc > 3 ? atoi(argv[3]) : 16;
    std::vector<qg_gate> g = line_gateset(N);
    const int64_t A = (int64_t)g.size();
    qg_config cfg; qg_config_default(&cfg, QG_CLIFFORD, N);   // the reference's defaults: add_inverts, add_perms, track_solution on
    cfg.difficulty = steps;  // episodes of up to depth_slope * difficulty steps; random actions rarely solve, so ~`steps` steps are taken
    qg_env *proto; CK(qg_env_create(&cfg, g.data(), g.size(), 0, &proto));
    printf("scalar API, CliffordEnv %d qubits, %lld actions, reference-default options; per thread: %d episodes x %d steps of {observe, masks, step, reward, is_final}\n",
           N, (long long)A, episodes, steps);
    for (int threads : {1, 4, 32}) {
        std::atomic<long long> n_steps{0};
        std::vector<double> t_clone(threads, 0.0), t_reset(threads, 0.0), t_loop(threads, 0.0);
        auto worker = [&](int tid) {
            std::vector<int64_t> obs(4 * N * N);
            std::vector<uint8_t> masks(A);
            uint64_t rng = 0x9E3779B97F4A7C15ull * (tid + 1);
            for (int ep = 0; ep < episodes; ++ep) {
                double t0 = now();
                qg_env *e; CK(qg_env_clone(proto, &e));
                double t1 = now();
                CK(qg_env_reset(e, (uint64_t)tid * 1000003u + ep));
                double t2 = now();
                int k = 0;
                for (; k < steps && !qg_env_is_final(e); ++k) {
                    (void)qg_env_observe(e, obs.data(), obs.size());
                    (void)qg_env_masks(e, masks.data(), masks.size());
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    CK(qg_env_step(e, (int64_t)((rng >> 33) % (uint64_t)A)));
                    (void)qg_env_reward(e);
                }
                double t3 = now();
                qg_env_destroy(e);
                double t4 = now();
                t_clone[tid] += (t1 - t0) + (t4 - t3);
                t_reset[tid] += t2 - t1;
                t_loop[tid] += t3 - t2;
                n_steps += k;
            }
        };
        {   // warm-up, not timed: kernel loads, and one pooled handle per thread (the first clone of a thread allocates; twisterl's workers
            // live for the whole run, so the steady state is what a collection sees)
            std::vector<std::thread> warm;
            for (int t = 0; t < threads; ++t) warm.emplace_back(worker, t);
            for (auto &t : warm) t.join();
        }
        n_steps = 0;
        for (auto *v : {&t_clone, &t_reset, &t_loop}) std::fill(v->begin(), v->end(), 0.0);
        const double w0 = now();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(worker, t);
        for (auto &t : pool) t.join();
        const double wall = now() - w0;
        double c = 0, r = 0, l = 0;
        for (int t = 0; t < threads; ++t) { c += t_clone[t]; r += t_reset[t]; l += t_loop[t]; }
        const long long ns = n_steps.load();
        printf("threads %2d: %8.1f env-steps/s wall | per thread: clone+destroy %7.1f us/episode, reset %7.1f us, {observe, masks, step, reward, is_final} %7.1f us/step\n",
               threads, ns / wall, c / (threads * episodes) * 1e6, r / (threads * episodes) * 1e6, l / (double)ns * 1e6);
    }
    qg_env_destroy(proto);
    return 0;
}
