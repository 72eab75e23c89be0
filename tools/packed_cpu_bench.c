/* packed_cpu_bench.c -- SURVEY.md 8(d) "also time the packed CPU backend for context": CliffordEnv's step (rust/src/envs/clifford.rs:321-347)
 * on bit-packed rows on the HOST -- one uint32 per tableau row, 32 rows = 128 bytes per env, env-major -- OpenMP over envs.  A context
 * figure next to the byte-per-entry port (oracle/, bench.py's cpu_baseline) and the GPU path; a development tool, not part of the product
 * and not the oracle (tools/packed_cpu_bench.py checks it against the oracle before it times it).
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC tools/packed_cpu_bench.c -o tools/bin/libpacked_cpu.so */
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint8_t type[2], dst[2], src[2]; /* two row operations: 0 none, 1 row[dst] ^= row[src], 2 swap (clifford.rs:64-82, 89-133) */
    uint8_t pad[2];
    float penalty;                   /* metrics-weighted penalty of the action, f32 (metrics.rs:135-146; default weights: a function of the gate) */
} pk_gate;

/* one env.step() for every env; rows[e * D + r]; out-of-range actions are no-ops that still use depth (clifford.rs:324,342) */
void pk_step(uint32_t *rows, int32_t *depth, float *reward, uint8_t *done, uint8_t *success, const int32_t *actions, const pk_gate *gates,
             int32_t num_actions, int64_t B, int32_t D, int32_t threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < B; ++e) {
        uint32_t *m = rows + e * D;
        const int32_t a = actions[e];
        float penalty = 0.0f;
        if (a >= 0 && a < num_actions) {
            const pk_gate g = gates[a];
            penalty = g.penalty;
            for (int k = 0; k < 2; ++k) {
                if (g.type[k] == 1) m[g.dst[k]] ^= m[g.src[k]];
                else if (g.type[k] == 2) { const uint32_t t = m[g.dst[k]]; m[g.dst[k]] = m[g.src[k]]; m[g.src[k]] = t; }
            }
        }
        int32_t d = depth[e];
        d = d > 0 ? d - 1 : 0;
        uint32_t diff = 0;
        for (int r = 0; r < D; ++r) diff |= m[r] ^ (1u << r); /* CFState::solved (clifford.rs:136-145) */
        const int solved = diff == 0;
        depth[e] = d;
        reward[e] = (solved ? 1.0f : 0.0f) - penalty;
        success[e] = (uint8_t)solved;
        done[e] = (uint8_t)(d == 0 || solved);
    }
}
