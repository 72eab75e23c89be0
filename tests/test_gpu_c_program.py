"""A plain C program (gcc, only include/qgym.h) drives the scalar `Env` flavour of the ABI on the GPU --
what a cgo / Rust-FFI host would do -- and its outputs are compared with the CPU oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import OracleEnv  # noqa: E402
from util import line_gateset  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include "qgym.h"
#define CHECK(x) do { int rc_ = (x); if (rc_ != QG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, qg_last_error()); return 10; } } while (0)
int main(void) {
    static const qg_gate gates[] = { GATES };
    static const int64_t state[] = { STATE };
    static const int64_t actions[] = { ACTIONS };
    static const int coins[] = { COINS };
    const size_t n_gates = sizeof gates / sizeof gates[0], n_act = sizeof actions / sizeof actions[0];
    qg_config cfg;
    qg_env *env = NULL, *twin = NULL;
    int64_t shape[2], obs[4096];
    uint64_t sol[1024];
    size_t t;
    qg_config_default(&cfg, QG_CLIFFORD, NQ);
    cfg.add_perms = 0;
    CHECK(qg_env_create(&cfg, gates, n_gates, 0, &env));
    CHECK(qg_env_obs_shape(env, shape));
    printf("shape %lld %lld actions %lld\n", (long long)shape[0], (long long)shape[1], (long long)qg_env_num_actions(env));
    CHECK(qg_env_set_state(env, state, sizeof state / sizeof state[0]));
    for (t = 0; t < n_act; ++t) {
        int64_t n, i;
        union { float f; uint32_t u; } r;
        if (t == n_act / 2) CHECK(qg_env_clone(env, &twin));
        CHECK(qg_env_step_coin(env, actions[t], coins[t]));
        r.f = qg_env_reward(env);
        n = qg_env_observe(env, obs, 4096);
        printf("step %u %d %d", (unsigned)r.u, qg_env_success(env), qg_env_is_final(env));
        for (i = 0; i < n; ++i) printf(" %lld", (long long)obs[i]);
        printf("\n");
    }
    {
        int64_t n = qg_env_solution(env, sol, 1024), i;
        printf("solution");
        for (i = 0; i < n; ++i) printf(" %llu", (unsigned long long)sol[i]);
        printf("\n");
        n = qg_env_observe(twin, obs, 4096);
        printf("twin");
        for (i = 0; i < n; ++i) printf(" %lld", (long long)obs[i]);
        printf("\n");
    }
    qg_env_destroy(twin);
    qg_env_destroy(env);
    return 0;
}
'''


def test_c_program_steps_a_clifford_env_like_the_oracle(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from qiskit_gym_amd import _lib
    from qiskit_gym_amd.envs.gateset import parse_gateset

    n = 5
    gs = line_gateset("clifford", n)
    rng = np.random.default_rng(11)
    ora = OracleEnv("clifford", n, gs, add_perms=0)  # add_inverts and track_solution default to on (clifford.rs:420-422)
    ora.difficulty = 12
    ora.reset_with(rng.integers(0, len(gs), size=12))
    start = ora.get_state().tolist()
    ora.set_state(start)
    actions = rng.integers(0, len(gs), size=14).tolist()
    coins = rng.integers(0, 2, size=14).tolist()
    src = (C_SRC.replace("GATES", ", ".join("{%d, %d, %d}" % g for g in parse_gateset(gs)))
           .replace("STATE", ", ".join(map(str, start))).replace("ACTIONS", ", ".join(map(str, actions)))
           .replace("COINS", ", ".join(map(str, coins))).replace("NQ", str(n)))
    (tmp_path / "host.c").write_text(src)
    _lib.load()
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "qiskit_gym_amd", "lib")
    exe = tmp_path / "host"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(tmp_path / "host.c"), "-o", str(exe), "-L", libdir, "-lqgym",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = out.stdout.strip().splitlines()
    assert lines[0] == f"shape {2 * n} {2 * n} actions {len(gs)}"
    twin_obs = None
    for t, (a, c) in enumerate(zip(actions, coins)):
        if t == len(actions) // 2:
            twin_obs = ora.observe()
        ora.step(int(a), int(c))
        want = f"step {ora.reward_bits()} {int(ora.success())} {int(ora.is_final())}" + "".join(f" {i}" for i in ora.observe())
        assert lines[1 + t] == want, (t, lines[1 + t], want)
    assert lines[1 + len(actions)] == "solution" + "".join(f" {i}" for i in ora.solution())
    assert lines[2 + len(actions)] == "twin" + "".join(f" {i}" for i in twin_obs)


LOOP_SRC = r'''
/* The whole collection loop from C: libqgym + the HIP runtime for memory, no tensor library, no Python. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "qgym.h"
#define CHECK(x) do { int rc_ = (x); if (rc_ != QG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, qg_last_error()); return 10; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 11; } } while (0)
static uint32_t lcg_state = 12345u;
static float lcg(float scale) { lcg_state = lcg_state * 1664525u + 1013904223u; return ((float)(lcg_state >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale; }
static float *upload(size_t n, float scale) {
    float *h = (float *)malloc(n * sizeof(float)), *d = NULL;
    size_t i;
    for (i = 0; i < n; ++i) h[i] = lcg(scale);
    if (hipMalloc((void **)&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    free(h);
    return d;
}
int main(void) {
    static const qg_gate gates[] = { GATES };
    const size_t n_gates = sizeof gates / sizeof gates[0];
    enum { B = 4096, OBS = 4 * NQ * NQ, H1 = 512, H2 = 256, T = 6 };
    const uint32_t A = (uint32_t)n_gates;
    qg_config cfg;
    qg_vec *v = NULL;
    float *w1, *b1, *w2, *b2, *w3, *b3, *logp, *entropy, *values, *reward_h;
    void *p1, *p2, *p3, *h1;
    int64_t *actions, *actions_h;
    qg_vec_info info;
    int t;
    qg_config_default(&cfg, QG_CLIFFORD, NQ);
    cfg.add_inverts = 0; cfg.add_perms = 0; cfg.track_solution = 0; cfg.difficulty = 6;
    CHECK(qg_vec_create(&cfg, gates, n_gates, B, 0, &v));
    CHECK(qg_vec_get_info(v, &info));
    w1 = upload((size_t)H1 * OBS, 0.2f); b1 = upload(H1, 0.1f);
    w2 = upload((size_t)H2 * H1, 0.1f);  b2 = upload(H2, 0.1f);
    w3 = upload((size_t)(A + 1) * H2, 0.2f); b3 = upload(A + 1, 0.1f);
    if (!w1 || !b1 || !w2 || !b2 || !w3 || !b3) return 12;
    HIP(hipMalloc(&p1, qg_vec_embed_packed_bytes(v, H1)));
    HIP(hipMalloc(&p2, qg_policy_mid_packed_bytes(H1, H2)));
    HIP(hipMalloc(&p3, qg_policy_head_packed_bytes(A, H2)));
    HIP(hipMalloc(&h1, (size_t)B * H1 * 2));
    HIP(hipMalloc((void **)&actions, B * sizeof(int64_t)));
    HIP(hipMalloc((void **)&logp, B * sizeof(float)));
    HIP(hipMalloc((void **)&entropy, B * sizeof(float)));
    HIP(hipMalloc((void **)&values, B * sizeof(float)));
    actions_h = (int64_t *)malloc(B * sizeof(int64_t));
    reward_h = (float *)malloc(B * sizeof(float));
    CHECK(qg_vec_pack_embedding(v, w1, QG_DT_F32, OBS, H1, p1, NULL));
    CHECK(qg_policy_pack_mid(w2, b2, QG_DT_F32, H1, H1, H2, p2, NULL));
    CHECK(qg_policy_pack_head(w3, b3, QG_DT_F32, H2, H2, A, (int32_t)A, 1, p3, NULL));
    for (t = 0; t < T; ++t) {
        unsigned long long asum = 0, rsum = 0;
        int i;
        CHECK(qg_vec_reset_done(v, 1000u + (unsigned)t, NULL));     /* a fresh env is final: step 0 scrambles all of them */
        CHECK(qg_vec_embed(v, p1, b1, H1, 1, h1, H1, NULL));
        CHECK(qg_policy_mid_head_sample(h1, H1, B, H1, p2, H2, p3, A, 77, (uint64_t)t, NULL, actions, QG_ACT_I64, logp, entropy, values, NULL));
        CHECK(qg_vec_step(v, actions, QG_ACT_I64, NULL, NULL));
        CHECK(qg_vec_sync(v, NULL));
        HIP(hipMemcpy(actions_h, actions, B * sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(reward_h, info.reward_dev, B * sizeof(float), hipMemcpyDeviceToHost));
        for (i = 0; i < B; ++i) {
            uint32_t bits;
            memcpy(&bits, &reward_h[i], 4);
            asum = asum * 31u + (unsigned long long)actions_h[i];
            rsum = rsum * 31u + bits;
        }
        printf("step %d actions %llu rewards %llu\n", t, asum, rsum);
    }
    qg_vec_destroy(v);
    return 0;
}
'''


def test_c_program_runs_the_policy_loop_without_a_tensor_library(tmp_path):
    """embed -> mid_head_sample -> step from plain C; the same loop through the Python wrappers gives the same stream."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    import torch

    from qiskit_gym_amd import _lib
    from qiskit_gym_amd.collector import embed, mid_head_sample, pack_embedding, pack_head, pack_mid
    from qiskit_gym_amd.envs.gateset import parse_gateset
    from qiskit_gym_amd.vec import VecEnv

    n, B, H1, H2, T = 7, 4096, 512, 256, 6
    gs = line_gateset("clifford", n)
    A, OBS = len(gs), 4 * n * n
    src = LOOP_SRC.replace("GATES", ", ".join("{%d, %d, %d}" % g for g in parse_gateset(gs))).replace("NQ", str(n))
    (tmp_path / "loop.c").write_text(src)
    _lib.load()
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "qiskit_gym_amd", "lib")
    exe = tmp_path / "loop"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", inc, "-I", "/opt/rocm/include", str(tmp_path / "loop.c"), "-o", str(exe), "-L", libdir, "-lqgym",
                    "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    lines = out.stdout.strip().splitlines()

    state = np.uint32(12345)

    def lcg_array(count, scale):
        nonlocal state
        vals = np.empty(count, dtype=np.float32)
        s = int(state)
        for i in range(count):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            vals[i] = (np.float32(s >> 8) * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * np.float32(scale)
        state = np.uint32(s)
        return torch.from_numpy(vals).cuda()

    w1, b1 = lcg_array(H1 * OBS, 0.2).view(H1, OBS), lcg_array(H1, 0.1)
    w2, b2 = lcg_array(H2 * H1, 0.1).view(H2, H1), lcg_array(H2, 0.1)
    w3, b3 = lcg_array((A + 1) * H2, 0.2).view(A + 1, H2), lcg_array(A + 1, 0.1)
    env = VecEnv("clifford", n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=6)
    p1, p2, p3 = pack_embedding(env, w1), pack_mid(w2, b2), pack_head(w3, b3, A, A, after_mid=True)
    for t in range(T):
        env.reset_done(1000 + t)
        h1 = embed(env, p1, b1, H1, relu=True)
        acts, *_ = mid_head_sample(h1, p2, H2, p3, A, 77, t)
        env.step(acts)
        env.sync()
        asum = rsum = 0
        for a, r in zip(acts.cpu().numpy().tolist(), env.reward.cpu().numpy().view(np.uint32).tolist()):
            asum = (asum * 31 + a) & 0xFFFFFFFFFFFFFFFF
            rsum = (rsum * 31 + r) & 0xFFFFFFFFFFFFFFFF
        assert lines[t] == f"step {t} actions {asum} rewards {rsum}", (t, lines[t])


HOST_PTR_SRC = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include "qgym.h"
#define CHECK(x) do { int rc_ = (x); if (rc_ != QG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, qg_last_error()); return 10; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 11; } } while (0)
enum { NQ = 12, B = 900, T = 7 };
int main(void) {
    static const qg_gate gates[] = { GATES };
    const size_t n_gates = sizeof gates / sizeof gates[0];
    qg_config cfg;
    qg_vec *v = NULL;
    qg_vec_info info;
    hipStream_t s;
    int64_t *act = NULL;      /* pinned */
    uint8_t *coin = NULL, *done = NULL, *succ = NULL;
    float *rew = NULL;
    uint32_t *packed = NULL;
    int8_t *dense = (int8_t *)malloc((size_t)B * 4 * NQ * NQ);   /* pageable on purpose */
    int t, e, k;
    qg_config_default(&cfg, QG_CLIFFORD, NQ);
    cfg.add_perms = 0; cfg.track_solution = 0; cfg.difficulty = 9;   /* add_inverts stays on: the coins travel too */
    CHECK(qg_vec_create(&cfg, gates, n_gates, B, 0, &v));
    CHECK(qg_vec_get_info(v, &info));
    HIP(hipStreamCreate(&s));
    HIP(hipHostMalloc((void **)&act, sizeof(int64_t) * B * T, 0));
    HIP(hipHostMalloc((void **)&coin, (size_t)B * T, 0));
    HIP(hipHostMalloc((void **)&rew, sizeof(float) * B * T, 0));
    HIP(hipHostMalloc((void **)&done, (size_t)B * T, 0));
    HIP(hipHostMalloc((void **)&succ, (size_t)B * T, 0));
    HIP(hipHostMalloc((void **)&packed, sizeof(uint32_t) * B * 2 * NQ, 0));
    for (t = 0; t < T; ++t) for (e = 0; e < B; ++e) {
        act[t * B + e] = (e * 5 + t * 11) % ((int)n_gates + 2) - 1;   /* includes -1 and n_gates: no gate, depth still decrements */
        coin[t * B + e] = (uint8_t)((e * 3 + t) % 3 == 0);
    }
    CHECK(qg_vec_reset(v, 7, s));
    for (t = 0; t < T; ++t)   /* nothing synchronises inside the loop */
        CHECK(qg_vec_step_host(v, act + t * B, QG_ACT_I64, coin + t * B, rew + t * B, done + t * B, succ + t * B, s));
    CHECK(qg_vec_observe_packed_host(v, packed, s));
    CHECK(qg_vec_observe_dense_host(v, dense, s));
    CHECK(qg_vec_sync(v, s));
    for (t = 0; t < T; ++t) for (e = 0; e < B; ++e) {
        uint32_t rb; memcpy(&rb, &rew[t * B + e], 4);
        printf("r %u %u %u\n", rb, done[t * B + e], succ[t * B + e]);
    }
    for (e = 0; e < B; ++e) {
        printf("o");
        for (k = 0; k < 2 * NQ; ++k) {
            uint32_t w = 0; int c;
            for (c = 0; c < 2 * NQ; ++c) w |= (uint32_t)(dense[((size_t)e * 2 * NQ + k) * 2 * NQ + c] != 0) << c;
            if (w != packed[e * 2 * NQ + k]) { fprintf(stderr, "dense and packed observation differ\n"); return 12; }
            printf(" %u", w);
        }
        printf("\n");
    }
    qg_vec_destroy(v);
    return 0;
}
'''


def test_c_host_steps_with_host_pointers(tmp_path):
    """qg_vec_step_host / qg_vec_observe_*_host (SURVEY 8b host-pointer variants): pinned actions and coins in, rewards and flags out, seven
    steps enqueued back to back with no synchronisation, then both observation formats; compared with the oracle step by step."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from oracle import OracleVec
    from qiskit_gym_amd.envs.gateset import parse_gateset
    from util import f32_bits, rng_actions

    NQ, B, T = 12, 900, 7
    gs = line_gateset("clifford", NQ)
    A = len(gs)
    src = HOST_PTR_SRC.replace("GATES", ", ".join("{%d, %d, %d}" % g for g in parse_gateset(gs)))
    (tmp_path / "hp.c").write_text(src)
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "qiskit_gym_amd", "lib")
    exe = tmp_path / "hp"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", inc, "-I", "/opt/rocm/include", str(tmp_path / "hp.c"), "-o", str(exe), "-L", libdir, "-lqgym",
                    "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout[-500:], out.stderr[-2000:])
    lines = out.stdout.splitlines()
    rl = [ln.split() for ln in lines if ln.startswith("r ")]
    ol = [ln.split() for ln in lines if ln.startswith("o ")]
    assert len(rl) == B * T and len(ol) == B
    proto = OracleEnv("clifford", NQ, gs, add_inverts=1, add_perms=0, track_solution=0, difficulty=9)
    ov = OracleVec(proto, B)
    ov.reset_with(rng_actions(7, B, 9, A))
    e = np.arange(B)
    for t in range(T):
        acts = (e * 5 + t * 11) % (A + 2) - 1
        coins = ((e * 3 + t) % 3 == 0).astype(np.uint8)
        r, s, f, _ = ov.step(acts, coins)
        got = np.array([[int(x) for x in row[1:]] for row in rl[t * B:(t + 1) * B]], dtype=np.int64)
        assert np.array_equal(got[:, 0], f32_bits(r).astype(np.int64)), t
        assert np.array_equal(got[:, 1], f) and np.array_equal(got[:, 2], s)
    dense = ov.observe_dense().reshape(B, 2 * NQ, 2 * NQ).astype(np.uint64)
    want = (dense << np.arange(2 * NQ, dtype=np.uint64)).sum(axis=2)
    assert np.array_equal(np.array([[int(x) for x in row[1:]] for row in ol], dtype=np.uint64), want)
