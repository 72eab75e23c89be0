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
