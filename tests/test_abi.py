"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/qgym.h declares, mirrors the reference's gate parsing / defaults, and fails loudly (no
CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import pytest

from qiskit_gym_amd import _lib
from qiskit_gym_amd.envs.gateset import parse_gate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "qgym.h")).read()
    declared = sorted(set(re.findall(r"\b(qg_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = _lib.load()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert L.qg_abi_version() == 3


def test_config_defaults_match_reference_constructors():
    cfg = _lib.make_config("clifford", 5)
    # clifford.rs:420-422, metrics.rs:157-166, envs/synthesis.py:186-188
    assert (cfg.difficulty, cfg.depth_slope, cfg.max_depth) == (1, 2, 128)
    assert (cfg.add_inverts, cfg.add_perms, cfg.track_solution) == (1, 1, 1)
    assert abs(cfg.w_n_cnots - 0.01) < 1e-9 and cfg.w_n_layers == 0.0 and cfg.w_n_layers_cnots == 0.0
    assert abs(cfg.w_n_gates - 0.0001) < 1e-9
    p = _lib.make_config("pauli", 4)
    assert (p.max_rotations, p.pauli_diff_scale, p.final_pauli_layers) == (5, 16, -1)
    assert abs(p.pauli_layer_reward - 0.01) < 1e-9 and p.num_qubits_decay == 0.5
    w = _lib.make_config("clifford", 5, metrics_weights={"n_layers": 0.5, "bogus": 3.0})  # unknown keys ignored
    assert w.w_n_layers == 0.5


def test_gate_parse_c_and_python_agree_with_reference_rules():
    L = _lib.load()
    g = _lib.QGGate()
    idx = (C.c_int64 * 2)(3, 1)
    assert L.qg_gate_parse(b"  cX ", idx, 2, C.byref(g)) == 0 and (g.kind, g.q0, g.q1) == (5, 3, 1)
    assert parse_gate(("  cX ", [3, 1])) == (5, 3, 1)
    assert L.qg_gate_parse(b"sdg", idx, 1, C.byref(g)) == 0 and (g.kind, g.q0) == (2, 3)
    assert L.qg_gate_parse(b"h", idx, 2, C.byref(g)) == -1
    assert b"expects 1 index, got 2" in L.qg_last_error()
    assert L.qg_gate_parse(b"toffoli", idx, 2, C.byref(g)) == -1
    assert b"Unknown gate name `toffoli`" in L.qg_last_error()
    with pytest.raises(ValueError, match="expects 2 indices, got 1"):
        parse_gate(("SWAP", [1]))
    with pytest.raises(ValueError, match="Unknown gate name"):
        parse_gate(("T", [1]))
    with pytest.raises(TypeError):
        parse_gate(("H", ["a"]))
    with pytest.raises(TypeError):
        parse_gate((5, [1]))
    with pytest.raises(ValueError, match="exactly 2 items"):
        parse_gate(("H", [0], 1))


def test_no_gpu_means_loud_failure_not_fallback():
    L = _lib.load()
    if L.qg_device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = _lib.make_config("clifford", 3)
    gates = _lib.make_gates([(0, 0, 0)])
    h = C.c_void_p()
    rc = L.qg_vec_create(C.byref(cfg), gates, 1, 4, 0, C.byref(h))
    assert rc == -4 and not h.value  # QG_ERR_DEVICE
    assert b"no CPU fallback" in L.qg_last_error()
    import torch

    if not torch.cuda.is_available():
        from qiskit_gym_amd.vec import VecEnv

        with pytest.raises(RuntimeError, match="no CPU fallback"):
            VecEnv("clifford", 3, [("H", (0,))], 4)


def test_product_package_never_imports_the_oracle():
    """No file under qiskit_gym_amd/ may import, include, link or dlopen anything from oracle/."""
    pkg = os.path.join(ROOT, "qiskit_gym_amd")
    pat = re.compile(r"(import\s+oracle|from\s+oracle|qgym_oracle|libqgym_oracle|oracle/|og_env_|og_vec_|OracleEnv|OracleVec)")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(text), (dirpath, f, pat.search(text).group(0))


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/qgym.h compiles as C99 (no C++ or torch types) and a C program that
    uses only the header links against libqgym.so and runs its host-only entry points."""
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "abi_user.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "qgym.h"
int main(void) {
    qg_config cfg;
    qg_gate g;
    int64_t idx[2] = {2, 0};
    qg_config_default(&cfg, QG_CLIFFORD, 16);
    if (cfg.max_depth != 128 || cfg.add_inverts != 1) return 1;
    if (qg_gate_parse(" Cz", idx, 2, &g) != QG_OK || g.kind != QG_CZ || g.q0 != 2 || g.q1 != 0) return 2;
    if (qg_gate_parse("cz", idx, 1, &g) != QG_ERR_INVALID || !strstr(qg_last_error(), "expects 2 indices")) return 3;
    if (qg_abi_version() != QG_ABI_VERSION) return 4;
    /* no GPU here: creation must fail loudly, never fall back */
    if (qg_device_count() == 0) {
        qg_vec *v = NULL;
        if (qg_vec_create(&cfg, &g, 1, 8, 0, &v) != QG_ERR_DEVICE || v != NULL) return 5;
    }
    printf("ok %d\n", qg_device_count());
    return 0;
}
''')
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "qiskit_gym_amd", "lib")
    _lib.load()  # builds the library when missing
    exe = tmp_path / "abi_user"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lqgym",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)


def test_bench_refuses_more_gpus_than_visible_and_mismatched_world_size():
    """bench.py --gpus N starts N ranks itself (or checks the launcher's WORLD_SIZE); it never runs fewer ranks and reports N."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch

    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(torch.cuda.device_count() + 1), "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 2 and "visible" in res.stderr and "metric" not in res.stdout
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], capture_output=True, text=True,
                         env=dict(env, WORLD_SIZE="2"), timeout=300)
    assert res.returncode == 2 and "disagree" in res.stderr
