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
    assert L.qg_abi_version() == 1


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
