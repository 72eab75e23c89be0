"""Shared helpers for the parity tests: build the same env on the CPU oracle and on the HIP path."""
from __future__ import annotations

import numpy as np

from oracle import OracleEnv, OracleVec
from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, grid_edges, line_edges

ALLOWED = {
    "linear_function": ["CX", "SWAP"],
    "permutation": ["SWAP"],
    "clifford": ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"],
    "pauli": ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"],
}


def line_gateset(kind: str, n: int, basis=None):
    return gateset_from_coupling_map(line_edges(n, True), basis, ALLOWED[kind])[1]


def grid_gateset(kind: str, rows: int, cols: int, basis=None, bidirectional=False):
    return gateset_from_coupling_map(grid_edges(rows, cols, bidirectional), basis, ALLOWED[kind])[1]


def oracle_cfg(cfg: dict) -> dict:
    """VecEnv config kwargs -> OracleEnv kwargs (bools to ints)."""
    out = {}
    for k, v in cfg.items():
        out[k] = int(v) if isinstance(v, bool) else v
    return out


def make_pair(kind, num_qubits, gateset, batch, **cfg):
    from qiskit_gym_amd.vec import VecEnv

    proto = OracleEnv(kind, num_qubits, gateset, **oracle_cfg(cfg))
    ov = OracleVec(proto, batch)
    gv = VecEnv(kind, num_qubits, gateset, batch, **cfg)
    return ov, gv


def f32_bits(x) -> np.ndarray:
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def rng_draw(seed: int, env: np.ndarray, t: int) -> np.ndarray:
    """qg::rng_draw of qiskit_gym_amd/csrc/qgym_internal.hpp on numpy uint64 arrays."""
    with np.errstate(over="ignore"):
        inner = splitmix64(env.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(t))
        return splitmix64(np.uint64(seed) ^ inner)


def rng_actions(seed: int, batch, n_draws: int, num_actions: int) -> np.ndarray:
    """The draws qg_vec_reset(seed) uses: mulhi64(rng_draw, num_actions); shape [n_draws, B].
    `batch`: an int (envs 0..B-1) or an array of env indices."""
    env = np.arange(batch, dtype=np.uint64) if np.isscalar(batch) else np.asarray(batch, dtype=np.uint64)
    out = np.zeros((n_draws, env.size), dtype=np.int64)
    a, lo32, s32 = np.uint64(num_actions), np.uint64(0xFFFFFFFF), np.uint64(32)
    assert 0 < num_actions < 2**32
    for t in range(n_draws):
        d = rng_draw(seed, env, t)
        out[t] = ((d >> s32) * a + (((d & lo32) * a) >> s32)) >> s32  # (d * a) >> 64 without 128-bit integers
    return out
