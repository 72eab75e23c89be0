"""Env::twists of the HIP library (qg_env_twists, qgym_env.cpp) against the oracle's restatement of the reference's
rust/src/envs/symmetry.rs (oracle/qgym_oracle_symmetry.c, pinned on CPU by tests/test_oracle_symmetry.py), and PauliEnv's
internal qubit / action permutations (pauli.rs:374-378, 443-485, 594-599) against the oracle env that computes its own."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import OracleEnv, qubit_perms
from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, grid_edges, line_edges
from qiskit_gym_amd.envs.raw import RawEnv
from test_gpu_pauli import random_labels, random_tableau
from test_oracle_symmetry import GRAPHS
from util import ALLOWED, f32_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", sorted(GRAPHS))
@pytest.mark.parametrize("kind", ["clifford", "linear_function", "permutation"])
def test_twists_equal_the_oracles(graph, kind):
    n, edges = GRAPHS[graph]
    gs = gateset_from_coupling_map(edges, None, ALLOWED[kind])[1]
    env = RawEnv(kind, n, gs, add_inverts=False, add_perms=True)
    ora = OracleEnv(kind, n, gs, add_inverts=0, add_perms=1)
    got, want = env.twists(), ora.twists()
    assert len(want[0]) >= 1
    assert got[1] == want[1], "action permutations"
    assert got[0] == want[0], "observation permutations"
    assert RawEnv(kind, n, gs, add_inverts=False, add_perms=False).twists() == ([], [])


def test_twists_of_the_reference_gatesets_and_odd_ones(golden_dir):
    gs_all = json.load(open(os.path.join(golden_dir, "gatesets.json")))
    cases = [("clifford", 3, [(g[0], tuple(g[1])) for g in gs_all["model_clifford_3q_custom"]["env"]["gateset"]]),  # asymmetric 1q gates
             ("permutation", 9, [(g[0], tuple(g[1])) for g in gs_all["model_perm_square_3x3"]["env"]["gateset"]]),
             ("linear_function", 5, [(g[0], tuple(g[1])) for g in gs_all["model_lf_5_line"]["env"]["gateset"]]),
             ("clifford", 4, [("H", (q,)) for q in range(4)] + [("S", (q,)) for q in range(4)]),  # no edge: all 24 perms, Heap's order
             ("clifford", 3, [("H", (0,)), ("H", (1,)), ("H", (2,)), ("CX", (1, 1))]),  # a two-qubit gate on one qubit is no edge
             ("linear_function", 4, gateset_from_coupling_map(line_edges(4, False), ["CX"], ALLOWED["linear_function"])[1]),
             ("permutation", 4, gateset_from_coupling_map(line_edges(4, False), ["SWAP"], ALLOWED["permutation"])[1]),
             ("clifford", 5, [("CX", (0, 1)), ("CX", (1, 2)), ("SWAP", (2, 1)), ("SWAP", (1, 2)), ("H", (3,)), ("H", (4,))])]  # duplicates, isolated qubits
    for kind, n, gs in cases:
        env = RawEnv(kind, n, gs, add_inverts=False, add_perms=True)
        ora = OracleEnv(kind, n, gs, add_inverts=0, add_perms=1)
        assert env.twists() == ora.twists(), (kind, n, gs)


@pytest.mark.parametrize("graph", ["ring6", "grid2x3", "heavy_hex7", "line5"])
def test_pauli_env_permutations_equal_the_oracles(graph):
    """PauliEnv keeps compute_qubit_perms' output to itself (twists() is empty, pauli.rs:675-679): compared through what it does with
    it -- observe() under every permutation index and the un-permuted action of the following step."""
    from qiskit_gym_amd.vec import VecEnv

    n, edges = GRAPHS[graph]
    gs = gateset_from_coupling_map(edges, None, ALLOWED["pauli"])[1]
    A = len(gs)
    qp, ap = qubit_perms(n, gs)
    P = len(qp)
    cfg = dict(add_perms=True, track_solution=True, max_rotations=4, max_depth=64)
    assert RawEnv("pauli", n, gs, **cfg).twists() == ([], [])
    batch = 4 * P
    gv = VecEnv("pauli", n, gs, batch, **cfg)
    assert gv.pauli_num_perms() == P
    envs = [OracleEnv("pauli", n, gs, **{k: int(v) for k, v in cfg.items()}) for _ in range(batch)]  # each computes its own perms
    rng = np.random.default_rng(P)
    pairs = [g[1] for g in gs if g[0] == "CX"]
    tabs, labs = [], []
    for o in envs:
        t = random_tableau(rng, n, 12, pairs)
        l = random_labels(rng, n, int(rng.integers(0, 5)), 3)
        o.pauli_reset_from(t, l)
        tabs.append(t)
        labs.append(l)
    gv.pauli_reset_from(np.stack(tabs), labs)
    for t in range(10):
        draws = (np.arange(batch) + t) % P  # every permutation index is exercised
        got = gv.pauli_observe(torch.as_tensor(draws, device="cuda", dtype=torch.int32)).cpu().numpy()
        want = np.stack([o.dense_obs(int(d)) for o, d in zip(envs, draws)])
        np.testing.assert_array_equal(got, want, err_msg=f"obs t={t}")
        acts = rng.integers(0, A, size=batch)
        for o, a in zip(envs, acts):
            o.step(int(a))
        gv.step(torch.as_tensor(acts, device="cuda", dtype=torch.int32))
        gv.sync()
        np.testing.assert_array_equal(f32_bits(gv.reward.cpu().numpy()), np.array([o.reward_bits() for o in envs], dtype=np.uint32))
    for e in range(batch):
        assert gv.solution(e) == envs[e].solution()  # the log holds the un-permuted (actual) actions
