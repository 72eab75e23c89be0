"""Pin the CPU oracle against every known-answer vector the reference ships
(tests/golden/*.json, extracted by tests/golden/make_golden.py from the reference's recorded
notebook outputs and example model configs)."""
import json
import os

import numpy as np
import pytest

from oracle import OracleEnv
from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map

ALLOWED = {
    "linear_function": ["CX", "SWAP"],
    "permutation": ["SWAP"],
    "clifford": ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"],
}


def load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def norm(gs):
    return [(n, tuple(q)) for n, q in gs]


def test_gateset_orderings(golden_dir):
    g = load(golden_dir, "gatesets.json")
    for key in ("lf_line3_bidirectional_cx_swap", "perm_grid3x3_unidirectional_swap"):
        rec = g[key]
        n, gs = gateset_from_coupling_map(rec["edges"], rec["basis_gates"], ALLOWED[rec["env"]])
        assert norm(gs) == norm(rec["gateset"]), key
    # model JSONs: the 3x3 grid model is the same 12 SWAPs; lf_5_line is line-5 with basis CX
    m = g["model_perm_square_3x3"]["env"]
    assert norm(m["gateset"]) == norm(g["perm_grid3x3_unidirectional_swap"]["gateset"])
    assert m["num_qubits"] == 9
    from qiskit_gym_amd.envs.gateset import line_edges
    n, gs = gateset_from_coupling_map(line_edges(5, True), ["CX"], ALLOWED["linear_function"])
    assert n == 5 and norm(gs) == norm(g["model_lf_5_line"]["env"]["gateset"])


def test_lf_notebook_transcripts(golden_dir):
    d = load(golden_dir, "lf_line3_transcripts.json")
    for seq in d["sequences"]:
        env = OracleEnv("linear_function", d["num_qubits"], d["gateset"], add_inverts=0, add_perms=0)
        assert env.num_actions() == d["action_space_n"]
        assert env.obs_shape() == d["obs_shape"]
        env.set_state(np.array(d["start_state"]).flatten().tolist())
        assert env.dense_obs().tolist() == d["start_state"]
        assert env.dense_obs().dtype == np.int8
        for a, want, fin in zip(seq["actions"], seq["states"], seq["is_final"]):
            assert not env.is_final()  # adapters.py:63-65
            env.step(a)
            assert env.dense_obs().tolist() == want, (seq["source"], a)
            assert env.is_final() == fin
        if seq["is_final"][-1]:
            assert env.success() and env.reward() > 0.9


@pytest.mark.parametrize("key", ["permutation_swap_0_8", "linear_function_cx_0_4", "clifford_h_2"])
def test_recorded_synthesis_outputs_solve_their_inputs(golden_dir, key):
    """The circuit the reference printed for an input, replayed as actions from get_state(input),
    must end in the solved state exactly at its last gate."""
    rec = load(golden_dir, "notebook_solutions.json")[key]
    gs = norm(rec["gateset"])
    env = OracleEnv(rec["env"], rec["num_qubits"], gs, add_inverts=0, add_perms=0)
    env.set_state(np.array(rec["state"]).flatten().tolist())
    assert not env.success()
    actions = [gs.index((n, tuple(q))) for n, q in rec["circuit"]]
    for i, a in enumerate(actions):
        assert not env.is_final(), i
        env.step(a)
    assert env.success() and env.is_final()
    assert env.solution() == actions  # track_solution default ON, no inversions


def test_default_weight_penalties_bit_patterns():
    """f32 constants implied by metrics.rs:135-166 for the default weights."""
    gs = [("H", (0,)), ("CX", (0, 1)), ("CZ", (0, 1)), ("SWAP", (0, 1))]
    want = {0: 0x38D1B717, 1: 0x3C257A78, 2: 0x3C28C155, 3: 0x3CF837B4}
    for a, bits in want.items():
        env = OracleEnv("clifford", 2, gs, add_inverts=0, add_perms=0)
        env.set_state([1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 0, 0, 1, 1, 1])  # any non-identity state
        env.step(a)
        pen = np.float32(0.0) - np.float32(env.reward())
        assert not env.success()
        assert int(np.float32(pen).view(np.uint32)) == bits, hex(int(np.float32(pen).view(np.uint32)))
