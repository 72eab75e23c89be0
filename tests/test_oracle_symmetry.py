"""Pins the oracle's restatement of the reference's twists (rust/src/envs/symmetry.rs, oracle/qgym_oracle_symmetry.c) on CPU:
the automorphism set against a brute force over all n! permutations, the Heap's-algorithm order of the no-edge branch against a
hand-derived sequence, the action / observation permutations against the rules written out in Python, and the gatesets the
reference ships (tests/golden/gatesets.json)."""
import itertools
import json
import os

import pytest

from oracle import OracleEnv, qubit_perms
from qiskit_gym_amd.envs.gateset import gateset_from_coupling_map, grid_edges, line_edges
from util import ALLOWED

HEAVY_HEX_7 = [(0, 1), (1, 2), (1, 3), (3, 5), (4, 5), (5, 6)]  # the 7-qubit H-shaped fragment of a heavy-hex lattice


def both_ways(edges):
    return sorted(set(edges) | {(b, a) for a, b in edges})


GRAPHS = {
    "line5": (5, line_edges(5, True)),
    "line4_one_way": (4, line_edges(4, False)),
    "ring6": (6, both_ways([(i, (i + 1) % 6) for i in range(6)])),
    "grid3x3": (9, grid_edges(3, 3, True)),
    "grid2x3": (6, grid_edges(2, 3, False)),
    "star5": (5, both_ways([(0, k) for k in range(1, 5)])),
    "heavy_hex7": (7, both_ways(HEAVY_HEX_7)),
    "two_components": (6, both_ways([(0, 1), (1, 2), (3, 4)])),  # qubit 5 isolated
}


def brute_force_automorphisms(n, edges):
    adj = {(a, b) for a, b in edges} | {(b, a) for a, b in edges}
    out = []
    for p in itertools.permutations(range(n)):
        if all(((p[a], p[b]) in adj) == ((a, b) in adj) for a in range(n) for b in range(n) if a != b):
            out.append(list(p))
    return sorted(out)


def key(name, qs):
    return (name.upper(), tuple(sorted(qs)) if name.upper() == "SWAP" else tuple(qs))


def expected_perms(n, gateset, autos):
    """compute_twists_with_builder / compute_qubit_perms (symmetry.rs:205-263, 307-361) written out: keep the automorphisms that map
    the gateset onto itself; the gate index keeps the LAST gate of each canonical key."""
    index = {}
    for i, (name, qs) in enumerate(gateset):
        index[key(name, qs)] = i
    qp, ap = [], []
    for p in autos:
        row = []
        for name, qs in gateset:
            k = key(name, [p[q] for q in qs])
            if k not in index:
                row = None
                break
            row.append(index[k])
        if row is not None:
            qp.append(list(p))
            ap.append(row)
    return qp, ap


@pytest.mark.parametrize("graph", sorted(GRAPHS))
@pytest.mark.parametrize("kind", ["clifford", "linear_function", "permutation"])
def test_qubit_and_action_perms_match_brute_force(graph, kind):
    n, edges = GRAPHS[graph]
    gs = gateset_from_coupling_map(edges, None, ALLOWED[kind])[1]
    autos = brute_force_automorphisms(n, edges)
    want_q, want_a = expected_perms(n, gs, autos)
    got_q, got_a = qubit_perms(n, gs)
    assert got_q == want_q and got_a == want_a
    assert len(got_q) >= 1 and got_q[0] == list(range(n))  # sorted: the identity comes first
    env = OracleEnv(kind, n, gs, add_perms=1)
    obs_perms, act_perms = env.twists()
    assert act_perms == want_a
    dim = 2 * n if kind == "clifford" else n
    for p, op in zip(want_q, obs_perms):
        full = list(p) + [n + x for x in p] if kind == "clifford" else list(p)  # symmetry.rs:276-295 / 265-274
        assert op == [full[r] * dim + full[c] for r in range(dim) for c in range(dim)]
    assert OracleEnv(kind, n, gs, add_perms=0).twists() == ([], [])  # clifford.rs:218-222


def test_expected_group_sizes():
    sizes = {g: len(qubit_perms(GRAPHS[g][0], gateset_from_coupling_map(GRAPHS[g][1], None, ALLOWED["clifford"])[1])[0]) for g in GRAPHS}
    # dihedral group of the hexagon 12, of the square grid 8, S4 on the star's leaves 24, the H shape 8 (swap each end pair, flip),
    # path x edge x isolated qubit 2 * 2; the one-way gatesets keep the identity only (a reflection reverses some CX)
    assert sizes == {"line5": 2, "line4_one_way": 1, "ring6": 12, "grid3x3": 8, "grid2x3": 1, "star5": 24, "heavy_hex7": 8, "two_components": 4}


def test_one_way_coupling_keeps_only_direction_preserving_automorphisms():
    # the undirected line-4 has the reversal, but CX(0,1) -> CX(3,2) is not in a one-way gateset (build_action_perm -> None)
    n, edges = GRAPHS["line4_one_way"]
    gs = gateset_from_coupling_map(edges, ["CX"], ALLOWED["linear_function"])[1]
    assert qubit_perms(n, gs)[0] == [[0, 1, 2, 3]]
    # ... while SWAP's key is order-free (symmetry.rs:66-71): a SWAP-only one-way gateset keeps the reversal
    gs = gateset_from_coupling_map(edges, ["SWAP"], ALLOWED["permutation"])[1]
    q, a = qubit_perms(n, gs)
    assert q == [[0, 1, 2, 3], [3, 2, 1, 0]] and a == [[0, 1, 2], [2, 1, 0]]


def test_asymmetric_one_qubit_gates_break_symmetries(golden_dir):
    gs_all = json.load(open(os.path.join(golden_dir, "gatesets.json")))
    gs = [(g[0], tuple(g[1])) for g in gs_all["model_clifford_3q_custom"]["env"]["gateset"]]
    q, a = qubit_perms(3, gs)
    assert q == [[0, 1, 2]]  # H / S exist on some qubits only in the reference's custom 3-qubit model
    assert len(a[0]) == len(gs)


def test_last_gate_wins_for_duplicate_keys():
    # bidirectional line-3, CX + SWAP: SWAP(0,1) and SWAP(1,0) share a key, the later index owns it (symmetry.rs:217-223)
    gs = gateset_from_coupling_map(line_edges(3, True), None, ALLOWED["linear_function"])[1]
    assert [g[0] for g in gs] == ["CX"] * 4 + ["SWAP"] * 4
    q, a = qubit_perms(3, gs)
    assert q == [[0, 1, 2], [2, 1, 0]]
    assert a[0] == [0, 1, 2, 3, 5, 5, 7, 7] and a[1] == [3, 2, 1, 0, 7, 7, 5, 5]


def test_no_two_qubit_gate_gives_every_permutation_in_heaps_order():
    gs = [("H", (q,)) for q in range(3)] + [("S", (q,)) for q in range(3)]
    q, a = qubit_perms(3, gs)
    # symmetry.rs:88-104 by hand: 012 -> swap(0,1) 102 -> swap(0,2) 201 -> swap(0,1) 021 -> swap(0,2) 120 -> swap(0,1) 210
    assert q == [[0, 1, 2], [1, 0, 2], [2, 0, 1], [0, 2, 1], [1, 2, 0], [2, 1, 0]]
    assert a == [p + [3 + x for x in p] for p in q]
    q4, _ = qubit_perms(4, [("H", (k,)) for k in range(4)])
    assert len(q4) == 24 and sorted(q4) == [list(p) for p in itertools.permutations(range(4))] and q4 != sorted(q4)
    assert q4[:4] == [[0, 1, 2, 3], [1, 0, 2, 3], [2, 0, 1, 3], [0, 2, 1, 3]]
    # a gateset whose only two-qubit gate acts on one qubit twice has no edge either (q1 != q2 guard, symmetry.rs:229)
    assert len(qubit_perms(3, [("H", (0,)), ("H", (1,)), ("H", (2,)), ("CX", (1, 1))])[0]) == 2  # perms fixing qubit 1


def test_pauli_env_twists_are_empty_but_its_perms_exist():
    n, edges = GRAPHS["ring6"]
    gs = gateset_from_coupling_map(edges, None, ALLOWED["pauli"])[1]
    env = OracleEnv("pauli", n, gs, add_perms=1)
    assert env.twists() == ([], [])  # pauli.rs:675-679
    assert len(qubit_perms(n, gs)[0]) == 12
