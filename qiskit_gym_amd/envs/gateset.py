"""Gate names, gate parsing and the `from_coupling_map` gateset builder.

Host-side mirror of two reference contracts:

* the gate argument format and its validation -- `(name, [indices])`, name trimmed and
  case-insensitive, arity checked (reference: rust/src/envs/common.rs:46-100);
* the gateset ordering produced by `BaseSynthesisEnv.from_coupling_map`
  (reference: src/qiskit_gym/envs/synthesis.py:83-103): edges sorted, `num_qubits = max + 1`,
  then for every basis gate in order: 1-qubit gates over all qubits, 2-qubit gates over the
  sorted edges.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

ONE_Q_GATES = ["H", "S", "Sdg", "SX", "SXdg"]  # envs/synthesis.py:28
TWO_Q_GATES = ["CX", "CZ", "SWAP"]  # envs/synthesis.py:29

# numeric gate kinds shared with include/qgym.h (enum order of common.rs:19-29)
GATE_KIND = {"h": 0, "s": 1, "sdg": 2, "sx": 3, "sxdg": 4, "cx": 5, "cz": 6, "swap": 7}
KIND_NAME = ["H", "S", "Sdg", "SX", "SXdg", "CX", "CZ", "SWAP"]

Gate = Tuple[str, Tuple[int, ...]]


def parse_gate(gate) -> Tuple[int, int, int]:
    """Validate one `(name, indices)` pair; returns `(kind, q0, q1)`.

    Error types and messages follow common.rs:49-98 (PyTypeError -> TypeError,
    PyValueError -> ValueError).
    """
    try:
        n_items = len(gate)
    except TypeError:
        raise TypeError("Each gate must be a 2-item sequence: (name, indices)")
    if isinstance(gate, (str, bytes)):
        raise TypeError("Each gate must be a 2-item sequence: (name, indices)")
    if n_items != 2:
        raise ValueError("Each gate must have exactly 2 items: (name, indices)")
    name, idx = gate[0], gate[1]
    if not isinstance(name, str):
        raise TypeError("Gate name must be a string")
    name = name.strip()
    key = name.lower()
    if isinstance(idx, (str, bytes)) or not hasattr(idx, "__len__"):
        raise TypeError("Gate indices must be a list/tuple of integers")
    qs = []
    for q in idx:
        if isinstance(q, bool) or not isinstance(q, int) and not hasattr(q, "__index__"):
            raise TypeError("Gate indices must be non-negative integers (usize)")
        q = int(q)
        if q < 0:
            raise TypeError("Gate indices must be non-negative integers (usize)")
        qs.append(q)
    one_q = key in ("h", "s", "sdg", "sx", "sxdg")
    two_q = key in ("cx", "cz", "swap")
    if one_q and len(qs) == 1:
        return GATE_KIND[key], qs[0], 0
    if two_q and len(qs) == 2:
        return GATE_KIND[key], qs[0], qs[1]
    if one_q:
        raise ValueError(f"Gate `{name}` expects 1 index, got {len(qs)}")
    if two_q:
        raise ValueError(f"Gate `{name}` expects 2 indices, got {len(qs)}")
    raise ValueError(f"Unknown gate name `{name}`. Allowed: H, S, Sdg, SX, SXdg, CX, CZ, SWAP")


def parse_gateset(gateset: Iterable) -> List[Tuple[int, int, int]]:
    return [parse_gate(g) for g in gateset]


def coupling_edges(coupling_map) -> List[Tuple[int, int]]:
    """Edge list of a coupling map: a qiskit `CouplingMap` (if qiskit is importable) or any
    iterable of `(q1, q2)` pairs (envs/synthesis.py:89-91)."""
    if hasattr(coupling_map, "get_edges"):
        coupling_map = list(coupling_map.get_edges())
    return sorted((int(a), int(b)) for a, b in coupling_map)


def gateset_from_coupling_map(
    coupling_map, basis_gates: Sequence[str] | None, allowed_gates: Sequence[str]
) -> Tuple[int, List[Gate]]:
    """Returns `(num_qubits, gateset)` exactly as envs/synthesis.py:83-103 builds them."""
    if basis_gates is None:
        basis_gates = tuple(allowed_gates)
    assert all(g in allowed_gates for g in basis_gates), (
        f"Some provided gates are not allowed (allowed: {list(allowed_gates)})."
    )
    edges = coupling_edges(coupling_map)
    num_qubits = max(max(q) for q in edges) + 1
    gateset: List[Gate] = []
    for gate_name in basis_gates:
        if gate_name in ONE_Q_GATES:
            for q in range(num_qubits):
                gateset.append((gate_name, (q,)))
        else:
            assert gate_name in TWO_Q_GATES, f"Gate {gate_name} not supported!"
            for q1, q2 in edges:
                gateset.append((gate_name, (q1, q2)))
    return num_qubits, gateset


def line_edges(n: int, bidirectional: bool = True) -> List[Tuple[int, int]]:
    """`CouplingMap.from_line(n, bidirectional)` edge list."""
    e = []
    for i in range(n - 1):
        e.append((i, i + 1))
        if bidirectional:
            e.append((i + 1, i))
    return e


def grid_edges(rows: int, cols: int, bidirectional: bool = True) -> List[Tuple[int, int]]:
    """`CouplingMap.from_grid(rows, cols, bidirectional)` edge set (row-major qubit ids)."""
    e = []
    for r in range(rows):
        for c in range(cols):
            q = r * cols + c
            if c + 1 < cols:
                e.append((q, q + 1))
                if bidirectional:
                    e.append((q + 1, q))
            if r + 1 < rows:
                e.append((q, q + cols))
                if bidirectional:
                    e.append((q + cols, q))
    return e
