"""Host-side mirror of `qiskit_gym.envs` for the MI355X batched env.step() path."""
from .gateset import (  # noqa: F401
    ONE_Q_GATES,
    TWO_Q_GATES,
    gateset_from_coupling_map,
    grid_edges,
    line_edges,
    parse_gate,
    parse_gateset,
)
