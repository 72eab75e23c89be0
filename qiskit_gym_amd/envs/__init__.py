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


def __getattr__(name):  # lazy: the env classes need the built library only when used
    if name in ("CliffordGym", "LinearFunctionGym", "PermutationGym", "PauliGym", "SYNTH_ENVS", "decode_pauli_solution"):
        from . import gyms as synthesis

        return getattr(synthesis, name)
    if name == "VecGym":
        from .vector import VecGym

        return VecGym
    if name == "RawEnv":
        from .raw import RawEnv

        return RawEnv
    raise AttributeError(name)
