"""CliffordGym / LinearFunctionGym / PermutationGym / PauliGym -- the reference's public env
classes (src/qiskit_gym/envs/synthesis.py:67-528) over the MI355X path.

Constructor signatures, defaults, `from_coupling_map`, `from_json`, `to_json`, the gateset ordering
and the solution encoding are the reference's.  The Qiskit-facing helpers (`get_state` from a
`QuantumCircuit`/`Clifford`, circuit reconstruction) need qiskit, which is an optional import:
plain matrices / permutations / `(tableau, labels)` inputs work without it.

Each class also has `.vec(batch, ...)`, which builds the batched `qiskit_gym_amd.VecEnv` with the
same configuration -- the path that actually uses the GPU.
"""
from __future__ import annotations

import inspect
from typing import Iterable, List, Tuple

import numpy as np

from .frontend import GymFrontEnd
from .gateset import ONE_Q_GATES, TWO_Q_GATES, gateset_from_coupling_map
from .raw import RawEnv

ROTATION_MARKER = 0x80000000  # must match the step kernel's solution encoding (reference pauli.rs:698)


def decode_pauli_solution(encoded_solution: List[int]) -> List[Tuple[str, int, int, int]]:
    """Inverse of the solution encoding (reference envs/synthesis.py:35-61, pauli.rs:685-719)."""
    result = []
    axis_names = ["rx", "ry", "rz"]
    for val in encoded_solution:
        if val >= ROTATION_MARKER:
            result.append((axis_names[(val >> 21) & 0x3], (val >> 11) & 0x3FF, (val >> 1) & 0x3FF, 1 if (val & 1) else -1))
        else:
            result.append(("gate", val, 0, 0))
    return result


class BaseSynthesisEnv:
    cls_name: str
    allowed_gates: List[str]
    env_kind: str

    @classmethod
    def from_coupling_map(cls, coupling_map, basis_gates=None, difficulty: int = 1, depth_slope: int = 2, max_depth: int = 128,
                          metrics_weights=None, add_inverts: bool = True, add_perms: bool = True):
        num_qubits, gateset = gateset_from_coupling_map(coupling_map, basis_gates, cls.allowed_gates)
        config = {
            "num_qubits": num_qubits, "difficulty": difficulty, "gateset": gateset, "depth_slope": depth_slope,
            "max_depth": max_depth, "metrics_weights": metrics_weights, "add_inverts": add_inverts, "add_perms": add_perms,
        }
        valid = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in config.items() if k in valid})

    @classmethod
    def from_json(cls, env_config):
        valid = set(inspect.signature(cls.__init__).parameters) - {"self"}
        return cls(**{k: v for k, v in env_config.items() if k in valid})

    def vec(self, batch: int, device=None, **overrides):
        """The batched, GPU-resident counterpart of this env (same configuration)."""
        from ..vec import VecEnv

        cfg = {k: v for k, v in self.config.items() if k not in ("num_qubits", "gateset")}
        cfg.update(overrides)
        return VecEnv(self.env_kind, self.config["num_qubits"], self.config["gateset"], batch, device=device, **cfg)

    def vec_gym(self, num_envs: int, seed: int = 0, device=None, **overrides):
        """`gymnasium.vector`-shaped front end over `vec(num_envs)` with same-step autoreset (envs/vector.py)."""
        from .vector import VecGym

        return VecGym(self.vec(num_envs, device=device, **overrides), seed=seed)

    def build_circuit_from_solution(self, actions: List[int], input=None):
        from qiskit import QuantumCircuit  # optional dependency

        qc = QuantumCircuit(self.config["num_qubits"])
        for a in actions:
            name, args = self.config["gateset"][a]
            getattr(qc, name.lower())(*args)
        return self.post_process_synthesis(qc, input)

    def post_process_synthesis(self, synth_circuit, _input_state):
        return synth_circuit


def _tableau_state(x) -> List[int]:
    return np.asarray(x).astype(int).flatten().tolist()


class CliffordGym(GymFrontEnd, BaseSynthesisEnv):
    cls_name = "CliffordEnv"
    env_kind = "clifford"
    allowed_gates = ONE_Q_GATES + TWO_Q_GATES

    def __init__(self, num_qubits: int, gateset, difficulty: int = 1, depth_slope: int = 2, max_depth: int = 128,
                 metrics_weights=None, add_inverts: bool = True, add_perms: bool = True, track_solution: bool = True):
        super().__init__(num_qubits=num_qubits, difficulty=difficulty, gateset=gateset, depth_slope=depth_slope,
                         max_depth=max_depth, metrics_weights=metrics_weights, add_inverts=add_inverts, add_perms=add_perms,
                         track_solution=track_solution)

    def get_state(self, input):
        """QuantumCircuit / qiskit Clifford -> adjoint tableau without the phase column, transposed
        (reference envs/synthesis.py:206-209); a 2N x 2N 0/1 matrix is passed through."""
        if hasattr(input, "adjoint") or hasattr(input, "data"):
            from qiskit import QuantumCircuit
            from qiskit.quantum_info import Clifford

            if isinstance(input, QuantumCircuit):
                input = Clifford(input)
            return input.adjoint().tableau[:, :-1].T.flatten().astype(int).tolist()
        return _tableau_state(input)


class LinearFunctionGym(GymFrontEnd, BaseSynthesisEnv):
    cls_name = "LinearFunctionEnv"
    env_kind = "linear_function"
    allowed_gates = ["CX", "SWAP"]

    def __init__(self, num_qubits: int, gateset, difficulty: int = 1, depth_slope: int = 2, max_depth: int = 128,
                 metrics_weights=None, add_inverts: bool = True, add_perms: bool = True, track_solution: bool = True):
        super().__init__(num_qubits=num_qubits, difficulty=difficulty, gateset=gateset, depth_slope=depth_slope,
                         max_depth=max_depth, metrics_weights=metrics_weights, add_inverts=add_inverts, add_perms=add_perms,
                         track_solution=track_solution)

    def get_state(self, input):
        if hasattr(input, "data") or hasattr(input, "linear"):  # reference envs/synthesis.py:254-258
            from qiskit.circuit.library.generalized_gates import LinearFunction
            from qiskit.quantum_info import Clifford

            input = LinearFunction(Clifford(input).adjoint())
            return np.array(input.linear).flatten().astype(int).tolist()
        return _tableau_state(input)


class PermutationGym(GymFrontEnd, BaseSynthesisEnv):
    cls_name = "PermutationEnv"
    env_kind = "permutation"
    allowed_gates = ["SWAP"]

    def __init__(self, num_qubits: int, gateset, difficulty: int = 1, depth_slope: int = 2, max_depth: int = 128,
                 metrics_weights=None, add_inverts: bool = True, add_perms: bool = True, track_solution: bool = True):
        super().__init__(num_qubits=num_qubits, difficulty=difficulty, gateset=gateset, depth_slope=depth_slope,
                         max_depth=max_depth, metrics_weights=metrics_weights, add_inverts=add_inverts, add_perms=add_perms,
                         track_solution=track_solution)

    def get_state(self, input: Iterable[int]):
        if hasattr(input, "data") and not isinstance(input, np.ndarray):  # QuantumCircuit (reference envs/synthesis.py:295-303)
            from qiskit.circuit.library.generalized_gates import LinearFunction

            input = LinearFunction(input).permutation_pattern()
        elif hasattr(input, "pattern"):
            input = input.pattern
        return np.argsort(np.array(input)).astype(int).tolist()


class PauliGym(GymFrontEnd, BaseSynthesisEnv):
    cls_name = "PauliNetworkEnv"
    env_kind = "pauli"
    allowed_gates = ONE_Q_GATES + TWO_Q_GATES

    def __init__(self, num_qubits: int, gateset, difficulty: int = 1, depth_slope: int = 2, max_depth: int = 128,
                 max_rotations: int = 5, pauli_diff_scale: int = 16, num_qubits_decay: float = 0.5,
                 final_pauli_layers=None, metrics_weights=None, add_perms: bool = True, pauli_layer_reward: float = 0.01,
                 track_solution: bool = True):
        super().__init__(num_qubits=num_qubits, difficulty=difficulty, gateset=gateset, depth_slope=depth_slope,
                         max_depth=max_depth, max_rotations=max_rotations, pauli_diff_scale=pauli_diff_scale,
                         num_qubits_decay=num_qubits_decay, final_pauli_layers=final_pauli_layers,
                         metrics_weights=metrics_weights, add_perms=add_perms, pauli_layer_reward=pauli_layer_reward,
                         track_solution=track_solution)
        object.__setattr__(self, "_rotation_params", [])

    def get_state(self, input, rotations: List[str] = None):
        """`(tableau, rotations)` -> the set_state wire format
        [rot_count, tableau..., len, chars, ...] (reference envs/synthesis.py:451-461)."""
        if isinstance(input, tuple):
            tableau, rotations = input
        else:
            tableau = input
        if hasattr(tableau, "tableau"):  # qiskit Clifford, already in adjoint form for tuple input
            tableau = tableau.tableau[:, :-1].T
        rotations = list(rotations or [])
        state = [len(rotations)] + _tableau_state(tableau)
        for rot in rotations:
            state.append(len(rot))
            state.extend(ord(c) for c in rot)
        return state


SYNTH_ENVS = {
    "CliffordEnv": CliffordGym,
    "LinearFunctionEnv": LinearFunctionGym,
    "PermutationEnv": PermutationGym,
    "PauliNetworkEnv": PauliGym,
}
