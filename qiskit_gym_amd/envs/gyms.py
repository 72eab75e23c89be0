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


def pauli_solution_operations(solution: List[int], gateset, rotation_params=None) -> List[Tuple[str, Tuple[int, ...], object]]:
    """A PauliGym solution as circuit operations in order, without qiskit: `(name, qubits, parameter)`.
    Gates come from the gateset -- CX with its qubits REVERSED, because PauliNetwork::cnot uses the opposite convention to the gateset's
    (control, target) (reference envs/synthesis.py:489-497) -- and a released rotation `(axis, qubit, index, sign)` becomes
    `("rx" | "ry" | "rz", (qubit,), sign * rotation_params[index])`, or `(index, sign)` when no parameters are given (:498-504)."""
    ops = []
    for kind, a1, a2, a3 in decode_pauli_solution(solution):
        if kind == "gate":
            name, args = gateset[a1]
            args = tuple(int(q) for q in args)
            ops.append((name.lower(), args[::-1] if name.lower() == "cx" else args, None))
        elif rotation_params is None:
            ops.append((kind, (a1,), (a2, a3)))
        elif a2 < len(rotation_params):
            ops.append((kind, (a1,), a3 * rotation_params[a2]))
        else:
            raise ValueError("too few rotation parameters stored for this solution")
    return ops


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


def _is_qiskit_object(x) -> bool:
    """QuantumCircuit / Clifford (anything that is not a plain matrix, list or tuple) -- without importing qiskit."""
    return not isinstance(x, (np.ndarray, list, tuple)) and (hasattr(x, "adjoint") or hasattr(x, "num_qubits"))


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
        if _is_qiskit_object(input):
            from qiskit import QuantumCircuit
            from qiskit.quantum_info import Clifford

            if isinstance(input, QuantumCircuit):
                input = Clifford(input)
            return input.adjoint().tableau[:, :-1].T.flatten().astype(int).tolist()
        return _tableau_state(input)

    def post_process_synthesis(self, synth_circuit, input):
        """The synthesised gates fix the tableau, not the Pauli phases: add the X / Y / Z layer that does, and return the circuit of the
        input itself rather than of its inverse (reference envs/synthesis.py:161-176, 210-217; needs qiskit).  Matrix inputs carry no phases:
        the circuit is returned as built."""
        if not _is_qiskit_object(input):
            return synth_circuit
        from qiskit import QuantumCircuit
        from qiskit.quantum_info import Clifford

        inverse = synth_circuit.inverse()
        target = Clifford(input) if isinstance(input, QuantumCircuit) else input
        rest = Clifford(inverse).compose(target)
        paulis = QuantumCircuit(rest.num_qubits)
        for q in range(rest.num_qubits):
            stab, destab = rest.stab_phase[q], rest.destab_phase[q]
            if destab and stab:
                paulis.y(q)
            elif stab:
                paulis.x(q)
            elif destab:
                paulis.z(q)
        return paulis.compose(inverse).inverse()


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
        if _is_qiskit_object(input) or hasattr(input, "linear"):  # reference envs/synthesis.py:254-258
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
        object.__setattr__(self, "_original_circuit", None)

    @staticmethod
    def _parse_circuit(circuit):
        """QuantumCircuit -> (Clifford, rotation labels, rotation angles) (reference envs/synthesis.py:316-364; needs qiskit): Clifford gates
        compose onto a running Clifford, an rx / ry / rz becomes the single-qubit Pauli evolved through it (adjoint label) plus its angle."""
        from qiskit.quantum_info import Clifford, Pauli

        n = circuit.num_qubits
        clifford = Clifford(np.eye(2 * n, dtype=bool))
        labels, params = [], []
        for inst in circuit.data:
            name = inst.operation.name.lower()
            qubits = [circuit.find_bit(q).index for q in inst.qubits]
            if name in ("rx", "ry", "rz"):
                chars = ["I"] * n
                chars[n - 1 - qubits[0]] = name[1].upper()
                labels.append(Pauli("".join(chars)).evolve(clifford).adjoint().to_label())
                params.extend(inst.operation.params)
            else:
                clifford = clifford.compose(inst.operation, qubits)
        return clifford, labels, params

    def get_state(self, input, rotations: List[str] = None):
        """`(tableau, rotations)` (or, with qiskit, a Clifford / a QuantumCircuit with rotations) -> the set_state wire format
        [rot_count, tableau..., len, chars, ...] (reference envs/synthesis.py:413-464)."""
        object.__setattr__(self, "_rotation_params", [])
        object.__setattr__(self, "_original_circuit", None)
        if isinstance(input, tuple):
            tableau, rotations = input  # a Clifford given in a tuple is taken as it is (already the adjoint)
        elif hasattr(input, "data") and hasattr(input, "num_qubits") and not hasattr(input, "tableau"):  # QuantumCircuit
            clifford, rotations, params = self._parse_circuit(input)
            object.__setattr__(self, "_rotation_params", list(params))
            object.__setattr__(self, "_original_circuit", input)
            tableau = clifford.adjoint()
        elif hasattr(input, "adjoint") and hasattr(input, "tableau"):  # a raw qiskit Clifford: the state holds its adjoint
            tableau = input.adjoint()
        else:
            tableau = input
        if hasattr(tableau, "tableau"):
            tableau = tableau.tableau[:, :-1].T
        rotations = list(rotations or [])
        state = [len(rotations)] + _tableau_state(tableau)
        for rot in rotations:
            state.append(len(rot))
            state.extend(ord(c) for c in rot)
        return state

    def solution_operations(self, actions: List[int]):
        """The decoded solution as `(name, qubits, parameter)` operations (no qiskit needed): `pauli_solution_operations`."""
        return pauli_solution_operations(actions, self.config["gateset"], self._rotation_params or None)

    def build_circuit_from_solution(self, actions: List[int], input=None):
        """Gates + released rotations as a QuantumCircuit, followed by the Clifford that corrects the Pauli phases when the original
        circuit is known (reference envs/synthesis.py:466-518; needs qiskit)."""
        from qiskit import QuantumCircuit
        from qiskit.quantum_info import Clifford

        qc = QuantumCircuit(self.config["num_qubits"])
        for name, qubits, param in pauli_solution_operations(actions, self.config["gateset"], self._rotation_params):
            if param is None:
                getattr(qc, name)(*qubits)
            else:
                getattr(qc, name)(param, qubits[0])
        original = input if (hasattr(input, "data") and hasattr(input, "num_qubits") and not hasattr(input, "tableau")) else self._original_circuit
        if original is not None:
            rest = qc.inverse().compose(original)
            only_clifford = QuantumCircuit.copy_empty_like(rest)
            for g in rest:
                if g.operation.name not in ("rx", "ry", "rz"):
                    only_clifford.append(g.operation, g.qubits)
            qc = qc.compose(Clifford(only_clifford).to_circuit())
        return qc


SYNTH_ENVS = {
    "CliffordEnv": CliffordGym,
    "LinearFunctionEnv": LinearFunctionGym,
    "PermutationEnv": PermutationGym,
    "PauliNetworkEnv": PauliGym,
}
