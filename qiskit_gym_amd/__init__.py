"""qiskit_gym_amd -- MI355X-native batched `env.step()` for qiskit-gym's synthesis environments.

Only the hot path lives here: hand-written gfx950 HIP kernels behind a C ABI
(`include/qgym.h`, `qiskit_gym_amd/csrc/`), and the host-side mirror of the reference's env API.
There is no CPU fallback: creating an environment without the HIP library or without a GPU raises.
"""
__version__ = "0.1.0"
