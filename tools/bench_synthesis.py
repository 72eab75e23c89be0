"""Time BatchedSynthesis with the reference's trained policies (tests/golden/policies): M targets x S searches as one batch.
Run on the GPU box: python tools/bench_synthesis.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from test_reference_policies import MODELS, load  # noqa: E402

import qiskit_gym_amd.envs as envs  # noqa: E402
from qiskit_gym_amd.synthesis import BatchedSynthesis, policy_from_reference_state_dict  # noqa: E402

GYMS = {"clifford": "CliffordGym", "linear_function": "LinearFunctionGym", "permutation": "PermutationGym"}

for name, M, S in (("clifford_3q_custom", 1024, 64), ("lf_5_line", 1024, 64), ("perm_square_3x3", 1024, 64), ("clifford_3q_custom", 64, 1024)):
    cfg, gateset, w = load(name)
    kind = MODELS[name]
    gym = getattr(envs, GYMS[kind])(cfg["num_qubits"], gateset, depth_slope=cfg["depth_slope"], max_depth=cfg["max_depth"])
    syn = BatchedSynthesis(gym, policy_from_reference_state_dict(w), seed=1)
    # targets: random scrambles made on the device, read back in the set_state wire format
    v = gym.vec(M, add_inverts=False, add_perms=False, track_solution=False, difficulty=32)
    v.reset(3)
    states = v.get_state("i64").cpu().numpy()
    syn.solve(states[:8], num_searches=S)  # warm-up (library handles)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sols = syn.solve(states, num_searches=S)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = syn.last_stats
    print(f"{name}{' (policy-layer kernels)' if st.get('kernels') else ''}: {M} targets x {S} searches: {dt * 1e3:.1f} ms ({dt / M * 1e6:.1f} us per target), solved {st['solved']}/{M}, "
          f"{st['searches_solved']:.1%} of searches, mean gates {st['mean_gates']:.1f}, {st['steps']} steps", flush=True)
