"""A/B: one-step gather/scatter kernel (default) vs the register-resident TILE kernel (QGYM_TILE_DENSE=1)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    from qiskit_gym_amd.vec import VecEnv
    from util import line_gateset
    for kind, n in (("clifford", 16), ("clifford", 20), ("clifford", 32), ("linear_function", 32), ("linear_function", 48), ("linear_function", 64)):
        gs = line_gateset(kind, n)
        for B in (65536, 262144):
            env = VecEnv(kind, n, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
            env.reset(1)
            acts = torch.randint(0, len(gs), (16, B), dtype=torch.int32, device="cuda")
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                env.rollout_ring(acts, 256); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
                for _ in range(8): env.rollout_ring(acts, 256)
                e1.record(s)
            torch.cuda.synchronize()
            print(f"{sys.argv[1]:7s} {kind}{n:<3d} B={B:8d} {e0.elapsed_time(e1) * 1e3 / (8 * 256):7.2f} us/step")
            env.close()
else:
    for mode in ("dense", "sparse"):
        envv = dict(os.environ)
        if mode == "dense": envv["QGYM_TILE_DENSE"] = "1"
        subprocess.run([sys.executable, __file__, mode], env=envv)
