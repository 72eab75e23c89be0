"""qg_vec_reset_done on a short list (512 finished envs of 65 536: scramble_tree) against the number of scramble gates: slope = the chain's cost
per gate, intercept = everything else (launch, list length, draws' latency, products, finish).  Kernel time from HIP events around the launch
pair (compact_done + reset) minus the same with nothing finished.  Development timing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

gs = line_gateset("clifford", 16); B = 65536
for count in (512, 64):
    for diff in (64, 128, 256, 512, 1024):
        env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=diff, max_depth=128)
        env.reset(1)
        mask = torch.zeros(B, dtype=torch.uint8, device="cuda"); mask[:: B // count] = 1
        res = {}
        for name, m in (("some", mask), ("none", torch.zeros_like(mask))):
            ts = []
            for i in range(20):
                env.done.copy_(m)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); env.reset_done(100 + i); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); res[name] = ts[len(ts) // 2]
        print(f"{count} finished envs, {diff:5d} gates: reset_done {res['some']:6.1f} us, with nothing to reset {res['none']:6.1f} us, difference {res['some'] - res['none']:6.1f} us", flush=True)
        env.close()
