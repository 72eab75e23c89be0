"""The desynchronised auto-reset leg of bench.py on its own (CliffordGym 16q x 65 536, difficulty 256, 1/128 of the batch finishing per step),
for rocprofv3 --kernel-trace --stats: what the step kernel with its done list and the reset kernel cost inside the captured graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if os.environ.get('QG_LIB'):
    from qiskit_gym_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ['QG_LIB'])
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

NQ = int(os.environ.get("N", "16"))  # > 16: 64-bit rows (no one-launch pair: two launches)
gs = line_gateset("clifford", NQ); B, AT, seed = int(os.environ.get("B", "65536")), 128, 7
A = len(gs)
DEFAULTS = os.environ.get("DEFAULTS") == "1"  # the reference's default options: add_inverts + solution log (the pair is then two launches)
env = VecEnv("clifford", NQ, gs, B, add_inverts=DEFAULTS, add_perms=False, track_solution=DEFAULTS, difficulty=256)
acts = torch.randint(0, A, (AT, B), dtype=torch.int32, device="cuda")
stream = torch.cuda.Stream()
with torch.cuda.stream(stream):
    env.reset(seed)
    cls = torch.arange(B, device="cuda") % AT
    for k in range(AT):
        env.set_counters(k, k); env.step(acts[k]); env.reset_done(seed + 0x51ED * (k + 1))
        env.done[cls == k] = 1; env.reset_done(seed + 0xA5A5 * (k + 1))
    FUSED = os.environ.get("UNFUSED") != "1"
    def episode():
        if not FUSED:
            for t in range(AT):
                env.set_counters(t, t); env.rollout(acts[t:t + 1]); env.reset_done(seed + 0x9E3779B9 * (t + 1))
            return
        env.set_counters(0, 0); env.rollout(acts[0:1])
        for t in range(1, AT):
            env.set_counters(t, t); env.reset_done_step(seed + 0x9E3779B9 * t, acts[t])  # reset_done + step: one launch
        env.reset_done(seed + 0x9E3779B9 * AT)
    episode(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        episode()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(8): g.replay()
    e1.record(stream)
torch.cuda.synchronize(); env.sync()
print(("one launch per (reset_done + step)" if FUSED and not DEFAULTS and NQ <= 16 else "two launches") + (" [reference-default options]" if DEFAULTS else "") + (f" [{NQ} qubits]" if NQ != 16 else "") + ": ", end="")
print(f"desynchronised auto-reset: {e0.elapsed_time(e1) * 1e3 / (8 * AT):.2f} us per (step + reset_done), {100.0 / AT:.2f} % of the batch finishes per step")
