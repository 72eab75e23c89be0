"""Why the multi-GPU gather is handed to its side stream through the host (OverlappedGather).

One rank, CliffordGym 16q x 65536, hipGraph replays of 256 steps with something between replays:
  none                  nothing                                           3.14 us per step
  record + stream wait  event recorded on the step stream, a second stream waits on it: every later replay
                        on the step stream runs ~40 % slower              4.3-4.5 us per step
  host-mediated         OverlappedGather: host waits for the event, then enqueues the all-gather on the side
                        stream; the gather overlaps the next replay       3.2 us per step
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist

for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29544"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from qiskit_gym_amd.distributed import OverlappedGather
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

gs = line_gateset("clifford", 16); B, G, SEG = 65536, 256, 16
env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
stream, comm = torch.cuda.Stream(), torch.cuda.Stream()
acts = torch.randint(0, len(gs), (16, B), dtype=torch.int32, device="cuda")
snap = torch.empty((B, 32), dtype=torch.int32, device="cuda")
ev = torch.cuda.Event()
og = OverlappedGather((B, 32), torch.int32, "cuda")
for mode in ("none", "record + stream wait", "host-mediated"):
    with torch.cuda.stream(stream):
        env.reset(1); env.rollout_ring(acts, G); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for s in range(SEG):
            env.rollout_ring(acts, G)
            if mode == "record + stream wait":
                env.observe_packed(out=snap); ev.record(stream); comm.wait_event(ev)
            if mode == "host-mediated":
                og.submit(lambda buf: env.observe_packed(out=buf))
        og.flush()
        e1.record(stream); torch.cuda.synchronize()
    print(f"{mode:22s}: {e0.elapsed_time(e1) * 1e3 / (SEG * G):.2f} us per step")
dist.destroy_process_group()
