"""End-to-end collection rate with a policy in the loop (CliffordGym 16q, BasicPolicy-shaped MLP,
bf16), everything on one MI355X.  Context for the reference's notebook timings (collect phase of
1 024 short episodes in ~15 ms = ~1.4e5 env-steps/s including its CPU policy inference)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

ONLY = os.environ.get("ONLY")  # e.g. ONLY=65536 to run the large batch only (profiling)
for B, store, graph in ((1024, "packed", False), (1024, "packed", True), (2048, "packed", True), (4096, "packed", True), (8192, "packed", False), (8192, "packed", True), (16384, "packed", True), (32768, "packed", True), (65536, "dense", False), (65536, "packed", False), (65536, "packed", True)):
    if ONLY and (B != int(ONLY) or store != "packed" or graph):
        continue
    if os.environ.get("GRAPH") == "1" and not graph:  # GRAPH=1: the hipGraph configurations only (small-batch A/Bs)
        continue
    NQ = int(os.environ.get("QUBITS", "16"))  # > 16: 64-bit row words, the first layer reads the packed observation (qg_policy_embed_words)
    gs = line_gateset("clifford", NQ)
    dflt = os.environ.get("DEFAULTS") == "1"  # DEFAULTS=1: the reference's defaults add_inverts=True, track_solution=True
    env = VecEnv("clifford", NQ, gs, B, add_inverts=dflt, add_perms=False, track_solution=dflt, difficulty=int(os.environ.get("DIFF", "32")))
    fused = {"1": True, "0": False}.get(os.environ.get("FUSED", ""), None)  # FUSED=1 / 0 forces the policy-layer kernels on / off
    embed = {"1": True, "0": False}.get(os.environ.get("EMBED", ""), fused)  # EMBED=0: library GEMM for the first layer only
    fstep = {"1": True, "0": False}.get(os.environ.get("FSTEP", ""), None)  # FSTEP=0: sampling kernel, then qg_vec_step (A/B of the fused launch)
    col = RolloutCollector(env, BasicPolicy(4 * NQ * NQ, len(gs)), dtype=torch.bfloat16, seed=1, store_obs=store, use_graph=graph,
                           use_bit_embedding=embed, use_fused_head=fused, use_fused_step=fstep)
    T = 32
    ro = col.collect(T)
    ro = col.collect(T, out=ro)  # with use_graph: the first replay (its one-time upload costs tens of ms now and then)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ro = col.collect(T, out=ro)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.sync()
    done_rate = float(ro.dones.float().mean())
    del col, env, ro  # the next configuration starts from an empty allocator (graph pools of earlier ones otherwise stall its first collections)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    print(f"CliffordGym {NQ}q B={B} obs stored {store}{' hipGraph' if graph else ''}{' FSTEP=' + os.environ['FSTEP'] if 'FSTEP' in os.environ else ''}: {3 * T * B / dt:.3e} env-steps/s with the policy in the loop ({dt / (3 * T) * 1e6:.0f} us per step), "
          f"success rate in last rollout {done_rate:.3f} done/step")
