"""Where an auto-reset pair (qg_vec_reset_done + the next qg_vec_step) spends its time, on the kernel device clock (qg_vec_set_kernel_clock):
the pair's launches in one captured graph of AT pairs, episode ends spread evenly over time (1 / AT of the batch finishes per step), every stamped
launch's own duration beside the graph's period per pair.

  python tools/auto_reset_breakdown.py [--out profiles/r05/auto_reset_breakdown.txt]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

if os.environ.get("QG_LIB"):  # development: a variant build
    from qiskit_gym_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["QG_LIB"])
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset

AT = 128


ONLY = None


def run(name, env, A, slots_per_pair, labels, fused, out, reset_slots=1):
    if ONLY and ONLY not in name:
        return
    dev = env.device
    B = env.batch
    stream = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    acts = torch.randint(0, A, (AT, B), dtype=torch.int32, device=dev, generator=gen)
    fin = torch.empty((AT, B), dtype=torch.uint8, device=dev)
    seed = 0x5EED0003
    with torch.cuda.stream(stream):
        env.reset(seed)
        cls = torch.arange(B, device=dev) % AT
        for k in range(AT):
            env.set_counters(k, k)
            env.step(acts[k])
            env.reset_done(seed + 0x51ED * (k + 1))
            env.done[cls == k] = 1
            env.reset_done(seed + 0xA5A5 * (k + 1))

        def episode():
            env.set_counters(0, 0)
            env.rollout(acts[0:1], dones_out=fin[0:1])
            for t in range(1, AT):
                env.set_counters(t, t)
                if fused:
                    env.reset_done_step(seed + 0x9E3779B9 * t, acts[t], dones_out=fin[t])
                else:
                    env.reset_done(seed + 0x9E3779B9 * t)
                    env.rollout(acts[t:t + 1], dones_out=fin[t:t + 1])
            env.reset_done(seed + 0x9E3779B9 * AT)

        def graph(n_slots=0):
            """Eager pass, then the capture; n_slots > 0: the captured launches (and only they) carry kernel-clock slots 0, 1, ..."""
            episode()
            torch.cuda.synchronize()
            slots = env.kernel_clock(n_slots) if n_slots else None
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                episode()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            return (g, slots) if n_slots else g

        def period(g):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(4):
                g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / (4 * AT)

        g = graph()
        plain = min(period(g) for _ in range(3))
        del g
        n_slots = 1 + slots_per_pair * (AT - 1) + reset_slots  # the first step, AT - 1 pairs, the last reset_done
        g, view = graph(n_slots + 8)
        stamped = min(period(g) for _ in range(3))
        per = [[] for _ in range(slots_per_pair)]
        for _ in range(4):
            view.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            t0, t1 = view[..., 0], view[..., 1]
            live = t1 != 0
            first = torch.where(live, t0, torch.full_like(t0, 2**62)).amin(dim=1)
            dur = ((t1.amax(dim=1) - first).double() / 100.0).cpu().numpy()  # 100 MHz -> us
            ok = live.any(dim=1).cpu().numpy()
            assert int(ok.sum()) == n_slots and ok[:n_slots].all(), (name, int(ok.sum()), n_slots)  # the capture's launches are what this tool thinks they are
            for j in range(slots_per_pair):
                idx = 1 + j + slots_per_pair * np.arange(3, AT - 1)  # pair t's j-th launch (the first pairs of a capture take other paths: skipped)
                per[j].append(dur[idx][ok[idx]])
        env.kernel_clock(0)
    env.sync()
    frac = fin.float().mean(dim=1)
    line = f"{name}: {plain:6.2f} us per pair (graph period; {stamped:.2f} with stamps), finished per step {float(frac.mean()):.4f}"
    parts = []
    for j in range(slots_per_pair):
        d = np.concatenate(per[j]) if per[j] and len(per[j][0]) else np.array([])
        parts.append(f"{labels[j]} {d.mean():.2f} us (median {np.median(d):.2f}, max {d.max():.2f})" if d.size else f"{labels[j]} (no stamps)")
    ksum = sum(np.concatenate(per[j]).mean() for j in range(slots_per_pair) if per[j] and len(per[j][0]))
    line += "\n      kernels on the device clock: " + "; ".join(parts) + f"; sum {ksum:.2f} us -> launch boundaries and unstamped launches {plain - ksum:.2f} us"
    print(line, flush=True)
    out.append(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="run the rows whose name holds this text")
    args = ap.parse_args()
    global ONLY
    ONLY = args.only
    torch.cuda.set_device(0)
    B = 65536
    gs3 = line_gateset("clifford", 16)
    out = []
    env = VecEnv("clifford", 16, gs3, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
    run("CliffordGym 16q, plain options, one launch per pair (qm_reset_step_kernel)", env, len(gs3), 1, ["reset + step"], True, out)
    del env
    env = VecEnv("clifford", 16, gs3, B, add_inverts=False, add_perms=False, track_solution=True, difficulty=256)
    run("CliffordGym 16q, no add_inverts but the solution log, one launch per pair (qm_reset_step_kernel<.., FEAT>)", env, len(gs3), 1, ["reset + step"], True, out)
    del env
    env = VecEnv("clifford", 16, gs3, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
    run("CliffordGym 16q, plain options, two launches per pair", env, len(gs3), 2, ["reset_done (qm_init_kernel)", "step (qm_step1_kernel<LIST>)"], False, out)
    del env
    env = VecEnv("clifford", 16, gs3, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=256)
    run("CliffordGym 16q, reference defaults (add_inverts, solution log), one launch per pair (qm_reset_inv2_step_kernel)", env, len(gs3), 1, ["reset + step"], True, out)
    del env
    env = VecEnv("clifford", 16, gs3, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=256)
    run("CliffordGym 16q, reference defaults (add_inverts, solution log), two launches per pair", env, len(gs3), 2,
        ["reset_done (qm_init_kernel)", "step (qm_inv2_kernel<LIST>)"], False, out)
    del env
    gs2 = line_gateset("linear_function", 8)
    for nb in (8192, B):
        env = VecEnv("linear_function", 8, gs2, nb, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
        run(f"LinearFunctionGym 8q x {nb}, one launch per pair (word_reset_step_kernel)", env, len(gs2), 1, ["reset + step"], True, out)
        del env
    gs24 = line_gateset("clifford", 24)
    env = VecEnv("clifford", 24, gs24, B, add_perms=False, difficulty=256, add_inverts=False, track_solution=False)
    run("CliffordGym 24q (64-bit rows), plain options, one launch per pair (q64_reset_step_kernel)", env, len(gs24), 1, ["reset + step"], True, out)
    del env
    env = VecEnv("clifford", 24, gs24, B, add_perms=False, difficulty=256, add_inverts=True, track_solution=True)
    run("CliffordGym 24q, reference defaults (add_inverts, solution log), two launches per pair (the step leaves its finishers as a mask)", env, len(gs24), 2,
        ["q64_reset_done_kernel", "q64_inv2_kernel"], True, out)
    del env
    gs5 = line_gateset("pauli", 20)
    env = VecEnv("pauli", 20, gs5, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
    run("PauliGym 20q (tree + generate + step per pair; the step leaves its finishers as a mask)", env, len(gs5), 3,
        ["ptile_reset_tree_kernel", "ptile_generate_kernel", "ptile_step1c_kernel"], True, out, reset_slots=2)
    del env
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        open(args.out, "w").write("Auto-reset pairs on the kernel device clock (tools/auto_reset_breakdown.py; 65 536 envs unless said, 1/128 of the batch finishes per step)\n\n"
                                  + "\n".join(out) + "\n")


if __name__ == "__main__":
    main()
