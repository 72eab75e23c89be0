"""Regenerate the numbers block of DESIGN.md section 4 (between the `numbers:begin` / `numbers:end` markers) and profiles/<round>/NUMBERS.md from
the files the GPU runs committed: kernel_device_clock.json (tools/kernel_device_clock.py: the kernel's own clock beside the launch period), traffic.json
(tools/profile_bench.sh + tools/pmc_traffic.py: rocprofv3 kernel statistics and PMC bytes) and the two bench lines (bench_driver_args.json = the driver's
arguments, bench_n1.json = the defaults).  No number in that block is typed by hand.

  python tools/design_numbers.py [profiles dir = profiles/r05]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r05")
HBM = 8000.0


def line_of(name):
    path = os.path.join(prof, name)
    if not os.path.exists(path):
        return None
    rows = [ln for ln in open(path).read().splitlines() if ln.startswith('{"metric"')]
    return json.loads(rows[-1]) if rows else None


def f(x, nd=2):
    return "–" if x is None else f"{x:.{nd}f}"


traffic = json.load(open(os.path.join(prof, "traffic.json")))["configs"]
clock = {r["config"]: r for r in json.load(open(os.path.join(prof, "kernel_device_clock.json")))["rows"]}
drv, dflt = line_of("bench_driver_args.json"), line_of("bench_n1.json")
out = []
out.append("**Three clocks, one table.**  `device clock` = last wave exit − first wave entry, stamped by the kernel's own waves (`qg_vec_set_kernel_clock`; mean over 8 × 128 launches): "
           "the kernel's duration, and what every fraction below divides by.  `launch period` = HIP events around hipGraph replays: kernel + the launch boundary "
           "(`boundary` = the difference: 0.7–3 µs, growing with the grid and with the dirty lines the end of a kernel writes back) — what throughput counts.  "
           "`rocprofv3` = the `--kernel-trace --stats` average of the same kernel in its own run: each dispatch wrapped in the tool's own packets, so it sits ABOVE the launch "
           "period for short kernels (C3: 5.1 against 3.1) and agrees for long ones (2²² envs: 188 against 187).")
out.append("")
out.append("| Configuration (kernel) | envs | device clock µs (median, p90) | launch period µs | boundary µs | rocprofv3 avg µs | PMC B/env (fetch + write) | needed B/env | 8d B/env | "
           "on the device clock: GB/s of 8d bytes · frac of 8 TB/s on 8d / moved / needed bytes |")
out.append("|---|---|---|---|---|---|---|---|---|---|")
names = {"C3": ("bench", "C3 CliffordGym 16q, headline"), "C3_1048576": ("C3_1048576", "C3 at 2²⁰ envs"), "C3_4194304": ("C3_4194304", "C3 at 2²² envs (beyond the Infinity Cache)"),
         "C3d": ("C3d", "C3 with the reference defaults (inverts + solution log)"), "C2": ("C2", "C2 LinearFunctionGym 8q"),
         "C5": ("C5", "C5 PauliGym 20q, device-generated targets"), "C5_1048576": ("C5_1048576", "C5 at 2²⁰ envs"), "C5_4194304": ("C5_4194304", "C5 at 2²² envs"),
         "dense": ("dense", "C3 dense observation, full rewrite"), "tracked": ("tracked", "C3 step that keeps a resident dense observation")}
for key, (tkey, label) in names.items():
    c, e = clock.get(key), traffic.get(tkey)
    if not c:
        continue
    d = c["device_clock_us"]
    us = d["mean"]
    envs = c["envs"]
    algo = c["bytes_8d_per_env"]
    moved = e["bytes_per_env"] if e else None
    needed = e.get("needed_bytes_per_env") if e else None

    def frac(b):
        return None if not b else b * envs / (us * 1e-6) / 1e9 / HBM

    fr = [frac(algo), frac(moved), frac(needed)]
    note = ""
    if any(x is not None and x > 1 for x in fr):  # more bytes per second than HBM delivers: no fraction of its peak is quoted
        note = " (cache: the launch's working set is absorbed by the 256 MiB Infinity Cache -- or 8d counts bytes the kernel does not move)"
        fr = [None if x is None or x > 1 else x for x in fr]
    out.append(f"| {label} (`{c['kernel']}`) | {envs} | **{f(us)}** ({f(d['median'])}, {f(d['p90'])}) | {f(c['launch_period_us'])} | {f(c['launch_boundary_us'])} | "
               f"{f(c.get('rocprof_committed_avg_us'))} | " + (f"{f(moved, 1)} ({f(e['fetch_per_env'], 1)} + {f(e['write_per_env'], 1)})" if e else "–") +
               f" | {needed or '–'} | {algo} | {algo * envs / (us * 1e-6) / 1e9:.0f} · " + " / ".join("cache" if (x is None and note) else "–" if x is None else f"{x:.3f}" for x in fr) + note + " |")
out.append("")
for label, d in (("driver's arguments (`--gpus 1 --steps 20 --warmup 5`)", drv), ("defaults (K = 2048, W = 128)", dflt)):
    if not d:
        continue
    r = d["roofline"]
    k = r["kernel_us_device_clock"]
    out.append(f"* Bench line, {label}: **{d['value']:.3e} env-steps/s**, {d['ms_per_step'] * 1e3:.2f} µs per step; `roofline.frac` **{r['frac']:.3f}** = 8d bytes ÷ the step kernel's "
               f"device-clock duration in that run ({f(k['mean'])} µs mean, {f(k['median'])} median over {k['launches']} launches); launch period {f(r['launch_period_us'])} µs "
               f"(`frac_launch_period` {r['frac_launch_period']:.3f}), graph period {f(r['kernel_us_graph_period'])} µs, eager event {f(r['kernel_us_eager_event'])} µs, "
               f"committed rocprofv3 average {f((r.get('rocprof_committed') or {}).get('avg_us'))} µs; PMC traffic "
               f"{r['traffic'] / 1e6 if r.get('traffic') else float('nan'):.2f} MB per launch = {f(r.get('traffic_over_needed'))}× the needed bytes; parity replay: "
               f"{(d.get('parity') or {}).get('envs')} envs × {(d.get('parity') or {}).get('steps_replayed')} steps bit-exact.")
d = dflt or drv
if d:
    cb = d.get("cpu_baseline")
    if cb:
        out.append(f"* CPU baseline (C port of the reference's scalar path, OpenMP over envs): {cb['value']:.2e} env-steps/s on {cb['cores']} cores, "
                   f"{cb['one_core']['value']:.2e} on one.")
    ar = d.get("auto_reset")
    if ar:
        out.append(f"* Auto-reset (step + `reset_done` per step, one hipGraph of 128 pairs, {ar['finished_per_step'] * 100:.2f} % of the batch finishing per step): CliffordGym 16q "
                   f"**{ar['us_per_step']:.1f} µs** a pair (one launch), every env finishing in the same step {ar['synchronised']['us_per_step']:.1f} µs, with the reference-default options "
                   f"**{ar['reference_defaults']['us_per_step']:.1f} µs** (two launches); LinearFunctionGym 8q × 8 192 **{ar['C2']['us_per_step']:.1f} µs** (one launch; × 65 536: "
                   f"{ar['C2_x65536']['us_per_step']:.1f}; reference defaults: {ar['C2_reference_defaults']['us_per_step']:.1f}); PauliGym 20q × 65 536 **{ar['C5']['us_per_step']:.1f} µs** "
                   f"(tree + generator + step, three launches)" + (f"; CliffordGym 24q (64-bit rows) **{ar['clifford24']['us_per_step']:.1f} µs**" if 'clifford24' in ar else "") +
                   f"; PauliGym `reset_done` alone at 1 % finished, eager: {ar['pauli_reset_done']['us_per_call']:.0f} µs.")
    om = d.get("observation_modes")
    if om:
        out.append(f"* Observation after every step (SURVEY 8d, both modes): packed {om['packed']['us_per_step']:.2f} µs per step; dense, full rewrite "
                   f"**{om['dense']['us_per_step']:.2f} µs** ({om['dense']['roofline']['frac']:.2f} of 8 TB/s on 8d's 1 184 B per launch period; the rewrite kernel alone "
                   f"{om['dense_kernel']['us_per_launch']:.2f} µs = {om['dense_kernel']['roofline']['frac']:.2f} on written bytes); dense, tracked in the step "
                   f"**{om['dense_tracked']['us_per_step']:.2f} µs**; tracked with the reference-default options {om['dense_tracked_reference_defaults']['us_per_step']:.2f} µs.")
    dc = d.get("default_config")
    if dc:
        out.append(f"* Reference-default options (coins given): {dc['us_per_step']:.2f} µs per step.")
    fr = d.get("fused_rollout")
    if fr:
        out.append(f"* Fused rollout (128 steps per launch, rows in LDS): {fr['value']:.2e} env-steps/s.")
    lb = d.get("large_batch")
    if lb:
        out.append("* Larger batches (launch period): " + "; ".join(f"{b['envs']} envs {b['launch_us']:.1f} µs ({b['frac']:.2f} of 8 TB/s on 8d bytes)" for b in lb["by_batch"]) + ".")
    pl = d.get("policy_in_loop")
    if pl:
        out.append("* Policy in the loop (BasicPolicy 1024-512-256-{170, 1} bf16, one hipGraph per 32-step collection): " +
                   "; ".join(f"{b['envs']} envs {b['us_per_step']:.1f} µs per step = {b['value']:.2e} env-steps/s" for b in pl["by_batch"]) + ".")
block = "\n".join(out)
open(os.path.join(prof, "NUMBERS.md"), "w").write("# Numbers of DESIGN.md section 4 (generated by tools/design_numbers.py)\n\n" + block + "\n")
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s2 = re.sub(r"<!-- numbers:begin -->.*?<!-- numbers:end -->", lambda m: "<!-- numbers:begin -->\n" + block + "\n<!-- numbers:end -->", s, flags=re.S)
open(path, "w").write(s2)
print(block)
