"""Regenerate the numbers block of DESIGN.md section 4 (between the `numbers:begin` / `numbers:end` markers) and profiles/r04/NUMBERS.md from
the files the GPU runs committed: profiles/r04/traffic.json (tools/profile_bench.sh + tools/pmc_traffic.py) and the two bench lines
(bench_driver_args.json = the driver's arguments, bench_n1.json = the defaults).  No number in that block is typed by hand.

  python tools/design_numbers.py [profiles dir = profiles/r04]
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04")


def line_of(name):
    path = os.path.join(prof, name)
    if not os.path.exists(path):
        return None
    rows = [ln for ln in open(path).read().splitlines() if ln.startswith('{"metric"')]
    return json.loads(rows[-1]) if rows else None


def f(x, nd=2):
    return "–" if x is None else f"{x:.{nd}f}"


traffic = json.load(open(os.path.join(prof, "traffic.json")))["configs"]
drv, dflt = line_of("bench_driver_args.json"), line_of("bench_n1.json")
out = []
out.append("| Configuration (kernel) | envs | rocprof avg µs | live µs / step | env-steps/s (live) | PMC B/env (fetch + write) | needed B/env | 8d B/env | frac of 8 TB/s: 8d bytes / moved bytes / needed bytes |")
out.append("|---|---|---|---|---|---|---|---|---|")
names = {"bench": "C3 CliffordGym 16q, headline", "C3_1048576": "C3 at 2²⁰ envs", "C3_4194304": "C3 at 2²² envs (beyond the Infinity Cache)",
         "C3d": "C3 with the reference defaults (inverts + solution log)", "C2": "C2 LinearFunctionGym 8q", "C5": "C5 PauliGym 20q, device-generated targets",
         "dense": "C3 dense observation, full rewrite", "tracked": "C3 step that keeps a resident dense observation",
         "tracked_default": "... with the reference defaults"}
live_cfg = (dflt or drv or {}).get("configs") or {}
dense_live = {}
for fn, keys in (("dense_live.json", {"dense": "dense", "tracked": "tracked"}), ("dense_default_live.json", {"tracked_default": "tracked"})):
    try:
        modes = json.loads(open(os.path.join(prof, fn)).read().strip().splitlines()[-1])["modes"]
        for k, m in keys.items():
            dense_live[k] = modes[m]["us_per_step"]
    except Exception:
        pass
for key in ("bench", "C3_1048576", "C3_4194304", "C3d", "C2", "C5", "dense", "tracked", "tracked_default"):
    e = traffic.get(key)
    if not e:
        continue
    st = e.get("rocprof_kernel_stats") or {}
    live_us = None
    if key == "bench" and dflt:
        live_us = dflt["roofline"]["kernel_us_graph_period"]
    elif key in live_cfg and isinstance(live_cfg[key], dict):
        live_us = live_cfg[key].get("us_per_step")
    elif key in dense_live:
        live_us = dense_live[key]  # (dense: step + rewrite per graph step)
    elif e.get("live"):
        live_us = e["live"].get("launch_us")
    rate = e["envs"] / live_us * 1e6 if live_us else None
    out.append(f"| {names[key]} (`{e['kernel'].replace('qg::', '')}`) | {e['envs']} | {f(st.get('avg_us'))} | {f(live_us)} | {'–' if rate is None else f'{rate:.2e}'} | "
               f"{f(e['bytes_per_env'], 1)} ({f(e['fetch_per_env'], 1)} + {f(e['write_per_env'], 1)}) | {e.get('needed_bytes_per_env') or '–'} | {e['survey_8d_bytes_per_env']} | "
               f"{f(e.get('rocprof_frac_algorithmic'), 3)} / {f(e.get('rocprof_frac_moved'), 3)} / {f(e.get('rocprof_frac_needed'), 3)} |")
out.append("")
for label, d in (("driver's arguments (`--gpus 1 --steps 20 --warmup 5`)", drv), ("defaults (K = 2048, W = 128)", dflt)):
    if not d:
        continue
    r = d["roofline"]
    out.append(f"* Bench line, {label}: **{d['value']:.3e} env-steps/s**, {d['ms_per_step'] * 1e3:.2f} µs per step; `roofline.frac` {r['frac']:.3f} "
               f"(this run's own clock; committed rocprofv3 average {f((r.get('rocprof_committed') or {}).get('avg_us'))} µs); live clocks: timed region {f(r['kernel_us_timed_region'])} µs, graph period {f(r['kernel_us_graph_period'])} µs, "
               f"eager event {f(r['kernel_us_eager_event'])} µs; PMC traffic {r['traffic'] / 1e6 if r.get('traffic') else float('nan'):.2f} MB per launch = "
               f"{f(r.get('traffic_over_needed'))}× the needed bytes.")
d = dflt or drv
if d:
    cb = d.get("cpu_baseline")
    if cb:
        out.append(f"* CPU baseline (C port of the reference's scalar path, OpenMP over envs): {cb['value']:.2e} env-steps/s on {cb['cores']} cores, "
                   f"{cb['one_core']['value']:.2e} on one.")
    ar = d.get("auto_reset")
    if ar:
        out.append(f"* Auto-reset (step + `reset_done` per step, one hipGraph): desynchronised episodes ({ar['finished_per_step'] * 100:.2f} % of the batch finishes per step) "
                   f"**{ar['us_per_step']:.1f} µs**, synchronised {ar['synchronised']['us_per_step']:.1f} µs"
                   + (f"; with the reference-default options (two launches per pair) {ar['reference_defaults']['us_per_step']:.1f} µs" if ar.get("reference_defaults") else "")
                   + (f"; PauliGym 20q `reset_done` alone at 1 % finished, eager: {ar['pauli_reset_done']['us_per_call']:.0f} µs" if ar.get("pauli_reset_done") else "") + ".")
    om = d.get("observation_modes")
    if om:
        out.append(f"* Observation after every step (SURVEY 8d, both modes): packed {om['packed']['us_per_step']:.2f} µs per step; dense, full rewrite "
                   f"**{om['dense']['us_per_step']:.2f} µs** ({om['dense']['roofline']['frac']:.2f} of 8 TB/s on 8d's 1 184 B; the rewrite kernel alone "
                   f"{om['dense_kernel']['us_per_launch']:.2f} µs = {om['dense_kernel']['roofline']['frac']:.2f} on written bytes); dense, tracked in the step "
                   f"**{om['dense_tracked']['us_per_step']:.2f} µs**" +
                   (f"; tracked with the reference-default options {om['dense_tracked_reference_defaults']['us_per_step']:.2f} µs" if om.get('dense_tracked_reference_defaults') else "") + ".")
    dc = d.get("default_config")
    if dc:
        out.append(f"* Reference-default options (coins given): {dc['us_per_step']:.2f} µs per step.")
    fr = d.get("fused_rollout")
    if fr:
        out.append(f"* Fused rollout (128 steps per launch, rows in LDS): {fr['value']:.2e} env-steps/s.")
    lb = d.get("large_batch")
    if lb:
        out.append("* Larger batches (live): " + "; ".join(f"{b['envs']} envs {b['launch_us']:.1f} µs ({b['frac']:.2f} of 8 TB/s on 8d bytes)" for b in lb["by_batch"]) + ".")
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
