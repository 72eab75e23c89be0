"""bench_legs.py -- the legs bench.py reports BESIDE its headline (N = 1, the metric's own batch size): the fused rollout, the reference-default options,
SURVEY 8(d)'s auto-reset variants and observation modes, its other configurations, larger batches, a policy in the loop.  Each leg takes the run's context
`c` (device, stream, generator, the headline env's gateset and action ring, the parsed arguments) and returns the object that goes into the line, or
None when the leg does not apply (N > 1, another batch size, switched off)."""
from __future__ import annotations

import time

import numpy as np
import torch

from bench_common import (ALGO_BYTES_PER_STEP, CHUNK, ENVS_PER_GPU, HBM_PEAK_GBS, NEEDED_BYTES_PER_STEP, RING, SCRAMBLE, oracle_replay, pack_rows_u32, pmc_traffic,
                          profiled_configs, rocprof_kernel_avg_us)


def fused_rollout(c):
    """Fused rollout (state in LDS across steps), reported beside the headline."""
    env = c.env  # (the headline handle, after its timed run and snapshot)
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    # ---- fused rollout (state in LDS across steps), reported beside the headline --------
    fused = None
    if not multi:
        FT = 128
        facts = torch.randint(0, A, (FT, B), dtype=torch.int32, device=dev, generator=gen)
        with torch.cuda.stream(stream):
            env.rollout(facts, fused=True)
            torch.cuda.synchronize()
            f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            f0.record(stream)
            for _ in range(8):
                env.rollout(facts, fused=True)
            f1.record(stream)
        torch.cuda.synchronize()
        fms = f0.elapsed_time(f1)
        fused = {"value": B * FT * 8 / (fms * 1e-3), "unit": "env-steps/s", "steps_per_launch": FT,
                 "kernel": "qg::qm_fused_lds_kernel<16, true, false> (rows resident in LDS; actions known up front)"}

    return fused


def _graph_period(c, venv, acts, replays=2):
    """Launch period of a handle's step kernel: HIP events around `replays` replays of its CHUNK-launch rollout graph."""
    stream = c.stream
    with torch.cuda.stream(stream):
        venv.rollout_ring(acts, CHUNK)
        torch.cuda.synchronize()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record(stream)
        for _ in range(replays):
            venv.rollout_ring(acts, CHUNK)
        b1.record(stream)
    torch.cuda.synchronize()
    venv.sync()
    return b0.elapsed_time(b1) * 1e3 / (replays * CHUNK)



def default_config(c):
    """The reference's DEFAULT configuration (envs/synthesis.py:182-204: add_inverts=True, track_solution=True)."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    # ---- the reference's DEFAULT configuration (envs/synthesis.py:182-204: add_inverts=True, track_solution=True) -------
    default_cfg = None
    if not multi and B == ENVS_PER_GPU and not args.no_default_config:
        DT = 128  # = max_depth (envs/synthesis.py:188): what one episode, and its solution log, can hold
        denv = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE)
        coins = torch.randint(0, 2, (DT, B), dtype=torch.uint8, device=dev, generator=gen)
        dacts = torch.randint(0, A, (DT, B), dtype=torch.int32, device=dev, generator=gen)
        dms = 0.0
        with torch.cuda.stream(stream):
            denv.reset(seed)
            denv.rollout(dacts, coins=coins)  # builds the graph
            for _ in range(4):
                denv.reset(seed)  # a new episode: the log is empty again (not timed)
                d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                d0.record(stream)
                denv.rollout(dacts, coins=coins)
                d1.record(stream)
                torch.cuda.synchronize()
                dms += d0.elapsed_time(d1)
        denv.sync()
        dus = dms * 1e3 / (4 * DT)
        default_cfg = {"us_per_step": dus, "value": B / (dus * 1e-6), "unit": "env-steps/s",
                       "config": "add_inverts=True (coin per env per step), track_solution=True, otherwise the headline workload; "
                                 f"episodes of {DT} steps, each one hipGraph of {DT} launches, coins given"}
        del denv, coins, dacts

    return default_cfg


def auto_reset(c):
    """SURVEY 8(d)'s auto-reset variant of C3, C2 and C5: qg_vec_reset_done after every step, one captured graph of 128 pairs."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    # ---- SURVEY 8(d)'s auto-reset variant of the headline workload: qg_vec_reset_done after every step (finished episodes start over on the
    # device: scramble from the identity), one captured graph of 128 x (step, reset_done).  Two episode schedules: "desynchronised" -- what a
    # collector sees: episode ends spread evenly over time, 1/128 of the batch finishes in every step (Env::reset called for class
    # env % 128 == k at warm-up step k) -- and "synchronised" (every env finishes in the same step, 127 of 128 reset_done calls find nothing) -----
    auto_reset = None
    if not multi and B == ENVS_PER_GPU and not args.no_default_config:
        AT = 128  # = min(depth_slope * difficulty, max_depth) for every configuration below: the episode length

        def auto_reset_leg(aenv, num_actions, spread=True):
            """One captured graph of AT x (step, reset_done), the pair issued as qg_vec_reset_done_step (ONE launch where the layout has it: TILE without
            add_inverts, LF8, PERM), replayed 4 times.  `spread`: Env::reset for class env % AT == k at warm-up step k, so 1 / AT of the batch finishes in
            every step of the graph (what a collector sees); else every env finishes in the same step."""
            nb = aenv.batch
            aacts = torch.randint(0, num_actions, (AT, nb), dtype=torch.int32, device=dev, generator=gen)
            afin = torch.empty((AT, nb), dtype=torch.uint8, device=dev)
            with torch.cuda.stream(stream):
                aenv.reset(seed)
                if spread:
                    cls = torch.arange(nb, device=dev) % AT
                    for k in range(AT):  # eager warm-up: spreads the episode ends (the done flags are caller-owned memory, qg_vec_bind_outputs)
                        aenv.set_counters(k, k)
                        aenv.step(aacts[k])
                        aenv.reset_done(seed + 0x51ED * (k + 1))
                        aenv.done[cls == k] = 1
                        aenv.reset_done(seed + 0xA5A5 * (k + 1))

                def episode():  # step, then AT - 1 x (reset_done, step) as qg_vec_reset_done_step, and the last reset_done
                    aenv.set_counters(0, 0)
                    aenv.rollout(aacts[0:1], dones_out=afin[0:1])
                    for t in range(1, AT):
                        aenv.set_counters(t, t)
                        aenv.reset_done_step(seed + 0x9E3779B9 * t, aacts[t], dones_out=afin[t])
                    aenv.reset_done(seed + 0x9E3779B9 * AT)

                episode()  # eager pass (allocations, kernel loads)
                torch.cuda.synchronize()
                ag = torch.cuda.CUDAGraph()
                with torch.cuda.graph(ag, stream=stream):
                    episode()
                torch.cuda.synchronize()
                ag.replay()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(stream)
                for _ in range(4):
                    ag.replay()
                a1.record(stream)
            torch.cuda.synchronize()
            aenv.sync()
            aus = a0.elapsed_time(a1) * 1e3 / (4 * AT)
            per_step = afin.float().mean(dim=1)
            return {"us_per_step": aus, "value": nb / (aus * 1e-6), "unit": "env-steps/s", "envs": nb, "finished_per_step": float(per_step.mean()),
                    "finished_per_step_min_max": [float(per_step.min()), float(per_step.max())]}

        from util import line_gateset as _line_gateset  # (the gateset builder the configs leg below uses)

        legs = {}
        for schedule in ("desynchronised", "synchronised", "desynchronised_reference_defaults"):
            ref_defaults = schedule.endswith("reference_defaults")  # add_inverts + solution log: the pair is two launches behind the one call
            aenv = VecEnv("clifford", n, gateset, B, add_inverts=ref_defaults, add_perms=False, track_solution=ref_defaults, difficulty=SCRAMBLE)
            legs[schedule] = auto_reset_leg(aenv, A, spread=schedule != "synchronised")
            del aenv
        # "an auto-reset variant reported separately" for the other configurations too, same schedule (episode ends spread evenly over time)
        gs2a = _line_gateset("linear_function", 8)
        for name, nb, kw in (("C2", 8192, dict(add_inverts=False, track_solution=False)), ("C2_x65536", B, dict(add_inverts=False, track_solution=False)),
                             ("C2_reference_defaults", 8192, dict(add_inverts=True, track_solution=True))):
            aenv = VecEnv("linear_function", 8, gs2a, nb, add_perms=False, difficulty=64, **kw)
            legs[name] = dict(auto_reset_leg(aenv, len(gs2a)), config=f"LinearFunctionGym 8q x {nb}, difficulty 64, {kw}: word_reset_step_kernel (one launch per pair)")
            del aenv
        gs24 = _line_gateset("clifford", 24)
        aenv = VecEnv("clifford", 24, gs24, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
        legs["clifford24"] = dict(auto_reset_leg(aenv, len(gs24)), config=f"CliffordGym 24q x {B} (64-bit rows), difficulty 256: q64_reset_step_kernel (one launch per pair)")
        del aenv
        gs5a = _line_gateset("pauli", 20)
        aenv = VecEnv("pauli", 20, gs5a, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
        legs["C5"] = dict(auto_reset_leg(aenv, len(gs5a)), config=f"PauliGym 20q x {B}, difficulty 256 (targets regenerated on the device), "
                                                                  "ptile_reset_tree_kernel + ptile_generate_kernel + ptile_step1c_kernel<LIST> (finishers as a mask) per pair")
        del aenv
        # ... and qg_vec_reset_done by itself where it is not a tree of row operations on a bit matrix: PauliGym 20q (config 5's env: a fresh target is
        # generated on the device), 1 % of the batch finished, eager calls (memset + compaction + two kernels), median of 12
        pg_n = 20
        pg_gs = gs5a
        penv = VecEnv("pauli", pg_n, pg_gs, B, add_perms=False, track_solution=False, difficulty=128)
        with torch.cuda.stream(stream):
            penv.reset(seed)
            pmask = (torch.rand(B, device=dev, generator=gen) < 0.01).to(torch.uint8)
            ptimes = []
            for i in range(12):
                penv.done.copy_(pmask)
                p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                p0.record(stream)
                penv.reset_done(seed + 100 + i)
                p1.record(stream)
                torch.cuda.synchronize()
                ptimes.append(p0.elapsed_time(p1) * 1e3)
        penv.sync()
        legs["pauli_reset_done"] = {"us_per_call": sorted(ptimes)[len(ptimes) // 2], "finished": float(pmask.float().mean()),
                                    "config": f"PauliGym {pg_n}q x {B} envs, difficulty 128, qg_vec_reset_done with 1 % of the batch finished, eager, median of 12"}
        del penv
        auto_reset = dict(legs["desynchronised"], synchronised=legs["synchronised"], reference_defaults=legs["desynchronised_reference_defaults"],
                          C2=legs["C2"], C2_x65536=legs["C2_x65536"], C2_reference_defaults=legs["C2_reference_defaults"], C5=legs["C5"], clifford24=legs["clifford24"],
                          pauli_reset_done=legs["pauli_reset_done"],
                          config=f"the headline workload with qg_vec_reset_done after every step (the pair reset_done + next step issued as qg_vec_reset_done_step: one launch), episodes of min(depth_slope * difficulty, max_depth) = {AT} steps, "
                                 f"a captured graph of {AT} x (step, reset_done) replayed 4 times; headline figures: episode ends spread evenly over time "
                                 "(parity of this schedule: tests/test_gpu_fullsize.py::test_auto_reset_with_desynchronised_episodes_at_full_size)")

    return auto_reset


def observation_modes(c):
    """SURVEY 8(d) "report both modes": the observation handed to the learner after EVERY step of the headline workload."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    # ---- SURVEY 8(d) "report both modes": the observation handed to the learner after EVERY step of the headline workload.  The Gym adapter
    # returns the dense int8 [32, 32] matrix per env on every step() (adapters.py:50-54,62-72).  Four graphs of 128 steps each:
    #   packed        step + qg_vec_observe_packed (the bit-packed [B, 32] row words in env-major order: what the all-gather moves)
    #   dense         step + qg_vec_observe_dense  (8d's 1 184 B per env-step: a full 1 KiB rewrite per env)
    #   dense_tracked step on a handle with qg_vec_track_dense: the step kernel itself rewrites the <= 4 rows its gate changed in a RESIDENT
    #                 dense observation (same bytes in the tensor after every step, <= 128 of the 1 024 written)
    #   dense_kernel  qg_vec_observe_dense alone (launch period): the HBM-write roofline of the rewrite kernel on the bytes it writes
    obs_modes = None
    if not multi and B == ENVS_PER_GPU and not args.no_dense_obs:
        OT = 128
        D2 = 4 * n * n  # dense bytes per env
        oenv = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
        tenv = VecEnv("clifford", n, gateset, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
        obs_d = torch.empty((B, 2 * n, 2 * n), dtype=torch.int8, device=dev)
        obs_p = torch.empty((B, 2 * n), dtype=torch.int32, device=dev)

        def graph_of(body, replays=4):
            with torch.cuda.stream(stream):
                body()  # eager pass
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=stream):
                    body()
                torch.cuda.synchronize()
                gr.replay()
                torch.cuda.synchronize()
                o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                o0.record(stream)
                for _ in range(replays):
                    gr.replay()
                o1.record(stream)
            torch.cuda.synchronize()
            return o0.elapsed_time(o1) * 1e3 / (replays * OT)

        def steps_with(env_, after):
            def body():
                for t in range(OT):
                    env_.step(actions[t % RING])
                    after()
            return body

        with torch.cuda.stream(stream):
            oenv.reset(seed)
            tenv.reset(seed)
            tracked = tenv.track_dense()
        us_packed = graph_of(steps_with(oenv, lambda: oenv.observe_packed(out=obs_p)))
        us_dense = graph_of(steps_with(oenv, lambda: oenv.observe(out=obs_d)))
        us_kernel = graph_of(lambda: [oenv.observe(out=obs_d) for _ in range(OT)])
        us_tracked = graph_of(steps_with(tenv, lambda: None))
        # ... and with the reference's default options (add_inverts=True, track_solution=True; coins given): the two-lanes-per-env step rewrites
        # an env's whole observation when its coin inverts the matrix (half of the envs per step), the gate's rows otherwise
        denv2 = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE, max_depth=7 * OT)
        dcoins = torch.randint(0, 2, (RING, B), dtype=torch.uint8, device=dev, generator=gen)
        with torch.cuda.stream(stream):
            denv2.reset(seed)
            tracked2 = denv2.track_dense()

        def default_steps():
            for t in range(OT):
                denv2.step(actions[t % RING], dcoins[t % RING])

        us_tracked_default = graph_of(default_steps)
        denv2.sync()
        with torch.cuda.stream(stream):
            same2 = bool(torch.equal(tracked2, denv2.observe()))
        if not same2:
            raise SystemExit("bench.py: the tracked dense observation (reference-default options) differs from a full rewrite of the same state")
        del denv2, tracked2, dcoins
        oenv.sync()
        tenv.sync()
        with torch.cuda.stream(stream):
            oenv.observe_packed(out=obs_p)  # the packed words of the final state (the packed graph ran before the dense one)
            same = bool(torch.equal(tracked, tenv.observe())) and bool(torch.equal(obs_d, oenv.observe()))
        if not same:
            raise SystemExit("bench.py: the tracked dense observation differs from a full rewrite of the same state")
        # oracle check of every env: oenv and tenv took the same steps from the same reset (eager pass + 6 replays of each graph)
        obs_parity = None
        if rank == 0 and not args.no_parity:
            o_steps = 2 * 6 * OT  # oenv: two stepping graphs (eager pass + 5 replays each); tenv: one
            t_steps = 6 * OT
            sample = np.arange(B)  # every env
            acts_np = host_actions[:, sample].numpy()
            ov_o, _ = oracle_replay(gateset, seed, env_base + sample, acts_np, [t % RING for t in range(OT)] * (o_steps // OT))
            ov_t, _ = oracle_replay(gateset, seed, env_base + sample, acts_np, [t % RING for t in range(OT)] * (t_steps // OT))
            sidx = torch.as_tensor(sample, device=dev)
            ok_d = bool(np.array_equal(obs_d[sidx].cpu().numpy().reshape(len(sample), -1), ov_o.observe_dense()))
            ok_t = bool(np.array_equal(tracked[sidx].cpu().numpy().reshape(len(sample), -1), ov_t.observe_dense()))
            ok_p = bool(np.array_equal(obs_p[sidx].cpu().numpy().view(np.uint32), pack_rows_u32(ov_o.observe_dense().reshape(len(sample), 2 * n, 2 * n))))
            obs_parity = {"envs": int(len(sample)), "dense_rewrite": ok_d, "dense_tracked": ok_t, "packed": ok_p, "bit_exact": ok_d and ok_t and ok_p}
            if not obs_parity["bit_exact"]:
                raise SystemExit(f"bench.py: observation modes differ from the CPU oracle replay: {obs_parity}")

        def mode(us, bytes_per_env, what):
            gbs = bytes_per_env * B / (us * 1e-6) / 1e9
            return {"us_per_step": us, "value": B / (us * 1e-6), "unit": "env-steps/s", "bytes_per_env_step": bytes_per_env,
                    "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}, "what": what}

        obs_modes = {
            "packed": mode(us_packed, ALGO_BYTES_PER_STEP + 2 * 4 * 2 * n,
                           "step + qg_vec_observe_packed per step: 8d's 160 B + the env-major copy of the 32 row words (128 B read, 128 B written)"),
            "dense": mode(us_dense, ALGO_BYTES_PER_STEP + D2, "step + qg_vec_observe_dense per step: SURVEY 8d's 1 184 B per env-step (full 1 KiB int8 rewrite)"),
            "dense_tracked": {"us_per_step": us_tracked, "value": B / (us_tracked * 1e-6), "unit": "env-steps/s",
                              "bytes_moved_per_env_step": NEEDED_BYTES_PER_STEP + 64,
                              "frac_moved": (NEEDED_BYTES_PER_STEP + 64) * B / (us_tracked * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              "what": "qg_vec_track_dense: the step kernel rewrites the rows its gate changed in a resident dense observation; the tensor "
                                      "holds the same bytes as after the full rewrite.  No fraction on 8d's 1 184 B: this form does not move most of them; "
                                      "frac_moved is on the bytes it has to move"},
            "dense_tracked_reference_defaults": {
                "us_per_step": us_tracked_default, "value": B / (us_tracked_default * 1e-6), "unit": "env-steps/s",
                "what": "qg_vec_track_dense with add_inverts=True, track_solution=True (coins given): qm_inv2_kernel rewrites the whole env when its coin fires "
                        "(wave-contiguous 1 KiB stores), the gate's rows otherwise; equals a full rewrite of the final state (checked)"},
            "dense_kernel": {"kernel": "qg::qm_dense_stream_kernel<2>", "us_per_launch": us_kernel, "bytes_written_per_launch": D2 * B,
                             "roofline": {"bound": "hbm", "achieved": D2 * B / (us_kernel * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": D2 * B / (us_kernel * 1e-6) / 1e9 / HBM_PEAK_GBS},
                             "what": "launch period of qg_vec_observe_dense alone (hipGraph of 128 launches); written bytes only (it reads 128 B per env)"},
            "parity": obs_parity,
            "config": f"the headline workload, one observation after every step, hipGraphs of {OT} steps replayed 4 times; the average changed-row count of the "
                      "170-action gateset is 1.98 rows of 32 B per step",
        }
        del oenv, tenv, obs_d, obs_p, tracked

    return obs_modes


def other_configs(c):
    """SURVEY 8(d)'s other configurations (C2, C5, C3d): live launch period beside the committed rocprofv3 / PMC figures."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    graph_period = lambda venv, acts, replays=2: _graph_period(c, venv, acts, replays)  # noqa: E731
    # ---- SURVEY 8(d)'s other configurations: live launch period (hipGraph of 128 single-step launches) beside the committed rocprofv3 / PMC
    # figures of the same kernels (profiles/r05/traffic.json, tools/profile_bench.sh) -----
    configs = None
    if not multi and B == ENVS_PER_GPU and not args.no_configs:
        from util import line_gateset

        prof = profiled_configs()
        configs = {}

        def leg(name, venv, acts, coins=None):
            with torch.cuda.stream(stream):
                if coins is None:
                    us = graph_period(venv, acts)
                else:
                    venv.rollout(acts, coins=coins)  # builds the graph
                    tot = 0.0
                    for _ in range(3):
                        venv.reset(seed)  # a new episode: the solution log is empty again (not timed)
                        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        c0.record(stream)
                        venv.rollout(acts, coins=coins)
                        c1.record(stream)
                        torch.cuda.synchronize()
                        tot += c0.elapsed_time(c1)
                    venv.sync()
                    us = tot * 1e3 / (3 * acts.shape[0])
            p = prof.get(name) or {}
            st = p.get("rocprof_kernel_stats") or {}
            row = {"kernel": p.get("kernel"), "envs": venv.batch, "us_per_step": us, "value": venv.batch / (us * 1e-6), "unit": "env-steps/s",
                   "rocprof_avg_us": st.get("avg_us"), "pmc_bytes_per_env": p.get("bytes_per_env"), "needed_bytes_per_env": p.get("needed_bytes_per_env"),
                   "survey_8d_bytes_per_env": p.get("survey_8d_bytes_per_env"), "frac_moved": p.get("rocprof_frac_moved"),
                   "frac_survey_8d": p.get("rocprof_frac_algorithmic"), "source": "profiles/r05/traffic.json" if p else None}
            configs[name] = row

        gs2 = line_gateset("linear_function", 8)
        e2 = VecEnv("linear_function", 8, gs2, 8192, add_inverts=False, add_perms=False, track_solution=False, difficulty=64)
        with torch.cuda.stream(stream):
            e2.reset(0x5EED0002)
        leg("C2", e2, torch.randint(0, len(gs2), (RING, 8192), dtype=torch.int32, device=dev, generator=gen))
        del e2
        gs5 = line_gateset("pauli", 20)
        e5 = VecEnv("pauli", 20, gs5, B, add_perms=False, track_solution=False, max_rotations=5, difficulty=256, pauli_diff_scale=8)
        with torch.cuda.stream(stream):
            e5.reset(0x5EED0005)  # targets generated on the device: 1-7 rotations per env, tableau scrambled by 256 gates
        leg("C5", e5, torch.randint(0, len(gs5), (RING, B), dtype=torch.int32, device=dev, generator=gen))
        del e5
        e3d = VecEnv("clifford", n, gateset, B, add_inverts=True, add_perms=False, track_solution=True, difficulty=SCRAMBLE)
        with torch.cuda.stream(stream):
            e3d.reset(seed)
        leg("C3d", e3d, torch.randint(0, A, (128, B), dtype=torch.int32, device=dev, generator=gen),
            coins=torch.randint(0, 2, (128, B), dtype=torch.uint8, device=dev, generator=gen))
        del e3d
        configs["note"] = ("C2 LinearFunctionGym 8q x 8 192, C5 PauliGym 20q x 65 536 (device-generated targets), C3d CliffordGym 16q x 65 536 with the "
                           "reference's default add_inverts=True / track_solution=True; us_per_step is live (HIP events around hipGraph replays of "
                           "single-step launches), the other columns are the committed rocprofv3 --kernel-trace --stats and --pmc passes of tools/run_config.py")

    return configs


def large_batch(c):
    """The same step kernel at larger batches: where the launch boundary stops dominating, and beyond the Infinity Cache."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    graph_period = lambda venv, acts, replays=2: _graph_period(c, venv, acts, replays)  # noqa: E731
    # ---- the same step kernel at larger batches: where the launch boundary (1.6 us) stops dominating, and beyond the Infinity Cache -----
    large = None
    if not multi and B == ENVS_PER_GPU and not args.no_large_batch:
        sizes = []
        for LB in (1 << 18, 1 << 20, 1 << 22):
            big = VecEnv("clifford", n, gateset, LB, add_inverts=False, add_perms=False, track_solution=False, difficulty=SCRAMBLE)
            bacts = torch.randint(0, A, (RING, LB), dtype=torch.int32, device=dev, generator=gen)
            with torch.cuda.stream(stream):
                big.reset(seed)
            lus = graph_period(big, bacts)
            lgb = ALGO_BYTES_PER_STEP * LB / (lus * 1e-6) / 1e9
            sizes.append({"envs": LB, "launch_us": lus, "achieved": lgb, "unit": "GB/s", "frac": lgb / HBM_PEAK_GBS,
                          "state_MiB": LB * 128 / 2**20, "rocprof": rocprof_kernel_avg_us(LB), "traffic": pmc_traffic(LB)})
            del big, bacts
        large = {"note": "same kernel and layout at 4x / 16x / 64x the batch: per-launch time is kernel time, not launch boundary; "
                         "2^22 envs = 512 MiB of state, beyond the 256 MiB Infinity Cache", "by_batch": sizes}

    return large


def policy_in_loop(c):
    """With a policy in the loop: forward + categorical draw + env.step() + auto-reset per collection step, everything on the GPU."""
    multi, B, args, VecEnv, n, gateset, A, dev, gen, stream, seed, actions, host_actions, env_base, rank = (
        c.multi, c.B, c.args, c.VecEnv, c.n, c.gateset, c.A, c.dev, c.gen, c.stream, c.seed, c.actions, c.host_actions, c.env_base, c.rank)
    # ---- with a policy in the loop: the reference's default network shape (rl/configs.py:531-607, bf16, random weights) forward +
    # categorical draw + env.step() + auto-reset per collection step, everything on the GPU; informational, not the headline metric -----
    collector = None
    if not multi and B == ENVS_PER_GPU and not args.no_collector:
        from qiskit_gym_amd.collector import BasicPolicy, RolloutCollector
        rows = []
        for CB in (1024, B):  # the reference's num_episodes (rl/configs.py:134) and the headline batch
            cenv = VecEnv("clifford", n, gateset, CB, add_inverts=False, add_perms=False, track_solution=False, difficulty=32)
            col = RolloutCollector(cenv, BasicPolicy(4 * n * n, A), dtype=torch.bfloat16, seed=1, store_obs="packed", use_graph=True)
            CT = 32
            col.collect(CT)  # eager pass + capture
            col.collect(CT)  # first replay (the graph's one-time upload: tens of ms now and then)
            torch.cuda.synchronize()
            per_replay = []
            for _ in range(5):  # each replay timed on its own: a hipGraph launch now and then stalls for tens of ms (its upload), the median does not see it
                t0 = time.perf_counter()
                ro = col.collect(CT)
                torch.cuda.synchronize()
                per_replay.append((time.perf_counter() - t0) / CT * 1e6)
            cus = float(np.median(per_replay))
            cenv.sync()
            rows.append({"envs": CB, "us_per_step": cus, "value": CB / (cus * 1e-6), "unit": "env-steps/s", "done_per_step": float(ro.dones.float().mean())})
            del col, cenv, ro
            torch.cuda.empty_cache()
        collector = {"policy": f"BasicPolicy {4 * n * n}-512-256-{{{A}, 1}} bf16, random weights; packed observation stored per step; one hipGraph per 32-step collection",
                     "clock": "host wall clock around each of 5 replays (device idle before and after), median", "by_batch": rows}

    return collector
