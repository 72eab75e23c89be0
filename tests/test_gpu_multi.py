"""The multi-GPU leg as far as one GPU can show it (BASELINE config 4: 524 288 envs = 8 shards of 65 536):
a shard stepped with an env base is bit-identical to its slice of the whole batch, and bench.py's RCCL path -- one rank
standing in for rank 3 of 8 -- hands the learner a gathered shard that matches the CPU oracle for those env ids."""
import json
import os
import subprocess
import time
import sys

import numpy as np
import pytest
import torch

from oracle import OracleEnv, OracleVec
from util import f32_bits, line_gateset, rng_actions

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind,n,cfg", [
    ("clifford", 16, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=40)),
    ("clifford", 6, dict(add_inverts=True, add_perms=False, track_solution=True, difficulty=12)),   # counter-RNG coins
    ("linear_function", 8, dict(add_inverts=True, add_perms=False, track_solution=False, difficulty=20)),
    ("linear_function", 24, dict(add_inverts=True, add_perms=False, track_solution=False, difficulty=20)),  # lane-group kernels
    ("permutation", 9, dict(add_inverts=True, add_perms=False, track_solution=False, difficulty=10)),
    ("clifford", 24, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=30)),
    ("pauli", 6, dict(add_perms=True, track_solution=False, difficulty=48, pauli_diff_scale=8)),      # target generator + observe() permutations
])
def test_shard_with_env_base_equals_its_slice_of_the_whole_batch(kind, n, cfg):
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    A = len(gs)
    per, world, r = 192, 4, 2
    whole = VecEnv(kind, n, gs, per * world, **cfg)
    shard = VecEnv(kind, n, gs, per, env_base=r * per, **cfg)
    sl = slice(r * per, (r + 1) * per)
    whole.reset(77)
    shard.reset(77)
    g = torch.Generator(device="cuda").manual_seed(5)
    for t in range(12):
        if kind == "pauli":
            assert torch.equal(whole.observe()[sl], shard.observe())  # draws one permutation per env from the counter RNG
        acts = torch.randint(0, A, (per * world,), dtype=torch.int32, device="cuda", generator=g)
        whole.step(acts)
        shard.step(acts[sl].contiguous())
        assert torch.equal(whole.reward[sl].view(torch.int32), shard.reward.view(torch.int32)), (kind, t)
        assert torch.equal(whole.done[sl], shard.done) and torch.equal(whole.success[sl], shard.success)
    whole.sync()
    shard.sync()
    if kind == "pauli":
        assert torch.equal(whole.get_state("i64")[sl], shard.get_state("i64"))
    else:
        assert torch.equal(whole.get_state("packed")[sl], shard.get_state("packed"))
    # auto-reset of finished episodes draws by global env id too
    whole.reset_done(123)
    shard.reset_done(123)
    assert torch.equal(whole.depth[sl], shard.depth)
    fmt = "i64" if kind == "pauli" else "packed"
    assert torch.equal(whole.get_state(fmt)[sl], shard.get_state(fmt))


def _run_bench(*flags, timeout=600, extra_env=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, env=env)


def test_bench_rccl_path_gathered_shard_3_of_8_matches_oracle(tmp_path):
    """BASELINE config 4's shape on the hardware there is: one rank runs bench.py's multi-GPU path (process group, side-stream
    all-gather of the flat learner shard, --steps 20 so the driver's arguments put a collective inside the timed region) as
    rank 3 of 8, i.e. env ids [196 608, 262 144).  The gathered packed observation, rewards and flags of every env of the shard are
    replayed here on the oracle, independently of bench.py's own check."""
    dump = str(tmp_path / "gathered.npz")
    res = _run_bench("--force-multi", "--shard", "3/8", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-large-batch",
                     "--no-default-config", "--dump-gathered", dump)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    cfg = line["config"]
    assert cfg["ranks_seen"] == 1 and line["n_gpus"] == 1
    assert cfg["env_ids_of_rank0"] == [3 * 65536, 4 * 65536] and cfg["total_envs"] == 65536
    assert cfg["collective"]["collectives_in_timed_region"] >= 1 and cfg["collective"]["every_steps"] == 20
    assert cfg["collective"]["per_step_gather_us"] > 0 and cfg["collective"]["segment_gather_us"] > 0
    assert line["parity"]["bit_exact"] and line["parity"]["gathered_shard"]["bit_exact"]
    assert "hipGraph of 20 launches" in cfg["launch"]

    d = np.load(dump)
    ids = d["global_ids"]
    assert ids.min() >= 3 * 65536 and ids.max() < 4 * 65536 and int(d["total_envs"]) == 8 * 65536
    gs = line_gateset("clifford", 16)
    proto = OracleEnv("clifford", 16, gs, add_inverts=0, add_perms=0, track_solution=0, difficulty=int(d["scramble"]))
    ov = OracleVec(proto, len(ids))
    assert len(ids) == 65536  # every env of the shard
    ov.reset_with(rng_actions(int(d["seed"]), ids, int(d["scramble"]), len(gs)))  # (draws made in numpy: independent of og_vec_reset_seeded)
    r = s = f = None
    assert len(d["trace"]) == 5 + 20  # warmup + the one timed segment
    for ring_idx in d["trace"]:
        r, s, f, _ = ov.step(d["actions"][ring_idx])
    dense = ov.observe_dense().reshape(len(ids), 32, 32).astype(np.uint64)
    want = (dense << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32)
    assert np.array_equal(d["obs"].view(np.uint32), want)
    assert np.array_equal(f32_bits(d["reward"]), f32_bits(r))
    assert np.array_equal(d["done"], f) and np.array_equal(d["success"], s)


def test_bench_line_survives_a_failed_direct_write_leg():
    """The optional direct-write cadence leg fails on this rank (its hipIpcOpenMemHandle, injected): the line still goes out, complete, with
    the RCCL cadences measured and `direct_write: {error: ...}` naming the phase."""
    res = _run_bench("--force-multi", "--shard", "3/8", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-large-batch",
                     "--no-default-config", "--no-configs", "--no-collector", "--no-dense-obs", "--inject-p2p-open-failure", "0")
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    col = line["config"]["collective"]
    assert col["per_step_gather_us"] > 0 and line["parity"]["bit_exact"] and line["value"] > 1e7
    assert col["direct_write"]["error"]["phase"] == "p2p_open" and col["direct_write"]["error"]["failed_on_this_rank"]
    assert "injected" in col["direct_write"]["error"]["error"]


def test_bench_gpus_flag_fails_loudly_without_that_many_gpus():
    n = torch.cuda.device_count() + 1
    res = _run_bench("--gpus", str(n), "--steps", "20", "--warmup", "5", timeout=120)
    assert res.returncode != 0 and "visible" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')]


# small- and large-batch sampling kernels; world 5: the whole batch (81 920 envs) takes two trips of the large-batch kernel's grid, the
# second one ragged, and the shard is its last fifth
@pytest.mark.parametrize("inverts,per,world", [(False, 256, 2), (False, 16384, 2), (True, 256, 2), (True, 16384, 2), (False, 16384, 5)])
def test_sharded_policy_in_the_loop_draws_what_the_whole_batch_draws(inverts, per, world):
    """The sampling + step call keys its action draw on the GLOBAL env id (qg_vec_set_env_base), like every env-side draw: one seed across
    ranks, and rank r's actions, log-probs and env results are the slice [r * per, (r + 1) * per) of the unsharded batch's."""
    from qiskit_gym_amd.collector import embed, mid_head_sample_step, pack_embedding, pack_head, pack_mid
    from qiskit_gym_amd.vec import VecEnv

    n, r, H1, H2 = 6, world - 1, 512, 256
    gs = line_gateset("clifford", n)
    A = len(gs)
    cfg = dict(add_inverts=inverts, add_perms=False, track_solution=False, difficulty=6)
    whole = VecEnv("clifford", n, gs, per * world, seed=3, **cfg)
    shard = VecEnv("clifford", n, gs, per, env_base=r * per, seed=3, **cfg)
    g = torch.Generator(device="cuda").manual_seed(1)
    w1 = (torch.randn(H1, 4 * n * n, device="cuda", generator=g) * 0.2).to(torch.bfloat16)
    w2 = (torch.randn(H2, H1, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    wh = (torch.randn(A + 1, H2, device="cuda", generator=g) * 0.2).to(torch.bfloat16)
    pm, ph = pack_mid(w2, None), pack_head(wh, None, A, A, after_mid=True)
    sl = slice(r * per, (r + 1) * per)
    out = {}
    for name, env in (("whole", whole), ("shard", shard)):
        env.reset(21)
        pe = pack_embedding(env, w1)
        B = env.batch
        acts = torch.empty(B, dtype=torch.int32, device="cuda")
        logp, ent, val = (torch.empty(B, dtype=torch.float32, device="cuda") for _ in range(3))
        trace = []
        for t in range(4):
            env.set_counters(t, t)
            h = embed(env, pe, None, H1)
            mid_head_sample_step(env, h, pm, H2, ph, 77, t, acts, logp, ent, val, reset_seed=500 + t)
            trace.append((acts.clone(), logp.clone(), env.reward.clone(), env.done.clone()))
        env.sync()
        out[name] = (trace, env.get_state("packed"))
    for (wa, wl, wr, wd), (sa, slp, sr, sd) in zip(out["whole"][0], out["shard"][0]):
        assert torch.equal(wa[sl], sa) and torch.equal(wl[sl].view(torch.int32), slp.view(torch.int32))
        assert torch.equal(wr[sl].view(torch.int32), sr.view(torch.int32)) and torch.equal(wd[sl], sd)
    assert torch.equal(out["whole"][1][sl], out["shard"][1])


def test_bench_as_a_two_rank_job_on_one_gpu_with_the_direct_write_handover():
    """A real N = 2 run of bench.py on the one GPU there is: two ranks under torch.distributed.run (gloo control plane, ids and barriers),
    each stepping its shard of one 131 072-env batch, the learner shard handed over by the direct write into both ranks' hipIpc windows
    inside the timed region; rank 0 replays its own part AND rank 1's part of what arrived on the oracle.  (RCCL refuses two ranks on one
    GPU, so the collective transport itself stays a world-1 test: tests/test_gpu_comm.py.)"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--handover", "direct", "--ranks-share-gpu0", "--steps", "20",
                          "--warmup", "5", "--no-cpu-baseline", "--no-large-batch", "--no-default-config", "--no-configs", "--no-collector"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    cfg = line["config"]
    assert line["ranks"] == 2 and cfg["ranks_seen"] == 2 and cfg["total_envs"] == 2 * 65536 and cfg["env_ids_of_rank0"] == [0, 65536]
    assert line["n_gpus"] == 1 and line["physical_gpus"] == 1 and line["ranks_share_gpu0"] and "ONE GPU" in line["metric"]  # said, not implied
    assert cfg["collective"]["handover"] == "direct" and cfg["collective"]["collectives_in_timed_region"] >= 1
    assert line["parity"]["bit_exact"] and line["parity"]["gathered_shard"]["bit_exact"]
    assert line["parity"]["gathered_shard_of_last_rank"]["bit_exact"] and line["parity"]["gathered_shard_of_last_rank"]["envs"] == 65536 and line["parity"]["envs"] == 65536
    assert line["value"] > 1e7 and line["scaling"] == "weak"  # (two processes share one GPU: a functional run, its rate means nothing)


def test_bench_as_a_four_rank_job_on_one_gpu_every_window_offset_and_flag():
    """The most ranks this pool lets a TEST put on one GPU: six processes may have it open at once -- this test runner, bench.py's launcher and four
    ranks (a run with one more is killed by the pool's process guard; standalone, without the test runner, five ranks fit:
    profiles/r05/bench_five_ranks_one_gpu_direct.json).  bench.py --gpus 4, every rank on GPU 0, 8 192 envs each, the learner shard handed over by the
    direct write -- four windows of 2 parities x 4 slots, four arrival flags per header, every peer offset in use -- inside the timed region; rank 0
    replays its own part and rank 3's part of what arrived on the oracle, every env.  (Config 4's world of 8 needs 8 GPUs:
    tests/test_distributed_cpu.py runs its control plane with 8 ranks.)"""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--handover", "direct", "--ranks-share-gpu0", "--envs", "8192",
                          "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-large-batch", "--no-default-config", "--no-configs", "--no-collector",
                          "--no-dense-obs"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1])
    cfg = line["config"]
    assert line["ranks"] == 4 and cfg["ranks_seen"] == 4 and cfg["total_envs"] == 4 * 8192 and cfg["env_ids_of_rank0"] == [0, 8192]
    assert line["n_gpus"] == 1 and line["physical_gpus"] == 1 and line["ranks_share_gpu0"]
    assert cfg["collective"]["handover"] == "direct" and cfg["collective"]["collectives_in_timed_region"] >= 1
    par = line["parity"]
    assert par["bit_exact"] and par["envs"] == 8192 and par["gathered_shard"]["bit_exact"]
    assert par["gathered_shard_of_last_rank"]["bit_exact"] and par["gathered_shard_of_last_rank"]["envs"] == 8192


def test_bench_two_ranks_with_one_rank_unable_to_map_its_peer_fails_fast_on_every_rank():
    """--handover direct with rank 1's hipIpcOpenMemHandle failing (injected): the vote after the phase stops BOTH ranks there -- the run ends
    within seconds with a non-zero exit code instead of one rank waiting in a barrier for the control plane's timeout."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--handover", "direct", "--ranks-share-gpu0", "--steps", "20",
                          "--warmup", "5", "--no-cpu-baseline", "--no-large-batch", "--no-default-config", "--no-configs", "--no-collector",
                          "--inject-p2p-open-failure", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    assert time.time() - t0 < 120, "a rank waited for a peer that had already failed"
    assert "p2p_open" in res.stderr and "injected" in res.stderr
