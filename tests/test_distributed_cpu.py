"""World-size-2 (and 3) gloo tests of the multi-GPU host logic: shard ranges partition the batch,
each rank's slice of the actions is its own, and the all-gathered packed observation is the
global tensor in env order."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from qiskit_gym_amd.distributed import (OverlappedGather, all_gather_observation, fill_learner_shard, learner_shard_words, local_actions,
                                        shard_range, split_learner_shards, unpack_rows_u32)


def test_shard_ranges_partition_the_batch():
    for total in (1, 7, 64, 65536, 524288, 1000003):
        for world in (1, 2, 3, 8):
            covered = 0
            for r in range(world):
                start, count = shard_range(total, r, world)
                assert start == covered
                covered += count
            assert covered == total
    assert shard_range(524288, 3, 8) == (3 * 65536, 65536)
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_packed(total, d):
    g = torch.Generator().manual_seed(1234)
    return torch.randint(-(2**31), 2**31 - 1, (total, d), dtype=torch.int64, generator=g).to(torch.int32)


def _worker(rank, world, port, per_rank, d, result_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        total = per_rank * world
        full = _global_packed(total, d)
        start, count = shard_range(total, rank, world)
        assert count == per_rank
        local = full[start:start + count].clone()
        gathered = all_gather_observation(local)
        ok = torch.equal(gathered, full)
        acts = torch.arange(3 * total).reshape(3, total)
        mine = local_actions(acts, rank, world)
        ok = ok and mine.shape == (3, per_rank) and int(mine[0, 0]) == start
        # the double-buffered gatherer bench.py uses: snapshot k is gathered when snapshot k + 1 is submitted (or on flush)
        og = OverlappedGather((per_rank, d), torch.int32, "cpu")
        ok = ok and og.latest() is None
        for k in range(5):
            og.submit(lambda buf, k=k: buf.copy_(local + k))
            if k >= 1:
                ok = ok and torch.equal(og.latest(), full + (k - 1))
        og.flush()
        ok = ok and torch.equal(og.latest(), full + 4)
        # the flat hand-over shard bench.py gathers: packed observation + rewards + is_final / success flags in one collective
        g = torch.Generator().manual_seed(99)
        rew_full = torch.randn(total, generator=g)
        done_full = torch.randint(0, 2, (total,), generator=g, dtype=torch.uint8)
        succ_full = torch.randint(0, 2, (total,), generator=g, dtype=torch.uint8)
        sl = slice(start, start + count)
        sg = OverlappedGather((learner_shard_words(per_rank, d),), torch.int32, "cpu")
        sg.submit(lambda buf: fill_learner_shard(buf, per_rank, d, lambda view: view.copy_(local), rew_full[sl], done_full[sl], succ_full[sl]))
        sg.flush()
        obs_g, rew_g, done_g, succ_g = split_learner_shards(sg.latest(), world, per_rank, d)
        ok = ok and torch.equal(obs_g, full) and torch.equal(rew_g, rew_full) and torch.equal(done_g, done_full) and torch.equal(succ_g, succ_full)
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and float(t.item()) == float(world)
        result_q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_all_gather_of_packed_observations_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 97, 32, q)) for r in range(world)]  # 97: the flag sections of the shard need padding
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=10) for _ in range(world))
    assert results == [(r, True) for r in range(world)]


def test_unpack_rows_matches_numpy():
    rng = np.random.default_rng(0)
    dense = rng.integers(0, 2, size=(5, 32, 32)).astype(np.uint64)
    packed = (dense << np.arange(32, dtype=np.uint64)).sum(axis=2).astype(np.uint32).view(np.int32)
    got = unpack_rows_u32(torch.from_numpy(packed.copy()), 32).numpy()
    np.testing.assert_array_equal(got, dense.astype(np.int8))


def test_split_gathered_follows_the_c_layout():
    """`split_gathered` on a hand-built buffer laid out as include/qgym.h's qg_shard_layout says (sections 4-byte aligned, shard padded to
    16 bytes), for a batch that needs every kind of padding and for 1- / 4- / 8-byte observation words; `learner_shard_words` agrees with it."""
    from types import SimpleNamespace

    from qiskit_gym_amd.distributed import split_gathered

    rng = np.random.default_rng(3)
    for batch, words, wb in ((97, 32, 4), (5, 9, 1), (130, 40, 8)):
        world = 3
        obs_bytes = batch * words * wb
        ro = (obs_bytes + 3) // 4 * 4
        fo = ro + 4 * batch
        so = fo + (batch + 3) // 4 * 4
        total = (so + batch + 15) // 16 * 16
        lay = SimpleNamespace(batch=batch, bytes=total, obs_offset=0, obs_bytes=obs_bytes, reward_offset=ro, final_offset=fo, success_offset=so)
        if wb == 4:
            assert learner_shard_words(batch, words) * 4 == total
        dt = {1: np.uint8, 4: np.int32, 8: np.int64}[wb]
        obs = rng.integers(0, 100, size=(world, batch, words)).astype(dt)
        rew = rng.standard_normal((world, batch)).astype(np.float32)
        fin = rng.integers(0, 2, size=(world, batch)).astype(np.uint8)
        suc = rng.integers(0, 2, size=(world, batch)).astype(np.uint8)
        buf = np.zeros((world, total), dtype=np.uint8)
        for r in range(world):
            buf[r, :obs_bytes] = obs[r].view(np.uint8).reshape(-1)
            buf[r, ro:ro + 4 * batch] = rew[r].view(np.uint8)
            buf[r, fo:fo + batch] = fin[r]
            buf[r, so:so + batch] = suc[r]
        o, rw, f, s = split_gathered(torch.from_numpy(buf.reshape(-1)), lay, world, wb)
        assert np.array_equal(o.numpy().view(dt), obs.reshape(world * batch, words))
        assert np.array_equal(rw.numpy(), rew.reshape(-1)) and np.array_equal(f.numpy(), fin.reshape(-1)) and np.array_equal(s.numpy(), suc.reshape(-1))


def _guarded_worker(rank, world, port, scenario, result_q):
    """bench.py's optional N > 1 legs run as voted phases (qiskit_gym_amd.distributed.run_guarded_phases): what every rank sees when one
    rank's phase fails (scenario "open_fails": rank 1 cannot map its peer's window) or a rank disappears (scenario "rank_dies")."""
    import datetime
    import time

    from qiskit_gym_amd.distributed import run_guarded_phases

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=8))
    reached = []

    def export():
        reached.append("export")

    def exchange():
        got = [None] * world
        dist.all_gather_object(got, f"handle-of-{rank}")
        assert got == [f"handle-of-{r}" for r in range(world)]
        reached.append("exchange")

    def open_():
        if scenario == "open_fails" and rank == (5 if world == 8 else 1):
            raise RuntimeError("injected: hipIpcOpenMemHandle failed")
        if scenario == "rank_dies" and rank == 1:
            os._exit(0)  # no vote, no goodbye
        reached.append("open")

    def timed():  # holds a barrier: must never be entered by a subset of the ranks
        dist.barrier()
        reached.append("timed")

    t0 = time.time()
    err = run_guarded_phases([("p2p_export", export), ("handle_exchange", exchange), ("p2p_open", open_), ("timed", timed)])
    elapsed = time.time() - t0
    usable = None
    if err is not None and "control_plane" not in err:  # a voted stop leaves the control plane intact: the run goes on to its final barrier
        dist.barrier()
        usable = True
    result_q.put((rank, err, reached, elapsed, usable))
    if err is None or "control_plane" not in err:
        dist.destroy_process_group()


def _run_guarded(scenario, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guarded_worker, args=(r, world, port, scenario, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = {}
    expect = world - (1 if scenario == "rank_dies" else 0)
    for _ in range(expect):
        r = q.get(timeout=60)
        out[r[0]] = r[1:]
    for p in procs:
        p.join(30)
    return out


def test_a_failed_phase_on_one_rank_stops_every_rank_at_the_same_phase():
    out = _run_guarded("open_fails")
    for rank in (0, 1):
        err, reached, elapsed, usable = out[rank]
        assert err is not None and err["phase"] == "p2p_open" and "timed" not in reached and usable
        assert err["failed_on_this_rank"] == (rank == 1)
        assert elapsed < 5.0, "a rank waited for a peer that had already failed"
    assert "injected" in out[1][0]["error"] and "peer rank failed" in out[0][0]["error"]


def test_world_of_eight_with_rank_five_failing_stops_all_eight_at_the_same_phase():
    """Config 4's shape (8 ranks, one per GPU) on the control plane alone: every rank exchanges its handle with the seven others, rank 5 cannot map
    its peers -- all eight stop after that phase, within seconds, with the control plane still usable for the final barrier."""
    out = _run_guarded("open_fails", world=8)
    assert sorted(out) == list(range(8))
    for rank in range(8):
        err, reached, elapsed, usable = out[rank]
        assert err is not None and err["phase"] == "p2p_open" and "timed" not in reached and usable, rank
        assert reached[:2] == ["export", "exchange"] and err["failed_on_this_rank"] == (rank == 5)
        assert elapsed < 10.0, "a rank waited for a peer that had already failed"
    assert "injected" in out[5][0]["error"] and all("peer rank failed" in out[r][0]["error"] for r in range(8) if r != 5)


def test_world_of_eight_passes_every_phase_when_nothing_fails():
    out = _run_guarded("ok", world=8)
    for rank in range(8):
        err, reached, elapsed, usable = out[rank]
        assert err is None and reached == ["export", "exchange", "open", "timed"], rank


def test_ranks_take_the_gpu_of_their_local_rank():
    """bench.py / the driver's launcher: rank r of an N-rank job on a node that shows V GPUs uses GPU LOCAL_RANK -- N = 2 and N = 4 on an 8-GPU node
    take GPUs 0..N-1, never "whatever is current"; fewer visible GPUs than ranks is an error unless every rank is told to share GPU 0 (a
    functional rehearsal on a one-GPU box)."""
    from qiskit_gym_amd.distributed import device_for_rank

    for world in (1, 2, 4, 8):
        assert [device_for_rank(r, world, visible=8) for r in range(world)] == list(range(world))
    assert [device_for_rank(r, 8, visible=1, share_gpu0=True) for r in range(8)] == [0] * 8
    for local_rank, world, visible in ((1, 2, 1), (4, 8, 4), (7, 8, 7)):
        with pytest.raises(RuntimeError, match="visible"):
            device_for_rank(local_rank, world, visible)
    with pytest.raises(RuntimeError, match="visible"):
        device_for_rank(0, 1, visible=0)
    assert device_for_rank(3, 8, visible=4) == 3  # (a launcher that gave this process LOCAL_RANK 3 on a 4-GPU node: its business)


def test_all_phases_pass_when_nothing_fails():
    out = _run_guarded("ok")
    for rank in (0, 1):
        err, reached, elapsed, usable = out[rank]
        assert err is None and reached == ["export", "exchange", "open", "timed"]


def test_a_vanished_rank_costs_the_control_plane_timeout_not_forever():
    out = _run_guarded("rank_dies")
    err, reached, elapsed, usable = out[0]
    assert err is not None and err["phase"] == "p2p_open" and "control_plane" in err and "timed" not in reached
    assert elapsed < 30.0
