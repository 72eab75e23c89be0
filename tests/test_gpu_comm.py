"""The multi-GPU hand-over behind the C ABI (include/qgym.h "Multi-GPU hand-over") as far as one GPU can show it:
RCCL's all-gather called from libqgym (world 1), the direct-write transport on one rank and on TWO ranks -- two processes that
share this GPU and map each other's windows through hipIpc -- and a plain C host.  What arrives is compared with the CPU oracle
(Env::observe / reward / is_final / success of clifford.rs:353-368 for the same env ids)."""
import os
import shutil
import subprocess
import time

import numpy as np
import pytest
import torch

from oracle import OracleEnv, OracleVec
from util import f32_bits, line_gateset, oracle_cfg, rng_actions

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [
    ("clifford", 16, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=24), 4),
    ("clifford", 20, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=24), 8),
    ("linear_function", 8, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=12), 4),
    ("permutation", 9, dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=8), 1),
]


def _pack_rows(dense, word_bytes):
    """dense [n, rows, cols] {0,1} -> [n, rows] words, bit c of word r = entry (r, c) (QG_FMT_PACKED)."""
    w = (dense.astype(np.uint64) << np.arange(dense.shape[2], dtype=np.uint64)).sum(axis=2)
    return w.astype({4: np.uint32, 8: np.uint64}[word_bytes])


def _oracle_shard(kind, n, gs, cfg, env_ids, seed, actions):
    """(packed observation words, reward, is_final, success) of envs `env_ids` after reset(seed) and the given steps."""
    proto = OracleEnv(kind, n, gs, **oracle_cfg(cfg))
    ov = OracleVec(proto, len(env_ids))
    ov.reset_with(rng_actions(seed, env_ids, cfg["difficulty"], len(gs)))
    r = s = f = None
    for a in actions:
        r, s, f, _ = ov.step(a)
    dense = ov.observe_dense()
    if kind == "permutation":
        obs = dense.reshape(len(env_ids), n, n).argmax(axis=2).astype(np.uint8)  # one byte per entry: the set column
    else:
        rows = 2 * n if kind == "clifford" else n
        obs = _pack_rows(dense.reshape(len(env_ids), rows, rows), 8 if rows > 32 else 4)
    return obs, r, f, s


def _check_gathered(gathered, env, world, word_bytes, want):
    from qiskit_gym_amd.distributed import split_gathered

    obs, rew, fin, suc = split_gathered(gathered, env.shard_layout(), world, word_bytes)
    dt = {1: np.uint8, 4: np.uint32, 8: np.uint64}[word_bytes]
    assert np.array_equal(obs.cpu().numpy().view(dt), want[0])
    assert np.array_equal(f32_bits(rew.cpu().numpy()), f32_bits(want[1]))
    assert np.array_equal(fin.cpu().numpy(), want[2]) and np.array_equal(suc.cpu().numpy(), want[3])


@pytest.mark.parametrize("kind,n,cfg,word_bytes", CASES)
def test_shard_layout_and_rccl_all_gather_world_1_match_oracle(kind, n, cfg, word_bytes):
    from qiskit_gym_amd.distributed import Communicator
    from qiskit_gym_amd.vec import VecEnv

    gs = line_gateset(kind, n)
    B, base, seed = 777, 5 * 777, 41  # odd batch: every section of the shard needs padding; rank 5 of a larger job
    env = VecEnv(kind, n, gs, B, env_base=base, **cfg)
    lay = env.shard_layout()
    assert lay.batch == B and lay.obs_offset == 0 and lay.obs_bytes == B * env.packed_words_per_env * env.packed_word_bytes
    assert lay.reward_offset % 4 == 0 and lay.final_offset == lay.reward_offset + 4 * B and lay.bytes % 16 == 0
    assert lay.success_offset >= lay.final_offset + B and lay.bytes >= lay.success_offset + B
    env.reset(seed)
    g = torch.Generator(device="cuda").manual_seed(3)
    acts = torch.randint(0, len(gs), (6, B), dtype=torch.int32, device="cuda", generator=g)
    for t in range(6):
        env.step(acts[t])
    want = _oracle_shard(kind, n, gs, cfg, np.arange(base, base + B), seed, acts.cpu().numpy())
    comm = Communicator(0, 1, unique_id=Communicator.unique_id())
    gathered = comm.gather(env)  # pack + ncclAllGather on the current stream
    env.sync()
    _check_gathered(gathered, env, 1, word_bytes, want)
    # the same shard without a communicator
    assert torch.equal(env.pack_learner_shard(), gathered)
    # overlapped form: snapshot k is gathered when k + 1 is submitted (or on flush)
    assert comm.latest() is None
    comm.submit(env)
    env.step(acts[0])
    comm.submit(env)
    _check_gathered(comm.latest().clone(), env, 1, word_bytes, want)
    want2 = _oracle_shard(kind, n, gs, cfg, np.arange(base, base + B), seed, np.concatenate([acts.cpu().numpy(), acts[:1].cpu().numpy()]))
    comm.flush()
    _check_gathered(comm.latest().clone(), env, 1, word_bytes, want2)
    comm.close()


def test_pauli_shard_is_one_observe_per_pack():
    """PauliEnv's packed observation counts as an observe() (it draws the add_perms permutation): the shard holds that observation."""
    from qiskit_gym_amd.distributed import split_gathered
    from qiskit_gym_amd.vec import VecEnv

    n, B = 6, 256
    gs = line_gateset("pauli", n)
    cfg = dict(add_perms=True, track_solution=False, difficulty=40, pauli_diff_scale=8)
    a, b = VecEnv("pauli", n, gs, B, seed=9, **cfg), VecEnv("pauli", n, gs, B, seed=9, **cfg)
    a.reset(5)
    b.reset(5)
    shard = a.pack_learner_shard()
    obs, rew, fin, suc = split_gathered(shard, a.shard_layout(), 1, 8)
    assert torch.equal(obs, b.observe_packed().view(B, -1))
    assert torch.equal(rew, b.reward) and torch.equal(fin, b.done) and torch.equal(suc, b.success)


def test_direct_write_world_1_epochs_match_oracle():
    from qiskit_gym_amd.distributed import Communicator
    from qiskit_gym_amd.vec import VecEnv

    kind, n, cfg, wb = CASES[0]
    gs = line_gateset(kind, n)
    B, seed = 4096 + 64, 77
    env = VecEnv(kind, n, gs, B, **cfg)
    env.reset(seed)
    comm = Communicator(0, 1, local=True)
    comm.p2p_connect(int(env.shard_layout().bytes))
    g = torch.Generator(device="cuda").manual_seed(4)
    acts = torch.randint(0, len(gs), (7, B), dtype=torch.int32, device="cuda", generator=g)
    for t in range(7):  # seven epochs: both parities, and the release / overwrite protocol from epoch 3 on
        env.step(acts[t])
        comm.push(env)
        view = comm.wait()
        got = view.clone()
        comm.release()
        if t in (0, 1, 6):
            _check_gathered(got, env, 1, wb, _oracle_shard(kind, n, gs, cfg, np.arange(B), seed, acts[: t + 1].cpu().numpy()))
    comm.check()
    comm.close()


def _p2p_rank(rank, world, B, conn, result_q):
    """One of `world` processes on the SAME GPU: steps its shard, pushes it into every window, checks what arrived from everybody."""
    import torch as th

    from qiskit_gym_amd.distributed import Communicator, split_gathered
    from qiskit_gym_amd.vec import VecEnv

    try:
        th.cuda.set_device(0)
        kind, n, cfg, wb = CASES[0]
        gs = line_gateset(kind, n)
        seed, T = 123, 6
        shard = VecEnv(kind, n, gs, B, env_base=rank * B, **cfg)
        whole = VecEnv(kind, n, gs, B * world, **cfg)  # what every shard must look like, stepped locally
        comm = Communicator(rank, world, local=True)
        mine = comm.p2p_export(int(shard.shard_layout().bytes))
        conn.send(mine)
        handles = conn.recv()  # all handles, rank order
        comm.p2p_open(handles)
        conn.send("opened")
        assert conn.recv() == "go"
        shard.reset(seed)
        whole.reset(seed)
        g = th.Generator(device="cuda").manual_seed(8)
        acts = th.randint(0, len(gs), (T, B * world), dtype=th.int32, device="cuda", generator=g)
        ok = True
        for t in range(T):
            whole.step(acts[t])
            shard.step(acts[t, rank * B:(rank + 1) * B].contiguous())
            comm.push(shard)
            if rank == 1 and t == 2:
                th.cuda.synchronize()
                time.sleep(0.3)  # one rank falls behind: the other's wait (and, two epochs on, its push) really waits
            view = comm.wait()
            obs, rew, fin, suc = split_gathered(view.clone(), shard.shard_layout(), world, wb)
            comm.release()
            ok = ok and th.equal(obs, whole.observe_packed().view(B * world, -1))
            ok = ok and th.equal(rew.view(th.int32), whole.reward.view(th.int32)) and th.equal(fin, whole.done) and th.equal(suc, whole.success)
        comm.check()
        shard.sync()
        conn.send("stepped")
        conn.recv()  # both ranks are through their T epochs
        if rank == 0:
            # a peer that never writes epoch T + 1: the wait gives up at its deadline (2 s of the device's wall clock), the stream drains, and
            # the check names the cause -- a dead rank becomes an error on the host, not a hung GPU
            comm.push(shard)
            t0 = time.time()
            comm.wait()
            try:
                comm.check()
                ok = False
            except RuntimeError as exc:
                ok = ok and "did not arrive" in str(exc) and 1.5 < time.time() - t0 < 20.0
        conn.send("done")
        conn.recv()  # nobody unmaps a window a peer may still write
        comm.close()
        result_q.put((rank, bool(ok), ""))
    except Exception as exc:  # noqa: BLE001
        result_q.put((rank, False, repr(exc)))


@pytest.mark.parametrize("world", [2, 4])  # four ranks: every (rank + y) % world peer index, four arrival flags and releases per window
def test_direct_write_ranks_sharing_this_gpu(world):
    """`world` processes, each a rank with its own shard, map each other's window (hipIpcGetMemHandle / hipIpcOpenMemHandle) and write
    their shards into all of them; every rank must read every shard, epoch after epoch, with no host synchronisation between the ranks
    inside the loop (arrival flags and releases are the only ordering).  Then rank 1 stops: rank 0's next wait must end at its deadline
    with an error the host can read."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    B = 2048
    q = ctx.Queue()
    pipes = [ctx.Pipe() for _ in range(world)]
    procs = [ctx.Process(target=_p2p_rank, args=(r, world, B, pipes[r][1], q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        def recv_all(what=None):
            out = []
            for r in range(world):
                assert pipes[r][0].poll(180), f"rank {r} did not answer"
                out.append(pipes[r][0].recv())
                if what is not None:
                    assert out[-1] == what
            return out

        handles = recv_all()
        for r in range(world):
            pipes[r][0].send(handles)
        recv_all("opened")
        for r in range(world):
            pipes[r][0].send("go")
        recv_all("stepped")
        for r in range(world):
            pipes[r][0].send("rank 1 stays silent for one epoch")
        recv_all("done")
        for r in range(world):
            pipes[r][0].send("bye")
        results = sorted(q.get(timeout=60) for _ in range(world))
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    assert results == [(r, True, "") for r in range(world)], results


C_HOST = r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "qgym.h"
#define CHECK(x) do { int rc_ = (x); if (rc_ != QG_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, qg_last_error()); return 10; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 11; } } while (0)
enum { NQ = 16, B = 1000, T = 5 };
int main(void) {
    static qg_gate gates[512];
    size_t n_gates = 0;
    int k, q;
    /* line-16 bidirectional, H S Sdg SX SXdg CX CZ SWAP in from_coupling_map order (envs/synthesis.py:89-103) */
    for (k = 0; k < 5; ++k) for (q = 0; q < NQ; ++q) { gates[n_gates].kind = k; gates[n_gates].q0 = q; gates[n_gates].q1 = 0; ++n_gates; }
    for (k = 5; k < 8; ++k) for (q = 0; q + 1 < NQ; ++q) {
        gates[n_gates].kind = k; gates[n_gates].q0 = q; gates[n_gates].q1 = q + 1; ++n_gates;
        gates[n_gates].kind = k; gates[n_gates].q0 = q + 1; gates[n_gates].q1 = q; ++n_gates;
    }
    qg_config cfg;
    qg_vec *v = NULL;
    qg_comm *c = NULL;
    qg_shard_layout lay;
    uint8_t id[QG_COMM_ID_BYTES];
    const void *win = NULL;
    qg_config_default(&cfg, QG_CLIFFORD, NQ);
    cfg.add_inverts = 0; cfg.add_perms = 0; cfg.track_solution = 0; cfg.difficulty = 20;
    CHECK(qg_vec_create(&cfg, gates, n_gates, B, 0, &v));
    CHECK(qg_vec_set_env_base(v, 3 * B));
    CHECK(qg_vec_learner_shard_layout(v, &lay));
    CHECK(qg_comm_unique_id(id));
    CHECK(qg_comm_init(id, 0, 1, 0, &c));
    if (qg_comm_rank(c) != 0 || qg_comm_world(c) != 1) return 12;
    int32_t *act_h = (int32_t *)malloc(sizeof(int32_t) * B), *act_d = NULL;
    uint8_t *gath_d = NULL, *gath_h = (uint8_t *)malloc(lay.bytes), *p2p_h = (uint8_t *)malloc(lay.bytes);
    HIP(hipMalloc((void **)&act_d, sizeof(int32_t) * B));
    HIP(hipMalloc((void **)&gath_d, lay.bytes));
    CHECK(qg_vec_reset(v, 99, NULL));
    for (k = 0; k < T; ++k) {
        for (q = 0; q < B; ++q) act_h[q] = (int32_t)((q * 7 + k * 13) % (int)n_gates);
        HIP(hipMemcpy(act_d, act_h, sizeof(int32_t) * B, hipMemcpyHostToDevice));
        CHECK(qg_vec_step(v, act_d, QG_ACT_I32, NULL, NULL));
    }
    CHECK(qg_vec_gather_learner_shard(v, c, gath_d, NULL));   /* RCCL */
    CHECK(qg_comm_p2p_connect(c, lay.bytes));                  /* direct write on the same communicator */
    CHECK(qg_vec_push_learner_shard(v, c, NULL));
    CHECK(qg_comm_p2p_wait(c, &win, NULL));
    CHECK(qg_comm_p2p_check(c, NULL));
    CHECK(qg_vec_sync(v, NULL));
    HIP(hipMemcpy(gath_h, gath_d, lay.bytes, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(p2p_h, win, lay.bytes, hipMemcpyDeviceToHost));
    CHECK(qg_comm_p2p_release(c, NULL));
    if (memcmp(gath_h, p2p_h, lay.bytes)) { fprintf(stderr, "transports disagree\n"); return 13; }
    printf("layout %llu %llu %llu %llu %llu\n", (unsigned long long)lay.bytes, (unsigned long long)lay.obs_bytes, (unsigned long long)lay.reward_offset,
           (unsigned long long)lay.final_offset, (unsigned long long)lay.success_offset);
    for (q = 0; q < B; ++q) {
        uint32_t rb;
        memcpy(&rb, gath_h + lay.reward_offset + 4 * q, 4);
        printf("env %u %u %u", rb, gath_h[lay.final_offset + q], gath_h[lay.success_offset + q]);
        for (k = 0; k < 2 * NQ; ++k) { uint32_t w; memcpy(&w, gath_h + 4 * ((size_t)q * 2 * NQ + k), 4); printf(" %u", w); }
        printf("\n");
    }
    qg_comm_destroy(c);
    qg_vec_destroy(v);
    return 0;
}
'''


def test_c_host_gathers_its_shard_over_both_transports(tmp_path):
    """What a Rust / C host would link: create a batch, step it, all-gather the learner shard through qg_comm (RCCL), then through
    the direct-write window; no Python, no torch in that process.  Every env's line is compared with the oracle."""
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    (tmp_path / "host.c").write_text(C_HOST)
    exe = tmp_path / "host"
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "qiskit_gym_amd", "lib")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", inc, "-I", "/opt/rocm/include", str(tmp_path / "host.c"), "-o", str(exe),
                    "-L", libdir, "-lqgym", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith(("layout ", "env "))]  # RCCL prints a version banner on stdout
    NQ, B, T = 16, 1000, 5
    gs = line_gateset("clifford", NQ)
    cfg = dict(add_inverts=False, add_perms=False, track_solution=False, difficulty=20)
    acts = np.array([[(q * 7 + k * 13) % len(gs) for q in range(B)] for k in range(T)])
    obs, r, f, s = _oracle_shard("clifford", NQ, gs, cfg, np.arange(3 * B, 4 * B), 99, acts)
    assert lines[0].split()[0] == "layout" and int(lines[0].split()[2]) == B * 2 * NQ * 4
    assert len(lines) == B + 1
    for q in range(B):
        tok = lines[1 + q].split()
        assert int(tok[1]) == int(f32_bits(r)[q]) and int(tok[2]) == int(f[q]) and int(tok[3]) == int(s[q])
        assert [int(x) for x in tok[4:]] == [int(x) for x in obs[q]]
