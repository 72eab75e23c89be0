// qgym_env.cpp -- scalar `Env`-trait flavour of the C ABI (qg_env_*) and the "twists".
//
// A qg_env is a batch of one on the same HIP kernels (no second implementation of the env);
// each call synchronises, so this flavour is for API parity and tests, not for throughput.
//
// Twists (coupling-graph symmetries handed to the learner for data augmentation) are
// constructor-time host data in the reference: rust/src/envs/symmetry.rs:115-361.  They are
// restated here: automorphisms of the undirected coupling graph (sorted + de-duplicated, so the
// enumeration order of petgraph's VF2 is unobservable; Heap's-algorithm order when the gateset
// has no two-qubit gate, symmetry.rs:84-113,121-123), the induced action permutation
// (symmetry.rs:178-203) and observation index permutation (symmetry.rs:265-295).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "qgym_host.hpp"

using namespace qg;

namespace {

typedef std::vector<int64_t> Perm;

void heap_permute(size_t k, Perm &perm, std::vector<Perm> &results) {  // symmetry.rs:88-104
    if (k == 1) {
        results.push_back(perm);
        return;
    }
    heap_permute(k - 1, perm, results);
    for (size_t i = 0; i < k - 1; ++i) {
        if (k % 2 == 0) std::swap(perm[i], perm[k - 1]);
        else std::swap(perm[0], perm[k - 1]);
        heap_permute(k - 1, perm, results);
    }
}

void automorphism_search(const std::vector<std::vector<uint8_t>> &adj, Perm &map, std::vector<uint8_t> &used, size_t pos,
                         std::vector<Perm> &out) {
    const size_t n = adj.size();
    if (pos == n) {
        out.push_back(map);
        return;
    }
    for (size_t cand = 0; cand < n; ++cand) {
        if (used[cand]) continue;
        bool ok = true;
        for (size_t prev = 0; prev < pos && ok; ++prev) ok = adj[pos][prev] == adj[cand][(size_t)map[prev]];
        if (!ok) continue;
        used[cand] = 1;
        map[pos] = (int64_t)cand;
        automorphism_search(adj, map, used, pos + 1, out);
        used[cand] = 0;
    }
}

std::vector<Perm> compute_automorphisms(const std::vector<std::vector<uint8_t>> &adj, bool has_edge) {  // symmetry.rs:115-176
    const size_t n = adj.size();
    std::vector<Perm> results;
    if (n == 0) {
        results.push_back(Perm());
        return results;
    }
    if (!has_edge) {
        Perm p(n);
        for (size_t i = 0; i < n; ++i) p[i] = (int64_t)i;
        heap_permute(n, p, results);
        return results;
    }
    Perm map(n, -1);
    std::vector<uint8_t> used(n, 0);
    automorphism_search(adj, map, used, 0, results);
    if (results.empty()) {
        Perm id(n);
        for (size_t i = 0; i < n; ++i) id[i] = (int64_t)i;
        results.push_back(id);
    }
    std::sort(results.begin(), results.end());
    results.erase(std::unique(results.begin(), results.end()), results.end());
    return results;
}

typedef std::pair<int, std::vector<int64_t>> GateKey;
GateKey canonical_key(int kind, std::vector<int64_t> q) {  // symmetry.rs:66-71
    if (kind == QG_SWAP) std::sort(q.begin(), q.end());
    return GateKey(kind, q);
}
std::vector<int64_t> gate_qubits(const qg_gate &g) {
    if (g.kind >= QG_CX) return {g.q0, g.q1};
    return {g.q0};
}

bool build_action_perm(const std::vector<qg_gate> &gates, const std::map<GateKey, int64_t> &index, const Perm &perm, Perm &out) {
    out.clear();
    for (const qg_gate &g : gates) {  // symmetry.rs:185-200
        std::vector<int64_t> q = gate_qubits(g);
        for (auto &x : q) {
            if ((size_t)x >= perm.size()) return false;
            x = perm[(size_t)x];
        }
        auto it = index.find(canonical_key(g.kind, q));
        if (it == index.end()) return false;
        out.push_back(it->second);
    }
    return true;
}

}  // namespace

namespace qg {

// (qubit perms, act perms) of symmetry.rs:205-263 / 307-361
void compute_qubit_and_action_perms(uint32_t N, const std::vector<qg_gate> &gates, std::vector<Perm> &qubit_perms,
                                    std::vector<Perm> &act_perms) {
    qubit_perms.clear();
    act_perms.clear();
    if (N == 0) return;
    std::map<GateKey, int64_t> index;
    for (size_t i = 0; i < gates.size(); ++i) index[canonical_key(gates[i].kind, gate_qubits(gates[i]))] = (int64_t)i;  // later wins
    std::vector<std::vector<uint8_t>> adj(N, std::vector<uint8_t>(N, 0));
    bool has_edge = false;
    for (const qg_gate &g : gates)
        if (g.kind >= QG_CX && g.q0 != g.q1) {
            adj[g.q0][g.q1] = adj[g.q1][g.q0] = 1;
            has_edge = true;
        }
    std::set<Perm> seen;
    for (const Perm &p : compute_automorphisms(adj, has_edge)) {
        if (!seen.insert(p).second) continue;
        Perm ap;
        if (build_action_perm(gates, index, p, ap)) {
            qubit_perms.push_back(p);
            act_perms.push_back(ap);
        }
    }
    if (qubit_perms.empty()) {
        Perm id(N);
        for (uint32_t i = 0; i < N; ++i) id[i] = i;
        Perm ap;
        if (build_action_perm(gates, index, id, ap)) {
            qubit_perms.push_back(id);
            act_perms.push_back(ap);
        }
    }
}

}  // namespace qg

// ---- scalar env ---------------------------------------------------------------------------------------------------------------
// What a call costs is what crosses the host / device boundary, so a qg_env keeps that to one kernel launch and one stream
// synchronisation per mutating call, and nothing at all per getter:
//   * its own non-blocking stream: envs stepped from different host threads (twisterl's rayon workers) do not serialise on the null stream;
//   * one pinned, device-mapped I/O block: the action and the coin are written there by the host and read by the step kernel in place;
//     reward / is_final / success are the handle's output arrays (qg_vec_bind_outputs) and live there too, so the kernel's results are in
//     host memory when the stream has drained and reward() / is_final() / success() / masks() are plain loads; the dense observation of
//     the new state is written behind the I/O block by the same call (env_sync), so the loop's observe() is a scan of host memory;
//   * clone() takes a finished env's handle from a pool when one with the same constructor arguments exists (twisterl clones the
//     prototype once per episode): no allocation, no constructor launch -- the clone is a handful of device copies.
struct EnvIO {  // pinned host memory, device-mapped
    int64_t action;
    uint8_t coin;
    uint8_t pad0[7];
    float reward;
    uint8_t done, success, pad1[2];
    int32_t depth;
    uint32_t error;
};

struct qg_env {
    qg_vec *v = nullptr;
    hipStream_t st = nullptr;
    EnvIO *io = nullptr;      // host address
    EnvIO *io_dev = nullptr;  // the same block as the device sees it
    int8_t *obs = nullptr, *obs_dev = nullptr;  // the dense observation, in the same pinned allocation behind the I/O block
    size_t obs_bytes = 0;
    bool obs_ahead = false;  // observe() is a pure function of the state (every env but PauliEnv with add_perms, whose observe() draws a
                             // permutation per call, pauli.rs:653-665): the call that changes the state writes the next observation too
    bool obs_valid = false;  // the pinned buffer holds the observation of the current state
    bool tracked = false;    // qg_vec_track_dense on the pinned buffer: every call that changes the state keeps it current itself (the step kernel
                             // rewrites the rows its gate changed, over PCIe) -- no observe launch per step
    bool twists_done = false;
    std::vector<Perm> obs_perms, act_perms;
};

namespace {
std::mutex g_pool_mutex;
std::vector<qg_env *> g_pool;            // destroyed envs kept for the next clone
constexpr size_t POOL_CAP = 256;         // a few per host thread (rl/configs.py:135 num_cores = 32); beyond it destroy() frees
constexpr size_t POOL_CAP_PER_KEY = 64;  // ... and at most this many with the same constructor arguments (a trainer that destroys thousands of
                                         // clones of one prototype does not push every other configuration's handles out, nor pin 256 of its own)

bool same_ctor(const qg_vec *a, const qg_vec *b) {
    return a->device == b->device && memcmp(&a->cfg, &b->cfg, sizeof a->cfg) == 0 && a->gates.size() == b->gates.size() &&
           (a->gates.empty() || memcmp(a->gates.data(), b->gates.data(), a->gates.size() * sizeof(qg_gate)) == 0);
}

void env_free(qg_env *e) {
    if (!e) return;
    if (e->v) {
        qg::DeviceGuard guard(e->v->device);
        if (e->st) (void)hipStreamSynchronize(e->st);
        qg_vec_destroy(e->v);  // the output arrays are bound to the I/O block, which is released below
        if (e->st) (void)hipStreamDestroy(e->st);
        if (e->io) (void)hipHostFree(e->io);  // the observation buffer lives in the same allocation
    }
    delete e;
}

// stream drained; a fault the reference panics on becomes QG_ERR_PANIC (qg_vec_sync's rule, from the copy of the error word in the I/O block).
// Called at the end of every call that changes the state: the next observe() is enqueued behind it first, so that the collection loop's
// observe() / step() pair costs one stream synchronisation, not two.
int env_sync(qg_env *e) {
    e->obs_valid = false;
    if (e->tracked) {
        e->obs_valid = true;
    } else if (e->obs_ahead) {
        if (int rc = qg_vec_observe_dense(e->v, e->obs_dev, e->st)) return rc;
        e->obs_valid = true;
    }
    // (the fault word lives in the I/O block: the kernels write it there, nothing to copy)
    HIP_TRY(hipStreamSynchronize(e->st));
    const uint32_t err = e->io->error;
    if (err) {
        const char *what = (err & QG_FAULT_SINGULAR)      ? "singular matrix in inverse() (reference panics, clifford.rs:155)"
                           : (err & QG_FAULT_ZERO_WEIGHT) ? "weight-0 rotation in the front layer (reference panics, pauli_network.rs:114)"
                           : (err & QG_FAULT_BAD_STATE)   ? "set_state produced an unusable state"
                                                          : "solution log overflow (QG_FAULT_SOLUTION_OVERFLOW)";
        return set_error(QG_ERR_PANIC, "env 0: %s (fault bits 0x%x)", what, err);
    }
    return QG_OK;
}
}  // namespace

extern "C" {

int qg_env_create(const qg_config *cfg, const qg_gate *gates, size_t n_gates, int device, qg_env **out) {
    if (!out) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    qg_vec *v = nullptr;
    int rc = qg_vec_create(cfg, gates, n_gates, 1, device, &v);
    if (rc) return rc;
    qg_env *e = new qg_env();
    e->v = v;
    auto fail = [&](int code) {
        env_free(e);
        return code;
    };
    qg::DeviceGuard guard(v->device);
    if (guard.err != hipSuccess) {
        (void)hipGetLastError();
        return fail(set_error(QG_ERR_DEVICE, "cannot select device %d: %s", v->device, hipGetErrorString(guard.err)));
    }
#define HIP_TRY_E(expr)                                                                                          \
    do {                                                                                                         \
        hipError_t _e = (expr);                                                                                  \
        if (_e != hipSuccess) {                                                                                  \
            (void)hipGetLastError();                                                                             \
            return fail(set_error(QG_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)));                \
        }                                                                                                        \
    } while (0)
    HIP_TRY_E(hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking));
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    e->obs_bytes = (size_t)info.obs_rows * info.obs_cols;
    static_assert(sizeof(EnvIO) <= 64, "the observation starts 64 bytes into the pinned block");
    HIP_TRY_E(hipHostMalloc((void **)&e->io, 64 + e->obs_bytes, hipHostMallocMapped));  // one pinned allocation per env
    memset(e->io, 0, 64 + e->obs_bytes);
    HIP_TRY_E(hipHostGetDevicePointer((void **)&e->io_dev, e->io, 0));
    e->obs = reinterpret_cast<int8_t *>(e->io) + 64;
    e->obs_dev = reinterpret_cast<int8_t *>(e->io_dev) + 64;
#undef HIP_TRY_E
    e->obs_ahead = !(v->layout == LAYOUT_PAULI && v->n_perms > 0);
    // the handle's per-env outputs live in the I/O block from here on (their constructor values are carried over)
    // (the remaining depth stays in device memory: the step kernel reads and writes it, the trait has no getter for it)
    rc = qg_vec_bind_outputs(v, &e->io_dev->reward, &e->io_dev->done, &e->io_dev->success, nullptr);
    if (rc) return fail(rc);
    if ((rc = qg::bind_error(v, &e->io_dev->error))) return fail(rc);
    // a step of the scalar env was two launches, a 4-byte copy and a stream synchronisation; where the layout can keep a resident dense
    // observation (16- / 32-row matrices) the step kernel rewrites the pinned observation's changed rows itself: one launch and the synchronisation
    if (e->obs_ahead && qg_vec_track_dense(v, e->obs_dev, e->st) == QG_OK) {
        e->tracked = true;
        e->obs_valid = true;
        if (hipStreamSynchronize(e->st) != hipSuccess) {
            (void)hipGetLastError();
            return fail(set_error(QG_ERR_DEVICE, "scalar env: first observation failed"));
        }
    }
    *out = e;
    return QG_OK;
}

void qg_env_destroy(qg_env *e) {
    if (!e) return;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t same = 0;
        for (const qg_env *p : g_pool) same += same_ctor(p->v, e->v) ? 1 : 0;
        if (g_pool.size() < POOL_CAP && same < POOL_CAP_PER_KEY) {  // kept for the next clone of an env built with the same arguments
            g_pool.push_back(e);
            return;
        }
    }
    env_free(e);
}

void qg_env_pool_clear(void) {
    std::vector<qg_env *> drained;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        drained.swap(g_pool);
    }
    for (qg_env *e : drained) env_free(e);
}

int qg_env_clone(const qg_env *e, qg_env **out) {  // Env: DynClone -- deep copy of every resident buffer
    if (!e || !out) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    qg_env *c = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = g_pool.size(); i-- > 0;)
            if (same_ctor(g_pool[i]->v, e->v)) {
                c = g_pool[i];
                g_pool.erase(g_pool.begin() + (ptrdiff_t)i);
                break;
            }
    }
    if (!c) {
        int rc = qg_env_create(&e->v->cfg, e->v->gates.data(), e->v->gates.size(), e->v->device, &c);
        if (rc) return rc;
    }
    const qg_vec *s = e->v;
    qg_vec *d = c->v;
    QG_ON_DEVICE(d);
    d->difficulty = s->difficulty;
    d->step_index = s->step_index;
    d->coin_seed = s->coin_seed;
    d->maybe_nonsymplectic = s->maybe_nonsymplectic;
    d->observe_counter = s->observe_counter;
    d->env_base = s->env_base;
    // a handle from the pool carries its previous owner's host-side session state: everything a fresh handle starts with, it starts with
    d->auto_list = d->done_list_fresh = d->mask_fresh = false;
    d->list_zero_known = false;  // (unknown is always safe: the next appending launch zeroes the length first)
    d->list_tainted = false;
    d->list_session = 0;
    d->perm_draw = false;
    d->perm_in = nullptr;
    d->clock_dev = s->clock_dev;
    d->dense = c->tracked ? c->obs_dev : nullptr;  // (the clone's own pinned observation, made current below)
    for (auto &g : d->graphs) {  // cached rollout graphs have the previous owner's pointers and counters baked in
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    d->graphs.clear();
    struct { void *dst; const void *src; size_t bytes; } copies[] = {
        {d->state, s->state, s->state_bytes},
        {d->depth, s->depth, 4},
        {d->inverted, s->inverted, 1},
        {d->sol_len, s->sol_len, 8},
        {d->sol, s->sol, (size_t)s->sol_cap * 4},
        {d->layers, s->layers, (size_t)s->layers_len * 4 * 64},  // the env's tile (layer_rec)
        {d->bad, s->bad, s->layout == LAYOUT_LFD ? (size_t)16 : s->layout == LAYOUT_TILE64 ? (size_t)8 : (size_t)4},  // incremental solved masks
        {d->perm_idx, s->perm_idx, (size_t)4},                                 // PauliEnv current_perm_idx (pauli.rs:661)
    };
    for (auto &cp : copies)
        if (cp.dst && cp.src && cp.bytes && hipMemcpyAsync(cp.dst, cp.src, cp.bytes, hipMemcpyDeviceToDevice, c->st) != hipSuccess) {
            (void)hipGetLastError();
            env_free(c);
            return set_error(QG_ERR_DEVICE, "clone copy failed");
        }
    // reward / is_final / success / depth: host memory on both sides (the source is idle: every call on it has synchronised)
    c->io->reward = e->io->reward;
    c->io->done = e->io->done;
    c->io->success = e->io->success;
    c->io->error = e->io->error;
    c->obs_valid = false;
    if (c->obs_ahead && e->obs_valid) {  // the source's observation is the clone's
        memcpy(c->obs, e->obs, e->obs_bytes);
        c->obs_valid = true;
    } else if (c->tracked) {  // (a source nobody has observed: rewrite the clone's tracked observation from the copied state)
        if (qg::dense_refresh_public(d, c->st) != QG_OK) {
            env_free(c);
            return set_error(QG_ERR_DEVICE, "clone: observation refresh failed");
        }
        c->obs_valid = true;
    }
    c->twists_done = false;
    c->obs_perms.clear();
    c->act_perms.clear();
    if (hipStreamSynchronize(c->st) != hipSuccess) {
        (void)hipGetLastError();
        env_free(c);
        return set_error(QG_ERR_DEVICE, "clone copy failed");
    }
    *out = c;
    return QG_OK;
}

int qg_env_set_seed(qg_env *e, uint64_t seed) { return e ? qg_vec_set_seed(e->v, seed) : set_error(QG_ERR_INVALID, "null argument"); }

int64_t qg_env_num_actions(const qg_env *e) { return e ? (int64_t)e->v->gates.size() : -1; }

int qg_env_obs_shape(const qg_env *e, int64_t out[2]) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    qg_vec_info info;
    qg_vec_get_info(e->v, &info);
    out[0] = info.obs_rows;
    out[1] = info.obs_cols;
    return QG_OK;
}

int qg_env_set_difficulty(qg_env *e, int64_t d) { return e ? qg_vec_set_difficulty(e->v, d) : set_error(QG_ERR_INVALID, "null"); }
int64_t qg_env_get_difficulty(const qg_env *e) { return e ? qg_vec_get_difficulty(e->v) : -1; }

int qg_env_set_state(qg_env *e, const int64_t *state, size_t n) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    if (n == 0 && e->v->layout == LAYOUT_PAULI) return QG_OK;  // pauli.rs:518-520
    if (!state) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(e->v);
    int rc = qg_vec_set_state(e->v, state, QG_FMT_I64, n, 0, e->st);
    if (rc) return rc;
    return env_sync(e);
}

int qg_env_reset(qg_env *e, uint64_t seed) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(e->v);
    int rc = qg_vec_reset(e->v, seed, e->st);
    if (rc) return rc;
    return env_sync(e);
}

static int env_step(qg_env *e, int64_t action, const uint8_t *coin) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    QG_ON_DEVICE(e->v);
    e->io->action = action;  // read in place by the step kernel
    if (coin) e->io->coin = *coin;
    int rc = qg_vec_step(e->v, &e->io_dev->action, QG_ACT_I64, coin ? &e->io_dev->coin : nullptr, e->st);
    if (rc) return rc;
    return env_sync(e);
}
int qg_env_step(qg_env *e, int64_t action) { return env_step(e, action, nullptr); }
int qg_env_step_coin(qg_env *e, int64_t action, int coin) {
    uint8_t c = coin ? 1 : 0;
    return env_step(e, action, &c);
}

// the results of the last mutating call are in the I/O block (every such call drains the env's stream before it returns)
int qg_env_is_final(const qg_env *e) { return e ? (int)e->io->done : -1; }
int qg_env_success(const qg_env *e) { return e ? (int)e->io->success : -1; }
float qg_env_reward(const qg_env *e) { return e ? e->io->reward : 0.0f; }
int64_t qg_env_masks(const qg_env *e, uint8_t *out, size_t cap) {  // clifford.rs:349-351
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    const int s = e->io->success;
    const size_t n = e->v->gates.size();
    for (size_t i = 0; i < n && i < cap; ++i) out[i] = s ? 0 : 1;
    return (int64_t)n;
}

int64_t qg_env_observe(qg_env *e, int64_t *out, size_t cap) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    qg_vec *v = e->v;
    if (!e->obs_valid) {  // PauliEnv with add_perms (a draw per call), or a state nobody has observed yet
        QG_ON_DEVICE(v);
        int rc = qg_vec_observe_dense(v, e->obs_dev, e->st);  // written straight into pinned host memory
        if (rc) return rc;
        if (hipStreamSynchronize(e->st) != hipSuccess) {
            (void)hipGetLastError();
            return set_error(QG_ERR_DEVICE, "observation failed");
        }
        e->obs_valid = e->obs_ahead;
    }
    const size_t n = e->obs_bytes;
    const int8_t *host = e->obs;
    int64_t cnt = 0;  // ascending flat indices of the set entries (clifford.rs:361-368)
    for (size_t i = 0; i < n; ++i)
        if (host[i]) {
            if ((size_t)cnt < cap && out) out[cnt] = (int64_t)i;
            ++cnt;
        }
    return cnt;
}

int qg_env_track_solution(const qg_env *e) { return e ? (e->v->cfg.track_solution != 0) : -1; }
int64_t qg_env_solution(const qg_env *e, uint64_t *out, size_t cap) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    return qg_vec_solution(e->v, 0, out, cap);
}

int64_t qg_env_twists(const qg_env *ce, int64_t *obs_out, int64_t *act_out) {
    if (!ce) return set_error(QG_ERR_INVALID, "null argument");
    qg_env *e = const_cast<qg_env *>(ce);
    const qg_vec *v = e->v;
    // add_perms off -> (empty, empty) (clifford.rs:218-222); PauliEnv always returns empty (pauli.rs:675-679)
    if (!v->cfg.add_perms || v->cfg.env_kind == QG_PAULI) return 0;
    if (!e->twists_done) {
        std::vector<Perm> qp;
        compute_qubit_and_action_perms(v->N, v->gates, qp, e->act_perms);
        const uint32_t N = v->N;
        for (const Perm &p : qp) {
            Perm op;
            if (v->cfg.env_kind == QG_CLIFFORD) {  // obs_perm_clifford (symmetry.rs:276-295)
                const uint32_t dim = 2 * N;
                op.resize((size_t)dim * dim);
                for (uint32_t r = 0; r < dim; ++r) {
                    const int64_t mr = r < N ? p[r] : N + p[r - N];
                    for (uint32_t c = 0; c < dim; ++c) {
                        const int64_t mc = c < N ? p[c] : N + p[c - N];
                        op[(size_t)r * dim + c] = mr * dim + mc;
                    }
                }
            } else {  // obs_perm_square (symmetry.rs:265-274)
                op.resize((size_t)N * N);
                for (uint32_t r = 0; r < N; ++r)
                    for (uint32_t c = 0; c < N; ++c) op[(size_t)r * N + c] = p[r] * N + p[c];
            }
            e->obs_perms.push_back(op);
        }
        e->twists_done = true;
    }
    const size_t n = e->obs_perms.size();
    for (size_t i = 0; i < n; ++i) {
        if (obs_out) std::copy(e->obs_perms[i].begin(), e->obs_perms[i].end(), obs_out + i * e->obs_perms[i].size());
        if (act_out) std::copy(e->act_perms[i].begin(), e->act_perms[i].end(), act_out + i * e->act_perms[i].size());
    }
    return (int64_t)n;
}

}  // extern "C"
