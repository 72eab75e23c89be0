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

struct qg_env {
    qg_vec *v = nullptr;
    bool twists_done = false;
    std::vector<Perm> obs_perms, act_perms;
};

static int env_scalar_u8(const qg_env *e, const uint8_t *dev, uint8_t *out) {
    if (hipMemcpy(out, dev, 1, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "device read failed");
    }
    (void)e;
    return QG_OK;
}

extern "C" {

int qg_env_create(const qg_config *cfg, const qg_gate *gates, size_t n_gates, int device, qg_env **out) {
    if (!out) return set_error(QG_ERR_INVALID, "null argument");
    *out = nullptr;
    qg_vec *v = nullptr;
    int rc = qg_vec_create(cfg, gates, n_gates, 1, device, &v);
    if (rc) return rc;
    qg_env *e = new qg_env();
    e->v = v;
    *out = e;
    return QG_OK;
}

void qg_env_destroy(qg_env *e) {
    if (!e) return;
    qg_vec_destroy(e->v);
    delete e;
}

int qg_env_clone(const qg_env *e, qg_env **out) {  // Env: DynClone -- deep copy of every resident buffer
    if (!e || !out) return set_error(QG_ERR_INVALID, "null argument");
    qg_env *c = nullptr;
    int rc = qg_env_create(&e->v->cfg, e->v->gates.data(), e->v->gates.size(), e->v->device, &c);
    if (rc) return rc;
    const qg_vec *s = e->v;
    qg_vec *d = c->v;
    d->difficulty = s->difficulty;
    d->step_index = s->step_index;
    d->coin_seed = s->coin_seed;
    d->maybe_nonsymplectic = s->maybe_nonsymplectic;
    d->observe_counter = s->observe_counter;
    struct { void *dst; const void *src; size_t bytes; } copies[] = {
        {d->state, s->state, s->state_bytes},
        {d->depth, s->depth, 4},
        {d->reward, s->reward, 4},
        {d->done, s->done, 1},
        {d->success, s->success, 1},
        {d->inverted, s->inverted, 1},
        {d->error, s->error, 4},
        {d->sol_len, s->sol_len, 8},
        {d->sol, s->sol, (size_t)s->sol_cap * 4},
        {d->layers, s->layers, (size_t)s->layers_len * 4 * 64},  // the env's tile (layer_rec)
        {d->bad, s->bad, s->layout == LAYOUT_LFD ? (size_t)16 : s->layout == LAYOUT_TILE64 ? (size_t)8 : (size_t)4},  // incremental solved masks
        {d->perm_idx, s->perm_idx, (size_t)4},                                 // PauliEnv current_perm_idx (pauli.rs:661)
    };
    for (auto &cp : copies)
        if (cp.dst && cp.src && cp.bytes && hipMemcpy(cp.dst, cp.src, cp.bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipGetLastError();
            qg_env_destroy(c);
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
    int rc = qg_vec_set_state(e->v, state, QG_FMT_I64, n, 0, nullptr);
    if (rc) return rc;
    return qg_vec_sync(e->v, nullptr);
}

int qg_env_reset(qg_env *e, uint64_t seed) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    int rc = qg_vec_reset(e->v, seed, nullptr);
    if (rc) return rc;
    return qg_vec_sync(e->v, nullptr);
}

static int env_step(qg_env *e, int64_t action, const uint8_t *coin) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    qg_vec *v = e->v;
    int rc = ensure_scratch_public(v, 16);
    if (rc) return rc;
    int64_t *a_dev = reinterpret_cast<int64_t *>(v->scratch);
    uint8_t *c_dev = reinterpret_cast<uint8_t *>(v->scratch) + 8;
    if (hipMemcpy(a_dev, &action, 8, hipMemcpyHostToDevice) != hipSuccess ||
        (coin && hipMemcpy(c_dev, coin, 1, hipMemcpyHostToDevice) != hipSuccess)) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "action upload failed");
    }
    rc = qg_vec_step(v, a_dev, QG_ACT_I64, coin ? c_dev : nullptr, nullptr);
    if (rc) return rc;
    return qg_vec_sync(v, nullptr);
}
int qg_env_step(qg_env *e, int64_t action) { return env_step(e, action, nullptr); }
int qg_env_step_coin(qg_env *e, int64_t action, int coin) {
    uint8_t c = coin ? 1 : 0;
    return env_step(e, action, &c);
}

int qg_env_is_final(const qg_env *e) {
    uint8_t x = 0;
    if (!e || env_scalar_u8(e, e->v->done, &x)) return -1;
    return x;
}
int qg_env_success(const qg_env *e) {
    uint8_t x = 0;
    if (!e || env_scalar_u8(e, e->v->success, &x)) return -1;
    return x;
}
float qg_env_reward(const qg_env *e) {
    float r = 0.0f;
    if (!e || hipMemcpy(&r, e->v->reward, 4, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        set_error(QG_ERR_DEVICE, "device read failed");
        return 0.0f;
    }
    return r;
}
int64_t qg_env_masks(const qg_env *e, uint8_t *out, size_t cap) {  // clifford.rs:349-351
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    int s = qg_env_success(e);
    if (s < 0) return QG_ERR_DEVICE;
    const size_t n = e->v->gates.size();
    for (size_t i = 0; i < n && i < cap; ++i) out[i] = s ? 0 : 1;
    return (int64_t)n;
}

int64_t qg_env_observe(qg_env *e, int64_t *out, size_t cap) {
    if (!e) return set_error(QG_ERR_INVALID, "null argument");
    qg_vec *v = e->v;
    qg_vec_info info;
    qg_vec_get_info(v, &info);
    const size_t n = (size_t)info.obs_rows * info.obs_cols;
    int rc = ensure_scratch_public(v, n + 16);
    if (rc) return rc;
    int8_t *dev = reinterpret_cast<int8_t *>(v->scratch) + 16;
    rc = qg_vec_observe_dense(v, dev, nullptr);
    if (rc) return rc;
    std::vector<int8_t> host(n);
    if (hipMemcpy(host.data(), dev, n, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return set_error(QG_ERR_DEVICE, "observation copy failed");
    }
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
