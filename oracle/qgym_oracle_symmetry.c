/*
 * qgym_oracle_symmetry.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): the "twists".
 *
 * Restatement of rust/src/envs/symmetry.rs of the reference: coupling-graph automorphisms and the
 * observation / action permutations they induce (Env::twists, clifford.rs:370-372; PauliEnv's
 * internal qubit permutations, pauli.rs:374-378).  Function by function:
 *
 *   all_permutations / heap_permute      symmetry.rs:84-113   (Heap's algorithm, its visiting order)
 *   compute_automorphisms                symmetry.rs:115-176
 *   build_action_perm                    symmetry.rs:178-203
 *   compute_twists_with_builder          symmetry.rs:205-263
 *   obs_perm_square / obs_perm_clifford  symmetry.rs:265-295
 *   compute_qubit_perms                  symmetry.rs:307-361
 *
 * Third-party step: symmetry.rs:146-167 enumerates the automorphisms with petgraph 0.6.5's VF2
 * (`subgraph_isomorphisms_iter(g, g)`, keeping the mappings that cover all n nodes).  A node-covering
 * monomorphism of a finite graph into itself is an automorphism, the set of automorphisms is closed
 * under inversion (so the direction of VF2's mapping does not matter), and the reference sorts and
 * de-duplicates the list (:173-175) -- the enumeration order of the library is unobservable.  Here
 * the same set is produced by plain backtracking over adjacency-preserving bijections and sorted the
 * same way.  tests/test_oracle_symmetry.py pins that set against a brute force over all n! permutations.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "qgym_oracle.h"

/* ---- a growable list of fixed-length index vectors ------------------------------------------ */
typedef struct {
    size_t *v;   /* n_items x len */
    size_t len;
    size_t n_items, cap;
} perm_list;

static void pl_init(perm_list *p, size_t len) {
    p->v = NULL;
    p->len = len;
    p->n_items = p->cap = 0;
}
static void pl_push(perm_list *p, const size_t *item) {
    if (p->n_items == p->cap) {
        p->cap = p->cap ? 2 * p->cap : 16;
        p->v = (size_t *)realloc(p->v, sizeof(size_t) * p->cap * (p->len ? p->len : 1));
    }
    memcpy(p->v + p->n_items * p->len, item, sizeof(size_t) * p->len);
    p->n_items++;
}
static void pl_free(perm_list *p) { free(p->v); }

/* symmetry.rs:88-104 */
static void heap_permute(size_t k, size_t *perm, perm_list *results) {
    if (k == 1) {
        pl_push(results, perm);
        return;
    }
    heap_permute(k - 1, perm, results);
    for (size_t i = 0; i < k - 1; i++) {
        size_t a = (k % 2 == 0) ? i : 0, t = perm[a];
        perm[a] = perm[k - 1];
        perm[k - 1] = t;
        heap_permute(k - 1, perm, results);
    }
}

/* lexicographic order of Vec<usize> (`results.sort()`, symmetry.rs:173) */
static size_t g_cmp_len;
static int cmp_perm(const void *a, const void *b) {
    const size_t *x = (const size_t *)a, *y = (const size_t *)b;
    for (size_t i = 0; i < g_cmp_len; i++)
        if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
    return 0;
}

/* every bijection node -> node with adj[i][j] == adj[map[i]][map[j]] (what VF2 yields, symmetry.rs:146-167) */
static void automorphism_search(const uint8_t *adj, const size_t *degree, size_t n, size_t pos, size_t *map, uint8_t *used, perm_list *out) {
    if (pos == n) {
        pl_push(out, map);
        return;
    }
    for (size_t cand = 0; cand < n; cand++) {
        if (used[cand] || degree[cand] != degree[pos]) continue;
        int ok = 1;
        for (size_t prev = 0; prev < pos && ok; prev++) ok = adj[pos * n + prev] == adj[cand * n + map[prev]];
        if (!ok) continue;
        used[cand] = 1;
        map[pos] = cand;
        automorphism_search(adj, degree, n, pos + 1, map, used, out);
        used[cand] = 0;
    }
}

/* symmetry.rs:115-176 */
static void compute_automorphisms(const uint8_t *adj, size_t n, int has_edge, perm_list *results) {
    pl_init(results, n);
    if (n == 0) { /* `return vec![Vec::new()]` */
        size_t dummy = 0;
        pl_push(results, &dummy);
        return;
    }
    size_t *perm = (size_t *)malloc(sizeof(size_t) * n);
    for (size_t i = 0; i < n; i++) perm[i] = i;
    if (!has_edge) { /* :121-123: every permutation, in Heap's order, NOT sorted */
        heap_permute(n, perm, results);
        free(perm);
        return;
    }
    size_t *degree = (size_t *)calloc(n, sizeof(size_t));
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) degree[i] += adj[i * n + j];
    uint8_t *used = (uint8_t *)calloc(n, 1);
    automorphism_search(adj, degree, n, 0, perm, used, results);
    if (results->n_items == 0) { /* :169-171 */
        for (size_t i = 0; i < n; i++) perm[i] = i;
        pl_push(results, perm);
    }
    g_cmp_len = n;
    qsort(results->v, results->n_items, sizeof(size_t) * n, cmp_perm); /* :173 */
    size_t w = 0;                                                      /* :174 dedup */
    for (size_t r = 0; r < results->n_items; r++)
        if (w == 0 || cmp_perm(results->v + (w - 1) * n, results->v + r * n) != 0) {
            if (w != r) memmove(results->v + w * n, results->v + r * n, sizeof(size_t) * n);
            w++;
        }
    results->n_items = w;
    free(used);
    free(degree);
    free(perm);
}

/* GateKey (symmetry.rs:33-37, 66-71): kind + qubits, SWAP's qubits sorted */
typedef struct {
    int32_t kind;
    size_t q[2];
    size_t nq;
} gate_key;

static gate_key canonical_key(int32_t kind, size_t q0, size_t q1) {
    gate_key k;
    k.kind = kind;
    k.nq = kind >= OG_CX ? 2 : 1;
    k.q[0] = q0;
    k.q[1] = k.nq == 2 ? q1 : 0;
    if (kind == OG_SWAP && k.q[0] > k.q[1]) {
        size_t t = k.q[0];
        k.q[0] = k.q[1];
        k.q[1] = t;
    }
    return k;
}
static int key_eq(const gate_key *a, const gate_key *b) {
    return a->kind == b->kind && a->nq == b->nq && a->q[0] == b->q[0] && (a->nq == 1 || a->q[1] == b->q[1]);
}
/* gate_index.get(&key): the HashMap was filled in gateset order with `insert`, so the LAST gate with a key owns it
 * (symmetry.rs:217-223) */
static int64_t gate_index_get(const og_gate *gates, size_t n_gates, const gate_key *key) {
    for (size_t i = n_gates; i-- > 0;) {
        gate_key k = canonical_key(gates[i].kind, (size_t)gates[i].q0, (size_t)gates[i].q1);
        if (key_eq(&k, key)) return (int64_t)i;
    }
    return -1;
}

/* symmetry.rs:178-203; returns 0 for None */
static int build_action_perm(const og_gate *gates, size_t n_gates, const size_t *perm, size_t perm_len, size_t *act_perm) {
    for (size_t g = 0; g < n_gates; g++) {
        size_t q0 = (size_t)gates[g].q0, q1 = (size_t)gates[g].q1;
        const int two = gates[g].kind >= OG_CX;
        if (q0 >= perm_len || (two && q1 >= perm_len)) return 0; /* :190-192 */
        gate_key key = canonical_key(gates[g].kind, perm[q0], two ? perm[q1] : 0);
        int64_t idx = gate_index_get(gates, n_gates, &key);
        if (idx < 0) return 0;
        act_perm[g] = (size_t)idx;
    }
    return 1;
}

/* the part compute_twists_with_builder (symmetry.rs:205-263) and compute_qubit_perms (:307-361) share: the qubit
 * permutations that map the gateset onto itself, with their action permutations */
static void qubit_and_action_perms(size_t n, const og_gate *gates, size_t n_gates, perm_list *qubit_perms, perm_list *act_perms) {
    pl_init(qubit_perms, n);
    pl_init(act_perms, n_gates);
    if (n == 0) return; /* :214-216 */
    uint8_t *adj = (uint8_t *)calloc(n * n, 1);
    int has_edge = 0;
    for (size_t g = 0; g < n_gates; g++) /* :225-236 */
        if (gates[g].kind >= OG_CX && gates[g].q0 != gates[g].q1) {
            adj[(size_t)gates[g].q0 * n + (size_t)gates[g].q1] = 1;
            adj[(size_t)gates[g].q1 * n + (size_t)gates[g].q0] = 1;
            has_edge = 1;
        }
    perm_list autos;
    compute_automorphisms(adj, n, has_edge, &autos);
    size_t *ap = (size_t *)malloc(sizeof(size_t) * (n_gates ? n_gates : 1));
    for (size_t i = 0; i < autos.n_items; i++) { /* :244-252 */
        const size_t *perm = autos.v + i * n;
        /* `if !seen.insert(perm.clone()) { continue; }` never fires: the list is de-duplicated (:174) or, in the
         * no-edge branch, holds every permutation exactly once */
        if (build_action_perm(gates, n_gates, perm, n, ap)) {
            pl_push(qubit_perms, perm);
            pl_push(act_perms, ap);
        }
    }
    if (qubit_perms->n_items == 0) { /* :254-260 */
        size_t *id = (size_t *)malloc(sizeof(size_t) * n);
        for (size_t i = 0; i < n; i++) id[i] = i;
        if (build_action_perm(gates, n_gates, id, n, ap)) {
            pl_push(qubit_perms, id);
            pl_push(act_perms, ap);
        }
        free(id);
    }
    free(ap);
    pl_free(&autos);
    free(adj);
}

int64_t og_qubit_perms(size_t num_qubits, const og_gate *gates, size_t n_gates, int64_t *qubit_out, int64_t *act_out) {
    perm_list qp, ap;
    qubit_and_action_perms(num_qubits, gates, n_gates, &qp, &ap);
    for (size_t i = 0; i < qp.n_items; i++) {
        if (qubit_out)
            for (size_t k = 0; k < num_qubits; k++) qubit_out[i * num_qubits + k] = (int64_t)qp.v[i * num_qubits + k];
        if (act_out)
            for (size_t k = 0; k < n_gates; k++) act_out[i * n_gates + k] = (int64_t)ap.v[i * n_gates + k];
    }
    const int64_t n = (int64_t)qp.n_items;
    pl_free(&qp);
    pl_free(&ap);
    return n;
}

int64_t og_twists(int32_t env_kind, size_t n, const og_gate *gates, size_t n_gates, int64_t *obs_out, int64_t *act_out) {
    /* PauliEnv::twists returns (vec![], vec![]) whatever the constructor computed (pauli.rs:675-679) */
    if (env_kind == OG_ENV_PAULI) return 0;
    perm_list qp, ap;
    qubit_and_action_perms(n, gates, n_gates, &qp, &ap);
    const size_t dim = env_kind == OG_ENV_CLIFFORD ? 2 * n : n, obs = dim * dim;
    for (size_t i = 0; i < qp.n_items; i++) {
        const size_t *perm = qp.v + i * n;
        if (obs_out) {
            int64_t *o = obs_out + i * obs;
            if (env_kind == OG_ENV_CLIFFORD) { /* obs_perm_clifford, symmetry.rs:276-295 */
                for (size_t row = 0; row < dim; row++) {
                    const size_t mapped_row = row < n ? perm[row] : n + perm[row - n];
                    for (size_t col = 0; col < dim; col++) {
                        const size_t mapped_col = col < n ? perm[col] : n + perm[col - n];
                        o[row * dim + col] = (int64_t)(mapped_row * dim + mapped_col);
                    }
                }
            } else { /* obs_perm_square, symmetry.rs:265-274 */
                for (size_t row = 0; row < n; row++)
                    for (size_t col = 0; col < n; col++) o[row * n + col] = (int64_t)(perm[row] * n + perm[col]);
            }
        }
        if (act_out)
            for (size_t k = 0; k < n_gates; k++) act_out[i * n_gates + k] = (int64_t)ap.v[i * n_gates + k];
    }
    const int64_t cnt = (int64_t)qp.n_items;
    pl_free(&qp);
    pl_free(&ap);
    return cnt;
}
