/*
 * qgym_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See qgym_oracle.h.
 *
 * Scalar restatement of the reference env state machines, one byte per GF(2) entry exactly like
 * the Rust (`Vec<bool>` / `DMatrix<u8>`), same operation order, same quirks.  Citations are
 * relative to /root/reference/.
 */
#include "qgym_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
const char *og_last_error(void) { return g_err; }
static int fail(const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * HashSet<usize> stand-in (metrics.rs:23-24 `cnot_layers`, `layers`): membership bitmap + len.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *has;
    size_t cap;
    size_t len;
} og_set;

static void set_clear(og_set *s) {
    if (s->has) memset(s->has, 0, s->cap);
    s->len = 0;
}
static void set_insert(og_set *s, size_t v) {
    if (v >= s->cap) {
        size_t ncap = s->cap ? s->cap : 64;
        while (ncap <= v) ncap *= 2;
        s->has = (uint8_t *)realloc(s->has, ncap);
        memset(s->has + s->cap, 0, ncap - s->cap);
        s->cap = ncap;
    }
    if (!s->has[v]) {
        s->has[v] = 1;
        s->len++;
    }
}
static void set_copy(og_set *d, const og_set *s) {
    d->cap = s->cap;
    d->len = s->len;
    d->has = NULL;
    if (s->cap) {
        d->has = (uint8_t *)malloc(s->cap);
        memcpy(d->has, s->has, s->cap);
    }
}

/* ------------------------------------------------------------------------------------------
 * MetricsTracker / MetricsCounts / MetricsWeights  (rust/src/envs/metrics.rs:18-185)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t n_cnots, n_layers_cnots, n_layers, n_gates;
} og_counts;

typedef struct {
    size_t num_qubits;
    size_t n_cnots, n_gates;
    og_set cnot_layers, layers;
    long *last_gates; /* Vec<isize>, -1 initial (metrics.rs:37-38) */
    long *last_cxs;
} og_metrics;

static void metrics_init(og_metrics *m, size_t nq) { /* metrics.rs:30-40 */
    memset(m, 0, sizeof *m);
    m->num_qubits = nq;
    m->last_gates = (long *)malloc(sizeof(long) * (nq ? nq : 1));
    m->last_cxs = (long *)malloc(sizeof(long) * (nq ? nq : 1));
    for (size_t i = 0; i < nq; i++) m->last_gates[i] = m->last_cxs[i] = -1;
}
static void metrics_free(og_metrics *m) {
    free(m->last_gates);
    free(m->last_cxs);
    free(m->cnot_layers.has);
    free(m->layers.has);
}
static void metrics_copy(og_metrics *d, const og_metrics *s) {
    *d = *s;
    size_t nq = s->num_qubits ? s->num_qubits : 1;
    d->last_gates = (long *)malloc(sizeof(long) * nq);
    d->last_cxs = (long *)malloc(sizeof(long) * nq);
    memcpy(d->last_gates, s->last_gates, sizeof(long) * nq);
    memcpy(d->last_cxs, s->last_cxs, sizeof(long) * nq);
    set_copy(&d->cnot_layers, &s->cnot_layers);
    set_copy(&d->layers, &s->layers);
}
static void metrics_reset(og_metrics *m) { /* metrics.rs:42-53 */
    m->n_cnots = 0;
    m->n_gates = 0;
    set_clear(&m->cnot_layers);
    set_clear(&m->layers);
    for (size_t i = 0; i < m->num_qubits; i++) m->last_gates[i] = m->last_cxs[i] = -1;
}
static og_counts metrics_snapshot(const og_metrics *m) { /* metrics.rs:55-62 */
    og_counts c = {m->n_cnots, m->cnot_layers.len, m->layers.len, m->n_gates};
    return c;
}
static void metrics_single(og_metrics *m, size_t t) { /* metrics.rs:83-95 */
    if (t >= m->num_qubits) return;
    m->n_gates += 1;
    long gate_layer = m->last_gates[t] + 1;
    m->last_gates[t] = gate_layer;
    if (gate_layer >= 0) set_insert(&m->layers, (size_t)gate_layer);
}
static void metrics_cx(og_metrics *m, size_t c, size_t t) { /* metrics.rs:97-123 */
    if (c == t || c >= m->num_qubits || t >= m->num_qubits) return;
    m->n_cnots += 1;
    m->n_gates += 1;
    long a = m->last_gates[c], b = m->last_gates[t];
    long gate_layer = (a > b ? a : b) + 1;
    m->last_gates[c] = gate_layer;
    m->last_gates[t] = gate_layer;
    if (gate_layer >= 0) set_insert(&m->layers, (size_t)gate_layer);
    a = m->last_cxs[c];
    b = m->last_cxs[t];
    long cx_layer = (a > b ? a : b) + 1;
    m->last_cxs[c] = cx_layer;
    m->last_cxs[t] = cx_layer;
    if (cx_layer >= 0) set_insert(&m->cnot_layers, (size_t)cx_layer);
}
static void metrics_apply_gate(og_metrics *m, const og_gate *g) { /* metrics.rs:64-81 */
    size_t a = (size_t)g->q0, b = (size_t)g->q1;
    switch (g->kind) {
    case OG_CX: metrics_cx(m, a, b); break;
    case OG_SWAP:
        metrics_cx(m, a, b);
        metrics_cx(m, b, a);
        metrics_cx(m, a, b);
        break;
    case OG_CZ:
        metrics_single(m, b);
        metrics_cx(m, a, b);
        metrics_single(m, b);
        break;
    default: metrics_single(m, a); break;
    }
}
static size_t sat_sub(size_t a, size_t b) { return a > b ? a - b : 0; }

/* metrics.rs:135-146 -- f32, left-to-right, no fused multiply-add (this file is compiled with
 * -ffp-contract=off; the volatile temporaries additionally pin each rounding). */
static float weighted_delta(const og_counts *now, const og_counts *prev, const float w[4]) {
    volatile float dc = (float)sat_sub(now->n_cnots, prev->n_cnots);
    volatile float dlc = (float)sat_sub(now->n_layers_cnots, prev->n_layers_cnots);
    volatile float dl = (float)sat_sub(now->n_layers, prev->n_layers);
    volatile float dg = (float)sat_sub(now->n_gates, prev->n_gates);
    volatile float t0 = w[0] * dc;
    volatile float t1 = w[1] * dlc;
    volatile float t2 = w[2] * dl;
    volatile float t3 = w[3] * dg;
    volatile float s = t0 + t1;
    s = s + t2;
    s = s + t3;
    return s;
}

/* ------------------------------------------------------------------------------------------
 * Byte matrices: CFState (clifford.rs:28-175) and LFState (linear_function.rs:29-151)
 * share row_xor / swap_rows / solved / inverse over a dim x dim row-major Vec<bool>.
 * ---------------------------------------------------------------------------------------- */
static void mat_identity(uint8_t *d, size_t dim) {
    memset(d, 0, dim * dim);
    for (size_t i = 0; i < dim; i++) d[i * dim + i] = 1;
}
static void row_xor(uint8_t *d, size_t dim, size_t dest, size_t src) { /* clifford.rs:64-72 */
    if (dest == src) return;
    for (size_t c = 0; c < dim; c++) d[dest * dim + c] ^= d[src * dim + c];
}
static void swap_rows(uint8_t *d, size_t dim, size_t r1, size_t r2) { /* clifford.rs:74-82 */
    if (r1 == r2) return;
    for (size_t c = 0; c < dim; c++) {
        uint8_t t = d[r1 * dim + c];
        d[r1 * dim + c] = d[r2 * dim + c];
        d[r2 * dim + c] = t;
    }
}
static int mat_solved(const uint8_t *d, size_t dim) { /* clifford.rs:136-145, linear_function.rs:91-100 */
    for (size_t i = 0; i < dim; i++)
        for (size_t j = 0; j < dim; j++)
            if (d[i * dim + j] != (uint8_t)(i == j)) return 0;
    return 1;
}
/* clifford.rs:147-170 / linear_function.rs:124-146: Gauss-Jordan, first pivot below.
 * Returns -1 where the reference panics ("... is singular; cannot invert"). */
static int mat_inverse(const uint8_t *src, size_t dim, uint8_t *inv) {
    uint8_t *mat = (uint8_t *)malloc(dim * dim);
    memcpy(mat, src, dim * dim);
    mat_identity(inv, dim);
    for (size_t col = 0; col < dim; col++) {
        if (!mat[col * dim + col]) {
            size_t pivot = dim;
            for (size_t row = col + 1; row < dim; row++)
                if (mat[row * dim + col]) {
                    pivot = row;
                    break;
                }
            if (pivot == dim) {
                free(mat);
                return fail("state is singular; cannot invert");
            }
            swap_rows(mat, dim, col, pivot);
            swap_rows(inv, dim, col, pivot);
        }
        for (size_t row = 0; row < dim; row++)
            if (row != col && mat[row * dim + col]) {
                row_xor(mat, dim, row, col);
                row_xor(inv, dim, row, col);
            }
    }
    free(mat);
    return 0;
}

/* clifford.rs:89-133,249-260 */
static void clifford_apply(uint8_t *d, size_t n, const og_gate *g) {
    size_t dim = 2 * n, a = (size_t)g->q0, b = (size_t)g->q1;
    switch (g->kind) {
    case OG_H: swap_rows(d, dim, a, n + a); break;           /* :89-92 */
    case OG_S:
    case OG_SDG: row_xor(d, dim, n + a, a); break;            /* :95-99 */
    case OG_SX:
    case OG_SXDG: row_xor(d, dim, a, n + a); break;           /* :102-106 */
    case OG_CX:                                                /* :111-116 */
        if (a == b) return;
        row_xor(d, dim, b, a);
        row_xor(d, dim, n + a, n + b);
        break;
    case OG_CZ:                                                /* :120-125 */
        if (a == b) return;
        row_xor(d, dim, n + a, b);
        row_xor(d, dim, n + b, a);
        break;
    case OG_SWAP:                                              /* :128-133 */
        if (a == b) return;
        swap_rows(d, dim, a, b);
        swap_rows(d, dim, n + a, n + b);
        break;
    }
}
/* linear_function.rs:62-83,237-243: cx = row q2 ^= row q1; swap = row swap; others ignored */
static void lf_apply(uint8_t *d, size_t n, const og_gate *g) {
    size_t a = (size_t)g->q0, b = (size_t)g->q1;
    if (g->kind == OG_CX) {
        if (a == b) return;
        for (size_t c = 0; c < n; c++) d[b * n + c] = d[a * n + c] ^ d[b * n + c];
    } else if (g->kind == OG_SWAP) {
        if (a == b) return;
        for (size_t c = 0; c < n; c++) {
            uint8_t av = d[a * n + c], bv = d[b * n + c];
            d[a * n + c] = bv;
            d[b * n + c] = av;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * Pauli (rust/src/pauli/pauli.rs:39-134)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *base_z, *base_x; /* Vec<bool>, length n */
    size_t n;
    int32_t base_phase, init_phase;
} og_pauli;

static void pauli_free(og_pauli *p) {
    free(p->base_z);
    free(p->base_x);
}
static void pauli_copy(og_pauli *d, const og_pauli *s) {
    *d = *s;
    d->base_z = (uint8_t *)malloc(s->n ? s->n : 1);
    d->base_x = (uint8_t *)malloc(s->n ? s->n : 1);
    memcpy(d->base_z, s->base_z, s->n);
    memcpy(d->base_x, s->base_x, s->n);
}
/* pauli.rs:22-37,48-81: label = ^([+-]?[ij1]?)([IXYZ]*)$ ; coefficient canonicalised by
 * dropping '1' and '+', 'j'->'i'; map ""->0, "-i"->1, "-"->2, "i"->3; the Pauli characters are
 * reversed (string position p <-> qubit len-1-p, :62). */
static int pauli_from_label(og_pauli *p, const char *label) {
    const char *s = label;
    int neg = 0, has_i = 0;
    if (*s == '+' || *s == '-') {
        neg = (*s == '-');
        s++;
    }
    if (*s == 'i' || *s == 'j') {
        has_i = 1;
        s++;
    } else if (*s == '1') {
        s++;
    }
    size_t len = strlen(s);
    for (size_t i = 0; i < len; i++)
        if (s[i] != 'I' && s[i] != 'X' && s[i] != 'Y' && s[i] != 'Z') return fail("Pauli string label is not valid.");
    int32_t phase = neg ? (has_i ? 1 : 2) : (has_i ? 3 : 0);
    p->n = len;
    p->base_x = (uint8_t *)calloc(len ? len : 1, 1);
    p->base_z = (uint8_t *)calloc(len ? len : 1, 1);
    int32_t ys = 0;
    for (size_t q = 0; q < len; q++) {
        char b = s[len - 1 - q];
        p->base_x[q] = (b == 'X' || b == 'Y');
        p->base_z[q] = (b == 'Z' || b == 'Y');
        ys += (b == 'Y');
    }
    p->base_phase = (phase + ys) % 4; /* :73 */
    p->init_phase = phase;
    return 0;
}
static void pauli_evolve_h(og_pauli *p, size_t q) { /* pauli.rs:83-90 */
    uint8_t x = p->base_x[q], z = p->base_z[q];
    p->base_x[q] = z;
    p->base_z[q] = x;
    p->base_phase = (p->base_phase + 2 * (int32_t)(x && z)) % 4;
}
static void pauli_evolve_s(og_pauli *p, size_t q) { /* pauli.rs:92-97 */
    uint8_t x = p->base_x[q];
    p->base_z[q] ^= x;
    p->base_phase = (p->base_phase + (int32_t)x) % 4;
}
static void pauli_evolve_cx(og_pauli *p, size_t qc, size_t qt) { /* pauli.rs:99-103 */
    p->base_x[qt] ^= p->base_x[qc];
    p->base_z[qc] ^= p->base_z[qt];
}
static void pauli_evolve_sx(og_pauli *p, size_t q) { /* pauli.rs:105-110 */
    pauli_evolve_h(p, q);
    pauli_evolve_s(p, q);
    pauli_evolve_h(p, q);
}
static int pauli_commutes(const og_pauli *a, const og_pauli *b) { /* pauli.rs:112-123 */
    int32_t acc = 0;
    size_t n = a->n < b->n ? a->n : b->n; /* zip() stops at the shorter */
    for (size_t i = 0; i < n; i++)
        acc += ((int32_t)(a->base_x[i] && b->base_z[i]) + (int32_t)(a->base_z[i] && b->base_x[i])) % 2;
    return acc % 2 == 0;
}
static int32_t pauli_phase(const og_pauli *p) { /* pauli.rs:125-133 */
    int32_t num_ys = 0;
    for (size_t i = 0; i < p->n; i++)
        if (p->base_z[i] && p->base_x[i]) num_ys++;
    return (p->base_phase + (4 * (int32_t)p->n - num_ys)) % 4;
}

/* ------------------------------------------------------------------------------------------
 * PauliDag (rust/src/pauli/pauli_dag.rs:18-58) on petgraph 0.6.5 DiGraph<usize,()>
 *   nodes[i] = weight (rotation index) of the node with NodeIndex i;
 *   edge[w1*R + w2] = 1 for an edge from the node of weight w1 to the node of weight w2.
 * Node removal = petgraph Graph::remove_node: drop the node's edges, then Vec::swap_remove.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t *nodes;
    size_t count;
    size_t R;       /* number of rotations at construction */
    uint8_t *edge;  /* R*R, keyed by weights (weights are unique) */
    uint8_t *present; /* R, node with this weight still in the graph */
} og_dag;

static void dag_free(og_dag *g) {
    free(g->nodes);
    free(g->edge);
    free(g->present);
}
static void dag_copy(og_dag *d, const og_dag *s) {
    *d = *s;
    size_t R = s->R ? s->R : 1;
    d->nodes = (size_t *)malloc(sizeof(size_t) * R);
    d->edge = (uint8_t *)malloc(R * R);
    d->present = (uint8_t *)malloc(R);
    memcpy(d->nodes, s->nodes, sizeof(size_t) * R);
    memcpy(d->edge, s->edge, R * R);
    memcpy(d->present, s->present, R);
}
static void dag_new(og_dag *g, const og_pauli *rot, size_t R) { /* pauli_dag.rs:25-44 */
    g->R = R;
    g->count = R;
    size_t Rs = R ? R : 1;
    g->nodes = (size_t *)malloc(sizeof(size_t) * Rs);
    g->edge = (uint8_t *)calloc(Rs * Rs, 1);
    g->present = (uint8_t *)calloc(Rs, 1);
    for (size_t i = 0; i < R; i++) {
        g->nodes[i] = i;
        g->present[i] = 1;
    }
    for (size_t i1 = 0; i1 < R; i1++)
        for (size_t i2 = 0; i2 < i1; i2++)
            if (!pauli_commutes(&rot[i1], &rot[i2])) g->edge[i1 * R + i2] = 1; /* :37-39 i1 -> i2 */
}
static int dag_out_degree_zero(const og_dag *g, size_t w) { /* pauli_dag.rs:50-55 */
    for (size_t w2 = 0; w2 < g->R; w2++)
        if (g->present[w2] && g->edge[w * g->R + w2]) return 0;
    return 1;
}
/* petgraph Graph::retain_nodes: `for index in self.node_indices().rev()` -> remove_node(index)
 * when the predicate is false; remove_node ends with `self.nodes.swap_remove(index)`. */
static void dag_remove_marked(og_dag *g, const uint8_t *doomed_by_index) {
    for (size_t i = g->count; i-- > 0;) {
        if (doomed_by_index[i]) {
            g->present[g->nodes[i]] = 0;
            g->nodes[i] = g->nodes[g->count - 1];
            g->count--;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * PauliNetwork (rust/src/pauli/pauli_network.rs:27-261)
 *   data: (2N) x (2N + R) bytes, row-major here (nalgebra's storage order is unobservable:
 *   from_vec is column-major and is followed by .transpose(), :44-45, so entry (r,c) of the
 *   tableau part is input[r*2N + c]).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    size_t n;        /* num_qubits */
    size_t R;        /* rotation count */
    size_t ncols;    /* 2n + R */
    uint8_t *data;
    og_pauli *rot;   /* rotation_qk */
    og_dag dag;
} og_net;

typedef struct {
    int axis; /* 0 X, 1 Y, 2 Z (pauli_network.rs:20-25) */
    size_t qubit;
    size_t index;
} og_removed;

static void net_free(og_net *w) {
    free(w->data);
    for (size_t i = 0; i < w->R; i++) pauli_free(&w->rot[i]);
    free(w->rot);
    dag_free(&w->dag);
}
static void net_copy(og_net *d, const og_net *s) {
    *d = *s;
    size_t sz = 2 * s->n * s->ncols;
    d->data = (uint8_t *)malloc(sz ? sz : 1);
    memcpy(d->data, s->data, sz);
    d->rot = (og_pauli *)malloc(sizeof(og_pauli) * (s->R ? s->R : 1));
    for (size_t i = 0; i < s->R; i++) pauli_copy(&d->rot[i], &s->rot[i]);
    dag_copy(&d->dag, &s->dag);
}
/* pauli_network.rs:37-77.  labels: n_rot strings laid end to end, each '\0'-terminated. */
static int net_new(og_net *w, size_t n, const uint8_t *tableau, const char *labels, size_t n_rot) {
    memset(w, 0, sizeof *w);
    w->n = n;
    w->R = n_rot;
    w->ncols = 2 * n + n_rot;
    w->rot = (og_pauli *)calloc(n_rot ? n_rot : 1, sizeof(og_pauli));
    const char *s = labels;
    size_t parsed = 0;
    int rc = 0;
    for (size_t i = 0; i < n_rot; i++) {
        if (pauli_from_label(&w->rot[i], s) != 0) { rc = -1; break; }
        parsed++;
        if (w->rot[i].n != n) { /* :52-58 panic */
            rc = fail("Number of qubits differ for Clifford and Paulis");
            break;
        }
        s += strlen(s) + 1;
    }
    if (rc != 0) {
        for (size_t i = 0; i < parsed; i++) pauli_free(&w->rot[i]);
        free(w->rot);
        memset(w, 0, sizeof *w);
        return -1;
    }
    size_t rows = 2 * n;
    w->data = (uint8_t *)calloc(rows * w->ncols ? rows * w->ncols : 1, 1);
    for (size_t r = 0; r < rows; r++)
        for (size_t c = 0; c < rows; c++) w->data[r * w->ncols + c] = tableau[r * rows + c];
    for (size_t i = 0; i < n_rot; i++)
        for (size_t q = 0; q < n; q++) {
            w->data[q * w->ncols + 2 * n + i] = w->rot[i].base_x[q];        /* :62-63 */
            w->data[(n + q) * w->ncols + 2 * n + i] = w->rot[i].base_z[q];  /* :64-65 */
        }
    dag_new(&w->dag, w->rot, n_rot);
    return 0;
}
#define NET(w, r, c) ((w)->data[(r) * (w)->ncols + (c)])

static int net_is_trivial(const og_net *w, size_t ri) { /* pauli_network.rs:79-93 (u8 sum) */
    uint8_t sum = 0;
    for (size_t q = 0; q < w->n; q++) sum = (uint8_t)(sum + (NET(w, q, 2 * w->n + ri) | NET(w, w->n + q, 2 * w->n + ri)));
    return sum <= 1;
}
/* pauli_network.rs:95-115; -1 where `.unwrap()` on None panics (weight-0 rotation) */
static long net_which_qubit(const og_net *w, size_t ri) {
    for (size_t q = 0; q < w->n; q++)
        if ((NET(w, q, 2 * w->n + ri) | NET(w, w->n + q, 2 * w->n + ri)) != 0) return (long)q;
    return -1;
}
static int net_which_axis(const og_net *w, size_t ri, size_t q) { /* pauli_network.rs:117-137 */
    if (NET(w, q, 2 * w->n + ri) == 1) return NET(w, w->n + q, 2 * w->n + ri) == 1 ? 1 : 0;
    if (NET(w, w->n + q, 2 * w->n + ri) == 1) return 2;
    return -1;
}
/* pauli_network.rs:139-165.  Appends to out[*n_out..]; returns -1 where the reference panics. */
static int net_clean(og_net *w, og_removed *out, size_t *n_out) {
    int removed = 1;
    size_t cap = w->R ? w->R : 1;
    uint8_t *doomed = (uint8_t *)malloc(cap);
    uint8_t *front = (uint8_t *)malloc(cap);
    while (removed) {
        removed = 0;
        memset(doomed, 0, cap);
        /* get_front_layer() is evaluated once, before the loop body runs (:146) */
        for (size_t i = 0; i < w->dag.count; i++) front[i] = (uint8_t)dag_out_degree_zero(&w->dag, w->dag.nodes[i]);
        for (size_t i = 0; i < w->dag.count; i++) {
            if (!front[i]) continue;
            size_t rindex = w->dag.nodes[i];
            if (net_is_trivial(w, rindex)) {
                long qno = net_which_qubit(w, rindex);
                if (qno < 0) {
                    free(doomed);
                    free(front);
                    return fail("which_qubit: rotation has weight 0 (Option::unwrap on None)");
                }
                int ax = net_which_axis(w, rindex, (size_t)qno);
                out[*n_out].axis = ax;
                out[*n_out].qubit = (size_t)qno;
                out[*n_out].index = rindex;
                (*n_out)++;
                doomed[i] = 1;
                for (size_t r = 0; r < 2 * w->n; r++) NET(w, r, 2 * w->n + rindex) = 0; /* :153-156 */
                removed = 1;
            }
        }
        dag_remove_marked(&w->dag, doomed); /* :160-161 */
    }
    free(doomed);
    free(front);
    return 0;
}
static int net_solved(const og_net *w) { /* pauli_network.rs:167-173 */
    if (w->dag.count != 0) return 0;
    for (size_t r = 0; r < 2 * w->n; r++)
        for (size_t c = 0; c < 2 * w->n; c++)
            if (NET(w, r, c) != (uint8_t)(r == c)) return 0;
    return 1;
}
static void net_xor_rows(og_net *w, size_t a, size_t b) { /* pauli_network.rs:183-187 (no a==b guard) */
    for (size_t c = 0; c < w->ncols; c++) NET(w, a, c) ^= NET(w, b, c);
}
static void net_h(og_net *w, size_t i) { /* :189-194 */
    for (size_t c = 0; c < w->ncols; c++) {
        uint8_t t = NET(w, i, c);
        NET(w, i, c) = NET(w, w->n + i, c);
        NET(w, w->n + i, c) = t;
    }
    for (size_t k = 0; k < w->R; k++) pauli_evolve_h(&w->rot[k], i);
}
static int net_cnot(og_net *w, size_t i, size_t j, og_removed *out, size_t *n_out) { /* :196-207 */
    net_xor_rows(w, i, j);
    net_xor_rows(w, w->n + j, w->n + i);
    for (size_t k = 0; k < w->R; k++) pauli_evolve_cx(&w->rot[k], j, i);
    return net_clean(w, out, n_out);
}
static void net_s(og_net *w, size_t i) { /* :209-215 */
    net_xor_rows(w, w->n + i, i);
    for (size_t k = 0; k < w->R; k++) pauli_evolve_s(&w->rot[k], i);
}
static void net_sx(og_net *w, size_t i) { /* :217-223 */
    net_xor_rows(w, i, w->n + i);
    for (size_t k = 0; k < w->R; k++) pauli_evolve_sx(&w->rot[k], i);
}
/* pauli_network.rs:225-260.  `out` must hold 3*R entries. */
static int net_act(og_net *w, const og_gate *g, og_removed *out, size_t *n_out) {
    size_t a = (size_t)g->q0, b = (size_t)g->q1;
    *n_out = 0;
    switch (g->kind) {
    case OG_H: net_h(w, a); return 0;
    case OG_S: net_s(w, a); return 0;
    case OG_SDG: net_s(w, a); net_s(w, a); net_s(w, a); return 0;
    case OG_SX: net_sx(w, a); return 0;
    case OG_SXDG: net_sx(w, a); net_sx(w, a); net_sx(w, a); return 0;
    case OG_CX: return net_cnot(w, a, b, out, n_out);
    case OG_CZ: {
        net_h(w, b);
        int rc = net_cnot(w, a, b, out, n_out);
        if (rc) return rc;
        net_h(w, b);
        return 0;
    }
    case OG_SWAP: {
        int rc = net_cnot(w, a, b, out, n_out);
        if (rc) return rc;
        rc = net_cnot(w, b, a, out, n_out);
        if (rc) return rc;
        return net_cnot(w, a, b, out, n_out);
    }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * The env objects
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t *v;
    size_t len, cap;
} og_vecu;
static void vec_push(og_vecu *a, uint64_t x) {
    if (a->len == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 16;
        a->v = (uint64_t *)realloc(a->v, sizeof(uint64_t) * a->cap);
    }
    a->v[a->len++] = x;
}
static void vec_copy(og_vecu *d, const og_vecu *s) {
    d->len = s->len;
    d->cap = s->len;
    d->v = NULL;
    if (s->len) {
        d->v = (uint64_t *)malloc(sizeof(uint64_t) * s->len);
        memcpy(d->v, s->v, sizeof(uint64_t) * s->len);
    }
}

struct og_env {
    int32_t kind;
    size_t n;       /* num_qubits */
    size_t dim;     /* matrix dimension: 2n Clifford/Pauli, n LF; n for Permutation */
    uint8_t *mat;   /* Clifford / LF state bytes */
    size_t *perm;   /* Permutation state */
    og_net net;     /* Pauli */
    int has_net;

    size_t depth;
    int success;
    size_t difficulty;
    og_gate *gates;
    size_t n_gates;
    size_t depth_slope, max_depth;
    og_metrics metrics;
    og_counts metrics_values;
    float w[4];
    float reward_value;
    int add_inverts, track_solution, inverted, add_perms;
    og_vecu solution, solution_inv;
    /* Pauli */
    size_t max_rotations, pauli_diff_scale, final_pauli_layers;
    float num_qubits_decay, pauli_layer_reward;
    size_t n_perms;
    size_t *qubit_perms; /* n_perms x n */
    size_t *act_perms;   /* n_perms x n_gates */
    size_t current_perm_idx;
};

void og_config_default(og_config *c, int32_t env_kind, int32_t num_qubits) {
    memset(c, 0, sizeof *c);
    c->num_qubits = num_qubits;
    c->difficulty = 1;   /* envs/synthesis.py:186 */
    c->depth_slope = 2;  /* :187 */
    c->max_depth = 128;  /* :188 */
    c->w_n_cnots = 0.01f;       /* metrics.rs:160 */
    c->w_n_layers_cnots = 0.0f; /* :161 */
    c->w_n_layers = 0.0f;       /* :162 */
    c->w_n_gates = 0.0001f;     /* :163 */
    c->add_inverts = 1;     /* clifford.rs:420 */
    c->add_perms = 1;       /* :421 */
    c->track_solution = 1;  /* :422 */
    c->max_rotations = 5;           /* envs/synthesis.py:387 */
    c->pauli_diff_scale = 16;       /* envs/synthesis.py:388 (Rust-side default 8, pauli.rs:768) */
    c->num_qubits_decay = 0.5f;     /* pauli.rs:769 */
    c->final_pauli_layers = -1;     /* None -> max_rotations + 2 (pauli.rs:760) */
    c->pauli_layer_reward = 0.01f;  /* pauli.rs:773 */
    if (env_kind == OG_ENV_PAULI) c->add_inverts = 0; /* PauliEnv has no inversion path */
}

static void reset_internals(og_env *e);
static int env_solved(const og_env *e);

og_env *og_env_new(int32_t kind, const og_config *c, const og_gate *gates, size_t n_gates) {
    if (kind < 0 || kind > 3 || c->num_qubits < 0) {
        fail("bad env kind / num_qubits");
        return NULL;
    }
    og_env *e = (og_env *)calloc(1, sizeof *e);
    e->kind = kind;
    e->n = (size_t)c->num_qubits;
    e->gates = (og_gate *)malloc(sizeof(og_gate) * (n_gates ? n_gates : 1));
    memcpy(e->gates, gates, sizeof(og_gate) * n_gates);
    e->n_gates = n_gates;
    e->difficulty = (size_t)c->difficulty;
    e->depth_slope = (size_t)c->depth_slope;
    e->max_depth = (size_t)c->max_depth;
    e->w[0] = c->w_n_cnots;
    e->w[1] = c->w_n_layers_cnots;
    e->w[2] = c->w_n_layers;
    e->w[3] = c->w_n_gates;
    e->add_inverts = c->add_inverts != 0;
    e->track_solution = c->track_solution != 0;
    e->add_perms = c->add_perms != 0;
    metrics_init(&e->metrics, e->n);
    e->metrics_values = metrics_snapshot(&e->metrics);
    e->depth = 1; /* clifford.rs:228, linear_function.rs:204, permutation.rs:78, pauli.rs:384 */
    switch (kind) {
    case OG_ENV_CLIFFORD:
        e->dim = 2 * e->n;
        e->mat = (uint8_t *)malloc(e->dim * e->dim ? e->dim * e->dim : 1);
        mat_identity(e->mat, e->dim);
        break;
    case OG_ENV_LINEAR_FUNCTION:
        e->dim = e->n;
        e->mat = (uint8_t *)malloc(e->dim * e->dim ? e->dim * e->dim : 1);
        mat_identity(e->mat, e->dim);
        break;
    case OG_ENV_PERMUTATION:
        e->dim = e->n;
        e->perm = (size_t *)malloc(sizeof(size_t) * (e->n ? e->n : 1));
        for (size_t i = 0; i < e->n; i++) e->perm[i] = i; /* permutation.rs:77 */
        break;
    case OG_ENV_PAULI: {
        e->dim = 2 * e->n;
        e->add_inverts = 0;
        e->max_rotations = c->max_rotations > 1 ? (size_t)c->max_rotations : 1;          /* pauli.rs:387 */
        e->pauli_diff_scale = c->pauli_diff_scale > 1 ? (size_t)c->pauli_diff_scale : 1;  /* :392 */
        e->final_pauli_layers = c->final_pauli_layers >= 0 ? (size_t)c->final_pauli_layers
                                                           : (size_t)c->max_rotations + 2; /* :760 */
        e->num_qubits_decay = c->num_qubits_decay;
        e->pauli_layer_reward = c->pauli_layer_reward;
        uint8_t *id = (uint8_t *)malloc(e->dim * e->dim ? e->dim * e->dim : 1);
        mat_identity(id, e->dim);
        net_new(&e->net, e->n, id, "", 0); /* pauli.rs:355-356 */
        e->has_net = 1;
        free(id);
        if (e->add_perms) { /* pauli.rs:374-378: compute_qubit_perms only when enabled */
            const int64_t np = og_qubit_perms(e->n, e->gates, e->n_gates, NULL, NULL);
            if (np > 0) {
                int64_t *qp = (int64_t *)malloc(sizeof(int64_t) * (size_t)np * (e->n ? e->n : 1));
                int64_t *ap = (int64_t *)malloc(sizeof(int64_t) * (size_t)np * (e->n_gates ? e->n_gates : 1));
                og_qubit_perms(e->n, e->gates, e->n_gates, qp, ap);
                og_pauli_set_perms(e, qp, ap, (size_t)np);
                free(qp);
                free(ap);
            }
        }
        break;
    }
    }
    e->success = env_solved(e); /* clifford.rs:215 (Permutation: `true`, permutation.rs:75) */
    e->reward_value = e->success ? 1.0f : 0.0f;
    return e;
}

og_env *og_env_clone(const og_env *s) {
    og_env *e = (og_env *)malloc(sizeof *e);
    *e = *s;
    e->gates = (og_gate *)malloc(sizeof(og_gate) * (s->n_gates ? s->n_gates : 1));
    memcpy(e->gates, s->gates, sizeof(og_gate) * s->n_gates);
    if (s->mat) {
        e->mat = (uint8_t *)malloc(s->dim * s->dim ? s->dim * s->dim : 1);
        memcpy(e->mat, s->mat, s->dim * s->dim);
    }
    if (s->perm) {
        e->perm = (size_t *)malloc(sizeof(size_t) * (s->n ? s->n : 1));
        memcpy(e->perm, s->perm, sizeof(size_t) * s->n);
    }
    if (s->has_net) net_copy(&e->net, &s->net);
    metrics_copy(&e->metrics, &s->metrics);
    vec_copy(&e->solution, &s->solution);
    vec_copy(&e->solution_inv, &s->solution_inv);
    if (s->n_perms) {
        e->qubit_perms = (size_t *)malloc(sizeof(size_t) * s->n_perms * s->n);
        memcpy(e->qubit_perms, s->qubit_perms, sizeof(size_t) * s->n_perms * s->n);
        e->act_perms = (size_t *)malloc(sizeof(size_t) * s->n_perms * s->n_gates);
        memcpy(e->act_perms, s->act_perms, sizeof(size_t) * s->n_perms * s->n_gates);
    }
    return e;
}

void og_env_free(og_env *e) {
    if (!e) return;
    free(e->gates);
    free(e->mat);
    free(e->perm);
    if (e->has_net) net_free(&e->net);
    metrics_free(&e->metrics);
    free(e->solution.v);
    free(e->solution_inv.v);
    free(e->qubit_perms);
    free(e->act_perms);
    free(e);
}

int64_t og_env_twists(const og_env *e, int64_t *obs_out, int64_t *act_out) {
    if (!e->add_perms) return 0; /* clifford.rs:218-222, linear_function.rs:194-198, permutation.rs:67-71 */
    return og_twists(e->kind, e->n, e->gates, e->n_gates, obs_out, act_out);
}

static int env_solved(const og_env *e) {
    switch (e->kind) {
    case OG_ENV_CLIFFORD:
    case OG_ENV_LINEAR_FUNCTION: return mat_solved(e->mat, e->dim);
    case OG_ENV_PERMUTATION: /* permutation.rs:122-128 */
        for (size_t i = 0; i < e->n; i++)
            if (e->perm[i] != i) return 0;
        return 1;
    case OG_ENV_PAULI: return net_solved(&e->net);
    }
    return 0;
}

/* clifford.rs:272-283 / linear_function.rs:245-256 / permutation.rs:134-144 */
static void reset_internals(og_env *e) {
    e->success = env_solved(e);
    metrics_reset(&e->metrics);
    e->metrics_values = metrics_snapshot(&e->metrics);
    e->reward_value = e->success ? 1.0f : 0.0f;
    e->inverted = 0;
    if (e->track_solution) {
        e->solution.len = 0;
        e->solution_inv.len = 0;
    }
}

size_t og_env_num_actions(const og_env *e) { return e->n_gates; }
size_t og_env_obs_shape(const og_env *e, size_t out[2]) {
    switch (e->kind) {
    case OG_ENV_CLIFFORD: out[0] = out[1] = 2 * e->n; break;          /* clifford.rs:291-294 */
    case OG_ENV_PAULI:                                                   /* pauli.rs:505-507 */
        out[0] = 2 * e->n;
        out[1] = 2 * e->n + e->max_rotations;
        break;
    default: out[0] = out[1] = e->n; break; /* linear_function.rs:267-269, permutation.rs:156-158 */
    }
    return 2;
}
void og_env_set_difficulty(og_env *e, size_t d) { e->difficulty = d; }
size_t og_env_get_difficulty(const og_env *e) { return e->difficulty; }

static int pauli_finish_rebuild(og_env *e) {
    /* pauli.rs:544-551 (set_state) -- also the tail of reset() after its own depth rule */
    e->success = net_solved(&e->net);
    metrics_reset(&e->metrics);
    e->metrics_values = metrics_snapshot(&e->metrics);
    e->reward_value = e->success ? 1.0f : 0.0f;
    if (e->track_solution) e->solution.len = 0;
    return 0;
}

int og_env_set_state(og_env *e, const int64_t *st, size_t n) {
    switch (e->kind) {
    case OG_ENV_CLIFFORD:      /* clifford.rs:299-304 */
    case OG_ENV_LINEAR_FUNCTION: /* linear_function.rs:279-283 */
        /* `self.cf.data = state.iter().map(|&x| x > 0).collect()` replaces the Vec whatever its
         * length; a wrong length makes later indexing panic, so the oracle rejects it here. */
        if (n != e->dim * e->dim) return fail("set_state: wrong length (reference would index out of bounds)");
        for (size_t i = 0; i < n; i++) e->mat[i] = st[i] > 0;
        e->depth = e->max_depth;
        reset_internals(e);
        return 0;
    case OG_ENV_PERMUTATION: /* permutation.rs:168-173: `x as usize` */
        if (n != e->n) return fail("set_state: wrong length");
        for (size_t i = 0; i < n; i++) {
            if (st[i] < 0 || (size_t)st[i] >= e->n) return fail("set_state: entry out of range (reference would index out of bounds)");
            e->perm[i] = (size_t)st[i];
        }
        e->depth = e->max_depth;
        reset_internals(e);
        return 0;
    case OG_ENV_PAULI: { /* pauli.rs:517-552 */
        if (n == 0) return 0;
        size_t pos = 0;
        int64_t rc0 = st[pos++];
        size_t rotation_count = rc0 > 0 ? (size_t)rc0 : 0;
        size_t tl = 4 * e->n * e->n;
        uint8_t *tab = (uint8_t *)malloc(tl ? tl : 1);
        for (size_t i = 0; i < tl; i++) {
            int64_t v = pos < n ? st[pos++] : 0; /* unwrap_or(0) */
            tab[i] = v > 0;
        }
        size_t lab_cap = n + rotation_count + 1;
        char *labels = (char *)malloc(lab_cap);
        size_t lp = 0, kept = 0;
        for (size_t idx = 0; idx < rotation_count; idx++) {
            int64_t l0 = pos < n ? st[pos++] : 0;
            size_t len = l0 > 0 ? (size_t)l0 : 0;
            size_t start = lp;
            for (size_t k = 0; k < len; k++) {
                if (pos >= n) { /* :534 expect() */
                    free(tab);
                    free(labels);
                    return fail("malformed state: not enough characters for rotation string");
                }
                int64_t ch = st[pos++];
                if (ch <= 0 || ch > 127) { /* non-ASCII can never match the label regex */
                    free(tab);
                    free(labels);
                    return fail("malformed state: invalid character code");
                }
                labels[lp++] = (char)ch;
            }
            labels[lp++] = '\0';
            if (idx < e->max_rotations) kept++; /* :538-540 */
            else lp = start;
        }
        og_net nw;
        int rc = net_new(&nw, e->n, tab, labels, kept);
        free(tab);
        free(labels);
        if (rc) return -1;
        net_free(&e->net);
        e->net = nw;
        e->depth = e->max_depth;
        return pauli_finish_rebuild(e);
    }
    }
    return -1;
}

int og_env_reset_with(og_env *e, const int64_t *actions, size_t n) {
    if (n != e->difficulty) return fail("reset_with: need exactly `difficulty` action draws");
    switch (e->kind) {
    case OG_ENV_CLIFFORD: /* clifford.rs:306-319 */
        mat_identity(e->mat, e->dim);
        for (size_t i = 0; i < n; i++)
            if (actions[i] >= 0 && (size_t)actions[i] < e->n_gates) clifford_apply(e->mat, e->n, &e->gates[actions[i]]);
        break;
    case OG_ENV_LINEAR_FUNCTION: /* linear_function.rs:285-300 */
        mat_identity(e->mat, e->dim);
        for (size_t i = 0; i < n; i++)
            if (actions[i] >= 0 && (size_t)actions[i] < e->n_gates) lf_apply(e->mat, e->n, &e->gates[actions[i]]);
        break;
    case OG_ENV_PERMUTATION: /* permutation.rs:175-192 */
        for (size_t i = 0; i < e->n; i++) e->perm[i] = i;
        for (size_t i = 0; i < n; i++) {
            if (actions[i] < 0 || (size_t)actions[i] >= e->n_gates) return fail("reset: action out of range");
            const og_gate *g = &e->gates[actions[i]];
            if (g->kind == OG_SWAP) {
                size_t t = e->perm[g->q0];
                e->perm[g->q0] = e->perm[g->q1];
                e->perm[g->q1] = t;
            }
        }
        break;
    default: return fail("reset_with: use og_pauli_reset_from for PauliEnv");
    }
    size_t d = e->depth_slope * e->difficulty;
    e->depth = d < e->max_depth ? d : e->max_depth;
    reset_internals(e);
    return 0;
}

int og_pauli_reset_from(og_env *e, const uint8_t *tableau, const char *labels, size_t n_rot) {
    if (e->kind != OG_ENV_PAULI) return fail("not a PauliEnv");
    og_net nw;
    if (net_new(&nw, e->n, tableau, labels, n_rot)) return -1; /* pauli.rs:573 */
    net_free(&e->net);
    e->net = nw;
    og_removed *tmp = (og_removed *)malloc(sizeof(og_removed) * (n_rot ? n_rot : 1));
    size_t cnt = 0;
    int rc = net_clean(&e->net, tmp, &cnt); /* pauli.rs:576 */
    free(tmp);
    if (rc) return rc;
    size_t d = e->depth_slope * e->difficulty;
    e->depth = d < e->max_depth ? d : e->max_depth; /* :578 */
    return pauli_finish_rebuild(e);                  /* :579-585 */
}


/* ------------------------------------------------------------------------------------------
 * PauliEnv::reset target generator (rust/src/envs/pauli.rs:54-271, 554-586).
 * The reference draws from rand::thread_rng(); here every draw comes from the counter RNG, two streams per
 * env (the same ones libqgym uses): the rotation labels (generate_paulis_with_difficulty) take
 * `rng_draw(seed ^ 0x7061756C, env_index, k)`, k = 0,1,2,..., and random_clifford_tableau's gate `it` takes
 * `rng_draw(seed ^ 0x7461626C, env_index, 2 it)` (its kind) and `(.., 2 it + 1)` (which pair / qubit) --
 * thread_rng's draws are i.i.d., so which draw feeds which decision leaves the distribution unchanged -- with
 *   gen_range(0..n) = mulhi64(u, n)        gen::<f32>() = (u >> 40) * 2^-24.
 * `for q in &qubits` iterates a HashSet in the reference (random order); the per-qubit axis draws
 * are i.i.d., so iterating in ascending order leaves the distribution unchanged.
 * ---------------------------------------------------------------------------------------- */
static uint64_t og_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
typedef struct {
    uint64_t seed, env, k;
} og_rng;
static uint64_t og_next(og_rng *r) {
    uint64_t v = og_splitmix64(r->seed ^ og_splitmix64(r->env * 0x9E3779B97F4A7C15ull + r->k));
    r->k++;
    return v;
}
static size_t og_range(og_rng *r, size_t n) { return (size_t)(((unsigned __int128)og_next(r) * (unsigned __int128)n) >> 64); }
static float og_f32(og_rng *r) { return (float)(og_next(r) >> 40) * (1.0f / 16777216.0f); }

int og_pauli_reset_seeded(og_env *e, uint64_t seed, uint64_t env_index) {
    if (e->kind != OG_ENV_PAULI) return fail("not a PauliEnv");
    const size_t n = e->n;
    og_rng rng = {seed ^ 0x7061756Cull, env_index, 0};
    /* valid_pairs: the CX gates of the gateset, in order (pauli.rs:360-366) */
    size_t np = 0;
    size_t *pa = (size_t *)malloc(sizeof(size_t) * (e->n_gates ? e->n_gates : 1) * 2);
    for (size_t i = 0; i < e->n_gates; i++)
        if (e->gates[i].kind == OG_CX) {
            pa[2 * np] = (size_t)e->gates[i].q0;
            pa[2 * np + 1] = (size_t)e->gates[i].q1;
            np++;
        }
    /* compute_graph_distances (pauli.rs:56-91): BFS over the undirected CX graph */
    uint8_t *adj = (uint8_t *)calloc(n * n ? n * n : 1, 1);
    for (size_t i = 0; i < np; i++) {
        adj[pa[2 * i] * n + pa[2 * i + 1]] = 1;
        adj[pa[2 * i + 1] * n + pa[2 * i]] = 1;
    }
    long *dist = (long *)malloc(sizeof(long) * (n * n ? n * n : 1));
    for (size_t i = 0; i < n * n; i++) dist[i] = -1;
    size_t *queue = (size_t *)malloc(sizeof(size_t) * (n ? n : 1));
    for (size_t st = 0; st < n; st++) {
        size_t qh = 0, qt = 0;
        queue[qt++] = st;
        dist[st * n + st] = 0;
        while (qh < qt) {
            size_t u = queue[qh++];
            for (size_t w = 0; w < n; w++)
                if (adj[u * n + w] && dist[st * n + w] < 0) {
                    dist[st * n + w] = dist[st * n + u] + 1;
                    queue[qt++] = w;
                }
        }
    }
    /* build_dist_pairs (pauli.rs:95-111): pairs q1<q2 grouped by distance, distances ascending */
    size_t maxd = 0;
    for (size_t a = 0; a < n; a++)
        for (size_t b = a + 1; b < n; b++)
            if (dist[a * n + b] > (long)maxd) maxd = (size_t)dist[a * n + b];
    /* pairs of distance d listed in (q1, q2) lexicographic order, as the nested loops push them */
    const size_t pauli_difficulty = e->difficulty / e->pauli_diff_scale; /* pauli.rs:557 */
    const size_t max_paulis = e->final_pauli_layers;
    char *labels = (char *)malloc((n + 1) * (max_paulis ? max_paulis : 1) + 1);
    size_t n_lab = 0, lp = 0;
    size_t remaining = pauli_difficulty;
    uint8_t *inset = (uint8_t *)malloc(n ? n : 1);
    size_t *cand = (size_t *)malloc(sizeof(size_t) * (n * n ? n * n : 1) * 2);
    size_t *vd = (size_t *)malloc(sizeof(size_t) * (maxd + 2));
    while (remaining > 0 && n_lab < max_paulis) { /* generate_paulis_with_difficulty (pauli.rs:191-213) */
        /* get_pauli_under_diff(remaining) (pauli.rs:115-188) */
        const size_t difficulty = remaining;
        size_t nvd = 0;
        for (size_t d = 1; d <= maxd; d++) {
            int present = 0;
            for (size_t a = 0; a < n && !present; a++)
                for (size_t b = a + 1; b < n; b++)
                    if (dist[a * n + b] == (long)d) { present = 1; break; }
            if (present && d <= difficulty) vd[nvd++] = d;
        }
        if (nvd == 0) break; /* None */
        memset(inset, 0, n ? n : 1);
        size_t pauli_diff = difficulty;
        size_t next_dif = vd[og_range(&rng, nvd)];
        size_t nc = 0;
        for (size_t a = 0; a < n; a++)
            for (size_t b = a + 1; b < n; b++)
                if (dist[a * n + b] == (long)next_dif) { cand[2 * nc] = a; cand[2 * nc + 1] = b; nc++; }
        size_t pick = og_range(&rng, nc);
        inset[cand[2 * pick]] = 1;
        inset[cand[2 * pick + 1]] = 1;
        pauli_diff = sat_sub(pauli_diff, next_dif);
        for (;;) {
            size_t nv2 = 0, rem_q = 0;
            for (size_t i = 0; i < nvd; i++) if (vd[i] <= pauli_diff) nv2++;
            for (size_t q = 0; q < n; q++) if (!inset[q]) rem_q++;
            if (pauli_diff == 0 || nv2 == 0 || rem_q == 0) break;
            if (og_f32(&rng) <= e->num_qubits_decay) break; /* continue with probability 1 - decay */
            next_dif = vd[og_range(&rng, nv2)]; /* valid_diffs is a prefix of the ascending valid_dists */
            nc = 0;
            for (size_t a = 0; a < n; a++)
                for (size_t b = a + 1; b < n; b++)
                    if (dist[a * n + b] == (long)next_dif && (inset[a] || inset[b])) { cand[2 * nc] = a; cand[2 * nc + 1] = b; nc++; }
            if (nc == 0) continue;
            pick = og_range(&rng, nc);
            inset[cand[2 * pick]] = 1;
            inset[cand[2 * pick + 1]] = 1;
            pauli_diff = sat_sub(pauli_diff, next_dif);
        }
        for (size_t q = 0; q < n; q++) labels[lp + q] = inset[q] ? "XYZ"[og_range(&rng, 3)] : 'I';
        labels[lp + n] = '\0';
        lp += n + 1;
        n_lab++;
        const size_t cost = difficulty - pauli_diff;
        remaining = sat_sub(remaining, cost > 1 ? cost : 1);
    }
    labels[lp] = '\0';
    /* random_clifford_tableau (pauli.rs:220-271) */
    const size_t dim = 2 * n;
    uint8_t *tab = (uint8_t *)malloc(dim * dim ? dim * dim : 1);
    mat_identity(tab, dim);
    if (e->difficulty != 0 && np != 0) {
        rng.seed = seed ^ 0x7461626Cull; /* the tableau's own stream: gate `it` at k = 2 it, 2 it + 1 */
        for (size_t it = 0; it < e->difficulty; it++) {
            rng.k = 2 * (uint64_t)it;
            const float r = og_f32(&rng);
            if (r > 0.3f) {
                const size_t k = og_range(&rng, np), q0 = pa[2 * k], q1 = pa[2 * k + 1];
                for (size_t c = 0; c < dim; c++) tab[q1 * dim + c] ^= tab[q0 * dim + c];
                for (size_t c = 0; c < dim; c++) tab[(n + q0) * dim + c] ^= tab[(n + q1) * dim + c];
            } else if (r > 0.15f) {
                const size_t q = og_range(&rng, n);
                swap_rows(tab, dim, q, n + q);
            } else {
                const size_t q = og_range(&rng, n);
                for (size_t c = 0; c < dim; c++) tab[(n + q) * dim + c] ^= tab[q * dim + c];
            }
        }
    }
    int rc = og_pauli_reset_from(e, tab, labels, n_lab);
    free(pa); free(adj); free(dist); free(queue); free(labels); free(inset); free(cand); free(vd); free(tab);
    return rc;
}

int og_pauli_set_perms(og_env *e, const int64_t *qp, const int64_t *ap, size_t n_perms) {
    if (e->kind != OG_ENV_PAULI) return fail("not a PauliEnv");
    free(e->qubit_perms);
    free(e->act_perms);
    e->qubit_perms = e->act_perms = NULL;
    e->n_perms = n_perms;
    e->current_perm_idx = 0;
    if (!n_perms) return 0;
    e->qubit_perms = (size_t *)malloc(sizeof(size_t) * n_perms * e->n);
    e->act_perms = (size_t *)malloc(sizeof(size_t) * n_perms * e->n_gates);
    for (size_t i = 0; i < n_perms * e->n; i++) e->qubit_perms[i] = (size_t)qp[i];
    for (size_t i = 0; i < n_perms * e->n_gates; i++) e->act_perms[i] = (size_t)ap[i];
    return 0;
}

static int gate_in_range(const og_env *e, const og_gate *g) {
    if (g->q0 < 0 || (size_t)g->q0 >= e->n) return 0;
    if (g->kind >= OG_CX && (g->q1 < 0 || (size_t)g->q1 >= e->n)) return 0;
    return 1;
}

int og_env_step(og_env *e, size_t action, int coin) {
    volatile float penalty = 0.0f;
    if (e->kind == OG_ENV_PAULI) { /* pauli.rs:588-635 */
        size_t new_rotations = 0;
        size_t actual = action;
        if (e->n_perms) { /* :594-599 */
            if (action >= e->n_gates) return fail("act_perms index out of bounds");
            actual = e->act_perms[e->current_perm_idx * e->n_gates + action];
        }
        if (actual < e->n_gates) {
            const og_gate *g = &e->gates[actual];
            og_counts prev = e->metrics_values;
            metrics_apply_gate(&e->metrics, g);
            og_counts now = metrics_snapshot(&e->metrics);
            penalty = weighted_delta(&now, &prev, e->w);
            e->metrics_values = now;
            if (!gate_in_range(e, g)) return fail("gate qubit out of range (reference would index out of bounds)");
            size_t cap = 3 * (e->net.R ? e->net.R : 1);
            og_removed *rem = (og_removed *)malloc(sizeof(og_removed) * cap);
            size_t cnt = 0;
            int rc = net_act(&e->net, g, rem, &cnt);
            if (rc) {
                free(rem);
                return rc;
            }
            new_rotations = cnt;
            if (e->track_solution) { /* :612-626 */
                vec_push(&e->solution, (uint64_t)actual);
                for (size_t i = 0; i < cnt; i++) {
                    int32_t ph = pauli_phase(&e->net.rot[rem[i].index]); /* read AFTER act() (:618) */
                    uint64_t phase_code = (ph == 2) ? 0u : 1u;           /* phase_mult -1 -> 0, +1 -> 1 */
                    /* encoding of solution() (:698-716) */
                    vec_push(&e->solution, 0x80000000ull | ((uint64_t)rem[i].axis << 21) | ((uint64_t)rem[i].qubit << 11) |
                                               ((uint64_t)rem[i].index << 1) | phase_code);
                }
            }
            free(rem);
        }
        e->depth = sat_sub(e->depth, 1);
        e->success = net_solved(&e->net);
        volatile float achieved = e->success ? 1.0f : 0.0f;
        volatile float t = achieved - penalty;
        volatile float bonus = e->pauli_layer_reward * (float)new_rotations;
        e->reward_value = t + bonus; /* :634 */
        return 0;
    }
    if (e->kind == OG_ENV_PERMUTATION) { /* permutation.rs:194-225 */
        if (action < e->n_gates) {
            const og_gate *g = &e->gates[action];
            og_counts prev = e->metrics_values;
            metrics_apply_gate(&e->metrics, g);
            og_counts now = metrics_snapshot(&e->metrics);
            penalty = weighted_delta(&now, &prev, e->w);
            e->metrics_values = now;
            if (g->kind == OG_SWAP) {
                if (!gate_in_range(e, g)) return fail("gate qubit out of range");
                size_t t = e->perm[g->q0];
                e->perm[g->q0] = e->perm[g->q1];
                e->perm[g->q1] = t;
            }
            if (e->track_solution) vec_push(e->inverted ? &e->solution_inv : &e->solution, action); /* inside the branch, :210-216 */
        }
        if (e->add_inverts && coin) { /* :110-120, BEFORE the depth decrement (:219-221) */
            size_t *inv = (size_t *)malloc(sizeof(size_t) * (e->n ? e->n : 1));
            for (size_t i = 0; i < e->n; i++) inv[e->perm[i]] = i; /* :101-107 */
            memcpy(e->perm, inv, sizeof(size_t) * e->n);
            free(inv);
            e->inverted = !e->inverted;
        }
        e->depth = sat_sub(e->depth, 1);
        e->success = env_solved(e);
        volatile float achieved = e->success ? 1.0f : 0.0f;
        e->reward_value = achieved - penalty;
        return 0;
    }
    /* Clifford (clifford.rs:321-347) and LinearFunction (linear_function.rs:302-328) */
    if (action < e->n_gates) {
        const og_gate *g = &e->gates[action];
        og_counts prev = e->metrics_values;
        metrics_apply_gate(&e->metrics, g);
        og_counts now = metrics_snapshot(&e->metrics);
        penalty = weighted_delta(&now, &prev, e->w);
        e->metrics_values = now;
        if (e->kind == OG_ENV_CLIFFORD) {
            if (!gate_in_range(e, g)) return fail("gate qubit out of range");
            clifford_apply(e->mat, e->n, g);
        } else {
            if ((g->kind == OG_CX || g->kind == OG_SWAP) && !gate_in_range(e, g)) return fail("gate qubit out of range");
            lf_apply(e->mat, e->n, g);
        }
    }
    if (e->track_solution) vec_push(e->inverted ? &e->solution_inv : &e->solution, action); /* :334-340, regardless of validity */
    e->depth = sat_sub(e->depth, 1);                                                         /* :342 */
    if (e->add_inverts && coin) {                                                            /* :262-270 */
        uint8_t *inv = (uint8_t *)malloc(e->dim * e->dim ? e->dim * e->dim : 1);
        if (mat_inverse(e->mat, e->dim, inv)) {
            free(inv);
            return -1;
        }
        memcpy(e->mat, inv, e->dim * e->dim);
        free(inv);
        e->inverted = !e->inverted;
    }
    e->success = env_solved(e); /* :344 */
    volatile float achieved = e->success ? 1.0f : 0.0f;
    e->reward_value = achieved - penalty; /* :345-346 */
    return 0;
}

size_t og_env_masks(const og_env *e, uint8_t *out, size_t cap) { /* clifford.rs:349-351 */
    for (size_t i = 0; i < e->n_gates && i < cap; i++) out[i] = !e->success;
    return e->n_gates;
}
int og_env_is_final(const og_env *e) { return e->depth == 0 || e->success; } /* clifford.rs:353 */
float og_env_reward(const og_env *e) { return e->reward_value; }
int og_env_success(const og_env *e) { return e->success; }
int og_env_track_solution(const og_env *e) { return e->track_solution; }
size_t og_env_depth(const og_env *e) { return e->depth; }
int og_env_inverted(const og_env *e) { return e->inverted; }
void og_env_metrics(const og_env *e, size_t out[4]) {
    out[0] = e->metrics_values.n_cnots;
    out[1] = e->metrics_values.n_layers_cnots;
    out[2] = e->metrics_values.n_layers;
    out[3] = e->metrics_values.n_gates;
}

/* pauli.rs:411-437 */
static void pauli_pad_and_collect(const og_env *e, uint8_t *dense) {
    size_t rows = 2 * e->n, max_cols = 2 * e->n + e->max_rotations;
    memset(dense, 0, rows * max_cols);
    for (size_t c = 0; c < 2 * e->n; c++)
        for (size_t r = 0; r < rows; r++) dense[r * max_cols + c] = NET(&e->net, r, c);
    for (size_t i = 0; i < e->net.dag.count; i++) { /* active_rotation_indices(), DAG node order */
        if (i >= e->max_rotations) break;
        size_t src = 2 * e->n + e->net.dag.nodes[i], dst = 2 * e->n + i;
        for (size_t r = 0; r < rows; r++) dense[r * max_cols + dst] = NET(&e->net, r, src);
    }
}
/* pauli.rs:445-485 */
static void pauli_apply_perm(const og_env *e, const uint8_t *dense, const size_t *perm, uint8_t *result) {
    size_t n = e->n, rows = 2 * n, tc = 2 * n + e->max_rotations;
    uint8_t *temp = (uint8_t *)calloc(rows * tc ? rows * tc : 1, 1);
    memset(result, 0, rows * tc);
    for (size_t i = 0; i < n; i++)
        for (size_t c = 0; c < tc; c++) {
            temp[i * tc + c] = dense[perm[i] * tc + c];
            temp[(n + i) * tc + c] = dense[(n + perm[i]) * tc + c];
        }
    for (size_t r = 0; r < rows; r++) {
        for (size_t i = 0; i < n; i++) {
            result[r * tc + i] = temp[r * tc + perm[i]];
            result[r * tc + n + i] = temp[r * tc + n + perm[i]];
        }
        for (size_t c = 2 * n; c < tc; c++) result[r * tc + c] = temp[r * tc + c];
    }
    free(temp);
}

size_t og_env_observe(og_env *e, int64_t *out, size_t cap, size_t perm_idx) {
    size_t cnt = 0;
    switch (e->kind) {
    case OG_ENV_CLIFFORD:        /* clifford.rs:361-368 */
    case OG_ENV_LINEAR_FUNCTION: /* linear_function.rs:346-351 */
        for (size_t i = 0; i < e->dim * e->dim; i++)
            if (e->mat[i]) {
                if (cnt < cap) out[cnt] = (int64_t)i;
                cnt++;
            }
        return cnt;
    case OG_ENV_PERMUTATION: /* permutation.rs:241-243 */
        for (size_t i = 0; i < e->n; i++) {
            if (cnt < cap) out[cnt] = (int64_t)(i * e->n + e->perm[i]);
            cnt++;
        }
        return cnt;
    case OG_ENV_PAULI: { /* pauli.rs:653-673 */
        size_t rows = 2 * e->n, tc = 2 * e->n + e->max_rotations;
        uint8_t *dense = (uint8_t *)malloc(rows * tc ? rows * tc : 1);
        pauli_pad_and_collect(e, dense);
        if (e->n_perms) {
            size_t idx = perm_idx % e->n_perms;
            e->current_perm_idx = idx; /* :661 */
            uint8_t *res = (uint8_t *)malloc(rows * tc ? rows * tc : 1);
            pauli_apply_perm(e, dense, e->qubit_perms + idx * e->n, res);
            free(dense);
            dense = res;
        }
        for (size_t i = 0; i < rows * tc; i++)
            if (dense[i] > 0) {
                if (cnt < cap) out[cnt] = (int64_t)i;
                cnt++;
            }
        free(dense);
        return cnt;
    }
    }
    return 0;
}

size_t og_env_solution(const og_env *e, uint64_t *out, size_t cap) {
    size_t cnt = 0;
    for (size_t i = 0; i < e->solution.len; i++) { /* clifford.rs:376-381; pauli.rs:700-718 */
        if (cnt < cap) out[cnt] = e->solution.v[i];
        cnt++;
    }
    if (e->kind != OG_ENV_PAULI)
        for (size_t i = e->solution_inv.len; i-- > 0;) {
            if (cnt < cap) out[cnt] = e->solution_inv.v[i];
            cnt++;
        }
    return cnt;
}

size_t og_env_get_state_i64(const og_env *e, int64_t *out, size_t cap) {
    size_t cnt = 0;
    if (e->kind == OG_ENV_PERMUTATION) {
        for (size_t i = 0; i < e->n && cnt < cap; i++) out[cnt++] = (int64_t)e->perm[i];
    } else if (e->kind == OG_ENV_PAULI) {
        for (size_t r = 0; r < 2 * e->n; r++)
            for (size_t c = 0; c < 2 * e->n && cnt < cap; c++) out[cnt++] = NET(&e->net, r, c);
    } else {
        for (size_t i = 0; i < e->dim * e->dim && cnt < cap; i++) out[cnt++] = e->mat[i];
    }
    return cnt;
}

size_t og_pauli_active_rotations(const og_env *e, int64_t *out, size_t cap) {
    if (e->kind != OG_ENV_PAULI) return 0;
    for (size_t i = 0; i < e->net.dag.count && i < cap; i++) out[i] = (int64_t)e->net.dag.nodes[i];
    return e->net.dag.count;
}
size_t og_pauli_num_rotations(const og_env *e) { return e->kind == OG_ENV_PAULI ? e->net.R : 0; }
int og_pauli_rotation(const og_env *e, size_t k, uint8_t *x, uint8_t *z, int32_t *base_phase, int32_t *phase) {
    if (e->kind != OG_ENV_PAULI || k >= e->net.R) return fail("bad rotation index");
    memcpy(x, e->net.rot[k].base_x, e->n);
    memcpy(z, e->net.rot[k].base_z, e->n);
    *base_phase = e->net.rot[k].base_phase;
    *phase = pauli_phase(&e->net.rot[k]);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Batched driver
 * ---------------------------------------------------------------------------------------- */
struct og_vec {
    og_env **envs;
    size_t batch;
};

og_vec *og_vec_new(const og_env *proto, size_t batch) {
    og_vec *v = (og_vec *)malloc(sizeof *v);
    v->batch = batch;
    v->envs = (og_env **)malloc(sizeof(og_env *) * (batch ? batch : 1));
    for (size_t i = 0; i < batch; i++) v->envs[i] = og_env_clone(proto);
    return v;
}
void og_vec_free(og_vec *v) {
    if (!v) return;
    for (size_t i = 0; i < v->batch; i++) og_env_free(v->envs[i]);
    free(v->envs);
    free(v);
}
og_env *og_vec_env(og_vec *v, size_t i) { return i < v->batch ? v->envs[i] : NULL; }

int og_vec_set_state(og_vec *v, const int64_t *states, size_t per_env) {
    int rc = 0;
    for (size_t i = 0; i < v->batch; i++)
        if (og_env_set_state(v->envs[i], states + i * per_env, per_env)) rc = -1;
    return rc;
}
int og_vec_reset_with(og_vec *v, const int64_t *actions, size_t n_steps) {
    int rc = 0;
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (n_steps ? n_steps : 1));
    for (size_t i = 0; i < v->batch; i++) {
        for (size_t t = 0; t < n_steps; t++) tmp[t] = actions[t * v->batch + i];
        if (og_env_reset_with(v->envs[i], tmp, n_steps)) rc = -1;
    }
    free(tmp);
    return rc;
}
int og_vec_step(og_vec *v, const int32_t *actions, const uint8_t *coins, float *reward, uint8_t *success,
                uint8_t *is_final, int32_t *depth, int threads) {
    int rc = 0;
    long B = (long)v->batch;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads) reduction(| : rc)
#endif
    for (long i = 0; i < B; i++) {
        og_env *e = v->envs[i];
        /* a negative action is out of range: `usize` cannot hold it, the caller would have wrapped */
        size_t a = actions[i] < 0 ? (size_t)-1 : (size_t)actions[i];
        if (og_env_step(e, a, coins ? coins[i] : 0)) rc |= 1;
        if (reward) reward[i] = e->reward_value;
        if (success) success[i] = (uint8_t)e->success;
        if (is_final) is_final[i] = (uint8_t)og_env_is_final(e);
        if (depth) depth[i] = (int32_t)e->depth;
    }
    (void)threads;
    return rc ? -1 : 0;
}
int og_vec_observe_dense(og_vec *v, int8_t *out, int threads) {
    if (!v->batch) return 0;
    size_t shp[2];
    og_env_obs_shape(v->envs[0], shp);
    size_t per = shp[0] * shp[1];
    long B = (long)v->batch;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
#endif
    for (long i = 0; i < B; i++) {
        int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (per ? per : 1));
        size_t cnt = og_env_observe(v->envs[i], idx, per, 0);
        int8_t *o = out + (size_t)i * per; /* adapters.py:50-54: zeros; full[observe()] = 1 */
        memset(o, 0, per);
        for (size_t k = 0; k < cnt; k++) o[idx[k]] = 1;
        free(idx);
    }
    (void)threads;
    return 0;
}
int og_vec_get_state(og_vec *v, int64_t *out, size_t per_env) {
    int bad = 0;
    long B = (long)v->batch;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(| : bad)
#endif
    for (long i = 0; i < B; i++)
        if (og_env_get_state_i64(v->envs[i], out + (size_t)i * per_env, per_env) != per_env) bad |= 1;
    return bad ? fail("get_state: size mismatch") : 0;
}

/* --- batched conveniences for the full-size parity tests (no reference counterpart: the draws of Env::reset made explicit
 * through the counter RNG the HIP path uses, qgym_internal.hpp rng_draw).  env i of the batch is GLOBAL env env_ids[i] (or
 * env_base + i); `mask` (or NULL = all) selects the envs to reset. --- */
int og_vec_reset_seeded(og_vec *v, uint64_t seed, uint64_t env_base, const uint64_t *env_ids, const uint8_t *mask, int threads) {
    int rc = 0;
    long B = (long)v->batch;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads) reduction(| : rc)
#endif
    for (long i = 0; i < B; i++) {
        if (mask && !mask[i]) continue;
        og_env *e = v->envs[i];
        const size_t n = e->difficulty;
        int64_t *draws = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
        og_rng rng = {seed, env_ids ? env_ids[i] : env_base + (uint64_t)i, 0};
        for (size_t t = 0; t < n; t++) draws[t] = (int64_t)og_range(&rng, e->n_gates); /* clifford.rs:311-316 gen_range(0..num_actions) */
        if (og_env_reset_with(e, draws, n)) rc |= 1;
        free(draws);
    }
    (void)threads;
    return rc ? -1 : 0;
}
int og_vec_pauli_reset_seeded(og_vec *v, uint64_t seed, uint64_t env_base, const uint64_t *env_ids, const uint8_t *mask, int threads) {
    int rc = 0;
    long B = (long)v->batch;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads) reduction(| : rc)
#endif
    for (long i = 0; i < B; i++) {
        if (mask && !mask[i]) continue;
        if (og_pauli_reset_seeded(v->envs[i], seed, env_ids ? env_ids[i] : env_base + (uint64_t)i)) rc |= 1;
    }
    (void)threads;
    return rc ? -1 : 0;
}
/* Env::solution of every env: out[i * cap .. ] (entries beyond cap dropped), lens[i] = the full length */
int og_vec_solutions(og_vec *v, uint64_t *out, size_t cap, int64_t *lens) {
    for (size_t i = 0; i < v->batch; i++) lens[i] = (int64_t)og_env_solution(v->envs[i], out + i * cap, cap);
    return 0;
}
