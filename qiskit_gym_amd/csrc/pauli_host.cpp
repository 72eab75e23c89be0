// pauli_host.cpp -- PauliEnv (PauliNetworkGym): host side of the batched handle.
//
// Reference semantics (paths relative to the reference repo):
//   PauliEnv::new / set_state / reset tail               rust/src/envs/pauli.rs:339-409, 517-552, 573-585
//   PauliNetwork::new                                    rust/src/pauli/pauli_network.rs:37-77
//   Pauli::{from_label, commutes_with}                   rust/src/pauli/pauli.rs:48-81, 112-123
//   PauliDag::new                                        rust/src/pauli/pauli_dag.rs:25-45
// What lives here: parsing the trait's `set_state(Vec<i64>)` wire format and explicit targets into per-env
// records (tableau rows, rotation masks, base phases, DAG predecessor masks), and handing them to the
// device layout of kernels_pauli_tile.hip.  Stepping, cleaning, observing and reset()'s random target
// generator all run on the GPU (kernels_pauli_tile.hip); nothing here computes an env step.
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "pauli_common.hpp"

namespace qg {

int pauli_alloc(qg_vec *v) {
    int rc = ptile_alloc(v);
    if (rc) return rc;
    if (v->cfg.add_perms) {  // compute_qubit_perms (symmetry.rs:307-361)
        std::vector<std::vector<int64_t>> qp, ap;
        compute_qubit_and_action_perms(v->N, v->gates, qp, ap);
        if (!qp.empty()) {
            std::vector<uint8_t> hq(qp.size() * v->N);
            std::vector<int32_t> ha(qp.size() * std::max<size_t>(v->gates.size(), 1));
            for (size_t i = 0; i < qp.size(); ++i) {
                for (uint32_t q = 0; q < v->N; ++q) hq[i * v->N + q] = (uint8_t)qp[i][q];
                for (size_t g = 0; g < v->gates.size(); ++g) ha[i * v->gates.size() + g] = (int32_t)ap[i][g];
            }
            HIP_TRY(hipMalloc(&v->d_qubit_perms, hq.size()));
            HIP_TRY(hipMalloc(&v->d_act_perms, sizeof(int32_t) * ha.size()));
            HIP_TRY(hipMemcpy(v->d_qubit_perms, hq.data(), hq.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(v->d_act_perms, ha.data(), sizeof(int32_t) * ha.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMalloc(&v->perm_idx, sizeof(uint32_t) * v->B));
            HIP_TRY(hipMemset(v->perm_idx, 0, sizeof(uint32_t) * v->B));  // AtomicUsize::new(0) (pauli.rs:400)
            v->n_perms = (uint32_t)qp.size();
        }
    }
    return QG_OK;
}

hipError_t pauli_step(const qg_vec *v, const StepArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    return ptile_step(v, a, s);
}

hipError_t pauli_export(const qg_vec *v, const ObsArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    return ptile_export(v, a, s);
}

// Pauli::from_label (pauli.rs:48-81) -> masks; returns false for an invalid label
static bool parse_label(const std::string &label, uint32_t N, PauliRot &r, std::string &why) {
    size_t p = 0;
    bool neg = false, has_i = false;
    if (p < label.size() && (label[p] == '+' || label[p] == '-')) neg = label[p++] == '-';
    if (p < label.size() && (label[p] == 'i' || label[p] == 'j')) { has_i = true; ++p; }
    else if (p < label.size() && label[p] == '1') ++p;
    const std::string s = label.substr(p);
    for (char ch : s)
        if (ch != 'I' && ch != 'X' && ch != 'Y' && ch != 'Z') { why = "Pauli string label is not valid."; return false; }
    if (s.size() != N) { why = "Number of qubits differ for Clifford and Paulis"; return false; }  // pauli_network.rs:52-58
    uint32_t phase = neg ? (has_i ? 1u : 2u) : (has_i ? 3u : 0u);  // pauli.rs:28-37
    r.x = r.z = 0;
    uint32_t ys = 0;
    for (uint32_t q = 0; q < N; ++q) {
        const char b = s[N - 1 - q];  // reversed (pauli.rs:62)
        if (b == 'X' || b == 'Y') r.x |= 1u << q;
        if (b == 'Z' || b == 'Y') r.z |= 1u << q;
        ys += (b == 'Y');
    }
    r.phase = (phase + ys) & 3u;  // pauli.rs:73
    r.pred = 0;
    return true;
}
// !commutes_with (pauli.rs:112-123): parity of the symplectic product
static bool anticommute(const PauliRot &a, const PauliRot &b) { return (__builtin_popcount(a.x & b.z) + __builtin_popcount(a.z & b.x)) & 1; }

static void host_net_init(const qg_vec *v, HostNet &h) {
    h.tab.assign((size_t)v->B * v->N * 2, 0);
    h.rot.assign((size_t)v->B * v->rmax, PauliRot{0, 0, 0, 0});
    h.meta.assign(v->B, PauliMeta{});
}
// PauliNetwork::new (pauli_network.rs:37-77) for env e
static int host_net_build(const qg_vec *v, HostNet &h, uint64_t e, const uint8_t *tableau_rowmajor,
                          const std::vector<std::string> &labels) {
    const uint32_t N = v->N, D = 2 * N;
    for (uint32_t q = 0; q < N; ++q) {
        uint64_t xr = 0, zr = 0;
        for (uint32_t c = 0; c < D; ++c) {
            xr |= (uint64_t)(tableau_rowmajor[(size_t)q * D + c] != 0) << c;
            zr |= (uint64_t)(tableau_rowmajor[(size_t)(N + q) * D + c] != 0) << c;
        }
        h.tab[(e * N + q) * 2] = xr;
        h.tab[(e * N + q) * 2 + 1] = zr;
    }
    const size_t R = labels.size();
    for (size_t k = 0; k < R; ++k) {
        std::string why;
        PauliRot &r = h.rot[e * v->rmax + k];
        if (!parse_label(labels[k], N, r, why)) return set_error(QG_ERR_PANIC, "env %llu rotation %zu: %s", (unsigned long long)e, k, why.c_str());
        for (size_t k2 = 0; k2 < k; ++k2)  // PauliDag::new (pauli_dag.rs:35-41): edge k -> k2 iff they do not commute
            if (anticommute(r, h.rot[e * v->rmax + k2])) r.pred |= 1u << k2;
    }
    PauliMeta &m = h.meta[e];
    m.alive = R >= 32 ? ~0u : ((1u << R) - 1u);
    m.count = (uint32_t)R;
    memset(m.order, 0, sizeof m.order);
    for (size_t k = 0; k < R; ++k) m.order[k] = (uint8_t)k;
    return QG_OK;
}

static int host_net_upload(qg_vec *v, const HostNet &h, bool do_clean, int32_t depth_value, hipStream_t s) {
    return ptile_upload(v, h, do_clean, depth_value, s);
}

int pauli_init_identity(qg_vec *v, hipStream_t s) {  // PauliEnv::new (pauli.rs:355-357,384,404)
    HostNet h;
    host_net_init(v, h);
    const uint32_t N = v->N, D = 2 * N;
    std::vector<uint8_t> id((size_t)D * D, 0);
    for (uint32_t i = 0; i < D; ++i) id[(size_t)i * D + i] = 1;
    for (uint64_t e = 0; e < v->B; ++e) {
        int rc = host_net_build(v, h, e, id.data(), {});
        if (rc) return rc;
    }
    return host_net_upload(v, h, false, 1, s);
}

// PauliEnv::set_state (pauli.rs:517-552): wire format [rot_count, 4N^2 tableau ints, (len, chars...)*]
int pauli_set_state(qg_vec *v, const void *states, int format, size_t stride, int on_device, hipStream_t s) {
    if (format != QG_FMT_I64) return set_error(QG_ERR_UNSUPPORTED, "PauliEnv set_state takes the i64 wire format only");
    const uint32_t N = v->N, D = 2 * N;
    std::vector<int64_t> host;
    const int64_t *st = reinterpret_cast<const int64_t *>(states);
    if (on_device) {
        host.resize(stride * v->B);
        HIP_TRY(hipMemcpy(host.data(), states, sizeof(int64_t) * host.size(), hipMemcpyDeviceToHost));
        st = host.data();
    }
    if (stride == 0) return QG_OK;  // `if state.is_empty() { return; }` (:518-520)
    HostNet h;
    host_net_init(v, h);
    std::vector<uint8_t> tab((size_t)D * D);
    for (uint64_t e = 0; e < v->B; ++e) {
        const int64_t *p = st + e * stride;
        size_t pos = 0;
        auto next = [&](bool &have) -> int64_t {
            have = pos < stride;
            return have ? p[pos++] : 0;
        };
        bool have;
        int64_t rc0 = next(have);
        const size_t rotation_count = rc0 > 0 ? (size_t)rc0 : 0;
        for (size_t i = 0; i < (size_t)D * D; ++i) tab[i] = next(have) > 0;  // unwrap_or(0), > 0 => 1
        std::vector<std::string> labels;
        for (size_t idx = 0; idx < rotation_count; ++idx) {
            int64_t l0 = next(have);
            const size_t len = l0 > 0 ? (size_t)l0 : 0;
            std::string lab;
            for (size_t k = 0; k < len; ++k) {
                int64_t ch = next(have);
                if (!have) return set_error(QG_ERR_PANIC, "env %llu: malformed state: not enough characters for rotation string", (unsigned long long)e);
                if (ch <= 0 || ch > 127) return set_error(QG_ERR_PANIC, "env %llu: malformed state: invalid character code", (unsigned long long)e);
                lab.push_back((char)ch);
            }
            if (idx < (size_t)v->cfg.max_rotations) labels.push_back(lab);  // :538-540
        }
        int rc = host_net_build(v, h, e, tab.data(), labels);
        if (rc) return rc;
    }
    return host_net_upload(v, h, false, v->cfg.max_depth, s);  // :544 depth = max_depth; no clean
}

int pauli_reset_from(qg_vec *v, const uint8_t *tableaus, const char *labels, const int32_t *n_rot, hipStream_t s) {
    const uint32_t N = v->N, D = 2 * N;
    HostNet h;
    host_net_init(v, h);
    size_t lp = 0;
    for (uint64_t e = 0; e < v->B; ++e) {
        if (n_rot[e] < 0 || (uint32_t)n_rot[e] > v->rmax)
            return set_error(QG_ERR_INVALID, "env %llu: %d rotations, this batch was planned for at most %u", (unsigned long long)e, n_rot[e], v->rmax);
        std::vector<std::string> labs;
        for (int32_t k = 0; k < n_rot[e]; ++k) {
            if (!labels) return set_error(QG_ERR_INVALID, "labels is null");
            labs.emplace_back(labels + lp, N);
            lp += N;
        }
        int rc = host_net_build(v, h, e, tableaus + e * (size_t)D * D, labs);
        if (rc) return rc;
    }
    const int64_t d = (int64_t)v->cfg.depth_slope * v->difficulty;  // pauli.rs:578
    return host_net_upload(v, h, true, (int32_t)std::min<int64_t>(d, v->cfg.max_depth), s);
}


// PauliEnv::reset (pauli.rs:554-586) with its random target generator (pauli.rs:54-271): on the device
// (ptile_generate_kernel), every draw from the env's two counter-RNG streams (kernels_pauli_tile.hip PT_STREAM_LABELS / PT_STREAM_TABLEAU).
int pauli_reset_seeded(qg_vec *v, uint64_t seed, hipStream_t s) { return ptile_reset_seeded(v, seed, false, s); }

}  // namespace qg
