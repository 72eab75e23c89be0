// kernels_pauli.hip -- PauliEnv (PauliNetworkGym): Clifford tableau + Pauli-rotation tracking.
//
// Reference semantics (paths relative to the reference repo):
//   PauliEnv::step / observe / set_state / reset tail   rust/src/envs/pauli.rs:588-635, 411-437, 517-552, 573-585
//   PauliNetwork::{act,cnot,h,s,sx,clean_and_return_with_phases,solved}  rust/src/pauli/pauli_network.rs:139-260
//   Pauli::{from_label,evolve_h,evolve_s,evolve_cx,evolve_sx,commutes_with,phase}  rust/src/pauli/pauli.rs:48-133
//   PauliDag::{new,get_front_layer}                      rust/src/pauli/pauli_dag.rs:25-57
//   petgraph 0.6.5 Graph::retain_nodes/remove_node (reverse visit, Vec::swap_remove) -- third party,
//   restated from its published source; it fixes the order of the surviving DAG nodes and with it
//   the order of the observation's rotation columns.
//
// Mapping to the machine (PAULI layout): 32 lanes per env, two envs per wavefront.
//   lane q < N   owns qubit q's two tableau rows as uint64 {X row q, Z row N+q}: H/S/SX are
//                lane-local, CNOT(i,j) is two ds_bpermute shuffles;
//   lane k < R   owns rotation k as {x mask, z mask, base_phase, predecessor mask}: the reference's
//                rotation columns of `data` always equal rotation_qk's (x, z) while the rotation is
//                alive (row ops and evolve_* are the same linear maps), so the bits are kept once;
//                rotation weight is one popcount, the DAG front layer is `pred & alive == 0`, and a
//                clean pass is a wave ballot of (alive & front & weight<=1).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "device_common.hpp"
#include "pauli_common.hpp"

namespace qg {

#define PAULI_L 32u
#define PAULI_LOG2L 5u

static inline uint64_t mop(uint32_t kind, uint32_t a, uint32_t b) { return (uint64_t)(kind | (a << 4) | (b << 10)); }
static uint64_t gate_program(const qg_gate &g) {
    const uint32_t a = (uint32_t)g.q0, b = (uint32_t)g.q1;
    switch (g.kind) {
    case QG_H: return mop(M_H, a, 0);
    case QG_S: return mop(M_S, a, 0);
    case QG_SDG: return mop(M_S, a, 0) | (mop(M_S, a, 0) << 16) | (mop(M_S, a, 0) << 32);       // S applied 3x (:229-234)
    case QG_SX: return mop(M_SX, a, 0);
    case QG_SXDG: return mop(M_SX, a, 0) | (mop(M_SX, a, 0) << 16) | (mop(M_SX, a, 0) << 32);   // SX applied 3x (:236-241)
    case QG_CX: return mop(M_CNOT, a, b);                                                        // :242
    case QG_CZ: return mop(M_H, b, 0) | (mop(M_CNOT, a, b) << 16) | (mop(M_H, b, 0) << 32);      // :243-249
    case QG_SWAP: return mop(M_CNOT, a, b) | (mop(M_CNOT, b, a) << 16) | (mop(M_CNOT, a, b) << 32);  // :250-257
    }
    return 0;
}

__device__ inline uint32_t bperm32(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
__device__ inline uint64_t bperm64(uint64_t v, uint32_t src_lane) {
    uint32_t lo = bperm32((uint32_t)v, src_lane), hi = bperm32((uint32_t)(v >> 32), src_lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ inline uint32_t nib(uint64_t order, uint32_t i) { return (uint32_t)(order >> (4 * i)) & 0xFu; }

// Pauli::evolve_* on bit masks (pauli.rs:83-110)
__device__ inline void rot_h(uint32_t &x, uint32_t &z, uint32_t &ph, uint32_t q) {
    const uint32_t bx = (x >> q) & 1u, bz = (z >> q) & 1u;
    x ^= (bx ^ bz) << q;
    z ^= (bx ^ bz) << q;
    ph = (ph + 2u * (bx & bz)) & 3u;
}
__device__ inline void rot_s(uint32_t &x, uint32_t &z, uint32_t &ph, uint32_t q) {
    const uint32_t bx = (x >> q) & 1u;
    z ^= bx << q;
    ph = (ph + bx) & 3u;
}
__device__ inline void rot_cx(uint32_t &x, uint32_t &z, uint32_t qc, uint32_t qt) {
    x ^= ((x >> qc) & 1u) << qt;
    z ^= ((z >> qt) & 1u) << qc;
}
// Pauli::phase (pauli.rs:125-133)
__device__ inline uint32_t rot_phase(uint32_t x, uint32_t z, uint32_t ph, uint32_t N) {
    return (ph + 4u * N - (uint32_t)__popc(x & z)) & 3u;
}

struct PauliLane {
    uint64_t xr, zr;             // tableau rows of qubit `lie`
    uint32_t rx, rz, rph, rpred; // rotation `lie`
    uint32_t rem_code;           // (axis, qubit, index) of this rotation if it was removed this step
    uint32_t rem_seq;            // its position among this step's removals, or ~0
};

// clean_and_return_with_phases (pauli_network.rs:139-165) for the envs of this wave whose lanes
// pass `on`.  Updates meta, counts removals, tags removed rotations for the solution log.
__device__ inline void pauli_clean(PauliLane &p, PauliMeta &m, uint32_t &n_removed, uint32_t &fault, bool on, uint32_t lie,
                                   uint32_t base, uint32_t rmax, uint32_t N) {
    // a rotation's weight does not change while cleaning, only `alive` does
    const uint32_t support = p.rx | p.rz;
    const bool trivial = (lie < rmax) && __popc(support) <= 1;  // is_rotation_trivial (:79-93)
    for (;;) {
        const bool alive_k = (m.alive >> lie) & 1u;
        const bool front_k = alive_k && ((p.rpred & m.alive) == 0);  // get_front_layer (pauli_dag.rs:47-57)
        bool doomed = on && front_k && trivial;
        if (doomed && support == 0) {  // which_qubit(..).unwrap() on None (:113-114): the reference panics
            fault |= QG_FAULT_ZERO_WEIGHT;
            doomed = false;
        }
        const uint64_t bal = __ballot(doomed);
        if (bal == 0) break;  // wave-uniform: no env of this wave removed anything in this pass
        const uint32_t dmask = (uint32_t)(bal >> base);
        if (dmask) {
            if (doomed) {  // which_qubit / which_axis (:95-137) read at removal time
                const uint32_t q = (uint32_t)__ffs((int)support) - 1u;
                const uint32_t bx = (p.rx >> q) & 1u, bz = (p.rz >> q) & 1u;
                const uint32_t axis = bx ? (bz ? 1u : 0u) : 2u;  // X=0, Y=1, Z=2
                p.rem_code = 0x80000000u | (axis << 21) | (q << 11) | (lie << 1);
            }
            // removals are reported in DAG node order within a pass (:146-152)
            for (uint32_t i = 0; i < m.count; ++i) {
                const uint32_t r = nib(m.order, i);
                const uint32_t hit = (dmask >> r) & 1u;
                if (hit && lie == r) p.rem_seq = n_removed;
                n_removed += hit;
            }
            // retain_nodes: visit NodeIndex high -> low, swap_remove each doomed node (:160-161)
            for (uint32_t i = m.count; i-- > 0;) {
                if ((dmask >> nib(m.order, i)) & 1u) {
                    const uint64_t last = (uint64_t)nib(m.order, m.count - 1);
                    m.order = (m.order & ~(0xFull << (4 * i))) | (last << (4 * i));
                    m.count -= 1;
                }
            }
            m.alive &= ~dmask;
        }
    }
}

struct PauliArgs {
    StepArgs s;
    PauliRot *rot;
    PauliMeta *meta;
    const uint64_t *prog;
    uint32_t rmax;
    uint32_t do_clean;  // init kernel: run the initial clean (PauliEnv::reset, pauli.rs:576)
    int32_t depth_value;
    // add_perms (pauli.rs:594-599): action un-permutation through the perm chosen by the last observe()
    const int32_t *act_perms;  // [n_perms][num_actions] or null
    const uint32_t *perm_idx;  // [B] current_perm_idx
    uint32_t n_perms;
};

__global__ __launch_bounds__(256) void pauli_step_kernel(PauliArgs pa) {
    const StepArgs &a = pa.s;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env_raw = gid >> PAULI_LOG2L;
    const uint32_t lie = (uint32_t)gid & (PAULI_L - 1);
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const uint32_t base = lane & ~(PAULI_L - 1);
    const bool valid = env_raw < a.B;
    const uint64_t env = valid ? env_raw : a.B - 1;
    const bool leader = valid && lie == 0;
    const bool act64 = a.flags & F_ACT64;
    const uint32_t N = a.N, rmax = pa.rmax;

    PauliLane p;
    p.xr = p.zr = 0;
    p.rx = p.rz = p.rph = p.rpred = 0;
    ulonglong2 *tp = reinterpret_cast<ulonglong2 *>(a.state) + env * N + lie;
    PauliRot *rp = pa.rot + env * rmax + lie;
    if (lie < N) {
        ulonglong2 t = *tp;
        p.xr = t.x;
        p.zr = t.y;
    }
    if (lie < rmax) {
        PauliRot r = *rp;
        p.rx = r.x; p.rz = r.z; p.rph = r.phase; p.rpred = r.pred;
    }
    PauliMeta m = pa.meta[env];
    const PauliMeta m0 = m;
    const uint64_t xr0 = p.xr, zr0 = p.zr;
    const uint32_t rx0 = p.rx, rz0 = p.rz, rph0 = p.rph;
    int32_t depth = a.depth[env];
    int32_t sol_n = (a.flags & F_TRACK) ? a.sol_len[env * 2] : 0;
    const uint64_t idx = lie < N ? (1ull << lie) : 0ull, idz = lie < N ? (1ull << (N + lie)) : 0ull;
    const uint64_t gmask = 0xFFFFFFFFull << base;

    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;

    for (uint32_t t = 0; t < a.T; ++t) {
        int64_t act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
        if (pa.n_perms) {  // actual_action = act_perms[current_perm_idx][action] (pauli.rs:594-599)
            if (act >= 0 && act < (int64_t)a.num_actions) act = pa.act_perms[(uint64_t)pa.perm_idx[env] * a.num_actions + act];
            else fault |= 16u;  // the reference indexes act_perms out of bounds here and panics
        }
        const bool in_range = act >= 0 && act < (int64_t)a.num_actions;  // pauli.rs:601
        uint64_t prog = 0;
        float penalty = 0.0f;
        if (in_range) {
            prog = pa.prog[act];
            penalty = a.gates[act].penalty;
            if ((a.flags & F_LAYERS) && leader) penalty = layers_penalty(a.layers + env * (2 * N + 2), N, a.descs[act], a.w);
        }
        uint32_t n_removed = 0;
        p.rem_seq = ~0u;
        p.rem_code = 0;

#pragma unroll 1
        for (uint32_t k = 0; k < 3; ++k) {  // PauliNetwork::act (pauli_network.rs:225-260)
            const uint32_t mo = (uint32_t)(prog >> (16 * k)) & 0xFFFFu;
            const uint32_t mk = mo & 15u, qa = (mo >> 4) & 63u, qb = (mo >> 10) & 63u;
            if (!__any((int)mk)) continue;
            if (mk == M_H) {  // :189-194
                if (lie == qa) { uint64_t tmp = p.xr; p.xr = p.zr; p.zr = tmp; }
                rot_h(p.rx, p.rz, p.rph, qa);
            } else if (mk == M_S) {  // :209-215
                if (lie == qa) p.zr ^= p.xr;
                rot_s(p.rx, p.rz, p.rph, qa);
            } else if (mk == M_SX) {  // :217-223, Pauli::evolve_sx = h, s, h
                if (lie == qa) p.xr ^= p.zr;
                rot_h(p.rx, p.rz, p.rph, qa);
                rot_s(p.rx, p.rz, p.rph, qa);
                rot_h(p.rx, p.rz, p.rph, qa);
            }
            const bool cn = (mk == M_CNOT);
            if (__any((int)cn)) {  // cnot(i, j) (:196-207): row[i] ^= row[j]; row[N+j] ^= row[N+i]; evolve_cx(j, i); clean
                const uint64_t xj = bperm64(p.xr, base + (qb & 31u));
                const uint64_t zi = bperm64(p.zr, base + (qa & 31u));
                if (cn) {
                    // (i == j zeroes both rows, exactly as xor_rows(i, i) does in the reference)
                    if (lie == qa) p.xr ^= xj;
                    if (lie == qb) p.zr ^= zi;
                    rot_cx(p.rx, p.rz, qb, qa);
                }
                pauli_clean(p, m, n_removed, fault, cn, lie, base, rmax, N);
            }
        }

        if ((a.flags & F_TRACK) && in_range) {  // pauli.rs:612-626 (only for a valid action)
            const int32_t nf = sol_n;
            if ((uint32_t)nf + 1u + n_removed <= a.sol_cap) {
                if (leader) a.sol[env * a.sol_cap + (uint32_t)nf] = sol_word(act);
                if (valid && p.rem_seq != ~0u) {
                    // the phase is read after the whole gate has been applied (pauli.rs:618)
                    const uint32_t ph = rot_phase(p.rx, p.rz, p.rph, N);
                    a.sol[env * a.sol_cap + (uint32_t)nf + 1u + p.rem_seq] = p.rem_code | (ph == 2u ? 0u : 1u);
                }
                sol_n = nf + 1 + (int32_t)n_removed;
            } else {
                fault |= 8u;
            }
        }

        depth = depth > 0 ? depth - 1 : 0;  // pauli.rs:630
        const bool ok = (p.xr == idx) && (p.zr == idz);
        const uint64_t bal = __ballot(ok);
        solved = (m.count == 0) && ((bal & gmask) == gmask);  // PauliNetwork::solved (pauli_network.rs:167-173)
        const float achieved = solved ? 1.0f : 0.0f;
        const float tmp = achieved - penalty;
        const float bonus = a.pauli_layer_reward * (float)n_removed;
        reward = tmp + bonus;  // pauli.rs:634
        if (leader) {
            if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
            if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
        }
    }

    if (valid) {
        if (lie < N && (p.xr != xr0 || p.zr != zr0)) *tp = make_ulonglong2(p.xr, p.zr);
        if (lie < rmax && (p.rx != rx0 || p.rz != rz0 || p.rph != rph0)) {
            PauliRot r;
            r.x = p.rx; r.z = p.rz; r.phase = p.rph; r.pred = p.rpred;
            *rp = r;
        }
        if (fault) atomicOr(&a.error[env], fault);
    }
    if (leader) {
        if (m.alive != m0.alive || m.order != m0.order || m.count != m0.count) pa.meta[env] = m;
        a.depth[env] = depth;
        a.reward[env] = reward;
        a.done[env] = (uint8_t)(depth == 0 || solved);
        a.success[env] = (uint8_t)solved;
        if (a.flags & F_TRACK) a.sol_len[env * 2] = sol_n;
    }
}

// After a host upload of tableau/rotations: optional initial clean, then the scalar resets of
// set_state (pauli.rs:544-551) / reset (pauli.rs:576-585).
__global__ __launch_bounds__(256) void pauli_init_kernel(PauliArgs pa) {
    const StepArgs &a = pa.s;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env_raw = gid >> PAULI_LOG2L;
    const uint32_t lie = (uint32_t)gid & (PAULI_L - 1);
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const uint32_t base = lane & ~(PAULI_L - 1);
    const bool valid = env_raw < a.B;
    const uint64_t env = valid ? env_raw : a.B - 1;
    const uint32_t N = a.N, rmax = pa.rmax;
    PauliLane p;
    p.xr = p.zr = 0;
    p.rx = p.rz = p.rph = p.rpred = 0;
    p.rem_seq = ~0u;
    p.rem_code = 0;
    if (lie < N) {
        ulonglong2 t = reinterpret_cast<const ulonglong2 *>(a.state)[env * N + lie];
        p.xr = t.x;
        p.zr = t.y;
    }
    if (lie < rmax) {
        PauliRot r = pa.rot[env * rmax + lie];
        p.rx = r.x; p.rz = r.z; p.rph = r.phase; p.rpred = r.pred;
    }
    PauliMeta m = pa.meta[env];
    uint32_t n_removed = 0, fault = 0;
    if (pa.do_clean) pauli_clean(p, m, n_removed, fault, true, lie, base, rmax, N);
    const uint64_t idx = lie < N ? (1ull << lie) : 0ull, idz = lie < N ? (1ull << (N + lie)) : 0ull;
    const uint64_t gmask = 0xFFFFFFFFull << base;
    const bool ok = (p.xr == idx) && (p.zr == idz);
    const bool solved = (m.count == 0) && ((__ballot(ok) & gmask) == gmask);
    if (valid && fault) atomicOr(&a.error[env], fault);
    if (valid && lie == 0) {
        pa.meta[env] = m;
        a.depth[env] = pa.depth_value;
        a.success[env] = (uint8_t)solved;
        a.reward[env] = solved ? 1.0f : 0.0f;
        a.done[env] = (uint8_t)(pa.depth_value == 0 || solved);
        a.inverted[env] = 0;
        a.sol_len[env * 2] = 0;
        a.sol_len[env * 2 + 1] = 0;
        if (a.layers) {
            int32_t *lay = a.layers + env * (2 * N + 2);
            for (uint32_t i = 0; i < 2 * N; ++i) lay[i] = -1;
            lay[2 * N] = 0;
            lay[2 * N + 1] = 0;
        }
    }
}

// observe / get_state: one thread per (env, observation row)
struct PauliObsArgs {
    ObsArgs o;
    const PauliRot *rot;
    const PauliMeta *meta;
    uint32_t rmax;
    uint32_t max_rot;
    // observe() with add_perms (pauli.rs:653-665, 445-485): a qubit permutation per env
    const uint8_t *qubit_perms;  // [n_perms][N] or null
    uint32_t *perm_idx;          // [B] current_perm_idx, (re)drawn by this launch when draw != 0
    const int32_t *perm_in;      // [B] explicit draws or null (counter RNG)
    uint32_t n_perms;
    uint32_t draw;
    uint64_t seed;
    uint64_t counter;
    const uint64_t *clock;
};
__global__ __launch_bounds__(256) void pauli_export_kernel(PauliObsArgs pa) {
    const ObsArgs &a = pa.o;
    const uint32_t N = a.N, D = 2 * N;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / D;
    const uint32_t row = (uint32_t)(gid % D);
    if (env >= a.B) return;
    const uint32_t cols = a.obs_cols;
    // qubit permutation of this observation (identity when perms are off or for get_state)
    const uint8_t *perm = nullptr;
    if (pa.n_perms && cols > D && a.format != QG_FMT_PACKED) {
        uint32_t pi;
        if (pa.draw) {  // `rng.gen_range(0..qubit_perms.len())` (pauli.rs:660), made reproducible
            pi = pa.perm_in ? (uint32_t)pa.perm_in[env] % pa.n_perms
                            : (uint32_t)__umul64hi(rng_draw(pa.seed ^ 0x7065726Dull, env, pa.counter + clock_of(pa.clock)), (uint64_t)pa.n_perms);
            if (row == 0) pa.perm_idx[env] = pi;  // current_perm_idx.store (pauli.rs:661)
        } else {
            pi = pa.perm_idx[env];
        }
        perm = pa.qubit_perms + (uint64_t)pi * N;
    }
    const uint32_t q = row < N ? row : row - N;
    const uint32_t sq = perm ? perm[q] : q;  // row i takes data from row perm[i] (pauli.rs:455-464)
    const ulonglong2 t = reinterpret_cast<const ulonglong2 *>(a.state)[env * N + sq];
    uint64_t w = row < N ? t.x : t.y;
    if (perm) {  // column i takes data from column perm[i], X and Z halves alike (pauli.rs:469-477)
        uint64_t pw = 0;
        for (uint32_t i = 0; i < N; ++i) {
            pw |= ((w >> perm[i]) & 1ull) << i;
            pw |= ((w >> (N + perm[i])) & 1ull) << (N + i);
        }
        w = pw;
    }
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<uint64_t *>(a.out)[env * a.out_stride + row] = w;
        return;
    }
    // pad_and_collect (pauli.rs:411-437): tableau, then the active rotations in DAG node order;
    // rotation columns are permuted along rows only (pauli.rs:478-481)
    uint32_t extra = 0;
    if (cols > D) {
        const PauliMeta m = pa.meta[env];
        const uint32_t shown = m.count < pa.max_rot ? m.count : pa.max_rot;
        for (uint32_t i = 0; i < shown; ++i) {
            const PauliRot r = pa.rot[env * pa.rmax + nib(m.order, i)];
            const uint32_t bit = row < N ? (r.x >> sq) & 1u : (r.z >> sq) & 1u;
            extra |= bit << i;
        }
    }
    if (a.format == QG_FMT_I64) {
        int64_t *o = reinterpret_cast<int64_t *>(a.out) + env * a.out_stride + (uint64_t)row * cols;
        for (uint32_t c = 0; c < D; ++c) o[c] = (int64_t)((w >> c) & 1ull);
        for (uint32_t c = D; c < cols; ++c) o[c] = (int64_t)((extra >> (c - D)) & 1u);
    } else {
        int8_t *o = reinterpret_cast<int8_t *>(a.out) + env * a.out_stride + (uint64_t)row * cols;
        for (uint32_t c = 0; c < D; ++c) o[c] = (int8_t)((w >> c) & 1ull);
        for (uint32_t c = D; c < cols; ++c) o[c] = (int8_t)((extra >> (c - D)) & 1u);
    }
}

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

// ------------------------------------------------------------------------------------------------
// host hooks
// ------------------------------------------------------------------------------------------------

int pauli_plan(qg_vec *v) {
    if (v->N > 32) return set_error(QG_ERR_UNSUPPORTED, "PauliEnv: N <= 32 supported, got %u", v->N);
    const int max_rot = std::max(v->cfg.max_rotations, 1);  // pauli.rs:387
    const int final_layers = v->cfg.final_pauli_layers >= 0 ? v->cfg.final_pauli_layers : v->cfg.max_rotations + 2;  // :760
    const int rmax = std::max(max_rot, final_layers);
    if (rmax > (int)PAULI_RMAX)
        return set_error(QG_ERR_UNSUPPORTED, "PauliEnv: at most %u rotations per env supported (max_rotations=%d, final_pauli_layers=%d)",
                         PAULI_RMAX, max_rot, final_layers);
    v->rmax = (uint32_t)rmax;
    v->rmax_generate = (uint32_t)final_layers;  // reset() generates at most final_pauli_layers rotations (pauli.rs:563)
    v->cfg.max_rotations = max_rot;
    v->stride_bytes = (size_t)16 * v->N;
    v->log2L = PAULI_LOG2L;
    // thread-per-env PTILE family by default; QGYM_PAULI_LANES=1 keeps the lane-group kernels
    v->pauli_tile = getenv("QGYM_PAULI_LANES") == nullptr;
    if (v->pauli_tile) return ptile_plan(v);
    return QG_OK;
}

int pauli_alloc(qg_vec *v) {
    if (v->pauli_tile) {
        int rc = ptile_alloc(v);
        if (rc) return rc;
    } else {
        HIP_TRY(hipMalloc(&v->rot, sizeof(PauliRot) * v->rmax * v->B));
        HIP_TRY(hipMalloc(&v->pmeta, sizeof(PauliMeta) * v->B));
        std::vector<uint64_t> prog(std::max<size_t>(v->gates.size(), 1));
        for (size_t i = 0; i < v->gates.size(); ++i) prog[i] = gate_program(v->gates[i]);
        HIP_TRY(hipMalloc(&v->d_prog, sizeof(uint64_t) * prog.size()));
        HIP_TRY(hipMemcpy(v->d_prog, prog.data(), sizeof(uint64_t) * prog.size(), hipMemcpyHostToDevice));
    }
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

static void fill_pauli_args(const qg_vec *v, const StepArgs &a, PauliArgs &pa) {
    pa.s = a;
    pa.rot = reinterpret_cast<PauliRot *>(v->rot);
    pa.meta = reinterpret_cast<PauliMeta *>(v->pmeta);
    pa.prog = reinterpret_cast<const uint64_t *>(v->d_prog);
    pa.rmax = v->rmax;
    pa.do_clean = 0;
    pa.depth_value = 0;
    pa.act_perms = v->d_act_perms;
    pa.perm_idx = v->perm_idx;
    pa.n_perms = v->n_perms;
}

hipError_t pauli_step(const qg_vec *v, const StepArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (v->pauli_tile) return ptile_step(v, a, s);
    PauliArgs pa;
    fill_pauli_args(v, a, pa);
    hipLaunchKernelGGL(pauli_step_kernel, dim3(grid_for(a.B * PAULI_L, 256)), dim3(256), 0, s, pa);
    return hipGetLastError();
}

hipError_t pauli_export(const qg_vec *v, const ObsArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (v->pauli_tile) return ptile_export(v, a, s);
    PauliObsArgs pa;
    pa.o = a;
    pa.rot = reinterpret_cast<const PauliRot *>(v->rot);
    pa.meta = reinterpret_cast<const PauliMeta *>(v->pmeta);
    pa.rmax = v->rmax;
    pa.max_rot = (uint32_t)v->cfg.max_rotations;
    pa.qubit_perms = v->d_qubit_perms;
    pa.perm_idx = v->perm_idx;
    pa.perm_in = v->perm_in;
    pa.n_perms = v->n_perms;
    pa.draw = v->perm_draw ? 1u : 0u;
    pa.seed = v->coin_seed;
    pa.counter = v->observe_counter;
    pa.clock = v->clock_dev;
    hipLaunchKernelGGL(pauli_export_kernel, dim3(grid_for(a.B * 2ull * a.N, 256)), dim3(256), 0, s, pa);
    return hipGetLastError();
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
    h.meta.assign(v->B, PauliMeta{0, 0, 0});
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
    m.order = 0;
    for (size_t k = 0; k < R; ++k) m.order |= (uint64_t)k << (4 * k);
    return QG_OK;
}

static int host_net_upload(qg_vec *v, const HostNet &h, bool do_clean, int32_t depth_value, hipStream_t s) {
    if (v->pauli_tile) return ptile_upload(v, h, do_clean, depth_value, s);
    HIP_TRY(hipMemcpyAsync(v->state, h.tab.data(), sizeof(uint64_t) * h.tab.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(v->rot, h.rot.data(), sizeof(PauliRot) * h.rot.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(v->pmeta, h.meta.data(), sizeof(PauliMeta) * h.meta.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(v->error, 0, sizeof(uint32_t) * v->B, s));
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.state = v->state;
    a.depth = v->depth;
    a.reward = v->reward;
    a.done = v->done;
    a.success = v->success;
    a.inverted = v->inverted;
    a.error = v->error;
    a.sol_len = v->sol_len;
    a.layers = v->layers;
    a.B = v->B;
    a.N = v->N;
    PauliArgs pa;
    fill_pauli_args(v, a, pa);
    pa.do_clean = do_clean ? 1u : 0u;
    pa.depth_value = depth_value;
    hipLaunchKernelGGL(pauli_init_kernel, dim3(grid_for(v->B * PAULI_L, 256)), dim3(256), 0, s, pa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));  // host staging vectors die with the caller's frame
    return QG_OK;
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


// ------------------------------------------------------------------------------------------------
// PauliEnv::reset (pauli.rs:554-586) with its random target generator (pauli.rs:54-271).
// Targets are generated on the host (reset is not on the step path) from the handle's counter RNG:
// draw k of env e is rng_draw(seed ^ 0x7061756C, e, k); gen_range(0..n) = mulhi64(u, n),
// gen::<f32>() = (u >> 40) * 2^-24.  The reference's RNG cannot be seeded, so only the
// distribution is fixed by it; the stream definition is what tests replay.
// ------------------------------------------------------------------------------------------------
namespace {
struct Stream {
    uint64_t seed, env, k = 0;
    uint64_t next() { return rng_draw(seed, env, k++); }
    size_t range(size_t n) { return (size_t)(((unsigned __int128)next() * (unsigned __int128)n) >> 64); }
    float f32() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

struct CouplingInfo {  // constructor-time data of PauliEnv::new (pauli.rs:360-370)
    std::vector<std::pair<uint32_t, uint32_t>> valid_pairs;         // CX gates, gateset order
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> by_dist; // by_dist[d]: pairs q1<q2 at graph distance d
    std::vector<uint32_t> all_dists;                                  // ascending distances that occur
};

CouplingInfo coupling_info(const qg_vec *v) {
    CouplingInfo ci;
    const uint32_t n = v->N;
    for (const qg_gate &g : v->gates)
        if (g.kind == QG_CX) ci.valid_pairs.emplace_back((uint32_t)g.q0, (uint32_t)g.q1);
    std::vector<std::vector<uint32_t>> adj(n);
    for (auto &pr : ci.valid_pairs) {  // compute_graph_distances (pauli.rs:56-91)
        auto add = [&](uint32_t a, uint32_t b) { if (std::find(adj[a].begin(), adj[a].end(), b) == adj[a].end()) adj[a].push_back(b); };
        add(pr.first, pr.second);
        add(pr.second, pr.first);
    }
    std::vector<std::vector<int>> dist(n, std::vector<int>(n, -1));
    for (uint32_t st = 0; st < n; ++st) {
        std::vector<uint32_t> q{st};
        dist[st][st] = 0;
        for (size_t h = 0; h < q.size(); ++h)
            for (uint32_t w : adj[q[h]])
                if (dist[st][w] < 0) { dist[st][w] = dist[st][q[h]] + 1; q.push_back(w); }
    }
    for (uint32_t a = 0; a < n; ++a)  // build_dist_pairs (pauli.rs:95-111)
        for (uint32_t b = a + 1; b < n; ++b)
            if (dist[a][b] >= 0) {
                if (ci.by_dist.size() <= (size_t)dist[a][b]) ci.by_dist.resize(dist[a][b] + 1);
                ci.by_dist[dist[a][b]].emplace_back(a, b);
            }
    for (uint32_t d = 0; d < ci.by_dist.size(); ++d)
        if (!ci.by_dist[d].empty()) ci.all_dists.push_back(d);
    return ci;
}

// get_pauli_under_diff (pauli.rs:115-188); false = None
bool pauli_under_diff(const CouplingInfo &ci, uint32_t n, size_t difficulty, float decay, Stream &rng, std::string &label, size_t &cost) {
    std::vector<uint32_t> valid;
    for (uint32_t d : ci.all_dists) if (d <= difficulty) valid.push_back(d);
    if (valid.empty()) return false;
    std::vector<uint8_t> in(n, 0);
    size_t budget = difficulty;
    uint32_t d = valid[rng.range(valid.size())];
    auto pr = ci.by_dist[d][rng.range(ci.by_dist[d].size())];
    in[pr.first] = in[pr.second] = 1;
    budget = budget > d ? budget - d : 0;
    for (;;) {
        size_t n_ok = 0, free_q = 0;
        for (uint32_t x : valid) n_ok += x <= budget;  // `valid` is ascending: the usable ones are a prefix
        for (uint32_t q = 0; q < n; ++q) free_q += !in[q];
        if (budget == 0 || n_ok == 0 || free_q == 0) break;
        if (rng.f32() <= decay) break;
        d = valid[rng.range(n_ok)];
        std::vector<std::pair<uint32_t, uint32_t>> touching;
        for (auto &p2 : ci.by_dist[d]) if (in[p2.first] || in[p2.second]) touching.push_back(p2);
        if (touching.empty()) continue;
        pr = touching[rng.range(touching.size())];
        in[pr.first] = in[pr.second] = 1;
        budget = budget > d ? budget - d : 0;
    }
    label.assign(n, 'I');
    for (uint32_t q = 0; q < n; ++q) if (in[q]) label[q] = "XYZ"[rng.range(3)];
    cost = difficulty - budget;
    return true;
}
}  // namespace

int pauli_reset_seeded(qg_vec *v, uint64_t seed, hipStream_t s) {
    if (v->pauli_tile) return ptile_reset_seeded(v, seed, false, s);  // generated on the device
    const uint32_t n = v->N, D = 2 * n;
    const CouplingInfo ci = coupling_info(v);
    const size_t scale = (size_t)std::max(v->cfg.pauli_diff_scale, 1);  // pauli.rs:392
    const size_t pauli_difficulty = (size_t)v->difficulty / scale;      // pauli.rs:557
    HostNet h;
    host_net_init(v, h);
    std::vector<uint8_t> tab((size_t)D * D);
    for (uint64_t e = 0; e < v->B; ++e) {
        Stream rng{seed ^ 0x7061756Cull, e};
        std::vector<std::string> labels;  // generate_paulis_with_difficulty (pauli.rs:191-213)
        size_t remaining = pauli_difficulty;
        while (remaining > 0 && labels.size() < (size_t)v->rmax_generate) {
            std::string lab;
            size_t cost = 0;
            if (!pauli_under_diff(ci, n, remaining, v->cfg.num_qubits_decay, rng, lab, cost)) break;
            labels.push_back(lab);
            const size_t dec = std::max<size_t>(cost, 1);
            remaining = remaining > dec ? remaining - dec : 0;
        }
        std::fill(tab.begin(), tab.end(), 0);  // random_clifford_tableau (pauli.rs:220-271)
        for (uint32_t i = 0; i < D; ++i) tab[(size_t)i * D + i] = 1;
        if (v->difficulty != 0 && !ci.valid_pairs.empty()) {
            auto xor_rows = [&](uint32_t a, uint32_t b) { for (uint32_t c = 0; c < D; ++c) tab[(size_t)a * D + c] ^= tab[(size_t)b * D + c]; };
            for (int64_t it = 0; it < v->difficulty; ++it) {
                const float r = rng.f32();
                if (r > 0.3f) {
                    auto pr = ci.valid_pairs[rng.range(ci.valid_pairs.size())];
                    xor_rows(pr.second, pr.first);
                    xor_rows(n + pr.first, n + pr.second);
                } else if (r > 0.15f) {
                    const uint32_t q = (uint32_t)rng.range(n);
                    for (uint32_t c = 0; c < D; ++c) std::swap(tab[(size_t)q * D + c], tab[(size_t)(n + q) * D + c]);
                } else {
                    const uint32_t q = (uint32_t)rng.range(n);
                    xor_rows(n + q, q);
                }
            }
        }
        int rc = host_net_build(v, h, e, tab.data(), labels);
        if (rc) return rc;
    }
    const int64_t d = (int64_t)v->cfg.depth_slope * v->difficulty;  // pauli.rs:578
    return host_net_upload(v, h, true, (int32_t)std::min<int64_t>(d, v->cfg.max_depth), s);
}

}  // namespace qg
