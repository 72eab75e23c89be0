// kernels_pauli_tile.hip -- PauliEnv, thread-per-env ("PTILE" layout).  The default PauliEnv path.
//
// Reference semantics:
//   PauliEnv::step / observe / reset tail     rust/src/envs/pauli.rs:588-635, 411-485, 573-585
//   PauliNetwork::{act,cnot,h,s,sx,clean_and_return_with_phases,solved}   rust/src/pauli/pauli_network.rs:139-260
//   Pauli::{evolve_h,evolve_s,evolve_cx,evolve_sx,phase}                  rust/src/pauli/pauli.rs:83-133
//   PauliDag::get_front_layer                 rust/src/pauli/pauli_dag.rs:47-57
//   petgraph 0.6.5 Graph::retain_nodes / remove_node (reverse visit + Vec::swap_remove): third
//   party, restated from its published source (fixes the observation's rotation-column order).
//
// Why thread-per-env: a lane-group form (32 lanes per env) spends ~200 wave-instructions per env-step.  Here a
// lane owns a whole env -- 2N tableau rows as uint64 plus <= RM rotation records in VGPRs -- so a
// wave instruction advances 64 envs and nothing crosses lanes.
//
//  * Tableau: the micro-ops of one gate (pauli_network.rs:225-260) act on the rows of at most two
//    qubits and are linear, and rotation removal never touches the tableau, so the whole gate is
//    ONE 4x4 GF(2) map on {X[a], Z[a], X[b], Z[b]} compiled on the host (CX = the reference's
//    reversed `cnot`, CZ = H.cnot.H, SWAP = three cnots, Sdg/SXdg = S/SX on bits).
//  * Rotations: micro-op by micro-op, because `clean` runs after every cnot and phases are not
//    linear.  A rotation is (x mask, z mask, 2-bit phase); phases of all rotations live in two
//    bit-planes so one micro-op updates them with three logic ops.  Weight = popc(x|z), front layer
//    = pred & alive == 0, node order = nibble-packed word (swap_remove order, high to low).
//
// Memory (PTILE layout): tiles of 64 envs (one wavefront), every wave access a contiguous block.
//   wide    (N > 24 or more than 8 rotations): NQ + RM + 1 groups of 1 KiB, one uint4 per lane:
//           groups 0..NQ-1 = {X row q, Z row q} (two uint64), groups NQ.. = rotation k
//           {x, z, phase, pred}, last group = {alive | count << 16, bad, order} where bit q of `bad`
//           says that qubit q's rows differ from the identity's (kept incrementally: a gate changes
//           the rows of <= 2 qubits, so `solved` never needs the whole tableau);
//   compact (N <= 24 and <= 8 rotations, the common case): rows are <= 48 bits, so a qubit's two
//           rows are 12 bytes {X[0:32), X[32:48) | Z[0:16) << 16, Z[16:48)} (768 B per group,
//           dwordx3 accesses).  The rotations are stored TRANSPOSED in the 8 x 8-byte rotation groups
//           (64 bytes per env): bytes [0, 48) = {xs[q], zs[q]} for q < 24, where bit k of xs[q] is bit q
//           of rotation k's x mask -- a gate only reads and writes the bytes of its two qubits, which are
//           exactly the bit-slices its micro-ops work on; bytes [48, 56) = pred[k]; byte 56 / 57 = the two
//           phase bit-planes; bytes 58..62 = the rotations' weights popc(x | z) as five bit-planes (bit k
//           of plane j = bit j of rotation k's weight), so `clean` never needs the masks.  N = 20: 320 B
//           per env instead of 464.
//
// Two step kernels: ptile_step1_kernel (one step per launch, the env.step() path) keeps only the
// rotations in registers and gathers / scatters the rows of the gate's <= 2 qubits at per-lane
// addresses -- ~2x fewer instructions than holding the tableau, and the step is issue-bound at one
// wave per SIMD; ptile_step_kernel (fused rollouts, T steps per launch) holds everything in VGPRs.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "device_common.hpp"
#include "pauli_common.hpp"

namespace qg {

// ---- host: per-action program word --------------------------------------------------------------
// [0:5) qa  [5:10) qb  [10:26) tableau map M (bit 4k+i: output k takes input i, order Xa,Za,Xb,Zb)
// [26:38) three micro-ops, 4 bits each: kind | (operands swapped ? 8 : 0)
static uint64_t ptile_program(const qg_gate &g) {
    const uint32_t a = (uint32_t)g.q0, b = g.kind >= QG_CX ? (uint32_t)g.q1 : (uint32_t)g.q0;
    struct Mop { uint32_t kind; bool swapped; };
    Mop mops[3] = {{M_NOP, false}, {M_NOP, false}, {M_NOP, false}};
    switch (g.kind) {  // PauliNetwork::act (pauli_network.rs:225-260)
    case QG_H: mops[0] = {M_H, false}; break;
    case QG_S: mops[0] = {M_S, false}; break;
    case QG_SDG: mops[0] = mops[1] = mops[2] = {M_S, false}; break;
    case QG_SX: mops[0] = {M_SX, false}; break;
    case QG_SXDG: mops[0] = mops[1] = mops[2] = {M_SX, false}; break;
    case QG_CX: mops[0] = {M_CNOT, false}; break;
    case QG_CZ: mops[0] = {M_H, true}; mops[1] = {M_CNOT, false}; mops[2] = {M_H, true}; break;
    case QG_SWAP: mops[0] = {M_CNOT, false}; mops[1] = {M_CNOT, true}; mops[2] = {M_CNOT, false}; break;
    }
    // symbolic execution of the micro-ops on {Xa, Za, Xb, Zb}; with a == b the b-slots alias the a-slots
    uint32_t row[4] = {1, 2, 4, 8};
    auto X = [&](bool second) -> uint32_t & { return row[(second && a != b) ? 2 : 0]; };
    auto Z = [&](bool second) -> uint32_t & { return row[(second && a != b) ? 3 : 1]; };
    uint64_t mbits = 0;
    for (int k = 0; k < 3; ++k) {
        const bool p = mops[k].swapped, q = !mops[k].swapped;  // operand -> "is it qubit b"
        switch (mops[k].kind) {
        case M_H: std::swap(X(p), Z(p)); break;                         // :189-194
        case M_S: Z(p) ^= X(p); break;                                   // :209-215
        case M_SX: X(p) ^= Z(p); break;                                  // :217-223
        case M_CNOT: X(p) ^= X(q); Z(q) ^= Z(p); break;                  // cnot(i=p, j=q) :196-201
        default: break;
        }
        mbits |= (uint64_t)(mops[k].kind | (mops[k].swapped ? 8u : 0u)) << (26 + 4 * k);
    }
    const uint64_t M = row[0] | (row[1] << 4) | (row[2] << 8) | (row[3] << 12);
    return (uint64_t)(a & 31u) | ((uint64_t)(b & 31u) << 5) | (M << 10) | mbits;
}

// ---- device helpers ------------------------------------------------------------------------------
template <int n>
__device__ inline uint64_t tree_select64(const uint64_t (&t)[n], uint32_t q) {
    if constexpr (n == 1) {
        return t[0];
    } else {
        constexpr int m = (n + 1) / 2;
        uint64_t u[m];
        const uint64_t mb = 0ull - (uint64_t)(q & 1u);  // arithmetic blend, see kernels_qm.hip tree_select
#pragma unroll
        for (int k = 0; k < m; ++k) u[k] = (2 * k + 1 < n) ? ((t[2 * k + 1] & mb) | (t[2 * k] & ~mb)) : t[2 * k];
        return tree_select64<m>(u, q >> 1);
    }
}
__device__ inline uint32_t pnib(uint64_t order, uint32_t i) { return (uint32_t)(order >> (4 * i)) & 0xFu; }

// DAG node order (petgraph NodeIndex -> rotation index): up to 16 rotations as nibbles of one uint64 (the memory format of the one- and
// two-group layouts below), up to 32 as bytes of four
template <int RM>
struct PTOrder {
    uint64_t w;
    __device__ inline uint32_t get(uint32_t i) const { return (uint32_t)(w >> (4 * i)) & 0xFu; }
    __device__ inline void set(uint32_t i, uint32_t v) { w = (w & ~(0xFull << (4 * i))) | ((uint64_t)v << (4 * i)); }
    __device__ inline void clear() { w = 0; }
    __device__ inline bool operator!=(const PTOrder &o) const { return w != o.w; }
};
template <>
struct PTOrder<32> {
    uint64_t w[4];
    __device__ inline uint32_t get(uint32_t i) const {  // i may be a per-lane value: word select, then shift
        const uint32_t k = i >> 3;
        const uint64_t x = k == 0 ? w[0] : k == 1 ? w[1] : k == 2 ? w[2] : w[3];
        return (uint32_t)(x >> (8 * (i & 7u))) & 0xFFu;
    }
    __device__ inline void set(uint32_t i, uint32_t v) {
        const uint32_t k = i >> 3, sh = 8 * (i & 7u);
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) w[j] = k == j ? (w[j] & ~(0xFFull << sh)) | ((uint64_t)v << sh) : w[j];
    }
    __device__ inline void clear() { w[0] = w[1] = w[2] = w[3] = 0; }
    __device__ inline bool operator!=(const PTOrder &o) const { return ((w[0] ^ o.w[0]) | (w[1] ^ o.w[1]) | (w[2] ^ o.w[2]) | (w[3] ^ o.w[3])) != 0; }
};

template <int NQ, int RM>
struct PTState {
    uint64_t X[NQ], Z[NQ];
    uint32_t rx[RM], rz[RM], rpred[RM];
    uint32_t plo, phi;  // phase bit-planes: phase of rotation k = ((phi >> k) & 1) * 2 + ((plo >> k) & 1)
    uint32_t alive, count;
    uint32_t bad;  // bit q: rows X[q] / Z[q] differ from the identity tableau's
    PTOrder<RM> order;
};

template <int NQ, int RM>
struct PTLayout {
    static constexpr bool COMPACT = NQ <= 24 && RM == 8;
    static constexpr uint32_t QB = COMPACT ? 768u : 1024u;   // bytes of one qubit group
    static constexpr uint32_t RB = COMPACT ? 512u : 1024u;   // bytes of one rotation group
    static constexpr uint32_t META_GROUPS = RM > 16 ? 3u : 1u;  // > 16 rotations: {alive, count, bad}, then 32 order bytes in two groups
    static constexpr uint32_t TILE_BYTES = NQ * QB + RM * RB + META_GROUPS * 1024u;
    static __device__ inline char *tile(void *state, uint64_t env) { return reinterpret_cast<char *>(state) + (env >> 6) * (uint64_t)TILE_BYTES; }

    static __device__ inline void load_qubit(const char *t, uint32_t lane, uint32_t q, uint64_t &X, uint64_t &Z) {
        if constexpr (COMPACT) {
            const uint32_t *p = reinterpret_cast<const uint32_t *>(t + q * QB + lane * 12u);
            const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
            X = (uint64_t)d0 | ((uint64_t)(d1 & 0xFFFFu) << 32);
            Z = (uint64_t)(d1 >> 16) | ((uint64_t)d2 << 16);
        } else {
            const uint4 v = *reinterpret_cast<const uint4 *>(t + q * QB + lane * 16u);
            X = (uint64_t)v.x | ((uint64_t)v.y << 32);
            Z = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
    }
    static __device__ inline void store_qubit(char *t, uint32_t lane, uint32_t q, uint64_t X, uint64_t Z) {
        if constexpr (COMPACT) {
            uint32_t *p = reinterpret_cast<uint32_t *>(t + q * QB + lane * 12u);
            p[0] = (uint32_t)X;
            p[1] = ((uint32_t)(X >> 32) & 0xFFFFu) | ((uint32_t)Z << 16);
            p[2] = (uint32_t)(Z >> 16);
        } else {
            *reinterpret_cast<uint4 *>(t + q * QB + lane * 16u) = make_uint4((uint32_t)X, (uint32_t)(X >> 32), (uint32_t)Z, (uint32_t)(Z >> 32));
        }
    }
    static __device__ inline void load_rot(const char *t, uint32_t lane, int k, uint32_t &x, uint32_t &z, uint32_t &ph, uint32_t &pred) {
        if constexpr (COMPACT) {
            const uint2 v = *reinterpret_cast<const uint2 *>(t + NQ * QB + k * RB + lane * 8u);
            x = v.x & 0xFFFFFFu; pred = v.x >> 24; z = v.y & 0xFFFFFFu; ph = (v.y >> 24) & 3u;
        } else {
            const uint4 v = *reinterpret_cast<const uint4 *>(t + NQ * QB + k * RB + lane * 16u);
            x = v.x; z = v.y; ph = v.z & 3u; pred = v.w;
        }
    }
    static __device__ inline void store_rot(char *t, uint32_t lane, int k, uint32_t x, uint32_t z, uint32_t ph, uint32_t pred) {
        if constexpr (COMPACT) *reinterpret_cast<uint2 *>(t + NQ * QB + k * RB + lane * 8u) = make_uint2(x | (pred << 24), z | (ph << 24));
        else *reinterpret_cast<uint4 *>(t + NQ * QB + k * RB + lane * 16u) = make_uint4(x, z, ph, pred);
    }
    static __device__ inline uint4 *meta(char *t, uint32_t lane, uint32_t g = 0) {
        return reinterpret_cast<uint4 *>(t + NQ * QB + RM * RB + g * 1024u + lane * 16u);
    }
    // compact layout: the transposed rotation region (see the file header)
    static __device__ inline uint16_t *xz(char *t, uint32_t lane, uint32_t q) {
        return reinterpret_cast<uint16_t *>(t + NQ * QB + (q >> 2) * RB + lane * 8u + (q & 3u) * 2u);
    }
    static __device__ inline uint2 *rotgroup(char *t, uint32_t lane, uint32_t g) { return reinterpret_cast<uint2 *>(t + NQ * QB + g * RB + lane * 8u); }
};

// DAG bookkeeping: {alive | count << 16, bad, order as 16 nibbles} in one group, or (more than 16 rotations) {alive, count, bad, -}
// followed by the 32 order bytes in two groups
template <int NQ, int RM>
__device__ inline void pt_load_meta(char *t, uint32_t lane, PTState<NQ, RM> &s) {
    using L = PTLayout<NQ, RM>;
    const uint4 m = *L::meta(t, lane);
    if constexpr (RM > 16) {
        const uint4 o0 = *L::meta(t, lane, 1), o1 = *L::meta(t, lane, 2);
        s.alive = m.x;
        s.count = m.y;
        s.bad = m.z;
        s.order.w[0] = (uint64_t)o0.x | ((uint64_t)o0.y << 32);
        s.order.w[1] = (uint64_t)o0.z | ((uint64_t)o0.w << 32);
        s.order.w[2] = (uint64_t)o1.x | ((uint64_t)o1.y << 32);
        s.order.w[3] = (uint64_t)o1.z | ((uint64_t)o1.w << 32);
    } else {
        s.alive = m.x & 0xFFFFu;
        s.count = m.x >> 16;
        s.bad = m.y;
        s.order.w = (uint64_t)m.z | ((uint64_t)m.w << 32);
    }
}

// rotations and bookkeeping (dense kernels; the compact layout's transposed rotation bytes are turned back
// into one (x, z) mask pair per rotation here)
template <int NQ, int RM>
__device__ inline void pt_load_rotations(const char *tile, uint32_t lane, PTState<NQ, RM> &s) {
    using L = PTLayout<NQ, RM>;
    char *t = const_cast<char *>(tile);
    s.plo = s.phi = 0;
    if constexpr (L::COMPACT) {
        uint32_t w[12];
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const uint2 v = (2 * g < (NQ + 1) / 2) ? *L::rotgroup(t, lane, g) : make_uint2(0u, 0u);
            w[2 * g] = v.x;
            w[2 * g + 1] = v.y;
        }
#pragma unroll
        for (int k = 0; k < RM; ++k) s.rx[k] = s.rz[k] = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const uint32_t v = w[q >> 1] >> (16 * (q & 1));
#pragma unroll
            for (int k = 0; k < RM; ++k) {
                s.rx[k] |= ((v >> k) & 1u) << q;
                s.rz[k] |= ((v >> (8 + k)) & 1u) << q;
            }
        }
        const uint2 pv = *L::rotgroup(t, lane, 6), pw = *L::rotgroup(t, lane, 7);
#pragma unroll
        for (int k = 0; k < RM; ++k) s.rpred[k] = ((k < 4 ? pv.x : pv.y) >> (8 * (k & 3))) & 0xFFu;
        s.plo = pw.x & 0xFFu;
        s.phi = (pw.x >> 8) & 0xFFu;
    } else {
#pragma unroll
        for (int k = 0; k < RM; ++k) {
            uint32_t ph;
            L::load_rot(tile, lane, k, s.rx[k], s.rz[k], ph, s.rpred[k]);
            s.plo |= (ph & 1u) << k;
            s.phi |= ((ph >> 1) & 1u) << k;
        }
    }
    pt_load_meta<NQ, RM>(t, lane, s);
}
template <int NQ, int RM>
__device__ inline void pt_load(const char *tile, uint32_t lane, PTState<NQ, RM> &s) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) PTLayout<NQ, RM>::load_qubit(tile, lane, (uint32_t)q, s.X[q], s.Z[q]);
    pt_load_rotations<NQ, RM>(tile, lane, s);
}
// compact layout: the {xs, zs} bytes of qubit q from the per-rotation masks
template <int NQ, int RM>
__device__ inline void pt_store_xz(char *tile, uint32_t lane, const PTState<NQ, RM> &s, uint32_t q) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < RM; ++k) v |= (((s.rx[k] >> q) & 1u) << k) | (((s.rz[k] >> q) & 1u) << (8 + k));
    *PTLayout<NQ, RM>::xz(tile, lane, q) = (uint16_t)v;
}
// compact layout: pred bytes, phase planes and weight planes
template <int NQ, int RM>
__device__ inline void pt_store_rotmeta(char *tile, uint32_t lane, const PTState<NQ, RM> &s, bool with_pred) {
    using L = PTLayout<NQ, RM>;
    uint32_t wp[5] = {0, 0, 0, 0, 0}, p0 = 0, p1 = 0;
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        const uint32_t wk = (uint32_t)__popc(s.rx[k] | s.rz[k]);
#pragma unroll
        for (int j = 0; j < 5; ++j) wp[j] |= ((wk >> j) & 1u) << k;
        if (k < 4) p0 |= (s.rpred[k] & 0xFFu) << (8 * k);
        else p1 |= (s.rpred[k] & 0xFFu) << (8 * (k - 4));
    }
    if (with_pred) *L::rotgroup(tile, lane, 6) = make_uint2(p0, p1);
    *L::rotgroup(tile, lane, 7) = make_uint2((s.plo & 0xFFu) | ((s.phi & 0xFFu) << 8) | (wp[0] << 16) | (wp[1] << 24), wp[2] | (wp[3] << 8) | (wp[4] << 16));
}
// write back the rotations a dense kernel changed: `touched` = rotations whose record changed, `qubits` = the
// qubits gates were applied to (the only bit positions a micro-op can change)
template <int NQ, int RM>
__device__ inline void pt_store_rotations(char *tile, uint32_t lane, const PTState<NQ, RM> &s, uint32_t touched, uint32_t qubits, bool everything) {
    using L = PTLayout<NQ, RM>;
    if constexpr (L::COMPACT) {
        if (!touched && !everything) return;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (everything || ((qubits >> q) & 1u)) pt_store_xz<NQ, RM>(tile, lane, s, (uint32_t)q);
        pt_store_rotmeta<NQ, RM>(tile, lane, s, everything);
    } else {
#pragma unroll
        for (int k = 0; k < RM; ++k)
            if (everything || ((touched >> k) & 1u))
                L::store_rot(tile, lane, k, s.rx[k], s.rz[k], ((s.plo >> k) & 1u) | (((s.phi >> k) & 1u) << 1), s.rpred[k]);
    }
}
template <int NQ, int RM>
__device__ inline void pt_store_meta(char *tile, uint32_t lane, const PTState<NQ, RM> &s) {
    using L = PTLayout<NQ, RM>;
    if constexpr (RM > 16) {
        *L::meta(tile, lane) = make_uint4(s.alive, s.count, s.bad, 0u);
        *L::meta(tile, lane, 1) = make_uint4((uint32_t)s.order.w[0], (uint32_t)(s.order.w[0] >> 32), (uint32_t)s.order.w[1], (uint32_t)(s.order.w[1] >> 32));
        *L::meta(tile, lane, 2) = make_uint4((uint32_t)s.order.w[2], (uint32_t)(s.order.w[2] >> 32), (uint32_t)s.order.w[3], (uint32_t)(s.order.w[3] >> 32));
    } else {
        *L::meta(tile, lane) = make_uint4(s.alive | (s.count << 16), s.bad, (uint32_t)s.order.w, (uint32_t)(s.order.w >> 32));
    }
}

// the gate's composite tableau map on {X[qa], Z[qa], X[qb], Z[qb]}: out[k] = xor of the inputs bit 4k+i of m selects
__device__ inline void pt_mix(uint32_t m, uint64_t xa, uint64_t za, uint64_t xb, uint64_t zb, uint64_t (&out)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = m >> (4 * k);
        out[k] = ((0ull - (uint64_t)(b & 1u)) & xa) ^ ((0ull - (uint64_t)((b >> 1) & 1u)) & za) ^
                 ((0ull - (uint64_t)((b >> 2) & 1u)) & xb) ^ ((0ull - (uint64_t)((b >> 3) & 1u)) & zb);
    }
}
// `bad` after qubits qa / qb received the rows n[0..3] (qa's rows win when qa == qb)
__device__ inline uint32_t pt_bad_update(uint32_t bad, uint32_t N, uint32_t qa, uint32_t qb, const uint64_t (&n)[4]) {
    const uint32_t bb = (uint32_t)(n[2] != (1ull << qb) || n[3] != ((1ull << N) << qb));
    const uint32_t ba = (uint32_t)(n[0] != (1ull << qa) || n[1] != ((1ull << N) << qa));
    bad = (bad & ~(1u << qb)) | (bb << qb);
    return (bad & ~(1u << qa)) | (ba << qa);
}
template <int NQ, int RM>
__device__ inline void pt_apply_tableau(PTState<NQ, RM> &s, uint32_t N, uint32_t qa, uint32_t qb, uint32_t m) {
    const uint64_t xa = tree_select64<NQ>(s.X, qa), za = tree_select64<NQ>(s.Z, qa);
    const uint64_t xb = tree_select64<NQ>(s.X, qb), zb = tree_select64<NQ>(s.Z, qb);
    uint64_t n[4];
    pt_mix(m, xa, za, xb, zb, n);
    s.bad = pt_bad_update(s.bad, N, qa, qb, n);
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const bool h0 = qa == (uint32_t)j, h1 = qb == (uint32_t)j;
        uint64_t vx = s.X[j], vz = s.Z[j];
        vx = h1 ? n[2] : vx;  // flat selects; qa's value wins when qa == qb
        vz = h1 ? n[3] : vz;
        vx = h0 ? n[0] : vx;
        vz = h0 ? n[1] : vz;
        s.X[j] = vx;
        s.Z[j] = vz;
    }
}

// one micro-op on every rotation (Pauli::evolve_*, pauli.rs:83-110), branch-free in `kind`:
//   H(p):    x_p <-> z_p, phase += 2*(x_p & z_p)
//   S(p):    z_p ^= x_p,  phase += x_p
//   SX(p):   = H,S,H -> x_p ^= z_p, phase += 3*z_p
//   CNOT(i=p, j=q) = evolve_cx(ctrl=q, tgt=p): x_p ^= x_q ; z_q ^= z_p
// Returns the mask of rotations whose record changed (most gates miss most rotations' supports).
template <int NQ, int RM>
__device__ inline uint32_t pt_evolve(PTState<NQ, RM> &s, uint32_t kind, uint32_t p, uint32_t q) {
    const uint32_t isH = kind == M_H, isS = kind == M_S, isSX = kind == M_SX, isCN = kind == M_CNOT;
    uint32_t inc1 = 0, inc2 = 0, changed = 0;
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        const uint32_t bxp = (s.rx[k] >> p) & 1u, bzp = (s.rz[k] >> p) & 1u;
        const uint32_t bxq = (s.rx[k] >> q) & 1u;
        const uint32_t dxp = (isH & (bxp ^ bzp)) | (isSX & bzp) | (isCN & bxq);
        const uint32_t dzp = (isH & (bxp ^ bzp)) | (isS & bxp);
        const uint32_t dzq = isCN & bzp;
        s.rx[k] ^= dxp << p;
        s.rz[k] ^= (dzp << p) ^ (dzq << q);
        inc1 |= ((isS & bxp) | (isSX & bzp)) << k;
        inc2 |= ((isH & bxp & bzp) | (isSX & bzp)) << k;
        changed |= (dxp | dzp | dzq) << k;
    }
    const uint32_t carry = s.plo & inc1;  // phases += inc1 + 2*inc2 (mod 4), all rotations at once
    s.plo ^= inc1;
    s.phi ^= carry ^ inc2;
    return changed | inc1 | inc2;
}

// clean_and_return_with_phases (pauli_network.rs:139-165).  `log`: solution-log sink or null.
template <int NQ, int RM>
__device__ inline void pt_clean(PTState<NQ, RM> &s, uint32_t &n_removed, uint32_t &fault, SolLog log, uint64_t (&rem_pos)[(RM + 7) / 8]) {
    uint32_t trivial = 0, zero_w = 0;  // weights do not change while cleaning (:79-93)
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        const uint32_t sup = s.rx[k] | s.rz[k];
        trivial |= (uint32_t)(__popc(sup) <= 1) << k;
        zero_w |= (uint32_t)(sup == 0) << k;
    }
    for (;;) {
        uint32_t front = 0;  // get_front_layer (pauli_dag.rs:47-57): no out-edge to a live node
#pragma unroll
        for (int k = 0; k < RM; ++k) front |= (uint32_t)((s.rpred[k] & s.alive) == 0) << k;
        uint32_t doomed = front & trivial & s.alive;
        if (doomed & zero_w) {  // which_qubit(..).unwrap() on None (:113-114): the reference panics
            fault |= QG_FAULT_ZERO_WEIGHT;
            doomed &= ~zero_w;
        }
        if (!doomed) break;
        // removals are reported in DAG node order within a pass (:146-152)
        uint32_t dpos = 0;  // bit i: DAG node i is removed in this pass
#pragma unroll
        for (int i = 0; i < RM; ++i) dpos |= (((uint32_t)i < s.count) ? ((doomed >> s.order.get(i)) & 1u) : 0u) << i;
        if (log) {
#pragma unroll
            for (int k = 0; k < RM; ++k) {
                if ((doomed >> k) & 1u) {
                    uint32_t pos = 0;
#pragma unroll
                    for (int i = 0; i < RM; ++i) pos = (s.order.get(i) == (uint32_t)k && (uint32_t)i < s.count) ? (uint32_t)i : pos;
                    const uint32_t seq = n_removed + (uint32_t)__popc(dpos & ((1u << pos) - 1u));
                    const uint32_t sup = s.rx[k] | s.rz[k];
                    const uint32_t qb = (uint32_t)__ffs((int)sup) - 1u;  // which_qubit / which_axis (:95-137)
                    const uint32_t bx = (s.rx[k] >> qb) & 1u, bz = (s.rz[k] >> qb) & 1u;
                    const uint32_t axis = bx ? (bz ? 1u : 0u) : 2u;
                    log[seq] = 0x80000000u | (axis << 21) | (qb << 11) | ((uint32_t)k << 1);  // phase bit patched after the gate
                    rem_pos[k / 8] |= (uint64_t)(seq + 1u) << (8 * (k % 8));
                }
            }
        }
        n_removed += (uint32_t)__popc(doomed);
        // retain_nodes: visit NodeIndex high -> low, swap_remove each doomed node (:160-161)
#pragma unroll
        for (int i = RM - 1; i >= 0; --i) {
            if ((uint32_t)i < s.count && ((dpos >> i) & 1u)) {
                s.order.set((uint32_t)i, s.order.get(s.count - 1));
                s.count -= 1;
            }
        }
        s.alive &= ~doomed;
    }
}

// `bad` from the whole tableau (after an upload; the step kernels keep it incrementally)
template <int NQ, int RM>
__device__ inline uint32_t pt_badmask(const PTState<NQ, RM> &s, uint32_t N) {
    uint32_t bad = 0;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        const uint64_t ix = (uint32_t)j < N ? 1ull << j : 0ull, iz = (uint32_t)j < N ? (1ull << N) << j : 0ull;
        bad |= (uint32_t)(s.X[j] != ix || s.Z[j] != iz) << j;
    }
    return bad;
}
template <int NQ, int RM>
__device__ inline bool pt_solved(const PTState<NQ, RM> &s) {  // PauliNetwork::solved (:167-173)
    return s.count == 0 && s.bad == 0;
}

struct PTArgs {
    StepArgs s;
    const uint64_t *prog;
    const int32_t *act_perms;  // add_perms (pauli.rs:594-599)
    const uint32_t *perm_idx;
    uint32_t n_perms;
    uint32_t do_clean;
    int32_t depth_value;
};

template <int NQ, int RM, bool FEAT>
__global__ __launch_bounds__(256) void ptile_step_kernel(PTArgs pa) {
    const StepArgs &a = pa.s;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    if (env >= a.B) return;
    const bool act64 = a.flags & F_ACT64;
    const uint32_t N = a.N;
    char *tile = PTLayout<NQ, RM>::tile(a.state, env);

    int64_t act = load_action(a.actions, env, act64);
    PTState<NQ, RM> s;
    pt_load<NQ, RM>(tile, lane, s);
    int32_t depth = a.depth[env];
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    const uint32_t alive0 = s.alive, count0 = s.count, bad0 = s.bad;
    const PTOrder<RM> order0 = s.order;
    uint32_t touched_rot = 0;   // rotations whose record changed while they were alive
    uint32_t dirty_q = 0;       // qubits whose tableau rows changed
    bool solved = false;
    float reward = 0.0f;
    uint32_t fault = 0;

    for (uint32_t t = 0; t < a.T; ++t) {
        if (t) act = load_action(a.actions, (uint64_t)t * a.B + env, act64);
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
            if (FEAT && (a.flags & F_LAYERS)) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, a.descs[act], a.w);
        }
        const uint32_t qa = (uint32_t)prog & 31u, qb = (uint32_t)(prog >> 5) & 31u, m = (uint32_t)(prog >> 10) & 0xFFFFu;
        uint32_t n_removed = 0;
        uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
        for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
        SolLog log{nullptr, 0};
        if (FEAT && (a.flags & F_TRACK) && in_range && (uint32_t)sol_n + 1u + (uint32_t)s.count <= a.sol_cap)
            log = SolLog{&sol_at(a, env, (uint32_t)sol_n + 1u), a.B};  // slot sol_n is the gate itself

        if (in_range) {
            const uint32_t alive_at_gate = s.alive;
            pt_apply_tableau<NQ, RM>(s, N, qa, qb, m);
            dirty_q |= (1u << qa) | (1u << qb);
#pragma unroll 1
            for (uint32_t k = 0; k < 3; ++k) {  // PauliNetwork::act (pauli_network.rs:225-260)
                const uint32_t mo = (uint32_t)(prog >> (26 + 4 * k)) & 15u;
                const uint32_t kind = mo & 7u;
                if (kind == M_NOP) continue;
                const uint32_t p = (mo & 8u) ? qb : qa, q = (mo & 8u) ? qa : qb;
                touched_rot |= pt_evolve<NQ, RM>(s, kind, p, q) & alive_at_gate;  // dead rotations are never read again
                if (kind == M_CNOT) pt_clean<NQ, RM>(s, n_removed, fault, log, rem_pos);
            }
        }

        if (FEAT && (a.flags & F_TRACK) && in_range) {  // pauli.rs:612-626
            if (log) {
                sol_at(a, env, (uint32_t)sol_n) = sol_word(act);
                // phase_mult is read after the whole gate has been applied (pauli.rs:618)
#pragma unroll
                for (int k = 0; k < RM; ++k) {
                    const uint32_t pos = (uint32_t)(rem_pos[k / 8] >> (8 * (k % 8))) & 0xFFu;
                    if (pos) {
                        const uint32_t base = ((s.plo >> k) & 1u) | (((s.phi >> k) & 1u) << 1);
                        const uint32_t ph = (base + 4u * N - (uint32_t)__popc(s.rx[k] & s.rz[k])) & 3u;  // Pauli::phase (pauli.rs:125-133)
                        log[pos - 1u] |= (ph == 2u ? 0u : 1u);
                    }
                }
                sol_n += 1 + (int32_t)n_removed;
            } else {
                fault |= 8u;
            }
        }

        depth = depth > 0 ? depth - 1 : 0;  // pauli.rs:630
        solved = pt_solved<NQ, RM>(s);
        const float achieved = solved ? 1.0f : 0.0f;
        const float tmp = achieved - penalty;
        const float bonus = a.pauli_layer_reward * (float)n_removed;
        reward = tmp + bonus;  // pauli.rs:634
        if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
        if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
    }

    // write back what changed: the (<= 2 per step) touched qubits' row pairs, the rotations that
    // were alive when a gate was applied, the DAG bookkeeping
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if ((dirty_q >> q) & 1u) PTLayout<NQ, RM>::store_qubit(tile, lane, q, s.X[q], s.Z[q]);
    pt_store_rotations<NQ, RM>(tile, lane, s, touched_rot, dirty_q, false);
    if (s.alive != alive0 || s.count != count0 || s.order != order0 || s.bad != bad0) pt_store_meta<NQ, RM>(tile, lane, s);
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) a.sol_len[env * 2] = sol_n;
    if (fault) atomicOr(&a.error[env], fault);
}

// ---- the gate's micro-ops on all rotations at once (one-step kernel) ---------------------------------
// A gate's micro-ops only read and write bits qa / qb of every rotation's x and z masks.  Those bits are
// gathered once into four "slices" (bit k = rotation k), the micro-ops become a handful of logic ops on
// the slices (the 2-bit phases already are bit-planes), `clean` tests weights through the slices and two
// masks computed once per gate (b0 / b1: rotations whose support OUTSIDE {qa, qb} has 0 / exactly 1 qubit),
// and the slices are scattered back once.
template <int RM>
struct PTSlices {
    uint32_t xa, za, xb, zb;  // bit k: bit qa (qb) of rotation k's x (z) mask; with qa == qb the b-slices stay zero
    uint32_t b0, b1;
};

template <int NQ, int RM>
__device__ inline void pt_slices_gather(const PTState<NQ, RM> &s, uint32_t qa, uint32_t qb, PTSlices<RM> &v) {
    const uint32_t outside = ~((1u << qa) | (1u << qb));
    const uint32_t two = qa != qb;
    v.xa = v.za = v.xb = v.zb = v.b0 = v.b1 = 0;
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        v.xa |= ((s.rx[k] >> qa) & 1u) << k;
        v.za |= ((s.rz[k] >> qa) & 1u) << k;
        v.xb |= ((s.rx[k] >> qb) & two) << k;
        v.zb |= ((s.rz[k] >> qb) & two) << k;
        const uint32_t c = (uint32_t)__popc((s.rx[k] | s.rz[k]) & outside);
        v.b0 |= (uint32_t)(c == 0) << k;
        v.b1 |= (uint32_t)(c == 1) << k;
    }
}
template <int NQ, int RM>
__device__ inline void pt_slices_scatter(PTState<NQ, RM> &s, uint32_t qa, uint32_t qb, const PTSlices<RM> &v) {
    const uint32_t two = qa != qb, keep = ~((1u << qa) | (1u << qb));
#pragma unroll
    for (int k = 0; k < RM; ++k) {
        s.rx[k] = (s.rx[k] & keep) | (((v.xa >> k) & 1u) << qa) | ((((v.xb >> k) & 1u) & two) << qb);
        s.rz[k] = (s.rz[k] & keep) | (((v.za >> k) & 1u) << qa) | ((((v.zb >> k) & 1u) & two) << qb);
    }
}
// one micro-op (Pauli::evolve_*, pauli.rs:83-110), branch-free in the per-lane `kind` (M_NOP changes nothing);
// `on_b`: its first operand p is qubit qb.
//   H(p):  x_p <-> z_p, phase += 2 * (x_p & z_p)        S(p):  z_p ^= x_p, phase += x_p
//   SX(p): x_p ^= z_p, phase += 3 * z_p                  CNOT(i = p, j = q) = evolve_cx(ctrl = q, tgt = p): x_p ^= x_q ; z_q ^= z_p
template <int NQ, int RM>
__device__ inline void pt_slices_evolve(PTState<NQ, RM> &s, PTSlices<RM> &v, uint32_t kind, bool on_b) {
    const uint32_t ob = 0u - (uint32_t)on_b;
    const uint32_t xp = (v.xb & ob) | (v.xa & ~ob), zp = (v.zb & ob) | (v.za & ~ob);
    const uint32_t xq = (v.xa & ob) | (v.xb & ~ob), zq = (v.za & ob) | (v.zb & ~ob);
    const uint32_t H = 0u - (uint32_t)(kind == M_H), S = 0u - (uint32_t)(kind == M_S), SX = 0u - (uint32_t)(kind == M_SX),
                   CN = 0u - (uint32_t)(kind == M_CNOT);
    const uint32_t inc1 = (S & xp) | (SX & zp), inc2 = ((H & xp) | SX) & zp;
    const uint32_t nxp = xp ^ (H & (xp ^ zp)) ^ (SX & zp) ^ (CN & xq);
    const uint32_t nzp = zp ^ (H & (xp ^ zp)) ^ (S & xp);
    const uint32_t nzq = zq ^ (CN & zp);
    v.xa = (nxp & ~ob) | (xq & ob);
    v.za = (nzp & ~ob) | (nzq & ob);
    v.xb = (xq & ~ob) | (nxp & ob);
    v.zb = (nzq & ~ob) | (nzp & ob);
    const uint32_t carry = s.plo & inc1;  // phases += inc1 + 2 * inc2 (mod 4)
    s.plo ^= inc1;
    s.phi ^= carry ^ inc2;
}
// clean_and_return_with_phases (pauli_network.rs:139-165) on the slices; see pt_clean for the bookkeeping
// `outside(k, rx, rz)`: rotation k's masks with bits qa / qb cleared (only the solution log asks)
template <int NQ, int RM, typename Outside>
__device__ inline void pt_slices_clean(PTState<NQ, RM> &s, const PTSlices<RM> &v, uint32_t qa, uint32_t qb, uint32_t &n_removed, uint32_t &fault,
                                       SolLog log, uint64_t (&rem_pos)[(RM + 7) / 8], Outside outside) {
    const uint32_t sa = v.xa | v.za, sb = v.xb | v.zb;  // weights do not change while cleaning (:79-93)
    const uint32_t trivial = (v.b0 & ~(sa & sb)) | (v.b1 & ~(sa | sb)), zero_w = v.b0 & ~(sa | sb);
    for (;;) {
        uint32_t front = 0;  // get_front_layer (pauli_dag.rs:47-57): no out-edge to a live node
#pragma unroll
        for (int k = 0; k < RM; ++k) front |= (uint32_t)((s.rpred[k] & s.alive) == 0) << k;
        uint32_t doomed = front & trivial & s.alive;
        if (doomed & zero_w) {  // which_qubit(..).unwrap() on None (:113-114): the reference panics
            fault |= QG_FAULT_ZERO_WEIGHT;
            doomed &= ~zero_w;
        }
        if (!doomed) break;
        uint32_t dpos = 0;  // bit i: DAG node i is removed in this pass (removals are reported in node order, :146-152)
#pragma unroll
        for (int i = 0; i < RM; ++i) dpos |= (((uint32_t)i < s.count) ? ((doomed >> s.order.get(i)) & 1u) : 0u) << i;
        if (log) {
#pragma unroll
            for (int k = 0; k < RM; ++k) {
                if ((doomed >> k) & 1u) {
                    uint32_t pos = 0;
#pragma unroll
                    for (int i = 0; i < RM; ++i) pos = (s.order.get(i) == (uint32_t)k && (uint32_t)i < s.count) ? (uint32_t)i : pos;
                    const uint32_t seq = n_removed + (uint32_t)__popc(dpos & ((1u << pos) - 1u));
                    // which_qubit / which_axis (:95-137): the single support qubit is outside {qa, qb} (b1), qa or qb
                    uint32_t q, bx, bz;
                    if ((v.b1 >> k) & 1u) {
                        uint32_t ox, oz;
                        outside((uint32_t)k, ox, oz);
                        q = (uint32_t)__ffs((int)(ox | oz)) - 1u;
                        bx = (ox >> q) & 1u;
                        bz = (oz >> q) & 1u;
                    } else if ((sa >> k) & 1u) {
                        q = qa; bx = (v.xa >> k) & 1u; bz = (v.za >> k) & 1u;
                    } else {
                        q = qb; bx = (v.xb >> k) & 1u; bz = (v.zb >> k) & 1u;
                    }
                    const uint32_t axis = bx ? (bz ? 1u : 0u) : 2u;
                    log[seq] = 0x80000000u | (axis << 21) | (q << 11) | ((uint32_t)k << 1);  // phase bit patched after the gate
                    rem_pos[k / 8] |= (uint64_t)(seq + 1u) << (8 * (k % 8));
                }
            }
        }
        n_removed += (uint32_t)__popc(doomed);
#pragma unroll
        for (int i = RM - 1; i >= 0; --i) {  // retain_nodes: visit NodeIndex high -> low, swap_remove each doomed node (:160-161)
            if ((uint32_t)i < s.count && ((dpos >> i) & 1u)) {
                s.order.set((uint32_t)i, s.order.get(s.count - 1));
                s.count -= 1;
            }
        }
        s.alive &= ~doomed;
    }
}

// One step per launch (PauliEnv::step, pauli.rs:588-635): rotations and bookkeeping in registers,
// the rows of the gate's qubits gathered from / scattered to the tile at per-lane addresses.
template <int NQ, int RM, bool FEAT>
__global__ __launch_bounds__(256) void ptile_step1_kernel(PTArgs pa) {
    using L = PTLayout<NQ, RM>;
    const StepArgs &a = pa.s;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    asm volatile("" ::"s"(pa.prog), "s"(pa.n_perms));
    if (env >= a.B) return;
    const uint32_t N = a.N;
    char *tile = L::tile(a.state, env);

    int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    PTState<NQ, RM> s;
    pt_load_rotations<NQ, RM>(tile, lane, s);
    int32_t depth = a.depth[env];
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    const uint32_t alive0 = s.alive, count0 = s.count, bad0 = s.bad;
    const PTOrder<RM> order0 = s.order;
    uint32_t touched_rot = 0, fault = 0;

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
        if (FEAT && (a.flags & F_LAYERS)) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, a.descs[act], a.w);
    }
    const uint32_t qa = (uint32_t)prog & 31u, qb = (uint32_t)(prog >> 5) & 31u, m = (uint32_t)(prog >> 10) & 0xFFFFu;
    uint32_t n_removed = 0;
    uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
    for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
    SolLog log{nullptr, 0};
    if (FEAT && (a.flags & F_TRACK) && in_range && (uint32_t)sol_n + 1u + (uint32_t)s.count <= a.sol_cap)
        log = SolLog{&sol_at(a, env, (uint32_t)sol_n + 1u), a.B};  // slot sol_n is the gate itself

    if (in_range) {
        uint64_t xa, za, xb, zb, n[4];
        L::load_qubit(tile, lane, qa, xa, za);
        L::load_qubit(tile, lane, qb, xb, zb);  // one-qubit gates: qb == qa, the same (cached) record
        pt_mix(m, xa, za, xb, zb, n);
        if (qb != qa) L::store_qubit(tile, lane, qb, n[2], n[3]);
        L::store_qubit(tile, lane, qa, n[0], n[1]);
        s.bad = pt_bad_update(s.bad, N, qa, qb, n);
        const uint32_t alive_at_gate = s.alive, plo0 = s.plo, phi0 = s.phi;
        PTSlices<RM> v;
        pt_slices_gather<NQ, RM>(s, qa, qb, v);
        const PTSlices<RM> v0 = v;
#pragma unroll 1
        for (uint32_t k = 0; k < 3; ++k) {  // PauliNetwork::act (pauli_network.rs:225-260)
            const uint32_t mo = (uint32_t)(prog >> (26 + 4 * k)) & 15u;
            const uint32_t kind = mo & 7u;
            if (kind == M_NOP) continue;
            pt_slices_evolve<NQ, RM>(s, v, kind, (mo & 8u) != 0);
            if (kind == M_CNOT)
                pt_slices_clean<NQ, RM>(s, v, qa, qb, n_removed, fault, log, rem_pos, [&](uint32_t k, uint32_t &ox, uint32_t &oz) {
                    const uint32_t keep = ~((1u << qa) | (1u << qb));
                    ox = oz = 0;
#pragma unroll
                    for (int i = 0; i < RM; ++i) {  // k is a per-lane value
                        ox = (uint32_t)i == k ? s.rx[i] & keep : ox;
                        oz = (uint32_t)i == k ? s.rz[i] & keep : oz;
                    }
                });
        }
        pt_slices_scatter<NQ, RM>(s, qa, qb, v);
        // records to write back: rotations alive at the gate whose bits or phase changed (dead ones are never read again)
        touched_rot = ((v.xa ^ v0.xa) | (v.za ^ v0.za) | (v.xb ^ v0.xb) | (v.zb ^ v0.zb) | (s.plo ^ plo0) | (s.phi ^ phi0)) & alive_at_gate;
    }

    if (FEAT && (a.flags & F_TRACK) && in_range) {  // pauli.rs:612-626
        if (log) {
            sol_at(a, env, (uint32_t)sol_n) = sol_word(act);
#pragma unroll
            for (int k = 0; k < RM; ++k) {  // phase_mult is read after the whole gate has been applied (pauli.rs:618)
                const uint32_t pos = (uint32_t)(rem_pos[k / 8] >> (8 * (k % 8))) & 0xFFu;
                if (pos) {
                    const uint32_t base = ((s.plo >> k) & 1u) | (((s.phi >> k) & 1u) << 1);
                    const uint32_t ph = (base + 4u * N - (uint32_t)__popc(s.rx[k] & s.rz[k])) & 3u;  // Pauli::phase (pauli.rs:125-133)
                    log[pos - 1u] |= (ph == 2u ? 0u : 1u);
                }
            }
            sol_n += 1 + (int32_t)n_removed;
        } else {
            fault |= 8u;
        }
    }

    depth = depth > 0 ? depth - 1 : 0;  // pauli.rs:630
    const bool solved = pt_solved<NQ, RM>(s);
    const float achieved = solved ? 1.0f : 0.0f;
    const float tmp = achieved - penalty;
    const float bonus = a.pauli_layer_reward * (float)n_removed;
    const float reward = tmp + bonus;  // pauli.rs:634
    if (a.rewards_seq) a.rewards_seq[env] = reward;
    if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
    pt_store_rotations<NQ, RM>(tile, lane, s, touched_rot, in_range ? (1u << qa) | (1u << qb) : 0u, false);
    if (s.alive != alive0 || s.count != count0 || s.order != order0 || s.bad != bad0) pt_store_meta<NQ, RM>(tile, lane, s);
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) a.sol_len[env * 2] = sol_n;
    if (fault) atomicOr(&a.error[env], fault);
}

// bit-sliced counters: plane j holds bit j of eight small integers (one per rotation); +/- a 0/1 per rotation
__device__ inline void planes_sub(uint32_t (&w)[5], uint32_t bits) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t t = w[j];
        w[j] = t ^ bits;
        bits &= ~t;
    }
}
__device__ inline void planes_add(uint32_t (&w)[5], uint32_t bits) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t t = w[j];
        w[j] = t ^ bits;
        bits &= t;
    }
}

// One step per launch on the compact layout: the transposed rotation region hands the kernel the gate's
// bit-slices directly (two 16-bit gathers), the weights live as bit-planes, so no per-rotation mask is ever
// loaded: per env the step reads two qubit records, two slice pairs and 32 bytes of bookkeeping.
// (the body of ptile_step1c_kernel for one env; returns is_final)
template <int NQ, int RM, bool FEAT>
__device__ __forceinline__ bool ptile_step1c_body(const PTArgs &pa, uint64_t env, uint32_t lane) {
    using L = PTLayout<NQ, RM>;
    static_assert(L::COMPACT && RM == 8, "compact layout only");
    const StepArgs &a = pa.s;
    const uint32_t N = a.N;
    char *tile = L::tile(a.state, env);

    int64_t act = load_action(a.actions, env, a.flags & F_ACT64);
    PTState<NQ, RM> s;  // rpred / phases / bookkeeping only; the masks stay in memory
    uint4 m0 = *L::meta(tile, lane);
    uint2 pv = *L::rotgroup(tile, lane, 6), pw0 = *L::rotgroup(tile, lane, 7);
    int32_t depth = a.depth[env];
    // keep these loads up here, in flight together with the action load: left alone, the compiler sinks the ones only
    // `clean` reads into that branch and every cnot pays another memory round trip (measured: 1.3 us per step)
    asm volatile("" : "+v"(m0.x), "+v"(m0.y), "+v"(m0.z), "+v"(m0.w), "+v"(pv.x), "+v"(pv.y), "+v"(pw0.x), "+v"(pw0.y), "+v"(depth));
    s.alive = m0.x & 0xFFFFu;
    s.count = m0.x >> 16;
    s.bad = m0.y;
    s.order.w = (uint64_t)m0.z | ((uint64_t)m0.w << 32);
#pragma unroll
    for (int k = 0; k < RM; ++k) s.rpred[k] = ((k < 4 ? pv.x : pv.y) >> (8 * (k & 3))) & 0xFFu;
    s.plo = pw0.x & 0xFFu;
    s.phi = (pw0.x >> 8) & 0xFFu;
    uint32_t w[5] = {(pw0.x >> 16) & 0xFFu, pw0.x >> 24, pw0.y & 0xFFu, (pw0.y >> 8) & 0xFFu, (pw0.y >> 16) & 0xFFu};
    int32_t sol_n = (FEAT && (a.flags & F_TRACK)) ? a.sol_len[env * 2] : 0;
    uint32_t fault = 0;

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
        if (FEAT && (a.flags & F_LAYERS)) penalty = layers_penalty(layer_rec(a.layers, env, 2 * N + 2), N, a.descs[act], a.w);
    }
    const uint32_t qa = (uint32_t)prog & 31u, qb = (uint32_t)(prog >> 5) & 31u, m = (uint32_t)(prog >> 10) & 0xFFFFu;
    uint32_t n_removed = 0;
    uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
    for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
    SolLog log{nullptr, 0};
    if (FEAT && (a.flags & F_TRACK) && in_range && (uint32_t)sol_n + 1u + (uint32_t)s.count <= a.sol_cap)
        log = SolLog{&sol_at(a, env, (uint32_t)sol_n + 1u), a.B};  // slot sol_n is the gate itself

    // rotation k's masks outside {qa, qb}, from the transposed bytes (solution log only: a removal is rare)
    auto outside = [&](uint32_t k, uint32_t &ox, uint32_t &oz) {
        ox = oz = 0;
        for (uint32_t q = 0; q < (uint32_t)NQ; ++q) {
            const uint32_t v = *L::xz(tile, lane, q);
            ox |= ((v >> k) & 1u) << q;
            oz |= ((v >> (8u + k)) & 1u) << q;
        }
        ox &= ~((1u << qa) | (1u << qb));
        oz &= ~((1u << qa) | (1u << qb));
    };

    PTSlices<RM> v;
    v.xa = v.za = v.xb = v.zb = v.b0 = v.b1 = 0;
    if (in_range) {
        uint64_t xa, za, xb, zb, n[4];
        L::load_qubit(tile, lane, qa, xa, za);
        L::load_qubit(tile, lane, qb, xb, zb);  // one-qubit gates: qb == qa, the same (cached) record
        const uint32_t two = qa != qb;
#ifdef QG_ABLATE_SLICES  // development build (tools/build_variant.sh): what the step would cost if the gate's rotation slices came with its qubit records -- results are WRONG
        const uint32_t sva = 0u, svb = 0u;
#else
        const uint32_t sva = *L::xz(tile, lane, qa), svb = *L::xz(tile, lane, qb);
#endif
        pt_mix(m, xa, za, xb, zb, n);
        if (two) L::store_qubit(tile, lane, qb, n[2], n[3]);
        L::store_qubit(tile, lane, qa, n[0], n[1]);
        s.bad = pt_bad_update(s.bad, N, qa, qb, n);
        v.xa = sva & 0xFFu;
        v.za = sva >> 8;
        v.xb = two ? svb & 0xFFu : 0u;
        v.zb = two ? svb >> 8 : 0u;
        planes_sub(w, v.xa | v.za);  // weights outside {qa, qb}
        planes_sub(w, v.xb | v.zb);
        const uint32_t hi = w[1] | w[2] | w[3] | w[4];
        v.b0 = ~(w[0] | hi) & 0xFFu;
        v.b1 = w[0] & ~hi & 0xFFu;
        const PTSlices<RM> v0 = v;
#pragma unroll
        for (uint32_t k = 0; k < 3; ++k) {  // PauliNetwork::act (pauli_network.rs:225-260)
            const uint32_t mo = (uint32_t)(prog >> (26 + 4 * k)) & 15u;
            const uint32_t kind = mo & 7u;
            pt_slices_evolve<NQ, RM>(s, v, kind, (mo & 8u) != 0);
            if (kind == M_CNOT) pt_slices_clean<NQ, RM>(s, v, qa, qb, n_removed, fault, log, rem_pos, outside);
        }
        planes_add(w, v.xa | v.za);
        planes_add(w, v.xb | v.zb);
#ifndef QG_ABLATE_SLICES
        if ((v.xa ^ v0.xa) | (v.za ^ v0.za)) *L::xz(tile, lane, qa) = (uint16_t)(v.xa | (v.za << 8));
        if ((v.xb ^ v0.xb) | (v.zb ^ v0.zb)) *L::xz(tile, lane, qb) = (uint16_t)(v.xb | (v.zb << 8));
#endif
    }

    if (FEAT && (a.flags & F_TRACK) && in_range) {  // pauli.rs:612-626
        if (log) {
            sol_at(a, env, (uint32_t)sol_n) = sol_word(act);
#pragma unroll
            for (int k = 0; k < RM; ++k) {  // phase_mult is read after the whole gate has been applied (pauli.rs:618)
                const uint32_t pos = (uint32_t)(rem_pos[k / 8] >> (8 * (k % 8))) & 0xFFu;
                if (pos) {
                    uint32_t ox, oz;
                    outside((uint32_t)k, ox, oz);
                    const uint32_t ys = (uint32_t)__popc(ox & oz) + (((v.xa & v.za) >> k) & 1u) + (((v.xb & v.zb) >> k) & 1u);
                    const uint32_t base = ((s.plo >> k) & 1u) | (((s.phi >> k) & 1u) << 1);
                    const uint32_t ph = (base + 4u * N - ys) & 3u;  // Pauli::phase (pauli.rs:125-133)
                    log[pos - 1u] |= (ph == 2u ? 0u : 1u);
                }
            }
            sol_n += 1 + (int32_t)n_removed;
        } else {
            fault |= 8u;
        }
    }

    depth = depth > 0 ? depth - 1 : 0;  // pauli.rs:630
    const bool solved = pt_solved<NQ, RM>(s);
    const float achieved = solved ? 1.0f : 0.0f;
    const float tmp = achieved - penalty;
    const float bonus = a.pauli_layer_reward * (float)n_removed;
    const float reward = tmp + bonus;  // pauli.rs:634
    if (a.rewards_seq) a.rewards_seq[env] = reward;
    if (a.dones_seq) a.dones_seq[env] = (uint8_t)(depth == 0 || solved);
    const uint2 pw = make_uint2((s.plo & 0xFFu) | ((s.phi & 0xFFu) << 8) | (w[0] << 16) | (w[1] << 24), w[2] | (w[3] << 8) | (w[4] << 16));
    if (pw.x != pw0.x || pw.y != pw0.y) *L::rotgroup(tile, lane, 7) = pw;
    const uint4 m1 = make_uint4(s.alive | (s.count << 16), s.bad, (uint32_t)s.order.w, (uint32_t)(s.order.w >> 32));
    if (m1.x != m0.x || m1.y != m0.y || m1.z != m0.z || m1.w != m0.w) *L::meta(tile, lane) = m1;
    a.depth[env] = depth;
    a.reward[env] = reward;
    a.done[env] = (uint8_t)(depth == 0 || solved);
    a.success[env] = (uint8_t)solved;
    if (FEAT && (a.flags & F_TRACK)) a.sol_len[env * 2] = sol_n;
    if (fault) atomicOr(&a.error[env], fault);
    return depth == 0 || solved;
}
// LIST (F_DONE_LIST): the envs that finish are left as one bit each in StepArgs::done_mask for the qg_vec_reset_done that follows (device_common.hpp
// done_mask_store: no compaction launch); its own instantiation, the plain kernel's code stays as it is
template <int NQ, int RM, bool FEAT, bool LIST = false>
__global__ __launch_bounds__(256) void ptile_step1c_kernel(PTArgs pa) {
    KernelClock kclk(pa.s.kclk, pa.s.kclk_waves);  // device_common.hpp
    const StepArgs &a = pa.s;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    QG_PREFETCH_STEP_ARGS(a);  // qgym_internal.hpp
    asm volatile("" ::"s"(pa.prog), "s"(pa.n_perms));
    if constexpr (LIST) {  // every thread reaches the wave's ballot
        bool fin = false;
        if (env < a.B) fin = ptile_step1c_body<NQ, RM, FEAT>(pa, env, lane);
        done_mask_store(a.done_mask, a.B, fin, env, a.done_epoch);
    } else {
        if (env >= a.B) return;
        (void)ptile_step1c_body<NQ, RM, FEAT>(pa, env, lane);
    }
}

// Fused rollout on the compact layout (T steps per launch; plain configuration: no solution log, default weights, no
// action permutation).  The dense fused kernel keeps all 2N rows in VGPRs and pays select trees over them for every gate;
// here the wave's tile is copied verbatim into LDS (its [group][lane] form is bank-conflict-free for per-lane groups: 12- and
// 8-byte lane strides), every step is the one-step kernel's gather / mix / scatter against LDS, and the DAG bookkeeping,
// phase and weight planes stay in registers for the whole rollout.  Actions are fetched two steps ahead and the gate
// program one step ahead with unconditional (clamped) loads, so the per-step chain never waits for global memory.
template <int NQ, int RM>
__global__ __launch_bounds__(64) void ptile_fused1c_kernel(PTArgs pa) {
    using L = PTLayout<NQ, RM>;
    static_assert(L::COMPACT && RM == 8, "compact layout only");
    constexpr uint32_t VEC = L::TILE_BYTES / 16u;  // uint4 per tile, a multiple of 64
    static_assert(L::TILE_BYTES % 1024u == 0, "tile = whole 1 KiB groups");
    __shared__ uint4 lds_tile[VEC];
    const StepArgs &a = pa.s;
    const uint32_t lane = threadIdx.x;
    const uint64_t env = (uint64_t)blockIdx.x * QG_WAVE + lane;
    uint4 *gtile = reinterpret_cast<uint4 *>(L::tile(a.state, (uint64_t)blockIdx.x * QG_WAVE));
#pragma unroll 4
    for (uint32_t i = 0; i < VEC; i += QG_WAVE) lds_tile[i + lane] = gtile[i + lane];
    __syncthreads();
    char *tile = reinterpret_cast<char *>(lds_tile);
    const uint32_t N = a.N;

    if (env < a.B) {
        const bool act64 = a.flags & F_ACT64;
        PTState<NQ, RM> s;  // rpred / phases / bookkeeping only
        const uint4 m0 = *L::meta(tile, lane);
        const uint2 pv = *L::rotgroup(tile, lane, 6), pw0 = *L::rotgroup(tile, lane, 7);
        int32_t depth = a.depth[env];
        s.alive = m0.x & 0xFFFFu;
        s.count = m0.x >> 16;
        s.bad = m0.y;
        s.order.w = (uint64_t)m0.z | ((uint64_t)m0.w << 32);
#pragma unroll
        for (int k = 0; k < RM; ++k) s.rpred[k] = ((k < 4 ? pv.x : pv.y) >> (8 * (k & 3))) & 0xFFu;
        s.plo = pw0.x & 0xFFu;
        s.phi = (pw0.x >> 8) & 0xFFu;
        uint32_t w[5] = {(pw0.x >> 16) & 0xFFu, pw0.x >> 24, pw0.y & 0xFFu, (pw0.y >> 8) & 0xFFu, (pw0.y >> 16) & 0xFFu};
        uint32_t fault = 0;
        bool solved = false;
        float reward = 0.0f;

        auto fetch = [&](uint32_t t) { return load_action(a.actions, (uint64_t)(t < a.T ? t : a.T - 1u) * a.B + env, act64); };
        struct Decoded { uint64_t prog; float penalty; bool in_range; };
        auto decode = [&](int64_t act) {
            Decoded d;
            d.in_range = act >= 0 && act < (int64_t)a.num_actions;  // pauli.rs:601
            const int64_t i = d.in_range ? act : 0;
            d.prog = pa.prog[i];
            d.penalty = a.gates[i].penalty;
            return d;
        };
        Decoded nxt = decode(fetch(0));
        int64_t act1 = fetch(1);
        for (uint32_t t = 0; t < a.T; ++t) {
            const Decoded cur = nxt;
            const int64_t act2 = fetch(t + 2u);
            nxt = decode(act1);
            act1 = act2;
            const uint64_t prog = cur.prog;
            const float penalty = cur.in_range ? cur.penalty : 0.0f;
            const uint32_t qa = (uint32_t)prog & 31u, qb = (uint32_t)(prog >> 5) & 31u, m = (uint32_t)(prog >> 10) & 0xFFFFu;
            uint32_t n_removed = 0;
            uint64_t rem_pos[(RM + 7) / 8] = {};
            if (cur.in_range) {
                uint64_t xa, za, xb, zb, n[4];
                L::load_qubit(tile, lane, qa, xa, za);
                L::load_qubit(tile, lane, qb, xb, zb);
                const uint32_t two = qa != qb;
                const uint32_t sva = *L::xz(tile, lane, qa), svb = *L::xz(tile, lane, qb);
                pt_mix(m, xa, za, xb, zb, n);
                if (two) L::store_qubit(tile, lane, qb, n[2], n[3]);
                L::store_qubit(tile, lane, qa, n[0], n[1]);
                s.bad = pt_bad_update(s.bad, N, qa, qb, n);
                PTSlices<RM> v;
                v.xa = sva & 0xFFu;
                v.za = sva >> 8;
                v.xb = two ? svb & 0xFFu : 0u;
                v.zb = two ? svb >> 8 : 0u;
                planes_sub(w, v.xa | v.za);  // weights outside {qa, qb}
                planes_sub(w, v.xb | v.zb);
                const uint32_t hi = w[1] | w[2] | w[3] | w[4];
                v.b0 = ~(w[0] | hi) & 0xFFu;
                v.b1 = w[0] & ~hi & 0xFFu;
                const PTSlices<RM> v0 = v;
#pragma unroll
                for (uint32_t k = 0; k < 3; ++k) {  // PauliNetwork::act (pauli_network.rs:225-260)
                    const uint32_t mo = (uint32_t)(prog >> (26 + 4 * k)) & 15u;
                    const uint32_t kind = mo & 7u;
                    pt_slices_evolve<NQ, RM>(s, v, kind, (mo & 8u) != 0);
                    if (kind == M_CNOT)
                        pt_slices_clean<NQ, RM>(s, v, qa, qb, n_removed, fault, SolLog{nullptr, 0}, rem_pos, [](uint32_t, uint32_t &ox, uint32_t &oz) { ox = oz = 0; });
                }
                planes_add(w, v.xa | v.za);
                planes_add(w, v.xb | v.zb);
                if ((v.xa ^ v0.xa) | (v.za ^ v0.za)) *L::xz(tile, lane, qa) = (uint16_t)(v.xa | (v.za << 8));
                if ((v.xb ^ v0.xb) | (v.zb ^ v0.zb)) *L::xz(tile, lane, qb) = (uint16_t)(v.xb | (v.zb << 8));
            }
            depth = depth > 0 ? depth - 1 : 0;  // pauli.rs:630
            solved = pt_solved<NQ, RM>(s);
            const float achieved = solved ? 1.0f : 0.0f;
            const float tmp = achieved - penalty;
            const float bonus = a.pauli_layer_reward * (float)n_removed;
            reward = tmp + bonus;  // pauli.rs:634
            if (a.rewards_seq) a.rewards_seq[(uint64_t)t * a.B + env] = reward;
            if (a.dones_seq) a.dones_seq[(uint64_t)t * a.B + env] = (uint8_t)(depth == 0 || solved);
        }
        *L::rotgroup(tile, lane, 7) = make_uint2((s.plo & 0xFFu) | ((s.phi & 0xFFu) << 8) | (w[0] << 16) | (w[1] << 24), w[2] | (w[3] << 8) | (w[4] << 16));
        *L::meta(tile, lane) = make_uint4(s.alive | (s.count << 16), s.bad, (uint32_t)s.order.w, (uint32_t)(s.order.w >> 32));
        a.depth[env] = depth;
        a.reward[env] = reward;
        a.done[env] = (uint8_t)(depth == 0 || solved);
        a.success[env] = (uint8_t)solved;
        if (fault) atomicOr(&a.error[env], fault);
    }
    __syncthreads();
#pragma unroll 4
    for (uint32_t i = 0; i < VEC; i += QG_WAVE) gtile[i + lane] = lds_tile[i + lane];
}

// after a host upload: optional initial clean (PauliEnv::reset, pauli.rs:576), then the scalar
// resets of set_state (pauli.rs:544-551) / reset (pauli.rs:578-585)
template <int NQ, int RM>
__global__ __launch_bounds__(256) void ptile_init_kernel(PTArgs pa) {
    const StepArgs &a = pa.s;
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    if (env >= a.B) return;
    char *tile = PTLayout<NQ, RM>::tile(a.state, env);
    PTState<NQ, RM> s;
    pt_load<NQ, RM>(tile, lane, s);
    uint32_t n_removed = 0, fault = 0;
    uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
    for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
    if (pa.do_clean) pt_clean<NQ, RM>(s, n_removed, fault, SolLog{nullptr, 0}, rem_pos);
    s.bad = pt_badmask<NQ, RM>(s, a.N);
    const bool solved = pt_solved<NQ, RM>(s);
    pt_store_meta<NQ, RM>(tile, lane, s);
    a.depth[env] = pa.depth_value;
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(pa.depth_value == 0 || solved);
    a.inverted[env] = 0;
    a.error[env] = fault;
    a.sol_len[env * 2] = 0;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, (2 * a.N + 2));
        for (uint32_t i = 0; i < 2 * a.N; ++i) lay[i] = -1;
        lay[2 * a.N] = 0;
        lay[2 * a.N + 1] = 0;
    }
}

// observe / get_state: one thread per (env, observation row)
struct PTObsArgs {
    ObsArgs o;
    uint32_t nq, rm, max_rot;
    const uint8_t *qubit_perms;  // add_perms (pauli.rs:653-665, 445-485)
    uint32_t *perm_idx;
    const int32_t *perm_in;
    uint32_t n_perms, draw;
    uint64_t seed, counter;
    const uint64_t *clock;
    uint64_t env_base;  // qg_vec_set_env_base
};
// the two observation rows qubit `q` of one env owns: tableau rows q and N + q (2N bits each, qubit-permuted
// when add_perms) in wx / wz, and the active rotations' bits for those rows (DAG node order,
// pad_and_collect pauli.rs:411-437) in ex / ez
__device__ inline void ptile_obs_qubit(const PTObsArgs &pa, uint64_t env, uint32_t q, bool want_extra, uint64_t &wx, uint64_t &wz, uint32_t &ex,
                                       uint32_t &ez) {
    const ObsArgs &a = pa.o;
    const uint32_t N = a.N, D = 2 * N;
    const uint32_t cols = a.obs_cols, lane = (uint32_t)(env & 63);
    const bool compact = pa.nq <= 24 && pa.rm == 8;  // PTLayout::COMPACT
    const uint32_t QB = compact ? 768u : 1024u, RB = compact ? 512u : 1024u;
    const char *tile = reinterpret_cast<const char *>(a.state) + (env >> 6) * (uint64_t)(pa.nq * QB + pa.rm * RB + (pa.rm > 16 ? 3072u : 1024u));
    const uint8_t *perm = nullptr;
    if (pa.n_perms && cols > D && a.format != QG_FMT_PACKED) {
        uint32_t pi;
        if (pa.draw) {  // `rng.gen_range(0..qubit_perms.len())` (pauli.rs:660), made reproducible
            pi = pa.perm_in ? (uint32_t)pa.perm_in[env] % pa.n_perms
                            : (uint32_t)__umul64hi(rng_draw(pa.seed ^ 0x7065726Dull, pa.env_base + env, pa.counter + clock_of(pa.clock)), (uint64_t)pa.n_perms);
            if (q == 0) pa.perm_idx[env] = pi;  // current_perm_idx.store (pauli.rs:661)
        } else {
            pi = pa.perm_idx[env];
        }
        perm = pa.qubit_perms + (uint64_t)pi * N;
    }
    const uint32_t sq = perm ? perm[q] : q;  // row i takes data from row perm[i] (pauli.rs:455-464)
    if (compact) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(tile + sq * QB + lane * 12u);
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        wx = (uint64_t)d0 | ((uint64_t)(d1 & 0xFFFFu) << 32);
        wz = (uint64_t)(d1 >> 16) | ((uint64_t)d2 << 16);
    } else {
        const uint4 t = *reinterpret_cast<const uint4 *>(tile + sq * QB + lane * 16u);
        wx = (uint64_t)t.x | ((uint64_t)t.y << 32);
        wz = (uint64_t)t.z | ((uint64_t)t.w << 32);
    }
    if (perm) {  // column i takes data from column perm[i], X and Z halves alike (pauli.rs:469-477)
        uint64_t px = 0, pz = 0;
        for (uint32_t i = 0; i < N; ++i) {
            px |= (((wx >> perm[i]) & 1ull) << i) | (((wx >> (N + perm[i])) & 1ull) << (N + i));
            pz |= (((wz >> perm[i]) & 1ull) << i) | (((wz >> (N + perm[i])) & 1ull) << (N + i));
        }
        wx = px;
        wz = pz;
    }
    ex = ez = 0;
    if (want_extra && cols > D) {
        const char *mt = tile + pa.nq * QB + pa.rm * RB;  // DAG bookkeeping (pt_load_meta)
        const uint4 m = *reinterpret_cast<const uint4 *>(mt + lane * 16u);
        const uint64_t order = (uint64_t)m.z | ((uint64_t)m.w << 32);
        const uint32_t count = pa.rm > 16 ? m.y : m.x >> 16, shown = count < pa.max_rot ? count : pa.max_rot;
        auto node = [&](uint32_t i) -> uint32_t {  // rotation held by DAG node i
            return pa.rm > 16 ? (uint32_t) reinterpret_cast<const uint8_t *>(mt + 1024u * (1u + (i >> 4)) + lane * 16u)[i & 15u] : pnib(order, i);
        };
        if (compact) {  // transposed rotation bytes: bit k of xs / zs = rotation k at this qubit
            const uint32_t v = *reinterpret_cast<const uint16_t *>(tile + pa.nq * QB + (sq >> 2) * RB + lane * 8u + (sq & 3u) * 2u);
            for (uint32_t i = 0; i < shown; ++i) {
                const uint32_t k = node(i);
                ex |= ((v >> k) & 1u) << i;
                ez |= ((v >> (8 + k)) & 1u) << i;
            }
        } else {
            for (uint32_t i = 0; i < shown; ++i) {
                const uint4 r = *reinterpret_cast<const uint4 *>(tile + pa.nq * QB + node(i) * RB + lane * 16u);
                ex |= ((r.x >> sq) & 1u) << i;
                ez |= ((r.y >> sq) & 1u) << i;
            }
        }
    }
}

// dense observation, first half: one 64-bit word per (env, row) = tableau bits | rotation bits << 2N;
// one thread per (env, qubit) writes the qubit's two rows
__global__ __launch_bounds__(256) void ptile_rowwords_kernel(PTObsArgs pa, uint64_t *words) {
    const uint32_t N = pa.o.N, D = 2 * N;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid <= 0xFFFFFFFFull ? (uint64_t)((uint32_t)gid / N) : gid / N;
    if (env >= pa.o.B) return;
    const uint32_t q = (uint32_t)(gid - env * N);
    uint64_t wx, wz;
    uint32_t ex, ez;
    ptile_obs_qubit(pa, env, q, true, wx, wz, ex, ez);
    words[env * D + q] = wx | ((uint64_t)ex << D);
    words[env * D + N + q] = wz | ((uint64_t)ez << D);
}

__global__ __launch_bounds__(256) void ptile_export_kernel(PTObsArgs pa) {
    const ObsArgs &a = pa.o;
    const uint32_t N = a.N, D = 2 * N;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t env = gid / D;
    const uint32_t row = (uint32_t)(gid % D);
    if (env >= a.B) return;
    const uint32_t cols = a.obs_cols;
    uint64_t wx, wz;
    uint32_t ex, ez;
    ptile_obs_qubit(pa, env, row < N ? row : row - N, a.format != QG_FMT_PACKED, wx, wz, ex, ez);
    const uint64_t w = row < N ? wx : wz;
    const uint32_t extra = row < N ? ex : ez;
    if (a.format == QG_FMT_PACKED) {
        reinterpret_cast<uint64_t *>(a.out)[env * a.out_stride + row] = w;
        return;
    }
    // pad_and_collect (pauli.rs:411-437): tableau, then the active rotations in DAG node order
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


// ------------------------------------------------------------------------------------------------
// PauliEnv::reset on the device (pauli.rs:554-586): the random target generator
// (get_pauli_under_diff / generate_paulis_with_difficulty / random_clifford_tableau, pauli.rs:115-271)
// one env per lane, every draw from the env's two counter-RNG streams (PT_STREAM_LABELS, PT_STREAM_TABLEAU below)
// -- the streams the oracle-side tests replay -- followed
// by the initial clean and the scalar resets.  Coupling-graph tables come from the host.
// ------------------------------------------------------------------------------------------------
struct PTGenArgs {
    StepArgs s;
    const uint8_t *pairs;      // [n_pairs][2]: qubit pairs q1<q2 grouped by graph distance, (q1,q2) ascending
    const uint32_t *dvals;     // [nd] distances that occur, ascending
    const uint32_t *doff;      // [nd+1] offsets of each distance's group in `pairs`
    const uint8_t *cx_pairs;   // [n_cx][2]: the CX gates of the gateset, in order (pauli.rs:360-366)
    uint32_t nd, n_cx;
    uint64_t seed;
    uint32_t difficulty, pauli_difficulty, max_paulis;
    float decay;
    int32_t depth_value;
    uint32_t only_done;
    const uint32_t *list;  // only_done, compacted (compact_done): thread i regenerates env list[i]
    uint32_t *list_count;
    uint32_t n_pairs;      // doff[nd]
    uint32_t gen_grid;     // workgroups of ptile_generate_kernel (0: one per 64 envs); they walk the batch with the grid's stride
    uint32_t tree_grid;    // workgroups of the tree launch (entry i on workgroup i mod tree_grid); `count_out`: where it reports the list's length (host memory)
    uint32_t *count_out;
    uint32_t tree;         // ptile_reset_tree_kernel runs before ptile_generate_kernel and takes the lists pt_tree_takes says it takes
    unsigned long long *tree_kclk;  // qg_vec_set_kernel_clock: the tree launch's slot (the generate launch's is s.kclk)
    // only_done, the finished envs as the bits the step before left (StepArgs::done_mask) instead of a compacted list: every workgroup counts them itself
    // (device_common.hpp done_mask_*); `count_pub` (the list's length word, unused otherwise): where the tree launch leaves the count for the generate launch
    const uint64_t *mask;
    uint32_t mask_words, mask_epoch;
    uint32_t *count_pub;
};
constexpr uint32_t PT_CX_LDS = plan::PAULI_CX_LDS;
using plan::pauli_tree_takes;  // short lists of long scrambles: a workgroup per listed env (ptile_reset_tree_kernel), qgym_plan.hpp

// `pre`: the stream's first `n_pre` draws computed ahead, one per thread, and parked in LDS (ptile_reset_tree_kernel): the label generator's draws
// depend on each other through what they decide, two splitmix64 rounds each (~0.15 us) -- a counter RNG's draw k depends on k alone
// The generator's two streams of one env: the rotation labels draw rng_draw(seed ^ PT_STREAM_LABELS, env, 0, 1, 2, ...), the tableau's gate `it` draws
// rng_draw(seed ^ PT_STREAM_TABLEAU, env, 2 it) (its kind) and (.., 2 it + 1) (which pair / qubit).  Until round 5 the tableau went on where the labels
// had stopped in ONE stream, so its first draw's index was known only after the labels: ptile_reset_tree_kernel's scramble (7.4 us) waited for the
// labels (7.5 us).  The reference draws everything from rand::thread_rng() (pauli.rs:560-575): i.i.d. draws, whatever their order
// (the tests' CPU restatement of the generator draws from the same two streams).
constexpr uint64_t PT_STREAM_LABELS = 0x7061756Cull, PT_STREAM_TABLEAU = 0x7461626Cull;
struct PTStream {
    uint64_t seed, env, k;
    const uint64_t *pre = nullptr;
    uint32_t n_pre = 0;
    __device__ uint64_t at(uint64_t i) const { return (pre && i < n_pre) ? pre[i] : rng_draw(seed, env, i); }  // draw i, the stream stays where it is
    __device__ uint64_t next() { return at(k++); }
    __device__ uint32_t range(uint32_t n) { return (uint32_t)__umul64hi(next(), (uint64_t)n); }
    __device__ float f32() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

// The generator's tables in LDS: the scramble looks a CX pair up for most of its `difficulty` gates -- from a copy in LDS, not from global memory
// behind the draw that picks it (a dependent global load per gate was most of this kernel: 256 gates x ~0.5 us) -- and the label
// generator's loops walk the distance classes and their qubit pairs with dependent loads, ~40 per rotation label.
struct PTGenTables {
    alignas(4) uint8_t cx[2 * PT_CX_LDS];
    uint32_t dvals[32], doff[33];
    alignas(4) uint8_t pairs[2 * 496];  // N (N - 1) / 2 pairs, N <= 32
};
// copied by the whole workgroup, before any lane leaves; the caller makes the copy visible (wave barrier / __syncthreads).  Every load is issued before
// the first LDS store (four copy loops one after the other were four trips to memory one after the other, on every reset's critical path); the tables are
// 4-byte aligned and padded in the handle's blob (ptile_reset_seeded), so the byte tables move as words.
__device__ inline bool pt_gen_tables_load(const PTGenArgs &ga, PTGenTables &t) {
    const bool cx_in_lds = ga.n_cx <= PT_CX_LDS;
    const uint32_t nd = ga.nd < 32u ? ga.nd : 32u;
    const uint32_t n_pairs = ga.n_pairs < 496u ? ga.n_pairs : 496u;  // (= doff[nd]: from the host, not behind a load)
    const uint32_t cx_words = cx_in_lds ? (2u * ga.n_cx + 3u) / 4u : 0u, pair_words = (2u * n_pairs + 3u) / 4u;  // <= 512, <= 248
    const uint32_t *cxw = reinterpret_cast<const uint32_t *>(ga.cx_pairs), *pw = reinterpret_cast<const uint32_t *>(ga.pairs);
    uint32_t *t_cx = reinterpret_cast<uint32_t *>(t.cx), *t_pairs = reinterpret_cast<uint32_t *>(t.pairs);
    constexpr uint32_t CX_PER = 8, PAIR_PER = 4;  // (a 64-thread workgroup: 512 / 64, 248 / 64)
    uint32_t cv[CX_PER], pv[PAIR_PER];
    const uint32_t dv = threadIdx.x < nd ? ga.dvals[threadIdx.x] : 0u, df = threadIdx.x <= nd ? ga.doff[threadIdx.x] : 0u;
#pragma unroll
    for (uint32_t k = 0; k < CX_PER; ++k) {
        const uint32_t i = threadIdx.x + k * blockDim.x;
        cv[k] = i < cx_words ? cxw[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < PAIR_PER; ++k) {
        const uint32_t i = threadIdx.x + k * blockDim.x;
        pv[k] = i < pair_words ? pw[i] : 0u;
    }
    if (threadIdx.x < nd) t.dvals[threadIdx.x] = dv;
    if (threadIdx.x <= nd) t.doff[threadIdx.x] = df;
#pragma unroll
    for (uint32_t k = 0; k < CX_PER; ++k) {
        const uint32_t i = threadIdx.x + k * blockDim.x;
        if (i < cx_words) t_cx[i] = cv[k];
    }
#pragma unroll
    for (uint32_t k = 0; k < PAIR_PER; ++k) {
        const uint32_t i = threadIdx.x + k * blockDim.x;
        if (i < pair_words) t_pairs[i] = pv[k];
    }
    return cx_in_lds;
}

// generate_paulis_with_difficulty (pauli.rs:191-213): the env's rotations, their DAG and phases into `s`; returns the number of labels
template <int NQ, int RM>
__device__ inline uint32_t pt_gen_labels(const PTGenArgs &ga, const PTGenTables &t, PTStream &rng, PTState<NQ, RM> &s, uint32_t N) {
#pragma unroll
    for (int k = 0; k < RM; ++k) s.rx[k] = s.rz[k] = s.rpred[k] = 0;
    s.plo = s.phi = 0;
    uint32_t n_lab = 0, remaining = ga.pauli_difficulty;
    while (remaining > 0 && n_lab < ga.max_paulis) {
        const uint32_t difficulty = remaining;  // get_pauli_under_diff(remaining) (pauli.rs:115-188)
        uint32_t nvd = 0;
        for (uint32_t i = 0; i < ga.nd; ++i) nvd += t.dvals[i] <= difficulty;
        if (nvd == 0) break;
        uint32_t inset = 0, budget = difficulty;
        uint32_t di = rng.range(nvd);
        uint32_t d = t.dvals[di];
        uint32_t pick = t.doff[di] + rng.range(t.doff[di + 1] - t.doff[di]);
        inset |= (1u << t.pairs[2 * pick]) | (1u << t.pairs[2 * pick + 1]);
        budget = budget > d ? budget - d : 0;
        for (;;) {
            uint32_t nv2 = 0;
            for (uint32_t i = 0; i < nvd; ++i) nv2 += t.dvals[i] <= budget;
            if (budget == 0 || nv2 == 0 || (uint32_t)__popc(inset) >= N) break;
            if (rng.f32() <= ga.decay) break;  // continue with probability 1 - num_qubits_decay
            di = rng.range(nv2);
            d = t.dvals[di];
            uint32_t nc = 0;
            for (uint32_t p = t.doff[di]; p < t.doff[di + 1]; ++p) nc += ((inset >> t.pairs[2 * p]) | (inset >> t.pairs[2 * p + 1])) & 1u;
            if (nc == 0) continue;
            uint32_t want = rng.range(nc);
            for (uint32_t p = t.doff[di]; p < t.doff[di + 1]; ++p) {
                if (((inset >> t.pairs[2 * p]) | (inset >> t.pairs[2 * p + 1])) & 1u) {
                    if (want == 0) {
                        inset |= (1u << t.pairs[2 * p]) | (1u << t.pairs[2 * p + 1]);
                        break;
                    }
                    --want;
                }
            }
            budget = budget > d ? budget - d : 0;
        }
        // label: string index q carries the axis; Pauli::from_label reverses, so it is qubit N-1-q
        uint32_t x = 0, z = 0, ys = 0;
        for (uint32_t q = 0; q < N; ++q) {
            if ((inset >> q) & 1u) {
                const uint32_t ax = rng.range(3);  // "XYZ"
                const uint32_t bit = 1u << (N - 1u - q);
                if (ax != 2) x |= bit;
                if (ax != 0) z |= bit;
                ys += ax == 1;
            }
        }
        uint32_t pred = 0;  // PauliDag::new (pauli_dag.rs:35-41): edge to every earlier non-commuting rotation
#pragma unroll
        for (int k = 0; k < RM; ++k)
            if ((uint32_t)k < n_lab) pred |= (uint32_t)((__popc(x & s.rz[k]) + __popc(z & s.rx[k])) & 1) << k;
#pragma unroll
        for (int k = 0; k < RM; ++k) {
            const bool here = (uint32_t)k == n_lab;
            s.rx[k] = here ? x : s.rx[k];
            s.rz[k] = here ? z : s.rz[k];
            s.rpred[k] = here ? pred : s.rpred[k];
        }
        s.plo |= (ys & 1u) << n_lab;  // base_phase = (0 + #Y) mod 4 (pauli.rs:73)
        s.phi |= ((ys >> 1) & 1u) << n_lab;
        n_lab += 1;
        const uint32_t cost = difficulty - budget, dec = cost > 1 ? cost : 1;
        remaining = remaining > dec ? remaining - dec : 0;
    }
    s.alive = (uint32_t)((1ull << n_lab) - 1ull);  // n_lab <= RM <= 32
    s.count = n_lab;
    s.bad = 0;  // the scramble starts from the identity and keeps `bad` current
    s.order.clear();
#pragma unroll
    for (int k = 0; k < RM; ++k)
        if ((uint32_t)k < n_lab) s.order.set((uint32_t)k, (uint32_t)k);
    return n_lab;
}

// pt_gen_labels by the 64 lanes of ONE wave (ptile_reset_tree_kernel; all of them call it, converged): the same draws in the same order and
// the same choices, but every walk over a table -- distance classes within a budget, the pairs of a class that touch the set, the axes of the
// set's qubits -- is a ballot over the lanes instead of a loop of dependent LDS reads (~60 per added pair, 17 of the kernel's 31 us).
template <int NQ, int RM>
__device__ inline uint32_t pt_gen_labels_wave(const PTGenArgs &ga, const PTGenTables &t, PTStream &rng, PTState<NQ, RM> &s, uint32_t N) {
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1);
    const uint32_t nd = ga.nd < 32u ? ga.nd : 32u;
    const uint32_t my_dval = lane < nd ? t.dvals[lane] : 0xFFFFFFFFu;
    auto classes_within = [&](uint32_t limit, uint32_t budget) -> uint32_t {  // #{i < limit : dvals[i] <= budget}
        return (uint32_t)__popcll(__ballot(lane < limit && my_dval <= budget));
    };
#pragma unroll
    for (int k = 0; k < RM; ++k) s.rx[k] = s.rz[k] = s.rpred[k] = 0;
    s.plo = s.phi = 0;
    uint32_t n_lab = 0, remaining = ga.pauli_difficulty;
    while (remaining > 0 && n_lab < ga.max_paulis) {
        const uint32_t difficulty = remaining;  // get_pauli_under_diff(remaining) (pauli.rs:115-188)
        const uint32_t nvd = classes_within(nd, difficulty);
        if (nvd == 0) break;
        uint32_t inset = 0, budget = difficulty;
        uint32_t di = rng.range(nvd);
        uint32_t d = t.dvals[di];
        const uint32_t pick = t.doff[di] + rng.range(t.doff[di + 1] - t.doff[di]);
        inset |= (1u << t.pairs[2 * pick]) | (1u << t.pairs[2 * pick + 1]);
        budget = budget > d ? budget - d : 0;
        for (;;) {
            const uint32_t nv2 = classes_within(nvd, budget);
            if (budget == 0 || nv2 == 0 || (uint32_t)__popc(inset) >= N) break;
            if (rng.f32() <= ga.decay) break;  // continue with probability 1 - num_qubits_decay
            di = rng.range(nv2);
            d = t.dvals[di];
            const uint32_t p0 = t.doff[di], p1 = t.doff[di + 1];
            auto touching = [&](uint32_t base) -> uint64_t {  // the class's pairs base .. base + 63 that touch the set, one per lane
                const uint32_t p = base + lane;
                return __ballot(p < p1 && (((inset >> t.pairs[2 * p]) | (inset >> t.pairs[2 * p + 1])) & 1u));
            };
            uint32_t nc = 0;
            for (uint32_t base = p0; base < p1; base += QG_WAVE) nc += (uint32_t)__popcll(touching(base));
            if (nc == 0) continue;
            uint32_t want = rng.range(nc);
            for (uint32_t base = p0; base < p1; base += QG_WAVE) {  // the want-th touching pair, in ascending order
                const uint64_t m = touching(base);
                const uint32_t here = (uint32_t)__popcll(m);
                if (want < here) {
                    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    const uint64_t sel = __ballot(((m >> lane) & 1ull) && rank == want);
                    const uint32_t p = base + (uint32_t)__ffsll((long long)sel) - 1u;
                    inset |= (1u << t.pairs[2 * p]) | (1u << t.pairs[2 * p + 1]);
                    break;
                }
                want -= here;
            }
            budget = budget > d ? budget - d : 0;
        }
        // label: string index q carries the axis; Pauli::from_label reverses, so it is qubit N-1-q.  The set's qubits draw their axes in
        // ascending order: qubit q takes draw k + (set qubits below q)
        const bool mine = lane < N && ((inset >> lane) & 1u);
        const uint32_t ax = mine ? (uint32_t)__umul64hi(rng.at(rng.k + (uint64_t)__popc(inset & ((1u << lane) - 1u))), 3ull) : 3u;  // "XYZ"
        rng.k += (uint64_t)__popc(inset);
        const uint32_t xq = (uint32_t)__ballot(mine && ax != 2u), zq = (uint32_t)__ballot(mine && ax != 0u);  // bit q
        const uint32_t ys = (uint32_t)__popcll(__ballot(mine && ax == 1u));
        const uint32_t x = __brev(xq) >> (32u - N), z = __brev(zq) >> (32u - N);  // bit N - 1 - q
        uint32_t pred = 0;  // PauliDag::new (pauli_dag.rs:35-41): edge to every earlier non-commuting rotation
#pragma unroll
        for (int k = 0; k < RM; ++k)
            if ((uint32_t)k < n_lab) pred |= (uint32_t)((__popc(x & s.rz[k]) + __popc(z & s.rx[k])) & 1) << k;
#pragma unroll
        for (int k = 0; k < RM; ++k) {
            const bool here = (uint32_t)k == n_lab;
            s.rx[k] = here ? x : s.rx[k];
            s.rz[k] = here ? z : s.rz[k];
            s.rpred[k] = here ? pred : s.rpred[k];
        }
        s.plo |= (ys & 1u) << n_lab;  // base_phase = (0 + #Y) mod 4 (pauli.rs:73)
        s.phi |= ((ys >> 1) & 1u) << n_lab;
        n_lab += 1;
        const uint32_t cost = difficulty - budget, dec = cost > 1 ? cost : 1;
        remaining = remaining > dec ? remaining - dec : 0;
    }
    s.alive = (uint32_t)((1ull << n_lab) - 1ull);  // n_lab <= RM <= 32
    s.count = n_lab;
    s.bad = 0;  // the scramble starts from the identity and keeps `bad` current
    s.order.clear();
#pragma unroll
    for (int k = 0; k < RM; ++k)
        if ((uint32_t)k < n_lab) s.order.set((uint32_t)k, (uint32_t)k);
    return n_lab;
}

// what follows the scramble, on the lane that holds the env's tableau in s.X / s.Z: clean, store, bookkeeping (pauli.rs:576-585)
template <int NQ, int RM>
__device__ inline void pt_gen_finish(const PTGenArgs &ga, PTState<NQ, RM> &s, uint64_t env, uint32_t N) {
    const StepArgs &a = ga.s;
    const uint32_t lane = (uint32_t)(env & (QG_WAVE - 1));
    char *tile = PTLayout<NQ, RM>::tile(a.state, env);
    s.bad = pt_badmask<NQ, RM>(s, N);
    uint32_t n_removed = 0, fault = 0;  // clean initially trivial rotations (pauli.rs:576)
    uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
    for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
    pt_clean<NQ, RM>(s, n_removed, fault, SolLog{nullptr, 0}, rem_pos);
    const bool solved = pt_solved<NQ, RM>(s);
#pragma unroll
    for (int q = 0; q < NQ; ++q) PTLayout<NQ, RM>::store_qubit(tile, lane, q, s.X[q], s.Z[q]);
    pt_store_rotations<NQ, RM>(tile, lane, s, ~0u, ~0u, true);
    pt_store_meta<NQ, RM>(tile, lane, s);
    a.depth[env] = ga.depth_value;  // pauli.rs:578-585
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(ga.depth_value == 0 || solved);
    a.inverted[env] = 0;
    a.error[env] = fault;
    a.sol_len[env * 2] = 0;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, (2 * N + 2));
        for (uint32_t i = 0; i < 2 * N; ++i) lay[i] = -1;
        lay[2 * N] = 0;
        lay[2 * N + 1] = 0;
    }
}

// pt_gen_finish by the 64 lanes of one wave (ptile_reset_tree_kernel; all of them call it, converged, with the same labels in `s`): `rows[k]` = the
// tableau's row of slot k (X[q] = slot q, Z[q] = slot NQ + q, in LDS).  Lane q packs and stores qubit q's group and its {xs, zs} bytes -- one lane doing
// all NQ of them was ~2 300 instructions at 5 cycles each, 4.6 us of the kernel -- the clean is the same work on every lane, lane 0 writes the rest.
template <int NQ, int RM>
__device__ inline void pt_gen_finish_wave(const PTGenArgs &ga, PTState<NQ, RM> &s, uint64_t env, uint32_t N, const uint64_t *rows) {
    using L = PTLayout<NQ, RM>;
    const StepArgs &a = ga.s;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), le = (uint32_t)(env & (QG_WAVE - 1));
    char *tile = L::tile(a.state, env);
    const bool q_lane = lane < (uint32_t)NQ;
    const uint64_t X = q_lane ? rows[lane] : 0ull, Z = q_lane ? rows[(uint32_t)NQ + lane] : 0ull;
    const uint64_t ix = lane < N ? 1ull << lane : 0ull, iz = lane < N ? (1ull << N) << lane : 0ull;
    s.bad = (uint32_t)__ballot(q_lane && (X != ix || Z != iz));  // pt_badmask
    uint32_t n_removed = 0, fault = 0;  // clean initially trivial rotations (pauli.rs:576)
    uint64_t rem_pos[(RM + 7) / 8];
#pragma unroll
    for (int i = 0; i < (RM + 7) / 8; ++i) rem_pos[i] = 0;
    pt_clean<NQ, RM>(s, n_removed, fault, SolLog{nullptr, 0}, rem_pos);
    const bool solved = pt_solved<NQ, RM>(s);
    if (q_lane) L::store_qubit(tile, le, lane, X, Z);
    if constexpr (L::COMPACT) {
        if (q_lane) pt_store_xz<NQ, RM>(tile, le, s, lane);
        if (lane == 0) pt_store_rotmeta<NQ, RM>(tile, le, s, true);
    } else {
        if (lane == 0) pt_store_rotations<NQ, RM>(tile, le, s, ~0u, ~0u, true);
    }
    if (lane != 0) return;
    pt_store_meta<NQ, RM>(tile, le, s);
    a.depth[env] = ga.depth_value;  // pauli.rs:578-585
    a.success[env] = (uint8_t)solved;
    a.reward[env] = solved ? 1.0f : 0.0f;
    a.done[env] = (uint8_t)(ga.depth_value == 0 || solved);
    a.inverted[env] = 0;
    a.error[env] = fault;
    a.sol_len[env * 2] = 0;
    a.sol_len[env * 2 + 1] = 0;
    if (a.layers) {
        const LayerRec lay = layer_rec(a.layers, env, (2 * N + 2));
        for (uint32_t i = 0; i < 2 * N; ++i) lay[i] = -1;
        lay[2 * N] = 0;
        lay[2 * N + 1] = 0;
    }
}

// The 64 envs of "workgroup" vblock (one wave).  Returns false -- on every lane alike -- when neither this vblock nor any later one has work (a list / mask of
// finished envs that ends before it, or that ptile_reset_tree_kernel has taken): ptile_generate_kernel's workgroups walk the vblocks with the grid's stride, and
// the grid behind a tree launch is a few workgroups when the handle's lists have been trees' lists (PTGenArgs::gen_grid).
template <int NQ, int RM>
__device__ __forceinline__ bool ptile_generate_body(const PTGenArgs &ga, uint32_t vblock) {
    const StepArgs &a = ga.s;
    // the tableau scramble runs on LDS-resident rows ([row][lane], conflict-free for any per-lane row): a
    // random CX / H / S is one or two row operations instead of a select sweep over 2N 64-bit registers
    __shared__ uint64_t lds_tab[2 * NQ][QG_WAVE];
    __shared__ PTGenTables tb;
    const uint64_t tid = (uint64_t)vblock * QG_WAVE + threadIdx.x;
    // the list's only reader; compact_done re-initialises the length before every use, so nobody has to zero it here (no reader tickets)
    __shared__ uint32_t mask_part[64 + 1 + 5];
    uint32_t count = ga.list ? ga.list_count[0] : 0u;
    if (ga.mask) {  // (the workgroup is one wave)
        // after a tree launch its count is in count_pub: most calls leave here; else (no tree launch, or a list too long for it) count the mask
        const bool published = ga.tree != 0;
        count = published ? *ga.count_pub : 0u;
        if (published && (pauli_tree_takes(count, ga.difficulty, a.B, ga.n_cx) || (tid & ~(uint64_t)(QG_WAVE - 1)) >= count)) return false;
        DoneMaskShare share;
        done_mask_load<64>(ga.mask, a.B, ga.mask_words, share);
        count = *done_mask_hint(ga.mask, a.B) != ga.mask_epoch ? 0u : done_mask_scan<64>(share, mask_part);
        if ((tid & ~(uint64_t)(QG_WAVE - 1)) >= count) return false;
    } else if (ga.list && ((ga.tree && pauli_tree_takes(count, ga.difficulty, a.B, ga.n_cx)) || (tid & ~(uint64_t)(QG_WAVE - 1)) >= count)) {
        // (before the tables are brought in: most calls with a list leave here -- ptile_reset_tree_kernel has taken it, or the wave lies past it)
        return false;
    }
    const bool cx_in_lds = pt_gen_tables_load(ga, tb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    uint64_t env = tid;
    if (ga.mask) {
        if (tid >= count) return true;
        env = done_mask_nth<64>(ga.mask, ga.mask_words, mask_part, (uint32_t)tid);
    } else if (ga.list) {
        if (tid >= count) return true;
        env = ga.list[tid];
    } else {
        if (env >= a.B) return true;
        if (ga.only_done && !a.done[env]) return true;
    }
    const uint32_t L = threadIdx.x & (QG_WAVE - 1);
    const uint32_t N = a.N;
    const uint64_t base_seed = ga.seed + QG_CLOCK_SEED_STRIDE * clock_of(a.clock);
    PTStream rng{base_seed ^ PT_STREAM_LABELS, a.env_base + env, 0};
    PTState<NQ, RM> s;
    (void)pt_gen_labels<NQ, RM>(ga, tb, rng, s, N);

    // random_clifford_tableau (pauli.rs:220-271): H / S / CX row operations on the identity
    // (row q = X[q], row N + q = Z[q]; LDS slot NQ + q holds Z[q])
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        lds_tab[j][L] = (uint32_t)j < N ? 1ull << j : 0ull;
        lds_tab[NQ + j][L] = (uint32_t)j < N ? (1ull << N) << j : 0ull;
    }
    if (ga.difficulty != 0 && ga.n_cx != 0) {
        // gate `it` takes draws 2 it (kind) and 2 it + 1 (which pair / qubit) of the env's tableau stream: the counter RNG makes every draw a
        // function of its index, so four gates' draws are computed ahead of the dependent chain of LDS row operations
        const uint64_t tseed = base_seed ^ PT_STREAM_TABLEAU;
        for (uint32_t it0 = 0; it0 < ga.difficulty; it0 += 4) {
            uint64_t d1[4], d2[4];
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                d1[j] = rng_draw(tseed, rng.env, 2ull * (it0 + j));
                d2[j] = rng_draw(tseed, rng.env, 2ull * (it0 + j) + 1ull);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                if (it0 + j >= ga.difficulty) break;
                const float r = (float)(d1[j] >> 40) * (1.0f / 16777216.0f);
                if (r > 0.3f) {  // CX: row q1 ^= row q0 ; row n+q0 ^= row n+q1
                    const uint32_t k = (uint32_t)__umul64hi(d2[j], (uint64_t)ga.n_cx);
                    const uint32_t q0 = cx_in_lds ? tb.cx[2 * k] : ga.cx_pairs[2 * k], q1 = cx_in_lds ? tb.cx[2 * k + 1] : ga.cx_pairs[2 * k + 1];
                    const uint64_t x0 = lds_tab[q0][L], z1 = lds_tab[NQ + q1][L];
                    lds_tab[q1][L] ^= x0;       // a CX(q, q) entry xors the rows into themselves: both become zero
                    lds_tab[NQ + q0][L] ^= z1;
                } else if (r > 0.15f) {  // H: swap rows q, n+q
                    const uint32_t q = (uint32_t)__umul64hi(d2[j], (uint64_t)N);
                    const uint64_t x = lds_tab[q][L], z = lds_tab[NQ + q][L];
                    lds_tab[q][L] = z;
                    lds_tab[NQ + q][L] = x;
                } else {  // S: row n+q ^= row q
                    const uint32_t q = (uint32_t)__umul64hi(d2[j], (uint64_t)N);
                    lds_tab[NQ + q][L] ^= lds_tab[q][L];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
        s.X[j] = lds_tab[j][L];
        s.Z[j] = lds_tab[NQ + j][L];
    }
    pt_gen_finish<NQ, RM>(ga, s, env, N);
    return true;
}
template <int NQ, int RM>
__global__ __launch_bounds__(64) void ptile_generate_kernel(PTGenArgs ga) {
    KernelClock kclk(ga.s.kclk, ga.s.kclk_waves);  // device_common.hpp
    // (the way out most launches behind a tree launch take, ahead of everything the body sets up: the trees left the mask's count, and it says they took the list)
    if (ga.mask && ga.tree) {
        const uint32_t count = *ga.count_pub;
        if (!count || pauli_tree_takes(count, ga.difficulty, ga.s.B, ga.n_cx)) return;
    }
    const uint32_t total = (uint32_t)((ga.s.B + QG_WAVE - 1) / QG_WAVE);
    for (uint32_t vblock = blockIdx.x; vblock < total; vblock += gridDim.x) {
        if (!ptile_generate_body<NQ, RM>(ga, vblock)) break;
        __builtin_amdgcn_wave_barrier();  // (the workgroup is one wave: its LDS tables and rows are free again)
    }
}

// qg_vec_reset_done with a short list of long scrambles (pt_tree_takes): a workgroup per listed env.  A lane of the kernel above spends most
// of its time on the scramble's 2 x `difficulty` draws (four 64-bit multiplies each) and on the dependent chain of row operations behind
// them; here every thread of waves 0 .. 3 draws ONE gate and the chain is cut in four and multiplied back (scramble_tree64_ops, device_common.hpp; the
// tableau's 2 NQ <= 64 rows are the slots, X[q] = slot q, Z[q] = slot NQ + q) WHILE the fifth wave generates the labels (pt_gen_labels_wave: the two
// draw from separate streams), then takes the rows from LDS and finishes the env on its 64 lanes (pt_gen_finish_wave).  What is left is one lone wave's
// instruction stream at ~5 cycles an instruction: the labels' ~3 700.
constexpr uint32_t PT_TREE_THREADS = QG_TREE_THREADS + QG_WAVE;  // four waves of scramble + the labels' wave
template <int NQ, int RM>
__global__ __launch_bounds__(PT_TREE_THREADS) void ptile_reset_tree_kernel(PTGenArgs ga) {
    KernelClock kclk(ga.tree_kclk, ga.s.kclk_waves);  // device_common.hpp
    const StepArgs &a = ga.s;
    constexpr int R = 2 * NQ;
    __shared__ uint64_t prod[4][64];
    __shared__ RowopMasks64 tree_gates[4][QG_WAVE];
    __shared__ PTGenTables tb;
    __shared__ uint64_t rows_out[64];
    __shared__ uint64_t pre_draws[QG_TREE_THREADS];
    uint32_t count;
    __shared__ uint32_t mask_part[PT_TREE_THREADS + 2 + PT_TREE_THREADS / 64];
    DoneMaskShare share;
    // the mask's counts (or the list's length) and the generator's tables in flight together: `opaque_zero` keeps the uniform loads vector loads, which
    // are waited for where their values are used and not where they are issued (kernels_qm.hip qm_init_block)
    uint32_t opaque_zero, first_v = 0;
    asm("v_mov_b32 %0, 0" : "=v"(opaque_zero));
    if (ga.mask) {  // the step before left its finishers as bits: count them (a hint word with another number: nobody finished)
        first_v = done_mask_hint(ga.mask, a.B)[opaque_zero];
        done_mask_load<PT_TREE_THREADS>(ga.mask, a.B, ga.mask_words, share);
    } else {
        first_v = ga.list_count[opaque_zero];
    }
    asm volatile("" ::: "memory");
    (void)pt_gen_tables_load(ga, tb);
    if (ga.mask) {
        count = (uint32_t)__builtin_amdgcn_readfirstlane((int)first_v) != ga.mask_epoch ? 0u : done_mask_scan<PT_TREE_THREADS>(share, mask_part);
        if (blockIdx.x == 0 && threadIdx.x == 0) *ga.count_pub = count;  // (for the generate launch behind this one)
    } else {
        count = (uint32_t)__builtin_amdgcn_readfirstlane((int)first_v);
    }
    if (ga.count_out && blockIdx.x == 0 && threadIdx.x == 0) *ga.count_out = count;  // (host memory: sizes the next launches' tree grid)
    if (threadIdx.x == 0) phase_stamp(ga.tree_kclk, a.kclk_waves, 0);  // the count is known
    if (!pauli_tree_takes(count, ga.difficulty, a.B, ga.n_cx)) return;  // (uniform per workgroup)
    const uint32_t N = a.N, lane = threadIdx.x & (QG_WAVE - 1), wave = threadIdx.x >> 6;
    __shared__ uint32_t prod_ready[4];  // scramble_tree64_ops: the products' levels hand over through these (the labels' wave is not part of them)
    if (threadIdx.x < 4u) prod_ready[threadIdx.x] = 0u;  // (visible after the barrier below)
    uint32_t prod_seq = 0;
    // entry blockIdx.x of the list, then + gridDim.x, ...: the launch's grid follows the list lengths the handle has seen, a longer list is walked in rounds
    for (uint32_t item = blockIdx.x; item < count; item += gridDim.x) {
    if (item != blockIdx.x) __syncthreads();  // (the previous round's LDS has been read)
    const uint64_t env = !ga.mask ? ga.list[item]
                       : item == blockIdx.x ? done_mask_find<PT_TREE_THREADS>(ga.mask, ga.mask_words, share, mask_part, item)
                                            : done_mask_nth<PT_TREE_THREADS>(ga.mask, ga.mask_words, mask_part, item);
    const uint64_t base_seed = ga.seed + QG_CLOCK_SEED_STRIDE * clock_of(a.clock), renv = a.env_base + env;
    PTStream rng{base_seed ^ PT_STREAM_LABELS, renv, 0};
    if (threadIdx.x < QG_TREE_THREADS) pre_draws[threadIdx.x] = rng_draw(rng.seed, rng.env, threadIdx.x);  // the label generator's draws, one per thread (it rarely needs more)
    __syncthreads();  // (the tables, the draws -- and, with a list, the entry -- are in LDS)
    rng.pre = pre_draws;
    rng.n_pre = QG_TREE_THREADS;
    if ((threadIdx.x & 63u) == 0 && (wave == 0u || wave == 4u)) phase_stamp(ga.tree_kclk, a.kclk_waves, wave == 0u ? 1u : 2u);  // the env and the draws are in LDS
    if (wave == 4u) {  // the labels, beside the scramble
        PTState<NQ, RM> s;
        (void)pt_gen_labels_wave<NQ, RM>(ga, tb, rng, s, N);
        phase_stamp(ga.tree_kclk, a.kclk_waves, 3);  // the labels
        __syncthreads();  // rows_out is written
        phase_stamp(ga.tree_kclk, a.kclk_waves, 5);  // the rows have arrived
        pt_gen_finish_wave<NQ, RM>(ga, s, env, N, rows_out);
        phase_stamp(ga.tree_kclk, a.kclk_waves, 6);  // everything is stored
        continue;
    }
    const uint32_t n_cx = ga.n_cx;
    const uint8_t *cx = tb.cx;
    const uint64_t tseed = base_seed ^ PT_STREAM_TABLEAU;
    uint64_t row = 0;
    const bool finisher = scramble_tree64_ops<R>(
        ga.difficulty, row, prod, tree_gates,
        [N](uint32_t k) -> uint64_t { const uint32_t j = k < (uint32_t)NQ ? k : k - (uint32_t)NQ; return j < N ? (k < (uint32_t)NQ ? 1ull << j : (1ull << N) << j) : 0ull; },
        [=](uint32_t it) -> uint32_t {  // random_clifford_tableau's gate `it` as two row operations (pauli.rs:220-271)
            const uint64_t d1 = rng_draw(tseed, renv, 2ull * it), d2 = rng_draw(tseed, renv, 2ull * it + 1ull);
            const float r = (float)(d1 >> 40) * (1.0f / 16777216.0f);
            if (r > 0.3f) {  // CX: row q1 ^= row q0 ; row n+q0 ^= row n+q1 (q0 == q1: a row xor-ed into itself is zero, in this form too)
                const uint32_t k = (uint32_t)__umul64hi(d2, (uint64_t)n_cx);
                const uint32_t q0 = cx[2 * k], q1 = cx[2 * k + 1];
                return make_op(OP_XOR, q1, q0) | (make_op(OP_XOR, (uint32_t)NQ + q0, (uint32_t)NQ + q1) << 14);
            }
            const uint32_t q = (uint32_t)__umul64hi(d2, (uint64_t)N);
            if (r > 0.15f) return make_op(OP_SWAP, q, (uint32_t)NQ + q);  // H: swap rows q, n+q
            return make_op(OP_XOR, (uint32_t)NQ + q, q);                    // S: row n+q ^= row q
        }, prod_ready, prod_seq);
    prod_seq += 2u;  // (two levels)
    if (finisher) rows_out[lane] = row;  // wave 0: lane s holds the row of slot s
    if (finisher) phase_stamp(ga.tree_kclk, a.kclk_waves, 4);  // the scramble
    __syncthreads();
    }
}

static inline unsigned grid_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

// ---- host hooks ----------------------------------------------------------------------------------

int ptile_alloc(qg_vec *v) {
    std::vector<uint64_t> prog(std::max<size_t>(v->gates.size(), 1));
    for (size_t i = 0; i < v->gates.size(); ++i) prog[i] = ptile_program(v->gates[i]);
    HIP_TRY(hipMalloc(&v->d_prog, sizeof(uint64_t) * prog.size()));
    HIP_TRY(hipMemcpy(v->d_prog, prog.data(), sizeof(uint64_t) * prog.size(), hipMemcpyHostToDevice));
    return QG_OK;
}

static void fill_pt_args(const qg_vec *v, const StepArgs &a, PTArgs &pa) {
    pa.s = a;
    pa.prog = reinterpret_cast<const uint64_t *>(v->d_prog);
    pa.act_perms = v->d_act_perms;
    pa.perm_idx = v->perm_idx;
    pa.n_perms = v->n_perms;
    pa.do_clean = 0;
    pa.depth_value = 0;
}

template <int NQ, int RM>
static hipError_t pt_launch_step(const PTArgs &pa, hipStream_t s) {
    const dim3 grid(grid_for(pa.s.B, 256)), block(256);
    const bool feat = pa.s.flags & (F_TRACK | F_LAYERS);
    switch (plan::pauli_step_kernel_of(pa.s.flags, pa.s.T, PTLayout<NQ, RM>::COMPACT, pa.n_perms != 0)) {  // qgym_plan.hpp
    case plan::SK_PTILE_STEP1C:
        if constexpr (PTLayout<NQ, RM>::COMPACT) {
            const bool list = (pa.s.flags & F_DONE_LIST) && pa.s.done_mask;
            if (feat && list) hipLaunchKernelGGL((ptile_step1c_kernel<NQ, RM, true, true>), grid, block, 0, s, pa);
            else if (feat) hipLaunchKernelGGL((ptile_step1c_kernel<NQ, RM, true>), grid, block, 0, s, pa);
            else if (list) hipLaunchKernelGGL((ptile_step1c_kernel<NQ, RM, false, true>), grid, block, 0, s, pa);
            else hipLaunchKernelGGL((ptile_step1c_kernel<NQ, RM, false>), grid, block, 0, s, pa);
        }
        break;
    case plan::SK_PTILE_STEP1:
        if constexpr (!PTLayout<NQ, RM>::COMPACT) {
            if (feat) hipLaunchKernelGGL((ptile_step1_kernel<NQ, RM, true>), grid, block, 0, s, pa);
            else hipLaunchKernelGGL((ptile_step1_kernel<NQ, RM, false>), grid, block, 0, s, pa);
        }
        break;
    case plan::SK_PTILE_FUSED1C:
        if constexpr (PTLayout<NQ, RM>::COMPACT)
            hipLaunchKernelGGL((ptile_fused1c_kernel<NQ, RM>), dim3(grid_for(pa.s.B, QG_WAVE)), dim3(QG_WAVE), 0, s, pa);
        break;
    default:
        if (feat) hipLaunchKernelGGL((ptile_step_kernel<NQ, RM, true>), grid, block, 0, s, pa);
        else hipLaunchKernelGGL((ptile_step_kernel<NQ, RM, false>), grid, block, 0, s, pa);
        break;
    }
    return hipGetLastError();
}
template <int NQ, int RM>
static hipError_t pt_launch_init(const PTArgs &pa, hipStream_t s) {
    hipLaunchKernelGGL((ptile_init_kernel<NQ, RM>), dim3(grid_for(pa.s.B, 256)), dim3(256), 0, s, pa);
    return hipGetLastError();
}

#define PT_DISPATCH(FN)                                  \
    switch (v->pt_nq * 100 + v->pt_rm) {                 \
    case 408: return FN<4, 8>(pa, s);                    \
    case 416: return FN<4, 16>(pa, s);                   \
    case 808: return FN<8, 8>(pa, s);                    \
    case 816: return FN<8, 16>(pa, s);                   \
    case 1208: return FN<12, 8>(pa, s);                  \
    case 1216: return FN<12, 16>(pa, s);                 \
    case 1608: return FN<16, 8>(pa, s);                  \
    case 1616: return FN<16, 16>(pa, s);                 \
    case 2008: return FN<20, 8>(pa, s);                  \
    case 2016: return FN<20, 16>(pa, s);                 \
    case 2408: return FN<24, 8>(pa, s);                  \
    case 2416: return FN<24, 16>(pa, s);                 \
    case 2808: return FN<28, 8>(pa, s);                  \
    case 2816: return FN<28, 16>(pa, s);                 \
    case 3208: return FN<32, 8>(pa, s);                  \
    case 3216: return FN<32, 16>(pa, s);                 \
    case 432: return FN<4, 32>(pa, s);                   \
    case 832: return FN<8, 32>(pa, s);                   \
    case 1232: return FN<12, 32>(pa, s);                 \
    case 1632: return FN<16, 32>(pa, s);                 \
    case 2032: return FN<20, 32>(pa, s);                 \
    case 2432: return FN<24, 32>(pa, s);                 \
    case 2832: return FN<28, 32>(pa, s);                 \
    case 3232: return FN<32, 32>(pa, s);                 \
    }                                                    \
    return hipErrorInvalidValue;

hipError_t ptile_step(const qg_vec *v, const StepArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    PTArgs pa;
    fill_pt_args(v, a, pa);
    PT_DISPATCH(pt_launch_step)
}

static hipError_t ptile_init(const qg_vec *v, const PTArgs &pa, hipStream_t s) { PT_DISPATCH(pt_launch_init) }

static void fill_obs(const qg_vec *v, const ObsArgs &a, PTObsArgs &pa) {
    pa.o = a;
    pa.nq = v->pt_nq;
    pa.rm = v->pt_rm;
    pa.max_rot = (uint32_t)v->cfg.max_rotations;
    pa.qubit_perms = v->d_qubit_perms;
    pa.perm_idx = v->perm_idx;
    pa.perm_in = v->perm_in;
    pa.n_perms = v->n_perms;
    pa.draw = v->perm_draw ? 1u : 0u;
    pa.seed = v->coin_seed;
    pa.counter = v->observe_counter;
    pa.clock = v->clock_dev;
    pa.env_base = v->env_base;
}

// dense observation in `out_dtype`: row words into the handle's scratch, then the write-bound expansion
static hipError_t ptile_dense_via_words(qg_vec *v, const ObsArgs &a, void *out, int out_dtype, hipStream_t s) {
    const uint64_t n_rows = a.B * 2ull * a.N;
    if (ensure_scratch_public(v, n_rows * sizeof(uint64_t)) != QG_OK) return hipErrorOutOfMemory;
    PTObsArgs pa;
    fill_obs(v, a, pa);
    pa.o.format = QG_FMT_U8;
    hipLaunchKernelGGL(ptile_rowwords_kernel, dim3(grid_for(n_rows / 2, 256)), dim3(256), 0, s, pa, reinterpret_cast<uint64_t *>(v->scratch));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return expand_rows(v->scratch, 8, n_rows, a.obs_cols, out, out_dtype, s);
}

hipError_t ptile_export(const qg_vec *v, const ObsArgs &a, hipStream_t s) {
    if (!a.B) return hipSuccess;
    if (a.format == QG_FMT_U8 && a.obs_cols <= 64u && a.out_stride == (uint64_t)a.obs_rows * a.obs_cols && (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0)
        return ptile_dense_via_words(const_cast<qg_vec *>(v), a, a.out, QG_DT_I8, s);  // the scratch buffer is a cache, not state
    if (a.format == QG_FMT_I64 && a.B >= QG_STREAM_MIN_ENVS && a.obs_cols == 2 * a.N && a.out_stride == (uint64_t)a.obs_cols * a.obs_cols &&
        (reinterpret_cast<uintptr_t>(a.out) & 15u) == 0 && a.out != v->scratch) {
        // get_state in the trait's Vec<i64> format (the tableau, pauli.rs:517-552): row words + the streaming expansion (qgym_api.cpp does the
        // same for the other bit-matrix layouts); 839 MB for 20 qubits x 65 536 envs, which the row-per-thread kernel below writes at 1.8 TB/s
        qg_vec *mv = const_cast<qg_vec *>(v);  // the scratch buffer is a cache, not state
        const uint64_t n_rows = a.B * 2ull * a.N;
        if (ensure_scratch_public(mv, n_rows * sizeof(uint64_t)) != QG_OK) return hipErrorOutOfMemory;
        PTObsArgs pa;
        fill_obs(v, a, pa);
        hipLaunchKernelGGL(ptile_rowwords_kernel, dim3(grid_for(n_rows / 2, 256)), dim3(256), 0, s, pa, reinterpret_cast<uint64_t *>(mv->scratch));
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return expand_rows_i64(mv->scratch, 8, n_rows, a.obs_cols, reinterpret_cast<int64_t *>(a.out), s);
    }
    PTObsArgs pa;
    fill_obs(v, a, pa);
    hipLaunchKernelGGL(ptile_export_kernel, dim3(grid_for(a.B * 2ull * a.N, 256)), dim3(256), 0, s, pa);
    return hipGetLastError();
}

// PauliEnv::observe as one 64-bit word per observation row (bit c = column c: the tableau's 2N columns, then the active rotations'
// columns; pauli.rs:411-485) -- the packed form of the [2N, 2N + max_rotations] observation, 8 bytes per row instead of cols
hipError_t ptile_observe_words(qg_vec *v, void *out_dev, hipStream_t s) {
    ObsArgs a;
    memset(&a, 0, sizeof a);
    a.state = v->state;
    a.B = v->B;
    a.N = v->N;
    a.D = 2 * v->N;
    a.obs_rows = 2 * v->N;
    a.obs_cols = 2 * v->N + (uint32_t)std::max(v->cfg.max_rotations, 1);
    a.format = QG_FMT_U8;
    if (!a.B) return hipSuccess;
    PTObsArgs pa;
    fill_obs(v, a, pa);
    hipLaunchKernelGGL(ptile_rowwords_kernel, dim3(grid_for(a.B * 2ull * a.N / 2, 256)), dim3(256), 0, s, pa, reinterpret_cast<uint64_t *>(out_dev));
    return hipGetLastError();
}

hipError_t ptile_observe_typed(qg_vec *v, void *out_dev, int out_dtype, hipStream_t s) {
    ObsArgs a;
    memset(&a, 0, sizeof a);
    a.state = v->state;
    a.B = v->B;
    a.N = v->N;
    a.D = 2 * v->N;
    a.obs_rows = 2 * v->N;
    a.obs_cols = 2 * v->N + (uint32_t)std::max(v->cfg.max_rotations, 1);  // qg_vec_get_info
    a.format = QG_FMT_U8;
    if (!a.B) return hipSuccess;
    return ptile_dense_via_words(v, a, out_dev, out_dtype, s);
}


template <int NQ, int RM>
static hipError_t pt_launch_generate(const PTGenArgs &pa, hipStream_t s) {
    if (pa.tree) {  // up to B / 32 listed envs get a workgroup each; the workgroups past the list leave at once
        const uint64_t blocks = pa.tree_grid;
        if (blocks) hipLaunchKernelGGL((ptile_reset_tree_kernel<NQ, RM>), dim3((unsigned)blocks), dim3(PT_TREE_THREADS), 0, s, pa);
    }
    const unsigned all = grid_for(pa.s.B, 64);
    hipLaunchKernelGGL((ptile_generate_kernel<NQ, RM>), dim3(pa.gen_grid && pa.gen_grid < all ? pa.gen_grid : all), dim3(64), 0, s, pa);
    return hipGetLastError();
}
static hipError_t ptile_generate(const qg_vec *v, const PTGenArgs &pa, hipStream_t s) { PT_DISPATCH(pt_launch_generate) }

// PauliEnv::reset for the whole batch (or, with only_done, for the finished episodes) on the device
int ptile_reset_seeded(qg_vec *v, uint64_t seed, bool only_done, hipStream_t s, bool from_mask) {
    const uint32_t n = v->N;
    if (!v->d_gen_tables) {  // coupling-graph tables, once per handle (PauliEnv::new, pauli.rs:360-370)
        std::vector<uint8_t> cx;
        std::vector<std::vector<uint32_t>> adj(n);
        for (const qg_gate &g : v->gates)
            if (g.kind == QG_CX) {
                cx.push_back((uint8_t)g.q0);
                cx.push_back((uint8_t)g.q1);
                auto add = [&](uint32_t a, uint32_t b) { if (std::find(adj[a].begin(), adj[a].end(), b) == adj[a].end()) adj[a].push_back(b); };
                add((uint32_t)g.q0, (uint32_t)g.q1);
                add((uint32_t)g.q1, (uint32_t)g.q0);
            }
        std::vector<std::vector<int>> dist(n, std::vector<int>(n, -1));  // compute_graph_distances (pauli.rs:56-91)
        for (uint32_t st = 0; st < n; ++st) {
            std::vector<uint32_t> q{st};
            dist[st][st] = 0;
            for (size_t h = 0; h < q.size(); ++h)
                for (uint32_t w : adj[q[h]])
                    if (dist[st][w] < 0) { dist[st][w] = dist[st][q[h]] + 1; q.push_back(w); }
        }
        std::vector<uint8_t> pairs;  // build_dist_pairs (pauli.rs:95-111)
        std::vector<uint32_t> dvals, doff;
        for (int d = 1; d < (int)n; ++d) {
            const size_t before = pairs.size();
            for (uint32_t a = 0; a < n; ++a)
                for (uint32_t b = a + 1; b < n; ++b)
                    if (dist[a][b] == d) { pairs.push_back((uint8_t)a); pairs.push_back((uint8_t)b); }
            if (pairs.size() != before) { dvals.push_back((uint32_t)d); doff.push_back((uint32_t)(before / 2)); }
        }
        doff.push_back((uint32_t)(pairs.size() / 2));
        // one allocation: [dvals | doff | pairs | cx]
        const size_t o_dvals = 0, o_doff = o_dvals + 4 * std::max<size_t>(dvals.size(), 1), o_pairs = o_doff + 4 * doff.size();
        const size_t o_cx = o_pairs + ((pairs.size() + 3) & ~size_t(3)) + 4, total = o_cx + cx.size() + 4;
        std::vector<uint8_t> blob(total, 0);
        if (!dvals.empty()) memcpy(blob.data() + o_dvals, dvals.data(), 4 * dvals.size());
        memcpy(blob.data() + o_doff, doff.data(), 4 * doff.size());
        if (!pairs.empty()) memcpy(blob.data() + o_pairs, pairs.data(), pairs.size());
        if (!cx.empty()) memcpy(blob.data() + o_cx, cx.data(), cx.size());
        HIP_TRY(hipMalloc(&v->d_gen_tables, total));
        HIP_TRY(hipMemcpy(v->d_gen_tables, blob.data(), total, hipMemcpyHostToDevice));
        v->gen_nd = (uint32_t)dvals.size();
        v->gen_ncx = (uint32_t)(cx.size() / 2);
        v->gen_npairs = (uint32_t)(pairs.size() / 2);
        v->gen_off[0] = (uint32_t)o_dvals; v->gen_off[1] = (uint32_t)o_doff; v->gen_off[2] = (uint32_t)o_pairs; v->gen_off[3] = (uint32_t)o_cx;
    }
    PTGenArgs ga;
    memset(&ga, 0, sizeof ga);
    StepArgs &a = ga.s;
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
    a.N = n;
    const uint8_t *base = reinterpret_cast<const uint8_t *>(v->d_gen_tables);
    ga.dvals = reinterpret_cast<const uint32_t *>(base + v->gen_off[0]);
    ga.doff = reinterpret_cast<const uint32_t *>(base + v->gen_off[1]);
    ga.pairs = base + v->gen_off[2];
    ga.cx_pairs = base + v->gen_off[3];
    ga.nd = v->gen_nd;
    ga.n_cx = v->gen_ncx;
    ga.n_pairs = v->gen_npairs;
    ga.seed = seed;
    a.clock = v->clock_dev;
    a.env_base = v->env_base;
    ga.difficulty = (uint32_t)v->difficulty;
    ga.pauli_difficulty = (uint32_t)(v->difficulty / std::max(v->cfg.pauli_diff_scale, 1));  // pauli.rs:557,392
    ga.max_paulis = v->rmax_generate;
    ga.decay = v->cfg.num_qubits_decay;
    const int64_t d = (int64_t)v->cfg.depth_slope * v->difficulty;  // pauli.rs:578
    ga.depth_value = (int32_t)std::min<int64_t>(d, v->cfg.max_depth);
    ga.only_done = only_done ? 1u : 0u;
    if (only_done && from_mask && v->done_mask[0]) {  // the step before left its finishers as bits: no compaction launch
        ga.mask = v->done_mask[v->mask_cur];
        ga.mask_words = (uint32_t)(4 * ((v->B + 255) / 256));
        ga.mask_epoch = v->mask_epoch[v->mask_cur];
        ga.count_pub = v->mask_count;
        ga.tree = (ga.difficulty >= plan::TREE_MIN_DRAWS && v->B / 32u >= 1u && pauli_tree_takes(1u, ga.difficulty, v->B, ga.n_cx)) ? 1u : 0u;
    } else if (only_done && v->done_list && v->B > QG_COMPACT_MIN_ENVS) {  // pack the finished envs: full waves instead of one live lane in every wave
        HIP_TRY(compact_done(v->done, v->B, v->done_list, v->done_list + v->B, s));
        ga.list = v->done_list;
        ga.list_count = v->done_list + v->B;
        ga.tree = (ga.difficulty >= plan::TREE_MIN_DRAWS && v->B / 32u >= 1u && pauli_tree_takes(1u, ga.difficulty, v->B, ga.n_cx)) ? 1u : 0u;
    }
    if (ga.tree) {
        ga.tree_grid = reset_tree_grid_public(v, (uint32_t)(v->B / 32u));
        ga.gen_grid = reset_second_grid_public(v, [](uint32_t count, const qg_vec *h) -> bool {
            return pauli_tree_takes(count, (uint32_t)h->difficulty, h->B, h->gen_ncx);
        });
        ga.count_out = v->count_seen;
        ga.tree_kclk = kernel_clock_slot_public(v);
    }
    a.kclk = kernel_clock_slot_public(v);
    a.kclk_waves = v->kclk_waves;
    HIP_TRY(ptile_generate(v, ga, s));
    return QG_OK;
}

// scatter the per-env records into the tiled layout, upload, run the init kernel
int ptile_upload(qg_vec *v, const HostNet &h, bool do_clean, int32_t depth_value, hipStream_t s) {
    const uint32_t N = v->N, NQ = v->pt_nq, RM = v->pt_rm;
    const bool compact = NQ <= 24 && RM == 8;  // PTLayout::COMPACT
    const size_t QB = compact ? 768 : 1024, RB = compact ? 512 : 1024, tile_bytes = NQ * QB + RM * RB + (RM > 16 ? 3072 : 1024);
    std::vector<uint8_t> img(v->state_bytes, 0);
    for (uint64_t e = 0; e < v->B; ++e) {
        uint8_t *tile = img.data() + (e >> 6) * tile_bytes;
        const uint32_t lane = (uint32_t)(e & 63);
        for (uint32_t q = 0; q < N; ++q) {
            const uint64_t x = h.tab[(e * N + q) * 2], z = h.tab[(e * N + q) * 2 + 1];
            if (compact) {
                uint32_t g[3] = {(uint32_t)x, ((uint32_t)(x >> 32) & 0xFFFFu) | ((uint32_t)z << 16), (uint32_t)(z >> 16)};
                memcpy(tile + q * QB + lane * 12, g, 12);
            } else {
                uint32_t g[4] = {(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)z, (uint32_t)(z >> 32)};
                memcpy(tile + q * QB + lane * 16, g, 16);
            }
        }
        if (compact) {  // transposed rotation region (see the file header)
            uint8_t *rb = tile + NQ * QB;
            auto byte_at = [&](uint32_t b) -> uint8_t & { return rb[(b >> 3) * RB + lane * 8 + (b & 7u)]; };
            uint32_t wp[5] = {0, 0, 0, 0, 0}, plo = 0, phi = 0;
            for (uint32_t k = 0; k < v->rmax; ++k) {
                const PauliRot &r = h.rot[e * v->rmax + k];
                for (uint32_t q = 0; q < N; ++q) {
                    byte_at(2 * q) |= (uint8_t)(((r.x >> q) & 1u) << k);
                    byte_at(2 * q + 1) |= (uint8_t)(((r.z >> q) & 1u) << k);
                }
                byte_at(48 + k) = (uint8_t)r.pred;
                plo |= (r.phase & 1u) << k;
                phi |= ((r.phase >> 1) & 1u) << k;
                const uint32_t wk = (uint32_t)__builtin_popcount(r.x | r.z);
                for (int j = 0; j < 5; ++j) wp[j] |= ((wk >> j) & 1u) << k;
            }
            byte_at(56) = (uint8_t)plo;
            byte_at(57) = (uint8_t)phi;
            for (int j = 0; j < 5; ++j) byte_at(58 + j) = (uint8_t)wp[j];
        } else {
            for (uint32_t k = 0; k < v->rmax; ++k) {
                const PauliRot &r = h.rot[e * v->rmax + k];
                uint32_t g[4] = {r.x, r.z, r.phase, r.pred};
                memcpy(tile + NQ * QB + k * RB + lane * 16, g, 16);
            }
        }
        const PauliMeta &m = h.meta[e];
        uint8_t *mt = tile + NQ * QB + RM * RB;
        if (RM > 16) {  // {alive, count, bad (the init kernel computes it), -}, then the 32 order bytes
            const uint32_t g[4] = {m.alive, m.count, 0u, 0u};
            memcpy(mt + lane * 16, g, 16);
            memcpy(mt + 1024 + lane * 16, m.order, 16);
            memcpy(mt + 2048 + lane * 16, m.order + 16, 16);
        } else {
            uint64_t nib = 0;
            for (uint32_t k = 0; k < 16; ++k) nib |= (uint64_t)(m.order[k] & 0xFu) << (4 * k);
            const uint32_t g[4] = {m.alive | (m.count << 16), 0u /* bad: the init kernel computes it */, (uint32_t)nib, (uint32_t)(nib >> 32)};
            memcpy(mt + lane * 16, g, 16);
        }
    }
    HIP_TRY(hipMemcpyAsync(v->state, img.data(), v->state_bytes, hipMemcpyHostToDevice, s));
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
    a.N = N;
    PTArgs pa;
    fill_pt_args(v, a, pa);
    pa.do_clean = do_clean ? 1u : 0u;
    pa.depth_value = depth_value;
    HIP_TRY(ptile_init(v, pa, s));
    HIP_TRY(hipStreamSynchronize(s));  // the staging image dies with this frame
    return QG_OK;
}

}  // namespace qg
