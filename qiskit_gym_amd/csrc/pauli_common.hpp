// pauli_common.hpp -- PauliEnv records shared by the two PauliEnv kernel families and the host
// side that builds them (private to libqgym).
#pragma once

#include <string>
#include <vector>

#include "qgym_host.hpp"

namespace qg {

#define PAULI_RMAX 32u

struct PauliMeta {  // host-side staging record (the device layout is kernels_pauli_tile.hip's)
    uint32_t alive;             // bit k: rotation k still in the DAG
    uint32_t count;             // number of DAG nodes
    uint8_t order[PAULI_RMAX];  // order[i] = rotation index held by DAG node i (petgraph NodeIndex order)
};

struct PauliRot {
    uint32_t x, z;   // base_x / base_z bit q = qubit q (reference pauli/pauli.rs:41-42)
    uint32_t phase;  // base_phase mod 4 (pauli/pauli.rs:43)
    uint32_t pred;   // earlier rotations that do not commute with this one (DAG out-edges, pauli_dag.rs:35-41)
};
static_assert(sizeof(PauliRot) == 16, "PauliRot must be 16 bytes");

// per-env records as the host builds them (PauliNetwork::new, pauli_network.rs:37-77)
struct HostNet {
    std::vector<uint64_t> tab;    // [B][N][2]: X row q, Z row N+q (bit c = column c)
    std::vector<PauliRot> rot;    // [B][rmax]
    std::vector<PauliMeta> meta;  // [B]
};

// micro-ops a gate decomposes into (pauli_network.rs:225-260)
enum : uint32_t { M_NOP = 0, M_H = 1, M_S = 2, M_SX = 3, M_CNOT = 4 };

// PTILE (thread-per-env) family: kernels_pauli_tile.hip
int ptile_alloc(qg_vec *v);
int ptile_reset_seeded(qg_vec *v, uint64_t seed, bool only_done, hipStream_t s, bool from_mask = false);
int ptile_upload(qg_vec *v, const HostNet &h, bool do_clean, int32_t depth_value, hipStream_t s);
hipError_t ptile_step(const qg_vec *v, const StepArgs &a, hipStream_t s);
hipError_t ptile_export(const qg_vec *v, const ObsArgs &a, hipStream_t s);
// observe() as a dense tensor of `out_dtype` (obs_cols <= 64): row words, then expand_rows
hipError_t ptile_observe_typed(qg_vec *v, void *out_dev, int out_dtype, hipStream_t s);
hipError_t ptile_observe_words(qg_vec *v, void *out_dev, hipStream_t s);

}  // namespace qg
