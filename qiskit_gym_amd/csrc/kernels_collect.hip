// kernels_collect.hip -- the device side of a GPU-resident rollout collector, i.e. what sits
// between env.observe() and env.step() when a policy is in the loop:
//
//   qg_expand_packed / qg_vec_observe_dense_as   bit-packed observation -> dense {0,1} tensor in the
//                                                dtype the policy network eats (bf16/f16/f32/int8)
//   qg_sample_actions                            logits -> action, log-prob, entropy, value copy
//   qg_gae                                       generalised advantage estimation over [T, B]
//
// The reference does all of this on the CPU inside twisterl (`collect`, `data_to_torch`;
// rl/synthesis.py:128-138, collecting parameters rl/configs.py:134-144).  These are streaming
// kernels: expand is write-bound (B*obs*sizeof(dtype) bytes), the other two read their inputs once.
#include <hip/hip_fp16.h>

#include "device_common.hpp"
#include "qgym_host.hpp"
#include "pauli_common.hpp"

namespace qg {


static inline unsigned blocks_for(uint64_t threads, unsigned block) { return (unsigned)((threads + block - 1) / block); }

// ---------------------------------------------------------------------------------------------
// expand: one thread produces one 16-byte chunk of the dense output (EPC elements of one row), so
// every store instruction of a wave writes 1 KiB contiguously.  The packed input is 8x..64x
// smaller than the output and is read through the cache (neighbouring threads share a word).
// ---------------------------------------------------------------------------------------------
// word_bytes: 4 / 8 = bit rows (bit c = column c); 1 = PermutationEnv rows (byte = the column that is set)
template <int ES>
__global__ __launch_bounds__(256) void expand_chunks_kernel(const void *packed, int word_bytes, uint64_t n_rows, uint32_t cols,
                                                            uint32_t chunks_per_row, uint4 *out, uint32_t one) {
    constexpr uint32_t EPC = 16 / ES;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row = gid / chunks_per_row;
    if (row >= n_rows) return;
    const uint32_t c0 = (uint32_t)(gid - row * chunks_per_row) * EPC;
    uint32_t bits;
    if (word_bytes == 8) bits = (uint32_t)(reinterpret_cast<const uint64_t *>(packed)[row] >> c0);
    else if (word_bytes == 4) bits = reinterpret_cast<const uint32_t *>(packed)[row] >> c0;
    else {
        const uint32_t col = reinterpret_cast<const uint8_t *>(packed)[row];
        bits = (col >= c0 && col < c0 + EPC) ? 1u << (col - c0) : 0u;
    }
    out[gid] = expand_chunk<ES>(bits & ((1u << EPC) - 1u), one);
}

// rows whose length is not a multiple of the chunk (PauliEnv: 2N + max_rotations columns): the dense
// output is still one flat array, so a thread still owns one aligned 16-byte chunk of it; the chunk's
// elements are the tail of one row and the head of the next (or of several short rows)
template <int ES>
__global__ __launch_bounds__(256) void expand_ragged_kernel(const void *packed, int word_bytes, uint64_t n_rows, uint32_t cols, uint4 *out,
                                                            uint32_t one) {
    constexpr uint32_t EPC = 16 / ES;
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = n_rows * cols, e0 = gid * EPC;
    if (e0 >= total) return;
    uint64_t row;
    if (total <= 0xFFFFFFFFull) row = (uint32_t)e0 / cols;  // 32-bit division: the 64-bit one costs more than the rest of the thread
    else row = e0 / cols;
    uint32_t c = (uint32_t)(e0 - row * cols), bits = 0, filled = 0;
    while (filled < EPC && row < n_rows) {
        const uint64_t w = word_bytes == 8 ? reinterpret_cast<const uint64_t *>(packed)[row] : (uint64_t) reinterpret_cast<const uint32_t *>(packed)[row];
        const uint32_t take = (EPC - filled) < (cols - c) ? (EPC - filled) : (cols - c);
        bits |= ((uint32_t)(w >> c) & ((1u << take) - 1u)) << filled;
        filled += take;
        row += 1;
        c = 0;
    }
    const uint4 v = expand_chunk<ES>(bits, one);
    if (e0 + EPC <= total) {
        out[gid] = v;
    } else {  // the array's last, partial chunk
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint8_t *o = reinterpret_cast<uint8_t *>(out) + gid * 16ull;
        for (uint32_t b = 0; b < (uint32_t)(total - e0) * ES; ++b) o[b] = (uint8_t)(w[b >> 2] >> (8 * (b & 3u)));
    }
}

// any shape / alignment: one thread per element
template <typename T>
__global__ __launch_bounds__(256) void expand_elems_kernel(const void *packed, int word_bytes, uint64_t n_rows, uint32_t cols, T *out,
                                                           T one) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t row = gid / cols;
    if (row >= n_rows) return;
    const uint32_t c = (uint32_t)(gid - row * cols);
    bool bit;
    if (word_bytes == 8) bit = (reinterpret_cast<const uint64_t *>(packed)[row] >> c) & 1u;
    else if (word_bytes == 4) bit = (reinterpret_cast<const uint32_t *>(packed)[row] >> c) & 1u;
    else bit = reinterpret_cast<const uint8_t *>(packed)[row] == c;
    out[gid] = bit ? one : (T)0;
}

// dense int8 {0,1} -> dense dtype (PauliEnv families / widths without the row-word form): 16 input
// bytes per thread
template <int ES>
__global__ __launch_bounds__(256) void widen01_kernel(const uint8_t *in, uint64_t n, void *out, uint32_t one) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16ull;
    if (i0 >= n) return;
    if (i0 + 16 <= n && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + i0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) bits |= ((w[k >> 2] >> (8 * (k & 3))) & 1u) << k;
        uint4 *o = reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(out) + i0 * ES);
        constexpr uint32_t EPC = 16 / ES;
#pragma unroll
        for (uint32_t c = 0; c < (uint32_t)ES; ++c) o[c] = expand_chunk<ES>((bits >> (c * EPC)) & ((1u << EPC) - 1u), one);
        return;
    }
    for (uint64_t i = i0; i < n && i < i0 + 16; ++i) {
        const uint32_t b = in[i] & 1u;
        if constexpr (ES == 1) reinterpret_cast<uint8_t *>(out)[i] = (uint8_t)(b * one);
        else if constexpr (ES == 2) reinterpret_cast<uint16_t *>(out)[i] = (uint16_t)(b * one);
        else reinterpret_cast<uint32_t *>(out)[i] = b * one;
    }
}

// Env::masks (clifford.rs:349-351)
__global__ __launch_bounds__(256) void masks_kernel(const uint8_t *success, uint8_t *out, uint64_t total, uint32_t A) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < total) out[gid] = success[gid / A] ? 0 : 1;
}
// the same, 16 bytes per thread (A >= 16: a chunk spans at most two envs; `out` 16-byte aligned): one byte per thread reaches 0.7 TB/s
__global__ __launch_bounds__(256) void masks16_kernel(const uint8_t *__restrict__ success, uint8_t *__restrict__ out, uint64_t total, uint32_t A, uint64_t B) {
    const uint64_t g = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    if (g >= total) return;
    const uint64_t env0 = g / A;
    const uint32_t left = (uint32_t)((env0 + 1u) * A - g);  // bytes of this chunk that belong to env0 (>= 1)
    const uint32_t m0 = success[env0] ? 0u : 1u, m1 = (env0 + 1u < B && success[env0 + 1u]) ? 0u : 1u;
    uint32_t w[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) v |= ((4u * k + j) < left ? m0 : m1) << (8u * j);
        w[k] = v;
    }
    if (g + 16u <= total) {
        *reinterpret_cast<uint4 *>(out + g) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint32_t k = 0; g + k < total; ++k) out[g + k] = (uint8_t)(w[k >> 2] >> (8u * (k & 3u)));
    }
}
// Indices of the finished envs, packed (qg_vec_reset_done): a wave ballots its `done` flags, the workgroup's 16 waves add their counts up
// through LDS, ONE lane reserves the workgroup's slots of `list` with an atomic add and every done lane writes its env index at its
// prefix.  (One atomic per WAVE with a finished env -- ~650 on one address at 1 % of 65 536 envs, ~12 ns each, one after the other -- made
// this kernel 9 us; 64 workgroups of 1 024 threads take 64 turns.)  The order across workgroups is arbitrary -- every env's reset depends on
// (seed, env) only.
__global__ __launch_bounds__(1024) void compact_done_kernel(const uint8_t *done, uint64_t B, uint32_t *list, uint32_t *count) {
    __shared__ uint32_t wave_cnt[16], wave_base[16];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), wave = threadIdx.x >> 6;
    const bool d = env < B && done[env];
    const uint64_t m = __ballot(d);
    if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x < 16u) {  // the first 16 lanes of wave 0: an exclusive prefix over the waves, the total from lane 15
        const uint32_t c = wave_cnt[threadIdx.x];
        uint32_t incl = c;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 16);
            if ((int)threadIdx.x >= off) incl += up;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 15, 16);
        uint32_t base = 0;
        if (threadIdx.x == 0 && total) base = atomicAdd(count, total);
        base = (uint32_t)__shfl((int)base, 0, 16);
        wave_base[threadIdx.x] = base + incl - c;
    }
    __syncthreads();
    if (d) list[wave_base[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)env;
}
// The length (and the reader ticket behind it) is zeroed right here, on the stream, whatever ran before: the host's idea of it is
// stale as soon as a caller replays a graph that touches the list.  The last kernel that consumes the list zeroes it again (list_count_take).
hipError_t compact_done(const uint8_t *done, uint64_t B, uint32_t *list, uint32_t *count, hipStream_t s) {
    if (hipError_t e = hipMemsetAsync(count, 0, 2 * sizeof(uint32_t), s)) return e;
    hipLaunchKernelGGL(compact_done_kernel, dim3(blocks_for(B, 1024)), dim3(1024), 0, s, done, B, list, count);
    return hipGetLastError();
}

// qg_vec_step_host with pinned (device-mapped) buffers: the step's outputs written straight into the caller's host memory, four envs per
// thread -- one launch instead of three copies whose cost is their latency (0.4 MB at 65 536 envs).  Null pointers are skipped.
__global__ __launch_bounds__(256) void step_outputs_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done, const uint8_t *__restrict__ success,
                                                           float *rewards_out, uint8_t *dones_out, uint8_t *success_out, uint64_t B) {
    const uint64_t e = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (e + 4u <= B) {
        if (rewards_out) *reinterpret_cast<float4 *>(rewards_out + e) = *reinterpret_cast<const float4 *>(reward + e);
        if (dones_out) *reinterpret_cast<uint32_t *>(dones_out + e) = *reinterpret_cast<const uint32_t *>(done + e);
        if (success_out) *reinterpret_cast<uint32_t *>(success_out + e) = *reinterpret_cast<const uint32_t *>(success + e);
    } else {
        for (uint64_t i = e; i < B; ++i) {
            if (rewards_out) rewards_out[i] = reward[i];
            if (dones_out) dones_out[i] = done[i];
            if (success_out) success_out[i] = success[i];
        }
    }
}
hipError_t step_outputs(const float *reward, const uint8_t *done, const uint8_t *success, float *rewards_out, uint8_t *dones_out, uint8_t *success_out,
                        uint64_t B, hipStream_t s) {
    if (!B || (!rewards_out && !dones_out && !success_out)) return hipSuccess;
    hipLaunchKernelGGL(step_outputs_kernel, dim3(blocks_for((B + 3) / 4, 256)), dim3(256), 0, s, reward, done, success, rewards_out, dones_out, success_out, B);
    return hipGetLastError();
}

// qg_vec_sync: the OR of the per-env fault words in ONE word of pinned host memory -- a sync costs a launch and four bytes instead of a copy
// of the whole array and a scan on the host (65 536 envs: 256 KiB per call).  Every workgroup ORs its share (16 bytes per load, grid stride)
// into scratch[0]; the last one to finish (ticket scratch[1]) hands the word to the host and leaves both zero for the next call.
__global__ __launch_bounds__(256) void fault_any_kernel(const uint32_t *__restrict__ error, uint64_t B, uint32_t *scratch, uint32_t *out_host) {
    const uint64_t n4 = B / 4u, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint4 *e4 = reinterpret_cast<const uint4 *>(error);
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const uint4 q = e4[i];
        acc |= (q.x | q.y) | (q.z | q.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (B & 3u)) acc |= error[n4 * 4u + threadIdx.x];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc |= __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63u) == 0 && acc) atomicOr(&scratch[0], acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&scratch[1], 1u) == gridDim.x - 1u) {
            *out_host = atomicOr(&scratch[0], 0u);
            scratch[0] = 0;
            scratch[1] = 0;
        }
    }
}
hipError_t fault_any(const uint32_t *error, uint64_t B, uint32_t *scratch, uint32_t *out_host, hipStream_t s) {
    const uint64_t want = (B / 4u + 1023u) / 1024u;  // four loads of 16 bytes per thread
    hipLaunchKernelGGL(fault_any_kernel, dim3((unsigned)(want < 1 ? 1 : want > 256 ? 256 : want)), dim3(256), 0, s, error, B, scratch, out_host);
    return hipGetLastError();
}

hipError_t masks_fill(const uint8_t *success, uint8_t *out, uint64_t B, uint32_t A, hipStream_t s) {
    const uint64_t total = B * A;
    if (!total) return hipSuccess;
    if (A >= 16u && !(reinterpret_cast<uintptr_t>(out) & 15u))
        hipLaunchKernelGGL(masks16_kernel, dim3(blocks_for((total + 15u) / 16u, 256)), dim3(256), 0, s, success, out, total, A, B);
    else
        hipLaunchKernelGGL(masks_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, success, out, total, A);
    return hipGetLastError();
}


static bool dtype_info(int dtype, uint32_t &elem_size, uint32_t &one) {
    switch (dtype) {
    case QG_DT_I8: elem_size = 1; one = 1u; return true;
    case QG_DT_BF16: elem_size = 2; one = 0x3F80u; return true;
    case QG_DT_F16: elem_size = 2; one = 0x3C00u; return true;
    case QG_DT_F32: elem_size = 4; one = 0x3F800000u; return true;
    }
    return false;
}

static int expand_packed_impl(const void *packed_dev, int word_bytes, uint64_t n_rows, uint32_t cols, void *out_dev, int out_dtype,
                              hipStream_t s) {
    uint32_t es = 0, one = 0;
    if (!dtype_info(out_dtype, es, one)) return set_error(QG_ERR_INVALID, "unknown output dtype %d", out_dtype);
    if (word_bytes != 1 && word_bytes != 4 && word_bytes != 8) return set_error(QG_ERR_INVALID, "word_bytes must be 1, 4 or 8");
    if (cols == 0 || (word_bytes != 1 && cols > (uint32_t)word_bytes * 8u) || (word_bytes == 1 && cols > 256u))
        return set_error(QG_ERR_INVALID, "cols does not fit the packed word");
    if (n_rows == 0) return QG_OK;
    const uint32_t epc = 16 / es;
    if (cols % epc == 0 && (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0) {
        const uint32_t cpr = cols / epc;
        const unsigned grid = blocks_for(n_rows * cpr, 256);
        uint4 *o = reinterpret_cast<uint4 *>(out_dev);
        if (es == 1) hipLaunchKernelGGL(expand_chunks_kernel<1>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, cpr, o, one);
        else if (es == 2) hipLaunchKernelGGL(expand_chunks_kernel<2>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, cpr, o, one);
        else hipLaunchKernelGGL(expand_chunks_kernel<4>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, cpr, o, one);
    } else if (word_bytes != 1 && cols >= 1 && (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0) {
        const unsigned grid = blocks_for((n_rows * cols + epc - 1) / epc, 256);
        uint4 *o = reinterpret_cast<uint4 *>(out_dev);
        if (es == 1) hipLaunchKernelGGL(expand_ragged_kernel<1>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, o, one);
        else if (es == 2) hipLaunchKernelGGL(expand_ragged_kernel<2>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, o, one);
        else hipLaunchKernelGGL(expand_ragged_kernel<4>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols, o, one);
    } else {
        const unsigned grid = blocks_for(n_rows * cols, 256);
        if (es == 1)
            hipLaunchKernelGGL(expand_elems_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols,
                               reinterpret_cast<uint8_t *>(out_dev), (uint8_t)one);
        else if (es == 2)
            hipLaunchKernelGGL(expand_elems_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols,
                               reinterpret_cast<uint16_t *>(out_dev), (uint16_t)one);
        else
            hipLaunchKernelGGL(expand_elems_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, packed_dev, word_bytes, n_rows, cols,
                               reinterpret_cast<uint32_t *>(out_dev), one);
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

// ---- the trait's Vec<i64> wire format (Env::set_state / get_state, clifford.rs:299-304: D * D entries per env, > 0 means 1) ----------------
// out: one thread per 16-byte chunk = two int64 entries of the flat [n_rows * cols] array; the row words are tiny and stay in cache, the
// stores are wave-contiguous (1 KiB per instruction): 537 MB for CliffordEnv 16q x 65 536.  (The export kernels before wrote a row per
// thread, entry by entry: 0.6 TB/s.)
__global__ __launch_bounds__(256) void expand_i64_kernel(const void *packed, int word_bytes, uint64_t n_rows, uint32_t cols, int64_t *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t total = n_rows * cols, e0 = gid * 2u;
    if (e0 >= total) return;
    uint64_t row;
    if (total <= 0xFFFFFFFFull) row = (uint32_t)e0 / cols;  // 32-bit division: the 64-bit one costs more than the rest of the thread
    else row = e0 / cols;
    const uint32_t c = (uint32_t)(e0 - row * cols);
    auto word = [&](uint64_t r) -> uint64_t {
        return word_bytes == 8 ? reinterpret_cast<const uint64_t *>(packed)[r] : (uint64_t) reinterpret_cast<const uint32_t *>(packed)[r];
    };
    const uint64_t w = word(row);
    const uint64_t v0 = (w >> c) & 1ull;
    if (e0 + 1 >= total) {  // the array's last entry (odd total)
        out[e0] = (int64_t)v0;
        return;
    }
    const uint64_t v1 = (c + 1u < cols) ? (w >> (c + 1u)) & 1ull : word(row + 1) & 1ull;
    typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
    u64x2 v = {v0, v1};
    *reinterpret_cast<u64x2 *>(out + e0) = v;
}

hipError_t expand_rows_i64(const void *words_dev, int word_bytes, uint64_t n_rows, uint32_t cols, int64_t *out_dev, hipStream_t s) {
    if (!n_rows || !cols) return hipSuccess;
    hipLaunchKernelGGL(expand_i64_kernel, dim3(blocks_for((n_rows * cols + 1) / 2, 256)), dim3(256), 0, s, words_dev, word_bytes, n_rows, cols, out_dev);
    return hipGetLastError();
}

// in: the flat entry stream as a BIT stream (bit e of the stream = entry e > 0), 64 entries per wave instruction: lane l of the wave that
// owns entries [64 i, 64 i + 64) loads entry 64 i + l (512 contiguous bytes of int64, 64 of uint8) and the wave's ballot IS stream word i.
// The init kernels then cut their row words out of the stream (QG_FMT_BITS, bits_window): no thread walks a row of 8-byte entries.
template <typename T>
__global__ __launch_bounds__(256) void pack_bitstream_kernel(const T *src, uint64_t n_entries, uint64_t *out) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool one = gid < n_entries && src[gid] > 0;  // whole waves reach the ballot
    const uint64_t m = __ballot(one);
    if ((threadIdx.x & 63u) == 0 && (gid >> 6) < (n_entries + 63u) / 64u) out[gid >> 6] = m;
}
// int64 entries, 16 bytes per lane: lane l of the wave that owns entries [128 i, 128 i + 128) loads entries 128 i + 2 l and + 2 l + 1 (1 KiB per wave
// instruction instead of 512 bytes); the two ballots are the even and the odd entries' bits, interleaved on the scalar unit into stream
// words 2 i and 2 i + 1.  n_entries even and src 16-byte aligned (the launcher checks).
__device__ inline uint64_t spread_bits32(uint64_t x) {  // bit k of the low word -> bit 2 k
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
__global__ __launch_bounds__(256) void pack_bitstream_i64x2_kernel(const int64_t *src, uint64_t n_entries, uint64_t *out) {
    typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, e0 = 2u * gid;
    i64x2 v = {0, 0};
    if (e0 < n_entries) v = reinterpret_cast<const i64x2 *>(src)[gid];  // (n_entries is even)
    const uint64_t even = __ballot(v.x > 0), odd = __ballot(v.y > 0);
    if ((threadIdx.x & 63u) == 0) {
        const uint64_t w = gid >> 6, n_words = (n_entries + 63u) / 64u;  // this wave's entries: stream words 2 w and 2 w + 1
        if (2u * w < n_words) out[2u * w] = spread_bits32(even & 0xFFFFFFFFull) | (spread_bits32(odd & 0xFFFFFFFFull) << 1);
        if (2u * w + 1u < n_words) out[2u * w + 1u] = spread_bits32(even >> 32) | (spread_bits32(odd >> 32) << 1);
    }
}

hipError_t pack_bitstream(const void *src, int elem_bytes, uint64_t n_entries, uint64_t *out_words, hipStream_t s) {
    if (!n_entries) return hipSuccess;
    if (elem_bytes == 8 && !(n_entries & 1u) && !(reinterpret_cast<uintptr_t>(src) & 15u)) {
        const unsigned grid = blocks_for((n_entries / 2u + 63u) / 64u * 64u, 256);
        hipLaunchKernelGGL(pack_bitstream_i64x2_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const int64_t *>(src), n_entries, out_words);
        return hipGetLastError();
    }
    const unsigned grid = blocks_for((n_entries + 63u) / 64u * 64u, 256);
    if (elem_bytes == 8) hipLaunchKernelGGL(pack_bitstream_kernel<int64_t>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const int64_t *>(src), n_entries, out_words);
    else hipLaunchKernelGGL(pack_bitstream_kernel<int8_t>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const int8_t *>(src), n_entries, out_words);
    return hipGetLastError();
}

hipError_t expand_rows(const void *words_dev, int word_bytes, uint64_t n_rows, uint32_t cols, void *out_dev, int out_dtype, hipStream_t s) {
    return expand_packed_impl(words_dev, word_bytes, n_rows, cols, out_dev, out_dtype, s) == QG_OK ? hipSuccess : hipErrorInvalidValue;
}

static int widen01_impl(const uint8_t *in_dev, uint64_t n, void *out_dev, int out_dtype, hipStream_t s) {
    uint32_t es = 0, one = 0;
    if (!dtype_info(out_dtype, es, one)) return set_error(QG_ERR_INVALID, "unknown output dtype %d", out_dtype);
    if (n == 0) return QG_OK;
    const unsigned grid = blocks_for((n + 15) / 16, 256);
    if (es == 1) hipLaunchKernelGGL(widen01_kernel<1>, dim3(grid), dim3(256), 0, s, in_dev, n, out_dev, one);
    else if (es == 2) hipLaunchKernelGGL(widen01_kernel<2>, dim3(grid), dim3(256), 0, s, in_dev, n, out_dev, one);
    else hipLaunchKernelGGL(widen01_kernel<4>, dim3(grid), dim3(256), 0, s, in_dev, n, out_dev, one);
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

// ---------------------------------------------------------------------------------------------
// sampling: 16 lanes per env (4 envs per wavefront), two passes over the (cache-resident) row.
// An exponential race draws from softmax(logits) exactly:
//   base = rng_draw(seed, env, counter)                      one 64-bit counter-RNG draw per env
//   h[a] = hash32(base, a)                                   (lowbias32 rounds keyed by both halves)
//   u[a] = ((h[a] >> 9) + 0.5) * 2^-23                       in (0, 1), exact in f32
//   E[a] = -log(u[a]);   action = argmin E[a] / exp(logit[a] - max)      (lowest index on ties)
// P(a wins) = p_a / sum p: the minimum of independent exponentials with rates p_a.  The same
// exp(logit - max) terms give the log-prob and the entropy, so the row is exponentiated once.
// The result does not depend on how the row is split over lanes (min / max are associative; the
// two f32 sums are reduced in a fixed order).
// ---------------------------------------------------------------------------------------------
struct SampleArgs {
    const void *logits;
    const uint8_t *mask;  // [B][A] 1 = allowed, or null
    void *actions;
    float *logp;
    float *entropy;
    float *values;
    uint64_t ld, B, seed, counter;
    const uint64_t *clock;  // device clock added to `counter`, or null
    uint32_t A;
    int32_t value_col;
    int32_t act64;
};

template <typename LT>
__device__ inline float logit_to_float(LT v);
template <>
__device__ inline float logit_to_float<float>(float v) { return v; }
template <>
__device__ inline float logit_to_float<uint16_t>(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }  // bf16
template <>
__device__ inline float logit_to_float<__half>(__half v) { return __half2float(v); }

__host__ __device__ inline float sample_uniform(uint64_t base, uint32_t a) {
    uint32_t x = (uint32_t)(base >> 32) + a * 0x9E3779B9u;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= (uint32_t)base;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

template <typename LT>
__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    const uint32_t sl = threadIdx.x & 15u;
    const uint64_t env_raw = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const bool live = env_raw < a.B;
    const uint64_t env = live ? env_raw : a.B - 1;  // idle sub-groups recompute the last env, write nothing
    const LT *row = reinterpret_cast<const LT *>(a.logits) + env * a.ld;
    const uint8_t *mrow = a.mask ? a.mask + env * a.A : nullptr;
    const float INF = __builtin_huge_valf();
    float m = -INF;
    for (uint32_t i = sl; i < a.A; i += 16) {
        if (mrow && !mrow[i]) continue;
        m = fmaxf(m, logit_to_float<LT>(row[i]));
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 16));
    const uint64_t base = rng_draw(a.seed, env, a.counter + clock_of(a.clock));
    float best_q = INF, best_d = 0.0f, ssum = 0.0f, wsum = 0.0f;
    uint32_t best_a = 0xFFFFFFFFu;
    for (uint32_t i = sl; i < a.A; i += 16) {
        if (mrow && !mrow[i]) continue;
        const float d = logit_to_float<LT>(row[i]) - m;
        // raw hardware exp2 / log2 / rcp (1 ulp each): the race only needs its winner (log2 instead of ln scales every key by the
        // same constant), the sums feed a log-prob quoted to 1e-5
        const float ex = __builtin_amdgcn_exp2f(d * 1.44269504088896340736f);
        ssum += ex;
        wsum += ex * d;
        const float q = -__builtin_amdgcn_logf(sample_uniform(base, i)) * __builtin_amdgcn_rcpf(ex);  // ex == 0 (logit far below the max): q = inf, never wins
        if (q < best_q || best_a == 0xFFFFFFFFu) {
            best_q = q;
            best_a = i;
            best_d = d;
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
        const float oq = __shfl_xor(best_q, off, 16), od = __shfl_xor(best_d, off, 16);
        const uint32_t oa = __shfl_xor(best_a, off, 16);
        ssum += __shfl_xor(ssum, off, 16);
        wsum += __shfl_xor(wsum, off, 16);
        const bool take = oa != 0xFFFFFFFFu && (best_a == 0xFFFFFFFFu || oq < best_q || (oq == best_q && oa < best_a));
        if (take) {
            best_q = oq;
            best_a = oa;
            best_d = od;
        }
    }
    if (!live || sl != 0) return;
    const bool none = best_a == 0xFFFFFFFFu;  // every action masked: the env is finished (clifford.rs:349-351)
    const int64_t act = none ? 0 : (int64_t)best_a;
    if (a.act64) reinterpret_cast<int64_t *>(a.actions)[env] = act;
    else reinterpret_cast<int32_t *>(a.actions)[env] = (int32_t)act;
    const float log_s = logf(ssum);
    if (a.logp) a.logp[env] = none ? 0.0f : best_d - log_s;
    if (a.entropy) a.entropy[env] = none ? 0.0f : log_s - wsum / ssum;
    if (a.values && a.value_col >= 0) a.values[env] = logit_to_float<LT>(row[a.value_col]);
}

// ---------------------------------------------------------------------------------------------
// GAE: one thread per env walks its T steps backwards; loads of a wave are contiguous in env.
//   nd_t   = 1 - done_t
//   delta  = r_t + (gamma * V_{t+1}) * nd_t - V_t
//   A_t    = delta + ((gamma*lambda) * nd_t) * A_{t+1};   R_t = A_t + V_t
// (f32, this operation order, no FMA: the numpy restatement in tests/ reproduces it bit for bit)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gae_kernel(const float *rewards, const float *values, const uint8_t *dones, const float *last_values,
                                                  float gamma, float gamma_lambda, uint32_t T, uint64_t B, float *adv, float *ret) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B) return;
    float next_v = last_values ? last_values[e] : 0.0f;
    float acc = 0.0f;
    for (uint32_t t = T; t-- > 0;) {
        const uint64_t i = (uint64_t)t * B + e;
        const float nd = dones[i] ? 0.0f : 1.0f;
        const float v = values[i];
        const float delta = rewards[i] + (gamma * next_v) * nd - v;
        acc = delta + (gamma_lambda * nd) * acc;
        adv[i] = acc;
        if (ret) ret[i] = acc + v;
        next_v = v;
    }
}

}  // namespace qg

using namespace qg;

extern "C" {

int qg_expand_packed(const void *packed_dev, int word_bytes, uint64_t n_rows, uint32_t cols, void *out_dev, int out_dtype, void *stream) {
    if (!packed_dev || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    return expand_packed_impl(packed_dev, word_bytes, n_rows, cols, out_dev, out_dtype, (hipStream_t)stream);
}

int qg_widen_dense(const int8_t *obs_dev, uint64_t n_elems, void *out_dev, int out_dtype, void *stream) {
    if (!obs_dev || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    return widen01_impl(reinterpret_cast<const uint8_t *>(obs_dev), n_elems, out_dev, out_dtype, (hipStream_t)stream);
}

int qg_vec_observe_dense_as(qg_vec *v, void *out_dev, int out_dtype, void *stream) {
    if (!v || !out_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (out_dtype == QG_DT_I8) return qg_vec_observe_dense(v, reinterpret_cast<int8_t *>(out_dev), stream);
    qg_vec_info info;
    int rc = qg_vec_get_info(v, &info);
    if (rc != QG_OK) return rc;
    QG_ON_DEVICE(v);
    const uint64_t obs = (uint64_t)info.obs_rows * info.obs_cols;
    if (v->layout == LAYOUT_PAULI && (uint32_t)info.obs_cols <= 64u) {
        v->perm_draw = true;  // PauliEnv::observe draws a new qubit permutation (pauli.rs:657-662)
        const hipError_t e = ptile_observe_typed(v, out_dev, out_dtype, (hipStream_t)stream);
        v->perm_draw = false;
        v->observe_counter += 1;
        HIP_TRY(e);
        return QG_OK;
    }
    if (v->layout == LAYOUT_PAULI) {  // lane-group family or more than 64 columns: int8 observation, then widen
        rc = ensure_scratch_public(v, v->B * obs);
        if (rc != QG_OK) return rc;
        rc = qg_vec_observe_dense(v, reinterpret_cast<int8_t *>(v->scratch), stream);
        if (rc != QG_OK) return rc;
        return widen01_impl(reinterpret_cast<const uint8_t *>(v->scratch), v->B * obs, out_dev, out_dtype, (hipStream_t)stream);
    }
    uint32_t es = 0, one = 0;
    if (!dtype_info(out_dtype, es, one)) return set_error(QG_ERR_INVALID, "unknown output dtype %d", out_dtype);
    if (v->layout == LAYOUT_TILE && v->D % (16 / es) == 0 && (reinterpret_cast<uintptr_t>(out_dev) & 15u) == 0) {
        // hot layout: expand straight from the resident tiles, no packed intermediate
        HIP_TRY(qm_export_typed(v->state, v->B, v->N, v->D, v->nxp, v->has_z, out_dev, es, one, (hipStream_t)stream));
        return QG_OK;
    }
    rc = ensure_scratch_public(v, v->B * (uint64_t)info.packed_words_per_env * info.packed_word_bytes);
    if (rc != QG_OK) return rc;
    rc = qg_vec_observe_packed(v, v->scratch, stream);
    if (rc != QG_OK) return rc;
    return expand_packed_impl(v->scratch, (int)info.packed_word_bytes, v->B * (uint64_t)info.packed_words_per_env, (uint32_t)info.obs_cols,
                              out_dev, out_dtype, (hipStream_t)stream);
}

int qg_sample_actions(const void *logits_dev, int logits_dtype, uint64_t ld, uint64_t batch, uint32_t num_actions, const uint8_t *mask_dev,
                      uint64_t seed, uint64_t counter, const uint64_t *clock_dev, void *actions_dev, int action_dtype, float *logp_dev,
                      float *entropy_dev, int32_t value_col, float *values_dev, void *stream) {
    if (!logits_dev || !actions_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (num_actions == 0 || ld < num_actions) return set_error(QG_ERR_INVALID, "bad logits shape");
    if (action_dtype != QG_ACT_I32 && action_dtype != QG_ACT_I64) return set_error(QG_ERR_INVALID, "bad action dtype");
    if (value_col >= 0 && ((uint64_t)value_col >= ld || !values_dev)) return set_error(QG_ERR_INVALID, "bad value column");
    if (batch == 0) return QG_OK;
    SampleArgs a;
    a.logits = logits_dev;
    a.mask = mask_dev;
    a.actions = actions_dev;
    a.logp = logp_dev;
    a.entropy = entropy_dev;
    a.values = values_dev;
    a.ld = ld;
    a.B = batch;
    a.seed = seed ^ 0x73616D70ull;  // "samp": a stream of its own next to reset / coin / perm draws
    a.counter = counter;
    a.clock = clock_dev;
    a.A = num_actions;
    a.value_col = value_col;
    a.act64 = action_dtype == QG_ACT_I64;
    const dim3 grid(blocks_for(batch * 16ull, 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (logits_dtype) {
    case QG_DT_F32: hipLaunchKernelGGL(sample_kernel<float>, grid, block, 0, s, a); break;
    case QG_DT_BF16: hipLaunchKernelGGL(sample_kernel<uint16_t>, grid, block, 0, s, a); break;
    case QG_DT_F16: hipLaunchKernelGGL(sample_kernel<__half>, grid, block, 0, s, a); break;
    default: return set_error(QG_ERR_INVALID, "logits dtype must be f32, bf16 or f16");
    }
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

int qg_gae(const float *rewards_dev, const float *values_dev, const uint8_t *dones_dev, const float *last_values_dev, float gamma,
           float gae_lambda, size_t n_steps, uint64_t batch, float *advantages_dev, float *returns_dev, void *stream) {
    if (!rewards_dev || !values_dev || !dones_dev || !advantages_dev) return set_error(QG_ERR_INVALID, "null argument");
    if (n_steps > 0x7fffffffu) return set_error(QG_ERR_INVALID, "too many steps");
    if (n_steps == 0 || batch == 0) return QG_OK;
    const float gl = gamma * gae_lambda;
    hipLaunchKernelGGL(gae_kernel, dim3(blocks_for(batch, 256)), dim3(256), 0, (hipStream_t)stream, rewards_dev, values_dev, dones_dev,
                       last_values_dev, gamma, gl, (uint32_t)n_steps, batch, advantages_dev, returns_dev);
    HIP_TRY(hipGetLastError());
    return QG_OK;
}

}  // extern "C"
