// device_common.hpp -- device helpers shared by the step kernels (gfx950, wave64).
#pragma once

#include "qgym_internal.hpp"
#include "qgym_plan.hpp"

namespace qg {

#define QG_WAVE 64

// The kernel device clock (qg_vec_set_kernel_clock, include/qgym.h): how long a launch's waves were on the machine, measured by the waves
// themselves -- profiler-independent.  The constant-rate counter (s_memrealtime, 100 MHz on gfx950) is read as the kernel's first
// instruction by every wave (two SGPRs, no wait); with a slot, every wave waits for its own loads and stores at its end, reads the counter again
// and one lane stores the pair {entry, exit} into the wave's OWN record of the launch's slot (wave w of the grid -> record w; plain 16-byte stores to
// distinct addresses: atomics folding 1 024 waves into one pair of words queue at one L2 channel for ~12 ns each, and every wave's exit stamp
// waits for its stores behind them -- the headline kernel read 17 us that way).  The host takes min(entry) and max(exit) over the records.
// Without a slot (every ordinary launch) the cost is that one scalar instruction and a scalar branch.  Declare as the first statement of a kernel;
// the destructor runs on every return path (a ragged last wave whose lanes leave at different points writes its record more than once: the
// last write, the latest exit, stays).
struct KernelClock {
    unsigned long long *slot;
    unsigned long long t0;
    uint32_t waves;
    __device__ inline KernelClock(unsigned long long *s, uint32_t n) : slot(s), t0(wall_clock64()), waves(n) {}
    __device__ inline ~KernelClock() {
        if (slot) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the wave's own stores have reached the L2 (gfx9: stores count in vmcnt)
            const unsigned long long t1 = wall_clock64();
            const uint32_t wave = blockIdx.x * ((blockDim.x + 63u) >> 6) + (threadIdx.x >> 6);
            const unsigned long long m = __ballot(1);
            if (wave < waves && __lane_id() == (unsigned)__ffsll((long long)m) - 1u)
                *reinterpret_cast<ulonglong2 *>(slot + 2ull * wave) = make_ulonglong2(t0, t1);
        }
    }
    KernelClock(const KernelClock &) = delete;
    KernelClock &operator=(const KernelClock &) = delete;
};

// Development (-DQG_PHASE_CLOCK, tools/build_variant.sh + tools/phase_clock.py): where inside a kernel the time goes.  Lane 0 of the calling wave of workgroup b
// stores the clock into the launch's slot at record (waves - 1 - (8 b + idx)) -- the records' upper half, far beyond the grid's own waves (grids of up to
// waves / 16 workgroups) -- after waiting for the wave's outstanding memory operations, so the stamp says "everything before this line has happened".
// Compiled out otherwise.
#ifdef QG_PHASE_CLOCK
__device__ inline void phase_stamp(unsigned long long *slot, uint32_t waves, uint32_t idx) {
    const uint32_t at = 8u * blockIdx.x + idx;
    if (slot && __lane_id() == 0 && idx < 8u && at < waves / 2u) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t = wall_clock64();
        *reinterpret_cast<ulonglong2 *>(slot + 2ull * (waves - 1u - at)) = make_ulonglong2(t, t);
    }
}
#else
__device__ inline void phase_stamp(unsigned long long *, uint32_t, uint32_t) {}
#endif

__device__ inline int64_t load_action(const void *actions, uint64_t idx, bool act64) {
    return act64 ? reinterpret_cast<const int64_t *>(actions)[idx]
                 : (int64_t) reinterpret_cast<const int32_t *>(actions)[idx];
}

// compact_done's job done by the kernel that finishes the envs: the wave's finished envs go to the list with one atomic per wave
// (call from every lane still alive; `fin` on one lane per env).  The length must be zero when the kernel starts (qgym_api.cpp keeps it
// so); an entry that would land past the list's `cap` = B slots is dropped rather than written.
__device__ inline void done_list_append(uint32_t *list, uint32_t *count, bool fin, uint64_t env, uint64_t cap) {
    const uint64_t m = __ballot(fin);
    if (!m) return;
    const uint32_t lane = __lane_id(), first = (uint32_t)__ffsll((long long)m) - 1u;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, first);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (fin && slot < cap) list[slot] = (uint32_t)env;
}

// The same append for a whole workgroup (call from ALL of its threads): one atomic per workgroup with a finisher instead of one per wave --
// every append goes to one address, ~12 ns each, one after the other, and they are the tail of the launch.  WAVES = 4 (256 threads) or 16
// (1 024 threads: the LIST step kernels, whose 64 workgroups at 65 536 envs take 64 turns instead of 256).
template <int WAVES = 4>
__device__ inline void done_list_append_block(uint32_t *list, uint32_t *count, bool fin, uint64_t env, uint64_t cap) {
    static_assert(WAVES == 4 || WAVES == 16, "256 or 1 024 threads");
    __shared__ uint32_t wave_count[WAVES], wave_base[WAVES];
    const uint64_t m = __ballot(fin);
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if constexpr (WAVES == 4) {
        if (threadIdx.x == 0) {
            const uint32_t c0 = wave_count[0], c1 = wave_count[1], c2 = wave_count[2], c3 = wave_count[3], total = c0 + c1 + c2 + c3;
            const uint32_t base = total ? atomicAdd(count, total) : 0u;
            wave_base[0] = base;
            wave_base[1] = base + c0;
            wave_base[2] = base + c0 + c1;
            wave_base[3] = base + c0 + c1 + c2;
        }
    } else {
        if (threadIdx.x < 16u) {  // an exclusive prefix over the waves on the first 16 lanes, the total from lane 15
            const uint32_t c = wave_count[threadIdx.x];
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
    }
    __syncthreads();
    if (fin) {
        const uint32_t slot = wave_base[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < cap) list[slot] = (uint32_t)env;
    }
}

// The finished envs of a step as one bit each (StepArgs::done_mask): the wave's ballot, one 8-byte store per wave -- no counter, no atomics, nothing
// to zero (every launch rewrites every word) -- and, behind the words, one byte per 32 envs with the number of bits set there (`done_mask_counts`):
// what the reader sums, an eighth of the words' bytes.  Call from every lane of every wave of a thread-per-env grid (`fin` false past the batch's end).
// Layout of a mask buffer for a batch of B: W = 4 * ceil(B / 256) words, then 2 W count bytes, then the hint word: a wave WITH a finisher stores the
// launch's number there (plain stores of one value: no read-modify-write), so a reader that finds another number knows every count is zero and leaves
// without summing them (a stale equal number -- the same launch of a replayed graph's previous replay -- only costs the sum).
__host__ __device__ inline uint64_t done_mask_words(uint64_t B) { return 4ull * ((B + 255ull) / 256ull); }
__host__ __device__ inline uint64_t done_mask_bytes(uint64_t B) { return done_mask_words(B) * 10ull + 8ull; }
__device__ inline uint32_t *done_mask_hint(uint64_t *mask, uint64_t B) { return reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(mask) + done_mask_words(B) * 10ull); }
__device__ inline const uint32_t *done_mask_hint(const uint64_t *mask, uint64_t B) { return done_mask_hint(const_cast<uint64_t *>(mask), B); }
__device__ inline uint8_t *done_mask_counts(uint64_t *mask, uint64_t B) { return reinterpret_cast<uint8_t *>(mask + done_mask_words(B)); }
__device__ inline const uint8_t *done_mask_counts(const uint64_t *mask, uint64_t B) { return reinterpret_cast<const uint8_t *>(mask + done_mask_words(B)); }
__device__ inline void done_mask_store(uint64_t *mask, uint64_t B, bool fin, uint64_t env, uint32_t epoch) {
    const uint64_t m = __ballot(fin);
    if (__lane_id() == 0) {
        if (m) *done_mask_hint(mask, B) = epoch;
        mask[env >> 6] = m;
        reinterpret_cast<uint16_t *>(done_mask_counts(mask, B))[env >> 6] = (uint16_t)((uint32_t)__popc((uint32_t)m) | ((uint32_t)__popc((uint32_t)(m >> 32)) << 8));
    }
}
// ... of a two-lanes-per-env grid (`tid` = 2 env + half; `fin` on the even lane): a wave holds 32 envs, half a word and one count byte
__device__ inline void done_mask_store_pairs(uint64_t *mask, uint64_t B, bool fin, uint64_t tid, uint32_t epoch) {
    uint64_t m = __ballot(fin && !(tid & 1ull));  // bit 2k: env k of the wave
    m = (m | (m >> 1)) & 0x3333333333333333ull;
    m = (m | (m >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    m = (m | (m >> 4)) & 0x00FF00FF00FF00FFull;
    m = (m | (m >> 8)) & 0x0000FFFF0000FFFFull;
    m = (m | (m >> 16)) & 0x00000000FFFFFFFFull;
    if (__lane_id() == 0) {
        if (m) *done_mask_hint(mask, B) = epoch;
        reinterpret_cast<uint32_t *>(mask)[tid >> 6] = (uint32_t)m;
        done_mask_counts(mask, B)[tid >> 6] = (uint8_t)__popc((uint32_t)m);
    }
}
// The reader's side: every workgroup of the reset kernel sums the counts for itself (B / 32 bytes: 2 KB at 65 536 envs) -- thread t of the workgroup's T
// (256, or 64 for the one-wave kernels) owns the words [t c, (t + 1) c), c = ceil(words / T), i.e. 2 c count bytes (one 8-byte load at c = 4).
//   done_mask_load   this thread's share of the counts (its load flies with whatever the caller issues next)
//   done_mask_scan   call from all T threads: two barriers; part[t] = bits before thread t's chunk, part[T] = the total, which it returns
//   done_mask_find   "the i-th finished env" for an i that is the same on every thread of the workgroup (the tree path's entry): the thread whose
//                    chunk holds it loads its words and answers; one more barrier, no search
//   done_mask_nth    the same for any thread and any i < total: a search over the T partial sums and a walk over one chunk's words
struct DoneMaskShare {
    uint32_t bits, before;
};
__device__ inline uint32_t done_mask_pick(uint64_t m, uint32_t r) {  // position of the r-th set bit (r < popcount)
    for (; r; --r) m &= m - 1ull;
    return (uint32_t)__ffsll((long long)m) - 1u;
}
__device__ inline uint32_t byte_sum4(uint32_t x) { return (x * 0x01010101u) >> 24; }  // (each byte <= 32: no carry out of the top byte)
template <uint32_t T = 256>
__device__ inline void done_mask_load(const uint64_t *mask, uint64_t B, uint32_t words, DoneMaskShare &sh) {
    const uint32_t chunk = (words + T - 1u) / T;
    const uint8_t *cnt = done_mask_counts(mask, B);
    sh.bits = 0;
    if (chunk == 4u) {  // 8 count bytes
        if (4u * threadIdx.x < words) {  // (T need not divide the words: ptile_reset_tree_kernel's 320 threads)
            const uint2 v = reinterpret_cast<const uint2 *>(cnt)[threadIdx.x];
            sh.bits = byte_sum4(v.x) + byte_sum4(v.y);
        }
    } else if ((chunk & 3u) == 0u) {  // whole 8-byte pieces
        for (uint32_t k = 0; k < chunk; k += 4u) {
            const uint32_t w = threadIdx.x * chunk + k;
            if (w < words) {  // (words is a multiple of 4: a piece is inside or outside as a whole)
                const uint2 v = *reinterpret_cast<const uint2 *>(cnt + 2u * w);
                sh.bits += byte_sum4(v.x) + byte_sum4(v.y);
            }
        }
    } else {
        for (uint32_t k = 0; k < chunk; ++k) {
            const uint32_t w = threadIdx.x * chunk + k;
            if (w < words) {
                const uint32_t c2 = reinterpret_cast<const uint16_t *>(cnt)[w];
                sh.bits += (c2 & 0xFFu) + (c2 >> 8);
            }
        }
    }
}
template <uint32_t T = 256>
__device__ inline uint32_t done_mask_scan(DoneMaskShare &sh, uint32_t *part /* LDS [T + 2 + T / 64] (T + 1 + 5 holds 64 and 256) */) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    uint32_t incl = sh.bits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
        if ((int)lane >= off) incl += up;
    }
    if (lane == 63u) part[T + 1u + wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; ++w) base += part[T + 1u + w];
    sh.before = base + incl - sh.bits;
    part[threadIdx.x + 1u] = base + incl;
    if (threadIdx.x == 0) part[0] = 0;
    __syncthreads();
    return part[T];
}
// the r-th set bit of the chunk of thread `t`
template <uint32_t T = 256>
__device__ inline uint32_t done_mask_in_chunk(const uint64_t *mask, uint32_t words, uint32_t t, uint32_t rem) {
    const uint32_t chunk = (words + T - 1u) / T;
    if (chunk == 4u) {  // the four words at once
        const uint4 a = reinterpret_cast<const uint4 *>(mask)[2u * t], b = reinterpret_cast<const uint4 *>(mask)[2u * t + 1u];
        const uint64_t w[4] = {(uint64_t)a.x | ((uint64_t)a.y << 32), (uint64_t)a.z | ((uint64_t)a.w << 32), (uint64_t)b.x | ((uint64_t)b.y << 32),
                               (uint64_t)b.z | ((uint64_t)b.w << 32)};
        uint32_t env = 0;
        bool found = false;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t pc = (uint32_t)__popcll(w[k]);
            if (!found && rem < pc) {
                env = (t * 4u + k) * 64u + done_mask_pick(w[k], rem);
                found = true;
            }
            rem -= found ? 0u : pc;
        }
        return env;
    }
    for (uint32_t k = 0; k < chunk; ++k) {
        const uint32_t w = t * chunk + k;
        const uint64_t m = w < words ? mask[w] : 0ull;
        const uint32_t pc = (uint32_t)__popcll(m);
        if (rem < pc) return w * 64u + done_mask_pick(m, rem);
        rem -= pc;
    }
    return 0u;  // (unreachable for rem < the chunk's bits)
}
template <uint32_t T = 256>
__device__ inline uint32_t done_mask_nth(const uint64_t *mask, uint32_t words, const uint32_t *part, uint32_t i) {
    uint32_t lo = 0, hi = T;  // part[lo] <= i < part[hi]
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (part[mid] <= i) lo = mid;
        else hi = mid;
    }
    return done_mask_in_chunk<T>(mask, words, lo, i - part[lo]);
}
template <uint32_t T = 256>
__device__ inline uint32_t done_mask_find(const uint64_t *mask, uint32_t words, const DoneMaskShare &sh, uint32_t *part, uint32_t i) {
    if (sh.before <= i && i < sh.before + sh.bits) part[T + 1u + T / 64u] = done_mask_in_chunk<T>(mask, words, threadIdx.x, i - sh.before);  // exactly one thread
    __syncthreads();
    return part[T + 1u + T / 64u];
}

#define QG_COOP_LANES qg::plan::COOP_LANES  // lanes per env of the cooperative scramble (scramble_coop below; qgym_plan.hpp)

// Length of the compacted list of finished envs (compact_done), read by every thread of the LAST kernel that consumes it.  The last
// block WITH WORK to have read it zeroes the counter (and the ticket) for the next qg_vec_reset_done: counter[0] = length, counter[1] =
// blocks that have read it.  The number of blocks with work follows from the length alone (and the path the length selects), so every block
// computes it.  A block without work takes no ticket (256 tickets on one address cost the launch 3.4 us); it may find the counter already zeroed and then sees an empty list, which for it is
// the same thing.  An empty list needs neither tickets nor zeroing.  Call from all threads.
// `vblock`: this block's index among the blocks that consume the list (blockIdx.x, or less an offset when the kernel's grid starts with other work)
// `zero_other`: no ticket at all -- the handle keeps a list that nobody reads or appends to during this launch (qgym_api.cpp: the lists rotate),
// this launch zeroes THAT list's length for whoever appends to it next, and the list consumed here is left as it is.  All tickets go to one
// address, ~12 ns each, one after the other: with 512 workgroups the last answer comes 6 us after the first, the wave that took the ticket
// waits for it (the compiler compares the answer where the atomic is) and, at the next barrier, so does its workgroup.
// `count`: the length (the caller has loaded counter[0]); `threads`: how many threads of the launch have work with a list this long (one per
// env, 16 per env, a workgroup per env: the caller's path); returns `count`.
__device__ inline uint32_t list_count_take(uint32_t *counter, uint32_t count, uint64_t threads, uint32_t vblock, uint32_t *zero_other = nullptr) {
    if (zero_other && vblock == 0 && threadIdx.x == 0) {
        zero_other[0] = 0;
        zero_other[1] = 0;
    }
    if (count == 0) return 0;
    const uint32_t blocks = (uint32_t)((threads + blockDim.x - 1u) >> (31u - (uint32_t)__builtin_clz(blockDim.x)));  // blockDim.x: a power of two (a 64-bit
    if (vblock >= blocks) return count;  // (this block's threads all lie past the list)                               division costs ~150 scalar instructions)
    __syncthreads();
    if (!zero_other && threadIdx.x == blockDim.x - 1u) {
        if (atomicAdd(&counter[1], 1u) == blocks - 1u) {
            counter[0] = 0;
            counter[1] = 0;
        }
    }
    return count;
}

// ---------------------------------------------------------------------------------------------------
// Reset scramble on LDS-resident rows, shared by the TILE (uint32 rows) and TILE64 (uint64 rows) layouts.
// reset() = identity followed by `difficulty` random gates (clifford.rs:306-316): a dependent chain per env,
// so what matters is the latency of one gate.  The rows live in LDS, a gate is <= 2 row operations on
// disjoint slots (clifford.rs:111-133) read from the row-op table (InitArgs::rowops, make_op with slot
// indices), and the counter-RNG draws -- two splitmix64 rounds, ~300 cycles of 64-bit multiplies each --
// are issued ahead of the dependent LDS chain.
// ---------------------------------------------------------------------------------------------------
// qg_vec_reset_done: which of the three scrambles a list takes is decided in qgym_plan.hpp (list_reset_path)
using plan::coop_takes;
using plan::tree_takes;

template <typename W>
__device__ inline void lds_rowop(W *dst, W *src, uint32_t type) {
    const W va = *dst, vb = *src;
    const W swap = (W)0 - (W)(type == OP_SWAP);
    const W nd = (vb & swap) | ((va ^ vb) & ~swap), ns = (va & swap) | (vb & ~swap);
    if (type != OP_NONE) {
        *dst = nd;
        *src = ns;
    }
}

// one lane per env: `rows` = this wave's [slot][QG_WAVE] array, L = the lane's column in it
template <typename W>
__device__ inline void scramble_flat(W (*rows)[QG_WAVE], uint32_t L, const InitArgs &a, uint64_t env) {
    const uint64_t seed = init_seed(a);
    auto draw = [&](uint32_t t) -> uint32_t {
        if (t >= a.n_draws) return 0u;
        const int64_t act = a.actions ? (int64_t)a.actions[(uint64_t)t * a.B + env] : (int64_t)rng_action(seed, a.env_base + env, t, a.num_actions);
        return (act >= 0 && act < (int64_t)a.num_actions) ? a.rowops[act] : 0u;
    };
    auto gate = [&](uint32_t o) {
        lds_rowop<W>(&rows[o & 63u][L], &rows[(o >> 6) & 63u][L], (o >> 12) & 3u);
        lds_rowop<W>(&rows[(o >> 14) & 63u][L], &rows[(o >> 20) & 63u][L], (o >> 26) & 3u);
    };
    for (uint32_t t = 0; t < a.n_draws; t += 4) {
        const uint32_t o0 = draw(t), o1 = draw(t + 1), o2 = draw(t + 2), o3 = draw(t + 3);
        gate(o0);
        gate(o1);
        gate(o2);
        gate(o3);
    }
}

// 16 lanes per env (few finished envs: fills the otherwise idle SIMDs).  The draws of a 64-gate chunk are
// spread over the 16 lanes and parked in LDS as row-op words; two lanes then apply them in lockstep, one
// row operation each.  `lds`: R rows + 64 words per env, 64 / 16 envs per wave.  Returns the env's rows
// on the one lane per env that has to finish it (its index in `env`), nullptr on every other lane.
struct ListEntry {  // entry i of InitArgs::list (the default source of scramble_coop's envs)
    const uint32_t *list;
    __device__ uint32_t operator()(uint32_t i) const { return list[i]; }
};
// (`env` is set on every lane of a group that has an entry, also on the lanes that get nullptr back)
template <typename W, int R, typename Identity, typename EnvOf>
__device__ inline W *scramble_coop(const InitArgs &a, uint32_t count, void *lds, uint64_t &env, Identity identity, uint32_t vblock, EnvOf env_of,
                                   uint32_t block_threads = 0 /* the threads of a workgroup that take part (0: all of blockDim.x) */) {
    constexpr uint32_t S = QG_COOP_LANES, EPW = QG_WAVE / S, CH = 64, PER_ENV = R * sizeof(W) + CH * sizeof(uint32_t);
    const uint64_t item = ((uint64_t)vblock * (block_threads ? block_threads : blockDim.x) + threadIdx.x) / S;
    if (item >= count) return nullptr;  // whole lane groups leave together
    const uint32_t sl = threadIdx.x & (S - 1), w = threadIdx.x >> 6, g = (threadIdx.x & (QG_WAVE - 1)) / S;
    env = env_of((uint32_t)item);
    char *base = reinterpret_cast<char *>(lds) + (size_t)(w * EPW + g) * PER_ENV;
    W *rows = reinterpret_cast<W *>(base);
    uint32_t *ops = reinterpret_cast<uint32_t *>(base + R * sizeof(W));
    for (uint32_t k = sl; k < (uint32_t)R; k += S) rows[k] = identity(k);  // clifford.rs:307
    const uint64_t seed = init_seed(a);
    for (uint32_t c0 = 0; c0 < a.n_draws; c0 += CH) {
        const uint32_t len = a.n_draws - c0 < CH ? a.n_draws - c0 : CH;
        for (uint32_t k = sl; k < CH; k += S)  // the tail of the last chunk is padded with "no gate"
            ops[k] = k < len ? a.rowops[rng_action(seed, a.env_base + env, c0 + k, a.num_actions)] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (sl < 2) {
            const uint32_t sh = 14u * sl;
            auto rowop = [&](uint32_t o) {
                const uint32_t op = (o >> sh) & 0x3FFFu;
                lds_rowop<W>(&rows[op & 63u], &rows[(op >> 6) & 63u], op >> 12);
            };
            const uint32_t padded = (len + 3u) & ~3u;
            for (uint32_t k = 0; k < padded; k += 4) {  // four gate words fetched ahead of the dependent row reads
                const uint32_t o0 = ops[k], o1 = ops[k + 1], o2 = ops[k + 2], o3 = ops[k + 3];
                rowop(o0);
                rowop(o1);
                rowop(o2);
                rowop(o3);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return sl == 0 ? rows : nullptr;
}
template <typename W, int R, typename Identity>
__device__ inline W *scramble_coop(const InitArgs &a, uint32_t count, void *lds, uint64_t &env, Identity identity) {
    return scramble_coop<W, R>(a, count, lds, env, identity, blockIdx.x, ListEntry{a.list});
}
// One WORKGROUP (four waves) per env, the matrix held by COLUMNS, the gate sequence cut in EIGHT.
//
// Columns: lane j keeps column j in slot order (bit s of `col` = entry (slot s, column j); uint32 rows: R <= 32 slots, <= 32 columns).  A row
// operation -- row[dst] ^= row[src], or the two rows trade places (clifford.rs:64-82) -- is then bit arithmetic inside every lane's own
// word, as a parity test: t = col & test, col ^= -(popcount(t) & 1) & flip, with test = 1 << src, flip = 1 << dst for the xor and
// test = flip = both bits for the swap ("no gate": both zero).  The lane that draws a gate (two splitmix64 rounds, all lanes at once)
// decodes it into its four masks itself and parks them in LDS as one uint4; the serial loop reads them back (every lane of a half wave the
// same address: a broadcast), four gates ahead of the dependent chain, and decodes nothing: five dependent vector instructions per row
// operation, no scalar unit in the loop.  (Three v_readlane + scalar field extraction per gate measured 73 ns per gate, this form 38;
// tools/microbench_scramble.hip.)
// A 32-column matrix fills half a wave, so each wave runs TWO segments side by side -- lanes 0-31 and 32-63 read their own gate
// stream -- which halves the chain again (24 ns per gate and wave).  reset() is S = G_n ... G_1 S0 (S0 = the identity in slot order), a
// product of row-operation matrices, and matrix products associate: segment 0 applies its eighth of the gates to S0, the others to the
// R x R identity, giving P_k, and the eighths are multiplied back in a three-level tree through LDS: inside each wave P_upper P_lower, then
// (W3 W2) (W1 W0).  With columns on the lanes, column j of A B is A times column j of B: the xor of A's columns s over the set bits s of
// the lane's own word -- A's 32 column words read from LDS, 64 vector instructions (~0.3 us).  The dependent chain is n / 8 gates + three
// products (n = 256: 32 steps of ~45 ns instead of 64 gates of 73).
// The env needs ROWS (a row word per slot), the lanes hold columns -- so the tree computes the TRANSPOSE: S^T = S0^T G_1^T ... G_n^T, whose
// column j is row j of S.  A segment's leaf is then its gates in REVERSE order, each transposed (the xor trades test and flip, the swap is its
// own transpose), tree place p takes segment 7 - p, and the place that holds segment 0 multiplies S0^T on at the end of its chain (the xor of
// S0's rows over the set bits of the lane's word).  No row has to be gathered from 32 lanes' bits afterwards (32 ballots on the wave
// everything waits for), and S0 costs one half wave ~100 instructions instead of the long wave ~200.
// Returns true on the 64 lanes of wave 0, which finish the env together: lane s < R holds the row of slot s.
// `prod`: 4 x 32 words of LDS; `gates`: 4 x 64 uint4 of LDS (16-byte aligned); `table`: the row-operation table in LDS (the caller brings it
// in while the list length is still in flight), or null: read a.rowops; `env`: list[vblock], loaded by the caller.  blockDim.x must be 256.
constexpr uint32_t QG_TREE_THREADS = plan::TREE_THREADS;
constexpr uint32_t QG_TREE_TABLE_MAX = 1024;  // gatesets up to this many actions have their row-operation table in LDS
template <int R>
__device__ inline uint32_t gf2_cols_product(const uint32_t *a_cols, uint32_t b) {  // this lane's column of A B; a_cols[s] = column s of A
    uint32_t acc = 0;
#pragma unroll 8
    for (int sl = 0; sl < R; ++sl) acc ^= a_cols[sl] & (uint32_t)__builtin_amdgcn_sbfe((int32_t)b, (uint32_t)sl, 1u);
    return acc;
}
// The same with BOTH halves of the wave at work (a 32-column matrix fills half a wave; an instruction costs the same with 32 lanes idle):
// lanes hl and hl + 32 hold the same word b, half h sums the slots [h R/2, (h + 1) R/2), and the halves' partial columns are xor-ed across
// with one v_permlane32_swap (gfx950): half the instructions of the one-half form.  Returns the column on both halves.
__device__ inline uint32_t both_halves_xor(uint32_t v) {  // v of lane hl ^ v of lane hl + 32, on both
    const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // {lower half's value on both halves, upper half's on both}
    return sw[0] ^ sw[1];
}
__device__ inline uint32_t lower_half_on_both(uint32_t v) { return __builtin_amdgcn_permlane32_swap(v, v, false, false)[0]; }
// a ^ (b & c) as ONE v_bitop3_b32 (gfx950; truth table from a = 0xF0, b = 0xCC, c = 0xAA).  The scramble chains and the GF(2) products below are
// bound by their instruction count (a lone wave issues one instruction every ~5 cycles), and this pair is most of what they do
__device__ inline uint32_t xor_and(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x78); }
template <int R>
__device__ inline uint32_t gf2_cols_product_halves(const uint32_t *a_cols, uint32_t b, uint32_t half) {
    constexpr int H = (R + 1) / 2;
    const uint32_t *ac = a_cols + H * half;
    const uint32_t bs = b >> (H * half);
    uint32_t acc = 0;
#pragma unroll 8
    for (int i = 0; i < H; ++i) acc = xor_and(acc, H + i < R || !half ? ac[i] : 0u, (uint32_t)__builtin_amdgcn_sbfe((int32_t)bs, (uint32_t)i, 1u));
    return both_halves_xor(acc);
}
// one row operation in the parity form (see above)
__device__ inline void rowop_parity(uint32_t &col, uint32_t test, uint32_t flip) {
    col = xor_and(col, (uint32_t)__builtin_amdgcn_sbfe((int32_t)__builtin_popcount(col & test), 0u, 1u), flip);
}
// the four masks {test0, flip0, test1, flip1} of a gate word (two row operations, make_op with slot indices, 14 bits each)
// (`transposed`: the masks of the row operation's transpose -- row[src] ^= row[dst]; the swap is symmetric)
__device__ inline uint4 rowop_masks(uint32_t o, bool transposed = false) {
    auto half = [transposed](uint32_t op, uint32_t &test, uint32_t &flip) {
        const uint32_t type = (op >> 12) & 3u, bd = 1u << ((op >> (transposed ? 6 : 0)) & 63u), bs = 1u << ((op >> (transposed ? 0 : 6)) & 63u);
        test = type == OP_NONE ? 0u : (type == OP_SWAP ? bs | bd : bs);
        flip = type == OP_NONE ? 0u : (type == OP_SWAP ? bs | bd : bd);
    };
    uint4 r;
    half(o & 0x3FFFu, r.x, r.y);
    half(o >> 14, r.z, r.w);
    return r;
}
struct NoMid { __device__ void operator()() const {} };
// `mid`: called by wave 0 between the chain and the products (the caller's loads for what follows the tree are issued there: by then their
// addresses have arrived, and the products' time hides them)
template <int R, typename Identity, typename Mid = NoMid>
__device__ inline bool scramble_tree(const InitArgs &a, uint64_t env, uint32_t &row_out, uint32_t (*prod)[32], uint4 (*gates)[QG_WAVE],
                                     const uint32_t *table, Identity identity, Mid mid = Mid()) {
    static_assert(R <= 32, "one uint32 of slots per column");
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), w = threadIdx.x >> 6, half = lane >> 5, hl = lane & 31u;
    if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 1);  // the env is known
    // (`env` comes in from the caller: list[item], requested before the list's length was known)
    // tree place p = 2 w + half holds segment k = 7 - p: the gates [k seg, (k + 1) seg), last one first, transposed
    const uint32_t seg = (a.n_draws + 7u) / 8u, k = 7u - (2u * w + half);
    const uint32_t t0 = k * seg < a.n_draws ? k * seg : a.n_draws, t1 = (t0 + seg < a.n_draws) ? t0 + seg : a.n_draws, len = t1 - t0;
    uint32_t col = hl < (uint32_t)R ? 1u << hl : 0u;
    const uint64_t seed = init_seed(a);
    uint4 *mine = gates[w];
    for (uint32_t c0 = 0; c0 < seg; c0 += 32u) {  // 32 gates per half wave and pass (seg <= 32 up to 256 draws: one pass)
        const uint32_t u = c0 + hl;  // the u-th gate this place applies: the segment's gate len - 1 - u
        uint32_t o = 0u;  // past the segment's end: "no gate" (all four masks zero)
        if (u < len) {
            const uint32_t act = rng_action(seed, a.env_base + env, t0 + (len - 1u - u), a.num_actions);
            o = table ? table[act] : a.rowops[act];
        }
        __builtin_amdgcn_wave_barrier();  // (the previous pass has read its masks)
        mine[lane] = rowop_masks(o, true);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t left = seg - c0, steps = left < 32u ? left : 32u;  // wave-uniform (both halves walk `steps` gates; the tail is "no gate")
        const uint4 *p = mine + 32u * half;
        uint4 g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = p[q];
        for (uint32_t kk = 0; kk < steps; kk += 4u) {
            uint4 nx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[q] = p[(kk + 4u + q) & 31u];  // the next four gates fly while these four are applied
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // (entries past `steps` in the last group of four are past the segment's end: zero masks)
                rowop_parity(col, g[q].x, g[q].y);
                rowop_parity(col, g[q].z, g[q].w);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = nx[q];
        }
    }
    if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 2);  // wave 0's chain
    if (k == 0) {  // S0^T on the left (clifford.rs:307): the xor of S0's rows over the set bits of the word
        uint32_t acc = 0;
#pragma unroll 8
        for (int sl = 0; sl < R; ++sl) acc ^= identity((uint32_t)sl) & (uint32_t)__builtin_amdgcn_sbfe((int32_t)col, (uint32_t)sl, 1u);
        col = acc;
    }
    // P_upper P_lower inside the wave, then (W3 W2) (W1 W0): publish, multiply
    if (w == 0) mid();
    // (every product with both halves of the wave: gf2_cols_product_halves; from the first one on both halves hold the same column)
    if (half) prod[w][hl] = col;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    col = gf2_cols_product_halves<R>(prod[w], lower_half_on_both(col), half);
    __builtin_amdgcn_wave_barrier();  // (prod[w] is this wave's own: its reads above come before the write below)
    if ((w & 1u) && !half) prod[w][hl] = col;
    __syncthreads();
    if (!(w & 1u)) col = gf2_cols_product_halves<R>(prod[w + 1u], col, half);
    if (w == 2u && !half) prod[2][hl] = col;
    __syncthreads();
    if (w != 0) return false;
    col = gf2_cols_product_halves<R>(prod[2], col, half);
    if (threadIdx.x == 0) phase_stamp(a.kclk, a.kclk_waves, 3);  // the products
    // (the transpose's column hl IS the row of slot hl: lanes 0 .. R - 1 of the wave hold the env's rows, the upper half nothing)
    const uint32_t my_row = half ? 0u : col;
    row_out = my_row;
    return true;  // on all 64 lanes of wave 0
}

// The same for 64-bit rows (TILE64: CliffordEnv 16 < N <= 32, LinearFunctionEnv 32 < N <= 64): R <= 64 slots in a uint64 per column, all
// 64 lanes hold a column.  `prod`: 4 x 64 uint64 of LDS.  Called from the tree workgroups of q64_reset_done_kernel / q64_reset_step_kernel (q64_reset_tree_body), one workgroup per listed env.
__device__ inline int64_t bit_mask64(uint64_t v, uint32_t off) { return (int64_t)(v << (63u - off)) >> 63; }  // bit `off` as 0 / -1
template <int R>
__device__ inline uint64_t gf2_cols_product64(const uint64_t *a_cols, uint64_t b) {
    const uint32_t blo = (uint32_t)b, bhi = (uint32_t)(b >> 32);
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int sl = 0; sl < R; ++sl) {  // one v_bfe_i32 per slot (a 64-bit shift pair in round 4), then the column's halves
        const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int32_t)(sl < 32 ? blo : bhi), (uint32_t)(sl & 31), 1u);
        const uint64_t c = a_cols[sl];
        lo = xor_and(lo, (uint32_t)c, m);
        hi = xor_and(hi, (uint32_t)(c >> 32), m);
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// Returns true on ALL lanes of wave 0, whose `col_out` is then the env's ROW of slot `lane` (the tree runs on the transpose, see scramble_tree:
// a row per lane is what q64_reset_tree_body's finish wants -- 64 row words in one lane's registers cost 450 registers and scratch).
// 64-bit counterparts of rowop_parity / rowop_masks
__device__ inline void rowop_parity64(uint64_t &col, uint64_t test, uint64_t flip) {
    col ^= (uint64_t)(0ll - (long long)(__builtin_popcountll(col & test) & 1)) & flip;
}
struct RowopMasks64 {
    uint64_t t0, f0, t1, f1;
};
__device__ inline RowopMasks64 rowop_masks64(uint32_t o, bool transposed = false) {
    auto half = [transposed](uint32_t op, uint64_t &test, uint64_t &flip) {
        const uint32_t type = (op >> 12) & 3u;
        const uint64_t bd = 1ull << ((op >> (transposed ? 6 : 0)) & 63u), bs = 1ull << ((op >> (transposed ? 0 : 6)) & 63u);
        test = type == OP_NONE ? 0ull : (type == OP_SWAP ? bs | bd : bs);
        flip = type == OP_NONE ? 0ull : (type == OP_SWAP ? bs | bd : bd);
    };
    RowopMasks64 r;
    half(o & 0x3FFFu, r.t0, r.f0);
    half(o >> 14, r.t1, r.f1);
    return r;
}
// `gates`: WAVES x 64 RowopMasks64 of LDS (8 KiB at four waves): the drawing lane decodes its gate into the four parity-test masks (64 gates decoded
// at once), the serial loop reads them back as broadcasts.  A 64-column matrix fills the wave, so a wave runs one segment.  What bounds the chain is
// the INSTRUCTION COUNT per gate, not a latency: at 512 trees per launch every SIMD holds two of these waves, each issues one instruction every
// ~5 cycles, and the vector unit takes 4 cycles per instruction whatever wave it comes from.  Round 4's loop spent 22 vector instructions a gate on
// 64-bit C++ (95 ns a gate, 6.1 of the tree launch's 13.2 us at 256 gates); on 32-bit halves a parity is and, and, bcnt, bcnt, bfe and two
// v_bitop3 (a ^ (b & c)): 14 a gate.  Measured and dropped (EXPERIMENTS.md round 5): decoding on the scalar unit from a v_readlane of the gate word (no
// LDS; +4 us: ~50 scalar instructions a gate issue at the same 5 cycles each), eight waves of 32 gates (+4 us with it: twice the waves per SIMD, one more
// level of products).
// `op_of(t)`: the row-operation word of gate t (two make_op halves, slot indices), t < n_gates; call from all threads of the workgroup.
__device__ inline void rowop_parity64_halves(uint32_t &lo, uint32_t &hi, uint32_t tlo, uint32_t thi, uint32_t flo, uint32_t fhi) {
    const uint32_t c = (uint32_t)__builtin_popcount(hi & thi) + (uint32_t)__builtin_popcount(lo & tlo);  // v_bcnt_u32_b32, accumulating
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int32_t)c, 0u, 1u);  // parity as 0 / -1
    lo = xor_and(lo, m, flo);
    hi = xor_and(hi, m, fhi);
}
// `ready` (LDS, WAVES words, zero before the first call) and `seq` (a number that grows by log2(WAVES) from one call of the workgroup to the next, starting at 0):
// the products' levels hand over through these words instead of workgroup barriers -- the publishing wave stores seq + level + 1 behind its column, the
// multiplying wave polls for it -- so that a workgroup with MORE waves than the scramble's (ptile_reset_tree_kernel: a fifth one generates labels
// meanwhile) does not have to bring them to every level's barrier.  Null: barriers.  Either way the caller separates two calls by a barrier of its own.
template <int R, int WAVES = 4, typename Identity, typename OpOf>
__device__ inline bool scramble_tree64_ops(uint32_t n_gates, uint64_t &col_out, uint64_t (*prod)[64], RowopMasks64 (*gates)[QG_WAVE], Identity identity, OpOf op_of,
                                           uint32_t *ready = nullptr, uint32_t seq = 0) {
    static_assert(R <= 64, "one uint64 of slots per column");
    static_assert(WAVES == 4 || WAVES == 8 || WAVES == 16, "the workgroup's waves: a power of two");
    const uint32_t lane = threadIdx.x & (QG_WAVE - 1), w = threadIdx.x >> 6;
    // the transpose, as scramble_tree: tree place w holds segment k = WAVES - 1 - w, its gates last one first and transposed; segment 0's place
    // multiplies S0^T on at the end.  Lane s of wave 0 ends with the row of slot s.
    const uint32_t seg = (n_gates + (uint32_t)WAVES - 1u) / (uint32_t)WAVES, k = (uint32_t)WAVES - 1u - w;
    const uint32_t t0 = k * seg < n_gates ? k * seg : n_gates, t1 = (t0 + seg < n_gates) ? t0 + seg : n_gates, len = t1 - t0;
    uint64_t col = lane < (uint32_t)R ? 1ull << lane : 0ull;
    uint32_t lo = (uint32_t)col, hi = (uint32_t)(col >> 32);
    const uint4 *mine = reinterpret_cast<const uint4 *>(gates[w]);
    for (uint32_t c0 = 0; c0 < len; c0 += QG_WAVE) {  // 64 gates per pass
        const uint32_t u = c0 + lane;  // the u-th gate this place applies: the segment's gate len - 1 - u
#if defined(QG_ABLATE_TREE64) && (QG_ABLATE_TREE64 & 2)
        const uint32_t o = 0u;
#else
        const uint32_t o = u < len ? op_of(t0 + (len - 1u - u)) : 0u;  // past the end: "no gate"
#endif
        __builtin_amdgcn_wave_barrier();  // (the previous pass has read its masks)
        gates[w][lane] = rowop_masks64(o, true);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#if defined(QG_ABLATE_TREE64) && (QG_ABLATE_TREE64 & 1)
        const uint32_t steps = 0;
#else
        const uint32_t left = len - c0;
        const uint32_t steps = (uint32_t)__builtin_amdgcn_readfirstlane((int)(left < QG_WAVE ? left : QG_WAVE));  // wave-uniform: a scalar loop
#endif
        for (uint32_t kk = 0; kk < steps; kk += 4u) {  // (entries past `steps` in the last group of four are past the segment's end: zero masks)
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                const uint4 g0 = mine[2u * ((kk + q) & 63u)], g1 = mine[2u * ((kk + q) & 63u) + 1u];  // {test, flip} of the gate's first and second row operation
                rowop_parity64_halves(lo, hi, g0.x, g0.y, g0.z, g0.w);
                rowop_parity64_halves(lo, hi, g1.x, g1.y, g1.z, g1.w);
            }
        }
    }
    col = (uint64_t)lo | ((uint64_t)hi << 32);
    if (k == 0) {  // S0^T on the left (clifford.rs:307)
        uint64_t acc = 0;
#pragma unroll 8
        for (int sl = 0; sl < R; ++sl) acc ^= identity((uint32_t)sl) & (uint64_t)bit_mask64(col, (uint32_t)sl);
        col = acc;
    }
#if defined(QG_ABLATE_TREE64) && (QG_ABLATE_TREE64 & 4)
    if (w != 0) return false;
    col_out = col;
    return true;
#endif
    // (W1 W0), (W3 W2), ...; then pairs of those; ...: wave w + s publishes, wave w multiplies it on
#pragma unroll
    for (uint32_t s = 1; s < (uint32_t)WAVES; s <<= 1) {
        const uint32_t at = w & (2u * s - 1u);
        if (at == s) prod[w][lane] = col;  // (at == s: the wave's lower bits are zero, i.e. it multiplied at every level before)
        if (ready) {
            seq += 1u;
            if (at == s) {  // the column is in LDS before the word that says so (one wave's LDS operations complete in order; the fence keeps the compiler to it)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_store(&ready[w], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (at == 0u) {
                while (__hip_atomic_load(&ready[w + s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq) __builtin_amdgcn_s_sleep(1);
            }
        } else {
            __syncthreads();  // (every wave of the workgroup, at every level)
        }
        if (at == 0u) col = gf2_cols_product64<R>(prod[w + s], col);
    }
    if (w != 0) return false;
    col_out = col;
    return true;
}
template <int R, int WAVES, typename Identity>
__device__ inline bool scramble_tree64(const InitArgs &a, uint64_t env, uint64_t &col_out, uint64_t (*prod)[64], RowopMasks64 (*gates)[QG_WAVE],
                                       const uint32_t *rowops /* InitArgs::rowops, or the caller's copy of it in LDS */, Identity identity) {
    const uint64_t seed = init_seed(a), e = a.env_base + env;
    const uint32_t num_actions = a.num_actions;
    return scramble_tree64_ops<R, WAVES>(a.n_draws, col_out, prod, gates, identity,
                                         [=](uint32_t t) -> uint32_t { return rowops[rng_action(seed, e, t, num_actions)]; });
}

template <typename W, int R>
constexpr size_t scramble_coop_lds_bytes(int waves) { return (size_t)waves * (QG_WAVE / QG_COOP_LANES) * (R * sizeof(W) + 64 * sizeof(uint32_t)); }

// Solution log (StepArgs::sol): entry `slot` of env e lives at sol[slot * B + e] -- step-major, so the one entry every env appends
// per step is a coalesced store (the env-major form costs a 64-byte line per 4-byte entry: +7.5 us per step at 262 144 envs).
__device__ inline uint32_t &sol_at(const StepArgs &a, uint64_t env, uint32_t slot) { return a.sol[(uint64_t)slot * a.B + env]; }
// a window of one env's log, indexable like an array (PauliEnv logs several entries per step)
struct SolLog {
    uint32_t *p;
    uint64_t stride;
    __device__ uint32_t &operator[](uint32_t i) const { return p[(uint64_t)i * stride]; }
    __device__ explicit operator bool() const { return p != nullptr; }
};

// Solution-log word of an action (clifford.rs:334-340 pushes the action verbatim, valid or not).
// PauliEnv: 32 bits (bit 31 is the reference's own ROTATION_MARKER, pauli.rs:698-716), so anything a `usize` action could hold beyond
// 2^32 - 2 -- and a negative int64, which `as usize` turns into 2^64 - 1 -- saturates to 0xFFFFFFFF (read back as UINT64_MAX).
__device__ inline uint32_t sol_word(int64_t act) { return (act < 0 || act > 0xFFFFFFFEll) ? 0xFFFFFFFFu : (uint32_t)act; }
// Clifford / LinearFunction / Permutation: entries are kept in the order they were pushed, slot = solution.len() + solution_inv.len(), with
// bit 31 = "pushed to solution_inv" (clifford.rs:335-339); envs that were reset together then append to the SAME slot, which makes the
// store coalesced whatever their coin histories were (separate front / back lists drift apart per env: one 128-byte line per 4-byte
// entry, +4.4 us per step at 262 144 envs).  The action keeps 31 bits: beyond 2^31 - 2 it saturates to 0x7FFFFFFF (read back as UINT64_MAX).
__device__ inline uint32_t sol_word_framed(int64_t act, bool inverted_frame) {
    const uint32_t w = (act < 0 || act > 0x7FFFFFFEll) ? 0x7FFFFFFFu : (uint32_t)act;
    return w | ((uint32_t)inverted_frame << 31);
}

// Dense {0,1} elements from packed bits: one 16-byte chunk = 16 / ES elements of ES bytes each.
// `one`: the bit pattern of 1 in the output dtype (int8 1, bf16 0x3F80, f16 0x3C00, f32 0x3F800000)
template <int ES>
__device__ inline uint4 expand_chunk(uint32_t bits, uint32_t one) {
    uint32_t w[4];
    if constexpr (ES == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t nb = (bits >> (4 * k)) & 0xFu;
            w[k] = ((nb & 1u) | ((nb & 2u) << 7) | ((nb & 4u) << 14) | ((nb & 8u) << 21)) * one;
        }
    } else if constexpr (ES == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = (bits >> (2 * k)) & 1u, hi = (bits >> (2 * k + 1)) & 1u;
            w[k] = (lo * one) | ((hi * one) << 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = ((bits >> k) & 1u) * one;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16 bits -> 16 int8 {0,1} entries (adapters.py:50-54's dense observation).  A nibble times 0x204081 puts copies of it at bits 0, 7, 14
// and 21 (they do not overlap: no carries), so the mask keeps bit k of the nibble in byte k: three full-rate instructions (bfe,
// mul_u32_u24, and) per output word.
__device__ inline uint4 expand16_i8(uint32_t bits) {
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = __umul24((bits >> (4 * k)) & 0xFu, 0x204081u) & 0x01010101u;
    return make_uint4(w[0], w[1], w[2], w[3]);
}
// row `row` of env `env` in the resident dense int8 observation [B][D][D], D = 16 * D16: 16 * D16 bytes at 16-byte alignment
template <int D16>
__device__ inline void dense_row_store(int8_t *dense, uint64_t env, uint32_t row, uint32_t w) {
    uint4 *p = reinterpret_cast<uint4 *>(dense + (env * (16u * D16) + row) * (16u * D16));
#pragma unroll
    for (int k = 0; k < D16; ++k) p[k] = expand16_i8(w >> (16 * k));
}

struct LayerRec {
    int32_t *p;
    __device__ int32_t &operator[](uint32_t i) const { return p[(size_t)i * 64]; }
};
__device__ inline LayerRec layer_rec(int32_t *layers, uint64_t env, uint32_t len) {
    return LayerRec{layers + (env >> 6) * 64 * len + (env & 63)};
}
struct LayerDelta {
    int dc, dlc, dl, dg;
};
// The tracker entries one gate can touch, held in registers: every micro-op of a gate (metrics.rs:64-81: CX -> cx(c, t); SWAP -> three cx on
// the same pair; CZ -> single(t), cx(c, t), single(t); one-qubit gates -> single(q)) reads and writes the layer indices of its own <= 2
// qubits and the two running maxima only, so the record is gathered once, updated in registers and scattered once -- no store -> load
// round trips through memory between the micro-ops (that chain cost 5 us per step of 65 536 envs).
struct LayerRegs {
    int32_t g[2], c[2];  // last_gates / last_cxs of q0 (index 0) and q1 (index 1)
    int32_t nl, nlc;     // |layers|, |cnot_layers| as running maxima
};
__device__ inline void layers_single(LayerRegs &r, uint32_t N, uint32_t t, uint32_t ti, LayerDelta &d) {  // ti: t is q0 (0) or q1 (1)
    if (t >= N) return;  // metrics.rs:84-86
    d.dg += 1;
    const int32_t gl = r.g[ti] + 1;
    r.g[ti] = gl;
    if (gl + 1 > r.nl) {
        d.dl += gl + 1 - r.nl;
        r.nl = gl + 1;
    }
}
__device__ inline void layers_cx(LayerRegs &r, uint32_t N, uint32_t c, uint32_t t, LayerDelta &d) {  // the pair is always {q0, q1}
    if (c == t || c >= N || t >= N) return;  // metrics.rs:98-103
    d.dc += 1;
    d.dg += 1;
    const int32_t gl = (r.g[0] > r.g[1] ? r.g[0] : r.g[1]) + 1;
    r.g[0] = r.g[1] = gl;
    if (gl + 1 > r.nl) {
        d.dl += gl + 1 - r.nl;
        r.nl = gl + 1;
    }
    const int32_t cl = (r.c[0] > r.c[1] ? r.c[0] : r.c[1]) + 1;
    r.c[0] = r.c[1] = cl;
    if (cl + 1 > r.nlc) {
        d.dlc += cl + 1 - r.nlc;
        r.nlc = cl + 1;
    }
}
// metrics.rs:64-81 + 135-146: apply the gate to the tracker, return the f32 penalty.
// (This translation unit is compiled with -ffp-contract=off: no fused multiply-add.)
// `lay` = this env's record: last_gates[N], last_cxs[N], n_layers, n_layers_cnots (int32 each; last_* start at -1), held like the state:
// tiles of 64 envs, entry-major, so the 64 lanes of a wave touch at most 2N + 2 rows of 256 B instead of 64 separate records.
// |layers| == max(last_gates)+1 and |cnot_layers| == max(last_cxs)+1 because every inserted layer index is one more than an index
// already present (or 0): running maxima instead of the reference's HashSets (metrics.rs:83-123).
// Split in two so that a step kernel can issue the tracker's loads together with its state loads (one memory round trip, not two):
// layers_begin gathers, layers_commit updates in registers, scatters what changed and returns the penalty.
struct LayerTxn {
    LayerRec lay;
    LayerRegs r;
    uint32_t N, desc;
};
__device__ inline LayerTxn layers_begin(const LayerRec lay, uint32_t N, uint32_t desc) {
    const uint32_t kind = desc & 0xFFu, q0 = (desc >> 8) & 0xFFu;
    const uint32_t q1 = kind >= QG_CX ? (desc >> 16) & 0xFFu : q0;
    const uint32_t i0 = q0 < N ? q0 : 0u, i1 = q1 < N ? q1 : 0u;  // out-of-range qubits are never applied (the guards above); keep the loads in bounds
    LayerTxn t;
    t.lay = lay; t.N = N; t.desc = desc;
    t.r.g[0] = lay[i0]; t.r.g[1] = lay[i1];
    t.r.c[0] = lay[N + i0]; t.r.c[1] = lay[N + i1];
    t.r.nl = lay[2 * N]; t.r.nlc = lay[2 * N + 1];
    return t;
}
__device__ inline float layers_commit(const LayerTxn &t, const float w[4]) {
    const LayerRec lay = t.lay;
    const uint32_t N = t.N, kind = t.desc & 0xFFu, q0 = (t.desc >> 8) & 0xFFu;
    const bool two = kind >= QG_CX;
    const uint32_t q1 = two ? (t.desc >> 16) & 0xFFu : q0;
    const LayerRegs r0 = t.r;
    LayerRegs r = t.r;
    LayerDelta d = {0, 0, 0, 0};
    switch (kind) {
    case QG_CX: layers_cx(r, N, q0, q1, d); break;
    case QG_SWAP:
        layers_cx(r, N, q0, q1, d);
        layers_cx(r, N, q1, q0, d);
        layers_cx(r, N, q0, q1, d);
        break;
    case QG_CZ:  // with q0 == q1 only the two single(t) act, on q1's registers
        layers_single(r, N, q1, 1, d);
        layers_cx(r, N, q0, q1, d);
        layers_single(r, N, q1, 1, d);
        break;
    default: layers_single(r, N, q0, 0, d); break;
    }
    // q0's entries first, q1's second: when both are the same qubit (a two-qubit gate on one qubit) q1's registers hold the truth
    if (r.g[0] != r0.g[0] && q0 < N && q0 != q1) lay[q0] = r.g[0];
    if (r.c[0] != r0.c[0] && q0 < N && q0 != q1) lay[N + q0] = r.c[0];
    if (!two && r.g[0] != r0.g[0] && q0 < N) lay[q0] = r.g[0];
    if (two && r.g[1] != r0.g[1] && q1 < N) lay[q1] = r.g[1];
    if (two && r.c[1] != r0.c[1] && q1 < N) lay[N + q1] = r.c[1];
    if (r.nl != r0.nl) lay[2 * N] = r.nl;
    if (r.nlc != r0.nlc) lay[2 * N + 1] = r.nlc;
    const float t0 = w[0] * (float)d.dc;
    const float t1 = w[1] * (float)d.dlc;
    const float t2 = w[2] * (float)d.dl;
    const float t3 = w[3] * (float)d.dg;
    float s = t0 + t1;
    s = s + t2;
    s = s + t3;
    return s;
}
__device__ inline float layers_penalty(const LayerRec lay, uint32_t N, uint32_t desc, const float w[4]) {
    return layers_commit(layers_begin(lay, N, desc), w);
}

}  // namespace qg
