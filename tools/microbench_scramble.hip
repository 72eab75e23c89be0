// microbench_scramble.hip -- what one row operation of the reset scramble costs when a wave holds ONE env's matrix by columns
// (device_common.hpp scramble_wave): 512 waves (one per SIMD on half the chip), 256 gates = 512 row operations each, variants of
// how the gate reaches the lanes.  Development tool.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_scramble.hip -o /tmp/mb_scramble
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ inline void rowop(uint32_t &col, uint32_t src, uint32_t dst, int32_t swap, uint32_t m) {
    const int32_t bs = __builtin_amdgcn_sbfe((int32_t)col, src, 1u), bd = __builtin_amdgcn_sbfe((int32_t)col, dst, 1u);
    col ^= (uint32_t)(bs ^ (bd & swap)) & m;
}
__device__ inline uint32_t mask_of(uint32_t op) {
    const uint32_t type = (op >> 12) & 3u, dst = op & 63u, src = (op >> 6) & 63u;
    return ((uint32_t)(type != 0) << dst) | ((uint32_t)(type == 2) << src);
}

// MODE 0: three readlanes per gate (op word + two masks), fields decoded on the scalar unit  (scramble_wave as first written)
// MODE 1: one readlane per gate, masks rebuilt on the scalar unit
// MODE 2: no readlane: the gate words come from LDS (every lane reads the same address), everything per lane on the vector unit
// MODE 3: floor -- the same vector chain with constant operands (no gate fetch at all)
// MODE 4: scalar loads: the gate words sit in global memory, s_load_dwordx4 brings four gates at a time
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t *ops, uint32_t *out, int n) {
    __shared__ uint32_t lds[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t *my = ops + (size_t)blockIdx.x * 256;
    uint32_t col = 1u << (lane & 31u);
    uint32_t o[4], m0[4], m1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = my[j * 64 + lane];
        m0[j] = mask_of(o[j] & 0x3FFFu);
        m1[j] = mask_of(o[j] >> 14);
        lds[j * 64 + lane] = o[j];
    }
    __syncthreads();
    if (MODE == 4) {
        const uint32_t *sp = (const uint32_t *)__builtin_amdgcn_readfirstlane((int)(uintptr_t)my) ;  // placeholder: see below
        (void)sp;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        for (int kk = 0; kk < 64 && j * 64 + kk < n; ++kk) {
            if (MODE == 0) {
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)o[j], kk);
                const uint32_t ga = (uint32_t)__builtin_amdgcn_readlane((int)m0[j], kk), gb = (uint32_t)__builtin_amdgcn_readlane((int)m1[j], kk);
                rowop(col, (g >> 6) & 63u, g & 63u, __builtin_amdgcn_sbfe((int32_t)g, 13u, 1u), ga);
                rowop(col, (g >> 20) & 63u, (g >> 14) & 63u, __builtin_amdgcn_sbfe((int32_t)g, 27u, 1u), gb);
            } else if (MODE == 1) {
                const uint32_t g = (uint32_t)__builtin_amdgcn_readlane((int)o[j], kk);
                rowop(col, (g >> 6) & 63u, g & 63u, __builtin_amdgcn_sbfe((int32_t)g, 13u, 1u), mask_of(g & 0x3FFFu));
                rowop(col, (g >> 20) & 63u, (g >> 14) & 63u, __builtin_amdgcn_sbfe((int32_t)g, 27u, 1u), mask_of(g >> 14));
            } else if (MODE == 2) {
                const uint32_t g = lds[j * 64 + kk];
                rowop(col, (g >> 6) & 63u, g & 63u, __builtin_amdgcn_sbfe((int32_t)g, 13u, 1u), mask_of(g & 0x3FFFu));
                rowop(col, (g >> 20) & 63u, (g >> 14) & 63u, __builtin_amdgcn_sbfe((int32_t)g, 27u, 1u), mask_of(g >> 14));
            } else if (MODE == 3) {
                rowop(col, 3u, 7u, 0, 1u << 7);
                rowop(col, 19u, 23u, -1, (1u << 23) | (1u << 19));
            } else {
                const uint32_t g = __builtin_nontemporal_load(my + j * 64 + kk);  // uniform address: the compiler may pick s_load
                rowop(col, (g >> 6) & 63u, g & 63u, __builtin_amdgcn_sbfe((int32_t)g, 13u, 1u), mask_of(g & 0x3FFFu));
                rowop(col, (g >> 20) & 63u, (g >> 14) & 63u, __builtin_amdgcn_sbfe((int32_t)g, 27u, 1u), mask_of(g >> 14));
            }
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = col;
}

// MODE 5..8: the row operation as a parity test -- t = col & test, col ^= -(popcount(t) & 1) & flip; xor: test = 1 << src, flip = 1 << dst;
// swap: test = flip = both bits -- so a gate is four masks the drawing lane decodes itself, and nothing is decoded in the serial loop:
//   5  four readlanes per gate
//   7  the four masks of a gate parked in LDS as one uint4, read back by every lane (broadcast), four gates ahead of the chain
//   8  the same with TWO gate streams per wave: lanes 0-31 and 32-63 each hold a 32-column matrix and read their own stream (n / 2 steps)
__device__ inline void rowop_p(uint32_t &col, uint32_t test, uint32_t flip) {
    const uint32_t t = col & test;
    col ^= (uint32_t)__builtin_amdgcn_sbfe((int32_t)__builtin_popcount(t), 0u, 1u) & flip;
}
__device__ inline uint4 decode_p(uint32_t o) {
    auto half = [](uint32_t op, uint32_t &test, uint32_t &flip) {
        const uint32_t type = (op >> 12) & 3u, dst = op & 63u, src = (op >> 6) & 63u;
        const uint32_t bs = 1u << src, bd = 1u << dst;
        test = type == 0 ? 0u : (type == 2 ? bs | bd : bs);
        flip = type == 0 ? 0u : (type == 2 ? bs | bd : bd);
    };
    uint4 r;
    half(o & 0x3FFFu, r.x, r.y);
    half(o >> 14, r.z, r.w);
    return r;
}
template <int MODE>
__global__ __launch_bounds__(64) void kp(const uint32_t *ops, uint32_t *out, int n) {
    __shared__ uint4 lds[256];
    const uint32_t lane = threadIdx.x;
    const uint32_t *my = ops + (size_t)blockIdx.x * 256;
    uint32_t col = 1u << (lane & 31u);
    uint4 d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        d[j] = decode_p(my[j * 64 + lane]);
        lds[j * 64 + lane] = d[j];
    }
    __syncthreads();
    if (MODE == 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            for (int kk = 0; kk < 64 && j * 64 + kk < n; ++kk) {
                const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)d[j].x, kk), b = (uint32_t)__builtin_amdgcn_readlane((int)d[j].y, kk);
                const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)d[j].z, kk), e = (uint32_t)__builtin_amdgcn_readlane((int)d[j].w, kk);
                rowop_p(col, a, b);
                rowop_p(col, c, e);
            }
    } else {
        // MODE 8: the upper half wave walks the second half of the stream
        const int steps = MODE == 8 ? n / 2 : n;
        const uint4 *p = lds + ((MODE == 8 && lane >= 32u) ? steps : 0);
        uint4 g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = p[q];
        for (int kk = 0; kk < steps; kk += 4) {
            uint4 nx[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nx[q] = p[(kk + 4 + q) & 255];  // the next four gates fly while these four are applied
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rowop_p(col, g[q].x, g[q].y);
                rowop_p(col, g[q].z, g[q].w);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = nx[q];
        }
    }
    out[(size_t)blockIdx.x * 64 + lane] = col;
}
template <int MODE>
static void runp(const char *name, const uint32_t *ops, uint32_t *out, int waves, int n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kp<MODE>, dim3(waves), dim3(64), 0, 0, ops, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kp<MODE>, dim3(waves), dim3(64), 0, 0, ops, out, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("waves %5d gates %3d mode %d %-52s %8.2f us per launch  %6.1f ns per gate\n", waves, n, MODE, name, ms * 1e3 / 20, ms * 1e6 / 20 / n);
}

template <int MODE>
static void run(const char *name, const uint32_t *ops, uint32_t *out, int waves, int n) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, ops, out, n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(waves), dim3(64), 0, 0, ops, out, n);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("waves %5d gates %3d mode %d %-52s %8.2f us per launch  %6.1f ns per gate\n", waves, n, MODE, name, ms * 1e3 / 20, ms * 1e6 / 20 / n);
}

int main() {
    const int W = 4096;
    std::vector<uint32_t> h((size_t)W * 256);
    uint32_t s = 12345;
    for (auto &v : h) {
        auto r = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
        const uint32_t t0 = 1 + r() % 2, t1 = r() % 3, d0 = r() % 32, s0 = r() % 32, d1 = r() % 32, s1 = r() % 32;
        v = d0 | (s0 << 6) | (t0 << 12) | ((d1 | (s1 << 6) | (t1 << 12)) << 14);
    }
    uint32_t *ops, *out;
    CK(hipMalloc(&ops, h.size() * 4)); CK(hipMalloc(&out, (size_t)W * 64 * 4));
    CK(hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int waves : {512, 2048}) {
        for (int n : {0, 32, 256}) {
            run<0>("3 readlanes, scalar decode", ops, out, waves, n);
            run<1>("1 readlane, scalar decode + scalar masks", ops, out, waves, n);
            run<2>("LDS broadcast, vector decode", ops, out, waves, n);
            run<3>("constant gate (floor of the vector chain)", ops, out, waves, n);
            run<4>("global uniform load per gate", ops, out, waves, n);
            runp<5>("parity form, 4 readlanes of pre-decoded masks", ops, out, waves, n);
            runp<7>("parity form, masks from LDS (broadcast, 4 ahead)", ops, out, waves, n);
            runp<8>("... two streams per wave (half waves)", ops, out, waves, n);
        }
    }
    return 0;
}
