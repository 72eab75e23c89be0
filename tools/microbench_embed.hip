// Where do the cycles of the bit-expanding MFMA loop go?  One wave per SIMD, 256 blocks x 256 threads,
// the inner phase of embed_bits_kernel (kernels_policy.hip) with parts switched off:
//   bit 0: expand the A fragments from a word (2 VALU per VGPR)   bit 1: read the B fragments from LDS
//   bit 2: pin the MFMA/VALU interleave with sched_group_barrier
// hipcc --offload-arch=gfx950 -O3 tools/microbench_embed.hip -o tools/bin/microbench_embed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ inline bf16x8 expand(uint32_t w, const uint32_t (&sh)[4]) {
    u32x4 v;
    v.x = __builtin_amdgcn_alignbit(w, w, sh[0]) & 0x40004000u;
    v.y = __builtin_amdgcn_alignbit(w, w, sh[1]) & 0x40004000u;
    v.z = __builtin_amdgcn_alignbit(w, w, sh[2]) & 0x40004000u;
    v.w = __builtin_amdgcn_alignbit(w, w, sh[3]) & 0x40004000u;
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const uint4 *src, float *out, uint32_t iters, uint64_t *cycles) {
    extern __shared__ uint4 lds[];
    for (uint32_t i = threadIdx.x; i < 8192; i += 256) lds[i] = src[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, h = lane >> 5;
    uint32_t sh[2][4];
    for (int sp = 0; sp < 2; ++sp) for (int j = 0; j < 4; ++j) sh[sp][j] = 8 * sp + 4 * h + j;
    uint32_t w[4];
    for (int i = 0; i < 4; ++i) w[i] = src[threadIdx.x + 64 * i].x;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int nb = 0; nb < 2; ++nb) for (int q = 0; q < 16; ++q) acc[i][nb][q] = 0.f;
    bf16x8 a_c[4], b_c[2];
    for (int i = 0; i < 4; ++i) a_c[i] = expand(w[i], sh[0]);
    b_c[0] = __builtin_bit_cast(bf16x8, lds[lane]);
    b_c[1] = __builtin_bit_cast(bf16x8, lds[64 + lane]);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; ++it) {
        const uint4 *bl = lds + (it & 7u) * 1024u + lane;
#pragma unroll
        for (uint32_t ss = 0; ss < 8; ++ss) {
            __builtin_amdgcn_sched_barrier(0);
            bf16x8 a_n[4], b_n[2];
            if (MODE & 2) {
                b_n[0] = __builtin_bit_cast(bf16x8, bl[(ss * 2u + 0u) * 64u]);
                b_n[1] = __builtin_bit_cast(bf16x8, bl[(ss * 2u + 1u) * 64u]);
            } else { b_n[0] = b_c[0]; b_n[1] = b_c[1]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE & 1) { a_n[i] = expand(w[i], sh[ss & 1]); w[i] += 0x01010101u * (ss + 1); }
                else a_n[i] = a_c[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[0], acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_c[i], b_c[1], acc[i][1], 0, 0, 0);
            }
            if (MODE & 4) {
                if (MODE & 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) a_c[i] = a_n[i];
            b_c[0] = b_n[0]; b_c[1] = b_n[1];
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int nb = 0; nb < 2; ++nb) for (int q = 0; q < 16; ++q) s += acc[i][nb][q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const uint4 *src, float *out, uint64_t *cyc, const char *name) {
    const uint32_t iters = 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 131072, 0, src, out, iters, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    uint64_t c[256]; hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
    const double mfmas = iters * 64.0;
    printf("%-40s %8.1f us  %6.1f shader cycles / MFMA  (wall: %.1f ns / MFMA)  %.0f TFLOP/s\n", name, ms * 1e3, c[0] / mfmas, ms * 1e6 / mfmas,
           mfmas * 1024 * 32768 / (ms * 1e-3) * 1e-12);
}

int main() {
    uint4 *src; float *out; uint64_t *cyc;
    hipMalloc(&src, 8192 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    uint32_t *h = new uint32_t[8192 * 4];
    for (int i = 0; i < 8192 * 4; ++i) h[i] = (0x3C00u + (i * 2654435761u >> 20 & 0x3FF)) * 0x00010001u;
    hipMemcpy(src, h, 8192 * 16, hipMemcpyHostToDevice);
    run<0>(src, out, cyc, "MFMA only");
    run<1>(src, out, cyc, "MFMA + expansion");
    run<5>(src, out, cyc, "MFMA + expansion, pinned interleave");
    run<2>(src, out, cyc, "MFMA + LDS B reads");
    run<6>(src, out, cyc, "MFMA + LDS B reads, pinned");
    run<3>(src, out, cyc, "MFMA + expansion + LDS");
    run<7>(src, out, cyc, "MFMA + expansion + LDS, pinned");
    return 0;
}
