// What keeps the MFMA pipe of embed_words_kernel (kernels_policy.hip) idle?  Its chunk loop (per k-step: 2 A fragments x 4 B
// fragments = 8 v_mfma_f32_32x32x16_bf16; 6 k-steps per chunk; 4 waves per workgroup, 2 workgroups per CU) with parts switched on:
//   bit 0: A fragments expanded from bit words (5 VALU each)     bit 1: B fragments read from LDS (4 ds_read_b128 per k-step, pipelined)
//   bit 2: one workgroup barrier per chunk                        bit 3: the next chunk staged global -> LDS (24 KiB per chunk)
// hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma.hip -o /tmp/microbench_mfma && /tmp/microbench_mfma
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr uint32_t KPG = 6, NB = 4, MA = 2, CHUNK_VEC = KPG * NB * 64;

__device__ __forceinline__ bf16x8 expand(uint32_t w, uint32_t sh) {
    const uint32_t t = __builtin_amdgcn_alignbit(w, w, sh);
    u32x4 v;
    v.x = t & 0x40004000u; v.y = t & 0x20002000u; v.z = t & 0x10001000u; v.w = t & 0x08000800u;
    return __builtin_bit_cast(bf16x8, v);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const uint4 *src, float *out, uint32_t chunks) {
    __shared__ uint4 cbuf[2 * CHUNK_VEC];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, h = lane >> 5;
    for (uint32_t i = threadIdx.x; i < 2 * CHUNK_VEC; i += 256) cbuf[i] = src[i];
    __syncthreads();
    uint4 cur[MA];
    for (uint32_t i = 0; i < MA; ++i) cur[i] = src[threadIdx.x + 256 * i];
    f32x16 acc[MA][NB];
    for (uint32_t i = 0; i < MA; ++i) for (uint32_t f = 0; f < NB; ++f) for (uint32_t q = 0; q < 16; ++q) acc[i][f][q] = 0.f;
    bf16x8 af[2][MA], bfr[2][NB];
    for (uint32_t i = 0; i < MA; ++i) af[0][i] = af[1][i] = expand(cur[i].x, 4 * h);
    for (uint32_t f = 0; f < NB; ++f) bfr[0][f] = bfr[1][f] = __builtin_bit_cast(bf16x8, cbuf[f * 64 + lane]);
    for (uint32_t g = 0; g < chunks; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 8) {
            const uint4 *s = src + (uint64_t)((g * 7u + blockIdx.x) & 63u) * CHUNK_VEC;
            uint4 *dst = cbuf + ((g + 1u) & 1u) * CHUNK_VEC;
#pragma unroll
            for (uint32_t c = 0; c < CHUNK_VEC; c += 256)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s + c + wave * 64 + lane),
                                                 (__attribute__((address_space(3))) void *)(dst + c + wave * 64), 16, 0, 0);
        }
        const uint4 *bl = cbuf + (g & 1u) * CHUNK_VEC + lane;
#pragma unroll
        for (uint32_t s = 0; s < KPG; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t c = s & 1u, n = c ^ 1u;
            if (s + 1 < KPG) {
                if (MODE & 2) {
#pragma unroll
                    for (uint32_t f = 0; f < NB; ++f) bfr[n][f] = __builtin_bit_cast(bf16x8, bl[((s + 1) * NB + f) * 64]);
                }
                if (MODE & 1) {
#pragma unroll
                    for (uint32_t i = 0; i < MA; ++i) {
                        const uint32_t w = (s & 2) ? cur[i].y : cur[i].x;
                        af[n][i] = expand(w + g, (s & 1) ? 4 * h : 8 + 4 * h);
                    }
                }
            }
#pragma unroll
            for (uint32_t f = 0; f < NB; ++f)
#pragma unroll
                for (uint32_t i = 0; i < MA; ++i)
                    acc[i][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE & 1) ? af[c][i] : af[0][i], (MODE & 2) ? bfr[c][f] : bfr[0][f], acc[i][f], 0, 0, 0);
            if (s + 1 < KPG && (MODE & 3)) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (MODE & 2) __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
#pragma unroll
                for (uint32_t m = 1; m < NB * MA; ++m) {
                    if (MODE & 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE & 4) __syncthreads();
        if (MODE & 2) {
            const uint4 *bn = cbuf + ((g + 1u) & 1u) * CHUNK_VEC + lane;
#pragma unroll
            for (uint32_t f = 0; f < NB; ++f) bfr[0][f] = __builtin_bit_cast(bf16x8, bn[f * 64]);
        }
    }
    float r = 0.f;
    for (uint32_t i = 0; i < MA; ++i) for (uint32_t f = 0; f < NB; ++f) for (uint32_t q = 0; q < 16; ++q) r += acc[i][f][q];
    out[(uint64_t)blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(const uint4 *src, float *out, uint32_t chunks, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, chunks);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, src, out, chunks);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps;
    const double mfma_per_simd = (double)blocks / 256.0 * chunks * KPG * NB * MA;  // 4 waves of a block sit on 4 SIMDs
    printf("mode %2d (%s%s%s%s): %7.1f us, %5.1f cycles of 2.4 GHz per MFMA per SIMD (32 = pipe limit at 2.4 GHz)\n", MODE, (MODE & 1) ? "expand " : "", (MODE & 2) ? "lds " : "",
           (MODE & 4) ? "barrier " : "", (MODE & 8) ? "stage " : "", us, us * 2400.0 / mfma_per_simd);
}

int main(int argc, char **argv) {
    uint4 *src; float *out;
    const size_t bytes = 64 * CHUNK_VEC * 16 + 4096 * 16;
    hipMalloc(&src, bytes);
    const bool random = argc > 2 && atoi(argv[2]);  // random bits: bf16 weights of every magnitude, dense A bits (the data decides the power draw, hence the clock)
    if (random) {
        uint16_t *hbuf = (uint16_t *)malloc(bytes);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < bytes / 2; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            hbuf[i] = (uint16_t)(0x3C00u + (x & 0x3FFu)) | (uint16_t)((x >> 20) & 0x8000u);  // bf16 around +-0.01 .. 0.03
        }
        hipMemcpy(src, hbuf, bytes, hipMemcpyHostToDevice);
        free(hbuf);
    } else {
        hipMemset(src, 0x11, bytes);
    }
    hipMalloc(&out, 4096 * 256 * 4);
    const uint32_t chunks = argc > 1 ? (uint32_t)atoi(argv[1]) : 80;
    printf("data: %s\n", random ? "random" : "constant");
    for (int blocks : {512, 1024}) {
        printf("%d workgroups x 4 waves, %u chunks of %u k-steps\n", blocks, chunks, KPG);
        run<0>(src, out, chunks, blocks);
        run<1>(src, out, chunks, blocks);
        run<2>(src, out, chunks, blocks);
        run<3>(src, out, chunks, blocks);
        run<7>(src, out, chunks, blocks);
        run<11>(src, out, chunks, blocks);
        run<15>(src, out, chunks, blocks);
    }
    hipDeviceSynchronize();
    return 0;
}
