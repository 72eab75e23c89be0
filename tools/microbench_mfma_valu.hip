// microbench_mfma_valu.hip -- does the matrix pipe of gfx950 run under VALU instructions of the SAME wave (or of another wave of the same SIMD)?
// One workgroup of 256 threads per CU (one wave per SIMD; WAVES2=1: 512 threads, two per SIMD).  Per iteration 4 MFMAs (32x32x16 bf16,
// independent accumulators), each followed by K VALU instructions of one kind, all in asm volatile so the order written is the order issued.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench_mfma_valu.hip -o /tmp/mv && /tmp/mv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND 0: v_add_f32 (full rate), 1: v_exp_f32 (transcendental), 2: v_mul_lo_u32, 3: v_xor_b32
template <int KIND> __device__ __forceinline__ void valu(float &f, uint32_t &u, float g, uint32_t w) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"(g));
    if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f));
    if (KIND == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u) : "v"(w));
    if (KIND == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u) : "v"(w));
}

template <int KIND, int K, bool MFMA, bool VALU>
__global__ __launch_bounds__(512) void probe(float *out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
    bf16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (__bf16)(float)(threadIdx.x & 3); b[q] = (__bf16)1.0f; }
    float f[4] = {0.5f, 0.25f, 0.125f, 0.75f};
    uint32_t u[4] = {threadIdx.x, threadIdx.x + 1u, threadIdx.x + 2u, threadIdx.x + 3u};
    const float g = 1.0f + (float)blockIdx.x;
    const uint32_t w = 0x9E3779B9u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MFMA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(a), "v"(b));
            if (VALU) {
#pragma unroll
                for (int k = 0; k < K; ++k) valu<KIND>(f[k & 3], u[k & 3], g, w);
            }
        }
    }
    float s = 0.0f;
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 16; ++q) s += acc[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + f[0] + f[1] + f[2] + f[3] + (float)(u[0] ^ u[1] ^ u[2] ^ u[3]);
}

template <int KIND, int K, bool MFMA, bool VALU> static double run(int threads, float *out, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<KIND, K, MFMA, VALU>), dim3(256), dim3(threads), 0, 0, out, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<KIND, K, MFMA, VALU>), dim3(256), dim3(threads), 0, 0, out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6 / iters;  // ns per iteration (4 MFMAs + 4 K VALU per wave)
}

template <int KIND, int K> static void kind(const char *name, int threads, float *out) {
    const int iters = 200000;
    const double m = run<KIND, K, true, false>(threads, out, iters), v = run<KIND, K, false, true>(threads, out, iters), b = run<KIND, K, true, true>(threads, out, iters);
    printf("%-14s K=%2d  %d waves/SIMD: MFMA only %7.1f ns, VALU only %7.1f ns, interleaved %7.1f ns per iteration (sum %7.1f, max %7.1f)\n", name, K, threads / 256, m, v, b,
           m + v, m > v ? m : v);
}

int main() {
    float *out; CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    for (int threads : {256, 512}) {
        kind<0, 6>("v_add_f32", threads, out);
        kind<3, 6>("v_xor_b32", threads, out);
        kind<1, 2>("v_exp_f32", threads, out);
        kind<2, 2>("v_mul_lo_u32", threads, out);
        kind<1, 1>("v_exp_f32", threads, out);
        kind<2, 1>("v_mul_lo_u32", threads, out);
    }
    return 0;
}
