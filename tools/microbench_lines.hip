// microbench_lines.hip -- does a per-lane 16-byte gather from a lane-interleaved tile fetch whole 128-byte lines?  The model DESIGN.md section 4
// uses for the one-step kernels' over-fetch: a tile is G groups of 1 KiB (64 lanes x 16 B), a lane reads `picks` of its G groups, the 8 lanes
// that share a 128-byte line pick independently, so a line is fetched with probability 1 - (1 - picks / G)^8 and the fetch per env is that
// fraction of the tile's G x 16 B.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` (tools/pmc_lines.sh) and compare.
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_lines.hip -o /tmp/mb_lines
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int G, int PICKS>
__global__ __launch_bounds__(256) void gather(const uint4 *state, uint32_t *out, uint32_t B, uint32_t salt) {
    const uint32_t env = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    if (env >= B) return;
    const uint4 *tile = state + (size_t)(env >> 6) * (G * 64);
    uint32_t h = (env * 2654435761u) ^ (salt * 0x9E3779B9u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < PICKS; ++p) {
        const uint32_t g = (h >> (8 * p)) % G;
        const uint4 v = tile[g * 64 + lane];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[env] = acc;
}

template <int G, int PICKS>
static void run(const uint4 *state, uint32_t *out, uint32_t B) {
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL((gather<G, PICKS>), dim3((B + 255) / 256), dim3(256), 0, 0, state, out, B, (uint32_t)i);
    CK(hipDeviceSynchronize());
    double p = (double)PICKS / G, frac = 1.0, q = 1.0 - p;
    for (int i = 0; i < 8; ++i) frac *= q;
    // PICKS independent picks of one lane may coincide: P(lane touches a given group) = 1 - (1 - 1/G)^PICKS
    double pl = 1.0; for (int i = 0; i < PICKS; ++i) pl *= 1.0 - 1.0 / G; pl = 1.0 - pl;
    double f8 = 1.0; for (int i = 0; i < 8; ++i) f8 *= 1.0 - pl;
    printf("gather<%d, %d>  tile %4d B/env  needed %3d B/env  model (128-byte lines) %6.1f B/env  model (64-byte sectors) %6.1f B/env\n", G, PICKS, G * 16, PICKS * 16,
           (1.0 - f8) * G * 16, (1.0 - (1.0 - pl) * (1.0 - pl) * (1.0 - pl) * (1.0 - pl)) * G * 16);
    (void)frac;
}

int main() {
    const uint32_t B = 1u << 20;
    uint4 *state; uint32_t *out;
    CK(hipMalloc(&state, (size_t)(B / 64) * 20 * 64 * 16));
    CK(hipMemset(state, 1, (size_t)(B / 64) * 20 * 64 * 16));
    CK(hipMalloc(&out, (size_t)B * 4));
    run<8, 2>(state, out, B);    // the headline kernel's pattern (CliffordEnv 16q: 8 groups, a gate touches <= 2)
    run<8, 1>(state, out, B);
    run<4, 1>(state, out, B);
    run<16, 2>(state, out, B);
    run<20, 2>(state, out, B);   // PauliEnv 20q's qubit records (16-byte form)
    run<20, 1>(state, out, B);
    return 0;
}
