// microbench_step.hip -- ablation timings of the TILE-layout step kernel (development tool, not
// part of libqgym).  Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 \
//   -I qiskit_gym_amd/csrc tools/microbench_step.hip -o gpurun_out/microbench_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../qiskit_gym_amd/csrc/kernels_qm.hip"

using namespace qg;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// variants of the step kernel body, selected at compile time
template <int V, int BLOCK>
__global__ __launch_bounds__(BLOCK) void variant_kernel(StepArgs a) {
    using Rows = QmRows<16, true>;
    extern __shared__ GateEntry s_gates[];
    const uint64_t env = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    if (V == 0) return;  // empty kernel
    uint4 *tile = reinterpret_cast<uint4 *>(a.state) + (env >> 6) * (uint64_t)(Rows::G * 64);
    Rows s;
    qm_load<16, true>(tile, lane, s);
    int64_t act = reinterpret_cast<const int32_t *>(a.actions)[env];
    int32_t depth = a.depth[env];
    if (V == 3) {  // LDS table
        for (uint32_t i = threadIdx.x; i < a.num_actions; i += blockDim.x) s_gates[i] = a.gates[i];
        __syncthreads();
    }
    GateEntry *wtab = s_gates + (threadIdx.x >> 6) * 256;
    if (V == 14) {  // wave-private LDS copy of the table: no block barrier
        for (uint32_t i = lane; i < a.num_actions; i += 64) wtab[i] = a.gates[i];
    }
    GateEntry g = {QM_IDENTITY << 10, 0.0f};
    if (V == 3) g = s_gates[act]; else if (V == 14) g = wtab[act]; else if (V >= 2) g = a.gates[act];
    uint32_t dirty = 0;
    bool solved = false;
    if (V >= 2) {
        dirty = qm_apply<16, true>(s, g.ops);
        solved = qm_solved<16, true>(s, a.N);
    } else {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) acc |= s.r[i];
        solved = acc == 0x12345;
    }
    if (V == 1 || V == 2 || V == 3) {  // no row stores; keep results live
        if (solved) a.reward[env] = (float)depth + g.penalty + (float)dirty;
        return;
    }
    if (V == 12) dirty &= (0u - dirty);                     // keep one dirty group
    if (V == 13) { uint32_t lo = dirty & (0u - dirty); uint32_t rest = dirty ^ lo; dirty = lo | (rest & (0u - rest)); }  // two
    // V >= 4: full stores
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int gi = 0; gi < Rows::G; ++gi)
        if (V == 5 || ((dirty >> gi) & 1u)) {
            u32x4 val = {s.r[4 * gi], s.r[4 * gi + 1], s.r[4 * gi + 2], s.r[4 * gi + 3]};
            uint4 *p = &tile[gi * 64 + lane];
            if (V == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(val) : "memory");
            else if (V == 9) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(val) : "memory");
            else if (V == 10) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(val) : "memory");
            else *p = make_uint4(val.x, val.y, val.z, val.w);
        }
    depth = depth > 0 ? depth - 1 : 0;
    const float rew = (solved ? 1.0f : 0.0f) - g.penalty;
    if (V == 6) {  // packed scalars: one 8-byte and one 2-byte store
        reinterpret_cast<uint2 *>(a.layers)[env] = make_uint2(__float_as_uint(rew), (uint32_t)depth);
        reinterpret_cast<uint16_t *>(a.sol)[env] = (uint16_t)((depth == 0 || solved) | (solved << 8));
    } else if (V == 7) {  // no scalar stores at all except reward
        a.reward[env] = rew + (float)depth;
    } else {
        a.depth[env] = depth;
        a.reward[env] = rew;
        a.done[env] = (uint8_t)(depth == 0 || solved);
        a.success[env] = (uint8_t)solved;
    }
}

static bool g_distinct_actions = false;
template <int V, int BLOCK>
static float time_variant(const StepArgs &a, int iters, hipStream_t st) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t lds = (V == 14) ? (BLOCK / 64) * 256 * sizeof(GateEntry) : a.num_actions * sizeof(GateEntry);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((variant_kernel<V, BLOCK>), dim3((unsigned)(a.B / BLOCK)), dim3(BLOCK), lds, st, a);
    // capture `iters` back-to-back launches into a graph to remove host launch overhead
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) {
        StepArgs b = a;
        if (g_distinct_actions) b.actions = (const char *)a.actions + (size_t)i * a.B * 4;
        hipLaunchKernelGGL((variant_kernel<V, BLOCK>), dim3((unsigned)(a.B / BLOCK)), dim3(BLOCK), lds, st, b);
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    hipGraphExecDestroy(exec); hipGraphDestroy(graph);
    return ms * 1e3f / iters;
}

static int g_ring = 1 << 30;
static float time_real(const StepArgs &a, int iters, hipStream_t st, bool distinct) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) {
        StepArgs b = a;
        if (distinct) b.actions = (const char *)a.actions + (size_t)(i % g_ring) * a.B * 4;
        CK(qm_step(b, 16, true, st));
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    hipGraphExecDestroy(exec); hipGraphDestroy(graph);
    return ms * 1e3f / iters;
}

// S independent sub-batch chains inside one graph (fork at the start, join at the end)
static float time_real_chains(const StepArgs &a, int iters, hipStream_t st, int S) {
    hipEvent_t e0, e1, fork, joins[16];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    hipStream_t sub[16];
    for (int i = 0; i < S; ++i) { CK(hipStreamCreateWithFlags(&sub[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&joins[i], hipEventDisableTiming)); }
    hipGraph_t graph; hipGraphExec_t exec;
    const uint64_t Bs = a.B / S;  // multiple of 64 for the sizes used here
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, st));
    for (int c = 0; c < S; ++c) {
        CK(hipStreamWaitEvent(sub[c], fork, 0));
        for (int i = 0; i < iters; ++i) {
            StepArgs b = a;
            const uint64_t base = c * Bs;
            b.B = Bs;
            b.state = (char *)a.state + base * 128;
            b.actions = (const char *)a.actions + ((size_t)i * a.B + base) * 4;
            b.depth = a.depth + base; b.reward = a.reward + base; b.done = a.done + base; b.success = a.success + base;
            CK(qm_step(b, 16, true, sub[c]));
        }
        CK(hipEventRecord(joins[c], sub[c]));
        CK(hipStreamWaitEvent(st, joins[c], 0));
    }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    hipGraphExecDestroy(exec); hipGraphDestroy(graph);
    return ms * 1e3f / iters;
}

int main(int argc, char **argv) {
    const uint64_t B = argc > 1 ? strtoull(argv[1], 0, 10) : 65536;
    const uint32_t A = 170, N = 16;
    StepArgs a{};
    CK(hipMalloc(&a.state, B * 128)); CK(hipMemset(a.state, 0x5a, B * 128));
    const int it = 500;
    std::vector<int32_t> acts(B * it);
    for (uint64_t i = 0; i < B * it; ++i) acts[i] = (int32_t)((i * 2654435761u) % A);
    void *d_act; CK(hipMalloc(&d_act, B * 4 * it)); CK(hipMemcpy(d_act, acts.data(), B * 4 * it, hipMemcpyHostToDevice));
    a.actions = d_act;
    std::vector<GateEntry> table(A);
    for (uint32_t i = 0; i < A; ++i) { uint32_t q0 = i % 16, q1 = (i + 1) % 16; table[i].ops = q0 | (q1 << 5) | ((i % 3 ? 0x8521u : 0x2184u) << 10); table[i].penalty = 0.01f; }
    GateEntry *d_g; CK(hipMalloc(&d_g, A * 8)); CK(hipMemcpy(d_g, table.data(), A * 8, hipMemcpyHostToDevice));
    a.gates = d_g;
    CK(hipMalloc(&a.depth, B * 4)); CK(hipMemset(a.depth, 0, B * 4));
    CK(hipMalloc(&a.reward, B * 4)); CK(hipMalloc(&a.done, B)); CK(hipMalloc(&a.success, B));
    a.B = B; a.N = N; a.num_actions = A; a.T = 1;
    CK(hipMalloc(&a.layers, B * 8)); CK(hipMalloc(&a.sol, B * 2));
    hipStream_t st; CK(hipStreamCreate(&st));
    printf("B=%llu  (us per launch, back-to-back in a hipGraph)\n", (unsigned long long)B);
    printf("V0 empty           block256: %.2f\n", time_variant<0, 256>(a, it, st));
    printf("V1 loads only      block256: %.2f\n", time_variant<1, 256>(a, it, st));
    printf("V2 +compute(glob)  block256: %.2f\n", time_variant<2, 256>(a, it, st));
    printf("V3 +compute(LDS)   block256: %.2f\n", time_variant<3, 256>(a, it, st));
    printf("V4 full            block256: %.2f\n", time_variant<4, 256>(a, it, st));
    printf("V5 all-group store block256: %.2f\n", time_variant<5, 256>(a, it, st));
    printf("V6 packed scalars  block256: %.2f\n", time_variant<6, 256>(a, it, st));
    printf("V7 reward only     block256: %.2f\n", time_variant<7, 256>(a, it, st));
    printf("REAL qm_step_kernel same actions : %.2f\n", time_real(a, it, st, false));
    printf("REAL qm_step_kernel distinct     : %.2f\n", time_real(a, it, st, true));
    for (int ring : {2, 4, 16, 64, 256}) { g_ring = ring; printf("REAL distinct, ring of %3d action slices : %.2f\n", ring, time_real(a, it, st, true)); }
    g_ring = 1 << 30;
    for (int S : {1}) printf("REAL distinct, %2d sub-batch chains : %.2f us per full step\n", S, time_real_chains(a, it, st, S));
    g_distinct_actions = true;
    printf("-- distinct action slice per launch --\n");
    printf("V2 +compute(glob)  block256: %.2f\n", time_variant<2, 256>(a, it, st));
    printf("V4 full            block256: %.2f\n", time_variant<4, 256>(a, it, st));
    g_distinct_actions = false;
    printf("V14 wave-private LDS table: %.2f\n", time_variant<14, 256>(a, it, st));
    printf("V12 one dirty group block256: %.2f\n", time_variant<12, 256>(a, it, st));
    printf("V13 two dirty groups block256: %.2f\n", time_variant<13, 256>(a, it, st));
    printf("V8 rows sc1        block256: %.2f\n", time_variant<8, 256>(a, it, st));
    printf("V9 rows nt         block256: %.2f\n", time_variant<9, 256>(a, it, st));
    printf("V10 rows sc0 sc1   block256: %.2f\n", time_variant<10, 256>(a, it, st));
    printf("V4 full            block 64: %.2f\n", time_variant<4, 64>(a, it, st));
    printf("V6 packed scalars  block 64: %.2f\n", time_variant<6, 64>(a, it, st));
    return 0;
}
