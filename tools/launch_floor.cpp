// launch_floor.cpp -- what the HIP runtime itself allows many host threads: every thread has its own stream and loops { launch an empty kernel of one wave;
// hipStreamSynchronize } -- no libqgym, no memory traffic.  The scalar qg_env_* API (one launch + one synchronisation per Env::step) cannot go
// faster than this from the same number of threads; tools/api_bench.cpp is the same loop through the library.
//   hipcc -O2 -std=c++17 tools/launch_floor.cpp -lpthread -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void empty_kernel(int *p) {
    if (p && threadIdx.x == 9999) *p = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const int iters = 2000;
    for (int threads : {1, 4, 16, 32}) {
        std::vector<std::thread> pool;
        std::vector<double> per(threads, 0.0);
        const double t0 = now();
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                hipSetDevice(0);
                hipStream_t st;
                hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
                for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, nullptr); hipStreamSynchronize(st); }
                const double a = now();
                for (int i = 0; i < iters; ++i) {
                    hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, nullptr);
                    hipStreamSynchronize(st);
                }
                per[t] = (now() - a) / iters * 1e6;
                hipStreamDestroy(st);
            });
        for (auto &th : pool) th.join();
        const double wall = now() - t0;
        double mean = 0;
        for (double x : per) mean += x / threads;
        printf("threads %2d: %9.1f {launch, synchronise} pairs per second in total (wall, incl. stream set-up); %6.1f us per pair and thread\n", threads,
               threads * (double)(iters + 50) / wall, mean);
    }
    return 0;
}
