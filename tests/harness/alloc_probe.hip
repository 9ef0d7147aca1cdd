// GPU box probe (round 6): what does fresh device memory cost, and does it come faster from several threads at once?
// hipMalloc of G gigabytes as one call, as T threads x G / T, hipFree, a second round (memory the process gave back), first touch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t G = argc > 1 ? (size_t)atoll(argv[1]) : 32;
    hipSetDevice(0);
    hipFree(nullptr);
    for (int round = 0; round < 2; ++round)
        for (int T : {1, 4, 16}) {
            std::vector<void *> p(T, nullptr);
            std::vector<std::thread> th;
            const size_t each = (G << 30) / T;
            double t0 = now();
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] { hipSetDevice(0); if (hipMalloc(&p[t], each) != hipSuccess) p[t] = nullptr; });
            for (auto &x : th) x.join();
            double t1 = now();
            for (int t = 0; t < T; ++t) if (p[t]) hipMemsetAsync(p[t], 0, each, 0);
            hipDeviceSynchronize();
            double t2 = now();
            for (int t = 0; t < T; ++t) if (p[t]) hipMemsetAsync(p[t], 0, each, 0);
            hipDeviceSynchronize();
            double t3 = now();
            for (int t = 0; t < T; ++t) if (p[t]) hipFree(p[t]);
            double t4 = now();
            printf("round %d: %zu GB as %2d threads x %.2f GB: hipMalloc %.3f s (%.1f ms/GB), first memset %.3f s, second memset %.3f s, hipFree %.3f s\n", round, G, T,
                   each / 1073741824.0, t1 - t0, (t1 - t0) * 1e3 / G, t2 - t1, t3 - t2, t4 - t3);
        }
    return 0;
}
