// How many host threads a pool may start: the hardware threads of the host, shared evenly between the processes of one
// multi-GPU job (LOCAL_WORLD_SIZE, set by torchrun; at least 8 each), or PAGH_MAX_THREADS.
// (NOT clipped by a container's CPU quota: measured on a 256-thread host with a 16-CPU cgroup quota, the text dumps of a
// configs[1] block take ~85 ms with 16 threads, ~50 ms with 32 and ~35 ms with 64 — the quota is enforced per 100 ms
// period and a short burst of threads fits inside it.)
#pragma once
#include <algorithm>
#include <cstdlib>
#include <thread>

namespace pagh {

// a variable of the environment as an integer; `otherwise` when it is not set
inline long long envInt(const char *name, long long otherwise) {
    const char *e = std::getenv(name);
    return e ? std::atoll(e) : otherwise;
}
// PAGRAPH_TIMING: lap timers of the host stages on stderr
inline bool envTiming() { return std::getenv("PAGRAPH_TIMING") != nullptr; }
// PAGH_OVERLAP_THREADS=<n>: host threads of a block's host half while it runs beside the next block's device work (0: not given)
inline unsigned envOverlapThreads() {
    const char *e = std::getenv("PAGH_OVERLAP_THREADS");
    return e ? static_cast<unsigned>(std::max(1, std::atoi(e))) : 0u;
}


inline unsigned usableCpus() {
    static const unsigned n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        if (const char *e = std::getenv("LOCAL_WORLD_SIZE")) {
            const int w = std::atoi(e);
            if (w > 1) hw = std::max(8u, hw / (unsigned)w);
        }
        if (const char *e = std::getenv("PAGH_MAX_THREADS")) {
            const int v = std::atoi(e);
            if (v > 0) hw = (unsigned)v;
        }
        return hw;
    }();
    return n;
}

}  // namespace pagh
