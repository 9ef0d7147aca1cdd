// walker_grid.hpp — the host's side of the walker waves' forward-progress rule (k_walk_persistent, k5_travel.hip).
//
// The walks of pag_travel run as jobs taken from rings in host memory by walker waves.  The waves are ELASTIC: one that
// finds nothing to claim for `idle_us` leaves, and `ensure()` starts new ones whenever jobs are outstanding for which too
// few waves are left.  Nothing in the control flow depends on how many waves are resident:
//   * a job is claimed by exactly one wave (compare-and-swap on the ring's counter) and runs to its end on its own;
//   * a wave waits for the host for at most `idle_us`, so a launch that cannot be resident as a whole (other processes'
//     walkers on the same device, a grid forced larger than the device) stalls the dispatcher for a bounded time only;
//   * the grid's size is a throughput setting: the occupancy the runtime reports for the kernel x compute units, divided
//     by the processes that share the device (PAG_DEVICE_SHARERS), with a floor of one wave per XCD;
//     PAG_WALK_WAVES=<n> forces a total (tests run the walks with 8 waves), PAG_WALK_WAVES_PER_CU=<n> a per-unit figure.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pag_graph_impl.hpp"

namespace pagdev {

struct WalkerGrid {
    pag_graph *g = nullptr;
    TravGraph G{};
    const TravPosted *jobs = nullptr;
    TravJobOut *outs = nullptr;
    uint32_t *done = nullptr;
    TravQueue *q = nullptr;  // fine-grained host memory: posted[] / exit written here, started / exited by the waves
    uint32_t cap = 0, k = 0;
    uint32_t max_waves = 0;
    uint64_t idle_ticks = 0;  // 100 MHz
    uint32_t launched = 0;    // waves started by this session
    uint32_t launches = 0;
    bool up = false;
    bool trace = false;       // (WalkConfig::walk_trace: the launches are reported on stderr)

    // how many waves the device should carry for this process
    static uint32_t default_waves(int device) {
        if (const char *e = std::getenv("PAG_WALK_WAVES")) return (uint32_t)std::max(1, std::atoi(e));
        int n_cu = 256;
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
        const int per_cu = std::max(1, (int)env_int("PAG_WALK_WAVES_PER_CU", trav_walk_waves_per_cu()));
        const int sharers = (int)env_device_sharers();
        const int n_xcd = 8;
        return (uint32_t)std::max(n_xcd, n_cu * per_cu / sharers);
    }

    void init(pag_graph *gg, const TravGraph &tg, const TravPosted *hjobs, TravJobOut *houts, uint32_t *hdone, TravQueue *hq, uint32_t qcap,
              uint32_t kk) {
        g = gg;
        G = tg;
        jobs = hjobs;
        outs = houts;
        done = hdone;
        q = hq;
        cap = qcap;
        k = kk;
        max_waves = default_waves(g->device);
        // the pool's first streams are made here, once per handle: creating one costs ~6 ms (measured, round 5), and a pool that
        // grew inside the walks paid that in the tail of a block, on the control thread, with finished jobs waiting
        prepare_streams(g, 4);
        const double idle_us = (double)std::max<long long>(1, env_int("PAG_WALK_IDLE_US", 2000));
        idle_ticks = (uint64_t)(idle_us * 100.0);
        idle_ticks |= 1ull << 63;  // (the walker waves raise their issue priority: see k_walk_persistent)
        launched = launches = 0;
        up = false;
    }
    static void prepare_streams(pag_graph *gg, size_t n) {
        int lo = 0, hi = 0;
        if (gg->walk_streams.size() >= n || hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return;
        while (gg->walk_streams.size() < n) {
            hipStream_t st = nullptr;
            if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            gg->walk_streams.push_back(st);
        }
    }
    uint32_t exited() const { return __atomic_load_n(&q->exited, __ATOMIC_ACQUIRE); }
    uint32_t started() const { return __atomic_load_n(&q->started, __ATOMIC_ACQUIRE); }
    // waves this session still has (resident, or launched and waiting for room)
    uint32_t have() const { return launched - exited(); }

    // Called after jobs have been published and from the host's wait loop: `outstanding` jobs are posted and not yet
    // fetched.  Starts waves when fewer are left than the jobs could use (with some slack: a launch costs ~10 us of the
    // control thread, so a few missing waves are not replaced one by one).
    int ensure(uint32_t outstanding) {
        if (!outstanding) return PAG_OK;
        if (outstanding > (1u << 28)) {  // (a count that has wrapped: never start waves for it)
            set_error("pag_travel: job accounting is off (%u jobs outstanding)", outstanding);
            return PAG_EFAULT;
        }
        const uint32_t want = std::min(max_waves, outstanding);
        const uint32_t h = have();
        if (h >= want) return PAG_OK;
        const uint32_t missing = want - h;
        if (h != 0 && missing < std::max<uint32_t>(4u, want / 4)) return PAG_OK;
        return launch(missing);
    }
    int launch(uint32_t n) {
        if (!n) return PAG_OK;
        // a stream of the pool whose last launch has drained (launches on one stream would run one after the other);
        // the pool grows to 8 streams, all of the highest priority: the runtime multiplexes streams onto a few hardware
        // queues, and work of the call's own stream must never be queued behind walkers
        const auto tq0 = std::chrono::steady_clock::now();
        hipStream_t st = nullptr;
        for (hipStream_t c : g->walk_streams)
            if (hipStreamQuery(c) == hipSuccess) {
                st = c;
                break;
            }
        (void)hipGetLastError();  // (hipErrorNotReady of the queries)
        if (!st) {
            if (g->walk_streams.size() < 8) {
                int lo = 0, hi = 0;
                PAG_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
                PAG_HIP_TRY(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
                g->walk_streams.push_back(st);
            } else {
                st = g->walk_streams[launches % g->walk_streams.size()];
            }
        }
        const auto tq1 = std::chrono::steady_clock::now();
        const bool drained = hipStreamQuery(st) == hipSuccess;
        (void)hipGetLastError();
        trav_launch_walk_persistent(G, jobs, outs, done, q, g->wq_next, cap, k, n, idle_ticks, st);
        if (trace) {
            const auto tq2 = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[trace] walker launch %u: %u waves on stream %p (%s, pool of %zu), have %u, picking the stream %.3f ms, the launch call %.3f ms\n", launches, n, (void *)st,
                         drained ? "drained" : "NOT drained", g->walk_streams.size(), have(), std::chrono::duration<double, std::milli>(tq1 - tq0).count(),
                         std::chrono::duration<double, std::milli>(tq2 - tq1).count());
        }
        if (hipGetLastError() != hipSuccess) {
            set_error("pag_travel: walker launch failed");
            return PAG_EFAULT;
        }
        launched += n;
        launches += 1;
        up = true;
        return PAG_OK;
    }
    // no further jobs: the waves leave at their next look at the queue; waits for them
    void shutdown() {
        if (!up) return;
        __atomic_store_n(&q->exit, 1u, __ATOMIC_RELEASE);
        for (hipStream_t c : g->walk_streams) hipStreamSynchronize(c);
        up = false;
    }
};

}  // namespace pagdev
