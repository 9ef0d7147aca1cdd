// Internal: the handle behind include/pagraph_hip.h and its pooled device buffers.
#pragma once
#include <algorithm>
#include <chrono>
#include <vector>

#include "pag_device.hpp"
#include "pag_travel.hpp"

struct pag_graph {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t k = 0;
    uint64_t n_solid = 0;
    int all_solid = 0;
    uint32_t *solid_bits = nullptr;  // 4^k bits
    // finished graph (device): k-mer-sorted streams with in-place segment results
    uint64_t n_t = 0, n_e = 0;
    uint32_t *tkey = nullptr;
    uint64_t *tval = nullptr;
    uint32_t *tseg = nullptr;
    uint16_t *tcnt = nullptr;
    uint32_t *ekey = nullptr;
    uint64_t *eval = nullptr;
    uint32_t *eseg = nullptr;
    pag_build_stats stats{};
    // pag_shard_extract -> pag_shard_take: sizes of the partitioned streams (tuples, edges) and which buffer of each
    // ping-pong pair holds them
    uint64_t shard_x[2] = {0, 0};
    int shard_in0[2] = {1, 1};
    // device memory pool: every buffer of the pipeline lives in a named slot that is reused (and only
    // ever grown) across pag_process calls, so steady-state calls do no hipMalloc/hipFree at all
    struct Slot {
        void *p = nullptr;
        size_t cap = 0;
    };
    Slot pool[256];
    // what the pool cost (PAGRAPH_TIMING reports it per stage: at the sizes of BASELINE configs[2] / [3] a fresh handle's tens of
    // GB are seconds of hipMalloc / hipFree)
    double alloc_ms = 0;
    uint64_t alloc_bytes = 0, alloc_calls = 0;
    // per-contig walker buffers (k5_travel_host.hip), grown on demand like the slots above
    std::vector<Slot> cpool;
    // while the persistent walker is resident nothing may be hipFree'd (it synchronises the device): replaced
    // buffers are parked here and released afterwards
    bool defer_free = false;
    std::vector<void *> deferred;
    // job queue of the persistent walker (fine-grained host memory) and its device-side ticket counter
    void *wq_host = nullptr;
    size_t wq_bytes = 0;
    uint32_t *wq_next = nullptr;
    std::vector<hipStream_t> walk_streams;  // the walker launches' streams (WalkerGrid, walker_grid.hpp)
    // traversal state (k5_travel_host.hip)
    pagdev::TravGraph tg{};
    bool tg_ready = false;
    uint64_t tg_dev = 0;   // parameters the successor records were built with
    double tg_err = 0;
    // result of the last pag_travel: one array in pinned host memory (grown, never released before pag_destroy), slot
    // 2 * contig + (reverse ? 1 : 0) at path_off / path_len
    pag_path_node *path_store = nullptr;
    size_t path_cap = 0;
    std::vector<uint64_t> path_off, path_len;
    hipStream_t deliver_stream = nullptr;  // copies of finished contigs' paths while the walks run (pag_travel)
    std::vector<const pag_path_node *> path_ptr;  // non-null: the orientation's path, delivered while the walks ran (pinned fetch memory)
    std::vector<uint8_t> path_valid;  // that orientation was traversed by the last pag_travel
    // device arena of the walker's job buffers (bump pointer, reset by every pag_travel)
    void *walk_arena = nullptr;
    size_t walk_arena_cap = 0, walk_arena_used = 0;
    std::vector<void *> fetch_chunks;  // pinned chunks for the paths fetched from the walker (pag_travel)
    std::vector<size_t> fetch_chunk_bytes;
    // the graph holds only a region of the block (pag_shard_set_region after pag_shard_import): the reference bands of its
    // coordinate-free vertices, [lo, hi) pairs sorted, and which ends are open (the block goes on beyond them, on other ranks)
    bool regional = false;
    std::vector<uint32_t> region_ref_iv;
    std::vector<uint8_t> region_ref_open;
    // the traversal view (g->tg) was built for these orientations only (trav_view_region, k5_travel_host.hip): it leaves out
    // what no traversal of them can examine.  view_off: a walk did leave the view (pag_travel then rebuilds the whole graph's
    // view and walks again); both are reset with the graph.
    bool view_pruned = false, view_off = false;
    std::vector<int32_t> view_orient;
    uint64_t view_counts[3] = {0, 0, 0};  // nodes, vertices, edges of the view
    uint64_t view_fallbacks = 0;          // (since the handle was created)
    double succ_per_vertex = 3.0;  // emission-stream slots per vertex of the last traversal graph: sizes the stream of the next (trav_prepare_graph)
    uint64_t n_zero_ctg = 0;  // new ids below it: vertices without a contig coordinate, in reference-coordinate order
    // pinned host staging area of the traversal (packed job results, uploads)
    void *pin_host = nullptr;
    size_t pin_bytes = 0;
    // debug: raw emitted streams (host copies), kept when PAG_DEBUG_KEEP_STREAMS=1
    std::vector<uint32_t> dbg_tkey, dbg_ekey;
    std::vector<uint64_t> dbg_tval, dbg_eval;
};


namespace pagdev {

struct DevBuf {  // a view of one pool slot of the handle (never frees; pag_destroy does)
    pag_graph *g = nullptr;
    pag_graph::Slot *sl = nullptr;
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(pag_graph *gg, int s) : g(gg), sl(&gg->pool[s]) {}
    DevBuf(pag_graph *gg, pag_graph::Slot *slot) : g(gg), sl(slot) {}
    int alloc(size_t bytes) {
        if (bytes == 0) bytes = 16;
        if (sl->cap < bytes) {
            const auto t0 = std::chrono::steady_clock::now();
            if (sl->p) {
                if (g->defer_free) g->deferred.push_back(sl->p);
                else hipFree(sl->p);
            }
            sl->p = nullptr;
            sl->cap = 0;
            // (room to grow without another hipMalloc: an eighth, at most 512 MB — an eighth of every slot was 25 GB of a 90 Mb block at 30x)
            size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)512 << 20) + 256;
            hipError_t e = hipMalloc(&sl->p, want);
            g->alloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            g->alloc_bytes += want;
            g->alloc_calls += 1;
            if (e != hipSuccess) {
                sl->p = nullptr;
                pagdev::set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
                return PAG_ENOMEM;
            }
            sl->cap = want;
        }
        p = sl->p;
        return PAG_OK;
    }
    template <typename T>
    T *as() const {
        return (T *)p;
    }
};


// slot numbers 0..63 belong to pag_process (pag_api.hip), 64..127 to the traversal, 128..191 to pag_prepare (k_prepare.hip), 192.. to pag_shard_select
enum { TRAV_SLOT0 = 64 };

}  // namespace pagdev
