// C ABI of the host half (include/pagraph_host.h): export the device graph, traverse, write outputs.
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>

#include "host_threads.hpp"
#include "assembly.hpp"
#include "host_graph.hpp"
#include "pagraph_host.h"
#include "path_graph.hpp"
#include "position_mapper.hpp"
#include "seq_db.hpp"

namespace {
thread_local char g_err[512] = "";
void setErr(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
double nowMs() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
pagh::SeqDb fromPacked(const pag_seqs *s, const char *const *names, const char *stem, unsigned firstNo) {
    pagh::SeqDb db;
    for (std::uint64_t i = 0; i < s->n_seqs; ++i) {
        std::string name = names ? names[i] : stem + std::to_string(i + firstNo);
        db.addPacked(name, s->packed + s->byte_off[i], s->len[i]);
    }
    db.finish();
    return db;
}
}  // namespace

extern "C" {

const char *pagh_last_error(void) { return g_err; }

// storage kept between calls on one graph handle (one block after the other: releasing and re-faulting ~1 GB of host
// arrays per block costs tens of milliseconds); dropped by pagh_release()
namespace {
struct HandleCache {
    pagh::HostGraph graph;
    std::vector<pagh::TravelSequence> travelled;
    // the host half of a pagh_traverse_begin() that has not been collected by pagh_traverse_end() yet
    std::thread worker;
    bool pending = false;
    int rc = PAG_OK;
    pagh_traverse_stats stats{};
    std::string error;
    std::vector<const pag_path_node *> paths;
    std::vector<std::uint64_t> lens;
    double tBegin = 0, tHost = 0;
    pag_travel_stats tst{};
};
std::mutex g_cacheLock;
std::map<const pag_graph *, std::unique_ptr<HandleCache>> g_cache;
HandleCache &cacheOf(const pag_graph *g) {
    std::lock_guard<std::mutex> l(g_cacheLock);
    auto &slot = g_cache[g];
    if (!slot) slot.reset(new HandleCache());
    return *slot;
}
}  // namespace

void pagh_release(pag_graph *g) {
    std::unique_ptr<HandleCache> hc;
    {
        std::lock_guard<std::mutex> l(g_cacheLock);
        auto it = g_cache.find(g);
        if (it == g_cache.end()) return;
        hc = std::move(it->second);
        g_cache.erase(it);
    }
    if (hc && hc->worker.joinable()) hc->worker.join();
}

// chain selection + writers over finished travel sequences (PAssembly::testTravel5 after its travelSequence loop,
// PAssembly.cpp:81-336)
int pagh_assemble_paths(pag_graph *cache_key, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names, const pag_seqs *refs,
                        const char *const *ref_names, const int32_t *ctg_orient, const pag_path_node *const *paths,
                        const uint64_t *path_len, uint32_t ref_threads, uint64_t epsilon, uint64_t min_len, const char *out_dir,
                        const char *prefix, uint32_t host_threads, pagh_traverse_stats *stats) {
    if (!ctgs || !refs || !ctg_orient || !out_dir || !paths || !path_len) return PAG_EINVAL;
    try {
        const double t0 = nowMs();
        pagh::SeqDb contigDb = fromPacked(ctgs, ctg_names, "ctg", 0);
        pagh::SeqDb refDb = fromPacked(refs, ref_names, "ref", 1);
        pagh::PositionMapper ctgMapper(contigDb), refMapper(refDb);
        std::set<std::pair<std::string, bool>> ctgSet;
        for (std::uint64_t i = 0; i < ctgs->n_seqs; ++i) {
            const int32_t o = ctg_orient[i];
            if (o == PAG_ORIENT_FORWARD || o == PAG_ORIENT_BOTH) ctgSet.emplace(contigDb.name(i), true);
            if (o == PAG_ORIENT_REVERSE || o == PAG_ORIENT_BOTH) ctgSet.emplace(contigDb.name(i), false);
        }
        // (no key: nothing is kept between calls, and callers without a handle of their own do not share storage)
        std::unique_ptr<HandleCache> local;
        if (!cache_key) local.reset(new HandleCache());
        HandleCache &hc = cache_key ? cacheOf(cache_key) : *local;
        std::vector<std::pair<const pag_path_node *, std::uint64_t>> views(2 * ctgs->n_seqs, {nullptr, 0});
        for (std::uint64_t c = 0; c < 2 * ctgs->n_seqs; ++c)
            if (paths[c] && path_len[c]) views[c] = {paths[c], path_len[c]};
        const double tg0 = nowMs();
        pagh::buildPathGraph(views, k, hc.graph, hc.travelled, host_threads);
        const double t1 = nowMs();
        if (pagh::envTiming()) std::fprintf(stderr, "[timing] buildPathGraph %.1f ms\n", t1 - tg0);
        pagh::AssembleStats as;
        pagh::assemble(out_dir, prefix ? prefix : "0_", hc.graph, contigDb, refDb, ctgMapper, refMapper, ctgSet, epsilon * 2, 0.15,
                       0.90, min_len, ref_threads, host_threads, &as, true, hc.travelled);
        const double t2 = nowMs();
        if (stats) {
            stats->n_contigs = as.nContigs;
            stats->n_path_nodes = as.nPathNodes;
            stats->n_path_bases = as.nPathBases;
            stats->n_chains_emitted = as.nChains;
            stats->n_fasta_bases = as.nFastaBases;
            stats->path_checksum = as.pathChecksum;
            stats->ms_export = t1 - t0;
            stats->ms_traverse = t2 - t1;
            stats->ms_total = t2 - t0;
        }
        return PAG_OK;
    } catch (const std::exception &e) {
        setErr("pagh_assemble_paths: %s", e.what());
        return PAG_EFAULT;
    }
}

// pagh_traverse in two halves (see include/pagraph_host.h): the device traversal, then the host half on a thread of its own
static int traverse_begin(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names, const pag_seqs *refs,
                          const char *const *ref_names, const int32_t *ctg_orient, uint32_t ref_threads, uint64_t epsilon,
                          uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads, bool wait_at_once) {
    if (!g || !ctgs || !refs || !ctg_orient || !out_dir) return PAG_EINVAL;
    HandleCache &hc = cacheOf(g);
    const double t0 = nowMs();
    pag_travel_params tp{};
    tp.ref_threads = ref_threads;
    tp.deviation = epsilon * 2;
    tp.error_rate = 0.15;
    tp.start_split = 0.90;
    tp.min_len = min_len;
    // the traversal's view of the new graph (successor records: device work, this thread only waits) still runs beside the
    // previous block's host half ...
    double msPrep = 0;
    int rc = pag_travel_prepare_for(g, ctgs, ctg_orient, refs->len, refs->n_seqs, &tp, &msPrep);
    if (rc != PAG_OK) {
        setErr("pag_travel_prepare: %s", pag_last_error());
        return rc;
    }
    // ... which reads travel sequences in pinned memory that the walks of the next pag_travel reuse: wait for it here
    const double tA = nowMs();
    const bool overlapped = hc.worker.joinable();
    if (overlapped) hc.worker.join();
    const double tB = nowMs();
    if (hc.pending && hc.rc != PAG_OK) {  // (never collected: its error is this call's)
        hc.pending = false;
        setErr("%s", hc.error.c_str());
        return hc.rc;
    }
    hc.tst = pag_travel_stats{};
    rc = pag_travel(g, ctgs, ctg_orient, refs->len, refs->n_seqs, &tp, &hc.tst);
    const double tC = nowMs();
    hc.tst.ms_compact += msPrep;
    hc.tst.ms_total += msPrep;
    const pag_travel_stats &tst = hc.tst;
    if (pagh::envTiming())
        std::fprintf(stderr, "[timing] pag_travel total %.1f ms compact %.1f ms walk %.1f ms rounds %llu jobs %llu steps %llu classify %llu probes %llu records %llu\n", tst.ms_total,
                     tst.ms_compact, tst.ms_walk, (unsigned long long)tst.rounds, (unsigned long long)tst.jobs,
                     (unsigned long long)tst.walk_steps, (unsigned long long)tst.classify_calls, (unsigned long long)tst.probes,
                     (unsigned long long)tst.records);
    if (rc != PAG_OK) {
        setErr("pag_travel: %s", pag_last_error());
        return rc;
    }
    hc.paths.assign(2 * ctgs->n_seqs, nullptr);
    hc.lens.assign(2 * ctgs->n_seqs, 0);
    for (std::uint64_t c = 0; c < ctgs->n_seqs; ++c)
        for (int rev = 0; rev < 2; ++rev) {
            std::uint64_t len = 0;
            const pag_path_node *p = pag_travel_path_oriented(g, c, rev == 0, &len);
            if (p && len) {
                hc.paths[2 * c + rev] = p;
                hc.lens[2 * c + rev] = len;
            }
        }
    hc.tBegin = t0;
    hc.tHost = nowMs();
    if (pagh::envTiming())
        std::fprintf(stderr, "[timing] traverse_begin: successor stage %.1f ms, wait for the previous host half %.1f ms, pag_travel %.1f ms, views %.1f ms\n", tA - t0, tB - tA,
                     tC - tB, hc.tHost - tC);
    hc.pending = true;
    hc.rc = PAG_OK;
    hc.stats = pagh_traverse_stats{};
    const std::string outDir(out_dir), pre(prefix ? prefix : "0_");
    HandleCache *h = &hc;
    // While the host half runs beside the next block's device work, its pool stays small: a burst of 64 threads uses up a
    // container's CPU quota for the period (measured on the GPU box, 16-CPU quota: the caller's launches of the next block's
    // sort stalled, 24 -> 74 ms).  pagh_traverse() itself (begin + end back to back) keeps the full pool.
    unsigned poolThreads = host_threads;
    if (!wait_at_once && poolThreads == 0) {
        // (measured at configs[1] on the GPU box, 16-CPU quota.  Round 3: ms per block 512 / 453 / 443 / 437 / 430 with 6 / 8 / 12 / 16 / 24
        // threads, and 12 was chosen to leave the caller's thread and the walks' control thread their CPUs.  Round 5: the device work the
        // host half runs beside — the next block's prepare, build and successor stage — has shrunk to ~163 ms and the host half on 12
        // threads takes 157-164 ms beside it: the next walks WAITED for it, 10-21 ms per block.  8 / 10 / 12 / 14 / 16 / 20 / 24 / 32
        // threads: wait 72 / 39 / 21 / 3 / 0-6 / 1 / 0 / 2 ms, host half 217 / 192 / 157-164 / 145 / 126-143 / 127 / 114 / 102 ms
        // (profiles/r05_overlap_threads_probe.txt); the walks themselves begin after the join and are not touched by the pool)
        static const unsigned asked = pagh::envOverlapThreads();
        poolThreads = asked ? asked : std::min(20u, std::max(4u, pagh::usableCpus()));  // (ranks of a node share its cores: host_threads.hpp)
    }
    hc.worker = std::thread([=]() {
        h->rc = pagh_assemble_paths(g, k, ctgs, ctg_names, refs, ref_names, ctg_orient, h->paths.data(), h->lens.data(), ref_threads, epsilon,
                                    min_len, outDir.c_str(), pre.c_str(), poolThreads, &h->stats);
        if (h->rc != PAG_OK) h->error = g_err;  // (this thread's message)
    });
    return PAG_OK;
}

int pagh_traverse_begin(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names, const pag_seqs *refs,
                        const char *const *ref_names, const int32_t *ctg_orient, uint32_t ref_threads, uint64_t epsilon,
                        uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads) {
    return traverse_begin(g, k, ctgs, ctg_names, refs, ref_names, ctg_orient, ref_threads, epsilon, min_len, out_dir, prefix, host_threads, false);
}

int pagh_traverse_end(pag_graph *g, pagh_traverse_stats *stats) {
    if (!g) return PAG_EINVAL;
    HandleCache &hc = cacheOf(g);
    if (hc.worker.joinable()) hc.worker.join();
    if (!hc.pending) {
        setErr("pagh_traverse_end: no traversal has been begun on this handle");
        return PAG_EINVAL;
    }
    hc.pending = false;
    if (hc.rc != PAG_OK) {
        setErr("%s", hc.error.c_str());
        return hc.rc;
    }
    if (stats) {
        *stats = hc.stats;
        const double dev = hc.tHost - hc.tBegin;
        stats->ms_export += dev;  // (device traversal included, as before)
        stats->ms_total += dev;
        stats->ms_successors = hc.tst.ms_compact;
        stats->ms_walk = hc.tst.ms_walk;
        stats->walk_rounds = hc.tst.rounds;
        stats->walk_jobs = hc.tst.jobs;
        stats->walk_steps = hc.tst.walk_steps;
        stats->walk_classifications = hc.tst.classify_calls;
    }
    return PAG_OK;
}

int pagh_traverse(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names, const pag_seqs *refs,
                  const char *const *ref_names, const int32_t *ctg_orient, uint32_t ref_threads, uint64_t epsilon,
                  uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads, pagh_traverse_stats *stats) {
    const int rc = traverse_begin(g, k, ctgs, ctg_names, refs, ref_names, ctg_orient, ref_threads, epsilon, min_len, out_dir, prefix,
                                  host_threads, true);
    return rc != PAG_OK ? rc : pagh_traverse_end(g, stats);
}

}  // extern "C"
