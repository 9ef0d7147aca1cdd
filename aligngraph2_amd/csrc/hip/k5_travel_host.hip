// k5_travel_host.hip — pag_travel: the per-round control of PAlgorithm::travelSequence
// (reference PAGraph/src/tools/graph/PAlgorithm.cpp:144-426) around the device kernels of k5_travel.hip.
//
// All selected contigs advance in lock step: round r launches one walker wave per (contig, seed), the
// host then applies the reference's choice rule per contig (first leaping walk, else the longest; seeds
// after the first must reach minLen), splices the walk into the running path (appendSeq), records it in
// the contig's global visited set (device hash set + host mirror), checks the repeat / leap stop rules
// and prepares the next seeds (window scan on the device; ordering by edit distance with the same
// unstable std::sort as the reference on the host).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "pag_graph_impl.hpp"
#include "walk_config.hpp"
#include "walk_stitch.hpp"
#include "walker_grid.hpp"

using namespace pagdev;

namespace {

#include "trav_prepare_host.hpp"
#include "walk_session_jobs.hpp"
#include "walk_session_rounds.hpp"
#include "walk_session_loop.hpp"


}  // namespace


extern "C" {

// test hooks (host code only, no device needed): the library's own copies of PositionMapper and editDistance against the
// reference's function-level golden tables (tests/test_function_goldens.py)
uint64_t pag_debug_edit_distance(const char *a, const char *b) { return edit_distance(a, b); }
uint64_t pag_debug_mapper_d2s(const uint32_t *len, uint64_t n, int64_t idx, int64_t pos) { return Mapper(len, n).dualToSingle(idx, pos); }
void pag_debug_mapper_s2d(const uint32_t *len, uint64_t n, uint64_t single, int64_t *idx, int64_t *pos) {
    auto d = Mapper(len, n).singleToDual(single);
    *idx = d.first;
    *pos = d.second;
}
uint64_t pag_debug_mapper_extra(const uint32_t *len, uint64_t n) { return Mapper(len, n).starts.back(); }

// g->paths[2 * contig + (reverse ? 1 : 0)]
const pag_path_node *pag_travel_path_oriented(const pag_graph *g, uint64_t ctg_index, int forward, uint64_t *len) {
    const uint64_t slot = 2 * ctg_index + (forward ? 0 : 1);
    if (!g || slot >= g->path_valid.size() || !g->path_valid[slot]) {
        if (len) *len = 0;
        return nullptr;
    }
    if (len) *len = g->path_len[slot];
    if (slot < g->path_ptr.size() && g->path_ptr[slot]) return g->path_ptr[slot];
    return g->path_store + g->path_off[slot];
}

const pag_path_node *pag_travel_path(const pag_graph *g, uint64_t ctg_index, uint64_t *len) {
    if (g && 2 * ctg_index + 1 < g->path_valid.size() && !g->path_valid[2 * ctg_index]) return pag_travel_path_oriented(g, ctg_index, 0, len);
    return pag_travel_path_oriented(g, ctg_index, 1, len);
}

// test hooks: the successor records of the prepared traversal graph (after pag_travel_prepare), copied to the host:
// succ_off[n_pos + 1], then n_succ records of 16 bytes (target, contig coordinate, step | grade | flags | count, target's offset)
int pag_debug_succ_sizes(const pag_graph *g, uint64_t *n_pos, uint64_t *n_succ) {
    if (!g || !g->tg_ready || !n_pos || !n_succ) return PAG_EINVAL;
    *n_pos = g->tg.n_pos;
    *n_succ = g->tg.n_succ;
    return PAG_OK;
}
int pag_debug_succ(const pag_graph *g, uint32_t *succ_off, void *recs) {
    if (!g || !g->tg_ready || !succ_off || !recs) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    PAG_HIP_TRY(hipMemcpy(succ_off, g->tg.succ_off, (g->tg.n_pos + 1) * 4, hipMemcpyDeviceToHost));
    PAG_HIP_TRY(hipMemcpy(recs, g->tg.succ, g->tg.n_succ * sizeof(SuccRec), hipMemcpyDeviceToHost));
    return PAG_OK;
}

// ... and which vertex of the finished graph every id of the view is: its k-mer code and its position (ctg << 32 | ref) —
// (code, position) names a vertex of the reference's graph uniquely (the clustered positions of a k-mer are pairwise
// distinct), which is how tests/test_gpu_succ_golden.py holds every record against the reference's successors() dump
int pag_debug_trav_vertices(const pag_graph *g, uint32_t *code, uint64_t *pos) {
    if (!g || !g->tg_ready || !code || !pos) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    const TravGraph &G = g->tg;
    std::vector<uint32_t> uold(G.n_pos), vnode(G.n_pos), ncode(G.n_nodes);
    PAG_HIP_TRY(hipMemcpy(uold.data(), G.uold, G.n_pos * 4, hipMemcpyDeviceToHost));
    PAG_HIP_TRY(hipMemcpy(vnode.data(), G.vnode, G.n_pos * 4, hipMemcpyDeviceToHost));
    PAG_HIP_TRY(hipMemcpy(ncode.data(), G.ncode, G.n_nodes * 4, hipMemcpyDeviceToHost));
    PAG_HIP_TRY(hipMemcpy(pos, G.upos, G.n_pos * 8, hipMemcpyDeviceToHost));
    for (uint64_t u = 0; u < G.n_pos; ++u) code[u] = ncode[vnode[uold[u]]];
    return PAG_OK;
}

// PABruijnGraph::successors for one vertex of the prepared view (PABruijnGraph.cpp:370-373 -> searchSuccessors :167-197), see pagraph_hip.h
int64_t pag_successors(const pag_graph *g, uint32_t code, uint64_t pos, pag_succ *out, uint64_t cap) {
    if (!g || !g->tg_ready || (!out && cap)) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    if (g->view_pruned || g->regional) {
        set_error("pag_successors: the prepared view was cut for given traversals (pag_travel_prepare_for / a regional graph): its lists are restricted to what those traversals can examine");
        return PAG_ERANGE;
    }
    void *d = nullptr;
    const uint64_t room = std::min<uint64_t>(cap, g->tg.n_succ);
    PAG_HIP_TRY(hipMalloc(&d, 16 + room * sizeof(pag_succ)));
    unsigned long long *d_n = (unsigned long long *)d;
    void *d_recs = (char *)d + 16;
    int rc = trav_successors_of(g->tg, code, pos, d_recs, room, d_n, g->stream);
    unsigned long long n = 0;
    if (rc == PAG_OK && (hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, g->stream) != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess)) rc = PAG_EFAULT;
    if (rc == PAG_OK && n == ~0ull) {
        set_error("pag_successors: the graph holds no vertex with that k-mer and position");
        rc = PAG_EINVAL;
    } else if (rc == PAG_OK && n == ~1ull) {
        set_error("pag_successors: that vertex's list is a marker record");
        rc = PAG_ERANGE;
    } else if (rc == PAG_OK && std::min<uint64_t>(n, room) &&
               hipMemcpy(out, d_recs, std::min<uint64_t>(n, room) * sizeof(pag_succ), hipMemcpyDeviceToHost) != hipSuccess) {
        rc = PAG_EFAULT;
    }
    hipFree(d);
    return rc == PAG_OK ? (int64_t)n : rc;
}

// the first part of pag_travel on its own (the caller may have other work for the host between it and the walks)
int pag_travel_prepare(pag_graph *g, const pag_seqs *ctgs, const uint32_t *ref_len, uint64_t n_refs, const pag_travel_params *prm, double *ms) {
    return pag_travel_prepare_for(g, ctgs, nullptr, ref_len, n_refs, prm, ms);
}
// ... for the traversals pag_travel will be asked for (orient as pag_travel takes it; NULL: any): the view then holds what
// those traversals can examine and nothing else (trav_view_region above)
int pag_travel_prepare_for(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
                           const pag_travel_params *prm, double *ms) {
    if (!g || !ctgs || !prm || (!ref_len && n_refs)) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    TravGraph G{};
    return trav_prepare_graph(g, ctgs->len, ctgs->n_seqs, ref_len, n_refs, prm->deviation, prm->error_rate, &G, ms, orient, prm->start_split);
}
int pag_travel_view_sizes(const pag_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges, uint64_t *n_succ, int *cut, uint64_t *fallbacks) {
    if (!g || !g->tg_ready) return PAG_EINVAL;
    if (n_nodes) *n_nodes = g->view_counts[0];
    if (n_pos) *n_pos = g->view_counts[1];
    if (n_edges) *n_edges = g->view_counts[2];
    if (n_succ) *n_succ = g->tg.n_succ;
    if (cut) *cut = g->view_pruned ? 1 : 0;
    if (fallbacks) *fallbacks = g->view_fallbacks;
    return PAG_OK;
}

static int travel_once(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
                       const pag_travel_params *prm, pag_travel_stats *stats);
int pag_travel(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
               const pag_travel_params *prm, pag_travel_stats *stats) {
    if (!g || !ctgs || !orient || !prm || (!ref_len && n_refs)) return PAG_EINVAL;
    int rc = travel_once(g, ctgs, orient, ref_len, n_refs, prm, stats);
    if (rc == PAG_ERANGE && g->view_pruned && !g->regional) {
        // a walk examined a vertex whose successors this handle's own view left out (trav_view_region): nothing of that walk is
        // kept — the whole graph's view is built and every contig walked again (the outputs are those of the un-cut graph)
        if (WalkConfig::from_env().timing) std::fprintf(stderr, "[timing] a walk left the view: %s; walking again on the whole graph\n", pag_last_error());
        g->view_off = true;
        g->tg_ready = false;
        g->view_fallbacks += 1;
        rc = travel_once(g, ctgs, orient, ref_len, n_refs, prm, stats);
    }
    return rc;
}
static int travel_once(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
                       const pag_travel_params *prm, pag_travel_stats *stats) {
    WalkSession W(g, ctgs, orient, ref_len, n_refs, prm, stats);
    return W.run();
}


}  // extern "C"
