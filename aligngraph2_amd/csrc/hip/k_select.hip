// k_select.hip — pag_shard_select: the part of an owner's finished slice that ONE rank of a sharded build needs for the
// traversals it was dealt (SURVEY.md 8e level 2, the step after K2-K4).
//
// Contigs are traversed independently (PAssembly.cpp:30-79) and a traversal only ever examines
//   * vertices on the traversed strand of its contig,
//   * vertices with a contig coordinate elsewhere ONLY as leap targets, and those are dropped unless they lie in the first
//     (1 - startSplit) of their contig strand (classifySuccessors, PAlgorithm.tcc:60-67) — anything else with a contig
//     coordinate outside the own strand can be left out without changing a single classification,
//   * vertices without a contig coordinate near the reference stretch the contig maps to (every successor lies within
//     step + deviation or step x (1 + error rate) of its source on the contig or on the reference, checkPosition
//     PABruijnGraph.cpp:143-165).
// So a rank takes, from every owner, the vertices whose contig coordinate falls into its contig intervals, and the
// coordinate-free vertices inside its reference bands (bands = where its contigs map, plus a halo; a walk that gets within
// a successor's reach of an open band end is detected — TravGraph::incomplete — never silently wrong), with all edges of
// the k-mers that keep a vertex.  The clustering itself was done by the owner on ALL tuples of a k-mer: the selection
// only drops finished vertices.  The result has the layout of a slice (pag_shard_slice), so pag_shard_import takes it.
#include <algorithm>
#include <vector>

#include "pag_graph_impl.hpp"

namespace pagdev {
namespace {

constexpr int SEL_SLOT0 = 192;

__device__ __forceinline__ bool in_intervals(const uint32_t *__restrict__ iv, uint32_t n, uint32_t x) {
    if (!n) return false;
    uint32_t lo = 0, hi = n;  // last interval with lo <= x
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (iv[2 * mid] <= x) lo = mid;
        else hi = mid;
    }
    return x >= iv[2 * lo] && x < iv[2 * lo + 1];
}

// per k-mer segment head: which of its leaders stay (keep[] over the slots), how many; the k-mer's bit in `codes`
__global__ void sel_vertices(const uint32_t *__restrict__ tkey, const uint64_t *__restrict__ tval, const uint32_t *__restrict__ tseg, uint64_t T,
                             const uint32_t *__restrict__ civ, uint32_t n_civ, const uint32_t *__restrict__ riv, uint32_t n_riv,
                             uint32_t *__restrict__ keep, uint64_t *__restrict__ codes, unsigned long long *__restrict__ n_nodes) {
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t kx = tkey[i];
        if (i != 0 && tkey[i - 1] == kx) continue;
        const uint32_t len = tseg[i];
        uint32_t kept = 0;
        for (uint32_t l = 0; l < len; ++l) {
            const uint64_t p = tval[i + l];
            const uint32_t c = (uint32_t)(p >> 32), r = (uint32_t)p;
            const bool k = c != 0u ? in_intervals(civ, n_civ, c) : in_intervals(riv, n_riv, r);
            keep[i + l] = k ? 1u : 0u;
            kept += k ? 1u : 0u;
        }
        if (kept) {
            atomicOr((unsigned long long *)&codes[kx >> 6], 1ull << (kx & 63u));
            ++mine;
        }
    }
    if (mine) atomicAdd(n_nodes, mine);
}
__global__ void sel_write_vertices(const uint32_t *__restrict__ tkey, const uint64_t *__restrict__ tval, const uint32_t *__restrict__ tseg,
                                   const uint16_t *__restrict__ tcnt, uint64_t T, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ pos,
                                   uint32_t *__restrict__ okey, uint64_t *__restrict__ oval, uint32_t *__restrict__ oseg, uint16_t *__restrict__ ocnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t kx = tkey[i];
        if (i != 0 && tkey[i - 1] == kx) continue;
        const uint32_t len = tseg[i];
        uint32_t kept = 0;
        uint64_t first = 0;
        for (uint32_t l = 0; l < len; ++l) {
            if (!keep[i + l]) continue;
            const uint64_t o = pos[i + l];
            if (!kept) first = o;
            okey[o] = kx;
            oval[o] = tval[i + l];
            ocnt[o] = tcnt[i + l];
            oseg[o] = SEG_LEADER | kept;  // (a leader slot `kept` behind the head of the selected segment: the layout of K3, pag_device.hpp)
            ++kept;
        }
        if (kept) oseg[first] = kept;
    }
}
// edge segments of the k-mers that keep a vertex
__global__ void sel_edges(const uint32_t *__restrict__ ekey, const uint32_t *__restrict__ eseg, uint64_t E, const uint64_t *__restrict__ codes,
                          uint32_t *__restrict__ keep) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t kx = ekey[j];
        if (j != 0 && ekey[j - 1] == kx) continue;
        if (!((codes[kx >> 6] >> (kx & 63u)) & 1ull)) continue;
        const uint32_t len = eseg[j];
        for (uint32_t l = 0; l < len; ++l) keep[j + l] = 1u;
    }
}
__global__ void sel_write_edges(const uint32_t *__restrict__ ekey, const uint64_t *__restrict__ eval, const uint32_t *__restrict__ eseg, uint64_t E,
                                const uint32_t *__restrict__ keep, const uint64_t *__restrict__ pos, uint32_t *__restrict__ okey, uint64_t *__restrict__ oval,
                                uint32_t *__restrict__ oseg) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (uint64_t)gridDim.x * blockDim.x) {
        if (!keep[j]) continue;
        const uint64_t o = pos[j];
        okey[o] = ekey[j];
        oval[o] = eval[j];
        oseg[o] = eseg[j];  // (a kept segment is kept whole: its head keeps its count, the others are 0)
    }
}
unsigned sel_grid(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 16) + (n == 0); }

}  // namespace
}  // namespace pagdev

using namespace pagdev;

extern "C" int pag_shard_select(pag_graph *g, const pag_region *r, pag_shard_slice *out) {
    if (!g || !r || !out || (r->n_ctg_iv && !r->ctg_iv) || (r->n_ref_iv && !r->ref_iv)) return PAG_EINVAL;
    if (!g->tkey && g->n_t) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const uint64_t T = g->n_t, E = g->n_e;
    const uint64_t n_words = ((1ull << (2 * g->k)) + 63) / 64;
    int rc, slot = SEL_SLOT0;
    DevBuf b_civ(g, slot++), b_riv(g, slot++), b_keep(g, slot++), b_pos(g, slot++), b_codes(g, slot++), b_tmp(g, slot++), b_cnt(g, slot++), b_tk(g, slot++),
        b_tv(g, slot++), b_ts(g, slot++), b_tc(g, slot++), b_ek(g, slot++), b_ev(g, slot++), b_es(g, slot++);
    const uint64_t M = std::max(T, E);
    if ((rc = b_civ.alloc(r->n_ctg_iv * 8 + 16)) || (rc = b_riv.alloc(r->n_ref_iv * 8 + 16)) || (rc = b_keep.alloc((M + 2) * 4)) ||
        (rc = b_pos.alloc((M + 3) * 8)) || (rc = b_codes.alloc(n_words * 8)) || (rc = b_tmp.alloc(scan_tmp_bytes(M + 2) + 64)) || (rc = b_cnt.alloc(64)))
        return rc;
    if (r->n_ctg_iv) PAG_HIP_TRY(hipMemcpyAsync(b_civ.p, r->ctg_iv, r->n_ctg_iv * 8, hipMemcpyHostToDevice, s));
    if (r->n_ref_iv) PAG_HIP_TRY(hipMemcpyAsync(b_riv.p, r->ref_iv, r->n_ref_iv * 8, hipMemcpyHostToDevice, s));
    PAG_HIP_TRY(hipMemsetAsync(b_keep.p, 0, (M + 2) * 4, s));
    PAG_HIP_TRY(hipMemsetAsync(b_codes.p, 0, n_words * 8, s));
    PAG_HIP_TRY(hipMemsetAsync(b_cnt.p, 0, 64, s));
    uint64_t n_nodes = 0, n_pos = 0, n_edges = 0;
    // ---- vertices
    if (T) sel_vertices<<<dim3(sel_grid(T)), dim3(256), 0, s>>>(g->tkey, g->tval, g->tseg, T, b_civ.as<uint32_t>(), (uint32_t)r->n_ctg_iv, b_riv.as<uint32_t>(),
                                                                (uint32_t)r->n_ref_iv, b_keep.as<uint32_t>(), b_codes.as<uint64_t>(), b_cnt.as<unsigned long long>());
    if ((rc = scan_u32_to_u64(b_keep.as<uint32_t>(), b_pos.as<uint64_t>(), T + 1, nullptr, b_tmp.p, s))) return rc;
    PAG_HIP_TRY(hipMemcpyAsync(&n_pos, b_pos.as<uint64_t>() + T, 8, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipMemcpyAsync(&n_nodes, b_cnt.p, 8, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    if ((rc = b_tk.alloc((n_pos + 1) * 4)) || (rc = b_tv.alloc((n_pos + 1) * 8)) || (rc = b_ts.alloc((n_pos + 1) * 4)) || (rc = b_tc.alloc((n_pos + 1) * 2))) return rc;
    if (T) sel_write_vertices<<<dim3(sel_grid(T)), dim3(256), 0, s>>>(g->tkey, g->tval, g->tseg, g->tcnt, T, b_keep.as<uint32_t>(), b_pos.as<uint64_t>(),
                                                                      b_tk.as<uint32_t>(), b_tv.as<uint64_t>(), b_ts.as<uint32_t>(), b_tc.as<uint16_t>());
    // ---- edges of the k-mers that keep a vertex
    PAG_HIP_TRY(hipMemsetAsync(b_keep.p, 0, (M + 2) * 4, s));
    if (E) sel_edges<<<dim3(sel_grid(E)), dim3(256), 0, s>>>(g->ekey, g->eseg, E, b_codes.as<uint64_t>(), b_keep.as<uint32_t>());
    if ((rc = scan_u32_to_u64(b_keep.as<uint32_t>(), b_pos.as<uint64_t>(), E + 1, nullptr, b_tmp.p, s))) return rc;
    PAG_HIP_TRY(hipMemcpyAsync(&n_edges, b_pos.as<uint64_t>() + E, 8, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    if ((rc = b_ek.alloc((n_edges + 1) * 4)) || (rc = b_ev.alloc((n_edges + 1) * 8)) || (rc = b_es.alloc((n_edges + 1) * 4))) return rc;
    if (E) sel_write_edges<<<dim3(sel_grid(E)), dim3(256), 0, s>>>(g->ekey, g->eval, g->eseg, E, b_keep.as<uint32_t>(), b_pos.as<uint64_t>(), b_ek.as<uint32_t>(),
                                                                   b_ev.as<uint64_t>(), b_es.as<uint32_t>());
    PAG_HIP_TRY(hipStreamSynchronize(s));
    PAG_HIP_TRY(hipGetLastError());
    *out = pag_shard_slice{};
    out->n_t = n_pos;
    out->n_e = n_edges;
    out->tkey = b_tk.as<uint32_t>();
    out->tval = b_tv.as<uint64_t>();
    out->tseg = b_ts.as<uint32_t>();
    out->tcnt = b_tc.as<uint16_t>();
    out->ekey = b_ek.as<uint32_t>();
    out->eval = b_ev.as<uint64_t>();
    out->eseg = b_es.as<uint32_t>();
    out->stats = g->stats;  // the owner's share of the count lines ...
    out->stats.n_nodes = n_nodes;  // ... and what of its slice goes to this rank
    out->stats.n_pos = n_pos;
    out->stats.n_uniq_edges = n_edges;
    return PAG_OK;
}

// after pag_shard_import of selected slices: the reference bands of the region (for the detection of walks that leave it)
extern "C" int pag_shard_set_region(pag_graph *g, const pag_region *r) {
    if (!g || !r || (r->n_ref_iv && (!r->ref_iv || !r->ref_open))) return PAG_EINVAL;
    g->regional = true;
    g->region_ref_iv.assign(r->ref_iv, r->ref_iv + 2 * r->n_ref_iv);
    g->region_ref_open.assign(r->ref_open, r->ref_open + 2 * r->n_ref_iv);
    g->tg_ready = false;
    return PAG_OK;
}

// The build stages' device memory (inputs, tuple streams, sort and segment scratch, the owner's slice, the selections:
// pool slots 0 .. 51 and 192 ..) handed back once the rank has imported its region: at BASELINE configs[2] that is ~70 GB
// per GPU the traversal needs.  The imported graph (slots 52 .. 58) and the traversal's slots stay.
extern "C" int pag_shard_release_build(pag_graph *g) {
    if (!g) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    PAG_HIP_TRY(hipStreamSynchronize(g->stream));
    auto drop = [&](int a, int b) {
        for (int i = a; i < b; ++i) {
            pag_graph::Slot &sl = g->pool[i];
            if (sl.p && sl.p != (void *)g->tkey && sl.p != (void *)g->tval && sl.p != (void *)g->tseg && sl.p != (void *)g->tcnt && sl.p != (void *)g->ekey &&
                sl.p != (void *)g->eval && sl.p != (void *)g->eseg) {
                hipFree(sl.p);
                sl.p = nullptr;
                sl.cap = 0;
            }
        }
    };
    drop(0, 52);
    drop(128, 256);
    return PAG_OK;
}
