// k5_view.hip — the traversal graph: what of a finished graph the traversals of a handle can examine (the view), as a compact CSR
// with dense node / vertex ids, a 4^k-bit node bitmap + rank directory (code -> node id in two loads), and the vertices
// renumbered by contig coordinate.  Reference: the graph PAlgorithm walks is PABruijnGraph's per-k-mer node table
// (PAGraph/src/tools/graph/PABruijnGraph.hpp:90-131, KMerAdjNode.hpp:19-23); findAll (contig k-mers -> nodes) PABruijnGraph.cpp:339-353.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pag_device.hpp"
#include "pag_travel.hpp"
#include "trav_device.hpp"

namespace pagdev {

// =================================================================================================
// graph compaction
// =================================================================================================
// ---- a view that leaves out what no traversal of this handle can examine (trav_view_region, k5_travel_host.hip) --------
// [lo, hi) pairs, sorted and disjoint
__device__ __forceinline__ bool iv_contains(const uint32_t *__restrict__ iv, uint32_t n, uint32_t x, uint32_t *which = nullptr) {
    if (!n) return false;
    uint32_t lo = 0, hi = n;  // last interval with lo <= x
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (iv[2 * mid] <= x) lo = mid;
        else hi = mid;
    }
    if (which) *which = lo;
    return x >= iv[2 * lo] && x < iv[2 * lo + 1];
}
// lowest / highest reference coordinate among the positions whose contig coordinate lies in zone z (lo[z] preset to all
// ones, hi[z] to 0).  One thread per tuple slot: slots behind a segment's leaders still hold positions of the k-mer's
// reads (members of the clusters), which lie within epsilon of a leader — they widen nothing.
constexpr uint32_t ZONE_LDS = 2048;
__global__ void k_zone_bands(const uint64_t *__restrict__ tval, uint64_t T, const uint32_t *__restrict__ zones, uint32_t n_z,
                             uint32_t *__restrict__ lo, uint32_t *__restrict__ hi) {
    __shared__ uint32_t s_lo[ZONE_LDS], s_hi[ZONE_LDS], s_z[2 * ZONE_LDS];
    const bool lds = n_z <= ZONE_LDS;
    if (lds) {
        for (uint32_t z = threadIdx.x; z < n_z; z += blockDim.x) {
            s_lo[z] = 0xFFFFFFFFu;
            s_hi[z] = 0u;
            s_z[2 * z] = zones[2 * z];
            s_z[2 * z + 1] = zones[2 * z + 1];
        }
        __syncthreads();
    }
    const uint32_t z_first = n_z ? zones[0] : 0u, z_last = n_z ? zones[2 * n_z - 1] : 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = tval[i];
        const uint32_t c = (uint32_t)(p >> 32), r = (uint32_t)p;
        uint32_t z;
        if (c < z_first || c >= z_last || r == 0u) continue;
        if (!(lds ? iv_contains(s_z, n_z, c, &z) : iv_contains(zones, n_z, c, &z))) continue;  // (two searches: a pointer "LDS or global" means flat loads)
        if (lds) {
            atomicMin(&s_lo[z], r);
            atomicMax(&s_hi[z], r);
        } else {
            atomicMin(&lo[z], r);
            atomicMax(&hi[z], r);
        }
    }
    if (lds) {
        __syncthreads();
        for (uint32_t z = threadIdx.x; z < n_z; z += blockDim.x)
            if (s_hi[z] != 0u) {
                atomicMin(&lo[z], s_lo[z]);
                atomicMax(&hi[z], s_hi[z]);
            }
    }
}
// The view's vertices out of the tuple slots, in two sweeps over tiles of VC_TILE slots (round 5; until then a flags kernel, two
// full-length scans of u32 flags into u64 offsets and a compaction kernel that read all of it back: 46 GB and 15.6 ms at BASELINE
// configs[1] for 12 GB of work):
//   k_view_mark   per slot: keep = the slot holds a vertex the view takes — one with a contig coordinate by that coordinate,
//                 one without by its reference coordinate; every vertex when there are no tables —, first = it is the first
//                 such slot of its k-mer segment (= a node).  Left behind as the ballots of every wave and round (2 bits per
//                 slot) and as the two counts of every tile.
//   (exclusive prefix over the tiles: two scans of T / 2048 counters)
//   k_view_write  the ballots again, their prefix inside the tile, the vertices / nodes written out.
// One thread per tuple slot (the slots say whether they hold a leader and how far behind their segment's head they lie: K3's
// seg_len layout, pag_device.hpp).  The interval tables are searched in LDS (from global memory the ~8 dependent loads per
// search were the whole cost: 34 ms at BASELINE configs[1] with a thread per segment head).  "First of its segment" comes from
// the tile's own prefix of the keep flags — no kept slot between the segment's head and this one —: a vertex without a contig
// coordinate sorts first in its segment and is what the view mostly leaves out, so a backward scan over the slots from the
// head (until round 5) ran its full length for every kept vertex behind one: most of the 7.8 ms of the flags kernel.
constexpr uint32_t PRUNE_LDS = 4096;  // interval ends (u32) the block keeps in LDS
constexpr uint32_t VC_T = 256, VC_R = 8, VC_TILE = VC_T * VC_R, VC_W = VC_T / 64;
__global__ __launch_bounds__(VC_T) void k_view_mark(const uint32_t *__restrict__ tkey, const uint64_t *__restrict__ tval, const uint32_t *__restrict__ tseg,
                                                    uint64_t T, const uint32_t *__restrict__ civ, uint32_t n_civ, const uint32_t *__restrict__ riv,
                                                    uint32_t n_riv, int whole, uint64_t *__restrict__ ballots, uint32_t *__restrict__ tile_first,
                                                    uint32_t *__restrict__ tile_keep, uint64_t n_tiles) {
    __shared__ uint32_t s_iv[PRUNE_LDS];
    __shared__ uint32_t s_ck[VC_R][VC_W], s_pk[VC_R][VC_W], s_cf[VC_R][VC_W];
    __shared__ uint16_t s_pre[VC_TILE];  // kept slots of the tile before this one
    const bool lds = !whole && 2u * (n_civ + n_riv) <= PRUNE_LDS;
    if (lds) {
        for (uint32_t x = threadIdx.x; x < 2u * n_civ; x += blockDim.x) s_iv[x] = civ[x];
        for (uint32_t x = threadIdx.x; x < 2u * n_riv; x += blockDim.x) s_iv[2u * n_civ + x] = riv[x];
        __syncthreads();
    }
    // (two searches, not one through a pointer that is "LDS or global": that pointer is generic to the compiler, and the eight
    // dependent loads of a search became flat loads — the slow way into LDS — 81 of them in this kernel)
    auto inside = [&](uint64_t p) {
        const uint32_t c = (uint32_t)(p >> 32), r = (uint32_t)p;
        if (whole) return true;
        if (lds) return c != 0u ? iv_contains(s_iv, n_civ, c) : iv_contains(s_iv + 2u * n_civ, n_riv, r);
        return c != 0u ? iv_contains(civ, n_civ, c) : iv_contains(riv, n_riv, r);
    };
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t tile_base = tile * VC_TILE;
        uint32_t kb = 0;          // bit r: the slot of round r is kept
        uint32_t off_r[VC_R];     // ... how far behind its segment's head it lies
        uint64_t bkr[VC_R];
#pragma unroll
        for (uint32_t r = 0; r < VC_R; ++r) {
            const uint64_t i = tile_base + (uint64_t)r * VC_T + threadIdx.x;
            bool kk = false;
            off_r[r] = 0;
            if (i < T) {
                const uint32_t kx = tkey[i], v = tseg[i];
                const bool head = i == 0 || tkey[i - 1] != kx;
                const bool leader = head ? v != 0u : (v & SEG_LEADER) != 0u;
                if (leader) {
                    kk = inside(tval[i]);
                    off_r[r] = head ? 0u : (v & ~SEG_LEADER);
                }
            }
            bkr[r] = __ballot(kk);
            kb |= kk ? 1u << r : 0u;
            if (lane == 0) s_ck[r][w] = (uint32_t)__popcll(bkr[r]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {  // exclusive prefix over (round, wave) = slot order, tile total
            uint32_t ak = 0;
            for (uint32_t r = 0; r < VC_R; ++r)
                for (uint32_t ww = 0; ww < VC_W; ++ww) {
                    s_pk[r][ww] = ak;
                    ak += s_ck[r][ww];
                }
            tile_keep[tile] = ak;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < VC_R; ++r)
            s_pre[r * VC_T + threadIdx.x] = (uint16_t)(s_pk[r][w] + (uint32_t)__popcll(bkr[r] & lanemask_lt()));
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < VC_R; ++r) {
            const uint32_t d = r * VC_T + threadIdx.x;  // slot inside the tile
            bool ff = false;
            if ((kb >> r) & 1u) {
                const uint32_t off = off_r[r];
                if (off <= d) {
                    ff = s_pre[d] == s_pre[d - off];  // no kept slot in [head, this one)
                } else {  // the segment began in an earlier tile: the slots before this tile by their own test (one segment per tile)
                    ff = s_pre[d] == 0u;
                    const uint64_t i = tile_base + d;
                    for (uint64_t j = i - off; j < tile_base && ff; ++j) {
                        const uint32_t kx = tkey[j], v = tseg[j];
                        const bool head = j == 0 || tkey[j - 1] != kx;
                        const bool leader = head ? v != 0u : (v & SEG_LEADER) != 0u;
                        ff = !(leader && inside(tval[j]));
                    }
                }
            }
            const uint64_t bf = __ballot(ff);
            if (lane == 0) {
                s_cf[r][w] = (uint32_t)__popcll(bf);
                const uint64_t at = ((tile * VC_R + r) * VC_W + w) * 2u;
                ballots[at] = bkr[r];
                ballots[at + 1] = bf;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t af = 0;
            for (uint32_t r = 0; r < VC_R; ++r)
                for (uint32_t ww = 0; ww < VC_W; ++ww) af += s_cf[r][ww];
            tile_first[tile] = af;
        }
    }
}
__global__ __launch_bounds__(VC_T) void k_view_write(const uint32_t *__restrict__ tkey, const uint64_t *__restrict__ tval, const uint16_t *__restrict__ tcnt,
                                                     const uint64_t *__restrict__ ballots, const uint64_t *__restrict__ base_first,
                                                     const uint64_t *__restrict__ base_keep, uint64_t n_tiles, TravGraph G) {
    __shared__ uint64_t s_b[VC_R * VC_W * 2];
    __shared__ uint32_t s_pk[VC_R][VC_W], s_pf[VC_R][VC_W];
    const uint32_t w = threadIdx.x >> 6;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x < VC_R * VC_W * 2) s_b[threadIdx.x] = ballots[tile * (VC_R * VC_W * 2) + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t ak = 0, af = 0;
            for (uint32_t r = 0; r < VC_R; ++r)
                for (uint32_t ww = 0; ww < VC_W; ++ww) {
                    s_pk[r][ww] = ak;
                    s_pf[r][ww] = af;
                    ak += (uint32_t)__popcll(s_b[(r * VC_W + ww) * 2]);
                    af += (uint32_t)__popcll(s_b[(r * VC_W + ww) * 2 + 1]);
                }
        }
        __syncthreads();
        const uint64_t bk0 = base_keep[tile], bf0 = base_first[tile];
#pragma unroll
        for (uint32_t r = 0; r < VC_R; ++r) {
            const uint64_t bk = s_b[(r * VC_W + w) * 2], bf = s_b[(r * VC_W + w) * 2 + 1];
            const uint64_t me = 1ull << lane_id();
            if (!(bk & me)) continue;
            const bool ff = (bf & me) != 0ull;
            const uint64_t i = tile * VC_TILE + (uint64_t)r * VC_T + threadIdx.x;
            const uint64_t p = bk0 + s_pk[r][w] + (uint32_t)__popcll(bk & lanemask_lt());
            const uint64_t n = bf0 + s_pf[r][w] + (uint32_t)__popcll(bf & lanemask_lt()) + (ff ? 1u : 0u) - 1u;
            G.vpos[p] = tval[i];
            G.vcnt[p] = tcnt[i];
            G.vnode[p] = (uint32_t)n;
            if (ff) {
                const uint32_t kx = tkey[i];
                G.ncode[n] = kx;
                G.npos_off[n] = (uint32_t)p;
                atomicOr((unsigned long long *)&G.bitmap[kx >> 6], 1ull << (kx & 63u));
            }
        }
        __syncthreads();
    }
}

__global__ void k_popc_words(const uint64_t *__restrict__ bitmap, uint64_t n_words, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)__popcll(bitmap[i]);
}

__global__ void k_narrow(const uint64_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)in[i];
}

__global__ void k_edge_counts(const uint32_t *__restrict__ ekey, const uint32_t *__restrict__ eseg, uint64_t E, TravGraph G,
                              uint32_t *__restrict__ necnt) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t kx = ekey[j];
        if (j != 0 && ekey[j - 1] == kx) continue;
        uint32_t n = node_of_code(G, kx);
        if (n != PAG_NONE) necnt[n] = eseg[j];
    }
}

// code -> (first position | number of positions << 32) of the k-mer's node, all ones: no node.  A direct table over the 4^k codes
// (2 GB at k = 14, scratch of the compaction): an edge's target then costs ONE random sector instead of the three dependent
// gathers of bitmap word, rank and position range (k_compact_edges: 9.9 -> ms at BASELINE configs[1], round 5); built
// from the node arrays, which are ascending in the code.  Larger k: no table, the three gathers.
__global__ void k_code_table(TravGraph G, uint64_t *__restrict__ tab) {
    for (uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; n < G.n_nodes; n += (uint64_t)gridDim.x * blockDim.x) {
        const U32x2 r = *(const U32x2 *)(G.npos_off + n);
        tab[G.ncode[n]] = (uint64_t)r.a[0] | ((uint64_t)(r.a[1] - r.a[0]) << 32);
    }
}

__global__ void k_compact_edges(const uint32_t *__restrict__ ekey, const uint64_t *__restrict__ eval,
                                const uint32_t *__restrict__ eseg, uint64_t E, TravGraph G, const uint64_t *__restrict__ tab) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t kx = ekey[j];
        if (j != 0 && ekey[j - 1] == kx) continue;
        uint32_t n = node_of_code(G, kx);
        if (n == PAG_NONE) continue;
        uint32_t dst = G.nedge_off[n], len = eseg[j];
        // (the edge carries what the successor kernels need of its target: where the target node's positions begin and how
        // many they are — one random sector less per edge in each of their two passes, see edge_target.  Four edges per
        // turn, their lookups in flight together)
        for (uint32_t l0 = 0; l0 < len; l0 += 4u) {
            uint64_t v4[4];
            uint32_t p04[4], q4[4];
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) v4[t] = l0 + t < len ? eval[j + l0 + t] : 0ull;
            if (tab) {
                uint64_t e4[4];
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) e4[t] = l0 + t < len ? tab[(uint32_t)(v4[t] >> 32)] : ~0ull;
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) {
                    p04[t] = (uint32_t)e4[t];
                    q4[t] = (uint32_t)(e4[t] >> 32);
                }
            } else {  // code -> bitmap word + rank -> position range: three dependent gathers
                uint32_t to4[4];
                U32x2 r4[4];
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) to4[t] = l0 + t < len ? node_of_code(G, (uint32_t)(v4[t] >> 32)) : PAG_NONE;
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) r4[t] = *(const U32x2 *)(G.npos_off + (to4[t] != PAG_NONE ? to4[t] : 0u));
#pragma unroll
                for (uint32_t t = 0; t < 4u; ++t) {
                    p04[t] = to4[t] != PAG_NONE ? r4[t].a[0] : PAG_NONE;
                    q4[t] = r4[t].a[1] - r4[t].a[0];
                }
            }
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) {
                const uint32_t l = l0 + t;
                if (l >= len) break;
                const uint32_t step = (((uint32_t)v4[t]) >> 1) & EDGE_STEP_MASK;
                if (p04[t] == PAG_NONE) {
                    G.eto[dst + l] = PAG_NONE;
                    G.estep[dst + l] = step;
                } else {
                    G.eto[dst + l] = p04[t];
                    G.estep[dst + l] = step | ((q4[t] < EDGE_Q_MANY ? q4[t] : EDGE_Q_MANY) << 24);
                }
            }
        }
    }
}

// contig strand k-mers -> node ids (PABruijnGraph::findAll).  One thread per k-mer start.
__global__ void k_ctg_nodes(const uint8_t *__restrict__ packed, const TravCtgNodesJob *__restrict__ jobs, uint32_t k, TravGraph G,
                            uint32_t *__restrict__ out_all) {
    const TravCtgNodesJob J = jobs[blockIdx.y];
    const uint32_t len = J.len;
    const bool forward = J.forward != 0;
    uint32_t *__restrict__ out = out_all + J.out_off;
    const uint32_t n_pos = len >= k ? len - k + 1 : 0;
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const uint32_t *words = (const uint32_t *)(packed + J.byte_off);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += gridDim.x * blockDim.x) {
        uint32_t a = forward ? i : len - k - i;  // first contig base (forward numbering) covered by the k-mer
        uint32_t w = a >> 4, sh = (a & 15u) * 2u;
        uint64_t W = (uint64_t)words[w] | ((uint64_t)words[w + 1] << 32);
        uint32_t x = (uint32_t)(W >> sh) & kmask;
        uint32_t code = forward ? (rev2(x) >> (32 - 2 * k)) : ((~x) & kmask);
        out[i] = node_of_code(G, code);
    }
}

// =================================================================================================
// coordinate order + precomputed successor lists
// =================================================================================================
// A walk advances along the contig coordinate, but vertex ids are k-mer major, i.e. random with respect
// to the coordinate: every step of a walk on the k-mer-major CSR is a chain of ~8 dependent random HBM
// accesses (TLB misses included, ~3 us each).  So the vertices are renumbered by contig coordinate
// (stable radix sort on DualPos.first: [ctg == 0 vertices] ++ [ctg != 0 ascending]) and the static part of
// the epsilon-join — searchSuccessors + checkPosition + isEdgeSimilar for EVERY vertex — is evaluated
// once, in parallel, into per-vertex successor records stored in that order.  A walk then streams
// through nearly consecutive memory: records, visit stamps and offsets of consecutive path vertices are
// neighbours.
// sort records: key = contig coordinate, payload = reference coordinate << 32 | vertex id (the position travels with the
// record, so that applying the order does not have to gather it back)
__global__ void k_order_keys(const uint64_t *__restrict__ vpos, uint64_t n, uint32_t *__restrict__ key, uint64_t *__restrict__ val) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = vpos[i];
        key[i] = (uint32_t)(p >> 32);
        val[i] = (p << 32) | i;
    }
}

// (slice / slice_shift: the random half of the work — newid[v], vcnt[v] — for the vertices v of one slice of the k-mer-major id
// range per launch, so that the slice of both arrays stays in the Infinity Cache; the streamed half with slice 0)
__global__ void k_order_apply(const uint32_t *__restrict__ sorted_ctg, const uint64_t *__restrict__ sorted_val, uint64_t n, uint64_t n_zero, TravGraph G,
                              uint32_t slice, uint32_t slice_shift) {
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v;
        if (slice == 0u) {
            const uint64_t x = sorted_val[u];
            v = (uint32_t)x;
            G.uold[u] = v;
            // (the first n_zero keys were overwritten by the second sort: their contig coordinate is 0)
            G.upos[u] = ((uint64_t)(u < n_zero ? 0u : sorted_ctg[u]) << 32) | (x >> 32);
        } else {
            v = G.uold[u];  // (what the first launch left: four bytes per vertex for the seven other slices instead of the eight of the sort record)
        }
        if (slice_shift >= 32u || (v >> slice_shift) == slice) {
            G.newid[v] = (uint32_t)u;
            G.ucnt[u] = G.vcnt[v];
        }
    }
}

int trav_compact(const uint32_t *tkey, const uint64_t *tval, const uint32_t *tseg, const uint16_t *tcnt, uint64_t T,
                 const uint32_t *ekey, const uint64_t *eval, const uint32_t *eseg, uint64_t E, uint32_t k, uint64_t n_nodes,
                 uint64_t n_pos, uint64_t n_edges, TravGraph G, void *tmp, size_t tmp_bytes, hipStream_t s, const TravView *view,
                 uint64_t *counts_out) {
    // tmp: flags u32[m] | scan out u64[m] | tile offsets + the view's ballots u64[n_tiles x 65 + ..] | tile counters u32[n_tiles] | scan
    // tmp | code table, m = max(tiles, bitmap words, nodes + 1): nothing here is as long as the tuple streams (until round 6 every
    // array was: 24 bytes per tuple slot, 37 GB for a 90 Mb block at 30x)
    // view != null: only the vertices inside its intervals (device arrays) are taken; counts_out[3] = nodes, vertices, edges
    // of the view (n_nodes / n_pos / n_edges are then upper bounds: what the arrays of G were sized for)
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    const uint64_t n_tiles_all = (T + VC_TILE - 1) / VC_TILE;
    const uint64_t m = std::max(n_tiles_all, std::max(n_words, n_nodes + 1)) + 1;
    const uint64_t m_ballots = n_tiles_all * (VC_TILE / 64 * 2 + 1) + 64;
    char *p = (char *)tmp;
    auto take = [&](size_t bytes) {
        char *q = p;
        p += (bytes + 255) & ~(size_t)255;
        return (void *)q;
    };
    uint32_t *flags = (uint32_t *)take(m * 4);
    uint64_t *sc1 = (uint64_t *)take(m * 8);
    uint64_t *sc2 = (uint64_t *)take(m_ballots * 8);
    uint32_t *keep = (uint32_t *)take((n_tiles_all + 1) * 4);
    uint64_t *totals = (uint64_t *)take(64);
    void *scan_tmp = take(scan_tmp_bytes(m));
    uint64_t *code_tab = k <= TRAV_CODE_TABLE_MAX_K ? (uint64_t *)take((size_t)8 << (2 * k)) : nullptr;
    if ((size_t)(p - (char *)tmp) > tmp_bytes) {
        set_error("trav_compact: scratch too small");
        return PAG_EINVAL;
    }
    PAG_HIP_TRY(hipMemsetAsync(G.bitmap, 0, n_words * 8, s));
    int rc;
    if (T) {
        const uint64_t n_tiles = (T + VC_TILE - 1) / VC_TILE;
        uint32_t *tile_first = flags, *tile_keep = keep;  // (n_tiles counters each)
        const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, 256u * 8u);
        const uint32_t *civ = view ? view->civ : nullptr, *riv = view ? view->riv : nullptr;
        const uint32_t n_civ = view ? view->n_civ : 0u, n_riv = view ? view->n_riv : 0u;
        uint64_t *ballots = sc2 + n_tiles + 16;  // (2 words per 64 slots, behind the tiles' offsets)
        k_view_mark<<<dim3(grid), dim3(VC_T), 0, s>>>(tkey, tval, tseg, T, civ, n_civ, riv, n_riv, view ? 0 : 1, ballots, tile_first, tile_keep, n_tiles);
        if ((rc = scan_u32_to_u64(tile_first, sc1, n_tiles, totals, scan_tmp, s))) return rc;
        if ((rc = scan_u32_to_u64(tile_keep, sc2, n_tiles, totals + 1, scan_tmp, s))) return rc;
        k_view_write<<<dim3(grid), dim3(VC_T), 0, s>>>(tkey, tval, tcnt, ballots, sc1, sc2, n_tiles, G);
    }
    if (view) {
        uint64_t h[2] = {0, 0};
        if (T) {
            PAG_HIP_TRY(hipMemcpyAsync(h, totals, 16, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
        if (h[0] > n_nodes || h[1] > n_pos) {
            set_error("trav_compact: the view holds more than the graph");
            return PAG_EFAULT;
        }
        n_nodes = h[0];
        n_pos = h[1];
        G.n_nodes = n_nodes;
        G.n_pos = n_pos;
    }
    uint32_t np32 = (uint32_t)n_pos;
    PAG_HIP_TRY(hipMemcpyAsync(G.npos_off + n_nodes, &np32, 4, hipMemcpyHostToDevice, s));
    // rank directory
    k_popc_words<<<dim3(grid_for(n_words)), dim3(256), 0, s>>>(G.bitmap, n_words, flags);
    if ((rc = scan_u32_to_u64(flags, sc1, n_words, nullptr, scan_tmp, s))) return rc;
    k_narrow<<<dim3(grid_for(n_words)), dim3(256), 0, s>>>(sc1, n_words, G.rank);
    // edges (of the k-mers that own a node: node_of_code finds no node for the others)
    PAG_HIP_TRY(hipMemsetAsync(flags, 0, (n_nodes + 1) * 4, s));
    if (E) k_edge_counts<<<dim3(grid_for(E)), dim3(256), 0, s>>>(ekey, eseg, E, G, flags);
    if ((rc = scan_u32_to_u64(flags, sc1, n_nodes + 1, view ? totals + 2 : nullptr, scan_tmp, s))) return rc;
    k_narrow<<<dim3(grid_for(n_nodes + 1)), dim3(256), 0, s>>>(sc1, n_nodes + 1, G.nedge_off);
    if (E && code_tab) {
        PAG_HIP_TRY(hipMemsetAsync(code_tab, 0xFF, (size_t)8 << (2 * k), s));
        TravGraph Gn = G;
        Gn.n_nodes = n_nodes;
        if (n_nodes) k_code_table<<<dim3(grid_for(n_nodes)), dim3(256), 0, s>>>(Gn, code_tab);
    }
    if (E) k_compact_edges<<<dim3(grid_for(E)), dim3(256), 0, s>>>(ekey, eval, eseg, E, G, code_tab);
    if (view) {
        uint64_t ne = 0;
        PAG_HIP_TRY(hipMemcpyAsync(&ne, totals + 2, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        if (ne > n_edges) {
            set_error("trav_compact: the view holds more edges than the graph");
            return PAG_EFAULT;
        }
        n_edges = ne;
    }
    if (counts_out) {
        counts_out[0] = n_nodes;
        counts_out[1] = n_pos;
        counts_out[2] = n_edges;
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

size_t trav_compact_tmp_bytes(uint64_t T, uint64_t E, uint32_t k, uint64_t n_nodes) {
    (void)E;
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    const uint64_t n_tiles = (T + VC_TILE - 1) / VC_TILE;
    const uint64_t m = std::max(n_tiles, std::max(n_words, n_nodes + 1)) + 1, m_ballots = n_tiles * (VC_TILE / 64 * 2 + 1) + 64;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al(m * 4) + al(m * 8) + al(m_ballots * 8) + al((n_tiles + 1) * 4) + al(64) + al(scan_tmp_bytes(m)) + (k <= TRAV_CODE_TABLE_MAX_K ? al((size_t)8 << (2 * k)) : 0) + 1024;
}

// reference bands of the zones (k_zone_bands): lo / hi [n_z] device arrays, preset here
int trav_zone_bands(const uint64_t *tval, uint64_t T, const uint32_t *zones_dev, uint32_t n_z, uint32_t *lo_dev, uint32_t *hi_dev, hipStream_t s) {
    if (!n_z) return PAG_OK;
    PAG_HIP_TRY(hipMemsetAsync(lo_dev, 0xFF, (size_t)n_z * 4, s));
    PAG_HIP_TRY(hipMemsetAsync(hi_dev, 0, (size_t)n_z * 4, s));
    if (T) k_zone_bands<<<dim3(grid_for(T)), dim3(256), 0, s>>>(tval, T, zones_dev, n_z, lo_dev, hi_dev);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

void trav_launch_ctg_nodes(const uint8_t *packed, const TravCtgNodesJob *jobs, uint32_t n_jobs, uint32_t max_len, uint32_t k, TravGraph G, uint32_t *out,
                           hipStream_t s) {
    const uint32_t n = max_len >= k ? max_len - k + 1 : 0;
    if (!n || !n_jobs) return;
    for (uint32_t at = 0; at < n_jobs; at += 65535u) {  // (gridDim.y)
        const uint32_t m = std::min(n_jobs - at, 65535u);
        k_ctg_nodes<<<dim3(std::min(grid_for(n), 256u), m), dim3(256), 0, s>>>(packed, jobs + at, k, G, out);
    }
}

// ---- a graph that holds a region of the block only (one rank of a sharded build) ----------------------------------
// largest step of any edge (bounds how far a successor's coordinate can lie from its source's)
__global__ void k_max_step(const uint32_t *__restrict__ estep, uint64_t n, uint32_t *__restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = (estep[i] & EDGE_STEP_MASK) > m ? (estep[i] & EDGE_STEP_MASK) : m;
    m = wave_max_u32(m);
    if (lane_id() == 0 && m) atomicMax(out, m);
}
// incomplete[u] for every vertex u (new ids; 0 .. n_zero: the coordinate-free ones, ordered by reference coordinate): its
// REFERENCE coordinate lies within `margin` of an OPEN end of the reference band it is in (iv: sorted disjoint [lo, hi)
// pairs; open[2 i], open[2 i + 1]: the graph goes on beyond that end, on another rank) — or in no band at all.  For a
// coordinate-free vertex the latter cannot happen (it was selected by its band); a vertex WITH a contig coordinate was
// selected by that coordinate whatever its reference coordinate is, and its coordinate-free successors (grade Skip,
// checkPosition with pos2.first == 0: PABruijnGraph.cpp:143-165) live around its reference coordinate — outside the bands
// they are on another rank.  A vertex without a reference coordinate has no successor that is found through one.
__global__ void k_mark_incomplete(TravGraph G, uint32_t n_zero, const uint32_t *__restrict__ iv, const uint8_t *__restrict__ open, uint32_t n_iv,
                                  uint32_t margin, uint32_t *__restrict__ bits) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (u < G.n_pos) bad = d_incomplete_by_position(iv, open, n_iv, margin, (uint32_t)G.upos[u], u >= n_zero);
    const uint64_t m = __ballot(bad);
    if ((threadIdx.x & 63u) == 0 && u < ((G.n_pos + 63ull) & ~63ull)) {
        bits[u >> 5] = (uint32_t)m;
        bits[(u >> 5) + 1] = (uint32_t)(m >> 32);
    }
}
int trav_mark_incomplete(TravGraph &G, uint32_t n_zero, const uint32_t *iv_host, const uint8_t *open_host, uint32_t n_iv, uint32_t dev, double err,
                         uint32_t *bits, void *tmp, hipStream_t s) {
    // tmp: u32 max step | intervals | open flags
    uint32_t *d_max = (uint32_t *)tmp;
    uint32_t *d_iv = d_max + 64;
    uint8_t *d_open = (uint8_t *)(d_iv + 2 * (size_t)n_iv + 2);
    PAG_HIP_TRY(hipMemsetAsync(d_max, 0, 4, s));
    if (G.n_edges) k_max_step<<<dim3(grid_for(G.n_edges)), dim3(256), 0, s>>>(G.estep, G.n_edges, d_max);
    if (n_iv) {
        PAG_HIP_TRY(hipMemcpyAsync(d_iv, iv_host, 2 * (size_t)n_iv * 4, hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(d_open, open_host, 2 * (size_t)n_iv, hipMemcpyHostToDevice, s));
    }
    uint32_t max_step = 0;
    PAG_HIP_TRY(hipMemcpyAsync(&max_step, d_max, 4, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    // a successor's coordinate lies within step + deviation, or step x (1 + error rate), of its source's (checkPosition)
    const uint64_t margin = (uint64_t)((double)max_step * (1.0 + err)) + dev + 2;
    // the bit per new id is only read when the successor kernel cannot repeat the test itself (more bands than it stages in
    // LDS); `incomplete` stays the flag that the graph holds a region
    const bool bits_read = n_iv > INC_LDS_MAX;
    if (G.n_pos && bits_read) k_mark_incomplete<<<dim3((unsigned)((G.n_pos + 255) / 256)), dim3(256), 0, s>>>(G, n_zero, d_iv, d_open, n_iv, (uint32_t)std::min<uint64_t>(margin, 0x7FFFFFFFu), bits);
    PAG_HIP_TRY(hipGetLastError());
    G.incomplete = bits;
    G.inc_iv = d_iv;  // (the scratch slot lives as long as the traversal graph: the successor kernels repeat the test, d_incomplete_by_position)
    G.inc_open = d_open;
    G.inc_n = n_iv;
    G.inc_margin = (uint32_t)std::min<uint64_t>(margin, 0x7FFFFFFFu);
    return PAG_OK;
}
size_t trav_mark_incomplete_tmp_bytes(uint32_t n_iv) { return 256 + (2 * (size_t)n_iv + 2) * 4 + 2 * (size_t)n_iv + 64; }

// coordinate order + successor records.  key/val/key2/val2: u32/u64 [n_pos] scratch pairs for the sort;
// cnt: u32 [n_pos + 1]; *n_succ_out receives the number of successor records (call twice: first with
// G.succ == nullptr to size it, then with the allocation)
// where the sorted keys stop being zero (keys ascending; *n0 preset to 0, stays 0 when key[0] != 0)
__global__ void k_zero_prefix(const uint32_t *__restrict__ key, uint64_t n, unsigned long long *__restrict__ n0) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (key[i] == 0u && (i + 1 == n || key[i + 1] != 0u)) *n0 = i + 1;
}
// sort keys of the vertices without a contig coordinate: their reference coordinate (the payload's upper half)
__global__ void k_order_refkeys(const uint64_t *__restrict__ val, uint64_t n, uint32_t *__restrict__ key) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        key[i] = (uint32_t)(val[i] >> 32);
}

// New ids: [vertices without a contig coordinate, by reference coordinate] ++ [the others, by contig coordinate]; equal
// keys keep the k-mer-major order (stable sorts).  The order inside the first group is not needed by the walks — it makes
// neighbours on the reference neighbours in memory, and it lets a splice of two walks bound the vertices of that kind a
// walk has examined by an id (k5_travel_host.hip, try_merge_leap).  *n_zero receives the size of the first group.
int trav_order(TravGraph G, uint32_t *key, uint64_t *val, uint32_t *key2, uint64_t *val2, void *sort_tmp, uint64_t *n_zero, int ctg_bits,
               int ref_bits, hipStream_t s) {
    const uint64_t n = G.n_pos;
    if (n_zero) *n_zero = 0;
    if (!n) return PAG_OK;
    k_order_keys<<<dim3(grid_for(n)), dim3(256), 0, s>>>(G.vpos, n, key, val);
    int in0 = 1, rc;
    if ((rc = sort_pairs(key, val, key2, val2, n, ctg_bits, sort_tmp, &in0, s, nullptr, nullptr))) return rc;
    uint32_t *ks = in0 ? key : key2, *ko = in0 ? key2 : key;
    uint64_t *vs = in0 ? val : val2, *vo = in0 ? val2 : val;
    unsigned long long *d_n0 = (unsigned long long *)sort_tmp;  // (the sort is done with its scratch)
    unsigned long long n0 = 0;
    PAG_HIP_TRY(hipMemsetAsync(d_n0, 0, 8, s));
    k_zero_prefix<<<dim3(grid_for(n)), dim3(256), 0, s>>>(ks, n, d_n0);
    PAG_HIP_TRY(hipMemcpyAsync(&n0, d_n0, 8, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    if (n0 > 1) {
        k_order_refkeys<<<dim3(grid_for(n0)), dim3(256), 0, s>>>(vs, n0, ks);
        int in0b = 1;
        if ((rc = sort_pairs(ks, vs, ko, vo, n0, ref_bits, sort_tmp, &in0b, s, nullptr, nullptr))) return rc;
        if (!in0b) PAG_HIP_TRY(hipMemcpyAsync(vs, vo, n0 * 8, hipMemcpyDeviceToDevice, s));
    }
    if (n_zero) *n_zero = n0;
    {
        // (eight slices once the two arrays — 6 bytes per vertex — outgrow the Infinity Cache: 16.2 -> 12.5 ms at configs[1],
        // 13.2 with four or sixteen, tests/order_probe.sh; tests: PAG_ORDER_SLICES=<2^n> runs the sliced form on small graphs)
        uint32_t lg = n >= (32ull << 20) ? 3u : 0u;
        if (const long long want = env_int("PAG_ORDER_SLICES", 0)) {
            lg = 0;
            while ((1u << (lg + 1)) <= (uint32_t)std::max<long long>(1, want)) ++lg;
        }
        uint32_t bits = 1;
        while (bits < 32 && (n >> bits) != 0) ++bits;  // v < n < 2^bits
        if (lg >= bits) lg = 0;
        const uint32_t shift = lg ? bits - lg : 32u;
        for (uint32_t sl = 0; sl < (1u << lg); ++sl)
            k_order_apply<<<dim3(grid_for(n)), dim3(256), 0, s>>>(ks, vs, n, n0 > 1 ? n0 : 0, G, sl, shift);
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

}  // namespace pagdev
