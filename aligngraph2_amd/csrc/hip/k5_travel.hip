// k5_travel.hip — K5: epsilon-join traversal on the device.
//
// What runs where (reference PAGraph/src/tools/graph/):
//   device  PABruijnGraph::searchSuccessors + checkPosition + isEdgeSimilar   PABruijnGraph.cpp:143-197, 385-400
//           PAlgorithm::classifySuccessors / walkStraight / graphTravel        PAlgorithm.tcc:35-298
//           PAlgorithm::searchPANode / searchPANode2 (seed scans)             PAlgorithm.tcc:300-365
//           PABruijnGraph::findAll (contig k-mers -> graph nodes)              PABruijnGraph.cpp:339-353
//   host    the outer loop of PAlgorithm::travelSequence (PAlgorithm.cpp:144-426): per round pick the
//           longest / leaping seed walk, appendSeq, repeat detection, re-seeding incl. the unstable
//           std::sort by edit distance (same libstdc++ => same tie order), filterSequence, "Pump it".
//
// One wavefront (= one 64-thread workgroup) owns one (contig, seed) graphTravel.  A walk is a chain of
// dependent steps, so the kernel is latency-bound by design; the lanes share the work inside a step:
// expanding children x positions, the f64 match predicates, the three visited-set probes, and ordered
// compaction by ballot.  Visited sets are open-addressing hash tables in HBM; the per-probe set of
// walkStraight uses generation tags so it never needs clearing.
//
// Before traversal the k-mer-sorted streams are compacted into a CSR with dense node / vertex ids and a
// 4^k-bit node bitmap + rank directory (code -> node id in two loads).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>
#include <string>
#include <unordered_set>
#include <vector>

#include "pag_device.hpp"
#include "pag_travel.hpp"

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
    const uint32_t *zz = lds ? s_z : zones;
    const uint32_t z_first = n_z ? zones[0] : 0u, z_last = n_z ? zones[2 * n_z - 1] : 0u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t p = tval[i];
        const uint32_t c = (uint32_t)(p >> 32), r = (uint32_t)p;
        uint32_t z;
        if (c < z_first || c >= z_last || r == 0u || !iv_contains(zz, n_z, c, &z)) continue;
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
    const uint32_t *cv = lds ? s_iv : civ, *rv = lds ? s_iv + 2u * n_civ : riv;
    auto inside = [&](uint64_t p) {
        const uint32_t c = (uint32_t)(p >> 32), r = (uint32_t)p;
        return whole || (c != 0u ? iv_contains(cv, n_civ, c) : iv_contains(rv, n_riv, r));
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

__device__ __forceinline__ uint32_t node_of_code(const TravGraph &G, uint32_t code) {
    uint64_t w = G.bitmap[code >> 6];
    uint32_t b = code & 63u;
    if (!((w >> b) & 1ull)) return PAG_NONE;
    const uint32_t in_code_order = G.rank[code >> 6] + (uint32_t)__popcll(w & ((1ull << b) - 1ull));
    return G.nperm ? G.nperm[in_code_order] : in_code_order;
}

// ---- nodes numbered by place (round 5).  The compaction leaves the nodes in code order, the order of the build's sorted tuples:
// a node's children — the k-mers that follow it in the reads — then lie anywhere in the node-major arrays, and so do the
// coordinate-ordered slots its vertices' results go to: every candidate list, every count, every record of the successor stage
// was a random 64-byte sector.  Numbered by WHERE the k-mer lies (the reference coordinate of the node's first vertex that has
// one; a node known on a contig only: its contig coordinate, behind the others), a node's children are its neighbours, and the
// coordinate order of its vertices runs alongside the node order.  Inside a node nothing moves (positions ascending, edges in
// tuple order): the reference's order of a vertex's successors does not depend on how nodes are numbered.
__global__ void k_node_place_keys(const uint32_t *__restrict__ npos_off, const uint64_t *__restrict__ vpos, uint64_t n_nodes, uint32_t shift,
                                  uint32_t flag, uint32_t *__restrict__ key, uint64_t *__restrict__ val) {
    for (uint64_t n = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; n < n_nodes; n += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = npos_off[n], b = npos_off[n + 1];
        uint32_t kk = flag | ((uint32_t)(vpos[a] >> 32) >> shift);
        for (uint32_t p = a; p < b; ++p) {
            const uint32_t r = (uint32_t)vpos[p];
            if (r != 0u) {
                kk = r >> shift;
                break;
            }
        }
        key[n] = kk;
        val[n] = n;
    }
}
// new id i <- node perm[i] of the code order: its number of vertices, and where the code order's number finds it again
__global__ void k_node_place_counts(const uint64_t *__restrict__ perm, const uint32_t *__restrict__ npos_off_old, uint64_t n_nodes,
                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ nperm) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = (uint32_t)perm[i];
        cnt[i] = npos_off_old[o + 1] - npos_off_old[o];
        nperm[o] = (uint32_t)i;
    }
}
__global__ void k_node_place_move(const uint64_t *__restrict__ perm, const uint64_t *__restrict__ off_new, const uint32_t *__restrict__ ncode_old,
                                  const uint32_t *__restrict__ npos_off_old, const uint64_t *__restrict__ vpos_old, const uint16_t *__restrict__ vcnt_old,
                                  uint64_t n_nodes, TravGraph G) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_nodes; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o = (uint32_t)perm[i];
        const uint32_t a = npos_off_old[o], b = npos_off_old[o + 1], d = (uint32_t)off_new[i];
        G.ncode[i] = ncode_old[o];
        G.npos_off[i] = d;
        for (uint32_t j = 0; j < b - a; ++j) {
            G.vpos[d + j] = vpos_old[a + j];
            G.vcnt[d + j] = vcnt_old[a + j];
            G.vnode[d + j] = (uint32_t)i;
        }
    }
}

// An edge of the traversal graph: eto = first position (k-mer-major vertex id) of the target node (PAG_NONE: the target has no
// node), estep = step (24 bits) | number of the target's positions << 24, EDGE_Q_MANY = "255 or more: count them".
constexpr uint32_t EDGE_STEP_MASK = 0xFFFFFFu, EDGE_Q_MANY = 255u;
constexpr uint32_t TRAV_CODE_TABLE_MAX_K = 14;  // (the direct code table of k_compact_edges: 8 B x 4^k)
struct __attribute__((packed, aligned(4))) U32x2 { uint32_t a[2]; };
__device__ __forceinline__ void edge_target(const TravGraph &G, uint32_t eto, uint32_t estep, uint32_t *step, uint32_t *p0, uint32_t *q) {
    *step = estep & EDGE_STEP_MASK;
    if (eto == PAG_NONE) {
        *p0 = 0u;
        *q = 0u;
        return;
    }
    *p0 = eto;
    uint32_t n = estep >> 24;
    if (n == EDGE_Q_MANY) {  // (a k-mer with hundreds of positions: its node's range)
        const uint32_t node = G.vnode[eto];
        n = G.npos_off[node + 1] - G.npos_off[node];
    }
    *q = n;
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
// match predicates (f64 exactly as the reference; compiled with -ffp-contract=off, no fast-math)
// =================================================================================================
__device__ __forceinline__ bool d_coord_sim(uint32_t a, uint32_t b, uint64_t dev) {
    return a != 0 && b != 0 && (uint64_t)((a > b ? a : b) - (a > b ? b : a)) <= dev;
}
// the ratio test of both predicates: fabs(1.0 - (double)D * 1.0 / (double)dist) <= err, D = u32 difference of the
// coordinates.  The f64 division (a ~25-instruction sequence) is only executed when D is within one percent of the
// accepted band: outside of [(1 - err - 0.01) dist, (1 + err + 0.01) dist] the quotient misses the band by 0.01, fifteen
// orders of magnitude more than the rounding of the two multiplications, so the answer is "no" without dividing.  Nine
// in ten candidate pairs (other copies of a repeated k-mer) leave here.
__device__ __forceinline__ bool d_ratio_ok(uint32_t D, int dist, double err) {
    const double dd = (double)D, ds = (double)dist;
    if (dd < (1.0 - err - 0.01) * ds || dd > (1.0 + err + 0.01) * ds) return false;
    return fabs(1.0 - (dd * 1.0 / ds)) <= err;
}
enum { G_OOPS = 0, G_SKIP = 1, G_GOOD = 2, G_EXCELLENT = 3, G_AMAZING = 4 };
// checkPosition (PABruijnGraph.cpp:143-165) incl. the un-guarded second ratio test (quirk Q6), with isEdgeSimilar
// (PABruijnGraph.cpp:385-400; *edge_sim: bit0 contig side, bit1 reference side) evaluated on the way: both use the
// same two ratio tests, each computed once here
__device__ __forceinline__ int d_check_position(uint32_t ac, uint32_t ar, uint32_t bc, uint32_t br, uint32_t dist, uint32_t dev,
                                                double err, uint32_t *edge_sim) {
    const bool q1 = d_ratio_ok(bc - ac, (int)dist, err), q2 = d_ratio_ok(br - ar, (int)dist, err);
    const uint32_t tc = ac != 0 ? ac + dist : 0, tr = ar != 0 ? ar + dist : 0;
    bool s1 = d_coord_sim(tc, bc, dev) || (ac != 0 && bc != 0 && q1);
    bool s2 = d_coord_sim(tr, br, dev) || (ar != 0 && br != 0 && q2);
    *edge_sim = (s1 ? 1u : 0u) | (s2 ? 2u : 0u);
    s1 = s1 || q1;
    s2 = s2 || q2;
    if (ac == 0 || bc == 0) return s2 ? (bc != 0 ? G_EXCELLENT : (ac != 0 ? G_SKIP : G_GOOD)) : G_OOPS;
    if (ar == 0 || br == 0) return s1 ? (br != 0 ? G_EXCELLENT : G_GOOD) : G_OOPS;
    return (s1 && s2) ? G_AMAZING : (s1 ? G_EXCELLENT : (s2 ? G_SKIP : G_OOPS));
}

// The ratio test as a table: for a given dist the coordinate differences D that pass d_ratio_ok are an interval (a
// correctly rounded division is monotonic in its dividend, so are 1 - x and fabs on either side of 1), [lo, lo + rng],
// found by trying d_ratio_ok itself on the few integers around (1 -+ err) dist.  Entry = lo | rng << 16; RATIO_TAB_NONE:
// no entry (dist 0 — nothing passes — or an interval that was not pinned down): the caller uses d_ratio_ok.  With the
// table a candidate pair costs integer compares only; the successor kernels, which run this predicate over five billion
// pairs per block at configs[1] and were bound by their vector instruction issue (SQ counters, profiles/r03_pmc_kernel_mix.json),
// keep it in LDS.
#define RATIO_TAB_N 1024u
#define RATIO_TAB_NONE 0xFFFFFFFFu
__device__ __forceinline__ uint32_t d_ratio_entry(uint32_t dist, double err) {
    if (dist == 0u) return RATIO_TAB_NONE;
    const double ds = (double)dist;
    const int64_t e0 = (int64_t)((1.0 - err) * ds), e1 = (int64_t)((1.0 + err) * ds);
    int64_t lo = -1, hi = -1;
    for (int64_t D = e0 > 3 ? e0 - 3 : 0; D <= e0 + 3; ++D)
        if (d_ratio_ok((uint32_t)D, (int)dist, err)) {
            lo = D;
            break;
        }
    for (int64_t D = e1 + 3; D >= (e1 > 3 ? e1 - 3 : 0); --D)
        if (d_ratio_ok((uint32_t)D, (int)dist, err)) {
            hi = D;
            break;
        }
    // the interval must have been bracketed on both sides (the first D tried at either end fails) and fit the entry
    const bool lo_ok = lo >= 0 && (lo == 0 || lo > (e0 > 3 ? e0 - 3 : 0)), hi_ok = hi >= 0 && hi < e1 + 3;
    if (!lo_ok || !hi_ok || hi < lo || lo > 0xFFFF || hi - lo > 0xFFFE) return RATIO_TAB_NONE;
    return (uint32_t)lo | ((uint32_t)(hi - lo) << 16);
}
__device__ __forceinline__ void d_ratio_table_fill(uint32_t *tab, double err) {  // (all threads of the block; __syncthreads after it)
    for (uint32_t d = threadIdx.x; d < RATIO_TAB_N; d += blockDim.x) tab[d] = d_ratio_entry(d, err);
}
// d_check_position with the two ratio tests given by a table entry (never RATIO_TAB_NONE)
__device__ __forceinline__ int d_check_position_tab(uint32_t ac, uint32_t ar, uint32_t bc, uint32_t br, uint32_t dist, uint32_t dev,
                                                    uint32_t entry, uint32_t *edge_sim) {
    const uint32_t lo = entry & 0xFFFFu, rng = entry >> 16;
    const bool q1 = (uint32_t)(bc - ac - lo) <= rng, q2 = (uint32_t)(br - ar - lo) <= rng;
    const uint32_t tc = ac != 0 ? ac + dist : 0, tr = ar != 0 ? ar + dist : 0;
    bool s1 = d_coord_sim(tc, bc, dev) || (ac != 0 && bc != 0 && q1);
    bool s2 = d_coord_sim(tr, br, dev) || (ar != 0 && br != 0 && q2);
    *edge_sim = (s1 ? 1u : 0u) | (s2 ? 2u : 0u);
    s1 = s1 || q1;
    s2 = s2 || q2;
    if (ac == 0 || bc == 0) return s2 ? (bc != 0 ? G_EXCELLENT : (ac != 0 ? G_SKIP : G_GOOD)) : G_OOPS;
    if (ar == 0 || br == 0) return s1 ? (br != 0 ? G_EXCELLENT : G_GOOD) : G_OOPS;
    return (s1 && s2) ? G_AMAZING : (s1 ? G_EXCELLENT : (s2 ? G_SKIP : G_OOPS));
}
// ... for any dist: through the table (LDS) where it has an entry
__device__ __forceinline__ int d_check_position_any(uint32_t ac, uint32_t ar, uint32_t bc, uint32_t br, uint32_t dist, uint32_t dev, double err,
                                                    uint32_t entry, uint32_t *edge_sim) {
    return entry != RATIO_TAB_NONE ? d_check_position_tab(ac, ar, bc, br, dist, dev, entry, edge_sim)
                                   : d_check_position(ac, ar, bc, br, dist, dev, err, edge_sim);
}

// =================================================================================================
// visited sets
// =================================================================================================
#define HS_EMPTY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t hs_hash(uint32_t key, uint32_t mask) { return (key * 2654435761u) & mask; }
// lookups use agent-scope (sc1) loads: inserts are L2 atomics, which a CU's L1 does not observe
__device__ __forceinline__ bool hs_has(const uint32_t *tab, uint32_t mask, uint32_t key) {
    if (!tab) return false;
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        uint32_t x = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x == key) return true;
        if (x == HS_EMPTY) return false;
    }
}
// concurrent insert (distinct or equal keys, any lanes)
__device__ __forceinline__ void hs_insert(uint32_t *tab, uint32_t mask, uint32_t key) {
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        uint32_t old = atomicCAS(&tab[s], HS_EMPTY, key);
        if (old == HS_EMPTY || old == key) return;
    }
}
// epoch-tagged travel set: entry = key | epoch << 32, empty = all ones; lookup returns the epoch (0 = absent)
#define HS64_EMPTY 0xFFFFFFFFFFFFFFFFull
__device__ __forceinline__ uint32_t hs64_epoch(const uint64_t *tab, uint32_t mask, uint32_t key) {
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        const uint64_t x = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x == HS64_EMPTY) return 0u;
        if ((uint32_t)x == key) return (uint32_t)(x >> 32);
    }
}
// concurrent insert of distinct keys (a key is inserted once per job: a vertex is appended once)
__device__ __forceinline__ void hs64_insert(uint64_t *tab, uint32_t mask, uint32_t key, uint32_t epoch) {
    const unsigned long long want = (unsigned long long)key | ((unsigned long long)epoch << 32);
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        const unsigned long long old = atomicCAS((unsigned long long *)&tab[s], (unsigned long long)HS64_EMPTY, want);
        if (old == (unsigned long long)HS64_EMPTY || (uint32_t)old == key) return;
    }
}
// generation-tagged set: entry = key | gen << 32; an entry of another generation counts as free
__device__ __forceinline__ bool gs_has(const uint64_t *tab, uint32_t mask, uint32_t key, uint32_t gen) {
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        uint64_t x = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32) != gen) return false;
        if ((uint32_t)x == key) return true;
    }
}
__device__ __forceinline__ void gs_insert_single(uint64_t *tab, uint32_t mask, uint32_t key, uint32_t gen) {  // one lane only
    for (uint32_t s = hs_hash(key, mask);; s = (s + 1) & mask) {
        uint64_t x = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(x >> 32) != gen) {
            __hip_atomic_store(&tab[s], (uint64_t)key | ((uint64_t)gen << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if ((uint32_t)x == key) return;
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
        const uint64_t x = sorted_val[u];
        const uint32_t v = (uint32_t)x;
        if (slice == 0u) {
            G.uold[u] = v;
            // (the first n_zero keys were overwritten by the second sort: their contig coordinate is 0)
            G.upos[u] = ((uint64_t)(u < n_zero ? 0u : sorted_ctg[u]) << 32) | (x >> 32);
        }
        if (slice_shift >= 32u || (v >> slice_shift) == slice) {
            G.newid[v] = (uint32_t)u;
            G.ucnt[u] = G.vcnt[v];
        }
    }
}

// may successors of a vertex lie outside the region this graph holds?  r = its reference coordinate, on_contig = it has a contig
// coordinate (new id >= n_zero).  The ONE statement of the test: k_mark_incomplete leaves it as a bit per new id, the successor
// kernels — threads in k-mer-major order, to which a bit at the vertex's new id is a random sector — evaluate it again from the
// vertex's own position with the bands staged in LDS.
__device__ __forceinline__ bool d_incomplete_by_position(const uint32_t *iv, const uint8_t *open, uint32_t n_iv, uint32_t margin, uint32_t r, bool on_contig) {
    uint32_t lo = 0, hi = n_iv;  // last interval with lo <= r
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (iv[2 * mid] <= r) lo = mid;
        else hi = mid;
    }
    bool bad = n_iv == 0 || r < iv[2 * lo] || r >= iv[2 * lo + 1];
    if (!bad) bad = (open[2 * lo] && r - iv[2 * lo] < margin) || (open[2 * lo + 1] && iv[2 * lo + 1] - r <= margin);
    if (on_contig && r == 0u) bad = false;
    return bad;
}

constexpr uint32_t INC_LDS_MAX = 256;  // bands a successor kernel stages in LDS (more: the bit per new id is gathered)
struct IncLds {
    uint32_t iv[2 * INC_LDS_MAX];
    uint8_t open[2 * INC_LDS_MAX];
};
__device__ __forceinline__ bool inc_lds_fill(IncLds &I, const TravGraph &G) {  // (all threads of the block; __syncthreads after it)
    const bool use = G.incomplete && G.inc_iv && G.inc_n <= INC_LDS_MAX;
    if (use)
        for (uint32_t i = threadIdx.x; i < 2u * G.inc_n; i += blockDim.x) {
            I.iv[i] = G.inc_iv[i];
            I.open[i] = G.inc_open[i];
        }
    return use;
}
// poison / marker of vertex v (new id u): from its own position when the bands are in LDS, else the bit at its new id
__device__ __forceinline__ bool vertex_incomplete(const TravGraph &G, const IncLds &I, bool lds, uint64_t v, uint32_t u) {
    if (!G.incomplete) return false;
    if (lds) {
        const uint64_t pv = G.vpos[v];
        return d_incomplete_by_position(I.iv, I.open, G.inc_n, G.inc_margin, (uint32_t)pv, (pv >> 32) != 0u);
    }
    return ((G.incomplete[u >> 5] >> (u & 31u)) & 1u) != 0u;
}

// The candidate pairs of one vertex v (searchSuccessors + checkPosition + isEdgeSimilar, PABruijnGraph.cpp:143-197,385-400):
// every position of every target node of its k-mer node, in CSR order.  WHAT = 0: count the accepted ones, and note in
// `mask` which of the first 64 candidates they are.  WHAT = 1: write the accepted ones to out[0 ..] — the first 64
// candidates through `mask` (nine in ten candidates are rejects: only the accepted ones are touched again), the rest
// evaluated again.  WHAT = 2: evaluate every candidate once and write the accepted ones as they come.  Returns their number.
// four / two consecutive entries with one load instruction (4- resp. 8-byte aligned addresses: the hardware takes them; the
// counting pass issues a load per candidate otherwise, and with some twenty k-mer nodes per wave every load instruction is
// twenty requests to the L1 — what the pass was bound by, not its arithmetic)
struct __attribute__((packed, aligned(4))) U32x4 { uint32_t a[4]; };
struct __attribute__((packed, aligned(8))) U64x4 { uint64_t a[4]; };
template <int WHAT>
#define SUCC_HEAVY 0xFFFFFFFFu
__device__ __forceinline__ uint32_t succ_vertex(const TravGraph &G, uint64_t v, uint32_t dev, double err, uint64_t &mask, SuccRec *__restrict__ out,
                                                const uint32_t *__restrict__ ratio_tab, uint32_t heavy_limit = 0xFFFFFFFFu, uint32_t src_u = 0u) {
    // (0 counts, 1 fills through the mask, 2 writes while it evaluates; 3: 1 for a vertex known to have at most 64 candidates —
    // everything is under the mask, the loop that evaluates candidates is not even compiled in: registers, k_succ<3>)
    // 4: 3 with the grading left to k_succ_link — the record gets its target's new id and the step, the source's new id in `toff`
    // (src_u), grade 0; neither the source's nor the target's position is read here: the linking pass runs in coordinate order,
    // where both are neighbours of what it reads anyway (one random sector per record less than 3)
    constexpr int MODE = WHAT == 3 || WHAT == 4 ? 1 : WHAT;
    constexpr bool MASK_ONLY = WHAT == 3 || WHAT == 4;
    constexpr bool DEFER = WHAT == 4;
    const uint64_t rootp = DEFER ? 0ull : G.vpos[v];
    const uint32_t rc = (uint32_t)(rootp >> 32), rr = (uint32_t)rootp;
    const uint32_t node = G.vnode[v];
    uint32_t n = 0, base = 0;
    const bool amask = true;  // (the mask is always there; a caller without one passes 0 and WHAT = 2)
    auto emit = [&](uint32_t tgt, uint32_t pc, uint32_t step, int grade, uint32_t esim) {
        SuccRec r;
        r.tgt = tgt;
        r.pc = pc;
        r.meta = (step & 0xFFFFFFu) | ((uint32_t)grade << 24) | ((esim & 1u) << 27);
        r.toff = 0;  // the target's own record range is linked in afterwards
        out[n] = r;
    };
    const uint32_t e_lo = G.nedge_off[node], e_hi = G.nedge_off[node + 1];
    for (uint32_t eb = e_lo; eb < e_hi; eb += 4u) {
      // (targets and position ranges of up to four edges requested together: two round trips for the four; k_succ<0>
      // 36 -> 34 ms, k_succ<1> 49 -> 46 ms)
      uint32_t to4[4], st4[4], p04[4], q4[4];
      const U32x4 toL = *(const U32x4 *)(G.eto + eb), stL = *(const U32x4 *)(G.estep + eb);  // (the arrays are padded by four entries)
#pragma unroll
      for (uint32_t t = 0; t < 4u; ++t) {
          const bool have = eb + t < e_hi;
          to4[t] = have ? toL.a[t] : PAG_NONE;
          edge_target(G, to4[t], have ? stL.a[t] : 0u, &st4[t], &p04[t], &q4[t]);
      }
#pragma unroll
      for (uint32_t t4 = 0; t4 < 4u; ++t4) {
        const uint32_t step = st4[t4], p0 = p04[t4], q = q4[t4];
        if (q == 0u) continue;
        // a vertex with more candidates than the limit is left to a whole wave (k_succ_heavy): the lanes of a wave here run as
        // long as the one with the longest lists
        if (q > heavy_limit || base + q > heavy_limit) return SUCC_HEAVY;
        const uint32_t entry = step < RATIO_TAB_N ? ratio_tab[step] : RATIO_TAB_NONE;  // (the ratio tests of this edge, see d_ratio_entry)
        uint32_t j0 = 0;
        if (MODE == 1 && amask) {  // the candidates the mask covers: accepted ones only
            const uint32_t lim = base < 64u ? (q < 64u - base ? q : 64u - base) : 0u;
            uint64_t sub = lim ? (mask >> base) & (lim == 64u ? ~0ull : ((1ull << lim) - 1ull)) : 0ull;
            while (sub) {
                const uint32_t p = p0 + (uint32_t)(__ffsll((long long)sub) - 1);
                sub &= sub - 1ull;
                if (DEFER) {
                    SuccRec r;
                    r.tgt = G.newid[p];
                    r.pc = 0u;
                    r.meta = step & 0xFFFFFFu;  // (grade 0 = not graded yet)
                    r.toff = src_u;
                    out[n++] = r;
                    continue;
                }
                const uint64_t pp = G.vpos[p];
                const uint32_t tgt = G.newid[p];  // (asked for together with the position: one round trip per record, not two)
                uint32_t esim;
                const int grade = d_check_position_any(rc, rr, (uint32_t)(pp >> 32), (uint32_t)pp, step, dev, err, entry, &esim);
                emit(tgt, (uint32_t)(pp >> 32), step, grade, esim);
                ++n;
            }
            j0 = lim;
        }
        // four candidates per turn, their positions requested together: one at a time, every candidate cost a full memory
        // round trip (load -> f64 tests -> next load) and the kernel ran at a third of the miss rate the memory system
        // sustains (k_succ<0>: 45 -> 36 ms).  Also having the first positions of four edges in flight was slower (48 ms:
        // a node has 2.65 edges on average, the rest is wasted loads and registers), eight candidates per turn no better
        // (35.6 ms), four lanes per vertex slower (51 ms), and keeping the first four accepted candidates of every vertex
        // in a side array for the filling pass cost the counting pass more (+14 ms) than it saved the other (-5 ms).
        for (uint32_t jb = j0; !MASK_ONLY && jb < q; jb += 4u) {
            const U64x4 pqL = *(const U64x4 *)(G.vpos + p0 + jb);  // (padded by four entries)
            const uint64_t *pq = pqL.a;
#pragma unroll
            for (uint32_t t = 0; t < 4u; ++t) {
                const uint32_t j = jb + t;
                if (j >= q) break;
                const uint32_t p = p0 + j;
                const uint32_t pc = (uint32_t)(pq[t] >> 32), pr = (uint32_t)pq[t];
                uint32_t esim;
                int grade = d_check_position_any(rc, rr, pc, pr, step, dev, err, entry, &esim);
                if (grade == G_OOPS) continue;
                if (MODE == 0 && base + j < 64u) mask |= 1ull << (base + j);
                if (MODE != 0) emit(G.newid[p], pc, step, grade, esim);
                ++n;
            }
        }
        base += q;
      }
    }
    return n;
}

// ... by a whole wave: 64 candidates of a target node's position list per turn (one coalesced load), the accepted ones counted
// / placed through the ballot.  WHAT = 0: count + the mask of the first 64 candidates (as succ_vertex<0> leaves it);
// WHAT = 1: every candidate evaluated again, the accepted ones written in CSR order.  v is the same in all lanes.
template <int WHAT>
__device__ __forceinline__ uint32_t succ_vertex_wave(const TravGraph &G, uint64_t v, uint32_t dev, double err, uint64_t &mask, SuccRec *__restrict__ out,
                                                     const uint32_t *__restrict__ ratio_tab) {
    const uint32_t lane = lane_id();
    const uint64_t rootp = G.vpos[v];
    const uint32_t rc = (uint32_t)(rootp >> 32), rr = (uint32_t)rootp;
    const uint32_t node = G.vnode[v];
    uint32_t n = 0, base = 0;
    const uint32_t e_lo = G.nedge_off[node], e_hi = G.nedge_off[node + 1];
    for (uint32_t e = e_lo; e < e_hi; ++e) {
        const uint32_t to = G.eto[e];
        if (to == PAG_NONE) continue;
        uint32_t step, p0, q;
        edge_target(G, to, G.estep[e], &step, &p0, &q);
        const uint32_t entry = step < RATIO_TAB_N ? ratio_tab[step] : RATIO_TAB_NONE;
        for (uint32_t jb = 0; jb < q; jb += 64u) {
            const uint32_t j = jb + lane;
            const bool valid = j < q;
            const uint64_t pp = valid ? G.vpos[p0 + j] : 0ull;
            const uint32_t pc = (uint32_t)(pp >> 32), pr = (uint32_t)pp;
            uint32_t esim = 0;
            const int grade = valid ? d_check_position_any(rc, rr, pc, pr, step, dev, err, entry, &esim) : (int)G_OOPS;
            const uint64_t b = __ballot(grade != G_OOPS);
            if (WHAT == 0 && base + jb < 64u) mask |= b << (base + jb);
            if (WHAT == 1 && grade != G_OOPS) {
                SuccRec r;
                r.tgt = G.newid[p0 + j];
                r.pc = pc;
                r.meta = (step & 0xFFFFFFu) | ((uint32_t)grade << 24) | ((esim & 1u) << 27);
                r.toff = 0;
                out[n + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = r;
            }
            n += (uint32_t)__popcll(b);
        }
        base += q;
    }
    return n;
}

// Threads run over the vertices in k-mer-major order (the order of the CSR): neighbouring threads belong to the same k-mer
// node, so the node's edge list and the position lists of its target nodes are shared through the caches; only the
// per-vertex results go to coordinate-ordered (random) places.
// MODE 0: count the records of every vertex (cnt[u]) and leave the acceptance mask of its first 64 candidates (amask[v]).
// MODE 1: write them to their final place succ[succ_off[u] ..] (after MODE 0 + scan: the two-pass path).
// MODE 2: write them to a staging array at stage_off[v] — offsets from the cheap upper bound of k_succ_bound — and count;
//         k_succ_place moves them to coordinate order afterwards.
// (Measured and not kept, round 3: count, then APPEND to a staging array in the same kernel — a wave adds up its counts, takes
// that much of the array with one atomic add, every thread goes over its accepted candidates once more — and gather the
// records into coordinate order afterwards (k_succ_place): 82 + 31 ms at configs[1] against 37 + 45 + 6 ms for count, fill
// and link: the second walk over the candidates is what the fill pass costs, not its scattered writes, and it is no
// cheaper inside the counting thread.)
// heavy_list / heavy_n (MODE 0 and 1, may be null): the vertices with more than heavy_limit candidates are left out here — MODE 0
// appends them to the list — and done by k_succ_heavy, a wave each.
template <int MODE>
__global__ void __launch_bounds__(256, 8) k_succ(TravGraph G, uint32_t dev, double err, uint32_t *__restrict__ cnt, const uint64_t *__restrict__ stage_off,
                       SuccRec *__restrict__ stage, uint64_t *__restrict__ amask, uint32_t *__restrict__ heavy_list,
                       unsigned long long *__restrict__ heavy_n, uint32_t heavy_limit) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    __shared__ IncLds inc_bands;
    d_ratio_table_fill(ratio_tab, err);
    const bool inc_lds = inc_lds_fill(inc_bands, G);
    __syncthreads();
    constexpr bool FILL = MODE == 1 || MODE == 3 || MODE == 4;  // (3, 4: with the heavy list and a limit of at most 64 — see succ_vertex)
    if (!heavy_list || MODE == 2) heavy_limit = 0xFFFFFFFFu;
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < G.n_pos; v += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n = 0;
        uint64_t mask = 0ull;
        const uint32_t u = G.newid[v];
        // successors of it may lie outside the region this graph holds (k_mark_incomplete).  A coordinate-free vertex gets one
        // poison record IN PLACE of its successors; a vertex on a contig keeps its successors — those that follow the contig
        // are all here — and gets one marker record behind them that only counts where a walk could take a Skip grade
        const bool inc = vertex_incomplete(G, inc_bands, inc_lds, v, u);
        const bool poison = inc && u < G.n_zero, marker = inc && u >= G.n_zero;
        if (poison) n = 1u;
        else if (MODE == 0) {
            n = succ_vertex<0>(G, v, dev, err, mask, nullptr, ratio_tab, heavy_limit);
            const bool heavy = n == SUCC_HEAVY;
            const uint64_t hb = __ballot(heavy);
            if (hb) {  // (one atomic per wave)
                const int first = __ffsll((long long)hb) - 1;
                unsigned long long at = 0;
                if ((int)lane_id() == first) at = atomicAdd(heavy_n, (unsigned long long)__popcll(hb));
                at = __shfl(at, first);
                if (heavy) heavy_list[at + (uint32_t)__popcll(hb & ((1ull << lane_id()) - 1ull))] = (uint32_t)v;
            }
            if (heavy) continue;
            n += marker ? 1u : 0u;
        }
        SuccRec *out = nullptr;
        if (FILL) out = G.succ + G.succ_off[u];
        if (MODE == 2) out = stage + stage_off[v];
        if (MODE != 0 && out) {
            SuccRec r;
            r.tgt = u;
            r.pc = 0u;  // (no coordinate: neither a leap nor subject to the coordinate windows; the grade alone rejects it)
            r.meta = 1u | ((poison ? GRADE_POISON : GRADE_POISON_IF_LEAP) << 24);
            r.toff = 0;
            if (poison) {
                out[0] = r;
            } else if (FILL) {
                uint32_t m;
                if (amask) {
                    mask = amask[v];
                    m = succ_vertex<MODE == 3 || MODE == 4 ? MODE : 1>(G, v, dev, err, mask, out, ratio_tab, heavy_limit, u);
                    if (m == SUCC_HEAVY) continue;
                } else {
                    m = succ_vertex<2>(G, v, dev, err, mask, out, ratio_tab);
                }
                if (marker) out[m] = r;
            } else if (MODE == 2) {
                n = succ_vertex<2>(G, v, dev, err, mask, out, ratio_tab);
                if (marker) out[n++] = r;
            }
        }
        if (!FILL) cnt[u] = n;
        if (MODE == 0 && amask) amask[v] = mask;
    }
}

// the vertices k_succ left out (more candidates than its limit), a wave each
template <int MODE>
__global__ void __launch_bounds__(256) k_succ_heavy(TravGraph G, uint32_t dev, double err, uint32_t *__restrict__ cnt, uint64_t *__restrict__ amask,
                                                    const uint32_t *__restrict__ heavy_list, const unsigned long long *__restrict__ heavy_n) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    __shared__ IncLds inc_bands;
    d_ratio_table_fill(ratio_tab, err);
    const bool inc_lds = inc_lds_fill(inc_bands, G);
    __syncthreads();
    const uint64_t total = *heavy_n, n_waves = (uint64_t)gridDim.x * (blockDim.x / 64u);
    for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; i < total; i += n_waves) {
        const uint64_t v = heavy_list[i];
        const uint32_t u = G.newid[v];
        const bool marker = vertex_incomplete(G, inc_bands, inc_lds, v, u);  // (a poisoned vertex never comes here)
        uint64_t mask = 0ull;
        if (MODE == 0) {
            const uint32_t n = succ_vertex_wave<0>(G, v, dev, err, mask, nullptr, ratio_tab) + (marker ? 1u : 0u);
            if (lane_id() == 0) {
                cnt[u] = n;
                amask[v] = mask;
            }
        } else {
            SuccRec *out = G.succ + G.succ_off[u];
            const uint32_t m = succ_vertex_wave<1>(G, v, dev, err, mask, out, ratio_tab);
            if (marker && lane_id() == 0) {
                SuccRec r;
                r.tgt = u;
                r.pc = 0u;
                r.meta = 1u | (GRADE_POISON_IF_LEAP << 24);
                r.toff = 0;
                out[m] = r;
            }
        }
    }
}

// ONE evaluation of the candidate pairs, records staged DENSELY (round 5, PAG_SUCC_MODE=fused): a vertex's candidates are walked
// and counted (with the acceptance mask), the waves take room for their vertices' records from a cursor — one atomic per wave —
// and every thread writes its accepted records there straight away, through the mask, while its node's edge and position lists
// are still in the caches; k_succ_place moves them to coordinate order and links them.  Against the two passes: the edge /
// position lists are read once, the records leave as full lines (k-mer-major neighbours write neighbouring records) instead of
// 16-byte pieces at coordinate-ordered places, and nobody reads succ_off[u] at a random place.  stage_off[v] = where v's records
// begin; a vertex whose records would not fit the staging array (cap) writes none — the caller sees the cursor beyond the cap
// and falls back to the two passes.  GENERAL = false: every vertex has at most 64 candidates (the others are on the heavy list).
template <bool GENERAL>
__global__ void __launch_bounds__(256, 8) k_succ_fused(TravGraph G, uint32_t dev, double err, uint32_t *__restrict__ cnt, uint64_t *__restrict__ stage_off,
                                                       SuccRec *__restrict__ stage, uint64_t cap, unsigned long long *__restrict__ cursor,
                                                       uint32_t *__restrict__ heavy_list, unsigned long long *__restrict__ heavy_n, uint32_t heavy_limit) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    __shared__ IncLds inc_bands;
    d_ratio_table_fill(ratio_tab, err);
    const bool inc_lds = inc_lds_fill(inc_bands, G);
    __syncthreads();
    if (!heavy_list) heavy_limit = 0xFFFFFFFFu;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t v0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); v0 < G.n_pos; v0 += stride) {  // (wave-uniform trip count)
        const uint64_t v = v0 + lane_id();
        const bool active = v < G.n_pos;
        uint32_t n = 0, u = 0;
        uint64_t mask = 0ull;
        bool poison = false, marker = false, heavy = false;
        if (active) {
            u = G.newid[v];
            const bool inc = vertex_incomplete(G, inc_bands, inc_lds, v, u);
            poison = inc && u < G.n_zero;
            marker = inc && u >= G.n_zero;
            if (poison) n = 1u;
            else {
                n = succ_vertex<0>(G, v, dev, err, mask, nullptr, ratio_tab, heavy_limit);
                heavy = n == SUCC_HEAVY;
                n = heavy ? 0u : n + (marker ? 1u : 0u);
            }
        }
        const uint64_t hb = __ballot(heavy);
        if (hb) {  // (one atomic per wave)
            const int first = __ffsll((long long)hb) - 1;
            unsigned long long at = 0;
            if ((int)lane_id() == first) at = atomicAdd(heavy_n, (unsigned long long)__popcll(hb));
            at = __shfl(at, first);
            if (heavy) heavy_list[at + (uint32_t)__popcll(hb & ((1ull << lane_id()) - 1ull))] = (uint32_t)v;
        }
        uint32_t tot;
        const uint32_t ex = wave_excl_sum(n, &tot);
        unsigned long long base = 0;
        if (tot) {
            if (lane_id() == 0) base = atomicAdd(cursor, (unsigned long long)tot);
            base = __shfl(base, 0);
        }
        if (!active || heavy) continue;
        const uint64_t off = base + ex;
        stage_off[v] = off;
        cnt[u] = n;
        if (n == 0u || off + n > cap) continue;
        SuccRec *out = stage + off;
        SuccRec r;
        r.tgt = u;
        r.pc = 0u;
        r.meta = 1u | ((poison ? GRADE_POISON : GRADE_POISON_IF_LEAP) << 24);
        r.toff = 0;
        if (poison) {
            out[0] = r;
        } else {
            const uint32_t m = succ_vertex<GENERAL ? 1 : 3>(G, v, dev, err, mask, out, ratio_tab, heavy_limit);
            if (marker) out[m] = r;
        }
    }
}
// ... the vertices on the heavy list, a wave each: counted, room taken, written
__global__ void __launch_bounds__(256) k_succ_heavy_fused(TravGraph G, uint32_t dev, double err, uint32_t *__restrict__ cnt, uint64_t *__restrict__ stage_off,
                                                          SuccRec *__restrict__ stage, uint64_t cap, unsigned long long *__restrict__ cursor,
                                                          const uint32_t *__restrict__ heavy_list, const unsigned long long *__restrict__ heavy_n) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    __shared__ IncLds inc_bands;
    d_ratio_table_fill(ratio_tab, err);
    const bool inc_lds = inc_lds_fill(inc_bands, G);
    __syncthreads();
    const uint64_t total = *heavy_n, n_waves = (uint64_t)gridDim.x * (blockDim.x / 64u);
    for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; i < total; i += n_waves) {
        const uint64_t v = heavy_list[i];
        const uint32_t u = G.newid[v];
        const bool marker = vertex_incomplete(G, inc_bands, inc_lds, v, u);  // (a poisoned vertex never comes here)
        uint64_t mask = 0ull;
        const uint32_t n = succ_vertex_wave<0>(G, v, dev, err, mask, nullptr, ratio_tab) + (marker ? 1u : 0u);
        unsigned long long off = 0;
        if (lane_id() == 0) off = atomicAdd(cursor, (unsigned long long)n);
        off = __shfl(off, 0);
        if (lane_id() == 0) {
            cnt[u] = n;
            stage_off[v] = off;
        }
        if (n == 0u || off + n > cap) continue;
        SuccRec *out = stage + off;
        const uint32_t m = succ_vertex_wave<1>(G, v, dev, err, mask, out, ratio_tab);
        if (marker && lane_id() == 0) {
            SuccRec r;
            r.tgt = u;
            r.pc = 0u;
            r.meta = 1u | (GRADE_POISON_IF_LEAP << 24);
            r.toff = 0;
            out[m] = r;
        }
    }
}

// upper bound of a vertex's records: the positions of all target nodes of its k-mer node (every candidate pair)
__global__ void k_succ_bound(TravGraph G, uint32_t *__restrict__ ub) {
    for (uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; v < G.n_pos; v += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t node = G.vnode[v];
        uint32_t n = 0;
        for (uint32_t e = G.nedge_off[node]; e < G.nedge_off[node + 1]; ++e) {
            uint32_t step, p0, q;
            edge_target(G, G.eto[e], G.estep[e], &step, &p0, &q);
            n += q;
        }
        ub[v] = n + (G.incomplete ? 1u : 0u);  // (room for a poison / marker record, see k_succ)
    }
}

// staged records -> coordinate order, target ranges linked in on the way (see k_succ_link)
__global__ void k_succ_place(TravGraph G, const uint64_t *__restrict__ stage_off, const SuccRec *__restrict__ stage) {
    for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < G.n_pos; u += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t o0 = G.succ_off[u], o1 = G.succ_off[u + 1];
        if (o0 == o1) continue;
        const SuccRec *src = stage + stage_off[G.uold[u]];
        for (uint32_t j = 0; j < o1 - o0; ++j) {
            SuccRec r = src[j];
            const uint32_t t0 = G.succ_off[r.tgt], t1 = G.succ_off[r.tgt + 1];
            const uint32_t tc = t1 - t0 < 15u ? t1 - t0 : 15u;
            r.meta |= tc << 28;
            r.toff = t0;
            G.succ[o0 + j] = r;
        }
    }
}

// Every record learns its target's record range (offset + count clamped to 15 = "15 or more: look the range up"), so
// that a walk step never waits for succ_off.  A pass of its own over the records IN COORDINATE ORDER: the targets of
// neighbouring records are neighbours on the strand, so the succ_off reads hit the caches — inside k_succ (k-mer-major
// threads) the same reads were one random HBM access per record.
// GRADE: a record that k_succ<4> left ungraded (grade 0, the source's new id in `toff`) gets its target's contig coordinate, its
// grade and its edge-similarity bit here, from the positions of both ends IN COORDINATE ORDER (upos): the source's is the
// neighbour of the previous record's, the target's lies a step ahead of it on the strand.
template <bool GRADE>
__global__ void k_succ_link(TravGraph G, uint64_t n_rec, uint32_t dev, double err) {
    __shared__ uint32_t ratio_tab[GRADE ? RATIO_TAB_N : 1u];
    if (GRADE) {
        d_ratio_table_fill(ratio_tab, err);
        __syncthreads();
    }
    // four records per thread and trip, their loads issued together (one dependent gather each: latency-bound otherwise)
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += 4 * T) {
        SuccRec r[4];
        uint32_t t0[4], t1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t x = i + (uint64_t)q * T;
            r[q] = G.succ[x < n_rec ? x : i];  // one 16-byte load
        }
        uint64_t ps[4], pt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            t0[q] = G.succ_off[r[q].tgt];
            t1[q] = G.succ_off[r[q].tgt + 1];
            if (GRADE) {
                const bool todo = ((r[q].meta >> 24) & 7u) == 0u;
                pt[q] = todo ? G.upos[r[q].tgt] : 0ull;
                ps[q] = todo ? G.upos[r[q].toff] : 0ull;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t x = i + (uint64_t)q * T;
            if (x >= n_rec) continue;
            if (GRADE && ((r[q].meta >> 24) & 7u) == 0u) {
                const uint32_t step = r[q].meta & 0xFFFFFFu;
                const uint32_t entry = step < RATIO_TAB_N ? ratio_tab[step] : RATIO_TAB_NONE;
                uint32_t esim;
                const int grade = d_check_position_any((uint32_t)(ps[q] >> 32), (uint32_t)ps[q], (uint32_t)(pt[q] >> 32), (uint32_t)pt[q], step, dev, err, entry, &esim);
                r[q].pc = (uint32_t)(pt[q] >> 32);
                r[q].meta = step | ((uint32_t)grade << 24) | ((esim & 1u) << 27);
            }
            const uint32_t tc = t1[q] - t0[q] < 15u ? t1[q] - t0[q] : 15u;
            r[q].meta |= tc << 28;
            r[q].toff = t0[q];
            G.succ[x] = r[q];  // one 16-byte store
        }
    }
}

// [first, last) new-id range of the vertices whose contig coordinate lies in [lo, hi)
__global__ void k_ranges(TravGraph G, TravContig *ctgs, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto lower = [&](uint32_t x) {
        uint64_t lo = 0, hi = G.n_pos;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if ((uint32_t)(G.upos[mid] >> 32) < x) lo = mid + 1;
            else hi = mid;
        }
        return (uint32_t)lo;
    };
    ctgs[i].in_lo = lower(ctgs[i].ctg_left);
    ctgs[i].in_hi = lower(ctgs[i].ctg_right);
    ctgs[i].g_lo = ctgs[i].in_lo;
    ctgs[i].g_hi = ctgs[i].in_hi;
}

// first new id whose contig coordinate is >= coords[i] (the vertices with a coordinate are ordered by it)
__global__ void k_id_bounds(TravGraph G, const uint32_t *__restrict__ coords, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = coords[i];
    uint64_t lo = 0, hi = G.n_pos;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(G.upos[mid] >> 32) < x) lo = mid + 1;
        else hi = mid;
    }
    out[i] = (uint32_t)lo;
}

// =================================================================================================
// the walker
// =================================================================================================
// What a walker wave keeps in LDS decides how many of them a compute unit holds (160 KB / sizeof(WalkLds)): until round 5 the
// worst-case caps below were 256 / 256 / 64 and made up half of the 52 KB (three waves per CU).  Now the LDS copies hold the
// COMMON case and the rare long list lives in global memory with no cap but the job's own buffers, which the host doubles on
// overflow (k5_travel_host.hip): a classification of more than 64 records writes its chosen class behind the end of the
// job's sequence (free room until the next append, see classify), a probe that stops with more than PB_CAP accepted records
// leaves none behind and graphTravel classifies that vertex itself.
constexpr int LIST_CAP = 64;   // successors of one vertex kept in LDS (the classification of at most 64 records)
constexpr int BR_CAP = 64;     // alternatives of a graphTravel round kept in LDS (more: read from the global list)
constexpr int PB_CAP = 16;     // accepted records of the classification a probe stopped at, per slot (more: not kept)
constexpr int PROBE_GROUPS = TRAV_PROBE_GROUPS;  // probe slots = lane groups of a wave
constexpr uint32_t GL = 64u / PROBE_GROUPS;       // lanes per slot: successor records of one vertex evaluated side by side
constexpr uint32_t GL_SHIFT = GL == 8u ? 3u : 4u;
constexpr uint32_t GL_MASK = (1u << GL) - 1u;
static_assert(GL == 8u || GL == 16u, "slot geometry");
static_assert(GL <= (uint32_t)PB_CAP, "a slot's own lanes always fit their accepted records (slots_step)");
#define STAMP_TRAVEL 0xFFFFFFFFu

// A walk moves through the coordinate-ordered arrays almost monotonically, a few ids per step, and every
// step is a chain of two dependent reads (successor records, then the visited marks of their targets).
// Served from HBM/L2 that chain costs ~2.3 us per step; the walker therefore keeps a WINDOW of the arrays
// in LDS — the records, the four probe-stamp arrays and the two visited bitmaps of WIN_IDS consecutive
// vertices of the strand — refilled with wave-wide coalesced loads every couple of hundred steps.  The
// window is a pure read cache: every mark is written through to the global arrays, and any access that
// falls outside the window uses them directly.
// Three window sizes (make WALK_WINDOW=tiny|small|large): they leave room for three, two or one walker wave per compute
// unit.  A lone wave is fastest per step with the large window (the small one costs 2.5 %: three times as many refills),
// and more waves per compute unit slow each other down (two: x 0.72 per wave) — which decided for ONE wave while a
// contig's exact tail was the critical path (round 2, first cut: 348 ms, two waves 553 ms).  With the leaping zone cut
// into pieces as well the walks are bound by the throughput of the whole grid, and more, slower waves win: 246 ms (large,
// 256 waves), 180 ms (small, 512), 165 ms (tiny, 768) at BASELINE configs[1].  Default: tiny.
#if defined(PAG_WALK_TINY_WINDOW)
constexpr uint32_t WIN_IDS = 256;
constexpr uint32_t WIN_REC = 832;
constexpr uint32_t FILT_WORDS = 128;
constexpr uint32_t WIN_BACK = 32;
constexpr uint32_t WIN_AHEAD = 64;
#elif defined(PAG_WALK_SMALL_WINDOW)
constexpr uint32_t WIN_IDS = 512;
constexpr uint32_t WIN_REC = 1792;
constexpr uint32_t FILT_WORDS = 256;  // 16 Ki-bit membership filters in front of the outside-range hash sets
constexpr uint32_t WIN_BACK = 64;    // ids kept behind the anchor at a refill
constexpr uint32_t WIN_AHEAD = 96;
#else
constexpr uint32_t WIN_IDS = 1024;
constexpr uint32_t WIN_REC = 4096;
constexpr uint32_t FILT_WORDS = 1024;  // 64 Ki-bit membership filters in front of the outside-range hash sets
constexpr uint32_t WIN_BACK = 128;   // ids kept behind the anchor at a refill
constexpr uint32_t WIN_AHEAD = 160;
#endif
static_assert(WIN_IDS / 32u <= 64u, "one lane per word of the global-visited window");
constexpr uint64_t SPEC_MARGIN = 50000;  // bases: no zombie within this distance of the can-leap threshold  // refill when the anchor gets this close to the upper end

struct WalkLds {
    uint32_t lst_v[1][LIST_CAP], lst_s[1][LIST_CAP];
    uint32_t br_v[BR_CAP], br_s[BR_CAP];
    // per list entry of class 0 / per alternative: contig coordinate, first successor record, record count
    // (15 = look it up) of the vertex, taken from the record that led to it; valid when *_meta is set
    uint32_t lst_pc[LIST_CAP], lst_off[LIST_CAP], lst_cnt[LIST_CAP];
    uint32_t br_pc[BR_CAP], br_off[BR_CAP], br_cnt[BR_CAP];
    SuccRec wrec[WIN_REC];
    uint32_t wst[PROBE_GROUPS][WIN_IDS];
    uint32_t wts[WIN_IDS];       // travel-visited epoch of the window's vertices
    uint32_t wgb[WIN_IDS / 32];  // global-visited bits
    uint32_t wab[WIN_IDS + 256];  // abundance of the window's vertices, from id w_d0 - w_ab (choice among branching alternatives)
    // the classification a probe stopped at (END / BRANCH): every accepted record with its class, in record order.
    // After the chosen path is appended, graphTravel's own classification of its last vertex is this list minus
    // the records whose coordinate now falls into the (hull of the) travel window — see k_walk's main loop.
    uint32_t pb_cnt[PROBE_GROUPS];
    uint32_t pb_v[PROBE_GROUPS][PB_CAP], pb_meta[PROBE_GROUPS][PB_CAP], pb_pc[PROBE_GROUPS][PB_CAP], pb_off[PROBE_GROUPS][PB_CAP], pb_cls[PROBE_GROUPS][PB_CAP];
    // blocked Bloom filters (two bits inside one 64-bit word) over the vertices OUTSIDE the strand's id range
    // that are in the travel-visited set (ft, maintained on insert) and in the contig's global visited set
    // (fg, built once per job): a clear bit proves absence, so the random probe into the global hash table
    // is only made when both bits are set
    uint64_t ft[FILT_WORDS], fg[FILT_WORDS];
};
// the first outside-range vertices marked by the running probe, so that the usual case (none, one or two)
// never reads the probe's global hash set
struct ProbeOut {
    uint32_t n, v0, v1;
};
__device__ __forceinline__ void probe_out_add(ProbeOut &P, uint32_t v) {
    P.v0 = P.n == 0u ? v : P.v0;  // selects, not an indexed store: the struct has to stay in registers
    P.v1 = P.n == 1u ? v : P.v1;
    P.n += 1;
}
struct FiltKey {
    uint32_t word;
    uint64_t mask;
};
__device__ __forceinline__ FiltKey filt_key(uint32_t v) {
    v *= 0x85EBCA6Bu;
    v ^= v >> 15;
    return FiltKey{v & (FILT_WORDS - 1u), (1ull << ((v >> 10) & 63u)) | (1ull << ((v >> 16) & 63u))};
}
__device__ __forceinline__ void filt_set(uint64_t *f, uint32_t v) {
    const FiltKey k = filt_key(v);
    atomicOr((unsigned long long *)&f[k.word], (unsigned long long)k.mask);
}

struct WalkCtx {
    TravGraph G;
    TravContig C;
    // visited state of this job: stamps for vertices on the contig strand, small hash sets for the rest
    uint32_t *stamp;   // walkStraight uniqueTable marks: PROBE_GROUPS arrays of [in_hi - in_lo] generation stamps
    uint32_t stamp_stride;
    uint32_t *tbits;   // travelUniqueTable over [in_lo, in_hi): epoch of the append per vertex, 0 = not visited
    uint64_t *tset_o;  // travelUniqueTable, vertices outside the strand's id range: (vertex | epoch << 32)
    uint32_t epoch;    // graphTravel iteration: marks with a later epoch do not exist yet for a probe of this one
    uint32_t tmask_o;
    uint64_t *pset_o;  // walkStraight uniqueTable, outside the range: PROBE_GROUPS tables of pmask_o + 1 entries
    uint32_t pmask_o;
    uint32_t n_out;    // entries in the outside sets (load-factor guard)
    uint32_t gen;
    uint32_t win_g0, win_g1;  // ctgGlobalPosTable
    uint32_t win_t0, win_t1;  // ctgTravelPosTable
    uint32_t win_p0, win_p1;  // walkStraight's ctgPosTable
    // LDS window: vertices in_lo + [w_d0, w_d0 + w_nid), records [w_r0, w_r0 + w_nrec)
    uint32_t w_d0, w_nid, w_r0, w_nrec;
    uint32_t w_anchor;  // offset the window was last filled for
    uint32_t w_ab;      // the abundance copy L.wab starts this many ids before the window (16-byte aligned loads)
    uint32_t n_fill;
    uint32_t n_classify, n_probe, n_records;  // work counters
    // what a later splice of this job's path into another walk has to know (TravJob::seq_x, TravJobOut::wd_*), per lane:
    uint32_t x_elow, x_m0;         // running iteration: lowest coordinate of an examined contig-following record / lowest id
                                   // of an examined record without a contig coordinate
    uint32_t x_below, x_forced;    // whole job: window-dependent records below / at or above force_low
    uint32_t x_poison;             // whole job: a poison record was examined (TravGraph::incomplete)
    uint32_t force_low;
    int overflow;
    int spec_fail;  // a zombie probe ended in a leap (or could not be continued): the job has to be redone without speculation
    uint64_t max_probe;  // per lane: largest probe size (sum of steps) seen, wave maximum taken at the end of the job
#ifdef PAG_WALK_PROF
    uint64_t pt[14];
    uint32_t pc[14];
#endif
};

// development aid (make WALK_PROF=1): cycles (s_memtime) and counts per section of a walk, reported through TravJobOut
#ifdef PAG_WALK_PROF
#define PROF_BEGIN(name) const uint64_t name = __builtin_amdgcn_s_memtime()
#define PROF_END(X, i, name)                                  \
    do {                                                     \
        (X).pt[i] += __builtin_amdgcn_s_memtime() - (name); \
        (X).pc[i] += 1;                                      \
    } while (0)
#else
#define PROF_BEGIN(name) do { } while (0)
#define PROF_END(X, i, name) do { } while (0)
#endif

__device__ __forceinline__ bool in_win(uint32_t lo, uint32_t hi, uint32_t p) { return p >= lo && p <= hi; }
__device__ __forceinline__ void win_add(uint32_t &lo, uint32_t &hi, uint32_t p) {
    if (p == 0) return;
    lo = p < lo ? p : lo;
    hi = p > hi ? p : hi;
}
__device__ __forceinline__ bool in_range(const WalkCtx &X, uint32_t u) { return u >= X.C.in_lo && u < X.C.in_hi; }
__device__ __forceinline__ uint32_t stamp_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Stamps are private to one wavefront.  The store is a plain (workgroup-scope) write-through store, which
// KEEPS the line in the XCD's L2 — an agent-scope (sc1) store would drop it and send the next stamp load
// of the neighbouring vertex to memory; the loads stay L2-served (sc1) so they never read a stale L1 line.
__device__ __forceinline__ void stamp_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ SuccRec rec_load(const WalkLds &L, const WalkCtx &X, uint32_t idx) {
    const uint32_t e = idx - X.w_r0;
    const bool in = e < X.w_nrec;
    SuccRec r = L.wrec[in ? e : 0u];  // always an LDS read; the global read only under its own (rare) branch
    // pinning the LDS result in registers keeps the compiler from sinking the two reads into one FLAT load
    // through a selected pointer (a flat load of LDS data is slower and waits on both memory counters)
    asm volatile("" : "+v"(r.tgt), "+v"(r.pc), "+v"(r.meta), "+v"(r.toff));
    if (!in) r = X.G.succ[idx];
    return r;
}

// mark vertex u (on the strand) in probe group grp's generation stamps: global array + window copy
__device__ __forceinline__ void stamp_put(WalkLds &L, const WalkCtx &X, uint32_t grp, uint32_t u, uint32_t gen) {
    const uint32_t d = u - X.C.in_lo;
    stamp_store(&X.stamp[(uint64_t)grp * X.stamp_stride + d], gen);
    const uint32_t e = d - X.w_d0;
    if (e < X.w_nid) L.wst[grp][e] = gen;
}

// (Re)load the window around strand offset d = anchor - in_lo.  All lanes.  The copies are LDS-direct
// loads (global_load_lds: memory -> LDS without passing through registers), so the ~190 loads per lane of a
// refill are all in flight together and a refill costs a few memory round trips instead of one per batch.
// Lanes past the end of a range re-load its last element (the slots they fill are never read).
typedef __attribute__((address_space(3))) void *lds_ptr_t;
__device__ __forceinline__ void win_fill(WalkLds &L, WalkCtx &X, uint32_t anchor) {
    PROF_BEGIN(t_fill);
    const uint32_t lane = lane_id();
    const uint32_t d = anchor - X.C.in_lo, span = X.C.in_hi - X.C.in_lo;
    const uint32_t d0 = (d > WIN_BACK ? d - WIN_BACK : 0u) & ~31u;
    const uint32_t nid = span - d0 < WIN_IDS ? span - d0 : WIN_IDS;
    const uint32_t r0 = X.G.succ_off[X.C.in_lo + d0], r1 = X.G.succ_off[X.C.in_lo + d0 + nid];
    const uint32_t nrec = r1 - r0 < WIN_REC ? r1 - r0 : WIN_REC;
    uint32_t ab_shift = 0;
    __syncthreads();
    if (nrec) {
        const SuccRec *src = X.G.succ + r0;
        for (uint32_t b = 0; b < nrec; b += 64u) {
            const uint32_t i = b + lane < nrec ? b + lane : nrec - 1u;
            __builtin_amdgcn_global_load_lds((const void *)(src + i), (lds_ptr_t)&L.wrec[b], 16, 0, 0);
        }
    }
    if (nid) {
        // 4-byte arrays: four ids per lane and instruction (b128).  The stamp / epoch arrays are padded by the host to a
        // multiple of four ids and d0 is a multiple of 32, so the addresses are 16-byte aligned; a lane past the end
        // re-loads the last quad of the array (the LDS slots it fills are never read).
        const uint32_t last_quad = X.stamp_stride - 4u - d0;  // (stamp_stride >= span rounded up to 4, d0 < span)
#pragma unroll
        for (int g = 0; g < PROBE_GROUPS; ++g) {
            const uint32_t *src = X.stamp + (uint64_t)g * X.stamp_stride + d0;
            for (uint32_t b = 0; b < nid; b += 256u) {
                const uint32_t i = b + 4u * lane < last_quad ? b + 4u * lane : last_quad;
                // sc1: the stamps were written through to the L2 by this wave, an L1 line may be older
                __builtin_amdgcn_global_load_lds((const void *)(src + i), (lds_ptr_t)&L.wst[g][b], 16, 0, 16);
            }
        }
        {
            const uint32_t *src = X.tbits + d0;
            for (uint32_t b = 0; b < nid; b += 256u) {
                const uint32_t i = b + 4u * lane < last_quad ? b + 4u * lane : last_quad;
                __builtin_amdgcn_global_load_lds((const void *)(src + i), (lds_ptr_t)&L.wts[b], 16, 0, 16);
            }
        }
        {   // the abundances are indexed by vertex id: aligned down, the window copy starts `shift` ids early
            const uint32_t first = X.C.in_lo + d0, shift = first & 3u;
            const uint32_t *src = X.G.ucnt + (first - shift);
            const uint32_t n4 = (nid + shift + 3u) & ~3u, lastq = n4 - 4u;
            for (uint32_t b = 0; b < n4; b += 256u) {
                const uint32_t i = b + 4u * lane < lastq ? b + 4u * lane : lastq;
                __builtin_amdgcn_global_load_lds((const void *)(src + i), (lds_ptr_t)&L.wab[b], 16, 0, 0);
            }
            ab_shift = shift;
        }
        const uint32_t nw = (nid + 31u) / 32u;  // <= WIN_IDS / 32 <= 64
        const uint32_t w = lane < nw ? lane : nw - 1u;
        // (an LDS-direct load writes one slot PER ACTIVE LANE: only the lanes that own a word of wgb may take part,
        // the others would write past its end, into the abundances)
        if (lane < WIN_IDS / 32u) {
            if (X.C.gbits) __builtin_amdgcn_global_load_lds((const void *)(X.C.gbits + ((X.C.in_lo - X.C.g_lo + d0) >> 5) + w), (lds_ptr_t)&L.wgb[0], 4, 0, 0);
            else L.wgb[lane] = 0u;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    X.w_d0 = d0;
    X.w_nid = nid;
    X.w_r0 = r0;
    X.w_nrec = nrec;
    X.w_anchor = d;
    X.w_ab = ab_shift;
    X.n_fill += 1;
    PROF_END(X, 6, t_fill);
}

// true if vertex `cur` with records [off, off + cnt) is served by the window with room to spare (per lane)
__device__ __forceinline__ bool win_comfortable(const WalkCtx &X, uint32_t cur, uint32_t off, uint32_t cnt) {
    const uint32_t d = cur - X.C.in_lo, span = X.C.in_hi - X.C.in_lo;
    const uint32_t e = d - X.w_d0;
    const bool ok = (e < X.w_nid) & ((e + WIN_AHEAD <= X.w_nid) | (X.w_d0 + X.w_nid >= span)) & ((e >= 32u) | (X.w_d0 == 0u)) &
                    ((off - X.w_r0 + cnt <= X.w_nrec) | (X.w_nrec != WIN_REC));
    return ok | (d >= span);  // a vertex off the strand has nothing to follow
}

// true if vertex `cur` and its records are inside the window (per lane); vertices off the strand count as served
__device__ __forceinline__ bool win_serves(const WalkCtx &X, uint32_t cur, uint32_t off, uint32_t cnt) {
    const uint32_t d = cur - X.C.in_lo, span = X.C.in_hi - X.C.in_lo;
    const uint32_t e = d - X.w_d0;
    return ((e < X.w_nid) & (off - X.w_r0 + cnt <= X.w_nrec)) | (d >= span);
}

// Keep the window around vertex `cur` whose records [off, off + cnt) are about to be read (uniform call).
__device__ __forceinline__ void win_follow(WalkLds &L, WalkCtx &X, uint32_t cur, uint32_t off, uint32_t cnt) {
    if (!in_range(X, cur)) return;
    const uint32_t d = cur - X.C.in_lo, span = X.C.in_hi - X.C.in_lo;
    const uint32_t e = d - X.w_d0;
    bool need = e >= X.w_nid;                                                        // outside
    need = need || (e + WIN_AHEAD > X.w_nid && X.w_d0 + X.w_nid < span);               // close to the upper end
    need = need || (e < 32u && X.w_d0 > 0u);                                          // close to the lower end
    need = need || (off - X.w_r0 + cnt > X.w_nrec && X.w_nrec == WIN_REC);            // records cut short
    if (!need) return;
    const uint32_t moved = d > X.w_anchor ? d - X.w_anchor : X.w_anchor - d;
    if (X.w_nid != 0 && moved < 64u) return;  // dense repeat region: do not thrash, the global arrays serve it
    win_fill(L, X, cur);
}

// bookkeeping for a later splice (see WalkCtx::x_*): every examined record passes here
__device__ __forceinline__ void walk_note_record(WalkCtx &X, uint32_t v, uint32_t pc, bool ectg, uint32_t grade = 0u, bool can_leap = false) {
    // the marker records of a graph that holds a region only (TravGraph::incomplete): grade 7 stands for ALL successors of its
    // vertex; grade 6 stands for the coordinate-free ones of a vertex on a contig (grade Skip), which a classification only
    // admits once leaping is possible.  They are no successors: nothing else is noted about them.
    X.x_poison |= ((grade == GRADE_POISON) | ((grade == GRADE_POISON_IF_LEAP) & can_leap)) ? 1u : 0u;
    const bool real = grade < GRADE_POISON_IF_LEAP;
    pc = real ? pc : 0xFFFFFFFFu;
    ectg = real ? ectg : true;  // (pc != 0 and "follows the contig" with a coordinate above every window: no entry below moves)
    X.x_m0 = ((pc == 0u) & (v < X.x_m0)) ? v : X.x_m0;
    X.x_elow = ((pc != 0u) & ectg & (pc < X.x_elow)) ? pc : X.x_elow;
    const bool wd = (pc != 0u) & !ectg;
    X.x_below = (wd & (pc < X.force_low) & (pc > X.x_below)) ? pc : X.x_below;
    X.x_forced = (wd & (pc >= X.force_low) & (pc < X.x_forced)) ? pc : X.x_forced;
}

// classifySuccessors (PAlgorithm.tcc:35-90) over the precomputed successor records of `cur`.
// level 1: filter of graphTravel (global && travel); level 2: filter of walkStraight (&& probe).
// Result: L.lst_v/lst_s[0] hold the chosen class in reference order, return = its size.
// per-lane evaluation of one successor record: class 0 Amazing/leap, 1 Excellent, 2 Good, 3 Skip, -1 rejected
// Written without short-circuit logic: every lane issues the same five LDS reads (window marks for a target
// on the strand, filter words for a target off it) and combines plain bit operations; the rare cases — a
// strand vertex outside the window, a filter hit, a probe with more than two outside vertices, a leap —
// are resolved afterwards under one branch each.
__device__ __forceinline__ int eval_record(const WalkLds &L, WalkCtx &X, const SuccRec &rec, bool can_leap, int level,
                                           uint32_t grp, const ProbeOut po, uint32_t epoch, uint32_t gen) {
    const uint32_t v = rec.tgt, pc = rec.pc;
    const uint32_t grade = (rec.meta >> 24) & 7u;
    const bool ectg = (rec.meta >> 27) & 1u;
    const uint32_t d = v - X.C.in_lo, e = d - X.w_d0;
    const bool inr = d < X.C.in_hi - X.C.in_lo;
    const bool inw = inr & (e < X.w_nid);
    const uint32_t ei = inw ? e : 0u;
    const uint32_t bit = 1u << (ei & 31u);
    const uint32_t tsw = L.wts[ei], gww = L.wgb[ei >> 5], stw = L.wst[grp][ei];
    const FiltKey fk = filt_key(v);
    const uint64_t ftw = L.ft[fk.word], fgw = L.fg[fk.word];
    bool tvis = inw & (tsw != 0u) & (tsw <= epoch);
    bool gvis = inw & ((gww & bit) != 0u);
    bool pvis = (inw & (stw == gen)) | (!inr & (((po.n >= 1u) & (v == po.v0)) | ((po.n >= 2u) & (v == po.v1))));
    const bool fth = !inr & ((ftw & fk.mask) == fk.mask), fgh = !inr & ((fgw & fk.mask) == fk.mask);
    // a strand vertex outside the range of the job's own arrays (segment jobs): its global mark is in the strand's bitmap
    const uint32_t dg = v - X.C.g_lo;
    const bool ing = !inr & (dg < X.C.g_hi - X.C.g_lo) & (X.C.gbits != nullptr);
    if ((inr & !inw) | fth | fgh | ing | (!inr & (po.n > 2u))) {
        if (inr) {
            {
                const uint32_t ts = stamp_load(&X.tbits[d]);
                tvis = (ts != 0u) & (ts <= epoch);
            }
            gvis = X.C.gbits ? (X.C.gbits[dg >> 5] >> (dg & 31u)) & 1u : false;
            if (level == 2) pvis = stamp_load(&X.stamp[(uint64_t)grp * X.stamp_stride + d]) == gen;
        } else {
            if (ing) gvis = (X.C.gbits[dg >> 5] >> (dg & 31u)) & 1u;
            else if (fgh) gvis = hs_has(X.C.gset, X.C.gmask, v);
            if (fth) {
                const uint32_t ts = hs64_epoch(X.tset_o, X.tmask_o, v);
                tvis = (ts != 0u) & (ts <= epoch);
            }
            if (level == 2 && po.n > 2u) pvis = gs_has(X.pset_o + (uint64_t)grp * ((uint64_t)X.pmask_o + 1), X.pmask_o, v, gen);
        }
    }
    const bool free_pc = (pc == 0u) | ectg;  // no coordinate, or the edge follows the contig: the window tests do not apply
    walk_note_record(X, v, pc, ectg, grade, can_leap);
    const bool hit_g = !free_pc & in_win(X.win_g0, X.win_g1, pc), hit_t = !free_pc & in_win(X.win_t0, X.win_t1, pc);
    const bool rev = (pc != 0u) & (pc >= X.C.rev_left) & (pc < X.C.rev_right);
    bool ok = !(gvis | hit_g | rev | tvis | hit_t);
    if (level == 2) ok = ok & !(pvis | (!free_pc & in_win(X.win_p0, X.win_p1, pc)));
    const bool leap = (pc != 0u) & ((pc < X.C.ctg_left) | (pc >= X.C.ctg_right));
    if (ok & leap) {
        // landing rule (PAlgorithm.tcc:60-67): singleToDual (PositionMapper.cpp:44-64) on the start table
        uint32_t lo2 = 0, hi2 = X.C.n_ctgs + 1;
        while (lo2 < hi2) {  // upper_bound(starts, pc)
            uint32_t mid = (lo2 + hi2) >> 1;
            if (X.C.starts[mid] <= (uint64_t)pc) lo2 = mid + 1;
            else hi2 = mid;
        }
        uint32_t idx = lo2 ? lo2 - 1 : 0;
        uint64_t off = (uint64_t)pc - X.C.starts[idx];
        uint64_t sz = idx < X.C.n_ctgs ? X.C.sizes[idx] : 0;
        if (off >= 2 * sz) off -= 2 * sz;
        ok = !((double)(int64_t)off > (double)sz * X.C.leap_min) & can_leap;
    }
    // class by grade through a nibble table: Amazing 0, Excellent 1, Good 2, Skip 3 (only once leaping is allowed)
    // (grade 7 = the poison record of a vertex whose successors lie outside the region this rank holds: never accepted)
    const uint32_t table = can_leap ? 0xFFF0123Fu : 0xFFF012FFu;
    const uint32_t c4 = leap ? 0u : (table >> (grade * 4u)) & 0xFu;
    return (ok & (c4 != 0xFu)) ? (int)c4 : -1;
}

// classifySuccessors (PAlgorithm.tcc:35-90) over the precomputed successor records of `cur`.
// level 1: filter of graphTravel (global && travel); level 2: filter of walkStraight (&& probe).
// Returns the size n of the chosen class.  n == 1: the successor is returned in *one_v/*one_s/*one_pc and
// nothing touches LDS (the common case of a straight walk).  n > 1: the chosen class is in
// L.lst_v/lst_s[0] in reference order.
// The single survivor of a classification and, speculatively, its own only successor record
struct Step {
    uint32_t v, s, pc;   // vertex (new id), step, contig coordinate
    uint32_t off, cnt;   // its successor records [off, off + cnt); cnt == 15 means "15 or more"
    bool have_next;      // cnt == 1 and `next` already holds that record (loaded while the stamps were in flight)
    SuccRec next;
};

__device__ __forceinline__ uint32_t classify(WalkLds &L, WalkCtx &X, uint32_t r0, uint32_t cnt, bool have_pre, const SuccRec &pre, bool can_leap,
                             int level, const ProbeOut po, Step *one, bool *list_meta = nullptr, uint32_t *gl_v = nullptr, uint32_t *gl_s = nullptr,
                             uint64_t gl_cap = 0) {
    // list_meta != nullptr: the caller wants the chosen class as a list (graphTravel, level 1); walkStraight (level 2) only asks
    // how many there are.  The list of a classification of at most 64 records is L.lst_* (*list_meta = true: with every
    // vertex's own data); of more than 64 records it is gl_v / gl_s[0 .. n) in global memory (*list_meta = false) — n > gl_cap
    // sets X.overflow.
    const uint32_t lane = lane_id();
    const uint32_t r1 = r0 + cnt;
    X.n_classify += 1;
    X.n_records += cnt;
    if (cnt <= 64) {
        int cls = -1;
        SuccRec rec{0, 0, 0, 0}, nx{0, 0, 0, 0};
        if (lane < cnt) {
            rec = have_pre ? pre : rec_load(L, X, r0 + lane);
            // speculative: the target's only successor record, requested together with the stamp
            if ((rec.meta >> 28) == 1u) nx = rec_load(L, X, rec.toff);
            cls = eval_record(L, X, rec, can_leap, level, 0u, po, X.epoch, X.gen);
        }
        uint64_t m = __ballot(cls == 0);
        if (!m) m = __ballot(cls == 1);
        if (!m) m = __ballot(cls == 2);
        if (!m) m = __ballot(cls == 3);
        uint32_t n = (uint32_t)__popcll(m);
        if (n == 0) return 0;
        if (n == 1) {
            int src = __ffsll((long long)m) - 1;
            uint32_t meta = __shfl(rec.meta, src, 64);
            one->v = __shfl(rec.tgt, src, 64);
            one->s = meta & 0xFFFFFFu;
            one->pc = __shfl(rec.pc, src, 64);
            one->off = __shfl(rec.toff, src, 64);
            one->cnt = meta >> 28;
            one->have_next = one->cnt == 1u;
            one->next.tgt = __shfl(nx.tgt, src, 64);
            one->next.pc = __shfl(nx.pc, src, 64);
            one->next.meta = __shfl(nx.meta, src, 64);
            one->next.toff = __shfl(nx.toff, src, 64);
            return 1;
        }
        if (!list_meta) return n;
        __syncthreads();
        if ((m >> lane) & 1ull) {
            uint32_t at = (uint32_t)__popcll(m & lanemask_lt());
            L.lst_v[0][at] = rec.tgt;
            L.lst_s[0][at] = rec.meta & 0xFFFFFFu;
            L.lst_pc[at] = rec.pc;
            L.lst_off[at] = rec.toff;
            L.lst_cnt[at] = rec.meta >> 28;
        }
        *list_meta = true;
        __syncthreads();
        return n;
    }
    if (list_meta) *list_meta = false;
    // more than 64 successor records (repeats), 64 at a time.  First pass: how many records of every class, and the first of
    // each (all wave-uniform); the chosen class is written out by a second pass over the records, in reference order.
    uint32_t cn[4] = {0, 0, 0, 0}, fv[4] = {0, 0, 0, 0}, fs[4] = {0, 0, 0, 0};
    for (uint32_t rb = r0; rb < r1; rb += 64) {
        int cls = -1;
        SuccRec rec{0, 0, 0, 0};
        if (rb + lane < r1) {
            rec = rec_load(L, X, rb + lane);
            cls = eval_record(L, X, rec, can_leap, level, 0u, po, X.epoch, X.gen);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint64_t m = __ballot(cls == c);
            if (m == 0) continue;
            if (cn[c] == 0) {
                const int src = __ffsll((long long)m) - 1;
                fv[c] = __shfl(rec.tgt, src, 64);
                fs[c] = __shfl(rec.meta, src, 64) & 0xFFFFFFu;
            }
            cn[c] += (uint32_t)__popcll(m);
        }
    }
    const int chosen = cn[0] ? 0 : (cn[1] ? 1 : (cn[2] ? 2 : 3));
    const uint32_t n = cn[chosen];
    if (n == 0) return 0;
    if (n == 1) {
        one->v = chosen == 0 ? fv[0] : chosen == 1 ? fv[1] : chosen == 2 ? fv[2] : fv[3];
        one->s = chosen == 0 ? fs[0] : chosen == 1 ? fs[1] : chosen == 2 ? fs[2] : fs[3];
        one->pc = (uint32_t)(X.G.upos[one->v] >> 32);
        one->off = X.G.succ_off[one->v];
        uint32_t c2 = X.G.succ_off[one->v + 1] - one->off;
        one->cnt = c2 < 15u ? c2 : 15u;
        one->have_next = false;
        return 1;
    }
    if (!list_meta) return n;
    if ((uint64_t)n > gl_cap) {  // (the host doubles the job's buffers and posts it again)
        X.overflow = 1;
        return n;
    }
    uint32_t base = 0;
    for (uint32_t rb = r0; rb < r1; rb += 64) {
        int cls = -1;
        SuccRec rec{0, 0, 0, 0};
        if (rb + lane < r1) {
            rec = rec_load(L, X, rb + lane);
            cls = eval_record(L, X, rec, can_leap, level, 0u, po, X.epoch, X.gen);
        }
        const uint64_t m = __ballot(cls == chosen);
        if (cls == chosen) {
            const uint32_t at = base + (uint32_t)__popcll(m & lanemask_lt());
            gl_v[at] = rec.tgt;
            gl_s[at] = rec.meta & 0xFFFFFFu;
        }
        base += (uint32_t)__popcll(m);
    }
    __threadfence_block();
    __syncthreads();
    return n;
}

// mark a vertex in walkStraight's uniqueTable (one lane)
__device__ __forceinline__ void probe_mark(WalkLds &L, WalkCtx &X, uint32_t u) {
    if (in_range(X, u)) {
        stamp_put(L, X, 0u, u, X.gen);  // group 0 arrays serve the sequential mode
    } else {
        gs_insert_single(X.pset_o, X.pmask_o, u, X.gen);
    }
}

enum { WS_END = 0, WS_BRANCH = 1, WS_LIMIT = 2, WS_LEAP = 3 };

// walkStraight (PAlgorithm.tcc:93-170): writes the path to pv/ps (capacity cap), returns status
__device__ __forceinline__ int walk_straight(WalkLds &L, WalkCtx &X, uint32_t v0, uint32_t s0, uint64_t has_size, uint32_t *pv, uint32_t *ps,
                             uint64_t cap, uint64_t *out_len) {
    const uint32_t lane = lane_id();
    X.gen += 1;
    X.n_probe += 1;
    X.win_p0 = 0xFFFFFFFFu;
    X.win_p1 = 0;
    uint64_t now_size = s0, len = 0;
    if (cap == 0) {
        X.overflow = 1;
        *out_len = 0;
        return WS_END;
    }
    if (lane == 0) {
        pv[0] = v0;
        ps[0] = s0;
    }
    len = 1;
    uint32_t c = (uint32_t)(X.G.upos[v0] >> 32);
    X.max_probe = now_size > X.max_probe ? now_size : X.max_probe;
    if (c != 0 && (c < X.C.ctg_left || c >= X.C.ctg_right)) {
        *out_len = len;
        return WS_LEAP;
    }
    win_add(X.win_p0, X.win_p1, c);
    ProbeOut po{0, 0, 0};
    if (lane == 0) probe_mark(L, X, v0);
    if (!in_range(X, v0)) probe_out_add(po, v0);
    __syncthreads();
    uint32_t cur = v0;
    uint32_t off = X.G.succ_off[v0], cnt = X.G.succ_off[v0 + 1] - off;
    bool have_pre = false;
    SuccRec pre{0, 0, 0, 0};
    int status;
    for (;;) {
        win_follow(L, X, cur, off, cnt);
        Step st;
        uint32_t m = classify(L, X, off, cnt, have_pre, pre, (has_size + now_size) >= X.C.split_size, 2, po, &st);
        if (m == 0) {
            status = WS_END;
            break;
        }
        if (m > 1) {
            status = WS_BRANCH;
            break;
        }
        if (len >= cap || (uint64_t)(po.n + 1) * 2 > (uint64_t)X.pmask_o) {
            X.overflow = 1;
            status = WS_END;
            break;
        }
        // same-wave stores and later loads of one address stay ordered in the memory pipeline, so the
        // mark needs no wait before the next step's stamp loads
        if (lane == 0) {
            probe_mark(L, X, st.v);
            pv[len] = st.v;
            ps[len] = st.s;
        }
        if (!in_range(X, st.v)) probe_out_add(po, st.v);
        win_add(X.win_p0, X.win_p1, st.pc);
        len += 1;
        now_size += st.s;
        if (st.pc != 0 && (st.pc < X.C.ctg_left || st.pc >= X.C.ctg_right)) {
            status = WS_LEAP;
            break;
        }
        cur = st.v;
        off = st.off;
        cnt = st.cnt;
        have_pre = st.have_next;
        pre = st.next;
        if (cnt == 15u) {  // "15 or more": take the exact range from the offset table
            off = X.G.succ_off[cur];
            cnt = X.G.succ_off[cur + 1] - off;
            have_pre = false;
        }
    }
    __syncthreads();  // the path written by lane 0 is read by all lanes afterwards
    *out_len = len;
    X.max_probe = now_size > X.max_probe ? now_size : X.max_probe;
    return status;
}

// what a probe group knows when it stops (uniform inside the group)
struct ProbeRes {
    int status;
    uint32_t len;
    uint32_t last_v, last_pc;  // last path vertex and its contig coordinate
    uint32_t off, cnt;         // successor records of last_v (exact) — meaningful for END / BRANCH
    uint32_t w0, w1;           // min / max contig coordinate over the path (0 = none)
    uint32_t n_out;            // path vertices outside the strand's id range
    uint32_t ab;               // abundance of the first path vertex
    uint64_t size;             // sum of the steps
};

// ---------------------------------------------------------------------------------------------------------------
// Probe SLOTS.  The four lane groups are slots that outlive a graphTravel iteration.  Of the alternatives of a branch
// almost all stop at the next branching vertex, and graphTravel then takes the one with the most abundant first
// vertex (the first one to leap would win over all of them, PAlgorithm.tcc:268-296).  So once an alternative A has
// stopped at a branch, a still-running alternative j with a lower claim (smaller abundance, or equal and later in
// order) can change the choice in ONE way only: by ending in a leap.  Waiting for that costs the longest probe of
// every branch (10.7 steps at C2) where the choice is known after 3.  Instead j becomes a ZOMBIE: the main walk goes
// on with A, j keeps walking in its slot — against the state of ITS iteration: travel marks are epochs and it only
// sees those up to its own, it keeps its own copy of the travel window and of the accumulated size — and if it ever
// ends in a leap the job is reported as mis-speculated and the host runs it again with speculation off.  A job ends
// only after its zombies have.  An alternative that started after a finished leap of its iteration can never be
// chosen and is simply dropped.
struct Slot {
    int status;       // < 0: walking; otherwise WS_* of the last probe
    uint32_t fresh;   // the result belongs to the running iteration and has not been consumed
    uint32_t zombie;  // walking, but only a leap would still matter
    uint32_t epoch, gen, alt;
    uint32_t cur_v, off, cnt, len, last_pc, ab;
    uint32_t wp0, wp1, wt0, wt1;  // the probe's coordinate window (start vertex included), the travel window of its iteration
    ProbeOut po;
    uint64_t tot;  // size walked so far INCLUDING the size of the sequence when the probe started (leaping needs the sum)
    uint64_t base; // ... that size of the sequence (tot - base = the probe's own size)
    uint32_t pb_v, pb_s;
};

// A slot that reaches a vertex with more records than it has lanes cannot go on inside its lane group: the whole wave
// walks it to its end (scalar walk state, the other slots wait).  Of a zombie only the final status matters; a probe
// of the running iteration also leaves its path, its windows and the accepted records of its last vertex behind.
__device__ __forceinline__ bool slot_finish_wide(WalkLds &L, WalkCtx &X, Slot &S, uint32_t g, uint32_t *arena_v, uint32_t *arena_s,
                                                 uint64_t cap_each) {
    const uint32_t lane = lane_id();
    const int src = (int)(GL * g);
    uint64_t *pset = X.pset_o + (uint64_t)g * ((uint64_t)X.pmask_o + 1);
    uint32_t *pv = arena_v + (uint64_t)g * cap_each, *ps = arena_s + (uint64_t)g * cap_each;
    int status = __shfl(S.status, src, 64);
    const bool zombie = __shfl(S.zombie, src, 64) != 0u;
    const uint32_t epoch = __shfl(S.epoch, src, 64), gen = __shfl(S.gen, src, 64);
    uint32_t cur = __shfl(S.cur_v, src, 64), off = __shfl(S.off, src, 64), cnt = __shfl(S.cnt, src, 64);
    uint32_t wp0 = __shfl(S.wp0, src, 64), wp1 = __shfl(S.wp1, src, 64);
    uint32_t last_pc = __shfl(S.last_pc, src, 64);
    const uint32_t wt0 = __shfl(S.wt0, src, 64), wt1 = __shfl(S.wt1, src, 64);
    ProbeOut po{__shfl(S.po.n, src, 64), __shfl(S.po.v0, src, 64), __shfl(S.po.v1, src, 64)};
    uint64_t tot = __shfl(S.tot, src, 64);
    uint32_t len = __shfl(S.len, src, 64);
    if (!zombie && (lane >> GL_SHIFT) == g && (lane & (GL - 1u)) < (len & (GL - 1u))) {  // path entries waiting in registers
        pv[len - (len & (GL - 1u)) + (lane & (GL - 1u))] = S.pb_v;
        ps[len - (len & (GL - 1u)) + (lane & (GL - 1u))] = S.pb_s;
    }
    int fail = 0;
    bool too_wide = false;
    while (status < 0) {
        if (cnt > 64u) {
            // more successor records than lanes: a zombie fails the speculation; for a probe of the running iteration
            // the caller probes this iteration again sequentially (sticky flag, the slot is parked as a dead end)
            if (zombie) fail |= 2;
            else too_wide = true;
            status = WS_END;
            break;
        }
        win_follow(L, X, cur, off, cnt);
        X.n_classify += 1;
        const bool can_leap = tot >= X.C.split_size;
        int cls = -1;
        SuccRec rec{0, 0, 0, 0};
        if (lane < cnt) {
            rec = rec_load(L, X, off + lane);
            const uint32_t sp0 = X.win_p0, sp1 = X.win_p1, st0 = X.win_t0, st1 = X.win_t1;
            X.win_p0 = wp0;
            X.win_p1 = wp1;
            X.win_t0 = wt0;
            X.win_t1 = wt1;
            cls = eval_record(L, X, rec, can_leap, 2, g, po, epoch, gen);
            X.win_p0 = sp0;
            X.win_p1 = sp1;
            X.win_t0 = st0;
            X.win_t1 = st1;
        }
        uint64_t m = __ballot(cls == 0);
        if (!m) m = __ballot(cls == 1);
        if (!m) m = __ballot(cls == 2);
        if (!m) m = __ballot(cls == 3);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n != 1u) {
            status = n == 0 ? WS_END : WS_BRANCH;
            if (!zombie) {  // the accepted records, for the classification of the chosen path's last vertex
                const uint64_t ga = __ballot(cls >= 0);
                if (cls >= 0 && (uint32_t)__popcll(ga & lanemask_lt()) < (uint32_t)PB_CAP) {  // (more than PB_CAP: the count says so, see walk_job)
                    const uint32_t kk = (uint32_t)__popcll(ga & lanemask_lt());
                    L.pb_v[g][kk] = rec.tgt;
                    L.pb_meta[g][kk] = rec.meta;
                    L.pb_pc[g][kk] = rec.pc;
                    L.pb_off[g][kk] = rec.toff;
                    L.pb_cls[g][kk] = (uint32_t)cls;
                }
                if (lane == 0) L.pb_cnt[g] = (uint32_t)__popcll(ga);
            }
            break;
        }
        if ((!zombie && len >= cap_each) || (uint64_t)(po.n + 1) * 2 > (uint64_t)X.pmask_o) {
            if (zombie) fail |= 4;
            else X.overflow = 1;
            status = WS_END;
            break;
        }
        const int sl = __ffsll((long long)m) - 1;
        const uint32_t meta = __shfl(rec.meta, sl, 64), nv = __shfl(rec.tgt, sl, 64), npc = __shfl(rec.pc, sl, 64), noff = __shfl(rec.toff, sl, 64);
        if (lane == 0) {
            if (in_range(X, nv)) stamp_put(L, X, g, nv, gen);
            else gs_insert_single(pset, X.pmask_o, nv, gen);
            if (!zombie) {
                pv[len] = nv;
                ps[len] = meta & 0xFFFFFFu;
            }
        }
        if (!in_range(X, nv)) probe_out_add(po, nv);
        win_add(wp0, wp1, npc);
        last_pc = npc;
        len += 1;
        tot += meta & 0xFFFFFFu;
        cur = nv;
        if (npc != 0 && (npc < X.C.ctg_left || npc >= X.C.ctg_right)) {
            status = WS_LEAP;
            break;
        }
        off = noff;
        cnt = meta >> 28;
        if (cnt == 15u) {
            off = X.G.succ_off[nv];
            cnt = X.G.succ_off[nv + 1] - off;
        }
    }
    if (zombie && status == WS_LEAP) fail |= 1;
    X.spec_fail |= fail;
    {
        const uint64_t psz = tot - __shfl(S.base, src, 64);
        X.max_probe = psz > X.max_probe ? psz : X.max_probe;
    }
    if ((lane >> GL_SHIFT) == g) {
        S.status = status;
        S.zombie = 0;
        S.fresh = zombie ? 0u : 1u;
        S.cur_v = cur;
        S.off = off;
        S.cnt = cnt;
        S.len = len;
        S.tot = tot;
        S.last_pc = last_pc;
        S.wp0 = wp0;
        S.wp1 = wp1;
        S.po = po;
    }
    return !too_wide;
}

// one step of every walking slot
__device__ __forceinline__ void slots_step(WalkLds &L, WalkCtx &X, Slot &S, uint32_t *arena_v, uint32_t *arena_s, uint64_t cap_each,
                                           bool *wide, bool drain) {
    const uint32_t lane = lane_id(), g = lane >> GL_SHIFT, sub = lane & (GL - 1u);
    uint64_t *pset = X.pset_o + (uint64_t)g * ((uint64_t)X.pmask_o + 1);
    uint32_t *pv = arena_v + (uint64_t)g * cap_each, *ps = arena_s + (uint64_t)g * cap_each;
    PROF_BEGIN(t_a);
    for (;;) {  // slots too wide for their lanes, one at a time (one copy of the wide walk in the code)
        const uint64_t wz = __ballot(S.status < 0 && S.cnt > GL);
        if (!wz) break;
        PROF_BEGIN(t_w);
        const bool okw = slot_finish_wide(L, X, S, (uint32_t)(__ffsll((long long)wz) - 1) >> GL_SHIFT, arena_v, arena_s, cap_each);
        PROF_END(X, 7, t_w);
        if (!okw) *wide = true;  // (sticky; the step goes on, the iteration is redone by the caller)
    }
    // The window belongs to the probes of the running iteration.  A zombie steps along while the window serves it;
    // once it has wandered off it is SUSPENDED (its slow global reads would be paid by every slot of the wave) and
    // is only walked on when the wave has nothing better to do: waiting for a free slot, or at the end of the job.
    const bool walking = S.status < 0;
    const bool lead = drain ? walking : (walking && !S.zombie);  // who may move the window
    const bool running = walking && (lead || win_serves(X, S.cur_v, S.off, S.cnt));
    if (__ballot(lead && !win_comfortable(X, S.cur_v, S.off, S.cnt))) {
        // the window follows the lowest leading slot; the others use it while they are inside
        uint32_t av = lead ? S.cur_v : 0xFFFFFFFFu;
        if (GL == 8u) av = min(av, (uint32_t)__builtin_amdgcn_update_dpp((int)av, (int)av, 0x128, 0xF, 0xF, false));  // row_ror:8
        const uint32_t a = min(min((uint32_t)__builtin_amdgcn_readlane((int)av, 0), (uint32_t)__builtin_amdgcn_readlane((int)av, 16)),
                               min((uint32_t)__builtin_amdgcn_readlane((int)av, 32), (uint32_t)__builtin_amdgcn_readlane((int)av, 48)));
        uint32_t ao = 0, ac = 0;
        if (a != 0xFFFFFFFFu) {
            const int al = __ffsll((long long)__ballot(lead && S.cur_v == a)) - 1;
            ao = (uint32_t)__builtin_amdgcn_readlane((int)S.off, al);
            ac = (uint32_t)__builtin_amdgcn_readlane((int)S.cnt, al);
        }
        if (a != 0xFFFFFFFFu) win_follow(L, X, a, ao, ac);
    }
    PROF_END(X, 8, t_a);
    PROF_BEGIN(t_b);
    X.n_classify += 1;
    int cls = -1;
    SuccRec rec{0, 0, 0, 0};
    const bool can_leap = S.tot >= X.C.split_size;
    if (running && sub < S.cnt) {
        rec = rec_load(L, X, S.off + sub);
        // the tests of a probe use ITS windows: the probe's own, and the travel window of its iteration
        const uint32_t sp0 = X.win_p0, sp1 = X.win_p1, st0 = X.win_t0, st1 = X.win_t1;
        X.win_p0 = S.wp0;
        X.win_p1 = S.wp1;
        X.win_t0 = S.wt0;
        X.win_t1 = S.wt1;
        cls = eval_record(L, X, rec, can_leap, 2, g, S.po, S.epoch, S.gen);
        X.win_p0 = sp0;
        X.win_p1 = sp1;
        X.win_t0 = st0;
        X.win_t1 = st1;
    }
    PROF_END(X, 9, t_b);
    PROF_BEGIN(t_c);
    // the best class present among the records of the slot (lane-group minimum by DPP), then its members
    uint32_t key = cls < 0 ? 4u : (uint32_t)cls, mn = key;
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x141, 0xF, 0xF, false));  // row_half_mirror
    if (GL == 16u) mn = min(mn, (uint32_t)__builtin_amdgcn_update_dpp((int)mn, (int)mn, 0x140, 0xF, 0xF, false));  // row_mirror
    const uint32_t cm = (uint32_t)(__ballot(key == mn && key < 4u) >> (GL * g)) & GL_MASK;
    const uint32_t n = (uint32_t)__popc(cm);
    {   // a probe of the running iteration that stops here by classification leaves its accepted records behind
        const uint32_t ga = (uint32_t)(__ballot(cls >= 0) >> (GL * g)) & GL_MASK;
        if (running && !S.zombie && n != 1u) {
            if (cls >= 0) {
                const uint32_t kk = (uint32_t)__popc(ga & ((1u << sub) - 1u));
                L.pb_v[g][kk] = rec.tgt;
                L.pb_meta[g][kk] = rec.meta;
                L.pb_pc[g][kk] = rec.pc;
                L.pb_off[g][kk] = rec.toff;
                L.pb_cls[g][kk] = (uint32_t)cls;
            }
            if (sub == 0) L.pb_cnt[g] = (uint32_t)__popc(ga);
        }
    }
    const int src = (int)(GL * g) + (cm ? __ffs(cm) - 1 : 0);
    const uint32_t meta = __shfl(rec.meta, src, 64);
    const uint32_t nv = __shfl(rec.tgt, src, 64);
    const uint32_t npc = __shfl(rec.pc, src, 64);
    const uint32_t noff = __shfl(rec.toff, src, 64);
    PROF_END(X, 10, t_c);
    PROF_BEGIN(t_d);
    // State update in predicated form (selects, no nested divergent regions: every slot of the wave takes the same
    // instruction path, and a branchy version costs ~40 register copies per step at the merge points).
    const bool full = (!S.zombie && S.len >= cap_each) || (uint64_t)(S.po.n + 1) * 2 > (uint64_t)X.pmask_o;
    const bool adv = running && n == 1u && !full;                     // the walk moves on to nv
    const bool leap = adv && npc != 0 && (npc < X.C.ctg_left || npc >= X.C.ctg_right);
    const bool inr_nv = in_range(X, nv);
    const uint32_t ns = meta & 0xFFFFFFu;
    if (running && n == 1u && full) {  // (rare)
        if (S.zombie) X.spec_fail |= 4;  // out of room for a walk whose only purpose is the check
        else X.overflow = 1;
    }
    if (adv && sub == 0) {
        if (inr_nv) stamp_put(L, X, g, nv, S.gen);
        else gs_insert_single(pset, X.pmask_o, nv, S.gen);
    }
    if (adv && !inr_nv) probe_out_add(S.po, nv);
    {
        const bool hold = adv && !S.zombie && sub == (S.len & (GL - 1u));
        S.pb_v = hold ? nv : S.pb_v;
        S.pb_s = hold ? ns : S.pb_s;
    }
    {
        const bool w = adv && npc != 0;
        S.wp0 = w && npc < S.wp0 ? npc : S.wp0;
        S.wp1 = w && npc > S.wp1 ? npc : S.wp1;
    }
    S.last_pc = adv ? npc : S.last_pc;
    S.len += adv ? 1u : 0u;
    S.tot += adv ? ns : 0u;
    if (adv && !S.zombie && (S.len & (GL - 1u)) == 0) {  // GL entries pending: one coalesced store per array
        pv[S.len - GL + sub] = S.pb_v;
        ps[S.len - GL + sub] = S.pb_s;
    }
    S.cur_v = adv ? nv : S.cur_v;
    {
        uint32_t noff2 = noff, ncnt = meta >> 28;
        if (adv && !leap && ncnt == 15u) {  // "15 or more": the exact range (rare)
            noff2 = X.G.succ_off[nv];
            ncnt = X.G.succ_off[nv + 1] - noff2;
        }
        const bool go = adv && !leap;
        S.off = go ? noff2 : S.off;
        S.cnt = go ? ncnt : S.cnt;
    }
    {
        const int st = !running ? S.status : n == 0u ? (int)WS_END : n > 1u ? (int)WS_BRANCH : full ? (int)WS_END : leap ? (int)WS_LEAP : S.status;
        const bool stopped = running && st >= 0;  // stopped in this step
        if (stopped && S.zombie && st == WS_LEAP) X.spec_fail |= 1;
        if (stopped && S.tot - S.base > X.max_probe) X.max_probe = S.tot - S.base;
        if (stopped && !S.zombie && sub < (S.len & (GL - 1u))) {  // the entries still waiting in registers
            pv[S.len - (S.len & (GL - 1u)) + sub] = S.pb_v;
            ps[S.len - (S.len & (GL - 1u)) + sub] = S.pb_s;
        }
        S.fresh = stopped && !S.zombie ? 1u : S.fresh;
        S.zombie = stopped ? 0u : S.zombie;
        S.status = st;
    }
    PROF_END(X, 11, t_d);
}

// one bit per slot: its first lane
constexpr uint64_t SLOT_LEADS = GL == 8u ? 0x0101010101010101ull : 0x0001000100010001ull;

// the most abundant first vertex among the slots in `cand` (first lanes of slots; ties: the lowest slot, which within
// one iteration is the earliest alternative).  Uniform.
__device__ __forceinline__ int slots_best(uint64_t cand, uint32_t key, uint32_t *best_key) {
    int pick = -1;
    uint32_t best = 0;
    while (cand) {
        const int l = __ffsll((long long)cand) - 1;
        cand &= cand - 1ull;
        const uint32_t kq = (uint32_t)__builtin_amdgcn_readlane((int)key, l);
        if (pick < 0 || kq > best) {
            pick = l;
            best = kq;
        }
    }
    *best_key = best;
    return pick;
}

// After a step: what the finished alternatives of the running iteration mean for the ones still walking.
// (The alternatives of one iteration sit in slots of increasing index, so "an earlier alternative" = "a lower slot".)
__device__ __forceinline__ void slots_dominate(const WalkCtx &X, Slot &S, bool speculate) {
    const bool fin = S.fresh != 0u && S.epoch == X.epoch;
    const uint64_t lm = __ballot(fin && S.status == WS_LEAP) & SLOT_LEADS;
    if (lm) {  // an earlier alternative leaps: the later ones can never be chosen
        const uint32_t alt_l = (uint32_t)__builtin_amdgcn_readlane((int)S.alt, __ffsll((long long)lm) - 1);
        if (S.status < 0 && S.epoch == X.epoch && alt_l < S.alt) {
            S.status = WS_END;
            S.zombie = 0;
            S.fresh = 0;
        }
    }
    if (!speculate) return;
    const uint64_t bm = __ballot(fin && S.status == WS_BRANCH) & SLOT_LEADS;
    if (bm) {
        uint32_t bab;
        const int bl = slots_best(bm, S.ab, &bab);
        const uint32_t balt = (uint32_t)__builtin_amdgcn_readlane((int)S.alt, bl);
        // (only while the walk is far from the size at which leaping becomes possible at all: close to it,
        // leaps of side paths are common and every one would void the whole job)
        if (S.status < 0 && S.epoch == X.epoch && !S.zombie && (bab > S.ab || (bab == S.ab && balt < S.alt)) &&
            S.tot + SPEC_MARGIN < X.C.split_size)
            S.zombie = 1;
    }
}

// Start the m alternatives L.br_* in free slots and walk until the choice among them is determined.
// Returns false if an alternative met a vertex with more than 64 records (the caller probes sequentially).
__device__ __forceinline__ bool probe_slots(WalkLds &L, WalkCtx &X, Slot &S, uint32_t m, bool have_meta, uint64_t has_size, uint32_t *arena_v,
                            uint32_t *arena_s, uint64_t cap_each, bool speculate) {
    const uint32_t lane = lane_id(), g = lane >> GL_SHIFT, sub = lane & (GL - 1u);
    bool wide = false;
    if (cap_each == 0) {
        X.overflow = 1;
        return true;
    }
    // wait for m free slots (zombies occupy theirs until they stop)
    uint64_t free_leads;
    PROF_BEGIN(t_wait);
    for (;;) {
        free_leads = __ballot(S.status >= 0) & SLOT_LEADS;
        if ((uint32_t)__popcll(free_leads) >= m) break;
        slots_step(L, X, S, arena_v, arena_s, cap_each, &wide, true);
        wide = false;  // only zombies walk here, and a zombie that gets too wide is finished inside the step
    }
    PROF_END(X, 2, t_wait);
    PROF_BEGIN(t_setup);
    X.gen += 1;
    X.n_probe += m;
    const uint32_t rank = (uint32_t)__popcll(free_leads & ((1ull << (GL * g)) - 1ull));
    {   // the free slots of rank < m take the alternatives (predicated form: one instruction path for the whole wave)
        const bool take = S.status >= 0 && rank < m;
        const uint32_t rk = take ? rank : 0u;
        uint64_t *pset = X.pset_o + (uint64_t)g * ((uint64_t)X.pmask_o + 1);
        const uint32_t v0 = L.br_v[rk], s0 = L.br_s[rk];
        uint32_t c = L.br_pc[rk], off0 = L.br_off[rk], cnt0 = L.br_cnt[rk];
        if (take && !have_meta) {  // (the list came without the vertices' own data: rare)
            c = (uint32_t)(X.G.upos[v0] >> 32);
            cnt0 = 15u;
        }
        const bool leap0 = take && c != 0 && (c < X.C.ctg_left || c >= X.C.ctg_right);
        const bool go = take && !leap0;
        if (go && cnt0 == 15u) {  // "15 or more": the exact range (rare)
            off0 = X.G.succ_off[v0];
            cnt0 = X.G.succ_off[v0 + 1] - off0;
        }
        uint32_t ab0;
        {   // abundance of the alternative (needed if it ends in a branch)
            const uint32_t e0 = v0 - X.C.in_lo - X.w_d0;
            ab0 = L.wab[e0 < X.w_nid ? e0 + X.w_ab : 0u];
            if (take && !(e0 < X.w_nid)) ab0 = X.G.ucnt[v0];
        }
        const bool inr0 = in_range(X, v0);
        if (leap0 && sub == 0) {
            arena_v[(uint64_t)g * cap_each] = v0;
            arena_s[(uint64_t)g * cap_each] = s0;
        }
        if (go && sub == 0) {
            if (inr0) stamp_put(L, X, g, v0, X.gen);
            else gs_insert_single(pset, X.pmask_o, v0, X.gen);
        }
        S.status = take ? (leap0 ? (int)WS_LEAP : -1) : S.status;
        S.fresh = take ? (leap0 ? 1u : 0u) : S.fresh;
        S.zombie = take ? 0u : S.zombie;
        S.epoch = take ? X.epoch : S.epoch;
        S.gen = take ? X.gen : S.gen;
        S.alt = take ? rank : S.alt;
        S.cur_v = take ? v0 : S.cur_v;
        S.tot = take ? has_size + s0 : S.tot;
        S.base = take ? has_size : S.base;
        if (take && (uint64_t)s0 > X.max_probe) X.max_probe = s0;
        S.len = take ? 1u : S.len;
        S.off = take ? (go ? off0 : 0u) : S.off;
        S.cnt = take ? (go ? cnt0 : 0u) : S.cnt;
        S.wp0 = take ? (c != 0 ? c : 0xFFFFFFFFu) : S.wp0;
        S.wp1 = take ? c : S.wp1;
        S.wt0 = take ? X.win_t0 : S.wt0;
        S.wt1 = take ? X.win_t1 : S.wt1;
        S.po.n = take ? (go && !inr0 ? 1u : 0u) : S.po.n;
        S.po.v0 = take ? (go && !inr0 ? v0 : 0u) : S.po.v0;
        S.po.v1 = take ? 0u : S.po.v1;
        S.pb_v = take ? v0 : S.pb_v;  // entry 0 of the path waits in lane 0 of the group
        S.pb_s = take ? s0 : S.pb_s;
        S.ab = take ? ab0 : S.ab;
        S.last_pc = take ? c : S.last_pc;
    }
    slots_dominate(X, S, speculate);  // an alternative may have leapt right at its first vertex
    PROF_END(X, 3, t_setup);
    PROF_BEGIN(t_steps);
    uint64_t seen = __ballot(S.fresh != 0u && S.epoch == X.epoch);
    for (;;) {
        if (!__ballot(S.status < 0 && !S.zombie && S.epoch == X.epoch)) break;
        slots_step(L, X, S, arena_v, arena_s, cap_each, &wide, false);
        if (wide) break;
        const uint64_t now = __ballot(S.fresh != 0u && S.epoch == X.epoch);
        if (now != seen) {  // an alternative of this iteration has stopped: what does it mean for the others?
            slots_dominate(X, S, speculate);
            seen = now;
        }
    }
    PROF_END(X, 4, t_steps);
    X.overflow = __ballot(X.overflow != 0) ? 1 : 0;
    {
        int sf = X.spec_fail, pz = (int)X.x_poison;
        for (int d2 = 32; d2 >= 1; d2 >>= 1) {
            sf |= __shfl_xor(sf, d2, 64);
            pz |= __shfl_xor(pz, d2, 64);
        }
        X.spec_fail = sf;
        X.x_poison = (uint32_t)pz;
    }
    __syncthreads();  // paths written by the groups are read by all lanes afterwards
    return !wide;
}

// walkStraight for ONE alternative by a whole wave: every piece of walk state (current vertex, record range,
// length, windows, status) is wave-uniform, so it lives in scalar registers and is updated by the scalar
// unit; only the evaluation of the <= 64 successor records of a vertex is per-lane work.  `grp` selects the
// stamp array / outside set of this probe, `alt` the alternative (start vertex in L.br_*), the path goes to
// pv/ps.  Returns false if a vertex with more than 64 records was met (caller falls back to walk_straight).
__device__ __forceinline__ bool probe_wave(WalkLds &L, WalkCtx &X, uint32_t grp, uint32_t alt, bool have_meta, uint64_t has_size, uint32_t *pv,
                           uint32_t *ps, uint64_t cap, bool follow, ProbeRes *res) {
    const uint32_t lane = lane_id();
    uint64_t *pset = X.pset_o + (uint64_t)grp * ((uint64_t)X.pmask_o + 1);
    uint32_t aw0 = 0xFFFFFFFFu, aw1 = 0;
    const uint32_t sg0 = X.win_p0, sg1 = X.win_p1;
    X.win_p0 = 0xFFFFFFFFu;
    X.win_p1 = 0;
    ProbeOut po{0, 0, 0};
    X.n_probe += 1;
    const uint32_t v0 = L.br_v[alt], s0 = L.br_s[alt];
    uint32_t cur = v0, len = 1, off = 0, cnt = 0;
    uint64_t now_size = s0;
    int status = -1;
    bool wide = false;
    if (cap == 0) {
        X.overflow = 1;
        res->status = WS_END;
        res->len = 0;
        return true;
    }
    if (lane == 0) {
        pv[0] = v0;
        ps[0] = s0;
    }
    uint32_t ab;
    {
        const uint32_t e0 = v0 - X.C.in_lo - X.w_d0;
        ab = L.wab[e0 < X.w_nid ? e0 + X.w_ab : 0u];
        if (!(e0 < X.w_nid)) ab = X.G.ucnt[v0];
    }
    const uint32_t c0 = have_meta ? L.br_pc[alt] : (uint32_t)(X.G.upos[v0] >> 32);
    uint32_t last_pc = c0;
    win_add(aw0, aw1, c0);
    if (c0 != 0 && (c0 < X.C.ctg_left || c0 >= X.C.ctg_right)) {
        status = WS_LEAP;
    } else {
        win_add(X.win_p0, X.win_p1, c0);
        if (lane == 0) {
            if (in_range(X, v0)) stamp_put(L, X, grp, v0, X.gen);
            else gs_insert_single(pset, X.pmask_o, v0, X.gen);
        }
        if (!in_range(X, v0)) probe_out_add(po, v0);
        if (have_meta) {
            off = L.br_off[alt];
            cnt = L.br_cnt[alt];
        } else {
            cnt = 15u;
        }
        if (cnt == 15u) {
            off = X.G.succ_off[v0];
            cnt = X.G.succ_off[v0 + 1] - off;
        }
    }
    while (status < 0) {
        if (cnt > 64u) {
            wide = true;
            break;
        }
        if (follow) win_follow(L, X, cur, off, cnt);
        X.n_classify += 1;
        const bool can_leap = (has_size + now_size) >= X.C.split_size;
        int cls = -1;
        SuccRec rec{0, 0, 0, 0};
        if (lane < cnt) {
            rec = rec_load(L, X, off + lane);
            cls = eval_record(L, X, rec, can_leap, 2, grp, po, X.epoch, X.gen);
        }
        uint64_t m = __ballot(cls == 0);
        if (!m) m = __ballot(cls == 1);
        if (!m) m = __ballot(cls == 2);
        if (!m) m = __ballot(cls == 3);
        const uint32_t n = (uint32_t)__popcll(m);
        if (n != 1u) {  // the walk stops at this classification: leave the accepted records behind (see WalkLds)
            const uint64_t am = __ballot(cls >= 0);
            if (cls >= 0 && (uint32_t)__popcll(am & lanemask_lt()) < (uint32_t)PB_CAP) {
                const uint32_t kk = (uint32_t)__popcll(am & lanemask_lt());
                L.pb_v[grp][kk] = rec.tgt;
                L.pb_meta[grp][kk] = rec.meta;
                L.pb_pc[grp][kk] = rec.pc;
                L.pb_off[grp][kk] = rec.toff;
                L.pb_cls[grp][kk] = (uint32_t)cls;
            }
            if (lane == 0) L.pb_cnt[grp] = (uint32_t)__popcll(am);
            status = n == 0 ? WS_END : WS_BRANCH;
            break;
        }
        if (len >= cap || (uint64_t)(po.n + 1) * 2 > (uint64_t)X.pmask_o) {
            X.overflow = 1;
            status = WS_END;
            break;
        }
        const int src = __ffsll((long long)m) - 1;
        const uint32_t meta = __shfl(rec.meta, src, 64);
        const uint32_t nv = __shfl(rec.tgt, src, 64);
        const uint32_t npc = __shfl(rec.pc, src, 64);
        const uint32_t noff = __shfl(rec.toff, src, 64);
        const uint32_t ns = meta & 0xFFFFFFu;
        if (lane == 0) {
            if (in_range(X, nv)) stamp_put(L, X, grp, nv, X.gen);
            else gs_insert_single(pset, X.pmask_o, nv, X.gen);
            pv[len] = nv;
            ps[len] = ns;
        }
        if (!in_range(X, nv)) probe_out_add(po, nv);
        win_add(X.win_p0, X.win_p1, npc);
        win_add(aw0, aw1, npc);
        last_pc = npc;
        len += 1;
        now_size += ns;
        cur = nv;
        if (npc != 0 && (npc < X.C.ctg_left || npc >= X.C.ctg_right)) {
            status = WS_LEAP;
            break;
        }
        off = noff;
        cnt = meta >> 28;
        if (cnt == 15u) {
            off = X.G.succ_off[nv];
            cnt = X.G.succ_off[nv + 1] - off;
        }
    }
    X.win_p0 = sg0;
    X.win_p1 = sg1;
    res->status = status;
    res->len = len;
    res->last_v = cur;
    res->last_pc = last_pc;
    res->off = off;
    res->cnt = cnt;
    res->w0 = aw0;
    res->w1 = aw1;
    res->n_out = po.n;
    res->ab = ab;
    res->size = now_size;
    X.max_probe = now_size > X.max_probe ? now_size : X.max_probe;
    return !wide;
}

// graphTravel (PAlgorithm.tcc:172-298), one wave per job
// A job clears its own marks before it begins (TravJob::self_clear, PAG_WALK_SELFCLEAR=1; a measured variant, not the default): the
// control thread has ONE launch clear the stamps, travel epochs and hash sets of every job of a batch (25 GB at configs[1]) and
// waits for it before the first job is published, 2.7 ms at the head of every block's walks with the device otherwise idle.  With
// the jobs clearing their own the walks start at once and take LONGER (88.2 against 85.0 ms): a lone wave needs ~0.6 ms for its
// 4 MB.
__device__ __forceinline__ void wave_fill16(void *p, uint64_t bytes, uint32_t word) {  // (p 16-byte aligned, bytes a multiple of 16)
    uint4 *q = (uint4 *)p;
    const uint4 w = make_uint4(word, word, word, word);
    const uint64_t n = bytes / 16u;
    for (uint64_t i = lane_id(); i < n; i += 64u) q[i] = w;
}
__device__ __forceinline__ void job_self_clear(const TravJob &J) {
    const uint64_t PG = TRAV_PROBE_GROUPS;
    if ((J.mode & TRAV_MODE_LEAP) && J.seq_x)
        for (uint64_t i = lane_id(); i < J.seq_cap; i += 64u) J.seq_x[i] = 0ull;
    wave_fill16(J.tset, ((uint64_t)J.tmask + 1u) * 8u, 0xFFFFFFFFu);
    wave_fill16(J.pset, ((uint64_t)J.pmask + 1u) * PG * 8u, 0u);
    wave_fill16(J.stamp, PG * (uint64_t)J.stamp_stride * 4u, 0u);
    wave_fill16(J.tbits, ((uint64_t)J.stamp_stride + 4u) * 4u, 0u);
    // the stores are in the L2 before anything of the job reads or updates these arrays (L2 atomics, agent-scope loads, plain
    // loads through an L1 that holds none of these lines yet: the fence drops what it might)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void walk_job(WalkLds &L, const TravGraph &G, const TravContig &C, const TravJob &Jsrc, TravJobOut *out, uint32_t k) {
    const uint32_t lane = lane_id();
    const TravJob J = Jsrc;  // by value: the record may live in host memory
    if (J.self_clear) job_self_clear(J);
    WalkCtx X;
    X.G = G;
    X.C = C;
    X.stamp = J.stamp;
    X.stamp_stride = J.stamp_stride;
    X.tbits = J.tbits;
    X.tset_o = J.tset;
    X.tmask_o = J.tmask;
    X.pset_o = J.pset;
    X.pmask_o = J.pmask;
    X.n_out = 0;
    X.gen = 0;
    X.epoch = 0;
    X.win_g0 = X.C.gwin_lo;
    X.win_g1 = X.C.gwin_hi;
    X.win_t0 = 0xFFFFFFFFu;
    X.win_t1 = 0;
    X.overflow = 0;
    X.spec_fail = 0;
    X.max_probe = 0;
    if (J.mode & TRAV_MODE_SPEC) X.C.split_size = ~0ull;  // a piece walked ahead of its graphTravel: leaping is off
    if (J.mode & TRAV_MODE_LEAP) X.C.split_size = 0ull;   // ... inside the leaping zone: leaping is on from the first vertex
    X.force_low = (J.mode & TRAV_MODE_LEAP) ? J.win_low : 0u;
    X.x_elow = X.x_m0 = X.x_forced = 0xFFFFFFFFu;
    X.x_below = 0u;
    X.x_poison = 0u;
    X.w_d0 = X.w_nid = X.w_r0 = X.w_nrec = X.w_anchor = X.w_ab = X.n_fill = 0;
#ifdef PAG_WALK_PROF
    for (int q = 0; q < 14; ++q) {
        X.pt[q] = 0;
        X.pc[q] = 0;
    }
#endif
    X.n_classify = X.n_probe = X.n_records = 0;

    PROF_BEGIN(t_setup0);
    for (uint32_t i = lane; i < FILT_WORDS; i += 64) {
        L.ft[i] = 0;
        L.fg[i] = 0;
    }
    __syncthreads();
    if (X.C.gset) {  // members of the contig's global visited set (outside-range part), 16 probes in flight per lane
        const uint32_t cap = X.C.gmask + 1u;
        for (uint32_t b = 0; b < cap; b += 1024u) {
            uint32_t t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t i = b + (uint32_t)q * 64u + lane;
                t[q] = i < cap ? X.C.gset[i] : HS_EMPTY;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (t[q] != HS_EMPTY) filt_set(L.fg, t[q]);
        }
    }
    __syncthreads();
    PROF_END(X, 12, t_setup0);
    uint64_t seq_len = 0, now_size = k, seq_size = 0;
    const uint64_t has_size = J.has_size;
    const uint32_t start = G.newid[J.start];
    win_add(X.win_t0, X.win_t1, (uint32_t)(G.upos[start] >> 32));
    if ((J.mode & TRAV_MODE_LEAP) && J.win_low != 0u && J.win_low < X.win_t0) X.win_t0 = J.win_low;

    // stitch bookkeeping (see TravJobOut): the lowest coordinate the probes of the running iteration visited
    uint32_t it_low = 0xFFFFFFFFu, max_back = 0, max_chosen = 0, stopped = 0;
    const bool resume = (J.mode & TRAV_MODE_RESUME) != 0;
    uint64_t plen = 0;
    // the chosen path of an iteration lives in the arena; the path a RESUME job continues lives in the sequence itself
    const uint32_t *ch_v = J.arena_v, *ch_s = J.arena_s;
    if (!resume) {
        walk_straight(L, X, start, k, has_size + now_size, J.arena_v, J.arena_s, J.arena_cap, &plen);
        it_low = X.win_p0;
        const uint32_t c0 = (uint32_t)(G.upos[start] >> 32);
        if (it_low < c0 && c0 - it_low > max_back) max_back = c0 - it_low;
    } else {
        ch_v = J.seq_v;
        ch_s = J.seq_s;
        plen = J.init_len;
    }
    uint64_t ch_off = 0, ch_len = plen;  // chosen path inside the arena
    // what the probe that produced the chosen path already knows about it (fast == true): no need to read
    // the positions / offsets of its vertices back from memory
    bool fast = false;
    uint32_t f_w0 = 0, f_w1 = 0, f_nout = 0, f_last = 0, f_lpc = 0, f_off = 0, f_cnt = 0, f_grp = 0;
    bool f_list = false;  // L.pb_*[f_grp] holds the classification the chosen probe stopped at
    uint64_t f_size = 0;
    Slot S;
    S.status = WS_END;
    S.fresh = S.zombie = S.epoch = S.gen = S.alt = 0;
    S.cur_v = S.off = S.cnt = S.len = S.last_pc = S.ab = 0;
    S.wp0 = S.wp1 = S.wt0 = S.wt1 = 0;
    S.po = ProbeOut{0, 0, 0};
    S.tot = 0;
    S.base = 0;
    S.pb_v = S.pb_s = 0;
    const bool speculate = J.exact == 0;
    const uint64_t slot_cap = J.arena_cap / PROBE_GROUPS;

    // iteration log of a TRAV_MODE_LEAP job: the entry of the boundary the running iteration started at is written when the
    // iteration is over (next turn of the loop, or after the loop once the last zombie has stopped)
    uint64_t log_idx = ~0ull;
    auto flush_log = [&]() {
        uint32_t el = X.x_elow, m0 = X.x_m0;
        for (int d2 = 32; d2 >= 1; d2 >>= 1) {
            const uint32_t oe = (uint32_t)__shfl_xor((int)el, d2, 64), om = (uint32_t)__shfl_xor((int)m0, d2, 64);
            el = oe < el ? oe : el;
            m0 = om < m0 ? om : m0;
        }
        if (J.seq_x && log_idx != ~0ull && lane == 0)
            J.seq_x[log_idx] = (1ull << 63) | ((uint64_t)(el < 0x7FFFFFFFu ? el : 0x7FFFFFFFu) << 32) | (uint64_t)m0;
        X.x_elow = X.x_m0 = 0xFFFFFFFFu;
    };
    uint64_t n_main = 0;
    for (;;) {
        ++n_main;
        if (J.seq_x) flush_log();
        PROF_BEGIN(t_app);
        X.epoch = (uint32_t)n_main;  // marks appended in this iteration carry it; the probes launched after see all of them
        // append the chosen path to the sequence, mark it visited, widen the travel window
        if (seq_len + ch_len > J.seq_cap) {
            X.overflow = 1;
            break;
        }
        uint32_t last, lc, l_off, l_cnt;
        max_chosen = ch_len > max_chosen ? (uint32_t)(ch_len < 0xFFFFFFFFull ? ch_len : 0xFFFFFFFFull) : max_chosen;
        if (fast) {
            uint32_t n_outside_chk = 0;
            for (uint64_t i = lane; i < ch_len; i += 64) {
                const uint32_t v = J.arena_v[ch_off + i], st = J.arena_s[ch_off + i];
                J.seq_v[seq_len + i] = v;
                J.seq_s[seq_len + i] = st;
                if (in_range(X, v)) {
                    stamp_store(&X.tbits[v - X.C.in_lo], X.epoch);
                    const uint32_t e = v - X.C.in_lo - X.w_d0;
                    if (e < X.w_nid) L.wts[e] = X.epoch;
                } else {
                    hs64_insert(X.tset_o, X.tmask_o, v, X.epoch);
                    filt_set(L.ft, v);
                    ++n_outside_chk;
                }
            }
            (void)n_outside_chk;
            now_size += f_size;
            seq_size += f_size;
            X.n_out += f_nout;
            if (f_w1 != 0) {
                X.win_t0 = f_w0 < X.win_t0 ? f_w0 : X.win_t0;
                X.win_t1 = f_w1 > X.win_t1 ? f_w1 : X.win_t1;
            }
            seq_len += ch_len;
            if ((uint64_t)X.n_out * 2 > (uint64_t)X.tmask_o) {
                X.overflow = 1;
                break;
            }
            __syncthreads();
            last = f_last;
            lc = f_lpc;
            l_off = f_off;
            l_cnt = f_cnt;
        } else {
            uint64_t add = 0;
            uint32_t lo = 0xFFFFFFFFu, hi = 0, n_outside = 0;
            // Four entries per lane and turn, their loads issued together: the path a RESUME job takes over is the whole walk so
            // far — a quarter of a million vertices at the end of a 1 Mb contig — and one entry per turn (load, coordinate
            // gather, mark: three dependent round trips) made such a job spend 5 ms here before its first step, on the
            // critical path of its contig (every contig ends with one or two of them).
            for (uint64_t i0 = lane; i0 < ch_len; i0 += 256) {
                uint32_t v4[4], st4[4], c4[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t i = i0 + (uint64_t)q * 64u;
                    v4[q] = i < ch_len ? ch_v[ch_off + i] : 0u;
                    st4[q] = i < ch_len ? ch_s[ch_off + i] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) c4[q] = i0 + (uint64_t)q * 64u < ch_len ? (uint32_t)(G.upos[v4[q]] >> 32) : 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t i = i0 + (uint64_t)q * 64u;
                    if (i >= ch_len) break;
                    const uint32_t v = v4[q], st = st4[q], c = c4[q];
                    J.seq_v[seq_len + i] = v;  // (a RESUME job's first "chosen path" is the sequence itself: stored onto itself)
                    J.seq_s[seq_len + i] = st;
                    if (in_range(X, v)) {
                        stamp_store(&X.tbits[v - X.C.in_lo], X.epoch);
                        const uint32_t e = v - X.C.in_lo - X.w_d0;
                        if (e < X.w_nid) L.wts[e] = X.epoch;
                    } else {
                        hs64_insert(X.tset_o, X.tmask_o, v, X.epoch);
                        filt_set(L.ft, v);
                        ++n_outside;
                    }
                    add += st;
                    if (c != 0) {
                        lo = c < lo ? c : lo;
                        hi = c > hi ? c : hi;
                    }
                }
            }
            {
                uint64_t tot;
                wave_excl_sum64(add, &tot);
                now_size += tot;
                seq_size += tot;
                X.n_out += wave_sum(n_outside);
                for (int d = 32; d >= 1; d >>= 1) {
                    uint32_t ol = __shfl_xor(lo, d, 64), oh = __shfl_xor(hi, d, 64);
                    lo = ol < lo ? ol : lo;
                    hi = oh > hi ? oh : hi;
                }
                if (hi != 0) {
                    X.win_t0 = lo < X.win_t0 ? lo : X.win_t0;
                    X.win_t1 = hi > X.win_t1 ? hi : X.win_t1;
                }
            }
            seq_len += ch_len;
            if ((uint64_t)X.n_out * 2 > (uint64_t)X.tmask_o) {
                X.overflow = 1;
                break;
            }
            __threadfence_block();
            __syncthreads();
            last = J.seq_v[seq_len - 1];
            lc = (uint32_t)(G.upos[last] >> 32);
            l_off = G.succ_off[last];
            l_cnt = G.succ_off[last + 1] - l_off;
            ch_v = J.arena_v;
            ch_s = J.arena_s;
        }
        log_idx = seq_len - 1;
        if (lc != 0 && (lc < X.C.ctg_left || lc >= X.C.ctg_right)) break;
        if (J.stop_pc != 0u && lc >= J.stop_pc) {  // (lc is on the own strand here) the piece ends at this iteration boundary
            stopped = 1;
            break;
        }
        // (TRAV_MODE_UNTIL_LEAP: a resumed walk that only has to cross the point from which leaping is possible — the pieces
        // of the leaping zone take over there — ends at the first iteration boundary with hasSize + nowSize >= split size)
        if ((J.mode & TRAV_MODE_UNTIL_LEAP) && (has_size + now_size) >= X.C.split_size) {
            stopped = 1;
            break;
        }
        it_low = lc != 0u ? lc : 0xFFFFFFFFu;
        PROF_END(X, 0, t_app);
        PROF_BEGIN(t_cls);

        Step one;
        bool list_meta = false;
        uint32_t m;
        // the alternatives of this iteration: L.br_*[0 .. m) — or, beyond BR_CAP of them (a classification of more than 64
        // records: repeats), gl_v / gl_s[0 .. m): the room behind the end of the job's sequence, free until the chosen path
        // of this iteration is appended at the top of the next one
        uint32_t *const gl_v = J.seq_v + seq_len, *const gl_s = J.seq_s + seq_len;
        bool use_list = fast && f_list;
        if (use_list) {
            __syncthreads();
            use_list = L.pb_cnt[f_grp] <= (uint32_t)PB_CAP;  // (a probe that stopped with more accepted records left none behind)
        }
        if (use_list) {
            // graphTravel's classification of `last` (level 1) = the classification the chosen probe stopped at
            // (level 2): the probe's marks are exactly the vertices just appended and its coordinate window has just
            // been merged into the travel window.  The one difference: the merged window is the HULL of the two, so
            // an accepted record whose coordinate lies in the gap between them is rejected now.
            const uint32_t nc = L.pb_cnt[f_grp];
            int c = -1;
            uint32_t cv = 0, cmeta = 0, cpc = 0, coff = 0;
            if (lane < nc) {
                cv = L.pb_v[f_grp][lane];
                cmeta = L.pb_meta[f_grp][lane];
                cpc = L.pb_pc[f_grp][lane];
                coff = L.pb_off[f_grp][lane];
                const bool free_pc = (cpc == 0u) | (((cmeta >> 27) & 1u) != 0u);
                walk_note_record(X, cv, cpc, ((cmeta >> 27) & 1u) != 0u);  // (examined again, now at the top level)
                c = (free_pc || !in_win(X.win_t0, X.win_t1, cpc)) ? (int)L.pb_cls[f_grp][lane] : -1;
            }
            uint64_t mm = __ballot(c == 0);
            if (!mm) mm = __ballot(c == 1);
            if (!mm) mm = __ballot(c == 2);
            if (!mm) mm = __ballot(c == 3);
            m = (uint32_t)__popcll(mm);
            X.n_classify += 1;
            if (m == 0) break;
            __syncthreads();
            if ((mm >> lane) & 1ull) {
                const uint32_t at = (uint32_t)__popcll(mm & lanemask_lt());
                L.br_v[at] = cv;
                L.br_s[at] = cmeta & 0xFFFFFFu;
                L.br_pc[at] = cpc;
                L.br_off[at] = coff;
                L.br_cnt[at] = cmeta >> 28;
            }
            list_meta = true;
            __syncthreads();
        } else {
        const SuccRec none{0, 0, 0, 0};
        win_follow(L, X, last, l_off, l_cnt);
        m = classify(L, X, l_off, l_cnt, false, none, (has_size + now_size) >= X.C.split_size, 1, ProbeOut{0, 0, 0}, &one,
                              &list_meta, gl_v, gl_s, J.seq_cap - seq_len);
        if (m == 0) break;
        if (X.overflow) break;  // (the list of a wide classification did not fit behind the sequence: the host doubles the buffers)
        __syncthreads();
        if (m == 1) {  // the single-successor fast path of classify bypasses the LDS list
            if (lane == 0) {
                L.br_v[0] = one.v;
                L.br_s[0] = one.s;
                L.br_pc[0] = one.pc;
                L.br_off[0] = one.off;
                L.br_cnt[0] = one.cnt;
            }
            list_meta = true;
        } else {
            for (uint32_t i = lane; i < m && i < (uint32_t)BR_CAP; i += 64) {
                if (list_meta) {
                    L.br_v[i] = L.lst_v[0][i];
                    L.br_s[i] = L.lst_s[0][i];
                    L.br_pc[i] = L.lst_pc[i];
                    L.br_off[i] = L.lst_off[i];
                    L.br_cnt[i] = L.lst_cnt[i];
                } else {
                    L.br_v[i] = gl_v[i];
                    L.br_s[i] = gl_s[i];
                }
            }
        }
        __syncthreads();
        }
        const bool br_global = m > (uint32_t)BR_CAP;  // (then the list is the global one: only a wide classification gets that long)

        // probe every alternative (PAlgorithm.tcc:251-266): PROBE_GROUPS at a time side by side, each in
        // its own share of the arena; sequential full-wave probing only when a vertex has more than 64 records
        PROF_END(X, 1, t_cls);
        fast = false;
        bool multi_ok = m <= PROBE_GROUPS;  // larger fan-outs would need several arena generations: sequential
        const bool zombies = __ballot(S.status < 0) != 0ull;  // between iterations only zombies are walking
        if (m == 1 && !zombies) {  // a single alternative and no slot in use: the whole wave walks it (scalar walk state)
            const uint64_t cap_each = J.arena_cap / PROBE_GROUPS;
            ProbeRes R[PROBE_GROUPS];
            bool ok_all = true;
            X.gen += 1;
            PROF_BEGIN(t_pw);
            for (uint32_t i = 0; i < m && ok_all; ++i)
                ok_all = probe_wave(L, X, i, i, list_meta, has_size + now_size, J.arena_v + i * cap_each, J.arena_s + i * cap_each, cap_each,
                                    true, &R[i]);
            PROF_END(X, 7, t_pw);
            if (ok_all) {
                __syncthreads();
                if (X.overflow) break;
                for (uint32_t i = 0; i < m; ++i) it_low = R[i].w0 < it_low ? R[i].w0 : it_low;
                if (it_low < lc && lc - it_low > max_back) max_back = lc - it_low;
                int pick = -1;
                for (uint32_t i = 0; i < m && pick < 0; ++i)
                    if (R[i].status == WS_LEAP) pick = (int)i;
                if (pick < 0) {
                    uint32_t best_ab = 0;
                    for (uint32_t i = 0; i < m; ++i) {
                        if (R[i].status != WS_BRANCH) continue;
                        if (pick < 0 || R[i].ab > best_ab) {
                            pick = (int)i;
                            best_ab = R[i].ab;
                        }
                    }
                }
                if (pick < 0) {
                    uint32_t best_len = 0;
                    for (uint32_t i = 0; i < m; ++i)
                        if (pick < 0 || R[i].len > best_len) {
                            pick = (int)i;
                            best_len = R[i].len;
                        }
                }
                ProbeRes Q = R[0];
                for (uint32_t i = 1; i < m; ++i)
                    if ((int)i == pick) Q = R[i];
                ch_off = (uint64_t)pick * cap_each;
                ch_len = Q.len;
                fast = true;
                f_grp = (uint32_t)pick;
                f_list = Q.status == WS_END || Q.status == WS_BRANCH;
                f_w0 = Q.w0;
                f_w1 = Q.w1;
                f_nout = Q.n_out;
                f_last = Q.last_v;
                f_lpc = Q.last_pc;
                f_off = Q.off;
                f_cnt = Q.cnt;
                f_size = Q.size;
                continue;
            }
            X.gen += 1;
        }
        if (multi_ok) {
            multi_ok = probe_slots(L, X, S, m, list_meta, has_size + now_size, J.arena_v, J.arena_s, slot_cap, speculate);
            if (multi_ok) {
                if (X.overflow) break;
                {   // lowest coordinate visited by the alternatives of this iteration (zombies: up to now; what they do later
                    // can only matter through a leap, which fails the job)
                    uint32_t lw = S.epoch == X.epoch ? S.wp0 : 0xFFFFFFFFu;
                    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
                        const uint32_t o2 = (uint32_t)__shfl_xor((int)lw, d2, 64);
                        lw = o2 < lw ? o2 : lw;
                    }
                    it_low = lw < it_low ? lw : it_low;
                    if (it_low < lc && lc - it_low > max_back) max_back = lc - it_low;
                }
                PROF_BEGIN(t_choice);
                // choice (PAlgorithm.tcc:268-296) among the alternatives of this iteration that have stopped (the
                // zombies are taken not to leap): the first one that leaps; else the branching one with the most
                // abundant first vertex (first wins ties); else the longest dead end (first wins ties)
                const bool fin = S.fresh != 0u && S.epoch == X.epoch;
                const uint64_t lm = __ballot(fin && S.status == WS_LEAP) & SLOT_LEADS;
                const uint64_t bm = __ballot(fin && S.status == WS_BRANCH) & SLOT_LEADS;
                int src;
                uint32_t kq;
                if (lm) src = __ffsll((long long)lm) - 1;
                else if (bm) src = slots_best(bm, S.ab, &kq);
                else src = slots_best(__ballot(fin) & SLOT_LEADS, S.len, &kq);
                const int pick = src >> GL_SHIFT;
                ch_off = (uint64_t)pick * slot_cap;
                ch_len = __shfl(S.len, src, 64);
                fast = true;
                f_grp = (uint32_t)pick;
                {
                    const int stt = __shfl(S.status, src, 64);
                    f_list = stt == WS_END || stt == WS_BRANCH;
                }
                f_w0 = __shfl(S.wp0, src, 64);
                f_w1 = __shfl(S.wp1, src, 64);
                f_nout = __shfl(S.po.n, src, 64);
                f_last = __shfl(S.cur_v, src, 64);
                f_lpc = __shfl(S.last_pc, src, 64);
                f_off = __shfl(S.off, src, 64);
                f_cnt = __shfl(S.cnt, src, 64);
                f_size = __shfl(S.tot, src, 64) - (has_size + now_size);  // (the chosen slot was started in this iteration)
                S.fresh = 0;  // consumed
                PROF_END(X, 5, t_choice);
            } else if (S.epoch == X.epoch && (S.status < 0 || S.fresh != 0u)) {  // too wide: this iteration is probed sequentially
                S.status = WS_END;
                S.fresh = 0;
                S.zombie = 0;
            }
        }
        if (!multi_ok) {  // the sequential probes use the arrays of slot 0 and the whole arena: let the zombies finish first
            while (__ballot(S.status < 0)) {
                bool w = false;
                slots_step(L, X, S, J.arena_v, J.arena_s, slot_cap, &w, true);
            }
        }
        if (!multi_ok) {
            int first_leap = -1, best_branch = -1, best_tip = -1;
            uint32_t best_ab = 0;
            uint64_t best_tip_len = 0, leap_off = 0, leap_len = 0, br_off = 0, br_len = 0, tip_off = 0;
            uint64_t used = 0;
            for (uint32_t i = 0; i < m; ++i) {
                uint64_t l2 = 0;
                const uint32_t alt_v = br_global ? gl_v[i] : L.br_v[i], alt_s = br_global ? gl_s[i] : L.br_s[i];
                int stt = walk_straight(L, X, alt_v, alt_s, has_size + now_size, J.arena_v + used, J.arena_s + used,
                                        J.arena_cap - used, &l2);
                it_low = X.win_p0 < it_low ? X.win_p0 : it_low;
                if (it_low < lc && lc - it_low > max_back) max_back = lc - it_low;
                if (stt == WS_LEAP) {
                    if (first_leap < 0) {
                        first_leap = (int)i;
                        leap_off = used;
                        leap_len = l2;
                    }
                } else if (stt == WS_END) {
                    if (best_tip < 0 || l2 > best_tip_len) {
                        best_tip = (int)i;
                        best_tip_len = l2;
                        tip_off = used;
                    }
                } else {
                    uint32_t ab = G.ucnt[alt_v];
                    if (best_branch < 0 || ab > best_ab) {
                        best_branch = (int)i;
                        best_ab = ab;
                        br_off = used;
                        br_len = l2;
                    }
                }
                used += l2;
                if (X.overflow) break;
            }
            if (X.overflow) break;
            if (first_leap >= 0) {
                ch_off = leap_off;
                ch_len = leap_len;
            } else if (best_branch >= 0) {
                ch_off = br_off;
                ch_len = br_len;
            } else {
                ch_off = tip_off;
                ch_len = best_tip_len;
            }
        }
    }
    while (__ballot(S.status < 0)) {  // the speculation is only valid once every zombie has stopped without a leap
        bool w = false;
        slots_step(L, X, S, J.arena_v, J.arena_s, slot_cap, &w, true);
    }
    if (J.seq_x) flush_log();
    uint32_t wd_below = X.x_below, wd_forced = X.x_forced;
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
        const uint32_t ob = (uint32_t)__shfl_xor((int)wd_below, d2, 64), of = (uint32_t)__shfl_xor((int)wd_forced, d2, 64);
        wd_below = ob > wd_below ? ob : wd_below;
        wd_forced = of < wd_forced ? of : wd_forced;
    }
    {
        int sf = X.spec_fail, pz = (int)X.x_poison;
        for (int d2 = 32; d2 >= 1; d2 >>= 1) {
            sf |= __shfl_xor(sf, d2, 64);
            pz |= __shfl_xor(pz, d2, 64);
        }
        X.spec_fail = sf;
        X.x_poison = (uint32_t)pz;
    }
    uint64_t mp_all = X.max_probe;
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
        const uint64_t o2 = __shfl_xor(mp_all, d2, 64);
        mp_all = o2 > mp_all ? o2 : mp_all;
    }
    if (lane == 0) {
        TravJobOut o;
        o.seq_len = seq_len;
        o.seq_size = seq_size;
        o.overflow = X.overflow | (X.spec_fail ? 4 : 0);
        o.n_classify = __shfl(X.n_classify, 0, 64);
        o.n_probe = X.n_probe;
        o.n_records = X.n_records;
        o.n_fill = X.n_fill | ((uint64_t)X.spec_fail << 32);
        o.n_out = X.n_out;
        o.n_main = n_main;
        o.stopped = stopped;
        o.max_back = max_back;
        o.max_chosen = max_chosen;
        o.wd_below_max = wd_below;
        o.wd_forced_min = wd_forced;
        o.poison = X.x_poison;
        o.max_probe = mp_all;
#ifdef PAG_WALK_PROF
        for (int q = 0; q < 14; ++q) {
            o.prof_t[q] = X.pt[q];
            o.prof_c[q] = X.pc[q];
        }
#endif
        o.last_ctg = seq_len ? (uint32_t)(G.upos[J.seq_v[seq_len - 1]] >> 32) : 0;
        *out = o;
    }
}

// one launch = one batch of jobs, one wave each
__global__ __launch_bounds__(64) void k_walk(TravGraph G, const TravContig *__restrict__ ctgs, const TravJob *__restrict__ jobs,
                                             TravJobOut *__restrict__ outs, uint32_t n_jobs, uint32_t k) {
    __shared__ WalkLds L;
    const uint32_t jid = blockIdx.x;
    if (jid >= n_jobs) return;
    walk_job(L, G, ctgs[jobs[jid].ctg], jobs[jid], &outs[jid], k);
}

// Persistent form: the grid stays resident and takes jobs from a queue that the host keeps feeding, so that the
// contigs advance through their traversal rounds independently of each other (a round of one contig starts as
// soon as ITS previous round is done, instead of when the slowest contig of the batch is done).
// The queue lives in fine-grained host memory: q->posted (host, release) = number of valid entries of `jobs`;
// a wave claims the next index from a device counter, waits until that entry is posted, runs it, writes the
// result record, makes its device-memory writes visible (system release) and raises done[index].
// A job starts with a system-scope acquire: buffers of the job were prepared by other kernels / copies.
//
// FORWARD PROGRESS.  A wave never waits for the host longer than `idle_ticks` (100 MHz clock): a wave that has found no
// claimable job for that long leaves, and the host starts new waves when it posts work for which too few are left
// (WalkerGrid, k5_travel_host.hip).  A grid that cannot be resident as a whole — another process has filled the compute
// units' LDS with walkers of its own — therefore stalls the dispatcher only until running jobs end or idle waves leave,
// never until a host acts that may itself be queued behind the stalled dispatch.  q->started / q->exited (system scope)
// tell the host how many waves it has.
// (PAG_WALK_WAVES_PER_EU, make WALK_EU=: the register budget.  Left to itself the compiler takes 256 VGPRs + AGPRs for the walker:
// one wave per SIMD, four per compute unit whatever the LDS allows; with 2 it keeps to 256 in all and spills ~35 to scratch.)
#ifndef PAG_WALK_WAVES_PER_EU
#define PAG_WALK_WAVES_PER_EU 1
#endif
__global__ __launch_bounds__(64, PAG_WALK_WAVES_PER_EU) void k_walk_persistent(TravGraph G, const TravPosted *jobs, TravJobOut *outs, uint32_t *done,
                                                        TravQueue *q, uint32_t *next, uint32_t cap, uint32_t k,
                                                        uint64_t idle_ticks) {
    __shared__ WalkLds L;
    const uint32_t lane = lane_id();
    if (lane == 0) __hip_atomic_fetch_add(&q->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // TRAV_RINGS rings of job records, served in order: the jobs a contig's progress waits for (the walk of a seed, a
    // resumed walk) are taken before the segment jobs that only run ahead of it, and those of contigs in a later round
    // before those of the first round (see k5_travel_host.hip).  A wave claims the next job number of a
    // ring with a compare-and-swap on the ring's counter when the host has posted beyond it; slot = ring * cap + number
    // mod cap.  Single-exit scalar loop: every value that steers it is wave-uniform by construction (readfirstlane).
    bool alive = true;
    uint64_t t0 = wall_clock64();
    uint32_t naps = 1;
    // (bit 63 of idle_ticks, PAG_WALK_PRIO=0 clears it: the walker waves issue ahead of whatever else is resident on their SIMD —
    // the deliveries of finished contigs, k_gather_path, run beside the last walks of a block)
    if (idle_ticks >> 63) __builtin_amdgcn_s_setprio(3);
    idle_ticks &= ~(1ull << 63);
    while (alive) {
        // One relaxed 8-byte read of host memory per poll (posted[0] | posted[1] << 32), polls of an idle wave spaced out up
        // to ~100 us: hundreds of idle waves hammering the host link would slow the working waves down.  (An acquire here
        // would also invalidate the L2 on every poll; the acquire that matters follows below.)
        const uint64_t w = __hip_atomic_load((const uint64_t *)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t w2 = __hip_atomic_load((const uint64_t *)q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        static_assert(TRAV_RINGS == 3, "posted[] is read as two 8-byte words");
        const uint32_t posted[TRAV_RINGS] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w), (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w >> 32)),
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)w2)};
        const uint32_t bye = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(w2 >> 32));
        int ring = -1;
        uint32_t idx = 0;
        for (int r = 0; r < TRAV_RINGS && ring < 0; ++r) {
            for (;;) {
                const uint32_t cur = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&next[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if ((int32_t)(posted[r] - cur) <= 0) break;  // nothing posted beyond the counter
                uint32_t won = 0;
                if (lane == 0) {
                    uint32_t expect = cur;
                    won = __hip_atomic_compare_exchange_strong(&next[r], &expect, cur + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
                }
                if (__builtin_amdgcn_readfirstlane(won)) {
                    ring = r;
                    idx = cur;
                    break;
                }
            }
        }
        if (ring >= 0) {
            const uint32_t slot = (uint32_t)ring * cap + idx % cap;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // buffers of the job were prepared by other kernels / copies
            const uint64_t tb = wall_clock64();
            if (__hip_atomic_load(&jobs[slot].J.mode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) & TRAV_MODE_CANCELLED) {
                if (lane == 0) {  // (a job of a round that is over, k5_travel_host.hip: nobody reads its result)
                    TravJobOut z{};
                    outs[slot] = z;
                }
            } else {
                walk_job(L, G, jobs[slot].C, jobs[slot].J, &outs[slot], k);
            }
            if (lane == 0) {
                outs[slot].t_begin = tb;
                outs[slot].t_end = wall_clock64();
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(&done[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // same value from every lane
            t0 = wall_clock64();
            naps = 1;
        } else {
            if (bye != 0u || wall_clock64() - t0 > idle_ticks) {  // host done, or nothing to do for too long (see above)
                alive = false;
            } else {
                for (uint32_t z = 0; z < naps; ++z) __builtin_amdgcn_s_sleep(127);
                naps = naps < 32u ? naps * 2u : 32u;
            }
        }
    }
    if (lane == 0) __hip_atomic_fetch_add(&q->exited, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// =================================================================================================
// seeds
// =================================================================================================
// searchPANode(onlyFirst = true) (PAlgorithm.tcc:300-327): the first contig k-mer one of whose positions
// lies on this contig strand within `dev` of the k-mer's own offset; all such positions of that k-mer.
// One wave per contig.  out[0] = count, then (vertex, node) pairs.
__global__ __launch_bounds__(64) void k_seed_first(TravGraph G, const TravContig *__restrict__ ctgs, uint32_t n_ctgs_sel,
                                                   uint64_t dev, uint32_t *__restrict__ out, uint32_t out_stride) {
    const uint32_t c = blockIdx.x;
    if (c >= n_ctgs_sel) return;
    const TravContig C = ctgs[c];
    const uint32_t lane = lane_id();
    uint32_t *o = out + (uint64_t)c * out_stride;
    uint32_t found = 0;
    for (uint32_t base = 0; base < C.n_kmers && !found; base += 64) {
        uint32_t i = base + lane;
        bool hit = false;
        uint32_t node = PAG_NONE;
        if (i < C.n_kmers) {
            node = C.nodes[i];
            if (node != PAG_NONE) {
                for (uint32_t p = G.npos_off[node]; p < G.npos_off[node + 1] && !hit; ++p) {
                    uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
                    if (pc >= C.ctg_left && pc < C.ctg_right) {
                        uint64_t off = pc - C.ctg_left;
                        uint64_t d = off > i ? off - i : (uint64_t)i - off;
                        hit = d <= dev;
                    }
                }
            }
        }
        uint64_t m = __ballot(hit);
        if (m) {
            int src = __ffsll((long long)m) - 1;
            if ((int)lane == src) {
                uint32_t n = 0;
                for (uint32_t p = G.npos_off[node]; p < G.npos_off[node + 1]; ++p) {
                    uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
                    if (pc >= C.ctg_left && pc < C.ctg_right) {
                        uint64_t off = pc - C.ctg_left;
                        uint64_t d = off > i ? off - i : (uint64_t)i - off;
                        if (d <= dev && 1 + 2 * n + 1 < out_stride) {
                            o[1 + 2 * n] = p;
                            o[2 + 2 * n] = i;
                            ++n;
                        }
                    }
                }
                o[0] = n;
            }
            found = 1;
        }
    }
    if (!found && lane == 0) o[0] = 0;
}

// searchPANode2 (PAlgorithm.tcc:329-365): every (contig offset in [left, right], position) pair whose
// position lies on this strand within `dev` of `pos`, in order.  Duplicates of a vertex are removed on
// the host (first occurrence wins).  TRAV_SEED_PARTS waves per request, each scanning one part of the offset range
// (the window spans 1000 x deviation offsets on either side); part p of request r writes out[(r * PARTS + p) * stride]:
// [0] = count, then vertex ids; the host concatenates the parts in order.
__global__ __launch_bounds__(64) void k_seed_window(TravGraph G, const TravContig *__restrict__ ctgs,
                                                    const TravSeedReq *__restrict__ reqs, uint32_t n_req, uint64_t dev,
                                                    uint32_t *__restrict__ out, uint32_t out_stride) {
    const uint32_t r = blockIdx.x, part = blockIdx.y;
    if (r >= n_req) return;
    const TravSeedReq R = reqs[r];
    const TravContig C = ctgs[R.ctg];
    const uint32_t lane = lane_id();
    uint32_t *o = out + ((uint64_t)r * TRAV_SEED_PARTS + part) * out_stride;
    uint32_t n_out = 0;
    const uint64_t right_all = R.right < (uint64_t)C.n_kmers ? R.right + 1 : C.n_kmers;  // exclusive
    const uint64_t span = right_all > R.left ? right_all - R.left : 0;
    const uint64_t per = ((span + TRAV_SEED_PARTS - 1) / TRAV_SEED_PARTS + 63) & ~63ull;  // offsets per part (whole wave rows)
    const uint64_t left = R.left + (uint64_t)part * per;
    const uint64_t right = left + per < right_all ? left + per : right_all;
    for (uint64_t base = left; base < right; base += 64) {
        uint64_t i = base + lane;
        uint32_t node = i < right ? C.nodes[i] : PAG_NONE;
        uint32_t p0 = 0, p1 = 0;
        if (node != PAG_NONE) {
            p0 = G.npos_off[node];
            p1 = G.npos_off[node + 1];
        }
        // lanes emit in lane order, positions in order: serialise over the lanes that have matches
        // filterPANodes (PAlgorithm.cpp:97-105): vertices of the contig's globalUniqueTable are dropped here, on the device
        // (a per-vertex predicate: applying it before the host removes duplicates gives the same list)
        auto visited = [&](uint32_t p) -> bool {
            if (!C.gbits) return false;
            const uint32_t u = G.newid[p];
            if (u >= C.g_lo && u < C.g_hi) return ((C.gbits[(u - C.g_lo) >> 5] >> ((u - C.g_lo) & 31u)) & 1u) != 0u;
            return C.gset ? hs_has(C.gset, C.gmask, u) : false;
        };
        uint32_t cnt = 0;
        for (uint32_t p = p0; p < p1; ++p) {
            uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc >= C.ctg_left && pc < C.ctg_right) {
                uint64_t off = pc - C.ctg_left;
                uint64_t d = off > R.pos ? off - R.pos : R.pos - off;
                cnt += (d <= dev && !visited(p)) ? 1u : 0u;
            }
        }
        uint32_t tot;
        uint32_t ex = wave_excl_sum(cnt, &tot);
        uint32_t w = n_out + ex;
        for (uint32_t p = p0; p < p1 && cnt; ++p) {
            uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc >= C.ctg_left && pc < C.ctg_right) {
                uint64_t off = pc - C.ctg_left;
                uint64_t d = off > R.pos ? off - R.pos : R.pos - off;
                if (d <= dev && !visited(p)) {
                    if (1 + w < out_stride) o[1 + w] = p;
                    ++w;
                }
            }
        }
        n_out += tot;
    }
    if (lane == 0) o[0] = n_out;
}

// Checkpoints of the segment-parallel walk (k5_travel_host.hip): for every request (contig, contig offset) the most
// abundant vertex that lies ON the contig strand (its k-mer is the contig's k-mer at offset i and its contig coordinate
// is within `dev` of i, like a seed of searchPANode) for i in [left, right], and that is not in the contig's global
// visited set.  Ties: the lowest offset, then position order.  One wave per request; out = (old vertex id, contig
// coordinate, abundance) or (PAG_NONE, 0, 0).  Which vertex is picked has no influence on the results of the
// traversal, only on how soon the walk that arrives from behind meets the piece started here.
__global__ __launch_bounds__(64) void k_checkpoints(TravGraph G, const TravContig *__restrict__ ctgs, const TravSeedReq *__restrict__ reqs,
                                                    uint32_t n_req, uint64_t dev, uint32_t *__restrict__ out) {
    const uint32_t r = blockIdx.x;
    if (r >= n_req) return;
    const TravSeedReq R = reqs[r];
    const TravContig C = ctgs[R.ctg];
    const uint32_t lane = lane_id();
    const uint64_t right = R.right < (uint64_t)C.n_kmers ? R.right + 1 : C.n_kmers;  // exclusive
    uint64_t best = 0;  // abundance << 40 | (0xFFFFF - (offset - left)) << 20 | (0xFFFFF - position rank): larger is better
    uint32_t best_v = PAG_NONE, best_pc = 0;
    for (uint64_t base = R.left; base < right; base += 64) {
        const uint64_t i = base + lane;
        const uint32_t node = i < right ? C.nodes[i] : PAG_NONE;
        if (node == PAG_NONE) continue;
        const uint32_t p0 = G.npos_off[node], p1 = G.npos_off[node + 1];
        for (uint32_t p = p0; p < p1; ++p) {
            const uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc < C.ctg_left || pc >= C.ctg_right) continue;
            const uint64_t off = pc - C.ctg_left;
            const uint64_t d = off > i ? off - i : i - off;
            if (d > dev) continue;
            const uint32_t u = G.newid[p];
            if (C.gbits && u >= C.g_lo && u < C.g_hi && ((C.gbits[(u - C.g_lo) >> 5] >> ((u - C.g_lo) & 31u)) & 1u)) continue;
            const uint64_t key = ((uint64_t)G.vcnt[p] << 40) | ((uint64_t)(0xFFFFFu - (uint32_t)((i - R.left) & 0xFFFFFu)) << 20) |
                                 (uint64_t)(0xFFFFFu - ((p - p0) & 0xFFFFFu));
            if (key > best) {
                best = key;
                best_v = p;
                best_pc = pc;
            }
        }
    }
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
        const uint64_t ob = __shfl_xor(best, d2, 64);
        const uint32_t ov = (uint32_t)__shfl_xor((int)best_v, d2, 64), op = (uint32_t)__shfl_xor((int)best_pc, d2, 64);
        if (ob > best) {
            best = ob;
            best_v = ov;
            best_pc = op;
        }
    }
    if (lane == 0) {
        out[3 * r] = best_v;
        out[3 * r + 1] = best_pc;
        out[3 * r + 2] = (uint32_t)(best >> 40);
    }
}

// The new parts of the sequences of a batch of finished jobs, packed for ONE copy to the host: per job its vertices (new
// ids), its steps and the contig coordinates of its vertices (+ the two words of the iteration log of a TRAV_MODE_LEAP
// job), each `len` words, at out + off — and behind them the job's BLOCK TABLES (walk_stitch.hpp: AGG_WORDS words per 64
// entries, + AGG_XWORDS for a leap job): a wave copies 64 consecutive entries per turn and reduces them while it holds them.
__global__ void k_pack_paths(TravGraph G, const TravPackDesc *__restrict__ descs, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t j = blockIdx.y;
    if (j >= n) return;
    const TravPackDesc D = descs[j];
    uint32_t *o = out + D.off;
    const uint32_t lane = lane_id();
    const uint64_t n_arrays = D.seq_x ? 5 : 3;
    uint32_t *agg = o + n_arrays * D.len;
    uint32_t *xagg = agg + ((D.len + 63) / 64) * 5;
    // (wave-uniform loop: every lane of a wave takes part in the reductions of its block)
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < D.len; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + lane;
        const bool valid = i < D.len;
        uint32_t v = 0, st = 0, c = 0, xl = 0, xh = 0;
        if (valid) {
            v = D.seq_v[i];
            st = D.seq_s[i];
            c = (uint32_t)(G.upos[v] >> 32);
            o[i] = v;
            o[D.len + i] = st;
            o[2 * D.len + i] = c;
            if (D.seq_x) {
                const uint64_t x = D.seq_x[i];
                xl = (uint32_t)x;
                xh = (uint32_t)(x >> 32);
                o[3 * D.len + i] = xl;
                o[4 * D.len + i] = xh;
            }
        }
        const uint32_t mx = wave_max_u32(valid ? c : 0u);
        const uint32_t m0 = wave_max_u32(valid && c == 0u ? v + 1u : 0u);
        const uint32_t lo = wave_min_u32(valid ? c : 0xFFFFFFFFu);
        const uint32_t lnz = wave_min_u32(valid && c != 0u ? c : 0xFFFFFFFFu);
        const uint32_t sum = wave_sum(valid ? st : 0u);
        const uint64_t blk = i0 >> 6;
        if (lane == 0) {
            uint32_t *a = agg + blk * 5;
            a[0] = mx;
            a[1] = m0;
            a[2] = lo;
            a[3] = lnz;
            a[4] = sum;
        }
        if (D.seq_x) {
            const bool bd = valid && (xh >> 31) != 0u;
            const uint32_t elow = wave_min_u32(bd ? (xh & 0x7FFFFFFFu) : 0xFFFFFFFFu);
            const uint32_t xm0 = wave_min_u32(bd ? xl : 0xFFFFFFFFu);
            if (lane == 0) {
                xagg[blk * 2] = elow;
                xagg[blk * 2 + 1] = xm0;
            }
        }
    }
}

// contig coordinates of a path (new ids) for the host-side stitch
__global__ void k_gather_pc(TravGraph G, const uint32_t *__restrict__ seq_v, uint64_t len, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)(G.upos[seq_v[i]] >> 32);
}

// record a finished walk (new ids) in the contig's global visited structures
__global__ void k_commit(const uint32_t *__restrict__ seq_v, uint64_t len, uint32_t in_lo, uint32_t in_hi, uint32_t *gbits,
                         uint32_t *gset, uint32_t gmask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t u = seq_v[i];
        if (u >= in_lo && u < in_hi) atomicOr(&gbits[(u - in_lo) >> 5], 1u << ((u - in_lo) & 31u));
        else hs_insert(gset, gmask, u);
    }
}

// vertex attributes of a path (new ids) for the host
__global__ void k_gather_path(TravGraph G, const uint32_t *__restrict__ seq_v, const uint32_t *__restrict__ seq_s, uint64_t len,
                              pag_path_node *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = G.uold[seq_v[i]];
        uint64_t p = G.vpos[v];
        pag_path_node o;
        o.code = G.ncode[G.vnode[v]];
        o.ctg = (uint32_t)(p >> 32);
        o.ref = (uint32_t)p;
        o.cnt = G.vcnt[v];
        o.reserved = 0;
        o.step = (int32_t)seq_s[i];
        o.vid = v;
        out[i] = o;
    }
}

__global__ void k_gather_vertices(TravGraph G, const uint32_t *__restrict__ vids, uint32_t n, pag_path_node *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = vids[i];
    uint64_t p = G.vpos[v];
    pag_path_node o;
    o.code = G.ncode[G.vnode[v]];
    o.ctg = (uint32_t)(p >> 32);
    o.ref = (uint32_t)p;
    o.cnt = G.vcnt[v];
    o.reserved = 0;
    o.step = 0;
    o.vid = v;
    out[i] = o;
}

// -------------------------------------------------------------------------------------------------
// launch wrappers used by pag_travel.cpp-side orchestration in pag_api.hip
// -------------------------------------------------------------------------------------------------
static unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 16) + (n == 0); }

int trav_compact(const uint32_t *tkey, const uint64_t *tval, const uint32_t *tseg, const uint16_t *tcnt, uint64_t T,
                 const uint32_t *ekey, const uint64_t *eval, const uint32_t *eseg, uint64_t E, uint32_t k, uint64_t n_nodes,
                 uint64_t n_pos, uint64_t n_edges, TravGraph G, void *tmp, size_t tmp_bytes, hipStream_t s, const TravView *view,
                 uint64_t *counts_out, int place_bits) {
    // place_bits != 0 (and G.nperm, G.uold / newid / upos / ucnt / succ_off allocated: they are scratch here): the nodes are
    // numbered by place — place_bits = width of the wider of the two coordinate spaces
    // tmp: flags u32[max(T, E, words, nodes + 1)] | scan out u64[same] | scan out 2 u64[T] | keep u32[T] | scan tmp
    // view != null: only the vertices inside its intervals (device arrays) are taken; counts_out[3] = nodes, vertices, edges
    // of the view (n_nodes / n_pos / n_edges are then upper bounds: what the arrays of G were sized for)
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    uint64_t m = std::max(std::max(T, E), std::max(n_words, n_nodes + 1)) + 1;
    char *p = (char *)tmp;
    auto take = [&](size_t bytes) {
        char *q = p;
        p += (bytes + 255) & ~(size_t)255;
        return (void *)q;
    };
    uint32_t *flags = (uint32_t *)take(m * 4);
    uint64_t *sc1 = (uint64_t *)take(m * 8);
    uint64_t *sc2 = (uint64_t *)take(m * 8);
    uint32_t *keep = (uint32_t *)take(m * 4);
    uint64_t *totals = (uint64_t *)take(64);
    void *scan_tmp = take(std::max(scan_tmp_bytes(m), sort_tmp_bytes(n_nodes + 1)));
    uint64_t *code_tab = k <= TRAV_CODE_TABLE_MAX_K ? (uint64_t *)take((size_t)8 << (2 * k)) : nullptr;
    if ((size_t)(p - (char *)tmp) > tmp_bytes) {
        set_error("trav_compact: scratch too small");
        return PAG_EINVAL;
    }
    PAG_HIP_TRY(hipMemsetAsync(G.bitmap, 0, n_words * 8, s));
    int rc;
    const bool by_place = place_bits > 0 && G.nperm != nullptr;
    if (!by_place) G.nperm = nullptr;
    if (T) {
        const uint64_t n_tiles = (T + VC_TILE - 1) / VC_TILE;
        uint32_t *tile_first = flags, *tile_keep = keep;  // (n_tiles counters each)
        const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, 256u * 8u);
        const uint32_t *civ = view ? view->civ : nullptr, *riv = view ? view->riv : nullptr;
        const uint32_t n_civ = view ? view->n_civ : 0u, n_riv = view ? view->n_riv : 0u;
        uint64_t *ballots = sc2 + n_tiles + 16;  // (2 words per 64 slots, behind the tiles' offsets: m * 8 bytes hold both)
        k_view_mark<<<dim3(grid), dim3(VC_T), 0, s>>>(tkey, tval, tseg, T, civ, n_civ, riv, n_riv, view ? 0 : 1, ballots, tile_first, tile_keep, n_tiles);
        if ((rc = scan_u32_to_u64(tile_first, sc1, n_tiles, totals, scan_tmp, s))) return rc;
        if ((rc = scan_u32_to_u64(tile_keep, sc2, n_tiles, totals + 1, scan_tmp, s))) return rc;
        // (numbered by place: the code-ordered nodes and vertices go to arrays that are free until the coordinate order is made)
        TravGraph Gw = G;
        if (by_place) {
            Gw.ncode = G.newid;
            Gw.npos_off = G.uold;
            Gw.vpos = G.upos;
            Gw.vcnt = (uint16_t *)G.ucnt;
        }
        k_view_write<<<dim3(grid), dim3(VC_T), 0, s>>>(tkey, tval, tcnt, ballots, sc1, sc2, n_tiles, Gw);
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
    PAG_HIP_TRY(hipMemcpyAsync((by_place ? G.uold : G.npos_off) + n_nodes, &np32, 4, hipMemcpyHostToDevice, s));
    if (by_place && n_nodes) {
        const uint32_t *ncode_old = G.newid, *npos_off_old = G.uold;
        const uint64_t *vpos_old = G.upos;
        const uint16_t *vcnt_old = (const uint16_t *)G.ucnt;
        const int kb = std::min(32, place_bits + 1);
        const uint32_t shift = (uint32_t)(place_bits + 1 - kb), flag = 1u << (kb - 1);
        uint32_t *key0 = keep, *key1 = G.succ_off;
        uint64_t *val0 = sc2, *val1 = sc1;
        k_node_place_keys<<<dim3(grid_for(n_nodes)), dim3(256), 0, s>>>(npos_off_old, vpos_old, n_nodes, shift, flag, key0, val0);
        int in0 = 1;
        if ((rc = sort_pairs(key0, val0, key1, val1, n_nodes, kb, scan_tmp, &in0, s, nullptr, nullptr))) return rc;
        const uint64_t *perm = in0 ? val0 : val1;
        uint64_t *off_new = in0 ? val1 : val0;  // (the other value array is free)
        k_node_place_counts<<<dim3(grid_for(n_nodes)), dim3(256), 0, s>>>(perm, npos_off_old, n_nodes, flags, G.nperm);
        if ((rc = scan_u32_to_u64(flags, off_new, n_nodes, nullptr, scan_tmp, s))) return rc;  // (the sort is done with its scratch)
        k_node_place_move<<<dim3(grid_for(n_nodes)), dim3(256), 0, s>>>(perm, off_new, ncode_old, npos_off_old, vpos_old, vcnt_old, n_nodes, G);
        PAG_HIP_TRY(hipMemcpyAsync(G.npos_off + n_nodes, &np32, 4, hipMemcpyHostToDevice, s));
    }
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
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    uint64_t m = std::max(std::max(T, E), std::max(n_words, n_nodes + 1)) + 1;
    return 2 * ((m * 4 + 255) & ~(size_t)255) + 2 * ((m * 8 + 255) & ~(size_t)255) + ((std::max(scan_tmp_bytes(m), sort_tmp_bytes(n_nodes + 1)) + 255) & ~(size_t)255) + 1024 + 256 +
           (k <= TRAV_CODE_TABLE_MAX_K ? ((size_t)8 << (2 * k)) + 256 : 0);
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
void trav_launch_seed_first(TravGraph G, const TravContig *ctgs, uint32_t n, uint64_t dev, uint32_t *out, uint32_t stride,
                            hipStream_t s) {
    if (n) k_seed_first<<<dim3(n), dim3(64), 0, s>>>(G, ctgs, n, dev, out, stride);
}
void trav_launch_seed_window(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev,
                             uint32_t *out, uint32_t stride, hipStream_t s) {
    if (n) k_seed_window<<<dim3(n, TRAV_SEED_PARTS), dim3(64), 0, s>>>(G, ctgs, reqs, n, dev, out, stride);
}
void trav_launch_checkpoints(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev, uint32_t *out,
                             hipStream_t s) {
    if (n) k_checkpoints<<<dim3(n), dim3(64), 0, s>>>(G, ctgs, reqs, n, dev, out);
}
void trav_launch_id_bounds(TravGraph G, const uint32_t *coords, uint32_t n, uint32_t *out, hipStream_t s) {
    if (n) k_id_bounds<<<dim3((n + 63) / 64), dim3(64), 0, s>>>(G, coords, n, out);
}
void trav_launch_pack_paths(TravGraph G, const TravPackDesc *descs, uint32_t n, uint64_t max_len, uint32_t *out, hipStream_t s) {
    if (!n) return;
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((max_len + 255) / 256, 1), 64);
    k_pack_paths<<<dim3(gx, n), dim3(256), 0, s>>>(G, descs, n, out);
}
void trav_launch_gather_pc(TravGraph G, const uint32_t *seq_v, uint64_t len, uint32_t *out, hipStream_t s) {
    if (len) k_gather_pc<<<dim3(grid_for(len)), dim3(256), 0, s>>>(G, seq_v, len, out);
}
void trav_launch_walk(TravGraph G, const TravContig *ctgs, const TravJob *jobs, TravJobOut *outs, uint32_t n, uint32_t k,
                      hipStream_t s) {
    if (n) k_walk<<<dim3(n), dim3(64), 0, s>>>(G, ctgs, jobs, outs, n, k);
}
// walker waves (= 64-thread workgroups) that fit one compute unit: what the runtime's occupancy calculation says for
// the kernel as built (the LDS a wave's window takes decides), at most 4
int trav_walk_waves_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_walk_persistent, 64, 0) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = (int)((160 * 1024) / sizeof(WalkLds));
    }
    return n >= 8 ? 8 : n >= 1 ? n : 1;
}
void trav_launch_walk_persistent(TravGraph G, const TravPosted *jobs, TravJobOut *outs, uint32_t *done, TravQueue *q,
                                 uint32_t *next, uint32_t cap, uint32_t k, uint32_t n_waves, uint64_t idle_ticks, hipStream_t s) {
    k_walk_persistent<<<dim3(n_waves), dim3(64), 0, s>>>(G, jobs, outs, done, q, next, cap, k, idle_ticks);
}
void trav_launch_commit(const uint32_t *seq_v, uint64_t len, uint32_t in_lo, uint32_t in_hi, uint32_t *gbits, uint32_t *gset,
                        uint32_t gmask, hipStream_t s) {
    if (len) k_commit<<<dim3(grid_for(len)), dim3(256), 0, s>>>(seq_v, len, in_lo, in_hi, gbits, gset, gmask);
}
void trav_launch_ranges(TravGraph G, TravContig *ctgs, uint32_t n, hipStream_t s) {
    if (n) k_ranges<<<dim3((n + 63) / 64), dim3(64), 0, s>>>(G, ctgs, n);
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
    // the bit per new id is only read when the successor kernels cannot repeat the test themselves (more bands than they stage in
    // LDS, or PAG_SUCC_INC_BITS=1); `incomplete` stays the flag that the graph holds a region
    const bool bits_read = n_iv > INC_LDS_MAX || (std::getenv("PAG_SUCC_INC_BITS") && std::atoi(std::getenv("PAG_SUCC_INC_BITS")) != 0);
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
        // 13.2 with four or sixteen, tests/order_probe.sh; PAG_ORDER_SLICES=<2^n> overrides)
        // (nodes numbered by place: newid[v] / vcnt[v] are touched nearly in order — nothing to slice)
        uint32_t lg = n >= (32ull << 20) && !G.nperm ? 3u : 0u;
        if (const char *e = std::getenv("PAG_ORDER_SLICES")) {
            const uint32_t want = (uint32_t)std::max(1, std::atoi(e));
            lg = 0;
            while ((1u << (lg + 1)) <= want) ++lg;
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
// blockIdx.y = range; 16 bytes per lane and turn where the range allows (the job buffers are 256-byte aligned), bytes at its edges
__global__ void k_clear_ranges(const TravClear *__restrict__ ranges) {
    const TravClear c = ranges[blockIdx.y];
    uint8_t *p = (uint8_t *)c.p;
    const uint64_t head = (16u - ((uintptr_t)p & 15u)) & 15u, h = head < c.bytes ? head : c.bytes;
    const uint64_t n16 = (c.bytes - h) >> 4, tail = (c.bytes - h) & 15u;
    uint4 *q = (uint4 *)(p + h);
    const uint4 w = make_uint4(c.word, c.word, c.word, c.word);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) q[i] = w;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = (uint8_t)c.word;
        if (threadIdx.x < tail) p[h + (n16 << 4) + threadIdx.x] = (uint8_t)c.word;
    }
}
int trav_clear_ranges(const TravClear *ranges_dev, size_t n, hipStream_t s) {
    for (size_t at = 0; at < n; at += 32768) {
        const uint32_t m = (uint32_t)std::min<size_t>(32768, n - at);
        k_clear_ranges<<<dim3(32, m), dim3(256), 0, s>>>(ranges_dev + at);
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
// pag_successors: one wave looks one vertex up in the coordinate order — the vertices without a contig coordinate come first,
// by reference coordinate; the others by contig coordinate; equal keys in k-mer-major order (trav_order) — and hands its records
// back in the caller's terms
struct SuccOut {
    uint32_t code, step;
    uint64_t pos;
    uint32_t grade, ctg_similar;
};
__global__ void k_successors_of(TravGraph G, uint32_t code, uint64_t pos, SuccOut *__restrict__ recs, uint64_t cap, unsigned long long *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const bool zero = (pos >> 32) == 0;
    const uint64_t lo0 = zero ? 0 : G.n_zero, hi0 = zero ? G.n_zero : G.n_pos;
    const uint32_t want = zero ? (uint32_t)pos : (uint32_t)(pos >> 32);
    uint64_t lo = lo0, hi = hi0;  // first u of the stretch whose key is >= want
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t p = G.upos[mid];
        const uint32_t key = zero ? (uint32_t)p : (uint32_t)(p >> 32);
        if (key < want) lo = mid + 1;
        else hi = mid;
    }
    unsigned long long found = ~0ull;
    for (uint64_t base = lo; base < hi0; base += 64) {
        const uint64_t u = base + lane;
        bool same_key = false, hit = false;
        if (u < hi0) {
            const uint64_t p = G.upos[u];
            same_key = (zero ? (uint32_t)p : (uint32_t)(p >> 32)) == want;
            hit = p == pos && G.ncode[G.vnode[G.uold[u]]] == code;
        }
        const unsigned long long hits = __ballot(hit);
        if (hits) {
            found = base + (unsigned long long)__builtin_ctzll(hits);
            break;
        }
        if (__ballot(same_key) != ~0ull) break;  // (the run of this key ends inside these 64)
    }
    if (found == ~0ull) {
        if (lane == 0) out[0] = ~0ull;
        return;
    }
    const uint32_t a = G.succ_off[found], b = G.succ_off[found + 1];
    bool marker = false;
    for (uint32_t i = a + lane; i < b; i += 64) {
        const SuccRec r = G.succ[i];
        const uint32_t grade = (r.meta >> 24) & 7u;
        if (grade >= GRADE_POISON_IF_LEAP) {
            marker = true;
        } else if ((uint64_t)(i - a) < cap) {
            SuccOut o;
            o.code = G.ncode[G.vnode[G.uold[r.tgt]]];
            o.step = r.meta & 0xFFFFFFu;
            o.pos = G.upos[r.tgt];
            o.grade = grade;
            o.ctg_similar = (r.meta >> 27) & 1u;
            recs[i - a] = o;
        }
    }
    const bool any_marker = __ballot(marker) != 0ull;
    if (lane == 0) out[0] = any_marker ? ~1ull : (unsigned long long)(b - a);
}
int trav_successors_of(TravGraph G, uint32_t code, uint64_t pos, void *recs, uint64_t cap, unsigned long long *out, hipStream_t s) {
    static_assert(sizeof(SuccOut) == sizeof(pag_succ), "pag_succ layout");
    k_successors_of<<<dim3(1), dim3(64), 0, s>>>(G, code, pos, (SuccOut *)recs, cap, out);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
int trav_succ_count(TravGraph G, uint32_t dev, double err, uint32_t *cnt, uint64_t *scan_out, void *scan_tmp, uint64_t *total_dev,
                    const uint64_t *stage_off, SuccRec *stage, uint64_t *amask, uint32_t *heavy_list, unsigned long long *heavy_n,
                    uint32_t heavy_limit, hipStream_t s) {
    if (std::getenv("PAG_SUCC_INC_BITS") && std::atoi(std::getenv("PAG_SUCC_INC_BITS")) != 0) G.inc_iv = nullptr;  // (the bit per new id gathered, as until round 5)
    const uint64_t n = G.n_pos;
    if (!n) return PAG_OK;
    if (heavy_limit == 0 || !heavy_n) heavy_list = nullptr;
    if (heavy_list) PAG_HIP_TRY(hipMemsetAsync(heavy_n, 0, 8, s));
    if (stage) k_succ<2><<<dim3(grid_for(n)), dim3(256), 0, s>>>(G, dev, err, cnt, stage_off, stage, nullptr, nullptr, nullptr, 0u);
    else {
        k_succ<0><<<dim3(grid_for(n)), dim3(256), 0, s>>>(G, dev, err, cnt, nullptr, nullptr, amask, heavy_list, heavy_n, heavy_limit);
        if (heavy_list) k_succ_heavy<0><<<dim3(4096), dim3(256), 0, s>>>(G, dev, err, cnt, amask, heavy_list, heavy_n);
    }
    PAG_HIP_TRY(hipMemsetAsync(cnt + n, 0, 4, s));
    int rc;
    if ((rc = scan_u32_to_u64(cnt, scan_out, n + 1, total_dev, scan_tmp, s))) return rc;
    k_narrow<<<dim3(grid_for(n + 1)), dim3(256), 0, s>>>(scan_out, n + 1, G.succ_off);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
// the fused way (k_succ_fused): counts, dense staging, offsets; then the scan of the counts into succ_off.  *cursor_dev ends at the
// number of staged records (beyond `cap`: the staging array was too small, nothing usable was staged)
int trav_succ_fused(TravGraph G, uint32_t dev, double err, uint32_t *cnt, uint64_t *scan_out, void *scan_tmp, uint64_t *total_dev, uint64_t *stage_off,
                    SuccRec *stage, uint64_t cap, unsigned long long *cursor_dev, uint32_t *heavy_list, unsigned long long *heavy_n, uint32_t heavy_limit,
                    hipStream_t s) {
    if (std::getenv("PAG_SUCC_INC_BITS") && std::atoi(std::getenv("PAG_SUCC_INC_BITS")) != 0) G.inc_iv = nullptr;  // (the bit per new id gathered, as until round 5)
    const uint64_t n = G.n_pos;
    if (!n) return PAG_OK;
    if (heavy_limit == 0 || !heavy_n) heavy_list = nullptr;
    PAG_HIP_TRY(hipMemsetAsync(cursor_dev, 0, 8, s));
    if (heavy_n) PAG_HIP_TRY(hipMemsetAsync(heavy_n, 0, 8, s));
    if (heavy_list && heavy_limit <= 64u)
        k_succ_fused<false><<<dim3(grid_for(n)), dim3(256), 0, s>>>(G, dev, err, cnt, stage_off, stage, cap, cursor_dev, heavy_list, heavy_n, heavy_limit);
    else
        k_succ_fused<true><<<dim3(grid_for(n)), dim3(256), 0, s>>>(G, dev, err, cnt, stage_off, stage, cap, cursor_dev, heavy_list, heavy_n, heavy_limit);
    if (heavy_list) k_succ_heavy_fused<<<dim3(4096), dim3(256), 0, s>>>(G, dev, err, cnt, stage_off, stage, cap, cursor_dev, heavy_list, heavy_n);
    PAG_HIP_TRY(hipMemsetAsync(cnt + n, 0, 4, s));
    int rc;
    if ((rc = scan_u32_to_u64(cnt, scan_out, n + 1, total_dev, scan_tmp, s))) return rc;
    k_narrow<<<dim3(grid_for(n + 1)), dim3(256), 0, s>>>(scan_out, n + 1, G.succ_off);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
int trav_succ_bound(TravGraph G, uint32_t *ub, uint64_t *scan_out, void *scan_tmp, uint64_t *total_dev, hipStream_t s) {
    const uint64_t n = G.n_pos;
    if (!n) return PAG_OK;
    k_succ_bound<<<dim3(grid_for(n)), dim3(256), 0, s>>>(G, ub);
    PAG_HIP_TRY(hipMemsetAsync(ub + n, 0, 4, s));
    int rc;
    if ((rc = scan_u32_to_u64(ub, scan_out, n + 1, total_dev, scan_tmp, s))) return rc;
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
int trav_succ_fill(TravGraph G, uint32_t dev, double err, uint64_t n_rec, const uint64_t *stage_off, const SuccRec *stage,
                   uint64_t *amask, uint32_t *heavy_list, unsigned long long *heavy_n, uint32_t heavy_limit, hipStream_t s) {
    if (std::getenv("PAG_SUCC_INC_BITS") && std::atoi(std::getenv("PAG_SUCC_INC_BITS")) != 0) G.inc_iv = nullptr;  // (the bit per new id gathered, as until round 5)
    if (!G.n_pos) return PAG_OK;
    if (heavy_limit == 0 || !heavy_n) heavy_list = nullptr;
    if (stage) {
        k_succ_place<<<dim3(grid_for(G.n_pos)), dim3(256), 0, s>>>(G, stage_off, stage);
    } else {
        // (PAG_SUCC_DEFER=0: the filling pass grades its records itself, as until round 5)
        const bool defer = !(std::getenv("PAG_SUCC_DEFER") && std::atoi(std::getenv("PAG_SUCC_DEFER")) == 0);
        const bool mask_only = heavy_list && heavy_limit <= 64u && amask;
        if (mask_only && defer)
            k_succ<4><<<dim3(grid_for(G.n_pos)), dim3(256), 0, s>>>(G, dev, err, nullptr, nullptr, nullptr, amask, heavy_list, heavy_n, heavy_limit);
        else if (mask_only)
            k_succ<3><<<dim3(grid_for(G.n_pos)), dim3(256), 0, s>>>(G, dev, err, nullptr, nullptr, nullptr, amask, heavy_list, heavy_n, heavy_limit);
        else
            k_succ<1><<<dim3(grid_for(G.n_pos)), dim3(256), 0, s>>>(G, dev, err, nullptr, nullptr, nullptr, amask, heavy_list, heavy_n, heavy_limit);
        if (heavy_list) k_succ_heavy<1><<<dim3(4096), dim3(256), 0, s>>>(G, dev, err, nullptr, amask, heavy_list, heavy_n);
        if (n_rec && mask_only && defer) k_succ_link<true><<<dim3(grid_for(n_rec)), dim3(256), 0, s>>>(G, n_rec, dev, err);
        else if (n_rec) k_succ_link<false><<<dim3(grid_for(n_rec)), dim3(256), 0, s>>>(G, n_rec, dev, err);
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
void trav_launch_gather_path(TravGraph G, const uint32_t *seq_v, const uint32_t *seq_s, uint64_t len, pag_path_node *out,
                             hipStream_t s, unsigned max_blocks) {
    const unsigned grid = max_blocks ? std::min(grid_for(len), max_blocks) : grid_for(len);
    if (len) k_gather_path<<<dim3(grid), dim3(256), 0, s>>>(G, seq_v, seq_s, len, out);
}
// blockIdx.y = the part; the blocks of a row stride over its entries
__global__ void __launch_bounds__(256) k_concat_parts(const TravConcatPart *__restrict__ parts, uint32_t *__restrict__ out_v,
                                                      uint32_t *__restrict__ out_s, uint32_t first_step) {
    const TravConcatPart P = parts[blockIdx.y];
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < P.n; x += (uint64_t)gridDim.x * blockDim.x) {
        out_v[P.start + x] = P.v[x];
        out_s[P.start + x] = (P.start + x == 0) ? first_step : P.s[x];
    }
}
void trav_launch_concat_parts(const TravConcatPart *parts, uint32_t n_parts, uint32_t *out_v, uint32_t *out_s, uint32_t first_step,
                              hipStream_t s) {
    for (uint32_t at = 0; at < n_parts; at += 32768) {
        const uint32_t n = std::min<uint32_t>(32768, n_parts - at);
        k_concat_parts<<<dim3(16, n), dim3(256), 0, s>>>(parts + at, out_v, out_s, first_step);
    }
}
void trav_launch_gather_vertices(TravGraph G, const uint32_t *vids, uint32_t n, pag_path_node *out, hipStream_t s) {
    if (n) k_gather_vertices<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(G, vids, n, out);
}

// test hook: the device match predicates on caller-supplied rows (tests/test_gpu_predicates.py feeds the reference's
// truth table tests/golden/func_predicate.txt.gz and a dense sweep around the 0.15 ratio boundary)
__global__ void k_debug_predicates(const uint32_t *__restrict__ rows, uint64_t n, double err, uint8_t *__restrict__ grade,
                                   uint8_t *__restrict__ edge_sim) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + 6 * i;  // a_ctg a_ref b_ctg b_ref dist dev
    uint32_t es = 0;
    grade[i] = (uint8_t)d_check_position(r[0], r[1], r[2], r[3], r[4], r[5], err, &es);
    edge_sim[i] = (uint8_t)es;
}
// ... the way the successor kernels evaluate them: ratio tests through the LDS table (d_ratio_entry) where the step has an entry
__global__ void k_debug_predicates_tab(const uint32_t *__restrict__ rows, uint64_t n, double err, uint8_t *__restrict__ grade,
                                       uint8_t *__restrict__ edge_sim, unsigned long long *__restrict__ n_tab) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    d_ratio_table_fill(ratio_tab, err);
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + 6 * i;
    uint32_t es = 0;
    const uint32_t entry = r[4] < RATIO_TAB_N ? ratio_tab[r[4]] : RATIO_TAB_NONE;
    if (entry != RATIO_TAB_NONE) atomicAdd(n_tab, 1ull);
    grade[i] = (uint8_t)d_check_position_any(r[0], r[1], r[2], r[3], r[4], r[5], err, entry, &es);
    edge_sim[i] = (uint8_t)es;
}

}  // namespace pagdev

static int debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device, uint64_t *n_through_table);
extern "C" int pag_debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device) {
    return debug_predicates(rows, n, err, grade, edge_sim, device, nullptr);
}
// the same rows through the ratio table of the successor kernels; *n_through_table: how many rows had a table entry
extern "C" int pag_debug_predicates_tab(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device,
                                        uint64_t *n_through_table) {
    if (!n_through_table) return PAG_EINVAL;
    *n_through_table = 0;
    return debug_predicates(rows, n, err, grade, edge_sim, device, n_through_table);
}
static int debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device, uint64_t *n_through_table) {
    using namespace pagdev;
    if (!rows || !grade || !edge_sim) return PAG_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return PAG_ENODEV;
    uint32_t *d_rows = nullptr;
    uint8_t *d_out = nullptr;
    if (n == 0) return PAG_OK;
    PAG_HIP_TRY(hipMalloc((void **)&d_rows, n * 24));
    if (hipMalloc((void **)&d_out, 2 * n + 16) != hipSuccess) {
        hipFree(d_rows);
        return PAG_ENOMEM;
    }
    unsigned long long *d_cnt = (unsigned long long *)(d_out + ((2 * n + 7) & ~(uint64_t)7));
    hipError_t e = hipMemcpy(d_rows, rows, n * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, 8);
    if (e == hipSuccess) {
        if (n_through_table) k_debug_predicates_tab<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d_rows, n, err, d_out, d_out + n, d_cnt);
        else k_debug_predicates<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d_rows, n, err, d_out, d_out + n);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && n_through_table) e = hipMemcpy(n_through_table, d_cnt, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(grade, d_out, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(edge_sim, d_out + n, n, hipMemcpyDeviceToHost);
    hipFree(d_rows);
    hipFree(d_out);
    if (e != hipSuccess) {
        set_error("pag_debug_predicates: %s", hipGetErrorString(e));
        return PAG_EFAULT;
    }
    return PAG_OK;
}
