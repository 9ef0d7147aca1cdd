// Device helpers shared by the traversal's translation units (k5_view.hip: the traversal graph; k5_succ.hip: the successor
// records; k5_travel.hip: the walker; k5_walk_aux.hip: seeds, checkpoints, path delivery): the CSR's code -> node lookup and edge
// layout, the match predicates of the epsilon-join (f64 exactly as the reference), the walker's visited sets, the test for
// vertices whose successors a regional graph does not hold.
#pragma once
#include <algorithm>

#include "pag_device.hpp"
#include "pag_travel.hpp"

namespace pagdev {

static inline unsigned grid_for(uint64_t n) { return (unsigned)std::min<uint64_t>((n + 255) / 256, 256 * 16) + (n == 0); }

__device__ __forceinline__ uint32_t node_of_code(const TravGraph &G, uint32_t code) {
    uint64_t w = G.bitmap[code >> 6];
    uint32_t b = code & 63u;
    if (!((w >> b) & 1ull)) return PAG_NONE;
    return G.rank[code >> 6] + (uint32_t)__popcll(w & ((1ull << b) - 1ull));
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
    // (the three cases of checkPosition as selects: written as early returns they became three exec-mask branches per candidate,
    // in a kernel that is bound by the instructions it issues)
    const bool A = ac != 0, B = bc != 0, C = ar != 0, D = br != 0;
    const int g1 = s2 ? (B ? G_EXCELLENT : (A ? G_SKIP : G_GOOD)) : G_OOPS;
    const int g2 = s1 ? (D ? G_EXCELLENT : G_GOOD) : G_OOPS;
    const int g3 = (s1 & s2) ? G_AMAZING : (s1 ? G_EXCELLENT : (s2 ? G_SKIP : G_OOPS));
    return !(A & B) ? g1 : (!(C & D) ? g2 : g3);
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

}  // namespace pagdev
