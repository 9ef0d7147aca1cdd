// k5_travel.hip — K5: the walker of the epsilon-join traversal.
//
// What runs where (reference PAGraph/src/tools/graph/):
//   device  PAlgorithm::classifySuccessors / walkStraight / graphTravel        PAlgorithm.tcc:35-298          (this file)
//           PABruijnGraph::searchSuccessors + checkPosition + isEdgeSimilar   PABruijnGraph.cpp:143-197, 385-400  (k5_succ.hip)
//           PAlgorithm::searchPANode / searchPANode2 (seed scans)             PAlgorithm.tcc:300-365          (k5_walk_aux.hip)
//           PABruijnGraph::findAll (contig k-mers -> graph nodes)              PABruijnGraph.cpp:339-353       (k5_view.hip)
//   host    the outer loop of PAlgorithm::travelSequence (PAlgorithm.cpp:144-426): per round pick the
//           longest / leaping seed walk, appendSeq, repeat detection, re-seeding incl. the unstable
//           std::sort by edit distance (same libstdc++ => same tie order), filterSequence, "Pump it"   (k5_travel_host.hip)
//
// One wavefront (= one 64-thread workgroup) owns one walk job (a graphTravel, or a piece of one).  A walk is a chain of
// dependent steps, so the kernel is latency-bound by design; the lanes share the work inside a step: eight probe slots of
// eight lanes walk the alternatives of a branch side by side, the successor records of consecutive strand vertices wait in
// an LDS window, ordered compaction goes by ballot.  Visited sets are direct-mapped marks over the strand's id range and
// open-addressing hash tables in HBM outside it (trav_device.hpp); the per-probe set of walkStraight uses generation tags
// so it never needs clearing.

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
    // (both passes evaluate the records against the same marks, window and epoch — nothing is written between them; a list that
    // came out another length than it was counted would be read short or stale: the job is reported instead and walked again)
    if (base != n) X.overflow = 1;
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
__device__ __forceinline__ void walk_job(WalkLds &L, const TravGraph &G, const TravContig &C, const TravJob &Jsrc, TravJobOut *out, uint32_t k) {
    const uint32_t lane = lane_id();
    TravJob Jg = Jsrc;  // by value: the record may live in host memory
    TravContig Cg = C;
    // The job's buffers are device memory; as pointers READ FROM MEMORY they are "generic" to the compiler, which then uses FLAT
    // instructions for them: those count against the LDS counter as well, so every wait for an LDS read (the window: 600 sites)
    // also waited for whatever hash-set probe or write-through store of the job was in flight.  Saying what they are (as_global,
    // pag_device.hpp) makes them global loads / stores (322 flat instructions in the kernel before).
    Jg.seq_v = as_global(Jg.seq_v), Jg.seq_s = as_global(Jg.seq_s), Jg.arena_v = as_global(Jg.arena_v), Jg.arena_s = as_global(Jg.arena_s);
    Jg.stamp = as_global(Jg.stamp), Jg.tbits = as_global(Jg.tbits), Jg.tset = as_global(Jg.tset), Jg.pset = as_global(Jg.pset), Jg.seq_x = as_global(Jg.seq_x);
    Cg.nodes = as_global(Cg.nodes), Cg.starts = as_global(Cg.starts), Cg.sizes = as_global(Cg.sizes), Cg.gbits = as_global(Cg.gbits), Cg.gset = as_global(Cg.gset);
    const TravJob J = Jg;
    WalkCtx X;
    X.G = G;
    X.C = Cg;
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

}  // namespace pagdev
