// walk_stitch.hpp — the host bookkeeping of a graphTravel that was walked in PIECES (k5_travel_host.hip, "PIECES"): the
// validated path of a chain as a list of parts, and the conditions under which the rest of a finished segment's path is
// adopted by a chain (try_merge outside the leaping zone, try_merge_leap inside it).  No device code, no HIP: plain data in,
// a decision out — unit-tested on recorded job outputs (tests/test_walk_stitch.py through tests/harness/stitch_test.cpp).
//
// Reference semantics these conditions protect: PAlgorithm::graphTravel / walkStraight / classifySuccessors
// (PAGraph/src/tools/graph/PAlgorithm.tcc:35-298) — an adoption must leave the path vertex for vertex what the un-cut walk
// produces.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pagdev {
namespace stitch {

struct Piece {  // host copy of a path: vertices (new ids), steps, contig coordinates
    std::vector<uint32_t> v, s, pc;
};
// Block tables of a fetched path: the pack kernel (k_pack_paths) reduces every 64 consecutive entries of a job's path while
// it copies them, so that the conditions below — which ask for maxima / minima / sums over stretches of thousands of
// entries — read a few dozen table rows plus the entries of the two blocks at the ends of a stretch instead of every entry
// (the entries are in pinned memory the device has just written: reading them is DRAM latency on the one thread every
// chain waits for).  build_block_aggs / build_block_xaggs are the same reductions on the host: the definition, the unit
// tests' input, and what PAG_DEBUG_CHECK_AGGS=1 holds the device's tables against.
constexpr size_t AGG_BLOCK = 64;
constexpr size_t AGG_WORDS = 5;   // per block: highest coordinate, highest id + 1 of a coordinate-free vertex, lowest coordinate,
                                  // lowest non-zero coordinate (0xFFFFFFFF: none), sum of the steps
constexpr size_t AGG_XWORDS = 2;  // TRAV_MODE_LEAP, over the block's iteration boundaries: lowest contig-following coordinate,
                                  // lowest coordinate-free id examined (0xFFFFFFFF: none)
inline size_t agg_blocks(size_t n) { return (n + AGG_BLOCK - 1) / AGG_BLOCK; }
inline void build_block_aggs(const uint32_t *v, const uint32_t *s, const uint32_t *pc, size_t n, uint32_t *out) {
    for (size_t b = 0; b < agg_blocks(n); ++b) {
        uint32_t mx = 0, m0 = 0, lo = 0xFFFFFFFFu, lnz = 0xFFFFFFFFu, sum = 0;
        for (size_t x = b * AGG_BLOCK; x < std::min(n, (b + 1) * AGG_BLOCK); ++x) {
            const uint32_t c = pc[x];
            mx = std::max(mx, c);
            lo = std::min(lo, c);
            if (c == 0) m0 = std::max(m0, v[x] + 1u);
            else lnz = std::min(lnz, c);
            sum += s[x];
        }
        uint32_t *o = out + b * AGG_WORDS;
        o[0] = mx, o[1] = m0, o[2] = lo, o[3] = lnz, o[4] = sum;
    }
}
inline void build_block_xaggs(const uint32_t *xl, const uint32_t *xh, size_t n, uint32_t *out) {
    for (size_t b = 0; b < agg_blocks(n); ++b) {
        uint32_t elow = 0xFFFFFFFFu, m0 = 0xFFFFFFFFu;
        for (size_t x = b * AGG_BLOCK; x < std::min(n, (b + 1) * AGG_BLOCK); ++x)
            if (xh[x] >> 31) {
                elow = std::min(elow, xh[x] & 0x7FFFFFFFu);
                m0 = std::min(m0, xl[x]);
            }
        out[b * AGG_XWORDS] = elow;
        out[b * AGG_XWORDS + 1] = m0;
    }
}

struct View {  // a finished job's path where the fetch put it (pinned memory kept for the whole call)
    const uint32_t *v = nullptr, *s = nullptr, *pc = nullptr;
    const uint32_t *xl = nullptr, *xh = nullptr;  // TRAV_MODE_LEAP: iteration log, low / high words
    size_t n = 0;
    const uint32_t *agg = nullptr, *xagg = nullptr;  // block tables of the arrays above (null: none, every entry is read)
    const uint32_t *dv = nullptr, *ds = nullptr;     // where v / s lie on the DEVICE (the job's sequence buffers; null: unknown)
};
struct Seg {  // one segment job of a round
    uint32_t x = 0;      // checkpoint coordinate
    uint32_t stop = 0;   // its stop coordinate
    uint32_t vid = 0;    // start vertex (old id)
    uint32_t win_lo = 0, win_hi = 0;  // new-id range its job keeps direct-mapped marks for (around the segment)
    bool done = false, usable = false, stopped = false;
    View P;  // (everything the splices need of it is computed from the arrays when a chain arrives: a few thousand entries)
    uint32_t max_back = 0, max_chosen = 0;
    uint64_t max_probe = 0;
    // a segment inside the leaping zone (TRAV_MODE_LEAP), see try_merge_leap:
    bool leap = false;
    uint32_t wd_below_max = 0, wd_forced_min = 0xFFFFFFFFu, win_low = 0;
    // the round of the contig whose state the job was posted under.  A segment KEPT from an earlier round (the contig's walk
    // dead-ended, was committed, and the next round re-seeds behind it) was walked without the global marks and the global
    // coordinate window of the rounds since: the adoption conditions take those into account (MergeCtx::g_*)
    uint32_t round = 0;
};
struct Chain {  // one graphTravel: (contig, seed) of the running round
    // The validated path so far, T, is a list of parts that stay where the fetches put them (pinned memory kept for the
    // whole call): a job's new vertices, an adopted stretch of a segment's path.  Nothing is copied when T grows; the
    // flat arrays a resumed walk or the splice needs are put together when they are needed.  Per part: where it starts in
    // T and, over everything BEFORE it, the highest coordinate and the highest id (+ 1) of a vertex without a coordinate.
    struct Part {
        const uint32_t *v, *s, *pc;
        size_t n, start;
        uint32_t mx, m0;
        const uint32_t *agg;  // block table of the fetched path the part is a stretch of (null: none) ...
        size_t org;           // ... in which the part begins at entry `org`
        const uint32_t *dv = nullptr, *ds = nullptr;  // the part's vertices / steps on the DEVICE (null: unknown)
    };
    std::vector<Part> parts;
    size_t len = 0;  // vertices of T
    uint32_t mx_all = 0, m0_all = 0;  // ... over all of T
    uint32_t low_nz = 0xFFFFFFFFu;   // lowest non-zero coordinate of T
    uint64_t size = 0;  // sum of its steps
    bool final = false;
    int waiting_seg = -1;  // the segment whose job this chain waits for
    int next_seg = 0;    // the next segment that cannot leap (index into the round's segments) ...
    int next_leap = 0;   // ... and the next one of the leaping zone the chain has not been through yet
    uint32_t grow = 1;
    bool exact = false;
    int job = -1;          // outstanding job number, -1: none
    // the job that is outstanding (to repeat it with larger buffers / without speculation)
    uint32_t job_mode = 0, job_stop = 0;
};
struct RoundState {
    uint32_t round = 0;
    bool active = false;
    std::vector<Seg> segs;   // the segments that cannot leap, by checkpoint coordinate, then those of the leaping zone, by coordinate
    size_t n_spec = 0;       // how many of the first kind (the two kinds overlap around the coordinate where leaping begins)
    std::vector<Chain> chains;
    uint32_t zone_end = 0;  // coordinate from which the walk is no longer cut (0: no segments this round)
    uint32_t live_jobs = 0;
    uint64_t has_size = 0;
    bool slot_bufs = false;  // some job of the round has its buffers in per-contig slots (not in the walk arena)
    uint32_t seg_epoch = 0;  // counts the times the segment list was given up: jobs of an earlier list are orphans
    bool kept = false;       // segs is the list of an earlier round, kept for this one (Seg::round tells which)
};

// T grows by a job's new vertices or by an adopted stretch of a segment
struct PartAgg {  // over the vertices of a part: highest coordinate, highest id + 1 of a coordinate-free vertex, lowest
                  // non-zero coordinate, sum of the steps
    uint32_t mx = 0, m0 = 0, lo = 0xFFFFFFFFu;
    uint64_t sz = 0;
    void add(uint32_t v, uint32_t st, uint32_t c) {
        mx = std::max(mx, c);
        if (c == 0) m0 = std::max(m0, v + 1u);  // (id + 1: 0 stands for "none")
        else lo = std::min(lo, c);
        sz += st;
    }
};
// the reductions over entries [i, j) of a fetched path (v, s, pc: its arrays from entry 0; agg: its block table or null)
struct RangeAgg {
    uint32_t mx = 0, m0 = 0, lo_all = 0xFFFFFFFFu, lo_nz = 0xFFFFFFFFu;
    uint64_t sz = 0;
    void entries(const uint32_t *v, const uint32_t *s, const uint32_t *pc, size_t i, size_t j) {
        // (branch-free: the compiler vectorises it)
        uint32_t a = 0, b = 0, c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu;
        uint64_t sum = 0;
        for (size_t x = i; x < j; ++x) {
            const uint32_t c = pc[x];
            a = std::max(a, c);
            b = std::max(b, c == 0u ? v[x] + 1u : 0u);
            c0 = std::min(c0, c);
            c1 = std::min(c1, c == 0u ? 0xFFFFFFFFu : c);
            sum += s[x];
        }
        mx = std::max(mx, a);
        m0 = std::max(m0, b);
        lo_all = std::min(lo_all, c0);
        lo_nz = std::min(lo_nz, c1);
        sz += sum;
    }
};
inline RangeAgg range_agg(const uint32_t *v, const uint32_t *s, const uint32_t *pc, const uint32_t *agg, size_t i, size_t j) {
    RangeAgg r;
    if (i >= j) return r;
    const size_t bi = (i + AGG_BLOCK - 1) / AGG_BLOCK, bj = j / AGG_BLOCK;  // whole blocks [bi, bj)
    if (!agg || bi >= bj) {
        r.entries(v, s, pc, i, j);
        return r;
    }
    r.entries(v, s, pc, i, bi * AGG_BLOCK);
    for (size_t b = bi; b < bj; ++b) {
        const uint32_t *o = agg + b * AGG_WORDS;
        r.mx = std::max(r.mx, o[0]);
        r.m0 = std::max(r.m0, o[1]);
        r.lo_all = std::min(r.lo_all, o[2]);
        r.lo_nz = std::min(r.lo_nz, o[3]);
        r.sz += o[4];
    }
    r.entries(v, s, pc, bj * AGG_BLOCK, j);
    return r;
}
// lowest contig-following coordinate / coordinate-free id the iterations that start at a boundary in [i, j) examined
inline void range_xagg(const uint32_t *xl, const uint32_t *xh, const uint32_t *xagg, size_t i, size_t j, uint32_t *elow_out, uint32_t *m0_out) {
    uint32_t elow = 0xFFFFFFFFu, m0 = 0xFFFFFFFFu;
    auto entries = [&](size_t a, size_t b) {
        for (size_t x = a; x < b; ++x) {
            const uint32_t h = xh[x];
            const bool bd = (h >> 31) != 0u;
            elow = std::min(elow, bd ? (h & 0x7FFFFFFFu) : 0xFFFFFFFFu);
            m0 = std::min(m0, bd ? xl[x] : 0xFFFFFFFFu);
        }
    };
    const size_t bi = (i + AGG_BLOCK - 1) / AGG_BLOCK, bj = j / AGG_BLOCK;
    if (!xagg || bi >= bj) {
        if (i < j) entries(i, j);
    } else {
        entries(i, bi * AGG_BLOCK);
        for (size_t b = bi; b < bj; ++b) {
            elow = std::min(elow, xagg[b * AGG_XWORDS]);
            m0 = std::min(m0, xagg[b * AGG_XWORDS + 1]);
        }
        entries(bj * AGG_BLOCK, j);
    }
    *elow_out = elow;
    *m0_out = m0;
}

// (agg, org: the block table of the fetched path the new part is a stretch of, and the entry it begins at)
inline void extend_chain(Chain &ch, const uint32_t *v, const uint32_t *sv, const uint32_t *pc, size_t n, const PartAgg *known = nullptr,
                         const uint32_t *agg = nullptr, size_t org = 0, const uint32_t *dv = nullptr, const uint32_t *ds = nullptr) {
    if (n == 0) return;
    ch.parts.push_back(Chain::Part{v, sv, pc, n, ch.len, ch.mx_all, ch.m0_all, agg, org, dv, ds});
    ch.len += n;
    PartAgg a;
    if (known) {
        a = *known;  // (the caller has been over the part already)
    } else {
        const RangeAgg r = range_agg(v - org, sv - org, pc - org, agg, org, org + n);
        a.mx = r.mx;
        a.m0 = r.m0;
        a.lo = r.lo_nz;
        a.sz = r.sz;
    }
    ch.mx_all = std::max(ch.mx_all, a.mx);
    ch.m0_all = std::max(ch.m0_all, a.m0);
    ch.low_nz = std::min(ch.low_nz, a.lo);
    ch.size += a.sz;
}
// the part of T that holds index idx (idx < ch.len)
inline size_t part_of(const Chain &ch, size_t idx) {
    size_t lo = 0, hi = ch.parts.size() - 1;
    while (lo < hi) {
        const size_t mid = (lo + hi + 1) / 2;
        if (ch.parts[mid].start <= idx) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
// highest coordinate / highest id + 1 of a coordinate-free vertex over T[0 .. idx)
inline void chain_before(const Chain &ch, size_t idx, uint32_t *mx_out, uint32_t *m0_out) {
    uint32_t mx = 0, m0 = 0;
    if (idx > 0 && !ch.parts.empty()) {
        const size_t pi = part_of(ch, idx - 1);
        const Chain::Part &pt = ch.parts[pi];
        mx = pt.mx;
        m0 = pt.m0;
        // (a splice asks for the prefix up to a few hundred vertices before the end of a part of thousands)
        const size_t cnt = idx - pt.start;
        const RangeAgg r = range_agg(pt.v - pt.org, pt.s - pt.org, pt.pc - pt.org, pt.agg, pt.org, pt.org + cnt);
        mx = std::max(mx, r.mx);
        m0 = std::max(m0, r.m0);
    }
    *mx_out = mx;
    *m0_out = m0;
}
// T as flat arrays (vertices / steps / coordinates; a null destination is skipped)
inline void flatten_chain(const Chain &ch, uint32_t *dv, uint32_t *ds, uint32_t *dpc) {
    for (const Chain::Part &pt : ch.parts) {
        if (dv) std::memcpy(dv + pt.start, pt.v, pt.n * 4);
        if (ds) std::memcpy(ds + pt.start, pt.s, pt.n * 4);
        if (dpc) std::memcpy(dpc + pt.start, pt.pc, pt.n * 4);
    }
}
// how many vertices T and a segment's path P have in common going backwards from T's last vertex = P[be] (beyond that
// pair itself: same vertices, same steps from the second common vertex on)
inline size_t common_back(const Chain &ch, const View &P, size_t be) {
    size_t t = 0;
    if (ch.len == 0) return 0;
    size_t pi = ch.parts.size() - 1, off = ch.parts[pi].n;  // T[e - t] = parts[pi][off - 1] while walking back
    // invariant: (pi, off - 1) addresses T[e - t]
    while (t < ch.len - 1 && t < be) {
        // step of T[e - t] and vertex of T[e - t - 1]
        const uint32_t st_cur = ch.parts[pi].s[off - 1];
        size_t pj = pi, oj = off - 1;  // (pj, oj - 1) will address T[e - t - 1]
        if (oj == 0) {
            pj = pi - 1;
            oj = ch.parts[pj].n;
        }
        if (ch.parts[pj].v[oj - 1] != P.v[be - t - 1] || st_cur != P.s[be - t]) break;
        ++t;
        pi = pj;
        off = oj;
    }
    return t;
}


// what a splice needs to know about the round it happens in
struct MergeCtx {
    uint32_t k = 0;
    uint64_t deviation = 0;
    uint64_t split = 0;     // (uint64_t)(contig length * startSplit): from hasSize + nowSize >= split on a walk can leap
    uint64_t has_size = 0;  // the round's hasSize (sum of the steps of the contig's path so far)
    // what the rounds before this one have committed to the contig's global structures (globalUniqueTable / ctgGlobalPosTable,
    // PAlgorithm.cpp:146-149, 290-296): the coordinate window [g_lo, g_hi] of their paths (0xFFFFFFFF, 0: nothing yet) and the
    // highest id + 1 of a coordinate-free vertex on them.  They count for segments kept from an earlier round only.
    uint32_t round = 0;
    uint32_t g_lo = 0xFFFFFFFFu, g_hi = 0, g_free_hi = 0;
};

// Adoption of a finished segment by a chain whose last vertex lies inside it.
//
// Let T be the chain's path (its walk stands at an iteration boundary of graphTravel behind T's last vertex) and P the
// segment's path.  Condition: the last vertex of T is P[be], and going backwards T and P agree on t + 1 vertices
// (T[e - j] == P[be - j], same steps from the second common vertex on).  While leaping is impossible every successor a
// classification can accept follows the contig (isEdgeSimilar on the contig coordinate: PABruijnGraph.cpp:385-400 and the
// grade table of checkPosition :158-164 leave nothing else once Skip and leaps are excluded, PAlgorithm.tcc:69-86), so
//   (a) the coordinate windows (existCtgPos) never decide anything, and
//   (b) an accepted successor of a vertex at coordinate c lies at >= c + step - deviation > c - deviation.
// Hence the ONLY state through which the past acts on the continuation are the visited sets, and the two walks differ in
// them by D = vertices of T before the common stretch (+) vertices of P before it.  P's job reports how far below the
// coordinate of an iteration's branch vertex any of its probes went (max_back) and how long its chosen paths were
// (max_chosen): every candidate vertex the segment's walk examined in an iteration that contributes a vertex behind P[be]
// has a coordinate > min(coord of P[q ..]) - max_back - deviation with q = be + 1 - max_chosen.  If that bound exceeds
// the largest coordinate in D, no vertex of D was ever a candidate: the real walk, continuing from T, takes exactly P's
// decisions, and P[be + 1 ..] is its path — as far as leaping stays impossible for it, which is checked with the true
// sizes: hasSize + k + (steps of T) + (steps adopted) + (largest probe of the segment's walk) < split size.
// Returns 0: not adoptable (now), 1: adopted up to the end of P, 2: adopted up to where leaping may begin.
inline int try_merge(const MergeCtx &M, Chain &ch, const Seg &sg, uint64_t *adopted) {
    const View &P = sg.P;
    if (!sg.usable || ch.len == 0 || P.n == 0) return 0;
    const size_t e = ch.len - 1;
    const uint32_t last_v = ch.parts.back().v[ch.parts.back().n - 1];
    size_t be = P.n;  // the last vertex of T in P
    for (size_t x = 0; x < P.n; ++x)
        if (P.v[x] == last_v) {
            be = x;
            break;
        }
    if (be == P.n) return 0;
    const size_t t = common_back(ch, P, be);
    const size_t a = e - t, b = be - t;
    if (t < 4) return 0;
    uint32_t dT = 0, d0 = 0, dP = 0;
    chain_before(ch, a, &dT, &d0);
    for (size_t x = 0; x < b; ++x) dP = std::max(dP, P.pc[x]);
    uint64_t dmax = std::max(dT, dP);
    // a segment kept from an earlier round: the vertices the rounds since have marked globally are visited for the real walk
    // and were not for the segment's — they belong to D (all of them have a coordinate <= g_hi, or none: such a vertex is
    // never a candidate while leaping is impossible)
    if (sg.round < M.round) dmax = std::max<uint64_t>(dmax, M.g_hi);
    const size_t q = be + 1 > sg.max_chosen ? be + 1 - sg.max_chosen : 0;
    const uint64_t split = M.split;
    const uint64_t base = M.has_size + M.k + ch.size + sg.max_probe + 1;
    if (base >= split) return 0;  // (no room: the caller resumes exactly)
    const uint64_t room = split - base;  // steps that may still be adopted
    // ONE pass over P[q ..]: the lowest coordinate (condition 2), how far the steps behind P[be] stay below `room`
    // (condition 3: `last`), and what the chain has to know about the adopted stretch
    uint64_t low = 0xFFFFFFFFull;
    for (size_t x = q; x <= be; ++x) low = std::min<uint64_t>(low, P.pc[x]);
    size_t last = be;
    PartAgg agg;
    {
        // (these reductions run over every adopted vertex of a block, 14 M at configs[1], on the thread every contig waits for:
        // the block table answers them)
        const RangeAgg r = range_agg(P.v, P.s, P.pc, P.agg, be + 1, P.n);
        low = std::min<uint64_t>(low, r.lo_all);
        if (r.sz < room) {  // the whole rest fits below the size at which leaping begins (every prefix sum does)
            agg.mx = r.mx;
            agg.m0 = r.m0;
            agg.lo = r.lo_nz;
            agg.sz = r.sz;
            last = P.n - 1;
        } else {
            const size_t n_tail = P.n - (be + 1);
            const uint32_t *tv = P.v + (be + 1), *ts = P.s + (be + 1), *tp = P.pc + (be + 1);
            for (size_t x = 0; x < n_tail; ++x) {
                if (agg.sz + ts[x] < room) {
                    agg.add(tv[x], ts[x], tp[x]);
                    last = be + 1 + x;
                } else {
                    break;
                }
            }
        }
    }
    if (low <= dmax + sg.max_back + M.deviation) return 0;
    if (last == be && last + 1 < P.n) return 0;
    extend_chain(ch, P.v + (be + 1), P.s + (be + 1), P.pc + (be + 1), last - be, &agg, P.agg, be + 1, P.dv ? P.dv + (be + 1) : nullptr,
                 P.ds ? P.ds + (be + 1) : nullptr);
    if (adopted) *adopted += last - be;
    return last + 1 == P.n ? 1 : 2;
}

// Adoption of a finished segment of the LEAPING zone (TRAV_MODE_LEAP).  Leaping being possible, a classification admits
// more: Skip grades and landings on other contigs (PAlgorithm.tcc:69-86), hence vertices without a contig coordinate on
// the paths and records whose fate the coordinate windows decide (existCtgPos).  What an examined record's verdict
// depends on, besides the record: the contig's global marks and the landing rule (the same for both walks), the probe's
// own marks and window (fresh at an iteration boundary), the travel-visited set, the travel window, and whether leaping
// is possible.  The splice is exact if
//   (1) T ends at P[be], an iteration boundary of the segment's walk too (both walks classify that vertex at the top
//       level next, with no probe under way), and going backwards T and P agree on t + 1 >= 5 vertices and steps;
//   (2) the real walk can leap from here on (true sizes), as the segment's walk could all along;
//   (3) the travel windows agree: same upper end (the highest coordinate of T and of P[.. be]); the lower ends are lowM
//       (lowest coordinate of T) and the value forced on the segment's walk (win_low): no window-dependent record it
//       examined may lie between the two (the job reports the extremes of those records), and P[.. b) itself lies inside
//       the real walk's window;
//   (4) no examined record of an iteration that starts at P[be] or later leads to a vertex only ONE of the walks has
//       visited, D = T[.. a) + P[.. b).  A record is one of three kinds.  Window-dependent (a coordinate, not following
//       the contig): a target in D lies inside both travel windows by (3) and is rejected by both.  Contig-following: the
//       job logs, per iteration, the lowest coordinate of any such record it examined (elow): above every coordinate
//       in D.  No coordinate: the job logs the lowest new id of any such target (m0; these ids are ordered by the
//       reference coordinate, trav_order): above every id of that kind in D.  Records examined by probes of earlier
//       iterations that are still walking are logged with the iteration they are examined in.
// Then every verdict from P[be] on is the one the real walk reaches, and P[be + 1 ..] is its path.
// Returns 0: refused, 1: adopted to the end of P.
inline int try_merge_leap(const MergeCtx &M, Chain &ch, const Seg &sg, uint64_t *adopted, int *why_out = nullptr) {
    const View &P = sg.P;
    auto refuse = [&](int why) {
        if (why_out) *why_out = why;
        return 0;
    };
    if (!sg.usable || ch.len == 0 || P.n == 0 || !P.xl) return refuse(0);
    const size_t e = ch.len - 1;
    const uint32_t last_v = ch.parts.back().v[ch.parts.back().n - 1];
    size_t be = P.n;
    for (size_t x = 0; x < P.n; ++x)
        if (P.v[x] == last_v) {
            be = x;
            break;
        }
    if (be == P.n) return refuse(1);
    if (!(P.xh[be] >> 31)) return refuse(2);
    const size_t t = common_back(ch, P, be);
    const size_t a = e - t, b = be - t;
    if (t < 4) return refuse(1);
    const uint64_t split = M.split;
    if (M.has_size + M.k + ch.size < split) return refuse(3);
    // P[0 .. be]: highest coordinate; P[0 .. b): highest / lowest coordinate, highest id + 1 of a coordinate-free vertex
    uint32_t p_top = 0, p_dmax = 0, p_min = 0xFFFFFFFFu, p_d0 = 0;
    for (size_t x = 0; x <= be; ++x) {
        const uint32_t c = P.pc[x];
        p_top = std::max(p_top, c);
        if (x < b) {
            p_dmax = std::max(p_dmax, c);
            if (c == 0) p_d0 = std::max(p_d0, P.v[x] + 1u);
            else p_min = std::min(p_min, c);
        }
    }
    if (ch.mx_all != p_top) return refuse(4);
    uint32_t lowM = ch.low_nz;
    const bool kept = sg.round < M.round;
    // A segment kept from an earlier round: the real walk rejects a window-dependent record inside its travel window
    // [lowest coordinate of T, top] AND inside the global window [g_lo, g_hi] of the rounds since (existCtgPos on
    // ctgGlobalPosTable, PAlgorithm.cpp:160-168); when the two touch (g_hi >= lowM - 1) their union is one interval from
    // min(lowM, g_lo) up, and that is what the segment's forced window has to agree with.  A gap between them cannot be
    // told from the job's extremes: the travel window alone counts then (stricter).
    if (kept && M.g_hi != 0u && M.g_lo <= M.g_hi && (uint64_t)M.g_hi + 1u >= lowM) lowM = std::min(lowM, M.g_lo);
    if (sg.wd_below_max != 0u && sg.wd_below_max >= lowM) return refuse(5);
    if (sg.wd_forced_min < lowM) return refuse(5);
    if (p_min < lowM) return refuse(5);
    // ONE pass over P[be ..]: the iterations that start at a boundary >= be (lowest contig-following coordinate /
    // coordinate-free id examined), and what the chain has to know about the adopted stretch
    uint32_t elow = 0xFFFFFFFFu, m0 = 0xFFFFFFFFu;
    range_xagg(P.xl, P.xh, P.xagg, be, P.n, &elow, &m0);
    PartAgg agg;
    {
        const RangeAgg r = range_agg(P.v, P.s, P.pc, P.agg, be + 1, P.n);
        agg.mx = r.mx;
        agg.m0 = r.m0;
        agg.lo = r.lo_nz;
        agg.sz = r.sz;
    }
    uint32_t t_dmax = 0, t_d0 = 0;
    chain_before(ch, a, &t_dmax, &t_d0);
    uint32_t dmax = std::max(t_dmax, p_dmax);
    uint32_t d0 = std::max(t_d0, p_d0);  // (id + 1, 0: none)
    if (kept) {  // (the vertices the rounds since have marked globally belong to D)
        dmax = std::max(dmax, M.g_hi);
        d0 = std::max(d0, M.g_free_hi);
    }
    if (elow <= dmax) return refuse(6);
    if (d0 != 0u && m0 != 0xFFFFFFFFu && m0 + 1u <= d0) return refuse(7);
    const size_t last = P.n - 1;
    extend_chain(ch, P.v + (be + 1), P.s + (be + 1), P.pc + (be + 1), last - be, &agg, P.agg, be + 1, P.dv ? P.dv + (be + 1) : nullptr,
                 P.ds ? P.ds + (be + 1) : nullptr);
    if (adopted) *adopted += last - be;
    if (why_out) *why_out = -1;
    return 1;
}

// ---- what a chain does next --------------------------------------------------------------------------------------------
// A round's segments: R.segs[0 .. n_spec) cannot leap, R.segs[n_spec ..) are the pieces of the leaping zone, each kind by
// checkpoint coordinate; the kinds OVERLAP around the coordinate where leaping becomes possible.  Which kind a chain may
// adopt is decided by its TRUE size: before hasSize + k + (its steps) reaches the split size only segments that cannot leap
// (try_merge's condition 3), from then on only pieces of the leaping zone (try_merge_leap's condition 2).

// the coordinate at (or beyond) which the walk towards segment q stops: a little into the segment, so that the two paths
// have a stretch in common
inline uint32_t stop_for(const RoundState &R, size_t q, uint64_t seg_ov) {
    const uint64_t x = (uint64_t)R.segs[q].x + seg_ov;
    return (uint32_t)(R.segs[q].leap ? std::min<uint64_t>(x, 0xFFFFFFFFull) : std::min<uint64_t>(x, R.zone_end));
}

struct Next {  // what the caller has to do for the chain
    enum What { Nothing, Resume } what = Nothing;  // Nothing: it is final, waits for a job or for a segment (ch.waiting_seg)
    uint32_t stop = 0;        // Resume: continue the chain exactly, up to this coordinate (0: to the end) ...
    bool until_leap = false;  // ... or only to the first iteration boundary from which the walk can leap (TRAV_MODE_UNTIL_LEAP)
};
struct AdvanceStats {
    uint64_t adopted = 0, leap_adopted = 0, merge_fail = 0;
    uint64_t leap_refused[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// Called when a chain's job has ended at a stop coordinate, or a segment it waits for has finished: adopts what can be
// adopted, then says how the chain goes on.
inline Next advance_chain(RoundState &R, Chain &ch, const MergeCtx &M, uint64_t seg_ov, AdvanceStats &S, int *last_refusal = nullptr) {
    const int n_spec = (int)R.n_spec, n_all = (int)R.segs.size();
    for (;;) {
        if (ch.final || ch.job >= 0) return Next{};
        // (where the chain stands: its highest coordinate — its last vertex may have none in the leaping zone)
        const uint32_t cT = ch.mx_all;
        const bool can = M.has_size + M.k + ch.size >= M.split;  // the real walk can leap from here on
        const int lo = can ? std::max(ch.next_leap, n_spec) : std::min(ch.next_seg, n_spec), hi = can ? n_all : n_spec;
        // the last segment of that kind that starts at or before the chain's end
        int j = -1;
        for (int q = hi - 1; q >= lo; --q)
            if (R.segs[(size_t)q].x <= cT) {
                j = q;
                break;
            }
        if (j < 0 || cT == 0) {  // no segment to adopt here: walk on exactly
            ch.waiting_seg = -1;
            if (cT == 0) return Next{Next::Resume, 0u, false};
            if (lo < hi) return Next{Next::Resume, stop_for(R, (size_t)lo, seg_ov), false};  // ... to the next checkpoint of the kind
            // past the segments that cannot leap and not yet able to leap: across that point, where the pieces of the
            // leaping zone (if any) take over; otherwise to the end
            return Next{Next::Resume, 0u, !can && n_all > n_spec};
        }
        Seg &sg = R.segs[(size_t)j];
        if (!sg.leap && cT >= R.zone_end) {  // past the zone of the segments that cannot leap
            ch.next_seg = n_spec;
            continue;
        }
        if (!sg.done) {
            ch.waiting_seg = j;
            return Next{};
        }
        ch.waiting_seg = -1;
        uint64_t got = 0;
        int why = -1;
        const int m = sg.leap ? try_merge_leap(M, ch, sg, &got, &why) : try_merge(M, ch, sg, &got);
        S.adopted += got;
        if (sg.leap) {
            if (m) S.leap_adopted += 1;
            else {
                S.leap_refused[why & 7] += 1;
                if (last_refusal) *last_refusal = why;
            }
        }
        (sg.leap ? ch.next_leap : ch.next_seg) = j + 1;
        if (m == 1) {
            if (!sg.stopped) {  // the segment's walk ended by itself, and so does the real one
                ch.final = true;
                return Next{};
            }
            continue;  // on to the next segment
        }
        if (m == 2) {  // adopted up to where leaping may begin: nothing more of this kind
            ch.next_seg = n_spec;
            continue;
        }
        S.merge_fail += 1;
        // (the next turn of the loop finds no started segment of the kind any more and resumes up to the next checkpoint, or to the end)
    }
}

}  // namespace stitch
}  // namespace pagdev
