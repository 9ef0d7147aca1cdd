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

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// PositionMapper (position/PositionMapper.cpp:16-64) over contig lengths
struct Mapper {
    std::vector<uint64_t> starts, sizes;
    Mapper(const uint32_t *len, uint64_t n) {
        for (uint64_t i = 0; i < n; ++i) sizes.push_back(len[i]);
        if (sizes.empty()) return;
        starts.push_back(sizes[0]);
        for (size_t i = 1; i < sizes.size(); ++i) starts.push_back(starts.back() + 3 * sizes[i - 1] + std::max(sizes[i - 1], sizes[i]));
        starts.push_back(starts.back() + 4 * sizes.back());
    }
    uint64_t dualToSingle(int64_t idx, int64_t pos) const {
        if (idx == 0) return 0;
        size_t i = (size_t)(idx > 0 ? idx - 1 : -idx - 1);
        return starts[i] + (idx > 0 ? 0 : 2 * sizes[i]) + (uint64_t)pos;
    }
    std::pair<int64_t, int64_t> singleToDual(uint64_t single) const {
        if (single == 0) return {0, 0};
        auto it = std::upper_bound(starts.begin(), starts.end(), single);
        if (it != starts.begin()) it = std::prev(it);
        int64_t idx = it - starts.begin();
        uint64_t off = single - *it;
        uint64_t sz = (size_t)idx < sizes.size() ? sizes[(size_t)idx] : 0;
        if (off >= 2 * sz) {
            off -= 2 * sz;
            idx = -(idx + 1);
        } else {
            ++idx;
        }
        return {idx, (int64_t)off};
    }
};

std::string code2kmer(uint32_t code, uint32_t k) {
    std::string s(k, 'A');
    for (uint32_t i = 0; i < k; ++i) {
        s[k - 1 - i] = "ACGT"[code & 3u];
        code >>= 2;
    }
    return s;
}

// PAlgorithm::editDistance (PAlgorithm.cpp:46-69)
size_t edit_distance(const std::string &a, const std::string &b) {
    // (two rows of the table; on the stack for k-mer sized strings: this runs once per re-seed candidate)
    size_t stack_rows[2][40];
    std::vector<size_t> heap_rows;
    size_t *dp[2] = {stack_rows[0], stack_rows[1]};
    if (b.size() + 1 > 40) {
        heap_rows.assign(2 * (b.size() + 1), 0);
        dp[0] = heap_rows.data();
        dp[1] = heap_rows.data() + b.size() + 1;
    }
    size_t flag = 0;
    for (size_t j = 0; j <= b.size(); ++j) dp[flag][j] = j;
    flag ^= 1;
    for (size_t i = 1; i <= a.size(); ++i) {
        for (size_t j = 0; j <= b.size(); ++j) {
            if (j == 0) {
                dp[flag][j] = i;
            } else {
                dp[flag][j] = std::min(dp[flag ^ 1][j] + 1, dp[flag][j - 1] + 1);
                dp[flag][j] = std::min(dp[flag][j], dp[flag ^ 1][j - 1] + (a[i - 1] == b[j - 1] ? 0 : 1));
            }
        }
        flag ^= 1;
    }
    return dp[flag ^ 1][b.size()];
}

// a vertex of a running travel sequence as the per-round control needs it: new id, step, contig coordinate.  The full
// records (k-mer, reference coordinate, abundance) are gathered once, for the finished sequences.
struct LNode {
    uint32_t u;
    int32_t step;
    uint32_t ctg;
    LNode() {}  // (left as it is by vector::resize: a round's path is written over the new elements right away, 14 M of them at configs[1])
    LNode(uint32_t uu, int32_t st, uint32_t c) : u(uu), step(st), ctg(c) {}
};

struct CtgState {
    size_t pendingFirst = 0;       // (choose + gather: the first vertex of the round's path in `travel`, and the step it gets)
    int32_t pendingFirstStep = 0;
    uint32_t ci = 0;  // contig index
    bool forward = true;
    int64_t chosenOne = 0;
    uint32_t len = 0;
    uint32_t ctgLeft = 0, ctgRight = 0, revLeft = 0, revRight = 0;
    uint64_t nodesOff = 0;  // offset of this contig's node table
    std::vector<LNode> travel;
    std::vector<pag_path_node> seeds;
    int64_t varLen = 0;
    std::deque<uint32_t> ctgQ, refQ;
    bool finalLeap = false, done = false;
    bool delivered = false;  // its finished sequence has been filtered, gathered and sent to the host
    bool committed = false;  // a walk has been recorded in the global visited structures (device: gbits / gset)
    uint32_t gwinLo = 0xFFFFFFFFu, gwinHi = 0;
    uint32_t gFreeHi = 0;  // highest id + 1 of a coordinate-free vertex on the committed paths (walk_stitch.hpp MergeCtx::g_free_hi)
    uint32_t *gset = nullptr;   // device: global visited, vertices outside the strand's id range
    uint32_t gcap = 0;
    uint32_t *gbits = nullptr;  // device: global visited bitmap over [inLo, inHi)
    uint32_t inLo = 0, inHi = 0;
    std::vector<uint32_t> outsideU;  // the entries of gset (new ids)
    uint64_t seqCap = 0;
    uint32_t parentCode = 0;  // k-mer of the last contig-consistent path vertex (seed ordering key)
    uint32_t parentU = 0;     // ... that vertex (new id)
    bool haveParent = false;
    // The last round of a contig that leaps (the contig is finished by it): its walk never comes to `travel` — the parts are
    // put one behind the other on the device, behind room for what `travel` holds, and delivered from there.
    struct DevTail {
        bool on = false;
        uint32_t *d_ids = nullptr;  // ids at [0, cap), steps at [cap, 2 cap); the tail from entry m0 on
        size_t cap = 0, m0 = 0, n = 0;
        uint32_t last_ctg = 0;      // coordinate of the tail's last vertex (the "Pump it" test)
    } tail;
};

uint64_t pow2_at_least(uint64_t x) {
    uint64_t p = 1024;
    while (p < x) p <<= 1;
    return p;
}

// The traversal's view of a finished graph: compact CSR with dense ids, vertices renumbered by contig coordinate, and the
// successor records of every vertex (searchSuccessors + checkPosition for all of them, PABruijnGraph.cpp:143-197) — built
// once per graph and pair of (deviation, error rate), kept in the handle (g->tg).  Pool slots TRAV_SLOT0 .. + TRAV_GRAPH_SLOTS.
constexpr int TRAV_GRAPH_SLOTS = 22;  // (+ 2 behind them for a regional graph's incomplete-vertex bitmap, + 2 for the view's scratch)
constexpr int TRAV_EXTRA_SLOTS = 4;  // (incomplete-vertex bitmap + its scratch, the two of the view)

// ---- the view of ONE handle's traversals -----------------------------------------------------------------------------
// A traversal of contig strand S (PAlgorithm::travelSequence for one (contig, orientation)) only ever examines
//   * the vertices on S;
//   * vertices with a contig coordinate elsewhere as leap targets, and a leap that lands beyond the first (1 - startSplit)
//     of its strand is dropped (classifySuccessors, PAlgorithm.tcc:60-67): the landing zones of every strand are enough;
//   * vertices WITHOUT a contig coordinate once it can take a Skip grade, i.e. once hasSize + nowSize >= ctgLen x startSplit
//     (PAlgorithm.tcc:69-86) — in the last tenth of the strand and beyond its end, along the reference, until it lands.
// Present or absent, anything else never changes a classification (the argument of pag_shard_select, k_select.hip, which cuts
// a block's graph the same way for the ranks of a multi-GPU run), so the view is built from these alone: at BASELINE
// configs[1] 4 of 10 vertices — the opposite strand of every contig (every read is emitted on both strands, one is
// traversed) and the coordinate-free vertices along the first 9/10 of every contig are left out, and the successor stage —
// 42 % of a step in round 3 — runs over what is left.  NEVER SILENTLY WRONG: a coordinate-free vertex within a successor's
// reach of an open band end carries a poison record, and a vertex on a strand whose reference coordinate lies in no band (its
// coordinate-free successors were left out) carries a marker record that counts wherever a Skip grade could be taken
// (k_mark_incomplete, k_succ); a walk that examines one reports it and pag_travel walks again on the whole graph's view
// (g->view_off).  Zones start PAG_VIEW_MARGIN (3 % of the contig, at least 4 kb) before the coordinate where leaping
// would begin if steps and coordinates agreed: the pieces of the leaping zone start a little before it (PAG_LEAP_LEFT) and
// the sum of the steps runs ahead of the coordinate by ~0.6 %.  Bands reach PAG_VIEW_HALO (100 kb) beyond the reference
// stretch the zone's vertices map to.
struct ViewRegion {
    std::vector<uint32_t> civ, riv;  // [lo, hi) pairs, sorted, disjoint
    std::vector<uint8_t> ropen;      // per band end: the graph goes on beyond it
};
void merge_intervals(std::vector<std::pair<uint64_t, uint64_t>> &iv) {
    std::sort(iv.begin(), iv.end());
    size_t w = 0;
    for (size_t i = 0; i < iv.size(); ++i) {
        if (iv[i].second <= iv[i].first) continue;
        if (w && iv[i].first <= iv[w - 1].second) iv[w - 1].second = std::max(iv[w - 1].second, iv[i].second);
        else iv[w++] = iv[i];
    }
    iv.resize(w);
}
int trav_view_region(pag_graph *g, const WalkConfig &cfg, const uint32_t *ctg_len, uint64_t n_ctgs, const int32_t *orient, const uint32_t *ref_len,
                     uint64_t n_refs, double startSplit, DevBuf &scratch, ViewRegion *out) {
    hipStream_t s = g->stream;
    const Mapper cm(ctg_len, n_ctgs), rm(ref_len, n_refs);
    const uint64_t halo = cfg.view_halo;
    const double margin_frac = cfg.view_margin_set ? 0.0 : 0.03;
    const uint64_t margin_min = cfg.view_margin;
    std::vector<std::pair<uint64_t, uint64_t>> civ, zones;
    const double leap_min = 1.0 - startSplit;
    for (uint64_t c = 0; c < n_ctgs; ++c) {
        const uint64_t n = ctg_len[c];
        const uint64_t z = std::min<uint64_t>(n, (uint64_t)((double)n * leap_min) + 2);
        for (int rev = 0; rev < 2; ++rev) {
            const int64_t one = rev ? -(int64_t)c - 1 : (int64_t)c + 1;
            const uint64_t left = cm.dualToSingle(one, 0);
            civ.push_back({left, left + z});  // landing zone of every strand
            const int32_t o = orient[c];
            const bool walked = o == PAG_ORIENT_BOTH || (!rev && o == PAG_ORIENT_FORWARD) || (rev && o == PAG_ORIENT_REVERSE);
            if (!walked) continue;
            civ.push_back({left, left + n});
            const uint64_t split = (uint64_t)((double)n * startSplit);
            const uint64_t margin = std::max<uint64_t>(margin_min, (uint64_t)((double)n * margin_frac));
            zones.push_back({left + (split > margin ? split - margin : 0), left + n});
        }
    }
    merge_intervals(civ);
    std::sort(zones.begin(), zones.end());  // (strands are disjoint: so are their zones)
    // reference stretch every zone's vertices map to
    const uint32_t nz = (uint32_t)zones.size();
    std::vector<uint32_t> zflat(2 * (size_t)nz), zlo(nz), zhi(nz);
    for (uint32_t i = 0; i < nz; ++i) {
        zflat[2 * i] = (uint32_t)zones[i].first;
        zflat[2 * i + 1] = (uint32_t)zones[i].second;
    }
    int rc;
    if ((rc = scratch.alloc(((size_t)nz * 4 + 16) * 4))) return rc;
    uint32_t *d_z = scratch.as<uint32_t>(), *d_lo = d_z + 2 * (size_t)nz, *d_hi = d_lo + nz;
    if (nz) {
        PAG_HIP_TRY(hipMemcpyAsync(d_z, zflat.data(), zflat.size() * 4, hipMemcpyHostToDevice, s));
        if ((rc = trav_zone_bands(g->tval, g->n_t, d_z, nz, d_lo, d_hi, s))) return rc;
        PAG_HIP_TRY(hipMemcpyAsync(zlo.data(), d_lo, (size_t)nz * 4, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipMemcpyAsync(zhi.data(), d_hi, (size_t)nz * 4, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }
    // strand ranges of the references in the single-coordinate space (PositionMapper): [start, start + len) and
    // [start + 2 len, start + 3 len); no position lies between them
    auto range_of = [&](uint64_t x, uint64_t *lo, uint64_t *hi) {
        const auto d = rm.singleToDual(x);
        const size_t i = (size_t)(d.first > 0 ? d.first - 1 : -d.first - 1);
        if (d.first == 0 || i >= rm.sizes.size()) {
            *lo = 0;
            *hi = ~0ull;
            return;
        }
        *lo = rm.starts[i] + (d.first > 0 ? 0 : 2 * rm.sizes[i]);
        *hi = *lo + rm.sizes[i];
    };
    struct Band {
        uint64_t lo, hi;
        bool olo, ohi;
    };
    std::vector<Band> bands;
    for (uint32_t i = 0; i < nz; ++i) {
        if (zhi[i] == 0u || zlo[i] > zhi[i]) continue;  // (no vertex of the zone has a reference coordinate)
        uint64_t a0, a1, b0, b1;
        range_of(zlo[i], &a0, &a1);
        range_of(zhi[i], &b0, &b1);
        Band b;
        b.lo = zlo[i] > halo ? zlo[i] - halo : 0;
        b.hi = (uint64_t)zhi[i] + halo + 1;
        b.olo = b.lo > a0;
        b.ohi = b.hi < b1;
        b.lo = std::max(b.lo, a0);
        b.hi = std::min<uint64_t>(std::min(b.hi, b1), 0xFFFFFFFFull);
        bands.push_back(b);
    }
    std::sort(bands.begin(), bands.end(), [](const Band &x, const Band &y) { return x.lo < y.lo || (x.lo == y.lo && x.hi < y.hi); });
    std::vector<Band> merged;
    for (const Band &b : bands) {
        if (!merged.empty() && b.lo <= merged.back().hi) {
            if (b.hi > merged.back().hi) {
                merged.back().hi = b.hi;
                merged.back().ohi = b.ohi;
            }
        } else {
            merged.push_back(b);
        }
    }
    out->civ.clear();
    out->riv.clear();
    out->ropen.clear();
    for (auto &c : civ) {
        out->civ.push_back((uint32_t)c.first);
        out->civ.push_back((uint32_t)std::min<uint64_t>(c.second, 0xFFFFFFFFull));
    }
    for (const Band &b : merged) {
        out->riv.push_back((uint32_t)b.lo);
        out->riv.push_back((uint32_t)b.hi);
        out->ropen.push_back(b.olo ? 1 : 0);
        out->ropen.push_back(b.ohi ? 1 : 0);
    }
    return PAG_OK;
}
// the strands `want` traverses are among those the view was built for
bool view_serves(const std::vector<int32_t> &have, const int32_t *want, uint64_t n) {
    if (have.size() != n) return false;
    for (uint64_t c = 0; c < n; ++c) {
        const int32_t w = want[c], h = have[c];
        if (w == PAG_ORIENT_NONE || h == PAG_ORIENT_BOTH || w == h) continue;
        return false;
    }
    return true;
}

// orient == nullptr: the view of the whole graph (serves any traversal)
int trav_prepare_graph(pag_graph *g, const uint32_t *ctg_len, uint64_t n_ctgs, const uint32_t *ref_len, uint64_t n_refs, uint64_t deviation,
                       double errorRate, TravGraph *G_out, double *ms_out, const int32_t *orient = nullptr, double startSplit = 0.9) {
    hipStream_t s = g->stream;
    const uint32_t k = g->k;
    int rc;
    int slot = TRAV_SLOT0;
    auto buf = [&](void) { return DevBuf(g, slot++); };
    if (ms_out) *ms_out = 0;
    if (g->tg_ready && (g->tg_dev != deviation || g->tg_err != errorRate)) g->tg_ready = false;
    if (g->tg_ready && g->view_pruned && !(orient && view_serves(g->view_orient, orient, n_ctgs))) g->tg_ready = false;
    if (g->tg_ready) {
        *G_out = g->tg;
        return PAG_OK;
    }
    // a graph that is one rank's region of a sharded build is cut already (pag_shard_select); PAG_TRAVEL_VIEW=whole: never cut
    const WalkConfig cfg = WalkConfig::from_env();
    const double t_entry = now_ms(), alloc_entry = g->alloc_ms;
    const bool prune = orient && !g->regional && !g->view_off && !cfg.view_whole;
    // ---- compact CSR (once per built graph)
    DevBuf b_ncode = buf(), b_npos = buf(), b_nedge = buf(), b_vpos = buf(), b_vcnt = buf(), b_vnode = buf(), b_eto = buf(),
           b_estep = buf(), b_bitmap = buf(), b_rank = buf(), b_ctmp = buf(), b_uold = buf(), b_newid = buf(), b_upos = buf(), b_ucnt = buf(),
           b_soff = buf(), b_succ = buf(), b_ok0 = buf(), b_ov0 = buf(), b_ok1 = buf(), b_ov1 = buf(), b_otmp = buf();
    const uint64_t nn = g->stats.n_nodes, np = g->stats.n_pos, ne = g->stats.n_uniq_edges;
    if (np >= 0xFFFFFFF0ull || ne >= 0xFFFFFFF0ull) {
        set_error("pag_travel: more than 2^32 vertices/edges");
        return PAG_EINVAL;
    }
    const uint64_t n_words = ((1ull << (2 * k)) + 63) / 64;
    if ((rc = b_ncode.alloc((nn + 1) * 4)) || (rc = b_npos.alloc((nn + 2) * 4)) || (rc = b_nedge.alloc((nn + 2) * 4)) ||
        (rc = b_vpos.alloc((np + 4) * 8)) || (rc = b_vcnt.alloc((np + 1) * 2)) || (rc = b_vnode.alloc((np + 1) * 4)) ||
        (rc = b_eto.alloc((ne + 4) * 4)) || (rc = b_estep.alloc((ne + 4) * 4)) || /* (+ 4: k_succ reads the lists four entries at a time) */ (rc = b_bitmap.alloc(n_words * 8)) ||
        (rc = b_rank.alloc(n_words * 4)) || (rc = b_uold.alloc((np + 1) * 4)) || (rc = b_newid.alloc((np + 1) * 4)) ||
        (rc = b_upos.alloc((np + 1) * 8)) || (rc = b_ucnt.alloc((np + 1) * 4)) || (rc = b_soff.alloc((np + 2) * 4)))
        return rc;
    TravGraph G{};
    G.n_nodes = nn;
    G.n_pos = np;
    G.n_edges = ne;
    G.ncode = b_ncode.as<uint32_t>();
    G.npos_off = b_npos.as<uint32_t>();
    G.nedge_off = b_nedge.as<uint32_t>();
    G.vpos = b_vpos.as<uint64_t>();
    G.vcnt = b_vcnt.as<uint16_t>();
    G.vnode = b_vnode.as<uint32_t>();
    G.eto = b_eto.as<uint32_t>();
    G.estep = b_estep.as<uint32_t>();
    G.bitmap = b_bitmap.as<uint64_t>();
    G.rank = b_rank.as<uint32_t>();
    G.uold = b_uold.as<uint32_t>();
    G.newid = b_newid.as<uint32_t>();
    G.upos = b_upos.as<uint64_t>();
    G.ucnt = b_ucnt.as<uint32_t>();
    G.succ_off = b_soff.as<uint32_t>();
    double t_compact = 0;
    {
        const double t0 = t_entry;
        // (PAGRAPH_TIMING: the stage's laps — each ends with the stream idle — and what of them was hipMalloc / hipFree)
        double lap_t = t0, lap_alloc = alloc_entry;
        std::string lap_line;
        auto lap = [&](const char *what) {
            if (!cfg.timing) return;
            hipStreamSynchronize(s);
            const double t = now_ms();
            char b[96];
            std::snprintf(b, sizeof b, " %s %.1f ms (pool %.1f);", what, t - lap_t, g->alloc_ms - lap_alloc);
            lap_line += b;
            lap_t = t;
            lap_alloc = g->alloc_ms;
        };
        size_t tb = trav_compact_tmp_bytes(g->n_t, g->n_e, k, nn);
        if ((rc = b_ctmp.alloc(tb))) return rc;
        ViewRegion vr;
        TravView tv{};
        DevBuf b_view(g, TRAV_SLOT0 + TRAV_GRAPH_SLOTS + 2), b_viewiv(g, TRAV_SLOT0 + TRAV_GRAPH_SLOTS + 3);
        g->view_pruned = false;
        if (prune) {
            if ((rc = trav_view_region(g, cfg, ctg_len, n_ctgs, orient, ref_len, n_refs, startSplit, b_view, &vr))) return rc;
            if ((rc = b_viewiv.alloc((vr.civ.size() + vr.riv.size() + 8) * 4))) return rc;
            uint32_t *d = b_viewiv.as<uint32_t>();
            if (!vr.civ.empty()) PAG_HIP_TRY(hipMemcpyAsync(d, vr.civ.data(), vr.civ.size() * 4, hipMemcpyHostToDevice, s));
            if (!vr.riv.empty()) PAG_HIP_TRY(hipMemcpyAsync(d + vr.civ.size(), vr.riv.data(), vr.riv.size() * 4, hipMemcpyHostToDevice, s));
            tv.civ = d;
            tv.n_civ = (uint32_t)(vr.civ.size() / 2);
            tv.riv = d + vr.civ.size();
            tv.n_riv = (uint32_t)(vr.riv.size() / 2);
            if (cfg.timing) {
                uint64_t cl = 0, rl = 0;
                for (size_t i = 0; i + 1 < vr.civ.size(); i += 2) cl += vr.civ[i + 1] - vr.civ[i];
                for (size_t i = 0; i + 1 < vr.riv.size(); i += 2) rl += vr.riv[i + 1] - vr.riv[i];
                std::fprintf(stderr, "[timing] view region: %zu contig intervals covering %llu coordinates, %zu reference bands covering %llu\n", vr.civ.size() / 2,
                             (unsigned long long)cl, vr.riv.size() / 2, (unsigned long long)rl);
            }
        }
        uint64_t counts[3] = {nn, np, ne};
        // (key widths of the coordinate sorts: the single-coordinate spaces of the contigs and of the references)
        auto bits_of = [](const uint32_t *len, uint64_t n) {
            const uint64_t space = Mapper(len, n).starts.empty() ? 1 : Mapper(len, n).starts.back();
            int b = 1;
            while (b < 32 && (space >> b) != 0) ++b;
            return b;
        };
        const int ctg_bits = bits_of(ctg_len, n_ctgs), ref_bits = bits_of(ref_len, n_refs);
        if ((rc = trav_compact(g->tkey, g->tval, g->tseg, g->tcnt, g->n_t, g->ekey, g->eval, g->eseg, g->n_e, k, nn, np, ne, G,
                               b_ctmp.p, tb, s, prune ? &tv : nullptr, counts)))
            return rc;
        if (prune) {
            G.n_nodes = counts[0];
            G.n_pos = counts[1];
            G.n_edges = counts[2];
            g->view_pruned = true;
            g->view_orient.assign(orient, orient + n_ctgs);
        }
        lap("view + CSR");
        g->view_counts[0] = G.n_nodes;
        g->view_counts[1] = G.n_pos;
        g->view_counts[2] = G.n_edges;
        // coordinate order, then the static half of the epsilon-join for every vertex
        // The two (key, value) scratch pairs of the sorts that follow — the coordinate order of the view's vertices, then the emission
        // stream of the successor records, four to six times as long — and their scratch: on LOAN from the build where it has room.
        // pag_process leaves, beside the finished streams, the other half of each ping-pong pair, the segment kernels' scratch and the
        // sort's (28 bytes per tuple slot, 47 GB for a 90 Mb block at 30x) untouched until its next call; a loan never grows a slot.
        struct Lender {
            pag_graph *g;
            bool lent[64] = {false};
            bool take(DevBuf &b, size_t bytes) {  // smallest idle build slot that holds `bytes`; false: none (b keeps its own slot)
                static const int cand[] = {30, 31, 32, 33, 34, 35, 36, 37, 38, 43, 44};
                int best = -1;
                for (int c : cand) {
                    const pag_graph::Slot &sl = g->pool[c];
                    if (lent[c] || !sl.p || sl.cap < bytes) continue;
                    if (sl.p == (void *)g->tkey || sl.p == (void *)g->tval || sl.p == (void *)g->ekey || sl.p == (void *)g->eval) continue;
                    if (best < 0 || sl.cap < g->pool[best].cap) best = c;
                }
                if (best < 0) return false;
                lent[best] = true;
                b = DevBuf(g, best);
                b.p = g->pool[best].p;
                return true;
            }
            void give_back() { std::fill(lent, lent + 64, false); }
        } lender{g};
        const DevBuf own_ok0 = b_ok0, own_ov0 = b_ov0, own_ok1 = b_ok1, own_ov1 = b_ov1, own_otmp = b_otmp;
        auto scratch_pairs = [&](uint64_t n_elems, size_t tmp_bytes) -> int {  // (values first: the larger requests get the larger slots)
            lender.give_back();
            b_ok0 = own_ok0, b_ov0 = own_ov0, b_ok1 = own_ok1, b_ov1 = own_ov1, b_otmp = own_otmp;
            int r2;
            if (!lender.take(b_ov0, (n_elems + 8) * 8) && (r2 = b_ov0.alloc((n_elems + 8) * 8))) return r2;
            if (!lender.take(b_ov1, (n_elems + 8) * 8) && (r2 = b_ov1.alloc((n_elems + 8) * 8))) return r2;
            if (!lender.take(b_ok0, (n_elems + 8) * 4) && (r2 = b_ok0.alloc((n_elems + 8) * 4))) return r2;
            if (!lender.take(b_ok1, (n_elems + 8) * 4) && (r2 = b_ok1.alloc((n_elems + 8) * 4))) return r2;
            if (!lender.take(b_otmp, tmp_bytes) && (r2 = b_otmp.alloc(tmp_bytes))) return r2;
            return PAG_OK;
        };
        if ((rc = scratch_pairs(G.n_pos, std::max(sort_tmp_bytes(G.n_pos), scan_tmp_bytes(G.n_pos + 2) + 64)))) return rc;
        if ((rc = trav_order(G, b_ok0.as<uint32_t>(), b_ov0.as<uint64_t>(), b_ok1.as<uint32_t>(), b_ov1.as<uint64_t>(), b_otmp.p, &g->n_zero_ctg,
                             ctg_bits, ref_bits, s)))
            return rc;
        lap("coordinate order");
        // a graph that holds a region of the block only: which coordinate-free vertices may have successors beyond it
        G.incomplete = nullptr;
        G.n_zero = (uint32_t)g->n_zero_ctg;
        if (g->regional || prune) {
            // (one rank's region of a sharded build: the bands it was given; this handle's own view: the bands it took)
            const std::vector<uint32_t> &riv = prune ? vr.riv : g->region_ref_iv;
            const std::vector<uint8_t> &ropen = prune ? vr.ropen : g->region_ref_open;
            DevBuf b_inc(g, TRAV_SLOT0 + TRAV_GRAPH_SLOTS), b_inct(g, TRAV_SLOT0 + TRAV_GRAPH_SLOTS + 1);
            const uint32_t n_iv = (uint32_t)(riv.size() / 2);
            if ((rc = b_inc.alloc(((size_t)G.n_pos / 32 + 4) * 4)) || (rc = b_inct.alloc(trav_mark_incomplete_tmp_bytes(n_iv)))) return rc;
            if ((rc = trav_mark_incomplete(G, G.n_zero, riv.data(), ropen.data(), n_iv, (uint32_t)deviation, errorRate, b_inc.as<uint32_t>(), b_inct.p, s)))
                return rc;
        }
        // The successor records: one evaluation of the candidate pairs into an emission stream (12 bytes per slot, two arrays
        // of `cap` slots that the sort ping-pongs between: the sort scratch of the coordinate order), sorted by source, finished
        // into G.succ (k5_travel.hip, k_succ_emit).  The stream's size is not known before the evaluation: the handle remembers
        // the records per vertex of its last graph; a stream that turns out too small is made again with what it asked for.
        uint64_t n_succ = 0, n_slots = 0, n_heavy = 0;
        const uint32_t *sk = nullptr;
        const uint64_t *sv = nullptr;
        {
            const uint64_t nv = G.n_pos;
            uint64_t cap = (uint64_t)((double)nv * g->succ_per_vertex * 1.05) + EMIT_SLACK_SLOTS;
            if (cfg.debug_emit_cap) cap = cfg.debug_emit_cap;
            DevBuf b_heavy = b_ctmp;  // (the compaction's scratch is free: the list of the vertices done by a wave each)
            const size_t heavy_bytes = ((nv + 16) * 4 + 15) & ~(size_t)15;
            if ((rc = b_heavy.alloc(heavy_bytes + 64))) return rc;
            unsigned long long *counters = (unsigned long long *)((char *)b_heavy.p + heavy_bytes);  // (slots taken, records, heavy vertices)
            for (int attempt = 0; attempt < 3; ++attempt) {
                if ((rc = scratch_pairs(cap, sort_tmp_bytes(cap)))) return rc;
                if ((rc = trav_succ_emit(G, (uint32_t)deviation, errorRate, b_ok0.as<uint32_t>(), b_ov0.as<uint64_t>(), b_ok1.as<uint32_t>(), b_ov1.as<uint64_t>(), cap,
                                         b_otmp.p, counters, b_heavy.as<uint32_t>(), cfg.succ_heavy, &n_slots, &n_succ, &n_heavy, &sk, &sv, s)))
                    return rc;
                if (sk) break;
                if (cfg.timing) std::fprintf(stderr, "[timing] successor records: a stream of %llu slots was too small (%llu taken): again\n", (unsigned long long)cap, (unsigned long long)n_slots);
                cap = n_slots + n_slots / 64 + EMIT_SLACK_SLOTS;
            }
            if (!sk) {
                set_error("trav_prepare_graph: the emission stream of the successor records did not fit in three attempts");
                return PAG_EFAULT;
            }
            if (nv) g->succ_per_vertex = (double)n_slots / (double)nv;
        }
        lap("candidate pairs -> sorted stream");
        if (n_succ >= 0xFFFFFFF0ull) {
            set_error("pag_travel: more than 2^32 successor records");
            return PAG_EINVAL;
        }
        if ((rc = b_succ.alloc((n_succ + 1) * sizeof(SuccRec)))) return rc;
        G.succ = b_succ.as<SuccRec>();
        G.n_succ = n_succ;
        if ((rc = trav_succ_finish(G, sk, sv, n_succ, s))) return rc;
        PAG_HIP_TRY(hipStreamSynchronize(s));
        lap("records");
        if (cfg.timing) std::fprintf(stderr, "[timing] traversal graph:%s\n", lap_line.c_str());
        g->tg = G;
        g->tg_dev = deviation;
        g->tg_err = errorRate;
        g->tg_ready = true;
        if (cfg.timing)
            std::fprintf(stderr, "[timing] successor records %llu for %llu vertices (%llu without a contig coordinate) of %llu (%s view: %llu of %llu nodes, %llu of %llu edges); "
                                 "emission stream %llu slots, %llu vertices with more than %u candidate pairs by a wave each\n",
                         (unsigned long long)n_succ, (unsigned long long)G.n_pos, (unsigned long long)g->n_zero_ctg, (unsigned long long)np, g->view_pruned ? "cut" : "whole", (unsigned long long)G.n_nodes,
                         (unsigned long long)nn, (unsigned long long)G.n_edges, (unsigned long long)ne, (unsigned long long)n_slots, (unsigned long long)n_heavy, cfg.succ_heavy);
        t_compact = now_ms() - t0;
    }

    static_assert(TRAV_GRAPH_SLOTS == 22, "slots of the traversal graph");
    if (slot != TRAV_SLOT0 + TRAV_GRAPH_SLOTS) {
        set_error("trav_prepare_graph: slot bookkeeping");
        return PAG_EFAULT;
    }
    *G_out = G;
    if (ms_out) *ms_out = t_compact;
    return PAG_OK;
}

using namespace stitch;  // Piece, View, Seg, Chain, RoundState, PartAgg, extend_chain .. try_merge_leap (walk_stitch.hpp)

// One call of pag_travel (one graph, the contigs of one block): the state of the traversal and the steps it goes through.
// run() is the whole of it — the traversal view, the contigs' tables and first seeds, the job rings, the first rounds, then the
// event loop (finished jobs -> their paths -> the chains move on -> decided rounds are chosen from, spliced, re-seeded or
// delivered) and the epilogue; the members are what those steps share.
struct WalkSession {
    // ---- the call
    pag_graph *g;
    const pag_seqs *ctgs;
    const int32_t *orient;
    const uint32_t *ref_len;
    uint64_t n_refs;
    const pag_travel_params *prm;
    pag_travel_stats *stats;
    WalkSession(pag_graph *g_, const pag_seqs *ctgs_, const int32_t *orient_, const uint32_t *ref_len_, uint64_t n_refs_, const pag_travel_params *prm_,
                pag_travel_stats *stats_)
        : g(g_), ctgs(ctgs_), orient(orient_), ref_len(ref_len_), n_refs(n_refs_), prm(prm_), stats(stats_), cfg(WalkConfig::from_env()),
          mapper(ctgs_->len, ctgs_->n_seqs), refMapper(ref_len_, n_refs_) {}

    // ---- configuration, timing
    hipStream_t s = nullptr;
    double t_begin = 0;
    const WalkConfig cfg;
    bool timing = false, wdebug = false, wtrace = false;
    // PAG_WALK_TRACE: what = 0 job done (a, b = device begin / end in 10 ns ticks), 1 job posted (a = ring, b = mode), 2 round over
    // (a = round, b = leap), 3 round started (a = round, b = seeds)
    struct TraceEv {
        double t;
        uint32_t what, ctg;
        int32_t kind, idx;
        uint64_t a, b, len, classify;
    };
    std::vector<TraceEv> trace;
    double lap_t = 0;
    std::vector<std::pair<const char *, double>> laps;
    uint32_t k = 0;
    uint64_t deviation = 0;
    double errorRate = 0, startSplit = 0;
    size_t topK = 0;
    int slot = TRAV_SLOT0;  // pool slots of the handle are handed out in the order of the buf() calls
    void lap(const char *what) {
        if (!timing) return;
        const double t = now_ms();
        for (auto &l : laps)
            if (l.first == what) {
                l.second += t - lap_t;
                lap_t = t;
                return;
            }
        laps.emplace_back(what, t - lap_t);
        lap_t = t;
    }
    DevBuf buf() { return DevBuf(g, slot++); }
    // pinned host staging area (grown, kept in the handle): packed job results on their way in, uploads on their way out
    std::vector<void *> pinned_parked;  // (freeing host memory synchronises the device: never while the walker grid is resident)
    void *pinned(size_t bytes) {
        if (g->pin_bytes < bytes) {
            if (g->pin_host) {
                if (g->defer_free) pinned_parked.push_back(g->pin_host);
                else hipHostFree(g->pin_host);
            }
            g->pin_host = nullptr;
            g->pin_bytes = 0;
            const size_t want = bytes + bytes / 4 + (1u << 20);
            if (hipHostMalloc(&g->pin_host, want, hipHostMallocDefault) != hipSuccess) {
                set_error("pag_travel: hipHostMalloc(%zu) failed", want);
                return nullptr;
            }
            g->pin_bytes = want;
        }
        return g->pin_host;
    }

    // Pinned memory that keeps what it is given for the whole call: the fetched paths of finished jobs stay where the copy
    // from the device put them (segments and chains refer to them by pointer).  64 MB chunks kept by the handle.
    size_t fetch_chunk = 0, fetch_used = 0;
    static constexpr size_t FETCH_CHUNK = 64u << 20;
    void *fetch_alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (; fetch_chunk < g->fetch_chunks.size(); ++fetch_chunk, fetch_used = 0)
            if (fetch_used + bytes <= g->fetch_chunk_bytes[fetch_chunk]) {
                void *q = (char *)g->fetch_chunks[fetch_chunk] + fetch_used;
                fetch_used += bytes;
                return q;
            }
        void *q = nullptr;
        const size_t want = std::max(FETCH_CHUNK, bytes);
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) {
            set_error("pag_travel: hipHostMalloc(%zu) failed", want);
            return nullptr;
        }
        g->fetch_chunks.push_back(q);
        g->fetch_chunk_bytes.push_back(want);
        fetch_used = bytes;  // (fetch_chunk is the index of the new chunk)
        return q;
    }

    // ---- the traversal view, the contigs
    TravGraph G{};
    double t_compact = 0;
    Mapper mapper, refMapper;
    uint32_t n_ctgs = 0, n_sel = 0;
    std::vector<CtgState> st;  // one entry per (contig, orientation) that is walked
    uint64_t nodes_total = 0;
    DevBuf b_packed, b_nodes, b_starts, b_sizes, b_tc, b_seedout, b_req, b_gset, b_gather, b_vids, b_gbits, b_ckreq, b_ckout;
    std::vector<TravContig> tc;
    static constexpr uint32_t SEED_STRIDE = 4096;
    void fill_contigs() {
        for (uint32_t i = 0; i < n_sel; ++i) {
            CtgState &cs = st[i];
            TravContig &t = tc[i];
            t.nodes = b_nodes.as<uint32_t>() + cs.nodesOff;
            t.n_kmers = cs.len >= k ? cs.len - k + 1 : 0;
            t.ctg_left = cs.ctgLeft;
            t.ctg_right = cs.ctgRight;
            t.rev_left = cs.revLeft;
            t.rev_right = cs.revRight;
            t.split_size = (uint64_t)(cs.len * startSplit);
            t.leap_min = 1 - startSplit;
            t.starts = b_starts.as<uint64_t>();
            t.sizes = b_sizes.as<uint64_t>();
            t.n_ctgs = n_ctgs;
            t.in_lo = cs.inLo;
            t.in_hi = cs.inHi;
            t.g_lo = cs.inLo;
            t.g_hi = cs.inHi;
            t.gbits = cs.committed ? cs.gbits : nullptr;
            t.gset = cs.committed ? cs.gset : nullptr;
            t.gmask = cs.gcap - 1;
            t.gwin_lo = cs.gwinLo;
            t.gwin_hi = cs.gwinHi;
        }
    }
    int upload_contigs() {
        fill_contigs();
        PAG_HIP_TRY(hipMemcpyAsync(b_tc.p, tc.data(), n_sel * sizeof(TravContig), hipMemcpyHostToDevice, s));
        return PAG_OK;
    }
    // vertex attributes for a list of vertex ids
    int fetch_vertices(const std::vector<uint32_t> &vids, std::vector<pag_path_node> &out) {
        out.resize(vids.size());
        if (vids.empty()) return PAG_OK;
        int r;
        if ((r = b_vids.alloc(vids.size() * 4)) || (r = b_gather.alloc(vids.size() * sizeof(pag_path_node)))) return r;
        PAG_HIP_TRY(hipMemcpyAsync(b_vids.p, vids.data(), vids.size() * 4, hipMemcpyHostToDevice, s));
        trav_launch_gather_vertices(G, b_vids.as<uint32_t>(), (uint32_t)vids.size(), b_gather.as<pag_path_node>(), s);
        PAG_HIP_TRY(hipMemcpyAsync(out.data(), b_gather.p, vids.size() * sizeof(pag_path_node), hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        return PAG_OK;
    }

    // ---- statistics of the call
    uint64_t rounds = 0, jobs_total = 0, steps_total = 0, classify_total = 0, probe_total = 0, record_total = 0;
    double t_walk = 0;

    // ---- the walks.
    // The contigs are independent state machines (walk the seeds of the round, choose, splice, stop or re-seed); a persistent
    // walker grid executes whatever jobs are posted, and this loop posts the next piece of work of a contig as soon as what it
    // depends on is done.
    //
    // PIECES.  A graphTravel (PAlgorithm.tcc:172-298) is one chain of dependent steps: a quarter of a million path vertices on
    // a 1 Mb contig, walked by one wavefront at ~1.7 us per step, while the other 255 compute units idle.  The chain is cut
    // along the contig coordinate:
    //   * the walk of a seed (a CHAIN) runs as a job with a stop coordinate: it ends at the first iteration boundary of
    //     graphTravel whose last vertex lies at or beyond it;
    //   * ahead of it, SEGMENT jobs start at checkpoint vertices (the most abundant on-contig vertex near x0 + j * seg_len)
    //     and walk as if they were graphTravels of their own that can never leap (TRAV_MODE_SPEC), each up to the next
    //     checkpoint plus an overlap.  They only exist up to the coordinate at which the real walk could start leaping;
    //   * when a chain has reached the start of a finished segment and its tail COINCIDES, vertex for vertex, with a stretch of
    //     that segment's path, the rest of the segment's path is adopted (see `try_merge` for the condition under which that is
    //     exactly what the real walk would have done) and the chain goes on to the next segment;
    //   * where no segment can be adopted the chain continues as a RESUME job (the path so far is handed to the walker, which
    //     marks it visited and goes on exactly as graphTravel would), with the next checkpoint as its stop coordinate, or
    //     without one from the zone where leaping becomes possible to the end of the walk.
    // The result is vertex-for-vertex the path of the un-cut walk (PAG_WALK_PIECES=0 runs that, tests compare both with the
    // host restatement of the reference), and the critical path of a contig shrinks from the whole contig to one segment
    // plus the leaping zone.
    enum { CB_SEQV = 0, CB_SEQS, CB_ARV, CB_ARS, CB_TSET, CB_PSET, CB_STAMP, CB_TBITS, CB_SEQX, CB_N };
    enum { GRP_ROUND = 0, GRP_CHAIN0 = 1, GRP_FINAL = 9, GROUPS = 10 };  // buffer groups per contig (chains: top-K <= 8)
    DevBuf cbuf(uint32_t i, int grp, int b) { return DevBuf(g, &g->cpool[((size_t)i * GROUPS + grp) * CB_N + b]); }
    // rings of job records, served in order: 0 chain jobs (what a contig's progress waits for), 1 segment jobs of contigs
    // in a later round (they are further along their critical path), 2 segment jobs of first rounds.  slot = ring * QCAP +
    // number mod QCAP
    // A ring holds the jobs of a round that are in flight; a slot is reused QCAP postings later.  Sized by what the contigs
    // of this call can post in one round (segments every few kb of every strand, top-K <= 8 chains each), twice over.
    uint32_t QCAP = 32768;
    static constexpr uint32_t NR = TRAV_RINGS;
    TravQueue *hq = nullptr;
    TravPosted *hjobs = nullptr;
    TravJobOut *houts = nullptr;
    uint32_t *hdone = nullptr;
    double t_walk0 = 0, tw0 = 0;  // (debug time stamps count from the launch of the walker)
    bool use_pieces = true, use_leap_pieces = true, force_exact = false;
    uint64_t seg_len_env = 0, seg_ov = 0;

    double t_st[4] = {0, 0, 0, 0};  // stitch: bookkeeping / paths of finished jobs / chains moving on; posting (inside the others)
    // (The stitch below is serial on purpose.  Worker threads — spinning, polling or sleeping on a condition variable, 4 to
    // 12 of them — cut the path copies from 33 to 10 ms on the GPU box (16-CPU cgroup quota, busy host), but every HIP call
    // of this thread (posting, fetching, re-seeding) got several times slower while they were active and the walks took
    // 180-450 ms instead of 165-180 ms.)
    std::vector<stitch::RoundState> RS;
    struct JobRef {
        uint32_t ctg = 0;
        int kind = 0;  // 0: chain job (seed or resume), 1: segment
        int idx = 0;   // chain / segment number
        uint64_t init_len = 0;
        bool live = false;
        uint32_t epoch = 0;  // RoundState::seg_epoch of its contig when the job was posted (see is_orphan)
    };
    std::vector<JobRef> jref;
    // A round of a contig is decided when all its chains are final.  Segment jobs of the round that are still waiting or
    // walking then are ORPHANS: nobody will look at their paths (a chain that dead-ends at a fifth of its contig leaves four
    // fifths of the round's segments behind — at BASELINE configs[1] the next round of such a contig used to start when the
    // last of them had been walked, ~85 ms into the walks, and its own walk was the tail everything waited for).  An orphan
    // that no wave has taken yet is cancelled (the wave that takes it reports it done at once); one that is walking finishes
    // into its own buffers — a round's buffers come from the walk arena, which is never handed out twice within one
    // pag_travel; a round that had to fall back on the per-contig slots waits for its jobs as before (RoundState::slot_bufs).
    uint64_t n_orphans = 0;
    uint32_t n_posted[TRAV_RINGS] = {0, 0, 0}, n_live = 0, respeculated = 0;
    std::atomic<uint64_t> n_adopted{0}, n_merge_fail{0}, n_leap_adopted{0}, n_leap_refused[8];
    uint64_t n_seg_jobs = 0, n_resume_jobs = 0, n_leap_jobs = 0;
    WalkerGrid walkers;
    void shutdown_walker() {
        if (!walkers.up) return;
        walkers.shutdown();
        g->defer_free = false;
        for (void *q : g->deferred) hipFree(q);
        g->deferred.clear();
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
    }
    struct JobPlan {
        int kind, idx;
        uint64_t cap;       // sequence capacity (vertices)
        uint32_t start_vid; // old id of the start vertex
        uint32_t mode, stop_pc;
        const Chain *init;  // RESUME: the chain whose path so far the job continues
        bool exact;
        uint32_t win_lo = 0, win_hi = 0;  // id range of the job's direct-mapped marks (0, 0: the whole strand)
        uint32_t win_low = 0;             // TRAV_MODE_LEAP: forced lower end of the travel coordinate window
    };
    bool need_publish = false;
    // what the posted batches want cleared before their jobs become visible (hash sets, stamps, travel bits): collected, and
    // cleared by ONE launch when the batch is published (trav_clear_ranges) — five hipMemsetAsync per batch, 673 fill kernels of
    // ~14 us per block at configs[1], ran one after the other on the stream in front of the first job
    std::vector<TravClear> clears;
    void want_clear(void *p, size_t bytes, uint32_t byte_value) {
        if (bytes) clears.push_back(TravClear{p, (uint64_t)bytes, byte_value * 0x01010101u, 0u});
    }
    int flush_clears() {  // (asynchronous: the list is read from pinned memory that lives as long as the walks; publish() waits for the stream)
        if (clears.empty()) return PAG_OK;
        TravClear *d = (TravClear *)fetch_alloc(clears.size() * sizeof(TravClear));
        if (!d) return PAG_ENOMEM;
        std::memcpy(d, clears.data(), clears.size() * sizeof(TravClear));
        const int r = trav_clear_ranges(d, clears.size(), s);
        clears.clear();
        return r;
    }
    // a prepared job enters its ring (in posting order; the walker takes the rings' jobs in that order)
    struct Deferred {
        TravPosted P;
        JobRef jr;
    };
    std::vector<std::vector<Deferred>> deferred;
    bool defer_ring2 = false;
    // A job enters its ring when the slot it takes (its number mod QCAP) is free again; until then it waits in the ring's
    // backlog, in posting order (a ring smaller than the jobs of a round is a matter of flow control, not an error).
    struct Backlogged {
        TravPosted P;
        JobRef jr;
    };
    std::deque<Backlogged> backlog[TRAV_RINGS];
    bool place_job(uint32_t ring, const TravPosted &P, const JobRef &jr2) {
        const uint32_t jn = n_posted[ring], slot = ring * QCAP + jn % QCAP;
        if (jref[slot].live) return false;
        hjobs[slot] = P;
        hdone[slot] = 0;
        jref[slot] = jr2;
        if (jr2.kind == 0) RS[jr2.ctg].chains[(size_t)jr2.idx].job = (int)slot;
        n_posted[ring] += 1;
        return true;
    }
    int commit_job(uint32_t ring, const TravPosted &P, const JobRef &jr2, uint32_t mode, uint32_t stop_pc) {
        if (jr2.kind == 0) {
            Chain &ch = RS[jr2.ctg].chains[(size_t)jr2.idx];
            ch.job = 0x7FFFFFFF;  // (outstanding; the slot number follows when the job enters the ring)
            ch.job_mode = mode;
            ch.job_stop = stop_pc;
        }
        n_live += 1;
        RS[jr2.ctg].live_jobs += 1;
        jobs_total += 1;
        if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 1u, jr2.ctg, (int32_t)jr2.kind, (int32_t)jr2.idx, ring, mode, jr2.init_len, 0});
        if (!backlog[ring].empty() || !place_job(ring, P, jr2)) backlog[ring].push_back(Backlogged{P, jr2});
        return PAG_OK;
    }
    bool is_orphan(const JobRef &jr) const { return jr.kind == 1 && jr.epoch != RS[jr.ctg].seg_epoch; }
    void flush_backlog() {
        for (uint32_t ring = 0; ring < TRAV_RINGS; ++ring)
            while (!backlog[ring].empty()) {
                if (is_orphan(backlog[ring].front().jr)) {  // (never entered a ring: gone)
                    backlog[ring].pop_front();
                    n_live -= 1;
                    continue;
                }
                if (!place_job(ring, backlog[ring].front().P, backlog[ring].front().jr)) break;
                backlog[ring].pop_front();
                need_publish = true;
            }
    }
    // The segment list of contig i is given up (the contig is finished, or its next round plans its own): the jobs of the list
    // that no wave has taken are cancelled, those that are walking finish as orphans.
    void give_up_segments(uint32_t i) {
        RoundState &R = RS[i];
        if (R.live_jobs != 0)  // (nothing of the contig is in a ring otherwise: most contigs finish that way)
            for (uint32_t ring = 0; ring < TRAV_RINGS; ++ring) {
                // the live jobs of a ring are among its last QCAP postings, none below scan_from
                const uint32_t hi = n_posted[ring], lo = std::max(scan_from[ring], hi > QCAP ? hi - QCAP : 0u);
                for (uint32_t jn = lo; jn < hi; ++jn) {
                    const uint32_t q = jn % QCAP;
                    JobRef &jr = jref[ring * QCAP + q];
                    if (jr.live && jr.ctg == i && jr.kind == 1 && jr.epoch == R.seg_epoch) {
                        __atomic_fetch_or(&hjobs[ring * QCAP + q].J.mode, (uint32_t)TRAV_MODE_CANCELLED, __ATOMIC_RELEASE);
                        ++n_orphans;
                    }
                }
            }
        R.seg_epoch += 1;
        R.segs.clear();
        R.n_spec = 0;
        R.zone_end = 0;
        R.live_jobs = 0;
        R.kept = false;
    }
    // buffers + job records of a batch of jobs of contig i (memsets and uploads go to stream s; the records become visible to
    // the walker only by publish())
    int post_batch(uint32_t i, int grp, const std::vector<JobPlan> &plans) {
        if (plans.empty()) return PAG_OK;
        const double tp0 = now_ms();
        struct PostTimer {
            double t0, *acc;
            ~PostTimer() { *acc += now_ms() - t0; }
        } post_timer{tp0, &t_st[3]};
        CtgState &cs = st[i];
        RoundState &R = RS[i];
        const uint64_t PG = TRAV_PROBE_GROUPS;
        // one travel epoch / probe stamp per vertex of the job's id range (the whole strand, or the surroundings of a
        // segment); padded to a multiple of four so that the walker's window refills can use 16-byte loads
        const size_t nj = plans.size();
        std::vector<uint64_t> o_seq(nj + 1, 0), o_oc(nj + 1, 0), o_st(nj + 1, 0), o_tb(nj + 1, 0), spans(nj, 0), o_x(nj + 1, 0);
        for (size_t j = 0; j < nj; ++j) {
            o_x[j + 1] = o_x[j] + ((plans[j].mode & TRAV_MODE_LEAP) ? plans[j].cap : 0);
            const uint32_t lo = plans[j].win_hi ? plans[j].win_lo : cs.inLo, hi = plans[j].win_hi ? plans[j].win_hi : cs.inHi;
            spans[j] = ((uint64_t)(hi - lo) + 1 + 3) & ~3ull;
            o_seq[j + 1] = o_seq[j] + plans[j].cap;
            // (a walk in the leaping zone visits vertices without a contig coordinate all the time: they live in the hash sets)
            o_oc[j + 1] = o_oc[j] + pow2_at_least((plans[j].mode & TRAV_MODE_LEAP) ? plans[j].cap + 8192 : plans[j].cap / 4 + 4096);
            o_st[j + 1] = o_st[j] + PG * spans[j];
            o_tb[j + 1] = o_tb[j] + spans[j] + 4;
        }
        // the batch's buffers come out of the walk arena (one allocation of the handle, bump pointer, reset per pag_travel:
        // a cold process otherwise spends seconds in thousands of hipMalloc calls); the per-(contig, group) slots take over
        // when the arena is used up
        DevBuf b_sv = cbuf(i, grp, CB_SEQV), b_ss = cbuf(i, grp, CB_SEQS), b_av = cbuf(i, grp, CB_ARV), b_as = cbuf(i, grp, CB_ARS),
               b_ts = cbuf(i, grp, CB_TSET), b_ps = cbuf(i, grp, CB_PSET), b_st = cbuf(i, grp, CB_STAMP), b_tb = cbuf(i, grp, CB_TBITS),
               b_sx = cbuf(i, grp, CB_SEQX);
        int r;
        {
            const size_t need[9] = {(size_t)o_seq[nj] * 4, (size_t)o_seq[nj] * 4, (size_t)o_seq[nj] * PG * 4, (size_t)o_seq[nj] * PG * 4,
                                    (size_t)o_oc[nj] * 8, (size_t)o_oc[nj] * PG * 8, (size_t)o_st[nj] * 4, (size_t)o_tb[nj] * 4, (size_t)o_x[nj] * 8};
            DevBuf *bufs[9] = {&b_sv, &b_ss, &b_av, &b_as, &b_ts, &b_ps, &b_st, &b_tb, &b_sx};
            size_t tot = 0;
            for (size_t q = 0; q < 9; ++q) tot += (need[q] + 16 + 255) & ~(size_t)255;
            if (g->walk_arena && g->walk_arena_used + tot <= g->walk_arena_cap) {
                for (size_t q = 0; q < 9; ++q) {
                    bufs[q]->p = (char *)g->walk_arena + g->walk_arena_used;
                    g->walk_arena_used += (need[q] + 16 + 255) & ~(size_t)255;
                }
            } else {
                R.slot_bufs = true;  // (per-contig slots are handed out again by the next batch of the group)
                for (size_t q = 0; q < 9; ++q)
                    if ((r = bufs[q]->alloc(need[q]))) return r;
            }
        }
        want_clear(b_sx.p, o_x[nj] * 8, 0u);
        want_clear(b_ts.p, o_oc[nj] * 8, 0xFFu);
        want_clear(b_ps.p, o_oc[nj] * PG * 8, 0u);
        want_clear(b_st.p, o_st[nj] * 4, 0u);
        want_clear(b_tb.p, o_tb[nj] * 4, 0u);
        fill_contigs();
        for (size_t j = 0; j < nj; ++j) {
            const JobPlan &pl = plans[j];
            const uint32_t ring = pl.kind == 0 ? 0u : (R.round > 1 ? 1u : 2u);
            const uint64_t cap = pl.cap, oc = o_oc[j + 1] - o_oc[j];
            TravPosted P{};
            TravJob &J = P.J;
            J.ctg = i;
            J.start = pl.start_vid;
            J.has_size = R.has_size;
            J.seq_v = b_sv.as<uint32_t>() + o_seq[j];
            J.seq_s = b_ss.as<uint32_t>() + o_seq[j];
            J.seq_cap = cap;
            J.arena_v = b_av.as<uint32_t>() + o_seq[j] * PG;
            J.arena_s = b_as.as<uint32_t>() + o_seq[j] * PG;
            J.arena_cap = PG * cap;
            J.stamp = b_st.as<uint32_t>() + o_st[j];
            J.stamp_stride = (uint32_t)spans[j];
            J.tbits = b_tb.as<uint32_t>() + o_tb[j];
            J.tset = b_ts.as<uint64_t>() + o_oc[j];
            J.tmask = (uint32_t)oc - 1;
            J.pset = b_ps.as<uint64_t>() + o_oc[j] * PG;
            J.pmask = (uint32_t)oc - 1;
            J.exact = (pl.exact || force_exact) ? 1u : 0u;
            J.mode = pl.mode;
            J.stop_pc = pl.stop_pc;
            J.init_len = 0;
            J.win_low = pl.win_low;
            J.seq_x = (pl.mode & TRAV_MODE_LEAP) ? b_sx.as<uint64_t>() + o_x[j] : nullptr;
            if (pl.mode & TRAV_MODE_RESUME) {
                const uint64_t n0 = pl.init->len;
                if (n0 == 0 || n0 > cap) {
                    set_error("pag_travel: resume job with a %llu-vertex path in a %llu-vertex buffer", (unsigned long long)n0, (unsigned long long)cap);
                    return PAG_EFAULT;
                }
                J.init_len = n0;
                // (put together in pinned memory: the copies below are asynchronous for real)
                uint32_t *flat = (uint32_t *)fetch_alloc(n0 * 8);
                if (!flat) return PAG_ENOMEM;
                flatten_chain(*pl.init, flat, flat + n0, nullptr);
                PAG_HIP_TRY(hipMemcpyAsync(J.seq_v, flat, n0 * 4, hipMemcpyHostToDevice, s));
                PAG_HIP_TRY(hipMemcpyAsync(J.seq_s, flat + n0, n0 * 4, hipMemcpyHostToDevice, s));
            }
            P.C = tc[i];
            if (pl.win_hi) {  // a segment job: direct-mapped marks only around the segment
                P.C.in_lo = pl.win_lo;
                P.C.in_hi = pl.win_hi;
            }
            JobRef jr2;
            jr2.ctg = i;
            jr2.kind = pl.kind;
            jr2.idx = pl.idx;
            jr2.init_len = J.init_len;
            jr2.live = true;
            jr2.epoch = R.seg_epoch;
            if (pl.kind == 0) {
                if (pl.mode & TRAV_MODE_RESUME) ++n_resume_jobs;
            } else {
                ++n_seg_jobs;
                if (pl.mode & TRAV_MODE_LEAP) ++n_leap_jobs;
            }
            if (defer_ring2 && ring == 2u) {  // (first rounds before the walker starts: the ring order is decided later)
                deferred[i].push_back(Deferred{P, jr2});
                continue;
            }
            int r2;
            if ((r2 = commit_job(ring, P, jr2, pl.mode, pl.stop_pc))) return r2;
        }
        need_publish = true;
        return PAG_OK;
    }
    int publish() {  // after the prepared buffers are ready on the device
        if (!need_publish) return PAG_OK;
        auto tmark = [&](const char *what) {
            if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 4u, n_live, 0, 0, (uint64_t)(uintptr_t)what, 0, clears.size(), 0});
        };
        tmark("publish: begin");
        int rcl;
        if ((rcl = flush_clears())) return rcl;
        tmark("publish: clears launched");
        PAG_HIP_TRY(hipStreamSynchronize(s));
        tmark("publish: stream idle");
        for (uint32_t r = NR; r-- > 0;) __atomic_store_n(&hq->posted[r], n_posted[r], __ATOMIC_RELEASE);
        need_publish = false;
        const int rcw = walkers.g ? walkers.ensure(n_live) : PAG_OK;  // (before the first launch: pag_travel starts the waves itself)
        tmark("publish: waves");
        return rcw;
    }

    // ---- start of a round of contig i: its seeds are in cs.seeds.  Decides where the walk can be cut, finds the checkpoint
    //      vertices and posts the seed jobs and the segment jobs.
    // stop coordinate of a job that walks up to segment q of the round (its checkpoint + the overlap)
    uint32_t first_stop(const stitch::RoundState &R) const { return stitch::stop_for(R, 0, seg_ov); }
    // The rounds of several contigs are prepared together: their checkpoint vertices come from ONE launch of k_checkpoints and
    // the id ranges around their segments from ONE launch of k_id_bounds (two synchronisations per call; contig by contig
    // the 48 first rounds of configs[1] were ~100 small launches and synchronisations, ~10 ms before the first job).
    struct RoundPlan {
        std::vector<uint32_t> ck_x;
        size_t n_spec_ck = 0;
        uint32_t x0 = 0xFFFFFFFFu, seed_lo = 0, seed_hi = 0;
        size_t req_off = 0, co_off = 0;
        bool has_co = false;
        bool kept = false;        // the round adopts the segments of an earlier round (RoundState::kept): none are planned
        uint32_t kept_stop = 0;   // ... and its seeds walk up to this coordinate (0: to the end)
    };
    // ---- start of a round of contigs `which` (their seeds are in cs.seeds): four steps
    // (1) per contig: the round's state, the checkpoint coordinates of its segments (reqs: the checkpoint searches)
    void plan_rounds(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, std::vector<TravSeedReq> &reqs) {
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            std::vector<uint32_t> &ck_x = P.ck_x;
            size_t &n_spec_ck = P.n_spec_ck;
            uint32_t &x0 = P.x0;
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            const bool keep = R.kept && !R.segs.empty();
            P.kept = keep;
            R.round += 1;
            R.active = true;
            if (!keep) {
                R.segs.clear();
                R.n_spec = 0;
                R.zone_end = 0;
                R.live_jobs = 0;
                R.slot_bufs = false;
            }
            R.chains.assign(cs.seeds.size(), Chain{});
            R.has_size = (uint64_t)cs.varLen;  // int64 -> size_t conversion as in the reference call
            rounds = std::max<uint64_t>(rounds, R.round);
            const uint64_t split = (uint64_t)(cs.len * startSplit);
            // where leaping becomes possible: hasSize + nowSize >= split, nowSize = k + the steps walked.  The steps follow the
            // contig coordinate closely but not exactly, so the zone is left with a margin; WHERE the walk is cut only decides how
            // much of it runs in parallel, every adoption is checked against the true sizes (try_merge).
            for (auto &sd : cs.seeds) x0 = std::min(x0, sd.ctg);
            if (keep) {
                // where the seeds' own walks stop: a little into the first kept segment ahead of them, of the kind a chain
                // at their coordinate adopts (advance_chain) — or of the other kind when none of that kind lies ahead
                const bool can = R.has_size + k >= split;
                const int n_spec = (int)R.n_spec, n_all = (int)R.segs.size();
                auto first_ahead = [&](int lo, int hi) -> int {
                    for (int q = lo; q < hi; ++q)
                        if (R.segs[(size_t)q].x > x0 && (R.segs[(size_t)q].leap || x0 < R.zone_end)) return q;
                    return -1;
                };
                int q = can ? first_ahead(n_spec, n_all) : first_ahead(0, n_spec);
                if (q < 0) q = can ? first_ahead(0, n_spec) : first_ahead(n_spec, n_all);
                P.kept_stop = q >= 0 ? stitch::stop_for(R, (size_t)q, seg_ov) : 0u;
            } else if (use_pieces && cs.varLen >= 0 && x0 >= cs.ctgLeft && x0 < cs.ctgRight) {
                const uint64_t H = (uint64_t)cs.varLen + k;
                // (measured at BASELINE configs[1]: segments of 10-20 kb with 1.5 kb of overlap are the optimum, a few thousand
                // jobs; shorter ones pay more overlap and job start-up, longer ones lengthen the first piece of every chain)
                const uint64_t seg_len = seg_len_env ? seg_len_env : 12000;
                // The two kinds of segments OVERLAP around the coordinate where leaping becomes possible (x0 + split - H if the steps
                // followed the coordinate exactly; they do not quite: `margin` on either side).  A chain adopts segments that
                // cannot leap up to where its true size allows (try_merge cuts the adoption there), crosses the point with a short
                // exact walk (TRAV_MODE_UNTIL_LEAP) and goes on with the pieces of the leaping zone that were started before the
                // point.  (Until round 3 the kinds were kept apart by the margins and every contig walked the ~10 kb between them
                // exactly, 30-48 ms at the end of its round.)  Only decides how much is walked in parallel: every adoption is
                // checked against the true sizes.
                const uint64_t margin = cfg.seg_safety_set ? cfg.seg_safety : cs.len / 400 + 200;
                if (split > H + seg_len) {
                    const uint64_t zone = std::min<uint64_t>((uint64_t)x0 + (split - H) + margin, (uint64_t)cs.ctgRight - 1);
                    for (uint64_t x = (uint64_t)x0 + seg_len; x + seg_ov + seg_len / 4 < zone; x += seg_len) ck_x.push_back((uint32_t)x);
                    if (!ck_x.empty()) R.zone_end = (uint32_t)zone;
                }
                n_spec_ck = ck_x.size();
                if (use_leap_pieces) {
                    // the leaping zone gets segments of its own (TRAV_MODE_LEAP), from where the real walk has certainly begun to
                    // leap (the steps follow the coordinate closely, not exactly: a margin; every adoption is checked with the true
                    // size) to the end of the strand
                    // (a walk there makes three times the classifications per vertex: shorter pieces for the same job length)
                    const uint64_t lseg = std::max<uint64_t>(seg_len / 2, seg_ov * 2);
                    // (how far before x0 + split - H the first piece starts: at configs[1] the steps of a path add up to 0.6 % more than
                    // the coordinates it covers — leaping begins ~7 kb earlier than the coordinate says on a 1.2 Mb contig; pieces
                    // started too early cost a few jobs, pieces started too late an exact walk on the contig's critical path)
                    const uint64_t left = cfg.seg_safety_set ? cfg.seg_safety : cs.len / 64 + 500;
                    const uint64_t first = (uint64_t)x0 + (split > H + left + lseg ? split - H - left : lseg);
                    // (the last stretch of the strand in shorter pieces still: the job that reaches the end of the strand is the
                    // last one of its round, and a contig that needs a second round waits for it twice)
                    const uint64_t end_div = 2;
                    const uint64_t end_zone = (uint64_t)cs.ctgRight > 2 * lseg ? (uint64_t)cs.ctgRight - 2 * lseg : 0;
                    for (uint64_t x = std::max<uint64_t>(first, (uint64_t)x0 + lseg); x + lseg / 4 / end_div < (uint64_t)cs.ctgRight - 1; x += (x >= end_zone ? std::max<uint64_t>(lseg / end_div, seg_ov) : lseg))
                        if (ck_x.size() == n_spec_ck || x > (uint64_t)ck_x.back() + lseg / 4 / end_div) ck_x.push_back((uint32_t)x);
                }
            }
            P.req_off = reqs.size();
            for (size_t q = 0; q < ck_x.size(); ++q) {
                const uint64_t off = ck_x[q] - cs.ctgLeft;
                TravSeedReq rq;
                rq.ctg = i;
                rq.pad = 0;
                rq.pos = off;
                rq.left = off - std::min<uint64_t>(off, 64);
                rq.right = off + 64;
                reqs.push_back(rq);
            }
        }
    }
    // (2) the checkpoint vertices of all of them: one launch, one round trip
    int find_checkpoints(const std::vector<TravSeedReq> &reqs, std::vector<uint32_t> &out) {
        out.assign(reqs.size() * 3, 0u);
        int r;
        if (!reqs.empty()) {
            if ((r = b_ckreq.alloc(reqs.size() * sizeof(TravSeedReq))) || (r = b_ckout.alloc(reqs.size() * 12))) return r;
            if ((r = upload_contigs())) return r;
            PAG_HIP_TRY(hipMemcpyAsync(b_ckreq.p, reqs.data(), reqs.size() * sizeof(TravSeedReq), hipMemcpyHostToDevice, s));
            trav_launch_checkpoints(G, b_tc.as<TravContig>(), b_ckreq.as<TravSeedReq>(), (uint32_t)reqs.size(), deviation, b_ckout.as<uint32_t>(), s);
            PAG_HIP_TRY(hipMemcpyAsync(out.data(), b_ckout.p, out.size() * 4, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
        return PAG_OK;
    }
    // (3) per contig: its segments, and the contig coordinates their id ranges are asked for (co)
    void make_segments(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, const std::vector<uint32_t> &out, std::vector<uint32_t> &co) {
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            const std::vector<uint32_t> &ck_x = P.ck_x;
            const size_t n_spec_ck = P.n_spec_ck;
            const uint32_t x0 = P.x0;
            if (P.kept) {  // (the id range around the seeds' own first piece: [lowest seed - 2000, its stop + 3000])
                if (P.kept_stop != 0u) {
                    P.has_co = true;
                    P.co_off = co.size();
                    co.push_back((uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)x0 - std::min<uint64_t>(x0, 2000)));
                    co.push_back((uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)P.kept_stop + 3000));
                }
                continue;
            }
            if (ck_x.empty()) continue;
            const uint32_t *out_c = out.data() + 3 * P.req_off;
            for (size_t q = 0; q < ck_x.size(); ++q) {
                if (out_c[3 * q] == PAG_NONE) continue;
                Seg sg;
                sg.x = out_c[3 * q + 1];
                sg.vid = out_c[3 * q];
                sg.leap = q >= n_spec_ck;
                sg.win_low = x0;
                sg.round = R.round;
                if (!R.segs.empty() && R.segs.back().leap == sg.leap && sg.x <= R.segs.back().x) continue;  // (increasing within a kind)
                R.segs.push_back(std::move(sg));
            }
            R.n_spec = 0;
            for (auto &sg : R.segs) R.n_spec += sg.leap ? 0 : 1;
            for (size_t q = 0; q < R.segs.size(); ++q) {
                const bool more = q + 1 < R.segs.size();
                if (R.segs[q].leap) R.segs[q].stop = more ? (uint32_t)std::min<uint64_t>((uint64_t)R.segs[q + 1].x + seg_ov, 0xFFFFFFFFull) : 0u;  // 0: to the end
                else R.segs[q].stop = more && !R.segs[q + 1].leap ? (uint32_t)std::min<uint64_t>((uint64_t)R.segs[q + 1].x + seg_ov, R.zone_end) : R.zone_end;
            }
            {
                bool any_spec = false;
                for (auto &sg : R.segs) any_spec = any_spec || !sg.leap;
                if (!any_spec) R.zone_end = 0;
            }
            if (!R.segs.empty()) {  // id ranges around the segments: [checkpoint - 2000, stop + 3000] in contig coordinates
                const size_t nq = R.segs.size();
                P.has_co = true;
                P.co_off = co.size();
                co.resize(co.size() + 2 * nq + 2);
                uint32_t *cc = co.data() + P.co_off;
                for (size_t q = 0; q < nq; ++q) {
                    cc[2 * q] = (uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)R.segs[q].x - std::min<uint64_t>(R.segs[q].x, 2000));
                    cc[2 * q + 1] = R.segs[q].stop ? (uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)R.segs[q].stop + 3000) : cs.ctgRight;
                }
                // ... and around the seeds' own first piece: [lowest seed - 2000, first stop + 3000]
                cc[2 * nq] = (uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)x0 - std::min<uint64_t>(x0, 2000));
                cc[2 * nq + 1] = (uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)first_stop(R) + 3000);
            }
        }
    }
    int find_id_bounds(const std::vector<uint32_t> &co, std::vector<uint32_t> &ids) {
        ids.assign(co.size(), 0u);
        int r;
        if (!co.empty()) {
            if ((r = b_ckreq.alloc(co.size() * 4)) || (r = b_ckout.alloc(co.size() * 4))) return r;
            PAG_HIP_TRY(hipMemcpyAsync(b_ckreq.p, co.data(), co.size() * 4, hipMemcpyHostToDevice, s));
            trav_launch_id_bounds(G, b_ckreq.as<uint32_t>(), (uint32_t)co.size(), b_ckout.as<uint32_t>(), s);
            PAG_HIP_TRY(hipMemcpyAsync(ids.data(), b_ckout.p, ids.size() * 4, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
        return PAG_OK;
    }
    // (4) per contig: the id ranges, the jobs
    int post_round_jobs(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, const std::vector<uint32_t> &ids) {
        int r;
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            uint32_t seed_lo = 0, seed_hi = 0;  // id range for the walks of the seeds up to the first checkpoint (0, 0: the strand)
            if (P.has_co) {
                const size_t nq = P.kept ? 0 : R.segs.size();
                const uint32_t *idc = ids.data() + P.co_off;
                auto window = [&](size_t q, uint32_t *wlo, uint32_t *whi) {
                    uint32_t lo = std::max(idc[2 * q], cs.inLo), hi = std::min(idc[2 * q + 1], cs.inHi);
                    lo = cs.inLo + ((lo - cs.inLo) & ~31u);  // (the strand's global-visited bitmap is read word-wise from here)
                    if (hi <= lo) hi = std::min<uint32_t>(cs.inHi, lo + 64);
                    *wlo = lo;
                    *whi = hi;
                };
                for (size_t q = 0; q < nq; ++q) window(q, &R.segs[q].win_lo, &R.segs[q].win_hi);
                if (P.kept || first_stop(R) != 0u) window(nq, &seed_lo, &seed_hi);
            }
            std::vector<JobPlan> plans;
            const uint64_t cap_full = cs.seqCap;
            for (size_t sd = 0; sd < cs.seeds.size(); ++sd) {
                const uint32_t stop = P.kept ? P.kept_stop : (R.segs.empty() ? 0u : first_stop(R));
                // (a seed's walk that stops at the first checkpoint is a piece like the segments: direct-mapped marks around it,
                // a sequence buffer for its stretch; the full-strand arrays, 130 MB per job at configs[1], are for resumed walks)
                JobPlan pl{0, (int)sd, cap_full, cs.seeds[sd].vid, 0u, stop, nullptr, false};
                if (stop != 0u && seed_hi != 0u && cs.seeds[sd].ctg <= stop) {
                    pl.win_lo = seed_lo;
                    pl.win_hi = seed_hi;
                    pl.cap = std::min<uint64_t>(cap_full, ((uint64_t)stop - cs.seeds[sd].ctg) / 2 + 8192);
                }
                plans.push_back(pl);
            }
            // (the segments of the leaping zone first: they are the slowest, three times the classifications per vertex)
            // ... and of those the piece that runs to the end of the strand FIRST: it walks on from there until it leaps (at
            // configs[1] ~5 000 vertices and 20 000 classifications where the other pieces have 1 700 and 6 000: 55-70 ms, the
            // longest job of its contig by far and the one its round waits for; tests/walk_trace.py showed a fifth of them
            // starting 12-14 ms into the walks)
            std::vector<size_t> seg_order;
            for (size_t q = R.segs.size(); q-- > 0;)
                if (R.segs[q].leap && R.segs[q].stop == 0u) {
                    seg_order.push_back(q);
                    break;
                }
            for (size_t q = 0; q < R.segs.size(); ++q)
                if (seg_order.empty() || q != seg_order[0]) seg_order.push_back(q);
            for (int pass = 0; pass < 2 && !P.kept; ++pass)  // (kept segments have their jobs, or their paths, already)
                for (size_t q : seg_order) {
                    if (R.segs[q].leap != (pass == 0)) continue;
                    const uint64_t spanc = (R.segs[q].stop ? (uint64_t)R.segs[q].stop : (uint64_t)cs.ctgRight) - R.segs[q].x;
                    const uint64_t cap = std::min<uint64_t>(cap_full, spanc / 2 + 8192);
                    JobPlan pl{1, (int)q, cap, R.segs[q].vid, (uint32_t)(R.segs[q].leap ? TRAV_MODE_LEAP : TRAV_MODE_SPEC), R.segs[q].stop, nullptr, false, R.segs[q].win_lo, R.segs[q].win_hi};
                    pl.win_low = R.segs[q].leap ? R.segs[q].win_low : 0u;
                    plans.push_back(pl);
                }
            if (wdebug)
                std::fprintf(stderr, "[walk] t=%.1f ms contig %u round %u: %zu seeds, %zu segments%s, cut zone ends at %u (strand %u..%u)\n", now_ms() - t_walk0, i,
                             R.round, cs.seeds.size(), R.segs.size(), P.kept ? " kept from an earlier round" : "", R.zone_end, cs.ctgLeft, cs.ctgRight);
            if ((r = post_batch(i, GRP_ROUND, plans))) return r;
        }
        return PAG_OK;
    }
    int start_rounds(const std::vector<uint32_t> &which) {
        std::vector<RoundPlan> RP(which.size());
        std::vector<TravSeedReq> reqs;
        std::vector<uint32_t> out, co, ids;
        int r;
        plan_rounds(which, RP, reqs);
        if ((r = find_checkpoints(reqs, out))) return r;
        make_segments(which, RP, out, co);
        if ((r = find_id_bounds(co, ids))) return r;
        if ((r = post_round_jobs(which, RP, ids))) return r;
        return flush_clears();  // (one launch for the buffers of all these rounds, under way while this thread goes on)
    }

    // continue chain c of contig i exactly: the path so far goes to the walker as a RESUME job
    // (until_leap: only as far as the first iteration boundary from which the walk can leap, TRAV_MODE_UNTIL_LEAP)
    int post_resume(uint32_t i, int c, uint32_t stop, bool until_leap = false) {
        CtgState &cs = st[i];
        Chain &ch = RS[i].chains[(size_t)c];
        const uint64_t cap = std::max<uint64_t>(cs.seqCap * ch.grow, ch.len + cs.seqCap / 4 + 4096);
        std::vector<JobPlan> plans{JobPlan{0, c, cap, cs.seeds[(size_t)c].vid, (uint32_t)(TRAV_MODE_RESUME | (until_leap ? TRAV_MODE_UNTIL_LEAP : 0)), stop, &ch, ch.exact}};
        return post_batch(i, GRP_CHAIN0 + c, plans);  // (its buffers are cleared with those of the other resumed walks of this turn: stitch_finished)
    }

    // adoption of a finished segment by a chain (conditions and their justification: walk_stitch.hpp)
    stitch::MergeCtx merge_ctx(uint32_t i) {
        MergeCtx M;
        M.k = k;
        M.deviation = deviation;
        M.split = (uint64_t)(st[i].len * startSplit);
        M.has_size = RS[i].has_size;
        M.round = RS[i].round;
        if (st[i].committed) {
            M.g_lo = st[i].gwinLo;
            M.g_hi = st[i].gwinHi;
            M.g_free_hi = st[i].gFreeHi;
        }
        return M;
    }
    // totals of what the chains adopted (walk_stitch.hpp advance_chain, called from stitch_finished)
    stitch::AdvanceStats adv_stats;

    int fail(int rc2) {
        shutdown_walker();
        g->defer_free = false;
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
        // deliveries made while the walks ran (gather kernels writing pinned chunks the next call reuses) must have landed,
        // and nothing of a failed call may be handed out as a path
        if (g->deliver_stream) hipStreamSynchronize(g->deliver_stream);
        hipStreamSynchronize(s);
        std::fill(g->path_valid.begin(), g->path_valid.end(), (uint8_t)0);
        std::fill(g->path_ptr.begin(), g->path_ptr.end(), nullptr);
        return rc2;
    }

    // filterSequence / "Pump it" of a finished contig (PAlgorithm.cpp:409-423)
    bool pumped(const CtgState &cs, uint32_t last_ctg) {  // the last vertex of a path that ends in a leap is dropped?
        auto d = mapper.singleToDual(last_ctg);
        uint64_t a = (uint64_t)std::llabs(d.first);
        return a == (uint64_t)cs.ci + 1 || (a >= 1 && a <= mapper.sizes.size() && (double)d.second >= (double)mapper.sizes[a - 1] * (1 - startSplit));
    }
    void filter_travel(CtgState &cs) {
        auto &seq = cs.travel;
        if (!cs.finalLeap) {
            const size_t windowSize = 10;
            if (seq.size() >= windowSize) {
                size_t startIdx = seq.size() - seq.size() / 90;
                for (size_t i = startIdx; i < seq.size() - windowSize + 1; ++i) {
                    uint32_t firstPos = seq[i].ctg;
                    uint32_t secondPos = seq[std::min(seq.size(), i + windowSize) - 1].ctg;
                    if (secondPos != 0 && firstPos != 0 && secondPos < firstPos) {
                        seq.resize(i + 1);
                        break;
                    }
                }
            }
        } else if (!seq.empty()) {
            if (pumped(cs, seq.back().ctg)) seq.pop_back();
        }
    }
    // A contig whose traversal is over is DELIVERED while the others still walk: its sequence is filtered, the full records of
    // its vertices are gathered on the device and copied (asynchronously, stream s) into pinned memory that lives until the
    // next call — at configs[1] the one gather + 380 MB copy for all contigs used to follow the last walk (15 ms).
    // Device buffers from the walk arena; without room there the contig is left to the epilogue.
    // (a delivery issued while walk jobs are live runs on 24 blocks: its thousands of waves, each with stores to host memory in
    // flight, slowed every walker wave beside them — 2.5 -> 3.2-5 us per classification in the last 40 ms of a block, round 5)
    static constexpr unsigned DELIVER_BLOCKS = 24;
    int deliver_contig(uint32_t i) {
        CtgState &cs = st[i];
        if (cs.delivered || !cs.done) return PAG_OK;
        if (cs.tail.on) {  // (a path that ends in a leap: finalLeap, nothing but the last vertex to filter)
            const CtgState::DevTail &T = cs.tail;
            const size_t m0 = T.m0, m = m0 + T.n - (pumped(cs, T.last_ctg) ? 1 : 0);
            const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
            cs.delivered = true;
            g->path_off[slot2] = 0;
            g->path_len[slot2] = m;
            g->path_valid[slot2] = 1;
            if (m == 0) return PAG_OK;
            pag_path_node *dst = (pag_path_node *)fetch_alloc(m * sizeof(pag_path_node));
            if (!dst) return PAG_ENOMEM;
            if (m0) {
                uint32_t *hp = (uint32_t *)fetch_alloc(m0 * 8);
                if (!hp) return PAG_ENOMEM;
                for (size_t x = 0; x < m0; ++x) {
                    hp[x] = cs.travel[x].u;
                    hp[m0 + x] = (uint32_t)cs.travel[x].step;
                }
                PAG_HIP_TRY(hipMemcpyAsync(T.d_ids, hp, m0 * 4, hipMemcpyHostToDevice, g->deliver_stream));
                PAG_HIP_TRY(hipMemcpyAsync(T.d_ids + T.cap, hp + m0, m0 * 4, hipMemcpyHostToDevice, g->deliver_stream));
            }
            trav_launch_gather_path(G, T.d_ids, T.d_ids + T.cap, m, dst, g->deliver_stream, n_live ? DELIVER_BLOCKS : 0u);
            g->path_ptr[slot2] = dst;
            return PAG_OK;
        }
        const size_t n = cs.travel.size();
        const size_t need = ((n * 8 + 255) & ~(size_t)255) + 512;
        if (!g->walk_arena || g->walk_arena_used + need > g->walk_arena_cap) return PAG_OK;
        filter_travel(cs);
        const size_t m = cs.travel.size();
        const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
        cs.delivered = true;
        g->path_off[slot2] = 0;
        g->path_len[slot2] = m;
        g->path_valid[slot2] = 1;
        if (m == 0) return PAG_OK;
        uint32_t *hp = (uint32_t *)fetch_alloc(m * 8);
        pag_path_node *dst = (pag_path_node *)fetch_alloc(m * sizeof(pag_path_node));
        if (!hp || !dst) return PAG_ENOMEM;
        for (size_t x = 0; x < m; ++x) {
            hp[x] = cs.travel[x].u;
            hp[m + x] = (uint32_t)cs.travel[x].step;
        }
        uint32_t *d_ids = (uint32_t *)((char *)g->walk_arena + g->walk_arena_used);
        g->walk_arena_used += (m * 8 + 255) & ~(size_t)255;
        // A stream of its own (behind this work on stream s the fetches of finished jobs would wait), and the gather kernel
        // writes the records straight into the pinned host array: a device-to-host copy of 32 bytes per vertex would
        // occupy the copy engine the fetches need (measured: their lap 9 -> 24 ms per step).
        if (!g->deliver_stream) PAG_HIP_TRY(hipStreamCreateWithFlags(&g->deliver_stream, hipStreamNonBlocking));
        PAG_HIP_TRY(hipMemcpyAsync(d_ids, hp, m * 8, hipMemcpyHostToDevice, g->deliver_stream));
        trav_launch_gather_path(G, d_ids, d_ids + m, m, dst, g->deliver_stream, n_live ? DELIVER_BLOCKS : 0u);
        g->path_ptr[slot2] = dst;
        return PAG_OK;
    }

    // ---- the event loop
    uint32_t scan_from[TRAV_RINGS] = {0, 0, 0};  // per ring: every job number below it has been handled
    double t_progress = 0, t_first_fin = 0;
    // Waiting for the walker: a busy wait (pause instructions), not a sleep — on a loaded host a 20 us sleep comes back after
    // a millisecond or more, and every finished job that waits for this thread holds up the jobs that depend on it.  Only
    // after 5 ms without any news does the thread start yielding its time slice.
    double t_last_news = 0;
    void idle_wait(double us) {
        const double t0w = now_ms();
        if (t0w - t_last_news > 5.0) {
            std::this_thread::sleep_for(std::chrono::microseconds((long)us));
            return;
        }
        while ((now_ms() - t0w) * 1000.0 < us) {
            for (int q = 0; q < 32; ++q) __builtin_ia32_pause();
        }
    }
    // Contigs whose round is decided and not yet chosen / spliced / re-seeded.  While jobs are in flight they are taken a few at
    // a time, those that go on to another round first: the copy of a finished contig's walk (hundreds of thousands of vertices
    // out of pinned memory) keeps this thread — the one every chain waits for — away from the jobs that finish meanwhile; in
    // the last third of the walks, when the contigs that leapt finish in batches of a dozen, a job of a contig still walking
    // used to wait 10 - 15 ms for its turn.
    std::vector<uint32_t> over_queue;

    // ---- the steps of a call, in the order run() takes them
    // the traversal view: compact CSR, coordinate order, successor records (once per built graph)
    int begin() {
        PAG_HIP_TRY(hipSetDevice(g->device));
        s = g->stream;
        t_begin = now_ms();
        timing = cfg.timing;
        wdebug = cfg.walk_debug;
        wtrace = cfg.walk_trace;
        lap_t = t_begin;
        k = g->k;
        deviation = prm->deviation;
        errorRate = prm->error_rate;
        startSplit = prm->start_split;
        topK = std::min<uint32_t>(prm->ref_threads, 8u);
        int rc;
        if ((rc = trav_prepare_graph(g, ctgs->len, ctgs->n_seqs, ref_len, n_refs, deviation, errorRate, &G, &t_compact, orient, startSplit))) return rc;
        slot += TRAV_GRAPH_SLOTS + TRAV_EXTRA_SLOTS;
        lap("compact");
        return PAG_OK;
    }
    // contigs: packed bases, mapper tables, per-strand node tables, id ranges, global visited structures
    int setup_contigs() {
        int rc;
        n_ctgs = (uint32_t)ctgs->n_seqs;
        g->path_off.assign(2 * (size_t)n_ctgs, 0);
        g->path_len.assign(2 * (size_t)n_ctgs, 0);
        g->path_valid.assign(2 * (size_t)n_ctgs, 0);
        g->path_ptr.assign(2 * (size_t)n_ctgs, nullptr);
        // one entry per (contig, orientation): a contig selected with both orientations is two independent traversals
        // (PAssembly.cpp:28-36 walks every (name, forward) pair of its set)
        for (uint32_t c2 = 0; c2 < 2 * n_ctgs; ++c2) {
            const uint32_t c = c2 >> 1;
            const bool fwd = (c2 & 1u) == 0;
            const int32_t o = orient[c];
            if (!(o == PAG_ORIENT_BOTH || (fwd && o == PAG_ORIENT_FORWARD) || (!fwd && o == PAG_ORIENT_REVERSE))) continue;
            CtgState cs;
            cs.ci = c;
            cs.forward = fwd;
            cs.chosenOne = cs.forward ? (int64_t)c + 1 : -(int64_t)c - 1;
            cs.len = ctgs->len[c];
            cs.ctgLeft = (uint32_t)mapper.dualToSingle(cs.chosenOne, 0);
            cs.ctgRight = (uint32_t)mapper.dualToSingle(cs.chosenOne, cs.len);
            cs.revLeft = (uint32_t)mapper.dualToSingle(-cs.chosenOne, 0);
            cs.revRight = (uint32_t)mapper.dualToSingle(-cs.chosenOne, cs.len);
            cs.nodesOff = nodes_total;
            cs.seqCap = (uint64_t)cs.len / 2 + 8192;
            if (cfg.debug_seqcap) cs.seqCap = (uint64_t)cfg.debug_seqcap;  // tests: force the overflow / regrow path
            nodes_total += cs.len >= k ? cs.len - k + 1 : 0;
            st.push_back(std::move(cs));
        }
        n_sel = (uint32_t)st.size();
        if (n_sel == 0) return PAG_OK;

        b_packed = buf(), b_nodes = buf(), b_starts = buf(), b_sizes = buf(), b_tc = buf(), b_seedout = buf(), b_req = buf();
        b_gset = buf(), b_gather = buf(), b_vids = buf(), b_gbits = buf();
        if ((rc = b_packed.alloc(ctgs->packed_bytes + 64)) || (rc = b_nodes.alloc((nodes_total + 1) * 4)) ||
            (rc = b_starts.alloc(mapper.starts.size() * 8 + 8)) || (rc = b_sizes.alloc(mapper.sizes.size() * 8 + 8)) ||
            (rc = b_tc.alloc(n_sel * sizeof(TravContig))))
            return rc;
        PAG_HIP_TRY(hipMemcpyAsync(b_packed.p, ctgs->packed, ctgs->packed_bytes, hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(b_starts.p, mapper.starts.data(), mapper.starts.size() * 8, hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(b_sizes.p, mapper.sizes.data(), mapper.sizes.size() * 8, hipMemcpyHostToDevice, s));
        {   // the strands' node tables: one launch
            std::vector<TravCtgNodesJob> cj;
            uint32_t max_len = 0;
            for (auto &cs : st) {
                cj.push_back(TravCtgNodesJob{ctgs->byte_off[cs.ci], cs.nodesOff, (uint32_t)cs.len, cs.forward ? 1 : 0});
                max_len = std::max<uint32_t>(max_len, (uint32_t)cs.len);
            }
            DevBuf b_cj = buf();
            if ((rc = b_cj.alloc(cj.size() * sizeof(TravCtgNodesJob)))) return rc;
            PAG_HIP_TRY(hipMemcpyAsync(b_cj.p, cj.data(), cj.size() * sizeof(TravCtgNodesJob), hipMemcpyHostToDevice, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));  // (cj is a local)
            trav_launch_ctg_nodes(b_packed.as<uint8_t>(), b_cj.as<TravCtgNodesJob>(), (uint32_t)cj.size(), max_len, k, G, b_nodes.as<uint32_t>(), s);
        }

        tc.assign(n_sel, TravContig{});
        // id ranges of the strands, then the per-contig global visited structures
        {
            for (auto &cs : st) cs.gcap = 1024;  // placeholder so that gmask is well formed
            if ((rc = upload_contigs())) return rc;
            trav_launch_ranges(G, b_tc.as<TravContig>(), n_sel, s);
            PAG_HIP_TRY(hipMemcpyAsync(tc.data(), b_tc.p, n_sel * sizeof(TravContig), hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
            uint64_t tot_set = 0, tot_bits = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                CtgState &cs = st[i];
                cs.inLo = tc[i].in_lo;
                cs.inHi = tc[i].in_hi;
                cs.gcap = (uint32_t)pow2_at_least(cs.seqCap / 2 + 8192);
                tot_set += cs.gcap;
                tot_bits += ((uint64_t)(cs.inHi - cs.inLo) + 31) / 32 + 1;
            }
            if ((rc = b_gset.alloc(tot_set * 4)) || (rc = b_gbits.alloc(tot_bits * 4))) return rc;
            PAG_HIP_TRY(hipMemsetAsync(b_gset.p, 0xFF, tot_set * 4, s));
            PAG_HIP_TRY(hipMemsetAsync(b_gbits.p, 0, tot_bits * 4, s));
            uint64_t o1 = 0, o2 = 0;
            for (auto &cs : st) {
                cs.gset = b_gset.as<uint32_t>() + o1;
                o1 += cs.gcap;
                cs.gbits = b_gbits.as<uint32_t>() + o2;
                o2 += ((uint64_t)(cs.inHi - cs.inLo) + 31) / 32 + 1;
            }
        }

        lap("contig tables");
        return PAG_OK;
    }
    // round 0 seeds: searchPANode(onlyFirst) then top-K
    int first_seeds() {
        int rc;
        if ((rc = b_seedout.alloc((uint64_t)n_sel * SEED_STRIDE * 4))) return rc;
        if ((rc = upload_contigs())) return rc;
        trav_launch_seed_first(G, b_tc.as<TravContig>(), n_sel, deviation, b_seedout.as<uint32_t>(), SEED_STRIDE, s);
        std::vector<uint32_t> seedbuf((size_t)n_sel * SEED_STRIDE);
        PAG_HIP_TRY(hipMemcpyAsync(seedbuf.data(), b_seedout.p, seedbuf.size() * 4, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        {
            std::vector<uint32_t> vids;
            std::vector<size_t> cnt(n_sel);
            for (uint32_t i = 0; i < n_sel; ++i) {
                const uint32_t *o = &seedbuf[(size_t)i * SEED_STRIDE];
                size_t n = std::min<size_t>(std::min<size_t>(o[0], (SEED_STRIDE - 2) / 2), topK);
                cnt[i] = n;
                for (size_t j = 0; j < n; ++j) vids.push_back(o[1 + 2 * j]);
            }
            std::vector<pag_path_node> attrs;
            if ((rc = fetch_vertices(vids, attrs))) return rc;
            size_t at = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                st[i].seeds.assign(attrs.begin() + at, attrs.begin() + at + cnt[i]);
                at += cnt[i];
                if (st[i].seeds.empty()) st[i].done = true;
            }
        }

        lap("first seeds");
        return PAG_OK;
    }
    // the rings of job records (host memory the walker reads), the switches of the pieces
    int setup_rings() {
        if (g->cpool.size() < (size_t)n_sel * GROUPS * CB_N) g->cpool.resize((size_t)n_sel * GROUPS * CB_N);
        {
            const uint64_t sl = std::max<uint64_t>(128, cfg.seg_len ? cfg.seg_len : 12000);
            const uint64_t ll = std::max<uint64_t>(128, sl / 2);
            uint64_t est = 0;
            for (uint32_t i = 0; i < n_sel; ++i) est += (uint64_t)st[i].len / sl + (uint64_t)st[i].len / ll + 32;  // (every strand as if all of it were both zones)
            while (QCAP < 2 * est && QCAP < (1u << 24)) QCAP *= 2;
            if (cfg.debug_ring) QCAP = (uint32_t)cfg.debug_ring;  // tests: a ring far smaller than a round
        }
        const size_t q_need = 256 + NR * (size_t)QCAP * (sizeof(TravPosted) + sizeof(TravJobOut) + sizeof(uint32_t)) + 256;
        if (g->wq_bytes < q_need) {
            if (g->wq_host) hipHostFree(g->wq_host);
            g->wq_host = nullptr;
            g->wq_bytes = 0;
            PAG_HIP_TRY(hipHostMalloc(&g->wq_host, q_need, hipHostMallocCoherent | hipHostMallocMapped));
            g->wq_bytes = q_need;
        }
        if (!g->wq_next) PAG_HIP_TRY(hipMalloc((void **)&g->wq_next, 256));
        hq = (TravQueue *)g->wq_host;
        hjobs = (TravPosted *)((char *)g->wq_host + 256);
        houts = (TravJobOut *)(hjobs + NR * (size_t)QCAP);
        hdone = (uint32_t *)(houts + NR * (size_t)QCAP);
        std::memset(g->wq_host, 0, 256);
        std::memset(hdone, 0, NR * (size_t)QCAP * sizeof(uint32_t));
        PAG_HIP_TRY(hipMemsetAsync(g->wq_next, 0, 256, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        t_walk0 = now_ms();
        use_pieces = cfg.pieces;
        seg_len_env = cfg.seg_len;
        seg_ov = cfg.seg_overlap;
        force_exact = cfg.force_exact;
        RS.clear();
        RS.resize(n_sel);
        jref.assign(NR * (size_t)QCAP, JobRef{});
        for (auto &x : n_leap_refused) x = 0;
        use_leap_pieces = cfg.leap_pieces;
        deferred.clear();
        deferred.resize(n_sel);
        b_ckreq = buf(), b_ckout = buf();
        lap("rings");
        return PAG_OK;
    }
    // pinned staging + the walk arena
    int reserve_arena() {
        if (!pinned(64u << 20)) return PAG_ENOMEM;  // (grown later if a batch needs more)
        {   // the walk arena: sized for the first round of every contig (chain buffers over the whole strand + segment buffers)
            // plus half again for resumed walks and later rounds; at most 40 % of the free device memory; kept by the handle
            size_t want = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                const CtgState &cs = st[i];
                const size_t span = (size_t)(cs.inHi - cs.inLo) + 8, cap = cs.seqCap, oc = pow2_at_least(cap / 4 + 4096);
                const size_t chain = cap * 8 + cap * 8 * TRAV_PROBE_GROUPS + oc * 8 * (1 + TRAV_PROBE_GROUPS) + span * 4 * (1 + TRAV_PROBE_GROUPS);
                const size_t n_seg = cs.len / 12000 + 1, scap = 8192 + 8192, soc = pow2_at_least(scap / 4 + 4096), sspan = span / (n_seg ? n_seg : 1) * 2 + 4096;
                const size_t seg = scap * 8 + scap * 8 * TRAV_PROBE_GROUPS + soc * 8 * (1 + TRAV_PROBE_GROUPS) + sspan * 4 * (1 + TRAV_PROBE_GROUPS);
                // segments of the leaping zone (the last tenth of the strand + margin, half as long, far larger hash sets, a log)
                const size_t n_lseg = cs.len / 8 / 6000 + 2, lcap = 3000 + 8192, loc = pow2_at_least(lcap + 8192), lspan = sspan;
                const size_t lseg = lcap * 8 + lcap * 8 * TRAV_PROBE_GROUPS + lcap * 8 + loc * 8 * (1 + TRAV_PROBE_GROUPS) + lspan * 4 * (1 + TRAV_PROBE_GROUPS);
                // (full-strand buffers: the resumed walks — the seeds' own first pieces are sized like segments)
                want += chain * 3 / 2 + (seg * (n_seg + 8) + lseg * n_lseg) * 3 / 2;
            }
            size_t free_b = 0, total_b = 0;
            const size_t sharers = std::getenv("PAG_DEVICE_SHARERS") ? (size_t)std::max(1, std::atoi(std::getenv("PAG_DEVICE_SHARERS"))) : 1;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) want = std::min(want, (free_b + g->walk_arena_cap) * 2 / 5 / sharers);
            if (g->walk_arena_cap < want / 10 * 7) {  // (an arena that is there — pag_reserve_walk_arena, an earlier call — is kept
                                                      // unless it is much too small: what does not fit goes to the slots)
                if (g->walk_arena) hipFree(g->walk_arena);
                g->walk_arena = nullptr;
                g->walk_arena_cap = 0;
                if (hipMalloc(&g->walk_arena, want) == hipSuccess) g->walk_arena_cap = want;
                else g->walk_arena = nullptr;  // (the slots do all the work then)
            }
            g->walk_arena_used = 0;
        }
        lap("arena");
        return PAG_OK;
    }
    // the first round of every contig is posted, the walker grid launched
    int post_first_rounds() {
        int rc;
        tw0 = now_ms();
        t_walk0 = tw0;
        g->defer_free = true;
        {   // longest contigs first: their exact tails (the leaping zone is a tenth of the contig) are the longest, so their
            // segments should be through the queue first
            std::vector<uint32_t> order(n_sel);
            for (uint32_t i = 0; i < n_sel; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a2, uint32_t b2) { return st[a2].len > st[b2].len; });
            // First rounds: the contigs' segment jobs enter the ring interleaved, a few per contig and turn (a contig's leap
            // segments first).  Posted contig by contig, the last contigs of the list finish their first round when the grid
            // runs empty — and those of them that need a second round (a re-seed after a walk that ended early) start it then:
            // every contig's first round now ends at about the same time, earlier than the last ones did.
            const uint32_t interleave = 16;  // (the share of the contig with the fewest jobs)
            defer_ring2 = true;
            {
                std::vector<uint32_t> first_rounds;
                for (uint32_t i : order)
                    if (!st[i].done) first_rounds.push_back(i);
                if ((rc = start_rounds(first_rounds))) return fail(rc);
            }
            lap("first rounds planned");
            defer_ring2 = false;
            {
                std::vector<size_t> at(n_sel, 0);
                // a contig's share of a turn (round 5): in proportion to the jobs it has, so that every contig's first round runs
                // out of the ring in the same turn.  With equal shares the contigs with the most segments — the longest ones, whose
                // chains also take the longest to stitch — saw their last segments START when the grid was already running empty
                // (configs[1]: at 58 of 105 ms), and the ones among them that need a second round started it last of all.
                std::vector<uint32_t> share(n_sel, interleave);
                {
                    size_t least = 0;
                    for (uint32_t i : order)
                        if (!deferred[i].empty() && (least == 0 || deferred[i].size() < least)) least = deferred[i].size();
                    const size_t turns = least ? (least + interleave - 1) / interleave : 1;
                    for (uint32_t i : order)
                        if (!deferred[i].empty()) share[i] = (uint32_t)std::max<double>(1.0, std::ceil((double)deferred[i].size() / (double)turns));
                }
                // (a turn of its own for the contigs' longest jobs — the piece that runs to the end of the strand, first in every
                // contig's list: they all start with the first wave of the grid)
                for (uint32_t i : order) {
                    auto &dq = deferred[i];
                    if (at[i] < dq.size() && (dq[0].P.J.mode & TRAV_MODE_LEAP) && dq[0].P.J.stop_pc == 0u) {
                        if ((rc = commit_job(2u, dq[0].P, dq[0].jr, dq[0].P.J.mode, dq[0].P.J.stop_pc))) return fail(rc);
                        at[i] = 1;
                    }
                }
                for (bool more = true; more;) {
                    more = false;
                    for (uint32_t i : order) {
                        auto &dq = deferred[i];
                        for (uint32_t c = 0; c < share[i] && at[i] < dq.size(); ++c, ++at[i])
                            if ((rc = commit_job(2u, dq[at[i]].P, dq[at[i]].jr, dq[at[i]].P.J.mode, dq[at[i]].P.J.stop_pc))) return fail(rc);
                        more = more || at[i] < dq.size();
                    }
                }
                for (auto &dq : deferred) std::vector<Deferred>().swap(dq);
            }
        }
        lap("ring order");
        if (n_live) {
            walkers.init(g, G, hjobs, houts, hdone, hq, QCAP, k);
            if ((rc = publish())) return fail(rc);  // (the jobs' buffers are ready, the rings are visible)
            lap("marks cleared, rings published");
            if ((rc = walkers.ensure(n_live))) {
                g->defer_free = false;
                return rc;
            }
            if (wdebug) std::fprintf(stderr, "[walk] %u walker waves launched (at most %u), %u + %u + %u jobs posted\n", walkers.launched, walkers.max_waves, n_posted[0], n_posted[1], n_posted[2]);
        } else {
            g->defer_free = false;
        }
        lap("round prep");
        return PAG_OK;
    }

    // ---- the event loop, step by step
    // jobs that have finished since the last look (again: nothing to do yet, look again)
    int poll_finished(std::vector<uint32_t> &fin, bool &again) {
        int rc;
        again = false;
        for (uint32_t ring = 0; ring < NR; ++ring) {
            // (at most QCAP jobs of a ring are in flight: a slot is taken again only when the job QCAP numbers earlier has been
            // handled — without this clamp a ring far smaller than a round would be scanned around more than once)
            if (n_posted[ring] > QCAP && scan_from[ring] < n_posted[ring] - QCAP) scan_from[ring] = n_posted[ring] - QCAP;
            while (scan_from[ring] < n_posted[ring] && !jref[ring * QCAP + scan_from[ring] % QCAP].live) ++scan_from[ring];
            for (uint32_t jn = scan_from[ring]; jn < n_posted[ring]; ++jn) {
                const uint32_t slot = ring * QCAP + jn % QCAP;
                if (jref[slot].live && __atomic_load_n(&hdone[slot], __ATOMIC_ACQUIRE) != 0) fin.push_back(slot);
            }
        }
        // A batch costs a kernel launch, a copy and a stream synchronisation (~0.1 ms of this thread): finished SEGMENT jobs
        // that no chain is waiting for are left to accumulate (up to 64 of them or 1 ms); a finished chain job, or a
        // segment some chain of its contig waits for, is fetched at once.
        if (!fin.empty()) {
            bool urgent = fin.size() >= 64 || (t_first_fin > 0 && now_ms() - t_first_fin > 1.0);
            for (size_t x = 0; x < fin.size() && !urgent; ++x) {
                const JobRef &jr = jref[fin[x]];
                if (jr.kind == 0) urgent = true;
                else
                    for (const Chain &ch : RS[jr.ctg].chains)
                        if (ch.waiting_seg == jr.idx) urgent = true;
            }
            if (t_first_fin == 0) t_first_fin = now_ms();
            if (!urgent) {
                idle_wait(20.0);
                again = true;
                return PAG_OK;
            }
            t_first_fin = 0;
        }
        if (fin.empty() && over_queue.empty()) {
            // waves that found nothing to do have left (k_walk_persistent): jobs that are outstanding get new ones
            if ((rc = walkers.ensure(n_live))) return fail(rc);
            const double idle_limit_ms = cfg.idle_limit_ms;
            if (now_ms() - t_progress > idle_limit_ms) {  // no job finished for a minute: give up instead of hanging
                uint32_t ticket[TRAV_RINGS] = {0, 0, 0};
                hipMemcpyAsync(ticket, g->wq_next, sizeof(ticket), hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                set_error("pag_travel: no walk job finished within %.0f s (posted %u + %u + %u, claimed %u + %u + %u, jobs outstanding %u, walker waves started %u / left %u of %u launched)",
                          idle_limit_ms / 1000.0, n_posted[0], n_posted[1], n_posted[2], ticket[0], ticket[1], ticket[2], n_live, walkers.started(), walkers.exited(), walkers.launched);
                return fail(PAG_EFAULT);
            }
            idle_wait(30.0);
            again = true;
            return PAG_OK;
        }
        t_progress = now_ms();
        t_last_news = t_progress;
        lap("walk");
        return PAG_OK;
    }
    // ---- their paths: vertices, steps and contig coordinates to the host (one round trip for the batch)
    struct Got {
        uint32_t jn;
        uint64_t from, len, off;       // the part of the sequence that is new; word offset of its packed words (trav_pack_words)
        const uint32_t *v, *s, *pc;     // ... in pinned memory that lives as long as this call (fetch_alloc)
        const uint32_t *xl = nullptr, *xh = nullptr;  // TRAV_MODE_LEAP: low / high words of the iteration log
        const uint32_t *agg = nullptr, *xagg = nullptr;  // block tables of those arrays (walk_stitch.hpp; written by k_pack_paths)
    };
    int fetch_paths(const std::vector<uint32_t> &fin, std::vector<Got> &got) {
        static_assert(AGG_BLOCK == 64 && AGG_WORDS == 5 && AGG_XWORDS == 2, "k_pack_paths writes these tables");
        got.assign(fin.size(), Got{});
        {
            uint64_t tot = 0, max_len = 0;
            std::vector<TravPackDesc> descs(fin.size());
            for (size_t x = 0; x < fin.size(); ++x) {
                const uint32_t slot = fin[x];
                const TravJobOut &o = houts[slot];
                const TravJob &J = hjobs[slot].J;
                Got &G2 = got[x];
                G2.jn = fin[x];
                G2.from = std::min<uint64_t>(jref[slot].init_len, o.seq_len);
                G2.len = is_orphan(jref[slot]) ? 0 : o.seq_len - G2.from;  // (nobody reads an orphan's path)
                G2.off = tot;
                tot += trav_pack_words(G2.len, J.seq_x != nullptr);
                max_len = std::max(max_len, G2.len);
                descs[x] = TravPackDesc{J.seq_v + G2.from, J.seq_s + G2.from, G2.len, G2.off, J.seq_x ? J.seq_x + G2.from : nullptr};
            }
            uint32_t *hp = (uint32_t *)fetch_alloc(tot * 4 + fin.size() * sizeof(TravPackDesc) + 256);
            if (!hp) return fail(PAG_ENOMEM);
            TravPackDesc *hd = (TravPackDesc *)(hp + ((tot + 3) & ~3ull));
            std::memcpy(hd, descs.data(), descs.size() * sizeof(TravPackDesc));
            // the pack kernel reads its descriptors from, and writes the packed paths to, the pinned host memory directly: one
            // launch + one synchronisation per batch instead of copy + launch + copy + synchronisation (every call of this
            // thread is on the critical path of some chain)
            trav_launch_pack_paths(G, hd, (uint32_t)descs.size(), max_len, hp, s);
            if (hipStreamSynchronize(s) != hipSuccess) {
                set_error("pag_travel: stream failure while fetching paths");
                return fail(PAG_EFAULT);
            }
            const bool check_aggs = cfg.check_aggs;
            for (Got &G2 : got) {
                G2.v = hp + G2.off;
                G2.s = G2.v + G2.len;
                G2.pc = G2.s + G2.len;
                G2.agg = G2.pc + G2.len;
                if (hjobs[G2.jn].J.seq_x) {
                    G2.xl = G2.pc + G2.len;
                    G2.xh = G2.xl + G2.len;
                    G2.agg = G2.xh + G2.len;
                    G2.xagg = G2.agg + agg_blocks((size_t)G2.len) * AGG_WORDS;
                }
                if (check_aggs) {  // (tests: the device's block tables against the host's definition of them)
                    std::vector<uint32_t> want(agg_blocks((size_t)G2.len) * AGG_WORDS), wantx(agg_blocks((size_t)G2.len) * AGG_XWORDS);
                    build_block_aggs(G2.v, G2.s, G2.pc, (size_t)G2.len, want.data());
                    bool same = std::memcmp(want.data(), G2.agg, want.size() * 4) == 0;
                    if (G2.xagg) {
                        build_block_xaggs(G2.xl, G2.xh, (size_t)G2.len, wantx.data());
                        same = same && std::memcmp(wantx.data(), G2.xagg, wantx.size() * 4) == 0;
                    }
                    if (!same) {
                        set_error("pag_travel: block tables of a fetched path differ from their definition (job %u, %llu entries)", G2.jn, (unsigned long long)G2.len);
                        return fail(PAG_EFAULT);
                    }
                }
            }
        }
        lap("fetch");
        return PAG_OK;
    }
    // bookkeeping of the finished jobs, their paths into segments and chains, the chains move on; touched: the contigs with news
    int stitch_finished(std::vector<Got> &got, std::vector<uint32_t> &touched) {
        int rc;
        const double ts0 = now_ms();
        // serial part: bookkeeping, and the (rare) jobs that have to be posted again
        std::vector<size_t> heavy;  // items of `got` whose path has to be copied / indexed
        for (size_t gx = 0; gx < got.size(); ++gx) {
            Got &G2 = got[gx];
            const uint32_t slot = G2.jn;
            JobRef &jr = jref[slot];
            const TravJobOut o = houts[slot];
            const uint32_t i = jr.ctg;
            RoundState &R = RS[i];
            if (is_orphan(jr)) {  // a segment job of a round that is over: its slot is free again, nothing else
                jr.live = false;
                n_live -= 1;
                continue;
            }
            jr.live = false;
            n_live -= 1;
            R.live_jobs -= 1;
            steps_total += o.seq_len - G2.from;
            classify_total += o.n_classify;
            probe_total += o.n_probe;
            record_total += o.n_records;
            const bool overflow = (o.overflow & 3) != 0, misspec = (o.overflow & 4) != 0;
            if (o.poison) {
                // (a regional graph, pag_shard_select: the walk reached a vertex whose successors another rank holds)
                set_error(g->regional ? "pag_travel: a walk of contig %u left the region of the graph this rank holds (reference band halo too small: raise PAG_SHARD_HALO)"
                                      : "pag_travel: a walk of contig %u left the view built for this handle's traversals (PAG_VIEW_HALO / PAG_VIEW_MARGIN)",
                          st[i].ci);
                return fail(PAG_ERANGE);
            }
            touched.push_back(i);
            if (jr.kind == 1) {  // a segment
                Seg &sg = R.segs[(size_t)jr.idx];
                sg.usable = !overflow && !misspec && G2.len >= 8;
                sg.stopped = o.stopped != 0;
                if (sg.usable) heavy.push_back(gx);
                else sg.done = true;
                continue;
            }
            Chain &ch = R.chains[(size_t)jr.idx];
            ch.job = -1;
            if (misspec && !overflow) {  // a zombie probe leapt: the job is walked again, every probe to its end
                ch.exact = true;
                ++respeculated;
                if (wdebug) std::fprintf(stderr, "[walk] contig %u chain %d: speculation failed, exact walk\n", i, jr.idx);
            } else if (overflow) {
                if (misspec) ch.exact = true;
                if (ch.grow >= (1u << 24)) {  // (the buffers double until the walk fits; a device allocation that fails reports itself)
                    set_error("pag_travel: walker buffers overflow at %u times their first size", ch.grow);
                    return fail(PAG_ENOMEM);
                }
                ch.grow *= 2;
            }
            if (misspec || overflow) {  // the same job again (its path so far, if any, is still on the host)
                std::vector<JobPlan> plans;
                const uint64_t cap = std::max<uint64_t>(st[i].seqCap * ch.grow, ch.len + st[i].seqCap / 4 + 4096);
                plans.push_back(JobPlan{0, jr.idx, cap, st[i].seeds[(size_t)jr.idx].vid, ch.job_mode, ch.job_stop, (ch.job_mode & TRAV_MODE_RESUME) ? &ch : nullptr, ch.exact});
                if ((rc = post_batch(i, GRP_CHAIN0 + jr.idx, plans))) return fail(rc);
                continue;
            }
            if (!o.stopped) ch.final = true;
            heavy.push_back(gx);
        }
        const double ts1 = now_ms();
        // parallel part: the paths of the finished jobs (a job belongs to one segment or one chain: the items are independent)
        for (size_t hx = 0; hx < heavy.size(); ++hx) {
            Got &G2 = got[heavy[hx]];
            const uint32_t slot = G2.jn;
            const JobRef &jr = jref[slot];
            const TravJobOut o = houts[slot];
            RoundState &R = RS[jr.ctg];
            if (jr.kind != 1) {  // the new part of a chain's path
                extend_chain(R.chains[(size_t)jr.idx], G2.v, G2.s, G2.pc, (size_t)G2.len, nullptr, G2.agg, 0, hjobs[slot].J.seq_v + G2.from, hjobs[slot].J.seq_s + G2.from);
                continue;
            }
            Seg &sg = R.segs[(size_t)jr.idx];
            sg.P.v = G2.v;
            sg.P.s = G2.s;
            sg.P.pc = G2.pc;
            sg.P.xl = G2.xl;
            sg.P.xh = G2.xh;
            sg.P.n = (size_t)G2.len;
            sg.P.agg = G2.agg;
            sg.P.xagg = G2.xagg;
            sg.P.dv = hjobs[slot].J.seq_v + G2.from;
            sg.P.ds = hjobs[slot].J.seq_s + G2.from;
            // (a coordinate-free vertex cannot happen while leaping is off; never adopt such a path)
            if (!sg.leap && range_agg(sg.P.v, sg.P.s, sg.P.pc, sg.P.agg, 0, sg.P.n).lo_all == 0u) sg.usable = false;
            if (sg.leap) {
                sg.usable = sg.usable && G2.xl != nullptr;
                sg.wd_below_max = o.wd_below_max;
                sg.wd_forced_min = o.wd_forced_min;
            }
            sg.max_back = o.max_back;
            sg.max_chosen = std::max<uint32_t>(o.max_chosen, 1u);
            sg.max_probe = o.max_probe;
            sg.done = true;
        }
        const double ts2 = now_ms();
        if (wtrace)
            for (Got &G2 : got) {
                const JobRef &jr = jref[G2.jn];
                const TravJobOut &o = houts[G2.jn];
                trace.push_back(TraceEv{ts2 - tw0, 0u, jr.ctg, (int32_t)jr.kind, (int32_t)jr.idx, o.t_begin, o.t_end, G2.len, o.n_classify});
            }
        if (wdebug)
            for (Got &G2 : got) {
                const uint32_t slot = G2.jn;
                const JobRef &jr = jref[slot];
                const TravJobOut o = houts[slot];
                const uint32_t i = jr.ctg;
                if (jr.kind == 1) {
                    const Seg &sg = RS[i].segs[(size_t)jr.idx];
                    std::fprintf(stderr, "[walk] t=%.1f ms dev %.3f..%.3f contig %u segment %d done: %llu vertices, %s, %s (flags %d, outside %llu, classify %llu, back %u, chosen %u, probe %llu)\n", now_ms() - tw0,
                                 (double)(o.t_begin % 100000000000ull) * 1e-5, (double)(o.t_end % 100000000000ull) * 1e-5, i, jr.idx, (unsigned long long)o.seq_len, sg.stopped ? "stopped" : "ended", sg.usable ? "usable" : "NOT usable", o.overflow, (unsigned long long)o.n_out, (unsigned long long)o.n_classify,
                                 o.max_back, o.max_chosen, (unsigned long long)o.max_probe);
                } else if ((o.overflow & 7) == 0) {
                    std::fprintf(stderr, "[walk] t=%.1f ms dev %.3f..%.3f contig %u chain %d job done: +%llu vertices (%zu), %s (classify %llu)\n", now_ms() - tw0,
                                 (double)(o.t_begin % 100000000000ull) * 1e-5, (double)(o.t_end % 100000000000ull) * 1e-5, i, jr.idx, (unsigned long long)G2.len,
                                 RS[i].chains[(size_t)jr.idx].len, o.stopped ? "stopped" : "ended", (unsigned long long)o.n_classify);
#ifdef PAG_WALK_PROF
                    // (make WALK_PROF=1: 100 MHz ticks and counts per section of the job — 0 append, 1 classification, 2 wait for a
                    // free slot, 3 slot setup, 4 slot steps, 5 choice, 6 window refills, 7 whole-wave probes, 8-11 inside a slot step:
                    // window, evaluation, class minimum + shuffles, state update, 12 job setup: filters + the contig's global set)
                    std::fprintf(stderr, "[walk]    prof mode %u main %llu fills %llu probes %llu:", hjobs[slot].J.mode, (unsigned long long)o.n_main,
                                 (unsigned long long)(o.n_fill & 0xFFFFFFFFull), (unsigned long long)o.n_probe);
                    for (int q = 0; q < 14; ++q) std::fprintf(stderr, " [%d] %.2f ms / %u", q, (double)o.prof_t[q] * 1e-5, o.prof_c[q]);
                    std::fprintf(stderr, "\n");
#endif
                }
            }
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        // the chains of the touched contigs move on: adoptions (advance_chain reads the round's segments and writes its chain
        // only: the chains are independent, those of one contig too), then the resumed walks are posted by this thread.  Host
        // threads since round 5: when the device runs out of first-round work, ~900 segments finish within 15 ms and their
        // adoptions — 5 us each, 24 ms per block — were what the control thread was busy with while finished chain jobs waited
        // (tests/walk_trace.py: laps of 3-4 ms per loop iteration in the last 30 ms of the walks).
        {
            struct AdvTask {
                uint32_t i;
                int c;
                Next nx;
            };
            std::vector<AdvTask> tasks;
            for (uint32_t i : touched) {
                RoundState &R = RS[i];
                for (size_t c = 0; c < R.chains.size(); ++c) {
                    const Chain &ch = R.chains[c];
                    if (ch.final || ch.job >= 0) continue;
                    if (ch.waiting_seg >= 0 && !R.segs[(size_t)ch.waiting_seg].done) continue;
                    tasks.push_back(AdvTask{i, (int)c, Next{}});
                }
            }
            const unsigned nthr = (unsigned)std::min<size_t>(tasks.size() / 2, cfg.stitch_threads);
            const uint64_t fails_before = adv_stats.merge_fail;
            auto run = [&](std::atomic<size_t> &next, stitch::AdvanceStats &S) {
                for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1))
                    tasks[t].nx = advance_chain(RS[tasks[t].i], RS[tasks[t].i].chains[(size_t)tasks[t].c], merge_ctx(tasks[t].i), seg_ov, S);
            };
            std::atomic<size_t> next{0};
            if (nthr <= 1) {
                run(next, adv_stats);
            } else {
                std::vector<stitch::AdvanceStats> part(nthr);
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nthr; ++t) pool.emplace_back([&, t]() { run(next, part[t]); });
                run(next, part[0]);
                for (auto &th : pool) th.join();
                for (const stitch::AdvanceStats &S : part) {
                    adv_stats.adopted += S.adopted;
                    adv_stats.leap_adopted += S.leap_adopted;
                    adv_stats.merge_fail += S.merge_fail;
                    for (int w = 0; w < 8; ++w) adv_stats.leap_refused[w] += S.leap_refused[w];
                }
            }
            n_adopted = adv_stats.adopted;
            n_leap_adopted = adv_stats.leap_adopted;
            n_merge_fail = adv_stats.merge_fail;
            for (int w = 0; w < 8; ++w) n_leap_refused[w] = adv_stats.leap_refused[w];
            if (wdebug && adv_stats.merge_fail != fails_before)
                std::fprintf(stderr, "[walk] %llu segments were not adoptable, walking on exactly\n", (unsigned long long)(adv_stats.merge_fail - fails_before));
            for (const AdvTask &T : tasks)
                if (T.nx.what == Next::Resume && (rc = post_resume(T.i, T.c, T.nx.stop, T.nx.until_leap))) return fail(rc);
            if ((rc = flush_clears())) return fail(rc);
        }
        t_st[0] += ts1 - ts0;
        t_st[1] += ts2 - ts1;
        t_st[2] += now_ms() - ts2;
        lap("stitch");
        return PAG_OK;
    }
    // contigs whose chains are all final: the round is decided; batch: those taken now
    void decide_rounds(const std::vector<uint32_t> &touched, std::vector<uint32_t> &batch) {
        for (uint32_t i : touched) {
            RoundState &R = RS[i];
            if (!R.active) continue;
            bool all = true;
            for (auto &ch : R.chains) all = all && ch.final;
            // (segment jobs still waiting or walking stay with the contig or become orphans — unless the round's buffers are
            // per-contig slots, which the next round takes over: such a round waits for them)
            if (all && (R.live_jobs == 0 || !R.slot_bufs) && std::find(over_queue.begin(), over_queue.end(), i) == over_queue.end())
                over_queue.push_back(i);
        }
        {
            auto leaps = [&](uint32_t i) {  // the round ended on another contig: the contig is finished (PAlgorithm.cpp:254-262, 322-328)
                for (const Chain &ch : RS[i].chains) {
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    if (last_ctg != 0 && mapper.singleToDual(last_ctg).first != st[i].chosenOne) return true;
                }
                return false;
            };
            const size_t take = over_queue.size();
            std::stable_partition(over_queue.begin(), over_queue.end(), [&](uint32_t i) { return !leaps(i); });
            batch.assign(over_queue.begin(), over_queue.begin() + (long)take);
            over_queue.erase(over_queue.begin(), over_queue.begin() + (long)take);
        }
    }
    // ---- per contig: choose (PAlgorithm.cpp:238-262); the chosen walks are uploaded, gathered and committed on the
    //      device back to back, copied out, and spliced by a pool of host threads (contigs are independent)
    struct Pick {
        int chosen = -1;
        bool leap = false;
        size_t chooseCtgPos = 0, chooseRefPos = 0;
        uint64_t off = 0, len = 0;
    };
    int take_walks(const std::vector<uint32_t> &batch, std::vector<Pick> &picks) {
        int rc;
        {
            uint64_t tot = 0;
            for (uint32_t i : batch) {
                CtgState &cs = st[i];
                RoundState &R = RS[i];
                R.active = false;
                Pick &P = picks[i];
                size_t maxLen = 0;
                for (size_t sd = 0; sd < R.chains.size(); ++sd) {
                    const Chain &ch = R.chains[sd];
                    const size_t len = ch.size;
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    P.leap = last_ctg != 0 && mapper.singleToDual(last_ctg).first != cs.chosenOne;
                    if (!P.leap && sd > 0 && prm->min_len > 0 && len < prm->min_len) continue;
                    if (len > maxLen || P.leap) {
                        maxLen = len;
                        P.chosen = (int)sd;
                        P.chooseCtgPos = (size_t)mapper.singleToDual(cs.seeds[sd].ctg).second;
                        P.chooseRefPos = (size_t)refMapper.singleToDual(cs.seeds[sd].ref).second;
                        if (P.leap) break;
                    }
                }
                if (P.chosen >= 0) {
                    P.len = R.chains[(size_t)P.chosen].len;
                    tot += P.len;  // (an upper bound: walks that stay on the device take no room, see below)
                }
                if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 2u, i, 0, P.chosen, R.round, P.leap ? 1u : 0u, P.chosen >= 0 ? R.chains[(size_t)P.chosen].len : 0, 0});
                if (wdebug && P.chosen >= 0) {
                    const Chain &ch = R.chains[(size_t)P.chosen];
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    std::fprintf(stderr, "[walk] t=%.1f ms contig %u round %llu over: chain %d of %zu chosen, %llu vertices, size %llu, from offset %lld, ends at offset %lld (strand %u), mx %u%s\n",
                                 now_ms() - t_begin, i, (unsigned long long)R.round, P.chosen, R.chains.size(), (unsigned long long)ch.len, (unsigned long long)ch.size,
                                 (long long)cs.seeds[(size_t)P.chosen].ctg - (long long)cs.ctgLeft, last_ctg ? (long long)last_ctg - (long long)cs.ctgLeft : -1ll, cs.len,
                                 ch.mx_all >= cs.ctgLeft ? ch.mx_all - cs.ctgLeft : 0u, P.leap ? ", leap" : "");
                }
            }
            // the chosen walks go to the device through the pinned staging area (vertex ids only: the commit kernels need
            // nothing else), one copy for the batch
            uint32_t *hp = (uint32_t *)pinned(tot * 4 + 256);
            if (!hp) return fail(PAG_ENOMEM);
            if ((rc = b_gather.alloc(tot * 4 + 64))) return fail(rc);
            // ONE pass over the chosen walk of every contig of the batch (its parts lie where the fetches put them): the
            // vertex ids for the device, the walk appended to the contig's running path (appendSeq, PAlgorithm.cpp:110-142),
            // the coordinate window of the global table, the vertices outside the strand's id range.  (Five passes and two
            // copies of the walk before: 14 M path vertices per block at configs[1], on the thread every contig waits for.)
            // The chosen chains' parts are copied to the contigs' paths (cs.travel) and to the id list of the commit in chunks,
            // by a small pool of threads: a round of a long contig is millions of vertices in a handful of parts, and the
            // entries are in pinned memory the device wrote (first read = DRAM latency).  What the loop used to add up on
            // the way — the steps, the coordinate window — the chain knows already.
            struct CopyChunk {
                uint32_t i;            // contig
                const Chain::Part *pt;
                size_t x0, x1;         // entries of the part
                LNode *dst;            // of the part's first entry
                uint32_t *ids;
                std::vector<uint32_t> outside;  // vertices outside the strand's id range, in order
            };
            std::vector<CopyChunk> chunks;
            const size_t CHUNK = 1u << 17;
            uint64_t used = 0;
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0) continue;
                const Chain &ch = RS[i].chains[(size_t)P.chosen];
                CtgState &cs = st[i];
                std::vector<LNode> &base = cs.travel;
                int64_t dLen = 0;
                const uint32_t head_ctg = ch.parts.front().pc[0];
                int32_t dist = (int32_t)k;
                while (!base.empty() && (base.back().ctg == 0 || head_ctg <= base.back().ctg)) {
                    dLen -= base.back().step;
                    base.pop_back();
                }
                if (!base.empty()) dist = (int32_t)(head_ctg - base.back().ctg);
                const size_t at0 = base.size();
                if (P.leap && !RS[i].slot_bufs && g->walk_arena) {
                    // the contig is finished by this walk (splice below): nothing of it is needed on the host
                    bool on_dev = true;
                    for (const Chain::Part &pt : ch.parts) on_dev = on_dev && pt.dv && pt.ds;
                    const size_t cap = at0 + P.len, need = (cap * 8 + 255) & ~(size_t)255;
                    TravConcatPart *cp = on_dev && g->walk_arena_used + need <= g->walk_arena_cap ? (TravConcatPart *)fetch_alloc(ch.parts.size() * sizeof(TravConcatPart)) : nullptr;
                    if (cp) {
                        CtgState::DevTail &T = cs.tail;
                        T.on = true;
                        T.d_ids = (uint32_t *)((char *)g->walk_arena + g->walk_arena_used);
                        g->walk_arena_used += need;
                        T.cap = cap;
                        T.m0 = at0;
                        T.n = P.len;
                        T.last_ctg = ch.parts.back().pc[ch.parts.back().n - 1];
                        for (size_t x = 0; x < ch.parts.size(); ++x) cp[x] = TravConcatPart{ch.parts[x].dv, ch.parts[x].ds, ch.parts[x].start, ch.parts[x].n};
                        if (!g->deliver_stream && hipStreamCreateWithFlags(&g->deliver_stream, hipStreamNonBlocking) != hipSuccess) return fail(PAG_EFAULT);
                        trav_launch_concat_parts(cp, (uint32_t)ch.parts.size(), T.d_ids + at0, T.d_ids + cap + at0, (uint32_t)dist, g->deliver_stream);
                        continue;
                    }
                }
                base.resize(at0 + P.len);
                LNode *dst = base.data() + at0;
                picks[i].off = used;
                used += P.len;
                uint32_t *ids = hp + P.off;
                for (const Chain::Part &pt : ch.parts)
                    for (size_t x0 = 0; x0 < pt.n; x0 += CHUNK)
                        chunks.push_back(CopyChunk{i, &pt, x0, std::min(pt.n, x0 + CHUNK), dst + pt.start, ids + pt.start, {}});
                dLen += (int64_t)ch.size;
                if (ch.low_nz != 0xFFFFFFFFu) {  // (some vertex has a coordinate)
                    cs.gwinLo = std::min(cs.gwinLo, ch.low_nz);
                    cs.gwinHi = std::max(cs.gwinHi, ch.mx_all);
                }
                cs.gFreeHi = std::max(cs.gFreeHi, ch.m0_all);
                // the first vertex of the round's path: its step is the distance to the path so far (set after the copy)
                cs.varLen += dLen - ((int64_t)ch.parts.front().s[0] - dist);
                cs.pendingFirst = at0;
                cs.pendingFirstStep = dist;
            }
            {
                std::atomic<size_t> nxt{0};
                auto worker = [&]() {
                    for (size_t c; (c = nxt.fetch_add(1)) < chunks.size();) {
                        CopyChunk &C = chunks[c];
                        const Chain::Part &pt = *C.pt;
                        const uint32_t in_lo = st[C.i].inLo, in_hi = st[C.i].inHi;
                        for (size_t x = C.x0; x < C.x1; ++x) {
                            const uint32_t v = pt.v[x];
                            C.dst[x] = LNode(v, (int32_t)pt.s[x], pt.pc[x]);
                            C.ids[x] = v;
                            if (v < in_lo || v >= in_hi) C.outside.push_back(v);
                        }
                    }
                };
                // (one thread: measured at configs[1] on the GPU box, 16-CPU quota, the previous block's host half running beside — 430 ms
                // per block with one thread, 436 with six; the copy is not what the round waits for)
                worker();
            }
            for (CopyChunk &C : chunks)  // (in the order of the path)
                if (!C.outside.empty()) st[C.i].outsideU.insert(st[C.i].outsideU.end(), C.outside.begin(), C.outside.end());
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0 || st[i].tail.on) continue;
                st[i].travel[st[i].pendingFirst].step = st[i].pendingFirstStep;
            }
            if (used) hipMemcpyAsync(b_gather.p, hp, used * 4, hipMemcpyHostToDevice, s);
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0 || st[i].tail.on) continue;
                CtgState &cs = st[i];
                // globalUniqueTable on the device: its hash set (vertices outside the strand's id range) grows as needed
                if ((uint64_t)cs.outsideU.size() * 2 > cs.gcap) {
                    uint32_t ncap = cs.gcap;
                    while ((uint64_t)cs.outsideU.size() * 4 > ncap) ncap *= 2;
                    DevBuf b_ng = cbuf(i, GRP_FINAL, CB_TSET), b_ou = cbuf(i, GRP_FINAL, CB_PSET);
                    if ((rc = b_ng.alloc((size_t)ncap * 4)) || (rc = b_ou.alloc(cs.outsideU.size() * 4))) return fail(rc);
                    hipMemsetAsync(b_ng.p, 0xFF, (size_t)ncap * 4, s);
                    // (the vertices of this round's path are inserted by the commit below; the earlier ones here)
                    hipMemcpyAsync(b_ou.p, cs.outsideU.data(), cs.outsideU.size() * 4, hipMemcpyHostToDevice, s);
                    trav_launch_commit(b_ou.as<uint32_t>(), cs.outsideU.size(), 0u, 0u, nullptr, b_ng.as<uint32_t>(), ncap - 1, s);
                    cs.gset = b_ng.as<uint32_t>();
                    cs.gcap = ncap;
                }
                // record the walk in the device-side global visited set of this contig
                trav_launch_commit(b_gather.as<uint32_t>() + P.off, P.len, cs.inLo, cs.inHi, cs.gbits, cs.gset, cs.gcap - 1, s);
                cs.committed = true;
            }
        }
        for (uint32_t i : batch) RS[i].chains.clear();  // (the host copies of the round's chains are spent; its segments: below)
        lap("choose+gather");
        return PAG_OK;
    }
    // splice + stop rules (PAlgorithm.cpp:264-360); reqs / req_cs: the seed searches of the contigs that go on
    void splice_batch(const std::vector<uint32_t> &batch, const std::vector<Pick> &picks, std::vector<TravSeedReq> &reqs, std::vector<uint32_t> &req_cs) {
        std::vector<TravSeedReq> slot_req(n_sel);
        std::vector<uint8_t> slot_has(n_sel, 0);
        auto splice = [&](uint32_t i) {
            CtgState &cs = st[i];
            const Pick &P = picks[i];
            const bool leap = P.leap;
            // (the walk was appended to cs.travel by take_walk above)
            if (P.chooseCtgPos != 0) {
                cs.ctgQ.push_back((uint32_t)P.chooseCtgPos);
                while (cs.ctgQ.size() > 4) cs.ctgQ.pop_front();
            }
            if (P.chooseRefPos != 0) {
                cs.refQ.push_back((uint32_t)P.chooseRefPos);
                while (cs.refQ.size() > 4) cs.refQ.pop_front();
            }
            bool ctgRepeat = false, refRepeat = false;
            if (cs.ctgQ.size() >= 4) {
                auto mm = std::minmax_element(cs.ctgQ.begin(), cs.ctgQ.end());
                ctgRepeat = (uint64_t)(*mm.second - *mm.first) <= 2 * deviation;
            }
            if (cs.refQ.size() >= 4) {
                auto mm = std::minmax_element(cs.refQ.begin(), cs.refQ.end());
                refRepeat = (uint64_t)(*mm.second - *mm.first) <= 2 * deviation;
            }
            if (ctgRepeat || refRepeat || leap) {
                if (leap) cs.finalLeap = true;
                cs.done = true;
                return;
            }
            // last contig-consistent vertex of the running path (PAlgorithm.cpp:332-360)
            uint64_t lastCtgPos = 0;
            uint32_t lastU = 0;  // (its k-mer is looked up on the device together with the next seeds)
            bool haveKmer = false;
            for (auto it = cs.travel.rbegin(); it != cs.travel.rend(); ++it) {
                if (it->ctg != 0) {
                    auto d = mapper.singleToDual(it->ctg);
                    if (d.first == cs.chosenOne && d.second >= 0) {
                        lastCtgPos = (uint64_t)d.second;
                        lastU = it->u;
                        haveKmer = true;
                        break;
                    }
                }
            }
            TravSeedReq r{};
            r.ctg = i;
            r.pos = lastCtgPos;
            r.left = lastCtgPos - std::min<uint64_t>(lastCtgPos, 1000 * deviation);
            r.right = lastCtgPos + 1000 * deviation;
            slot_req[i] = r;
            slot_has[i] = 1;
            cs.seeds.clear();
            cs.parentU = lastU;
            cs.haveParent = haveKmer;
        };
        {
            unsigned nthr = std::min<unsigned>((unsigned)batch.size(), std::max(1u, std::min(32u, std::thread::hardware_concurrency())));
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (size_t x; (x = next.fetch_add(1)) < batch.size();) splice(batch[x]);
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
        }
        for (uint32_t i : batch)
            if (slot_has[i]) {
                reqs.push_back(slot_req[i]);
                req_cs.push_back(i);
            }
        // The round's SEGMENTS: a contig that goes on re-seeds behind the path it has just committed and walks the rest of the
        // strand — through the very checkpoints this round's segments were started from.  They are kept (those still waiting
        // or walking included): the next round's chains adopt them under the conditions of walk_stitch.hpp, which count the
        // marks committed since (Seg::round, MergeCtx::g_*).  A finished contig gives them up.
        for (uint32_t i : batch) {
            if (st[i].done) give_up_segments(i);
            else RS[i].kept = !RS[i].segs.empty();
        }
        lap("splice");
    }
    // next seeds: searchPANode2 + filterPANodes + sort by edit distance + top-K
    int reseed(std::vector<TravSeedReq> &reqs, const std::vector<uint32_t> &req_cs, std::vector<uint32_t> &next_round) {
        int rc;
        if (!reqs.empty()) {
            const uint32_t PARTS = TRAV_SEED_PARTS;
            uint32_t WSTRIDE = 2048;  // words per part of a request
            std::vector<uint32_t> wb;
            for (;;) {  // (a part with more candidates than the stride is searched again with a larger one)
                const size_t words = (size_t)reqs.size() * PARTS * WSTRIDE;
                if ((rc = b_req.alloc(reqs.size() * sizeof(TravSeedReq))) || (rc = b_seedout.alloc(words * 4))) return fail(rc);
                uint32_t *hp = (uint32_t *)pinned(words * 4 + 256);
                if (!hp) return fail(PAG_ENOMEM);
                hipError_t he = hipMemcpyAsync(b_req.p, reqs.data(), reqs.size() * sizeof(TravSeedReq), hipMemcpyHostToDevice, s);
                if (upload_contigs() != PAG_OK) he = hipErrorUnknown;
                trav_launch_seed_window(G, b_tc.as<TravContig>(), b_req.as<TravSeedReq>(), (uint32_t)reqs.size(), deviation,
                                        b_seedout.as<uint32_t>(), WSTRIDE, s);
                if (he == hipSuccess) he = hipMemcpyAsync(hp, b_seedout.p, words * 4, hipMemcpyDeviceToHost, s);
                if (he == hipSuccess) he = hipStreamSynchronize(s);
                if (he != hipSuccess) {
                    set_error("pag_travel: seed search failed: %s", hipGetErrorString(he));
                    return fail(PAG_EFAULT);
                }
                uint32_t most = 0;
                for (size_t q = 0; q < reqs.size() * PARTS; ++q) most = std::max(most, hp[q * WSTRIDE]);
                if (most <= WSTRIDE - 1) {
                    wb.assign(hp, hp + words);
                    break;
                }
                if (most > (1u << 28)) {
                    set_error("pag_travel: seed window with %u candidates", most);
                    return fail(PAG_ENOMEM);
                }
                WSTRIDE = (uint32_t)pow2_at_least((uint64_t)most + 2);
            }
            std::vector<uint32_t> vids;
            std::vector<size_t> cnt(reqs.size());
            for (size_t q = 0; q < reqs.size(); ++q) {
                std::unordered_set<uint32_t> seen;
                size_t n = 0;
                for (uint32_t part = 0; part < PARTS; ++part) {  // (the parts of the window, in offset order)
                    const uint32_t *o = &wb[(q * PARTS + part) * WSTRIDE];
                    for (uint32_t x = 0; x < o[0]; ++x) {
                        uint32_t v = o[1 + x];
                        if (!seen.insert(v).second) continue;         // std::set `unique` in searchPANode2
                        vids.push_back(v);                            // (filterPANodes was applied by the kernel)
                        ++n;
                    }
                }
                cnt[q] = n;
            }
            std::vector<pag_path_node> attrs;
            if ((rc = fetch_vertices(vids, attrs))) return fail(rc);
            {   // k-mers of the parents (last contig-consistent vertex of each running path)
                std::vector<uint32_t> pu(reqs.size());
                for (size_t q = 0; q < reqs.size(); ++q) pu[q] = st[req_cs[q]].parentU;
                std::vector<pag_path_node> pa(reqs.size());
                if ((rc = b_vids.alloc(pu.size() * 8)) || (rc = b_gather.alloc(pu.size() * sizeof(pag_path_node) + 64))) return fail(rc);
                hipMemcpyAsync(b_vids.p, pu.data(), pu.size() * 4, hipMemcpyHostToDevice, s);
                hipMemsetAsync(b_vids.as<uint32_t>() + pu.size(), 0, pu.size() * 4, s);
                trav_launch_gather_path(G, b_vids.as<uint32_t>(), b_vids.as<uint32_t>() + pu.size(), pu.size(), b_gather.as<pag_path_node>(), s);
                hipMemcpyAsync(pa.data(), b_gather.p, pa.size() * sizeof(pag_path_node), hipMemcpyDeviceToHost, s);
                if (hipStreamSynchronize(s) != hipSuccess) {
                    set_error("pag_travel: parent k-mer lookup failed");
                    return fail(PAG_EFAULT);
                }
                for (size_t q = 0; q < reqs.size(); ++q) st[req_cs[q]].parentCode = pa[q].code;
            }
            size_t at = 0;
            for (size_t q = 0; q < reqs.size(); ++q) {
                CtgState &cs = st[req_cs[q]];
                const std::string parent = cs.haveParent ? code2kmer(cs.parentCode, k) : std::string();
                std::vector<pag_path_node> cand(attrs.begin() + at, attrs.begin() + at + cnt[q]);
                at += cnt[q];
                // std::sort with the reference's comparator (edit distance to the parent k-mer), unstable:
                // precomputed keys give the same comparison outcomes, hence the same permutation
                struct Keyed {
                    size_t d;
                    pag_path_node n;
                };
                std::vector<Keyed> keyed;
                keyed.reserve(cand.size());
                for (auto &c : cand) keyed.push_back({edit_distance(parent, code2kmer(c.code, k)), c});
                std::sort(keyed.begin(), keyed.end(), [](const Keyed &a, const Keyed &b) { return a.d < b.d; });
                cs.seeds.clear();
                for (size_t x = 0; x < keyed.size() && x < topK; ++x) cs.seeds.push_back(keyed[x].n);
                if (cs.seeds.empty()) {
                    cs.done = true;
                    give_up_segments(req_cs[q]);
                }
                else next_round.push_back(req_cs[q]);
            }
        }
        return PAG_OK;
    }
    int event_loop() {
        int rc;
        t_progress = now_ms();
        t_first_fin = 0;
        t_last_news = now_ms();
        while (n_live || !over_queue.empty()) {
            std::vector<uint32_t> fin, touched, batch, req_cs, next_round;
            bool again = false;
            if ((rc = poll_finished(fin, again))) return rc;
            if (again) continue;
            auto mark = [&](const char *what) {  // (PAG_WALK_DEBUG: where this thread's time goes, iteration by iteration)
                if (wdebug) std::fprintf(stderr, "[walk] t=%.1f ms loop: %s (%zu jobs, %zu contigs decided, %u live)\n", now_ms() - tw0, what, fin.size(), batch.size(), n_live);
                if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 4u, n_live, 0, 0, (uint64_t)(uintptr_t)what, 0, fin.size(), batch.size()});
            };
            mark("polled");
            std::vector<Got> got;
            if ((rc = fetch_paths(fin, got))) return rc;
            mark("fetched");
            if ((rc = stitch_finished(got, touched))) return rc;
            mark("stitched");
            decide_rounds(touched, batch);
            flush_backlog();
            if (batch.empty()) {
                if ((rc = publish())) return fail(rc);
                mark("published");
                lap("round prep");
                continue;
            }
            // per contig: choose (PAlgorithm.cpp:238-262), commit, splice, stop or re-seed
            std::vector<Pick> picks(n_sel);
            if ((rc = take_walks(batch, picks))) return rc;
            mark("walks taken");
            std::vector<TravSeedReq> reqs;
            splice_batch(batch, picks, reqs, req_cs);
            mark("spliced");
            if ((rc = reseed(reqs, req_cs, next_round))) return rc;
            mark("re-seeded");
            for (uint32_t i : batch)
                if (st[i].done && (rc = deliver_contig(i))) return fail(rc);
            mark("delivered");
            lap("reseed");
            // the follow-up rounds
            if (!next_round.empty() && (rc = start_rounds(next_round))) return fail(rc);
            mark("next rounds posted");
            flush_backlog();
            if ((rc = publish())) return fail(rc);
            mark("published");
            lap("round prep");
        }
        return PAG_OK;
    }
    // the walker is sent home; whatever has not been delivered while the walks ran; statistics
    int finish() {
        int rc;
        shutdown_walker();
        g->defer_free = false;
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
        t_walk = now_ms() - tw0;
        lap("walk");
        if (wtrace) {
            uint64_t d0 = ~0ull;
            for (const TraceEv &e : trace)
                if (e.what == 0u) d0 = std::min(d0, e.a);
            std::fprintf(stderr, "[trace] walks %.2f ms, %zu events; device times relative to the first job's begin\n", t_walk, trace.size());
            for (const TraceEv &e : trace) {
                if (e.what == 0u)
                    std::fprintf(stderr, "[trace] done t=%.2f ctg %u %s %d dev %.2f..%.2f len %llu classify %llu\n", e.t, e.ctg, e.kind ? "seg" : "chain", e.idx, (double)(e.a - d0) * 1e-5,
                                 (double)(e.b - d0) * 1e-5, (unsigned long long)e.len, (unsigned long long)e.classify);
                else if (e.what == 4u)
                    std::fprintf(stderr, "[trace] loop t=%.2f %s jobs %llu decided %llu live %u\n", e.t, (const char *)(uintptr_t)e.a, (unsigned long long)e.len, (unsigned long long)e.classify, e.ctg);
                else if (e.what == 1u)
                    std::fprintf(stderr, "[trace] post t=%.2f ctg %u %s %d ring %llu mode %llu init %llu\n", e.t, e.ctg, e.kind ? "seg" : "chain", e.idx, (unsigned long long)e.a, (unsigned long long)e.b,
                                 (unsigned long long)e.len);
                else
                    std::fprintf(stderr, "[trace] over t=%.2f ctg %u round %llu chain %d len %llu%s\n", e.t, e.ctg, (unsigned long long)e.a, e.idx, (unsigned long long)e.len, e.b ? " leap" : "");
            }
            trace.clear();
        }
        if (timing || wdebug) {
            std::fprintf(stderr, "[timing] stitch: bookkeeping %.1f ms, paths %.1f ms, chains %.1f ms; posting jobs (all callers) %.1f ms; fetch memory: chunk %zu of %zu; walk arena: %.2f of %.2f GB used\n", t_st[0], t_st[1], t_st[2], t_st[3],
                         fetch_chunk + 1, g->fetch_chunks.size(), g->walk_arena_used / 1e9, g->walk_arena_cap / 1e9);
            std::fprintf(stderr, "[timing] leaping zone: %llu segment jobs, %llu adopted, refused by reason (unusable, no junction, not a boundary, cannot leap yet, window top, window bottom, contig-following record, coordinate-free record): %llu %llu %llu %llu %llu %llu %llu %llu\n",
                         (unsigned long long)n_leap_jobs, (unsigned long long)n_leap_adopted.load(), (unsigned long long)n_leap_refused[0].load(), (unsigned long long)n_leap_refused[1].load(), (unsigned long long)n_leap_refused[2].load(),
                         (unsigned long long)n_leap_refused[3].load(), (unsigned long long)n_leap_refused[4].load(), (unsigned long long)n_leap_refused[5].load(), (unsigned long long)n_leap_refused[6].load(), (unsigned long long)n_leap_refused[7].load());
            {
                size_t n_tail = 0, n_tail_behind = 0;
                for (const CtgState &cs : st) n_tail += cs.tail.on, n_tail_behind += cs.tail.on && cs.tail.m0;
                std::fprintf(stderr, "[timing] last rounds put together on the device: %zu of %u contigs (%zu behind an earlier round's path)\n", n_tail, n_sel, n_tail_behind);
            }
            std::fprintf(stderr, "[timing] pieces: %llu segment jobs, %llu resume jobs, %llu vertices adopted, %llu segments not adoptable\n", (unsigned long long)n_seg_jobs,
                         (unsigned long long)n_resume_jobs, (unsigned long long)n_adopted.load(), (unsigned long long)n_merge_fail.load());
        }

        // ---- epilogue: whatever has not been delivered while the walks ran (see deliver_contig)
        for (uint32_t i = 0; i < n_sel; ++i)
            if (!st[i].delivered) filter_travel(st[i]);
        {   // the full records of those sequences: one gather, results straight into the pinned array the handle keeps for
            // pag_travel_path()
            uint64_t tot = 0;
            for (auto &cs : st)
                if (!cs.delivered) tot += cs.travel.size();
            if (g->path_cap < tot + 1) {
                if (g->path_store) hipHostFree(g->path_store);
                g->path_store = nullptr;
                g->path_cap = 0;
                const size_t want = tot + tot / 8 + 1024;
                if (hipHostMalloc((void **)&g->path_store, want * sizeof(pag_path_node), hipHostMallocDefault) != hipSuccess) {
                    set_error("pag_travel: hipHostMalloc for %zu path records failed", want);
                    return fail(PAG_ENOMEM);
                }
                g->path_cap = want;
            }
            uint32_t *hp = (uint32_t *)pinned(tot * 8 + 256);
            if (!hp) return fail(PAG_ENOMEM);
            uint64_t at = 0;
            for (auto &cs : st) {
                if (cs.delivered) continue;
                const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
                g->path_off[slot2] = at;
                g->path_len[slot2] = cs.travel.size();
                g->path_valid[slot2] = 1;
                for (size_t x = 0; x < cs.travel.size(); ++x) {
                    hp[at + x] = cs.travel[x].u;
                    hp[tot + at + x] = (uint32_t)cs.travel[x].step;
                }
                at += cs.travel.size();
            }
            DevBuf b_fin = buf();
            if ((rc = b_fin.alloc(tot * 8 + 64)) || (rc = b_gather.alloc(tot * sizeof(pag_path_node) + 64))) return fail(rc);
            if (tot) {
                PAG_HIP_TRY(hipMemcpyAsync(b_fin.p, hp, tot * 8, hipMemcpyHostToDevice, s));
                trav_launch_gather_path(G, b_fin.as<uint32_t>(), b_fin.as<uint32_t>() + tot, tot, b_gather.as<pag_path_node>(), s);
                PAG_HIP_TRY(hipMemcpyAsync(g->path_store, b_gather.p, tot * sizeof(pag_path_node), hipMemcpyDeviceToHost, s));
            }
            PAG_HIP_TRY(hipStreamSynchronize(s));
            if (g->deliver_stream) PAG_HIP_TRY(hipStreamSynchronize(g->deliver_stream));  // (the deliveries made during the walks)
        }
        lap("epilogue");
        if (timing) {
            std::fprintf(stderr, "[timing] walks redone without speculation: %u\n", respeculated);
            std::fprintf(stderr, "[timing] segment jobs left behind by rounds that were decided without them: %llu\n", (unsigned long long)n_orphans);
            std::fprintf(stderr, "[timing] pag_travel laps:");
            for (auto &l : laps) std::fprintf(stderr, " %s %.1f ms;", l.first, l.second);
            std::fprintf(stderr, "\n");
        }
        if (stats) {
            stats->ms_compact = t_compact;
            stats->ms_walk = t_walk;
            stats->ms_total = now_ms() - t_begin;
            stats->rounds = rounds;
            stats->jobs = jobs_total;
            stats->walk_steps = steps_total;
            stats->classify_calls = classify_total;
            stats->probes = probe_total;
            stats->records = record_total;
        }
        return PAG_OK;
    }
    int run() {
        int rc;
        if ((rc = begin()) || (rc = setup_contigs())) return rc;
        if (n_sel == 0) return PAG_OK;
        if ((rc = first_seeds()) || (rc = setup_rings()) || (rc = reserve_arena()) || (rc = post_first_rounds()) || (rc = event_loop())) return rc;
        return finish();
    }
};

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
