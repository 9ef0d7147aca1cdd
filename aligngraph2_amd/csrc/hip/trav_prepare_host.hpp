// The traversal graph of a handle, host side (included by k5_travel_host.hip, inside its anonymous namespace): the coordinate
// mapper and small helpers of the per-round control, the view of a handle's traversals (trav_view_region) and trav_prepare_graph —
// compact CSR, coordinate order, successor records (kernels: k5_view.hip, k5_succ.hip).
#pragma once

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

