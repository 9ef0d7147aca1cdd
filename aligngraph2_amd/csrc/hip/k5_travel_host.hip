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
    std::vector<std::vector<size_t>> dp(2, std::vector<size_t>(b.size() + 1, 0));
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

// open-addressing set of vertex ids (PAlgorithm's globalUniqueTable on the host side)
struct VidSet {
    std::vector<uint32_t> slot;
    size_t n = 0;
    static uint32_t mix(uint32_t x) {
        x *= 0x9E3779B1u;
        return x ^ (x >> 15);
    }
    bool empty() const { return n == 0; }
    void grow() {
        std::vector<uint32_t> old;
        old.swap(slot);
        slot.assign(old.empty() ? 1024 : old.size() * 4, 0xFFFFFFFFu);
        n = 0;
        for (uint32_t v : old)
            if (v != 0xFFFFFFFFu) insert(v);
    }
    bool insert(uint32_t v) {  // true if newly inserted
        if ((n + 1) * 2 > slot.size()) grow();
        const size_t mask = slot.size() - 1;
        for (size_t h = mix(v) & mask;; h = (h + 1) & mask) {
            if (slot[h] == v) return false;
            if (slot[h] == 0xFFFFFFFFu) {
                slot[h] = v;
                ++n;
                return true;
            }
        }
    }
    bool contains(uint32_t v) const {
        if (slot.empty()) return false;
        const size_t mask = slot.size() - 1;
        for (size_t h = mix(v) & mask;; h = (h + 1) & mask) {
            if (slot[h] == v) return true;
            if (slot[h] == 0xFFFFFFFFu) return false;
        }
    }
};

struct CtgState {
    uint32_t ci = 0;  // contig index
    bool forward = true;
    int64_t chosenOne = 0;
    uint32_t len = 0;
    uint32_t ctgLeft = 0, ctgRight = 0, revLeft = 0, revRight = 0;
    uint64_t nodesOff = 0;  // offset of this contig's node table
    std::vector<pag_path_node> travel;
    std::vector<pag_path_node> seeds;
    int64_t varLen = 0;
    std::deque<uint32_t> ctgQ, refQ;
    bool finalLeap = false, done = false;
    VidSet globalUnique;
    uint32_t gwinLo = 0xFFFFFFFFu, gwinHi = 0;
    uint32_t *gset = nullptr;   // device: global visited, vertices outside the strand's id range
    uint32_t gcap = 0;
    uint32_t *gbits = nullptr;  // device: global visited bitmap over [inLo, inHi)
    uint32_t inLo = 0, inHi = 0;
    uint64_t nOutside = 0;      // entries in gset
    uint64_t seqCap = 0;
    uint32_t parentCode = 0;  // k-mer of the last contig-consistent path vertex (seed ordering key)
    bool haveParent = false;
};

uint64_t pow2_at_least(uint64_t x) {
    uint64_t p = 1024;
    while (p < x) p <<= 1;
    return p;
}

// PAlgorithm::appendSeq (PAlgorithm.cpp:110-142) on path records
int64_t append_seq(std::vector<pag_path_node> &base, const std::vector<pag_path_node> &tail, uint32_t k) {
    if (tail.empty()) return 0;
    int64_t dLen = 0;
    const pag_path_node &head = tail.front();
    int32_t dist = (int32_t)k;
    while (!base.empty() && (base.back().ctg == 0 || head.ctg <= base.back().ctg)) {
        dLen -= base.back().step;
        base.pop_back();
    }
    if (!base.empty()) dist = (int32_t)(head.ctg - base.back().ctg);
    for (auto &n : tail) {
        dLen += n.step;
        base.push_back(n);
    }
    pag_path_node &first = base[base.size() - tail.size()];
    dLen -= first.step - dist;
    first.step = dist;
    return dLen;
}

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
    if (!g || slot >= g->paths.size() || !g->path_valid[slot]) {
        if (len) *len = 0;
        return nullptr;
    }
    if (len) *len = g->paths[slot].size();
    return g->paths[slot].data();
}

const pag_path_node *pag_travel_path(const pag_graph *g, uint64_t ctg_index, uint64_t *len) {
    if (g && 2 * ctg_index + 1 < g->paths.size() && !g->path_valid[2 * ctg_index]) return pag_travel_path_oriented(g, ctg_index, 0, len);
    return pag_travel_path_oriented(g, ctg_index, 1, len);
}

int pag_travel(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
               const pag_travel_params *prm, pag_travel_stats *stats) {
    if (!g || !ctgs || !orient || !prm || (!ref_len && n_refs)) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const double t_begin = now_ms();
    const bool timing = std::getenv("PAGRAPH_TIMING") != nullptr;
    double lap_t = t_begin;
    std::vector<std::pair<const char *, double>> laps;
    auto lap = [&](const char *what) {
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
    };
    const uint32_t k = g->k;
    const uint64_t deviation = prm->deviation;
    const double errorRate = prm->error_rate, startSplit = prm->start_split;
    const size_t topK = std::min<uint32_t>(prm->ref_threads, 8u);
    int rc;
    int slot = TRAV_SLOT0;
    auto buf = [&](void) { return DevBuf(g, slot++); };

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
        (rc = b_vpos.alloc((np + 1) * 8)) || (rc = b_vcnt.alloc((np + 1) * 2)) || (rc = b_vnode.alloc((np + 1) * 4)) ||
        (rc = b_eto.alloc((ne + 1) * 4)) || (rc = b_estep.alloc((ne + 1) * 4)) || (rc = b_bitmap.alloc(n_words * 8)) ||
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
    if (g->tg_ready && (g->tg_dev != deviation || g->tg_err != errorRate)) g->tg_ready = false;
    if (g->tg_ready) {
        G.succ = g->tg.succ;
        G.n_succ = g->tg.n_succ;
    }
    if (!g->tg_ready) {
        const double t0 = now_ms();
        size_t tb = trav_compact_tmp_bytes(g->n_t, g->n_e, k, nn);
        if ((rc = b_ctmp.alloc(tb))) return rc;
        if ((rc = trav_compact(g->tkey, g->tval, g->tseg, g->tcnt, g->n_t, g->ekey, g->eval, g->eseg, g->n_e, k, nn, np, ne, G,
                               b_ctmp.p, tb, s)))
            return rc;
        // coordinate order, then the static half of the epsilon-join for every vertex
        if ((rc = b_ok0.alloc((np + 1) * 4)) || (rc = b_ov0.alloc((np + 4) * 8)) || (rc = b_ok1.alloc((np + 1) * 4)) ||
            (rc = b_ov1.alloc((np + 1) * 8)) || (rc = b_otmp.alloc(std::max(sort_tmp_bytes(np), scan_tmp_bytes(np + 2) + 64))))
            return rc;
        if ((rc = trav_order(G, b_ok0.as<uint32_t>(), b_ov0.as<uint64_t>(), b_ok1.as<uint32_t>(), b_ov1.as<uint64_t>(), b_otmp.p, s)))
            return rc;
        uint64_t n_succ = 0, n_cand = 0;
        // One evaluation of the candidate pairs instead of two when memory allows: the records are first written to a
        // staging array laid out by the candidate-pair bound (b_ok1 = bound per vertex, b_ov1 = its prefix,
        // b_ov0[np + 1] = total), then moved to coordinate order.  Staging = 16 B per CANDIDATE (about twice the
        // records); it reuses the compaction scratch slot.
        const SuccRec *stage = nullptr;
        const uint64_t *stage_off = nullptr;
        // (only attempted for graphs small enough that it can fit: at sequencing coverage the candidates are ~10x the
        // records, see DESIGN.md, and computing the bound is not free)
        if (!std::getenv("PAG_SUCC_TWO_PASS") && np <= (64ull << 20)) {
            if ((rc = trav_succ_bound(G, b_ok1.as<uint32_t>(), b_ov1.as<uint64_t>(), b_otmp.p, b_ov0.as<uint64_t>() + np + 1, s)))
                return rc;
            PAG_HIP_TRY(hipMemcpyAsync(&n_cand, b_ov0.as<uint64_t>() + np + 1, 8, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
            size_t free_b = 0, total_b = 0;
            PAG_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const size_t want = (n_cand + 1) * sizeof(SuccRec);
            // the final array (<= the staging size) has to fit as well; keep a margin for the walk buffers
            if (want <= b_ctmp.sl->cap || want < (free_b + b_ctmp.sl->cap) / 4) {
                if (b_ctmp.alloc(want) == PAG_OK) {
                    stage = b_ctmp.as<SuccRec>();
                    stage_off = b_ov1.as<uint64_t>();
                }
            }
        }
        // two passes: b_ov1 (free then) keeps, per vertex, which of its first 64 candidates the counting pass accepted
        uint64_t *amask = stage ? nullptr : b_ov1.as<uint64_t>();
        // b_ok0 doubles as the per-vertex count array, b_ov0 as the scan output, b_ov0[np + 2] as the total
        if ((rc = trav_succ_count(G, (uint32_t)deviation, errorRate, b_ok0.as<uint32_t>(), b_ov0.as<uint64_t>(), b_otmp.p,
                                  b_ov0.as<uint64_t>() + np + 2, stage_off, const_cast<SuccRec *>(stage), amask, s)))
            return rc;
        PAG_HIP_TRY(hipMemcpyAsync(&n_succ, b_ov0.as<uint64_t>() + np + 2, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        if (n_succ >= 0xFFFFFFF0ull) {
            set_error("pag_travel: more than 2^32 successor records");
            return PAG_EINVAL;
        }
        if ((rc = b_succ.alloc((n_succ + 1) * sizeof(SuccRec)))) return rc;
        G.succ = b_succ.as<SuccRec>();
        G.n_succ = n_succ;
        if ((rc = trav_succ_fill(G, (uint32_t)deviation, errorRate, n_succ, stage_off, stage, amask, s))) return rc;
        PAG_HIP_TRY(hipStreamSynchronize(s));
        g->tg = G;
        g->tg_dev = deviation;
        g->tg_err = errorRate;
        g->tg_ready = true;
        if (std::getenv("PAGRAPH_TIMING"))
            std::fprintf(stderr, "[timing] successor records %llu for %llu vertices (%s, %llu candidate pairs)\n", (unsigned long long)n_succ,
                         (unsigned long long)np, stage ? "staged, one evaluation" : "two passes", (unsigned long long)n_cand);
        t_compact = now_ms() - t0;
    }

    lap("compact");
    // ---- contigs: packed bases, mapper tables, per-strand node tables
    Mapper mapper(ctgs->len, ctgs->n_seqs);
    Mapper refMapper(ref_len, n_refs);
    const uint32_t n_ctgs = (uint32_t)ctgs->n_seqs;
    g->paths.assign(2 * (size_t)n_ctgs, {});
    g->path_valid.assign(2 * (size_t)n_ctgs, 0);
    std::vector<CtgState> st;
    uint64_t nodes_total = 0;
    // one entry per (contig, orientation): a contig selected with both orientations is two independent traversals
    // (PAssembly.cpp:28-36 walks every (name, forward) pair of its set)
    for (uint32_t c2 = 0; c2 < 2 * n_ctgs; ++c2) {
        const uint32_t c = c2 >> 1;
        const bool fwd = (c2 & 1u) == 0;
        const int32_t o = orient[c];
        if (!(o == PAG_ORIENT_BOTH || (fwd && o == PAG_ORIENT_FORWARD) || (!fwd && o == PAG_ORIENT_REVERSE))) continue;
        CtgState cs;
        cs.ci = c;
        if (c2 < g->paths_pool.size()) {  // storage of the previous block's path for this slot
            cs.travel.swap(g->paths_pool[c2]);
            cs.travel.clear();
        }
        cs.forward = fwd;
        cs.chosenOne = cs.forward ? (int64_t)c + 1 : -(int64_t)c - 1;
        cs.len = ctgs->len[c];
        cs.ctgLeft = (uint32_t)mapper.dualToSingle(cs.chosenOne, 0);
        cs.ctgRight = (uint32_t)mapper.dualToSingle(cs.chosenOne, cs.len);
        cs.revLeft = (uint32_t)mapper.dualToSingle(-cs.chosenOne, 0);
        cs.revRight = (uint32_t)mapper.dualToSingle(-cs.chosenOne, cs.len);
        cs.nodesOff = nodes_total;
        cs.seqCap = (uint64_t)cs.len / 2 + 8192;
        if (const char *e = std::getenv("PAG_DEBUG_SEQCAP")) cs.seqCap = std::max(16, std::atoi(e));  // tests: force the overflow / regrow path
        nodes_total += cs.len >= k ? cs.len - k + 1 : 0;
        st.push_back(std::move(cs));
    }
    const uint32_t n_sel = (uint32_t)st.size();
    if (n_sel == 0) return PAG_OK;

    DevBuf b_packed = buf(), b_nodes = buf(), b_starts = buf(), b_sizes = buf(), b_tc = buf(), b_seedout = buf(), b_req = buf(),
           b_gset = buf(), b_gather = buf(), b_vids = buf(), b_gbits = buf();
    if ((rc = b_packed.alloc(ctgs->packed_bytes + 64)) || (rc = b_nodes.alloc((nodes_total + 1) * 4)) ||
        (rc = b_starts.alloc(mapper.starts.size() * 8 + 8)) || (rc = b_sizes.alloc(mapper.sizes.size() * 8 + 8)) ||
        (rc = b_tc.alloc(n_sel * sizeof(TravContig))))
        return rc;
    PAG_HIP_TRY(hipMemcpyAsync(b_packed.p, ctgs->packed, ctgs->packed_bytes, hipMemcpyHostToDevice, s));
    PAG_HIP_TRY(hipMemcpyAsync(b_starts.p, mapper.starts.data(), mapper.starts.size() * 8, hipMemcpyHostToDevice, s));
    PAG_HIP_TRY(hipMemcpyAsync(b_sizes.p, mapper.sizes.data(), mapper.sizes.size() * 8, hipMemcpyHostToDevice, s));
    for (auto &cs : st)
        trav_launch_ctg_nodes(b_packed.as<uint8_t>(), ctgs->byte_off[cs.ci], cs.len, cs.forward ? 1 : 0, k, G,
                              b_nodes.as<uint32_t>() + cs.nodesOff, s);

    std::vector<TravContig> tc(n_sel);
    auto fill_contigs = [&]() {
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
            t.gbits = cs.globalUnique.empty() ? nullptr : cs.gbits;
            t.gset = cs.globalUnique.empty() ? nullptr : cs.gset;
            t.gmask = cs.gcap - 1;
            t.gwin_lo = cs.gwinLo;
            t.gwin_hi = cs.gwinHi;
        }
    };
    auto upload_contigs = [&]() -> int {
        fill_contigs();
        PAG_HIP_TRY(hipMemcpyAsync(b_tc.p, tc.data(), n_sel * sizeof(TravContig), hipMemcpyHostToDevice, s));
        return PAG_OK;
    };
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
    // vertex attributes for a list of vertex ids
    auto fetch_vertices = [&](const std::vector<uint32_t> &vids, std::vector<pag_path_node> &out) -> int {
        out.resize(vids.size());
        if (vids.empty()) return PAG_OK;
        int r;
        if ((r = b_vids.alloc(vids.size() * 4)) || (r = b_gather.alloc(vids.size() * sizeof(pag_path_node)))) return r;
        PAG_HIP_TRY(hipMemcpyAsync(b_vids.p, vids.data(), vids.size() * 4, hipMemcpyHostToDevice, s));
        trav_launch_gather_vertices(G, b_vids.as<uint32_t>(), (uint32_t)vids.size(), b_gather.as<pag_path_node>(), s);
        PAG_HIP_TRY(hipMemcpyAsync(out.data(), b_gather.p, vids.size() * sizeof(pag_path_node), hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        return PAG_OK;
    };

    // ---- round 0 seeds: searchPANode(onlyFirst) then top-K
    const uint32_t SEED_STRIDE = 4096;
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
    uint64_t rounds = 0, jobs_total = 0, steps_total = 0, classify_total = 0, probe_total = 0, record_total = 0;
    double t_walk = 0;

    // ---- the walks.  The contigs are independent state machines (walk the seeds of the round, choose, splice,
    //      stop or re-seed); a persistent walker grid executes whatever jobs are posted, and this loop posts the
    //      next round of a contig as soon as that contig's previous round is done.
    enum { CB_SEQV = 0, CB_SEQS, CB_ARV, CB_ARS, CB_TSET, CB_PSET, CB_STAMP, CB_TBITS, CB_N };
    if (g->cpool.size() < (size_t)n_sel * CB_N) g->cpool.resize((size_t)n_sel * CB_N);
    auto cbuf = [&](uint32_t i, int b) { return DevBuf(g, &g->cpool[(size_t)i * CB_N + b]); };
    const uint32_t QCAP = 8192;
    const size_t q_need = 256 + (size_t)QCAP * (sizeof(TravPosted) + sizeof(TravJobOut) + sizeof(uint32_t)) + 256;
    if (g->wq_bytes < q_need) {
        if (g->wq_host) hipHostFree(g->wq_host);
        g->wq_host = nullptr;
        g->wq_bytes = 0;
        PAG_HIP_TRY(hipHostMalloc(&g->wq_host, q_need, hipHostMallocCoherent | hipHostMallocMapped));
        g->wq_bytes = q_need;
    }
    if (!g->wq_next) PAG_HIP_TRY(hipMalloc((void **)&g->wq_next, 256));
    if (!g->walk_stream) {
        // the resident grid gets a stream of its own priority class: the runtime multiplexes streams onto a few
        // hardware queues, and work of this call's side stream must never be queued behind the walker
        int lo = 0, hi = 0;
        PAG_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PAG_HIP_TRY(hipStreamCreateWithPriority(&g->walk_stream, hipStreamNonBlocking, hi));
    }
    TravQueue *hq = (TravQueue *)g->wq_host;
    TravPosted *hjobs = (TravPosted *)((char *)g->wq_host + 256);
    TravJobOut *houts = (TravJobOut *)(hjobs + QCAP);
    uint32_t *hdone = (uint32_t *)(houts + QCAP);
    std::memset(g->wq_host, 0, q_need);
    PAG_HIP_TRY(hipMemsetAsync(g->wq_next, 0, 256, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));

    struct CRun {
        uint32_t first = 0, n = 0, grow = 1, round = 0;
        bool outstanding = false;
        bool exact = false;  // a speculation of this contig's walk failed once: walk without from now on
    };
    std::vector<CRun> run(n_sel);
    uint32_t n_posted = 0, n_outstanding = 0, respeculated = 0;
    bool walker_up = false;
    auto shutdown_walker = [&]() {
        if (!walker_up) return;
        __atomic_store_n(&hq->exit, 1u, __ATOMIC_RELEASE);
        hipStreamSynchronize(g->walk_stream);
        walker_up = false;
        g->defer_free = false;
        for (void *q : g->deferred) hipFree(q);
        g->deferred.clear();
    };
    auto out_cap = [&](const CtgState &cs, uint32_t grow) { return pow2_at_least((cs.seqCap / 4 + 4096) * grow); };
    // buffers + job records of the next round of contig i (memsets go to stream s; the records become visible
    // to the walker only by publish())
    auto prepare = [&](uint32_t i) -> int {
        CtgState &cs = st[i];
        CRun &R = run[i];
        const size_t ns = cs.seeds.size();
        if (n_posted + ns > QCAP) {
            set_error("pag_travel: more than %u walk jobs", QCAP);
            return PAG_ENOMEM;
        }
        const uint64_t PG = TRAV_PROBE_GROUPS;
        const uint64_t cap = cs.seqCap * R.grow, oc = out_cap(cs, R.grow);
        // one travel epoch / probe stamp per strand vertex; padded to a multiple of four so that the walker's window
        // refills can use 16-byte loads
        const uint64_t span = ((uint64_t)(cs.inHi - cs.inLo) + 1 + 3) & ~3ull, tbw = span + 4;
        DevBuf b_sv = cbuf(i, CB_SEQV), b_ss = cbuf(i, CB_SEQS), b_av = cbuf(i, CB_ARV), b_as = cbuf(i, CB_ARS), b_ts = cbuf(i, CB_TSET),
               b_ps = cbuf(i, CB_PSET), b_st = cbuf(i, CB_STAMP), b_tb = cbuf(i, CB_TBITS);
        int r;
        if ((r = b_sv.alloc(ns * cap * 4)) || (r = b_ss.alloc(ns * cap * 4)) || (r = b_av.alloc(ns * PG * cap * 4)) ||
            (r = b_as.alloc(ns * PG * cap * 4)) || (r = b_ts.alloc(ns * oc * 8)) || (r = b_ps.alloc(ns * PG * oc * 8)) ||
            (r = b_st.alloc(ns * PG * span * 4)) || (r = b_tb.alloc(ns * tbw * 4)))
            return r;
        PAG_HIP_TRY(hipMemsetAsync(b_ts.p, 0xFF, ns * oc * 8, s));
        PAG_HIP_TRY(hipMemsetAsync(b_ps.p, 0, ns * PG * oc * 8, s));
        PAG_HIP_TRY(hipMemsetAsync(b_st.p, 0, ns * PG * span * 4, s));
        PAG_HIP_TRY(hipMemsetAsync(b_tb.p, 0, ns * tbw * 4, s));
        fill_contigs();
        R.first = n_posted;
        R.n = (uint32_t)ns;
        R.outstanding = true;
        R.round += 1;
        for (size_t sd = 0; sd < ns; ++sd) {
            TravPosted &P = hjobs[n_posted + sd];
            TravJob &J = P.J;
            J.ctg = i;
            J.start = cs.seeds[sd].vid;
            J.has_size = (uint64_t)cs.varLen;  // int64 -> size_t conversion as in the reference call
            J.seq_v = b_sv.as<uint32_t>() + sd * cap;
            J.seq_s = b_ss.as<uint32_t>() + sd * cap;
            J.seq_cap = cap;
            J.arena_v = b_av.as<uint32_t>() + sd * PG * cap;
            J.arena_s = b_as.as<uint32_t>() + sd * PG * cap;
            J.arena_cap = PG * cap;
            J.stamp = b_st.as<uint32_t>() + sd * PG * span;
            J.stamp_stride = (uint32_t)span;
            J.tbits = b_tb.as<uint32_t>() + sd * tbw;
            J.tset = b_ts.as<uint64_t>() + sd * oc;
            J.tmask = (uint32_t)oc - 1;
            J.pset = b_ps.as<uint64_t>() + sd * PG * oc;
            J.pmask = (uint32_t)oc - 1;
            J.exact = (R.exact || std::getenv("PAG_WALK_EXACT")) ? 1u : 0u;
            P.C = tc[i];
            hdone[n_posted + sd] = 0;
        }
        n_posted += (uint32_t)ns;
        n_outstanding += 1;
        rounds = std::max<uint64_t>(rounds, R.round);
        jobs_total += ns;
        return PAG_OK;
    };
    auto publish = [&]() -> int {  // after the prepared buffers are ready on the device
        if (std::getenv("PAG_WALK_DEBUG")) std::fprintf(stderr, "[walk] publish: waiting for stream\n");
        PAG_HIP_TRY(hipStreamSynchronize(s));
        if (std::getenv("PAG_WALK_DEBUG")) std::fprintf(stderr, "[walk] publish: posting %u\n", n_posted);
        __atomic_store_n(&hq->posted, n_posted, __ATOMIC_RELEASE);
        return PAG_OK;
    };

    const bool wdebug = std::getenv("PAG_WALK_DEBUG") != nullptr;
    const double tw0 = now_ms();
    g->defer_free = true;
    for (uint32_t i = 0; i < n_sel; ++i)
        if (!st[i].done && (rc = prepare(i))) {
            g->defer_free = false;
            return rc;
        }
    if (n_outstanding) {
        int n_cu = 256;
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, g->device);
        trav_launch_walk_persistent(G, hjobs, houts, hdone, hq, g->wq_next, QCAP, k, (uint32_t)std::max(64, n_cu),
                                    (uint64_t)(std::getenv("PAG_WALK_IDLE_S") ? std::atoi(std::getenv("PAG_WALK_IDLE_S")) : 120) * 2400000000ull, g->walk_stream);
        if (hipGetLastError() != hipSuccess) {
            g->defer_free = false;
            set_error("pag_travel: walker launch failed");
            return PAG_EFAULT;
        }
        if (wdebug) std::fprintf(stderr, "[walk] walker launched, %u jobs prepared\n", n_posted);
        walker_up = true;
        if ((rc = publish())) {
            shutdown_walker();
            return rc;
        }
    } else {
        g->defer_free = false;
    }
    lap("round prep");

    double t_progress = now_ms();
    while (n_outstanding) {
        // contigs whose jobs of the running round are all done
        std::vector<uint32_t> batch;
        for (uint32_t i = 0; i < n_sel; ++i) {
            if (!run[i].outstanding) continue;
            bool all = true;
            for (uint32_t j = 0; j < run[i].n && all; ++j) all = __atomic_load_n(&hdone[run[i].first + j], __ATOMIC_ACQUIRE) != 0;
            if (all) batch.push_back(i);
        }
        if (batch.empty()) {
            if (hipStreamQuery(g->walk_stream) == hipSuccess) {  // the grid is gone although jobs are outstanding
                walker_up = false;
                shutdown_walker();
                g->defer_free = false;
                set_error("pag_travel: the walker stopped with jobs outstanding");
                return PAG_EFAULT;
            }
            if (wdebug && now_ms() - t_progress > 3000.0) {
                static double last = 0;
                if (now_ms() - last > 2000.0) {
                    last = now_ms();
                    uint32_t ticket = 0;
                    hipMemcpyAsync(&ticket, g->wq_next, 4, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    std::fprintf(stderr, "[walk] waiting: posted %u tickets %u done0 %u done1 %u query %d stage0 %llu stage1 %llu\n", n_posted, ticket, hdone[0], hdone[1],
                                 (int)hipStreamQuery(g->walk_stream), (unsigned long long)houts[0].n_main, (unsigned long long)houts[1].n_main);
                }
            }
            if (now_ms() - t_progress > 60000.0) {  // no job finished for a minute: give up instead of hanging
                uint32_t ticket = 0;
                hipMemcpyAsync(&ticket, g->wq_next, 4, hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                shutdown_walker();
                set_error("pag_travel: no walk job finished within 60 s (posted %u, tickets taken %u, contigs waiting %u)", n_posted, ticket,
                          n_outstanding);
                return PAG_EFAULT;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
            continue;
        }
        t_progress = now_ms();
        if (wdebug)
            for (uint32_t i : batch) {
                uint64_t mx = 0;
                for (uint32_t j = 0; j < run[i].n; ++j) mx = std::max<uint64_t>(mx, houts[run[i].first + j].n_classify);
                std::fprintf(stderr, "[walk] t=%.1f ms contig %u round %u done (%u jobs, max classify %llu), %u still walking\n", now_ms() - tw0, i, run[i].round, run[i].n,
                             (unsigned long long)mx, n_outstanding - (uint32_t)batch.size());
#ifdef PAG_WALK_PROF
                for (uint32_t j = 0; j < run[i].n; ++j) {
                    const TravJobOut &o = houts[run[i].first + j];
                    if (o.n_classify != mx) continue;
                    static const char *nm[12] = {"append", "classify", "wait", "setup", "steps", "choice", "fill", "probe_wave", "st.window", "st.eval", "st.ballot", "st.update"};
                    std::fprintf(stderr, "[prof] contig %u main %llu fills %llu:", i, (unsigned long long)o.n_main, (unsigned long long)(o.n_fill & 0xFFFFFFFFu));
                    for (int q = 0; q < 12; ++q) std::fprintf(stderr, " %s %.1f Mcyc /%u", nm[q], o.prof_t[q] * 1e-6, o.prof_c[q]);
                    std::fprintf(stderr, "\n");
                    break;
                }
#endif
            }
        lap("walk");
        std::vector<uint32_t> redo, next_round;
        for (uint32_t i : batch) {
            run[i].outstanding = false;
            n_outstanding -= 1;
            bool overflow = false, misspec = false;
            for (uint32_t j = 0; j < run[i].n; ++j) {
                overflow |= (houts[run[i].first + j].overflow & 3) != 0;
                misspec |= (houts[run[i].first + j].overflow & 4) != 0;
            }
            if (misspec && !overflow) {  // a zombie probe leapt: the round is walked again, every probe to its end
                if (wdebug)
                    std::fprintf(stderr, "[walk] contig %u: speculation failed (cause bits %llu), exact walk\n", i,
                                 (unsigned long long)(houts[run[i].first].n_fill >> 32));
                run[i].exact = true;
                run[i].round -= 1;
                jobs_total -= run[i].n;
                ++respeculated;
                redo.push_back(i);
                continue;
            }
            if (overflow) {
                if (misspec) run[i].exact = true;
                if (run[i].grow >= 64) {
                    shutdown_walker();
                    set_error("pag_travel: walker buffers overflow even at 64x capacity");
                    return PAG_ENOMEM;
                }
                run[i].grow *= 2;
                run[i].round -= 1;
                jobs_total -= run[i].n;
                redo.push_back(i);
            }
        }
        batch.erase(std::remove_if(batch.begin(), batch.end(), [&](uint32_t i) { return std::find(redo.begin(), redo.end(), i) != redo.end(); }),
                    batch.end());
        for (uint32_t i : batch)
            for (uint32_t j = 0; j < run[i].n; ++j) {
                const TravJobOut &o = houts[run[i].first + j];
                steps_total += o.seq_len;
                classify_total += o.n_classify;
                probe_total += o.n_probe;
                record_total += o.n_records;
            }

        // ---- per contig: choose (PAlgorithm.cpp:238-262); the chosen walks are gathered and committed on the
        //      device back to back, copied out, and spliced by a pool of host threads (contigs are independent)
        struct Pick {
            int chosen = -1;
            bool leap = false;
            size_t j = 0, chooseCtgPos = 0, chooseRefPos = 0;
            uint64_t off = 0, len = 0;
        };
        std::vector<Pick> picks(n_sel);
        {
            uint64_t tot = 0;
            for (uint32_t i : batch) {
                CtgState &cs = st[i];
                Pick &P = picks[i];
                const size_t ns = cs.seeds.size(), j0 = run[i].first;
                size_t maxLen = 0;
                for (size_t sd = 0; sd < ns; ++sd) {
                    const TravJobOut &o = houts[j0 + sd];
                    size_t len = o.seq_size;
                    P.leap = o.last_ctg != 0 && mapper.singleToDual(o.last_ctg).first != cs.chosenOne;
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
                    P.j = j0 + (size_t)P.chosen;
                    P.off = tot;
                    P.len = houts[P.j].seq_len;
                    tot += P.len;
                }
            }
            if ((rc = b_gather.alloc(tot * sizeof(pag_path_node) + 64))) {
                shutdown_walker();
                return rc;
            }
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0) continue;
                trav_launch_gather_path(G, hjobs[P.j].J.seq_v, hjobs[P.j].J.seq_s, P.len, b_gather.as<pag_path_node>() + P.off, s);
                // record the walk in the device-side global visited set of this contig
                trav_launch_commit(hjobs[P.j].J.seq_v, P.len, st[i].inLo, st[i].inHi, st[i].gbits, st[i].gset, st[i].gcap - 1, s);
            }
        }
        std::vector<std::vector<pag_path_node>> longest(n_sel);
        for (uint32_t i : batch) {
            const Pick &P = picks[i];
            if (P.chosen < 0 || P.len == 0) continue;
            longest[i].resize(P.len);
            if (hipMemcpyAsync(longest[i].data(), b_gather.as<pag_path_node>() + P.off, P.len * sizeof(pag_path_node), hipMemcpyDeviceToHost,
                               s) != hipSuccess) {
                shutdown_walker();
                set_error("pag_travel: path copy failed");
                return PAG_EFAULT;
            }
        }
        if (hipStreamSynchronize(s) != hipSuccess) {
            shutdown_walker();
            set_error("pag_travel: stream failure while gathering paths");
            return PAG_EFAULT;
        }
        lap("choose+gather");

        // splice + stop rules (PAlgorithm.cpp:264-360)
        std::vector<TravSeedReq> slot_req(n_sel);
        std::vector<uint8_t> slot_has(n_sel, 0), slot_full(n_sel, 0);
        auto splice = [&](uint32_t i) {
            CtgState &cs = st[i];
            const Pick &P = picks[i];
            const bool leap = P.leap;
            cs.varLen += append_seq(cs.travel, longest[i], k);
            if (P.chooseCtgPos != 0) {
                cs.ctgQ.push_back((uint32_t)P.chooseCtgPos);
                while (cs.ctgQ.size() > 4) cs.ctgQ.pop_front();
            }
            if (P.chooseRefPos != 0) {
                cs.refQ.push_back((uint32_t)P.chooseRefPos);
                while (cs.refQ.size() > 4) cs.refQ.pop_front();
            }
            for (auto &n : longest[i]) {
                if (cs.globalUnique.insert(n.vid) && (n.ctg < cs.ctgLeft || n.ctg >= cs.ctgRight)) ++cs.nOutside;
                if (n.ctg != 0) {
                    cs.gwinLo = std::min(cs.gwinLo, n.ctg);
                    cs.gwinHi = std::max(cs.gwinHi, n.ctg);
                }
            }
            std::vector<pag_path_node>().swap(longest[i]);
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
            uint32_t lastCode = 0;
            bool haveKmer = false;
            for (auto it = cs.travel.rbegin(); it != cs.travel.rend(); ++it) {
                if (it->ctg != 0) {
                    auto d = mapper.singleToDual(it->ctg);
                    if (d.first == cs.chosenOne && d.second >= 0) {
                        lastCtgPos = (uint64_t)d.second;
                        lastCode = it->code;
                        haveKmer = true;
                        break;
                    }
                }
            }
            if (cs.nOutside * 2 > cs.gcap) {
                slot_full[i] = 1;
                return;
            }
            TravSeedReq r{};
            r.ctg = i;
            r.pos = lastCtgPos;
            r.left = lastCtgPos - std::min<uint64_t>(lastCtgPos, 1000 * deviation);
            r.right = lastCtgPos + 1000 * deviation;
            slot_req[i] = r;
            slot_has[i] = 1;
            cs.seeds.clear();
            cs.parentCode = lastCode;
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
        std::vector<TravSeedReq> reqs;
        std::vector<uint32_t> req_cs;
        for (uint32_t i : batch) {
            if (slot_full[i]) {
                shutdown_walker();
                set_error("pag_travel: global visited set of contig %u is full", st[i].ci);
                return PAG_ENOMEM;
            }
            if (slot_has[i]) {
                reqs.push_back(slot_req[i]);
                req_cs.push_back(i);
            }
        }
        lap("splice");

        // ---- next seeds: searchPANode2 + filterPANodes + sort by edit distance + top-K
        if (!reqs.empty()) {
            const uint32_t WSTRIDE = 16384;
            if ((rc = b_req.alloc(reqs.size() * sizeof(TravSeedReq))) || (rc = b_seedout.alloc((uint64_t)reqs.size() * WSTRIDE * 4))) {
                shutdown_walker();
                return rc;
            }
            std::vector<uint32_t> wb((size_t)reqs.size() * WSTRIDE);
            hipError_t he = hipMemcpyAsync(b_req.p, reqs.data(), reqs.size() * sizeof(TravSeedReq), hipMemcpyHostToDevice, s);
            trav_launch_seed_window(G, b_tc.as<TravContig>(), b_req.as<TravSeedReq>(), (uint32_t)reqs.size(), deviation,
                                    b_seedout.as<uint32_t>(), WSTRIDE, s);
            if (he == hipSuccess) he = hipMemcpyAsync(wb.data(), b_seedout.p, wb.size() * 4, hipMemcpyDeviceToHost, s);
            if (he == hipSuccess) he = hipStreamSynchronize(s);
            if (he != hipSuccess) {
                shutdown_walker();
                set_error("pag_travel: seed search failed: %s", hipGetErrorString(he));
                return PAG_EFAULT;
            }
            std::vector<uint32_t> vids;
            std::vector<size_t> cnt(reqs.size());
            for (size_t q = 0; q < reqs.size(); ++q) {
                CtgState &cs = st[req_cs[q]];
                const uint32_t *o = &wb[q * WSTRIDE];
                if (o[0] > WSTRIDE - 1) {
                    shutdown_walker();
                    set_error("pag_travel: seed window overflow (%u candidates)", o[0]);
                    return PAG_ENOMEM;
                }
                std::unordered_set<uint32_t> seen;
                size_t n = 0;
                for (uint32_t x = 0; x < o[0]; ++x) {
                    uint32_t v = o[1 + x];
                    if (!seen.insert(v).second) continue;         // std::set `unique` in searchPANode2
                    if (cs.globalUnique.contains(v)) continue;     // filterPANodes
                    vids.push_back(v);
                    ++n;
                }
                cnt[q] = n;
            }
            std::vector<pag_path_node> attrs;
            if ((rc = fetch_vertices(vids, attrs))) {
                shutdown_walker();
                return rc;
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
                if (cs.seeds.empty()) cs.done = true;
                else next_round.push_back(req_cs[q]);
            }
        }
        lap("reseed");
        // ---- post the follow-up rounds (and the repeats with larger buffers)
        for (uint32_t i : redo) next_round.push_back(i);
        for (uint32_t i : next_round)
            if ((rc = prepare(i))) {
                shutdown_walker();
                return rc;
            }
        if (!next_round.empty() && (rc = publish())) {
            shutdown_walker();
            return rc;
        }
        lap("round prep");
    }
    shutdown_walker();
    t_walk = now_ms() - tw0;
    lap("walk");

    // ---- epilogue per contig: filterSequence / "Pump it" (PAlgorithm.cpp:409-423)
    for (auto &cs : st) {
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
            auto d = mapper.singleToDual(seq.back().ctg);
            uint64_t a = (uint64_t)std::llabs(d.first);
            if (a == (uint64_t)cs.ci + 1 || (a >= 1 && a <= mapper.sizes.size() &&
                                             (double)d.second >= (double)mapper.sizes[a - 1] * (1 - startSplit)))
                seq.pop_back();
        }
        g->paths[2 * (size_t)cs.ci + (cs.forward ? 0 : 1)] = std::move(seq);
        g->path_valid[2 * (size_t)cs.ci + (cs.forward ? 0 : 1)] = 1;
    }
    lap("epilogue");
    if (timing) {
        std::fprintf(stderr, "[timing] walks redone without speculation: %u\n", respeculated);
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

}  // extern "C"
