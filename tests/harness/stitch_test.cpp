// tests/harness/stitch_test.cpp — TEST SUPPORT (never part of the product).
// C wrappers over the host bookkeeping of a walk cut into pieces (aligngraph2_amd/csrc/hip/walk_stitch.hpp) so that
// tests/test_walk_stitch.py can drive the adoption conditions on recorded / constructed job outputs without a GPU.
#include <cstdint>
#include <vector>

#include "walk_stitch.hpp"

using namespace pagdev::stitch;

namespace {
// pagt_use_tables(1): the paths get block tables (walk_stitch.hpp: what the pack kernel attaches to a fetched path) — T's parts
// as stretches of ONE fetched path (the flat arrays), so that parts begin in the middle of blocks
bool g_tables = false;
struct Tables {
    std::vector<uint32_t> agg, xagg;
    const uint32_t *a(const uint32_t *v, const uint32_t *s, const uint32_t *pc, size_t n) {
        if (!g_tables) return nullptr;
        agg.assign(agg_blocks(n) * AGG_WORDS + 1, 0);
        build_block_aggs(v, s, pc, n, agg.data());
        return agg.data();
    }
    const uint32_t *x(const uint32_t *xl, const uint32_t *xh, size_t n) {
        if (!g_tables || !xl) return nullptr;
        xagg.assign(agg_blocks(n) * AGG_XWORDS + 1, 0);
        build_block_xaggs(xl, xh, n, xagg.data());
        return xagg.data();
    }
};
void build_chain(Chain &ch, Tables &tt, const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts) {
    const uint32_t *ta = tt.a(tv, ts, tpc, (size_t)part_off[n_parts]);
    for (uint32_t p = 0; p < n_parts; ++p)
        extend_chain(ch, tv + part_off[p], ts + part_off[p], tpc + part_off[p], (size_t)(part_off[p + 1] - part_off[p]), nullptr, ta, (size_t)part_off[p]);
}
}  // namespace

extern "C" {

void pagt_use_tables(int on) { g_tables = on != 0; }

// the reductions over entries [i, j) of a path, with or without its block table: out = mx, m0, lo_all, lo_nz, sz
void pagt_range_agg(const uint32_t *v, const uint32_t *s, const uint32_t *pc, uint64_t n, uint64_t i, uint64_t j, uint64_t *out) {
    Tables tt;
    const RangeAgg r = range_agg(v, s, pc, tt.a(v, s, pc, (size_t)n), (size_t)i, (size_t)j);
    out[0] = r.mx, out[1] = r.m0, out[2] = r.lo_all, out[3] = r.lo_nz, out[4] = r.sz;
}
void pagt_range_xagg(const uint32_t *xl, const uint32_t *xh, uint64_t n, uint64_t i, uint64_t j, uint32_t *out) {
    Tables tt;
    range_xagg(xl, xh, tt.x(xl, xh, (size_t)n), (size_t)i, (size_t)j, out, out + 1);
}

// A chain T made of `n_parts` parts (part p = vertices [part_off[p], part_off[p + 1]) of the flat arrays) meets a segment P.
// leap = 0: try_merge, 1: try_merge_leap (xl / xh: the iteration log).  out[0] = decision, out[1] = vertices adopted,
// out[2] = refusal reason (leap), out[3] = chain length afterwards, out[4] = chain size (sum of steps) afterwards,
// out[5] = mx_all afterwards.  tail_out (may be null): the adopted vertices.
// kept: {segment's round, the running round, g_lo, g_hi, g_free_hi} (null: a segment of the running round)
int pagt_stitch_kept(const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts, const uint32_t *pv,
                     const uint32_t *ps, const uint32_t *ppc, const uint32_t *pxl, const uint32_t *pxh, uint64_t pn, uint32_t max_back, uint32_t max_chosen,
                     uint64_t max_probe, uint32_t wd_below_max, uint32_t wd_forced_min, int usable, uint32_t k, uint64_t deviation, uint64_t split,
                     uint64_t has_size, int leap, uint64_t *out, uint32_t *tail_out, const uint32_t *kept);
int pagt_stitch(const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts, const uint32_t *pv,
                const uint32_t *ps, const uint32_t *ppc, const uint32_t *pxl, const uint32_t *pxh, uint64_t pn, uint32_t max_back, uint32_t max_chosen,
                uint64_t max_probe, uint32_t wd_below_max, uint32_t wd_forced_min, int usable, uint32_t k, uint64_t deviation, uint64_t split,
                uint64_t has_size, int leap, uint64_t *out, uint32_t *tail_out) {
    return pagt_stitch_kept(tv, ts, tpc, part_off, n_parts, pv, ps, ppc, pxl, pxh, pn, max_back, max_chosen, max_probe, wd_below_max, wd_forced_min, usable, k,
                            deviation, split, has_size, leap, out, tail_out, nullptr);
}
int pagt_stitch_kept(const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts, const uint32_t *pv,
                     const uint32_t *ps, const uint32_t *ppc, const uint32_t *pxl, const uint32_t *pxh, uint64_t pn, uint32_t max_back, uint32_t max_chosen,
                     uint64_t max_probe, uint32_t wd_below_max, uint32_t wd_forced_min, int usable, uint32_t k, uint64_t deviation, uint64_t split,
                     uint64_t has_size, int leap, uint64_t *out, uint32_t *tail_out, const uint32_t *kept) {
    Chain ch;
    Tables tt, tp, tx;
    build_chain(ch, tt, tv, ts, tpc, part_off, n_parts);
    Seg sg;
    sg.P.agg = tp.a(pv, ps, ppc, (size_t)pn);
    sg.P.xagg = tx.x(pxl, pxh, (size_t)pn);
    sg.P.v = pv;
    sg.P.s = ps;
    sg.P.pc = ppc;
    sg.P.xl = pxl;
    sg.P.xh = pxh;
    sg.P.n = (size_t)pn;
    sg.usable = usable != 0;
    sg.done = true;
    sg.leap = leap != 0;
    sg.max_back = max_back;
    sg.max_chosen = max_chosen;
    sg.max_probe = max_probe;
    sg.wd_below_max = wd_below_max;
    sg.wd_forced_min = wd_forced_min;
    MergeCtx M;
    M.k = k;
    M.deviation = deviation;
    M.split = split;
    M.has_size = has_size;
    if (kept) {
        sg.round = kept[0];
        M.round = kept[1];
        M.g_lo = kept[2];
        M.g_hi = kept[3];
        M.g_free_hi = kept[4];
    }
    const size_t before = ch.len;
    uint64_t adopted = 0;
    int why = -1;
    const int m = leap ? try_merge_leap(M, ch, sg, &adopted, &why) : try_merge(M, ch, sg, &adopted);
    out[0] = (uint64_t)m;
    out[1] = adopted;
    out[2] = (uint64_t)(int64_t)why;
    out[3] = ch.len;
    out[4] = ch.size;
    out[5] = ch.mx_all;
    if (tail_out && ch.len > before) {
        std::vector<uint32_t> flat(ch.len);
        flatten_chain(ch, flat.data(), nullptr, nullptr);
        for (size_t x = before; x < ch.len; ++x) tail_out[x - before] = flat[x];
    }
    return 0;
}

// advance_chain: a round with n_seg segments (the first n_spec cannot leap; segment q's path = entries [seg_off[q], seg_off[q + 1]) of
// the flat arrays, max_back 0 / max_chosen 2 / max_probe 50 for all) and one chain.  out: what (0 nothing, 1 resume), stop,
// until_leap, chain length, next_seg, next_leap, waiting_seg, final, vertices adopted, merges failed, leap pieces adopted,
// last leap refusal reason.
int pagt_advance(uint32_t n_seg, uint32_t n_spec, uint32_t zone_end, const uint32_t *seg_x, const uint32_t *seg_done, const uint32_t *seg_stopped,
                 const uint32_t *seg_usable, const uint64_t *seg_off, const uint32_t *pv, const uint32_t *ps, const uint32_t *ppc, const uint32_t *pxl,
                 const uint32_t *pxh, const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts, uint32_t k,
                 uint64_t deviation, uint64_t split, uint64_t has_size, uint64_t seg_ov, int64_t *out) {
    RoundState R;
    R.n_spec = n_spec;
    R.zone_end = zone_end;
    std::vector<Tables> tabs(2 * (size_t)n_seg + 1);
    for (uint32_t q = 0; q < n_seg; ++q) {
        Seg sg;
        sg.x = seg_x[q];
        sg.leap = q >= n_spec;
        sg.done = seg_done[q] != 0;
        sg.stopped = seg_stopped[q] != 0;
        sg.usable = seg_usable[q] != 0;
        const size_t o = (size_t)seg_off[q], n = (size_t)(seg_off[q + 1] - seg_off[q]);
        sg.P.v = pv + o;
        sg.P.s = ps + o;
        sg.P.pc = ppc + o;
        sg.P.n = n;
        if (sg.leap && pxl) {
            sg.P.xl = pxl + o;
            sg.P.xh = pxh + o;
            sg.P.xagg = tabs[2 * q + 1].x(sg.P.xl, sg.P.xh, n);
        }
        sg.P.agg = tabs[2 * q].a(sg.P.v, sg.P.s, sg.P.pc, n);
        sg.max_back = 0;
        sg.max_chosen = 2;
        sg.max_probe = 50;
        R.segs.push_back(sg);
    }
    R.chains.assign(1, Chain{});
    Chain &ch = R.chains[0];
    build_chain(ch, tabs[2 * (size_t)n_seg], tv, ts, tpc, part_off, n_parts);
    MergeCtx M;
    M.k = k;
    M.deviation = deviation;
    M.split = split;
    M.has_size = has_size;
    AdvanceStats S;
    int why = -1;
    const Next nx = advance_chain(R, ch, M, seg_ov, S, &why);
    out[0] = nx.what == Next::Resume ? 1 : 0;
    out[1] = nx.stop;
    out[2] = nx.until_leap ? 1 : 0;
    out[3] = (int64_t)ch.len;
    out[4] = ch.next_seg;
    out[5] = ch.next_leap;
    out[6] = ch.waiting_seg;
    out[7] = ch.final ? 1 : 0;
    out[8] = (int64_t)S.adopted;
    out[9] = (int64_t)S.merge_fail;
    out[10] = (int64_t)S.leap_adopted;
    out[11] = why;
    return 0;
}

// chain_before / common_back on their own
void pagt_chain_before(const uint32_t *tv, const uint32_t *ts, const uint32_t *tpc, const uint64_t *part_off, uint32_t n_parts, uint64_t idx, uint32_t *mx,
                       uint32_t *m0) {
    Chain ch;
    Tables tt;
    build_chain(ch, tt, tv, ts, tpc, part_off, n_parts);
    chain_before(ch, (size_t)idx, mx, m0);
}

}  // extern "C"
