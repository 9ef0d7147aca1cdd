#include "shard_plan.hpp"

#include <algorithm>
#include <set>

namespace pagh {

namespace {
// PositionMapper layout (position/PositionMapper.cpp:16-31): start of every sequence's forward strand
std::vector<std::uint64_t> starts(const std::uint32_t *len, std::uint64_t n) {
    std::vector<std::uint64_t> st;
    for (std::uint64_t i = 0; i < n; ++i) st.push_back(i == 0 ? len[0] : st.back() + 3ull * len[i - 1] + std::max<std::uint64_t>(len[i - 1], len[i]));
    return st;
}
std::vector<std::pair<std::uint64_t, std::uint64_t>> merged(std::vector<std::pair<std::uint64_t, std::uint64_t>> iv) {
    std::sort(iv.begin(), iv.end());
    std::vector<std::pair<std::uint64_t, std::uint64_t>> out;
    for (auto &x : iv) {
        if (x.second <= x.first) continue;
        if (!out.empty() && x.first <= out.back().second) out.back().second = std::max(out.back().second, x.second);
        else out.push_back(x);
    }
    return out;
}
}  // namespace

ShardPlan planShards(const RawInput &raw, const std::vector<std::int32_t> &orient, unsigned world, std::uint64_t halo, double startSplit) {
    const pag_raw_input &in = raw.view();
    const std::uint64_t nCtg = in.n_ctgs, nRef = in.n_refs;
    const auto cst = starts(in.ctg_len, nCtg), rst = starts(in.ref_len, nRef);
    ShardPlan P;
    P.deal.assign(world, {});
    P.ownerOf.assign(nCtg, -1);
    // where every contig maps to (all listed contig->reference alignments onto known references, whatever their strand:
    // a superset costs memory, never correctness)
    struct Aln {
        std::uint64_t ref, tb, te;
    };
    std::vector<std::vector<Aln>> alns(nCtg);
    std::vector<std::uint64_t> refBegin(nCtg, ~0ull);
    for (std::uint64_t i = 0; i < in.ctg_to_ref.n; ++i) {
        const pag_raw_aln &r = in.ctg_to_ref.rec[i];
        if (r.query == PAG_NONE || r.query >= nCtg || r.target == PAG_NONE || r.target >= nRef) continue;
        alns[r.query].push_back(Aln{r.target, r.t_begin, r.t_end});
        refBegin[r.query] = std::min(refBegin[r.query], rst[r.target] + r.t_begin);
    }
    // contiguous runs along the reference, balanced by length: a contig goes to the rank in whose share of the total its
    // middle falls
    std::vector<std::size_t> order;
    double total = 0;
    for (std::uint64_t c = 0; c < nCtg; ++c)
        if (orient[c] != PAG_ORIENT_NONE) {
            order.push_back(c);
            total += in.ctg_len[c];
        }
    std::sort(order.begin(), order.end(), [&](std::size_t a, std::size_t b) { return refBegin[a] != refBegin[b] ? refBegin[a] < refBegin[b] : a < b; });
    double acc = 0;
    for (std::size_t c : order) {
        const double mid = acc + in.ctg_len[c] / 2.0;
        const unsigned r = std::min<unsigned>(world - 1, total > 0 ? static_cast<unsigned>(mid / total * world) : 0u);
        P.deal[r].push_back(c);
        P.ownerOf[c] = static_cast<int>(r);
        acc += in.ctg_len[c];
    }
    // landing zones of every contig strand: offset <= len * (1 - startSplit) survives the leap rule
    const double leapMin = 1.0 - startSplit;
    std::vector<std::pair<std::uint64_t, std::uint64_t>> landing;
    for (std::uint64_t c = 0; c < nCtg; ++c) {
        const std::uint64_t n = in.ctg_len[c];
        const std::uint64_t z = std::min<std::uint64_t>(static_cast<std::uint64_t>(static_cast<double>(n) * leapMin) + 2, n);
        landing.emplace_back(cst[c], cst[c] + z);
        landing.emplace_back(cst[c] + 2 * n, cst[c] + 2 * n + z);
    }
    std::set<std::uint64_t> refEnds;
    for (std::uint64_t i = 0; i < nRef; ++i) {
        refEnds.insert(rst[i]);
        refEnds.insert(rst[i] + in.ref_len[i]);
    }
    P.ctgIv.resize(world);
    P.refIv.resize(world);
    P.refOpen.resize(world);
    for (unsigned r = 0; r < world; ++r) {
        auto civ = landing;
        std::vector<std::pair<std::uint64_t, std::uint64_t>> riv;
        for (std::size_t c : P.deal[r]) {
            const std::uint64_t n = in.ctg_len[c];
            if (orient[c] == PAG_ORIENT_FORWARD || orient[c] == PAG_ORIENT_BOTH) civ.emplace_back(cst[c], cst[c] + n);
            if (orient[c] == PAG_ORIENT_REVERSE || orient[c] == PAG_ORIENT_BOTH) civ.emplace_back(cst[c] + 2 * n, cst[c] + 3 * n);
            for (const Aln &a : alns[c]) {
                const std::uint64_t lo = rst[a.ref], hi = rst[a.ref] + in.ref_len[a.ref];
                const std::uint64_t b = rst[a.ref] + a.tb, e = rst[a.ref] + a.te;
                riv.emplace_back(b > lo + halo ? b - halo : lo, std::min(hi, e + halo));
            }
        }
        for (auto &x : merged(civ)) {
            P.ctgIv[r].push_back(static_cast<std::uint32_t>(x.first));
            P.ctgIv[r].push_back(static_cast<std::uint32_t>(x.second));
        }
        for (auto &x : merged(riv)) {
            P.refIv[r].push_back(static_cast<std::uint32_t>(x.first));
            P.refIv[r].push_back(static_cast<std::uint32_t>(x.second));
            P.refOpen[r].push_back(refEnds.count(x.first) ? 0 : 1);
            P.refOpen[r].push_back(refEnds.count(x.second) ? 0 : 1);
        }
    }
    for (unsigned r = 0; r < world; ++r)
        P.regions.push_back(pag_region{P.ctgIv[r].size() / 2, P.ctgIv[r].data(), P.refIv[r].size() / 2, P.refIv[r].data(), P.refOpen[r].data()});
    return P;
}

}  // namespace pagh
