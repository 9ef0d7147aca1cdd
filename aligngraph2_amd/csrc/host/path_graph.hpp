// Turns the paths returned by the device traversal (pag_travel) into the (graph view, travel sequences)
// pair that the chain selection / seqToString / writers of assembly.cpp work on: a HostGraph that
// contains exactly the vertices that lie on some path.  Sort-based (paths hold millions of vertices).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "host_graph.hpp"
#include "traversal.hpp"

namespace pagh {

inline void buildPathGraph(const std::vector<std::vector<pag_path_node>> &paths, const std::vector<int> &orient, unsigned k,
                           HostGraph &graph, std::vector<TravelSequence> &results) {
    struct Ref {
        std::uint32_t code, vid;
        const pag_path_node *n;
    };
    std::vector<Ref> all;
    std::size_t total = 0;
    for (auto &p : paths) total += p.size();
    all.reserve(total);
    for (auto &p : paths)
        for (auto &n : p) all.push_back({n.code, n.vid, &n});
    std::sort(all.begin(), all.end(), [](const Ref &a, const Ref &b) { return a.code != b.code ? a.code < b.code : a.vid < b.vid; });
    all.erase(std::unique(all.begin(), all.end(), [](const Ref &a, const Ref &b) { return a.vid == b.vid && a.code == b.code; }),
              all.end());

    std::size_t nNodes = 0;
    for (std::size_t i = 0; i < all.size(); ++i)
        if (i == 0 || all[i].code != all[i - 1].code) ++nNodes;
    graph.resize(nNodes, all.size(), 0);
    graph.k = k;
    // (vid -> vertex) lookup table, sorted by vid
    std::vector<std::pair<std::uint32_t, Vertex>> where(all.size());
    std::size_t ni = 0;
    std::uint32_t pi = 0;
    for (std::size_t i = 0; i < all.size(); ++i) {
        if (i == 0 || all[i].code != all[i - 1].code) {
            if (i != 0) ++ni;
            graph.nodeCode[ni] = all[i].code;
            graph.posOff[ni] = i;
            pi = 0;
        }
        graph.posCtg[i] = all[i].n->ctg;
        graph.posRef[i] = all[i].n->ref;
        graph.posCnt[i] = all[i].n->cnt;
        where[i] = {all[i].vid, Vertex{static_cast<std::uint32_t>(ni), pi++}};
    }
    if (nNodes) graph.posOff[nNodes] = all.size();
    std::sort(where.begin(), where.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    auto find = [&](std::uint32_t vid) {
        auto it = std::lower_bound(where.begin(), where.end(), vid, [](const auto &a, std::uint32_t v) { return a.first < v; });
        return it->second;
    };
    results.assign(paths.size() * 2, {});
    for (std::size_t c = 0; c < paths.size(); ++c) {
        if (c >= orient.size() || orient[c] < 0) continue;
        auto &res = results[2 * c + (orient[c] ? 0 : 1)];
        res.reserve(paths[c].size());
        for (auto &n : paths[c]) res.emplace_back(find(n.vid), n.step);
    }
}

}  // namespace pagh
