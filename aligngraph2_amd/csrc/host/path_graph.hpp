// Turns the paths returned by the device traversal (pag_travel) into the (graph view, travel sequences)
// pair that the chain selection / seqToString / writers of assembly.cpp work on: a HostGraph that
// contains exactly the vertices that lie on some path.
#pragma once
#include <algorithm>
#include <map>
#include <unordered_map>
#include <vector>

#include "host_graph.hpp"
#include "traversal.hpp"

namespace pagh {

inline void buildPathGraph(const std::vector<std::vector<pag_path_node>> &paths, const std::vector<int> &orient, unsigned k,
                           HostGraph &graph, std::vector<TravelSequence> &results) {
    // distinct vertices, grouped by k-mer code
    std::map<std::uint32_t, std::vector<const pag_path_node *>> byCode;
    std::unordered_map<std::uint32_t, bool> seen;
    for (auto &p : paths)
        for (auto &n : p)
            if (seen.emplace(n.vid, true).second) byCode[n.code].push_back(&n);
    std::size_t nPos = seen.size();
    graph.resize(byCode.size(), nPos, 0);
    graph.k = k;
    std::unordered_map<std::uint32_t, Vertex> where;
    std::size_t ni = 0, pi = 0;
    for (auto &kv : byCode) {
        graph.nodeCode[ni] = kv.first;
        graph.posOff[ni] = pi;
        std::uint32_t j = 0;
        for (auto *n : kv.second) {
            graph.posCtg[pi] = n->ctg;
            graph.posRef[pi] = n->ref;
            graph.posCnt[pi] = n->cnt;
            where[n->vid] = Vertex{static_cast<std::uint32_t>(ni), j++};
            ++pi;
        }
        ++ni;
    }
    graph.posOff[ni] = pi;
    results.assign(paths.size() * 2, {});
    for (std::size_t c = 0; c < paths.size(); ++c) {
        if (c >= orient.size() || orient[c] < 0) continue;
        auto &res = results[2 * c + (orient[c] ? 0 : 1)];
        res.reserve(paths[c].size());
        for (auto &n : paths[c]) res.emplace_back(where[n.vid], n.step);
    }
}

}  // namespace pagh
