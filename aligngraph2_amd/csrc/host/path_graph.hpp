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
    // every path element becomes its own single-position node: the post-processing only ever asks for the
    // k-mer, position and abundance of a path vertex, so no de-duplication or ordering is needed
    std::size_t total = 0;
    for (auto &p : paths) total += p.size();
    graph.resize(total, total, 0);
    graph.k = k;
    results.assign(paths.size() * 2, {});
    std::size_t i = 0;
    for (std::size_t c = 0; c < paths.size(); ++c) {
        const bool used = c < orient.size() && orient[c] >= 0;
        TravelSequence *res = used ? &results[2 * c + (orient[c] ? 0 : 1)] : nullptr;
        if (res) res->reserve(paths[c].size());
        for (auto &n : paths[c]) {
            graph.nodeCode[i] = n.code;
            graph.posOff[i] = i;
            graph.posCtg[i] = n.ctg;
            graph.posRef[i] = n.ref;
            graph.posCnt[i] = n.cnt;
            if (res) res->emplace_back(Vertex{static_cast<std::uint32_t>(i), 0u}, n.step);
            ++i;
        }
    }
    if (total) graph.posOff[total] = total;
}

}  // namespace pagh
