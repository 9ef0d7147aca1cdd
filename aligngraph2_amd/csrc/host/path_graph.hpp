// Turns the paths returned by the device traversal (pag_travel) into the (graph view, travel sequences)
// pair that the chain selection / seqToString / writers of assembly.cpp work on: a HostGraph that
// contains exactly the vertices that lie on some path, one single-position node per path element (the
// post-processing only ever asks for the k-mer, position and abundance of a path vertex, so no
// de-duplication or ordering is needed).  Contigs fill disjoint slices, so they are filled by a pool of threads.
#pragma once
#include <algorithm>
#include "host_threads.hpp"
#include <atomic>
#include <cstdint>
#include <thread>
#include <utility>
#include <vector>

#include "host_graph.hpp"
#include "traversal.hpp"

namespace pagh {

// paths[2 * c + (reverse ? 1 : 0)] = (pointer, length) of the path of contig c in that orientation (0 = not traversed)
inline void buildPathGraph(const std::vector<std::pair<const pag_path_node *, std::uint64_t>> &paths, unsigned k, HostGraph &graph,
                           std::vector<TravelSequence> &results, unsigned threads = 0) {
    std::vector<std::size_t> base(paths.size() + 1, 0);
    for (std::size_t c = 0; c < paths.size(); ++c) base[c + 1] = base[c] + paths[c].second;
    const std::size_t total = base.back();
    {   // the six arrays are allocated (and first-touched) side by side: a serial resize of ~600 MB costs ~60 ms
        std::thread t1([&] { graph.nodeCode.assign(total, 0); });
        std::thread t2([&] { graph.posOff.assign(total + 1, 0); });
        std::thread t3([&] { graph.edgeOff.assign(total + 1, 0); });
        std::thread t4([&] { graph.posCtg.assign(total, 0); });
        std::thread t5([&] { graph.posRef.assign(total, 0); });
        graph.posCnt.assign(total, 0);
        graph.edgeTo.clear();
        graph.edgeStep.clear();
        t1.join();
        t2.join();
        t3.join();
        t4.join();
        t5.join();
    }
    graph.k = k;
    results.resize(paths.size());  // (inner vectors keep their storage from the previous block)
    for (auto &r : results) r.clear();
    std::atomic<std::size_t> next{0};
    auto worker = [&]() {
        for (std::size_t c; (c = next.fetch_add(1)) < paths.size();) {
            TravelSequence *res = &results[c];
            const pag_path_node *p = paths[c].first;
            const std::size_t n = paths[c].second;
            res->resize(n);
            std::size_t i = base[c];
            for (std::size_t x = 0; x < n; ++x, ++i) {
                graph.nodeCode[i] = p[x].code;
                graph.posOff[i] = i;
                graph.posCtg[i] = p[x].ctg;
                graph.posRef[i] = p[x].ref;
                graph.posCnt[i] = p[x].cnt;
                (*res)[x] = {Vertex{static_cast<std::uint32_t>(i), 0u}, p[x].step};
            }
        }
    };
    unsigned nThreads = threads ? threads : std::max(1u, usableCpus());
    nThreads = static_cast<unsigned>(std::min<std::size_t>(nThreads, std::max<std::size_t>(1, paths.size())));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    if (total) graph.posOff[total] = total;
}

}  // namespace pagh
