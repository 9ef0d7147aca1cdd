// Per-contig traversal driver, chain selection and output writers.
// Restates PAssembly::testTravel5 + combatSeq + UnionSet (reference PAGraph/src/tools/graph/
// PAssembly.cpp:11-336, PAssembly.tcc:2-31, UnionSet.cpp:7-20).
#pragma once
#include <iosfwd>
#include <functional>
#include <set>
#include <string>
#include <utility>

#include <vector>

#include "host_graph.hpp"
#include "traversal.hpp"
#include "position_mapper.hpp"
#include "seq_db.hpp"

namespace pagh {

struct AssembleStats {
    std::uint64_t nContigs = 0, nPathNodes = 0, nPathBases = 0, nChains = 0, nFastaBases = 0, pathChecksum = 0;
};

// returns the (contig name, forward) pairs consumed by emitted chains.
// travelled: the travel sequences (PAlgorithm::travelSequence, produced by the device traversal), indexed
// 2 * contig + (reverse ? 1 : 0); `graph` only has to contain the vertices on those sequences.  The sequences are
// consumed (their storage is handed back for the next block).
// hostThreads: workers for the per-contig traversal loop (the reference uses max(1, t/8) threads there,
// PAssembly.cpp:30; results are independent of the worker count).
// ONE block built by several ranks (PAGRAPH_SHARD): every rank writes the path dumps (<prefix><contig>_<0|1>.txt) of the
// contigs IT walked, from its own travel sequences — the dumps are most of a block's output bytes; the chain selection and
// everything after it needs all travel sequences and stays with rank 0.  writesDump(contig id): does this call write that
// contig's dump; dumpsOnly: return after the dumps (a rank other than 0: ctgSet = its own contigs, the result is empty).
struct AssembleShare {
    std::function<bool(std::size_t)> writesDump;
    bool dumpsOnly = false;
};
std::set<std::pair<std::string, bool>> assemble(const std::string &outDir, const std::string &prefix, const HostGraph &graph,
                                                const SeqDb &contigs, const SeqDb &refs, const PositionMapper &ctgMapper,
                                                const PositionMapper &refMapper,
                                                const std::set<std::pair<std::string, bool>> &ctgSet, std::size_t deviation,
                                                double errorRate, double startSplit, std::size_t minLen, unsigned threadNum,
                                                unsigned hostThreads, AssembleStats *stats, bool quiet,
                                                std::vector<TravelSequence> &travelled, std::ostream *logTo = nullptr,
                                                const AssembleShare *share = nullptr);

}  // namespace pagh
