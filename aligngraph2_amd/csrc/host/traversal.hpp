// Host side of the traversal epilogue: what is done with a finished travel sequence.
// Restates PAlgorithm::seqToString / seqSize (reference PAGraph/src/tools/graph/PAlgorithm.cpp:428-497) and
// PABruijnGraph::toString(PANode) (PABruijnGraph.cpp:358-364).  The walk that produces the sequences runs on the
// device (csrc/hip/k5_travel*.hip); no host walk is part of the product.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "host_graph.hpp"
#include "position_mapper.hpp"
#include "seq_db.hpp"

namespace pagh {

using TravelSequence = std::vector<std::pair<Vertex, int>>;

class SeqTools {
public:
    SeqTools(const HostGraph &g, const SeqDb &contigs, const SeqDb &refs, const PositionMapper &ctgMapper,
             const PositionMapper &refMapper)
        : g_(g), contigs_(contigs), refs_(refs), ctgMapper_(ctgMapper), refMapper_(refMapper) {}

    // PAlgorithm::seqToString (PAlgorithm.cpp:428-489)
    std::string seqToString(const TravelSequence &seq, std::size_t deviation, double errorRate) const;
    // PAlgorithm::seqSize (PAlgorithm.cpp:491-497)
    static std::size_t seqSize(const TravelSequence &seq);
    // PABruijnGraph::toString(PANode) (PABruijnGraph.cpp:358-364)
    std::string vertexString(const Vertex &v) const;

private:
    const HostGraph &g_;
    const SeqDb &contigs_;
    const SeqDb &refs_;
    const PositionMapper &ctgMapper_;
    const PositionMapper &refMapper_;
};

}  // namespace pagh
