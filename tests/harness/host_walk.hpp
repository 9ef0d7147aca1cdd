// tests/harness/host_walk.hpp — TEST INFRASTRUCTURE (never linked into the product).
// Host restatement of PAlgorithm (reference PAGraph/src/tools/graph/PAlgorithm.{hpp,tcc,cpp}): successor
// classification, walkStraight, graphTravel (branch probing), travelSequence (outer loop with seeds, repeat
// detection, leap to the next contig), appendSeq, filterSequence, editDistance — over an exported CSR.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "host_graph.hpp"
#include "position_mapper.hpp"
#include "seq_db.hpp"
#include "traversal.hpp"

namespace pagh {

class HostWalk {
public:
    HostWalk(const HostGraph &g, const SeqDb &contigs, const SeqDb &refs, const PositionMapper &ctgMapper,
             const PositionMapper &refMapper, unsigned threadNum, std::string *log = nullptr)
        : g_(g), contigs_(contigs), refs_(refs), ctgMapper_(ctgMapper), refMapper_(refMapper), threadNum_(threadNum), log_(log) {}

    // PAlgorithm::travelSequence (PAlgorithm.cpp:144-426)
    TravelSequence travelSequence(std::size_t ctgIdx, bool forward, std::size_t deviation, double errorRate,
                                  double startSplit, std::size_t minLen);
    // PAlgorithm::editDistance (PAlgorithm.cpp:46-69)
    static std::size_t editDistance(const std::string &a, const std::string &b);

private:
    enum NodeStatus { End, Branch, Limit, Leap };
    using PosTable = std::pair<std::uint32_t, std::uint32_t>;
    using UniqueTable = std::set<std::uint64_t>;  // vertex slots

    template <typename Filter>
    void classifySuccessors(std::vector<std::pair<Vertex, int>> &results, const Vertex &v, std::uint32_t deviation,
                            double errorRate, std::pair<std::int64_t, std::int64_t> ctgRange, bool canLeap, double leapMin,
                            Filter filter) const;
    template <typename ParentFilter>
    NodeStatus walkStraight(const std::pair<Vertex, int> &start, std::vector<std::pair<Vertex, int>> &path, int deviation,
                            double errorRate, std::pair<std::int64_t, std::int64_t> ctgRange, std::size_t hasSize,
                            std::size_t splitSize, double splitMin, ParentFilter parentFilter, std::size_t limitation = 0) const;
    template <typename ParentFilter>
    TravelSequence graphTravel(const Vertex &start, int deviation, double errorRate,
                               std::pair<std::int64_t, std::int64_t> ctgRange, std::size_t hasSize, std::size_t splitSize,
                               double splitMin, ParentFilter parentFilter) const;
    void successors(std::vector<std::pair<Vertex, int>> &out, const Vertex &v, std::uint32_t deviation, double errorRate) const;
    std::int64_t appendSeq(TravelSequence &base, const TravelSequence &tail) const;
    void filterSequence(TravelSequence &seq) const;

    static bool existCtgPos(const PosTable &t, std::uint32_t pos) { return pos >= t.first && pos <= t.second; }
    static void resetCtgPosTable(PosTable &t) {
        t.first = 0xFFFFFFFFu;
        t.second = 0;
    }
    static void insertCtgPos(PosTable &t, std::uint32_t pos) {
        if (pos == 0) return;
        t.first = std::min(t.first, pos);
        t.second = std::max(t.second, pos);
    }
    void say(const std::string &s) const {
        if (log_) *log_ += s;
    }

    const HostGraph &g_;
    const SeqDb &contigs_;
    const SeqDb &refs_;
    const PositionMapper &ctgMapper_;
    const PositionMapper &refMapper_;
    unsigned threadNum_;
    std::string *log_;
};

// travel sequences of every (contig, orientation) of ctgSet by the host walk, indexed 2 * contig + (reverse ? 1 : 0)
std::vector<TravelSequence> hostWalkAll(const HostGraph &graph, const SeqDb &contigs, const SeqDb &refs,
                                        const PositionMapper &ctgMapper, const PositionMapper &refMapper,
                                        const std::set<std::pair<std::string, bool>> &ctgSet, std::size_t deviation,
                                        double errorRate, double startSplit, std::size_t minLen, unsigned threadNum,
                                        unsigned hostThreads);

}  // namespace pagh
