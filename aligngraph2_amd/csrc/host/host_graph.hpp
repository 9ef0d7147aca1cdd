// Host view of the finished positional A-Bruijn graph (compact CSR, ascending k-mer code) and the
// match predicates of the epsilon-join.  The CSR is what include/pagraph_hip.h: pag_export_csr returns.
//
// Predicates restate PABruijnGraph::isPosSimilar / isEdgeSimilar / checkPosition
// (reference PAGraph/src/tools/graph/PABruijnGraph.cpp:379-400, 143-165) with the reference's exact
// arithmetic: u32 wrap-around subtraction, double division, and the second, un-guarded ratio test of
// checkPosition.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "pagraph_hip.h"

namespace pagh {

struct DualPos {
    std::uint32_t first = 0;   // contig single coordinate
    std::uint32_t second = 0;  // reference single coordinate
};

enum MatchGrade { Oops, Skip, Good, Excellent, Amazing };

inline std::pair<bool, bool> isPosSimilar(const DualPos &l, const DualPos &r, std::size_t deviation) {
    bool s1 = l.first != 0 && r.first != 0 &&
              static_cast<std::size_t>(std::max(l.first, r.first) - std::min(l.first, r.first)) <= deviation;
    bool s2 = l.second != 0 && r.second != 0 &&
              static_cast<std::size_t>(std::max(l.second, r.second) - std::min(l.second, r.second)) <= deviation;
    return {s1, s2};
}

inline std::pair<bool, bool> isEdgeSimilar(const DualPos &l, const DualPos &r, int dist, std::size_t deviation,
                                           double errorRate) {
    DualPos tmp;
    tmp.first = l.first != 0 ? l.first + static_cast<std::uint32_t>(dist) : 0;
    tmp.second = l.second != 0 ? l.second + static_cast<std::uint32_t>(dist) : 0;
    auto st = isPosSimilar(tmp, r, deviation);
    bool s1 = st.first, s2 = st.second;
    s1 = s1 || (l.first != 0 && r.first != 0 &&
                std::abs(1.0 - (static_cast<std::uint32_t>(r.first - l.first) * 1.0 / dist)) <= errorRate);
    s2 = s2 || (l.second != 0 && r.second != 0 &&
                std::abs(1.0 - (static_cast<std::uint32_t>(r.second - l.second) * 1.0 / dist)) <= errorRate);
    return {s1, s2};
}

inline MatchGrade checkPosition(const DualPos &p1, const DualPos &p2, std::uint32_t dist, std::uint32_t deviation,
                                double errorRate) {
    auto st = isEdgeSimilar(p1, p2, static_cast<int>(dist), deviation, errorRate);
    bool s1 = st.first, s2 = st.second;
    s1 = s1 || std::abs(1.0 - (static_cast<std::uint32_t>(p2.first - p1.first) * 1.0 / dist)) <= errorRate;
    s2 = s2 || std::abs(1.0 - (static_cast<std::uint32_t>(p2.second - p1.second) * 1.0 / dist)) <= errorRate;
    if (p1.first == 0 || p2.first == 0) return s2 ? (p2.first != 0 ? Excellent : (p1.first != 0 ? Skip : Good)) : Oops;
    if (p1.second == 0 || p2.second == 0) return s1 ? (p2.second != 0 ? Excellent : Good) : Oops;
    return (s1 && s2) ? Amazing : (s1 ? Excellent : (s2 ? Skip : Oops));
}

// vertex handle = (k-mer node, position index); PABruijnNode::getUniqueHelper (PABruijnNode.hpp:69-71)
struct Vertex {
    std::uint32_t node = 0xFFFFFFFFu;
    std::uint32_t pi = 0;
};

class HostGraph {
public:
    std::uint32_t k = 0;
    std::vector<std::uint32_t> nodeCode;
    std::vector<std::uint64_t> posOff, edgeOff;
    std::vector<std::uint32_t> posCtg, posRef;
    std::vector<std::uint16_t> posCnt;
    std::vector<std::uint32_t> edgeTo;
    std::vector<std::int32_t> edgeStep;

    void resize(std::uint64_t nNodes, std::uint64_t nPos, std::uint64_t nEdges) {
        nodeCode.assign(nNodes, 0);
        posOff.assign(nNodes + 1, 0);
        edgeOff.assign(nNodes + 1, 0);
        posCtg.assign(nPos, 0);
        posRef.assign(nPos, 0);
        posCnt.assign(nPos, 0);
        edgeTo.assign(nEdges, 0);
        edgeStep.assign(nEdges, 0);
    }
    pag_csr view() {
        return pag_csr{nodeCode.size(), posCtg.size(), edgeTo.size(), nodeCode.data(), posOff.data(), posCtg.data(),
                       posRef.data(), posCnt.data(), edgeOff.data(), edgeTo.data(), edgeStep.data()};
    }
    // node index of a k-mer code, -1 if the k-mer has no vertex
    std::int64_t findNode(std::uint32_t code) const {
        auto it = std::lower_bound(nodeCode.begin(), nodeCode.end(), code);
        return (it != nodeCode.end() && *it == code) ? it - nodeCode.begin() : -1;
    }
    std::size_t nPositions(std::uint32_t node) const { return static_cast<std::size_t>(posOff[node + 1] - posOff[node]); }
    DualPos position(const Vertex &v) const {
        std::uint64_t s = posOff[v.node] + v.pi;
        return DualPos{posCtg[s], posRef[s]};
    }
    std::uint16_t abundance(const Vertex &v) const { return posCnt[posOff[v.node] + v.pi]; }
    std::uint64_t slot(const Vertex &v) const { return posOff[v.node] + v.pi; }
    // KmerHelper::code2Kmer (KmerHelper.cpp:27-37)
    std::string kmerString(std::uint32_t node) const {
        std::string s(k, 'A');
        std::uint32_t c = nodeCode[node];
        for (std::uint32_t i = 0; i < k; ++i) {
            s[k - 1 - i] = "ACGT"[c & 3u];
            c >>= 2;
        }
        return s;
    }
};

}  // namespace pagh
