// Sequence utilities of the traversal epilogue (host side of a17/a18): PAlgorithm::seqToString, seqSize and
// PABruijnGraph::toString(PANode).  The walk itself (classifySuccessors / walkStraight / graphTravel /
// travelSequence) runs on the device (csrc/hip/k5_travel*.hip); a host restatement of it exists only as test
// infrastructure under tests/harness/host_walk.cpp.
#include "traversal.hpp"

#include <charconv>
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <sstream>

namespace pagh {

std::string SeqTools::vertexString(const Vertex &v) const {
    DualPos p = g_.position(v);
    std::string out = g_.kmerString(v.node);
    char num[24];
    for (unsigned long long x : {static_cast<unsigned long long>(p.first), static_cast<unsigned long long>(p.second),
                                 static_cast<unsigned long long>(g_.abundance(v))}) {
        out.push_back(',');
        auto r = std::to_chars(num, num + sizeof num, x);
        out.append(num, static_cast<std::size_t>(r.ptr - num));
    }
    return out;
}

std::size_t SeqTools::seqSize(const TravelSequence &seq) {
    std::size_t n = 0;
    for (auto &s : seq) n += s.second;
    return n;
}

std::string SeqTools::seqToString(const TravelSequence &seq, std::size_t deviation, double errorRate) const {
    if (seq.empty()) return "";
    std::string str;
    str.append(g_.kmerString(seq[0].first.node));
    const int kmerSize = static_cast<int>(g_.k);

    {   // (the usual size: one base per unit of step, plus the first k-mer)
        std::size_t total = static_cast<std::size_t>(kmerSize);
        for (std::size_t i = 1; i < seq.size(); ++i) total += static_cast<std::size_t>(std::max(seq[i].second, 0));
        str.reserve(total + 16);
    }
    for (std::size_t i = 1; i < seq.size(); ++i) {
        const int kmerDist = seq[i].second;
        const std::uint32_t code = g_.nodeCode[seq[i].first.node];
        if (kmerDist <= kmerSize) {
            // the step is covered by the k-mer itself: its last kmerDist bases (nothing of what follows is needed — the
            // similarity tests and coordinate mappings only choose where the bases of a LONGER step are read from)
            for (int j = 0; j < kmerDist; ++j) {
                const int at = kmerSize - kmerDist + j;  // base index inside the k-mer, first base most significant
                str.push_back("ACGT"[(code >> (2 * (kmerSize - 1 - at))) & 3u]);
            }
            continue;
        }
        DualPos prev = g_.position(seq[i - 1].first), now = g_.position(seq[i].first);
        auto similar = isEdgeSimilar(prev, now, seq[i].second, deviation, errorRate);
        bool useCtg = similar.first;
        if (!similar.first && !similar.second) useCtg = isPosSimilar(prev, now, deviation).first;

        const SeqDb &db = useCtg ? contigs_ : refs_;
        const PositionMapper &mapper = useCtg ? ctgMapper_ : refMapper_;
        auto startPos = mapper.singleToDual(useCtg ? prev.first : prev.second);
        auto endPos = mapper.singleToDual(useCtg ? now.first : now.second);

        std::int64_t posDist = endPos.second - startPos.second;
        std::int64_t selIdx = std::llabs(endPos.first) - 1;
        bool selForward = endPos.first > 0;
        double moveLen = posDist * 1.0 / kmerDist;
        double refNow = static_cast<double>(startPos.second + kmerSize);
        std::string kmer = g_.kmerString(seq[i].first.node);

        for (int j = 0; j < kmerDist; ++j) {
            if (kmerSize - kmerDist + j >= 0) {
                str.push_back(kmer[static_cast<std::size_t>(kmerSize - kmerDist + j)]);
            } else {
                std::size_t refPos = static_cast<std::size_t>(std::round(refNow));
                char b = (selIdx >= 0 && static_cast<std::size_t>(selIdx) < db.size())
                             ? db.baseAt(static_cast<std::size_t>(selIdx), refPos, selForward)
                             : 'N';  // the reference indexes out of range here (undefined behaviour)
                str.push_back(static_cast<char>(std::tolower(b)));
            }
            refNow += moveLen;
        }
    }
    return str;
}

}  // namespace pagh
