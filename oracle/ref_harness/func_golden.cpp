// oracle/ref_harness/func_golden.cpp — TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Prints known-answer tables of the reference's own small functions on the hot path, captured once
// into tests/golden/func_*.txt by tests/golden/make_golden.py:
//   kmer      KmerHelper::kmer2Code / code2Kmer          (kmer/KmerHelper.cpp:7-37)
//   mapper    PositionMapper::dualToSingle / singleToDual (position/PositionMapper.cpp:37-64)
//   predicate PABruijnGraph::checkPosition / isEdgeSimilar / isPosSimilar (graph/PABruijnGraph.cpp:143-165, 379-400)
//   edit      PAlgorithm::editDistance                    (graph/PAlgorithm.cpp:46-69)
#include <cstdint>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "graph/PABruijnGraph.hpp"
#include "graph/PAlgorithm.hpp"
#include "kmer/KmerHelper.hpp"
#include "position/PositionMapper.hpp"
#include "seq/AbstractSeqDatabase.hpp"

namespace {
class MemSeqDb : public AbstractSeqDatabase {
public:
    void add(const std::string &name, const std::string &seq) {
        _nameToId[name] = _seqs.size();
        _seqs.emplace_back(seq, name);
    }
};
std::uint64_t rng_state = 88172645463325252ull;
std::uint64_t rnd() {  // xorshift64
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
}  // namespace

int main(int argc, char **argv) {
    std::string what = argc > 1 ? argv[1] : "";
    if (what == "kmer") {
        const char *seqs[] = {"ACGTACGTTTGACCA", "acgtnNxACGTTGCATGCAAACCCGGGTTT", "AC", "GATTACAGATTACAGATTACA"};
        for (auto s : seqs)
            for (std::size_t k : {3, 5, 8, 14}) {
                std::vector<std::uint64_t> codes;
                KmerHelper::kmer2Code(codes, s, k);
                std::cout << s << " " << k << " " << codes.size();
                for (auto c : codes) std::cout << " " << c << ":" << KmerHelper::code2Kmer(c, k);
                std::cout << "\n";
            }
    } else if (what == "mapper") {
        MemSeqDb db;
        db.add("a", std::string(100, 'A'));
        db.add("b", std::string(37, 'C'));
        db.add("c", std::string(250, 'G'));
        PositionMapper m(db);
        std::cout << "extra " << m.extraStart() << "\n";
        for (std::int64_t idx = -3; idx <= 3; ++idx)
            for (std::int64_t pos : {0, 1, 36, 99, 249}) {
                auto s = m.dualToSingle(idx, pos);
                std::cout << "d2s " << idx << " " << pos << " " << s << "\n";
            }
        for (std::size_t s = 0; s < m.extraStart(); s += 7) {
            auto d = m.singleToDual(s);
            std::cout << "s2d " << s << " " << d.first << " " << d.second << "\n";
        }
    } else if (what == "predicate") {
        // a grid around the interesting boundaries, zero coordinates and u32 wrap-around included
        std::vector<std::uint32_t> base = {0, 5, 1000, 4294967290u};
        std::vector<int> deltas = {-41, -21, -20, -11, -10, -1, 0, 3, 10, 11, 20, 21, 40};
        std::vector<std::uint32_t> dists = {3, 7, 20, 100};
        for (auto dev : {10u, 20u})
            for (auto a1 : base)
                for (auto a2 : base)
                    for (auto dist : dists)
                        for (auto d1 : deltas)
                            for (auto d2 : {-21, -3, 0, 3, 20}) {
                                PABruijnGraph::DualPos p1(a1, a2);
                                for (int z = 0; z < 4; ++z) {
                                    std::uint32_t b1 = (z & 1) ? 0u : a1 + dist + (std::uint32_t)d1;
                                    std::uint32_t b2 = (z & 2) ? 0u : a2 + dist + (std::uint32_t)d2;
                                    PABruijnGraph::DualPos p2(b1, b2);
                                    auto g = PABruijnGraph::checkPosition(p1, p2, dist, dev, 0.15);
                                    auto e = PABruijnGraph::isEdgeSimilar(p1, p2, (int)dist, dev, 0.15);
                                    auto s = PABruijnGraph::isPosSimilar(p1, p2, dev / 2);
                                    std::cout << a1 << " " << a2 << " " << b1 << " " << b2 << " " << dist << " " << dev << " "
                                              << (int)g << " " << e.first << e.second << " " << s.first << s.second << "\n";
                                }
                            }
    } else if (what == "edit") {
        const char *acgt = "ACGT";
        for (int i = 0; i < 400; ++i) {
            std::string a, b;
            std::size_t la = 1 + rnd() % 14, lb = 1 + rnd() % 14;
            for (std::size_t j = 0; j < la; ++j) a.push_back(acgt[rnd() % 4]);
            for (std::size_t j = 0; j < lb; ++j) b.push_back(i % 3 == 0 && j < la ? a[j] : acgt[rnd() % 4]);
            std::cout << a << " " << b << " " << PAlgorithm::editDistance(a, b) << "\n";
        }
    } else {
        std::cerr << "usage: func_golden kmer|mapper|predicate|edit\n";
        return 1;
    }
    return 0;
}
