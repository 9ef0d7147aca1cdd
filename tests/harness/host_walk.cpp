// tests/harness/host_walk.cpp — TEST INFRASTRUCTURE (never linked into the product).
// Host restatement of the reference's traversal (PAlgorithm, reference PAGraph/src/tools/graph/PAlgorithm.{tcc,cpp}) over
// an exported CSR: the checker the device walkers are compared with (tests/walk_check.py, test_gpu_big.py) and the walk
// of the oracle-backed harness programs.  Pinned to the compiled reference by the golden output directories.
#include "host_walk.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <deque>
#include <atomic>
#include <sstream>
#include <thread>

namespace pagh {

std::size_t HostWalk::editDistance(const std::string &a, const std::string &b) {
    std::vector<std::vector<std::size_t>> dp(2, std::vector<std::size_t>(b.size() + 1, 0));
    std::size_t flag = 0;
    for (std::size_t j = 0; j <= b.size(); ++j) dp[flag][j] = j;
    flag ^= 1;
    for (std::size_t i = 1; i <= a.size(); ++i) {
        for (std::size_t j = 0; j <= b.size(); ++j) {
            if (j == 0) {
                dp[flag][j] = i;
            } else {
                dp[flag][j] = std::min(dp[flag ^ 1][j] + 1, dp[flag][j - 1] + 1);
                dp[flag][j] = std::min(dp[flag][j], dp[flag ^ 1][j - 1] + (a[i - 1] == b[j - 1] ? 0 : 1));
            }
        }
        flag ^= 1;
    }
    return dp[flag ^ 1][b.size()];
}

// PABruijnGraph::searchSuccessors (PABruijnGraph.cpp:167-197): child order x position order
void HostWalk::successors(std::vector<std::pair<Vertex, int>> &out, const Vertex &v, std::uint32_t deviation,
                           double errorRate) const {
    DualPos root = g_.position(v);
    for (std::uint64_t e = g_.edgeOff[v.node]; e < g_.edgeOff[v.node + 1]; ++e) {
        std::int64_t child = g_.findNode(g_.edgeTo[e]);
        if (child < 0) continue;  // cannot happen: every edge target was sampled, hence owns positions
        int step = g_.edgeStep[e];
        std::size_t n = g_.nPositions(static_cast<std::uint32_t>(child));
        for (std::size_t j = 0; j < n; ++j) {
            Vertex c{static_cast<std::uint32_t>(child), static_cast<std::uint32_t>(j)};
            if (checkPosition(root, g_.position(c), static_cast<std::uint32_t>(step), deviation, errorRate) != Oops)
                out.emplace_back(c, step);
        }
    }
}

// PAlgorithm::classifySuccessors (PAlgorithm.tcc:35-90)
template <typename Filter>
void HostWalk::classifySuccessors(std::vector<std::pair<Vertex, int>> &results, const Vertex &v, std::uint32_t deviation,
                                   double errorRate, std::pair<std::int64_t, std::int64_t> ctgRange, bool canLeap,
                                   double leapMin, Filter filter) const {
    DualPos root = g_.position(v);
    std::vector<std::pair<Vertex, int>> origin;
    successors(origin, v, deviation, errorRate);
    {
        std::size_t p = 0;
        for (std::size_t i = 0; i < origin.size(); ++i)
            if (filter(v, origin[i])) origin[p++] = origin[i];
        origin.resize(p);
    }
    std::vector<std::size_t> amazing, excellent, great, skip;
    for (std::size_t i = 0; i < origin.size(); ++i) {
        DualPos pos = g_.position(origin[i].first);
        MatchGrade check = checkPosition(root, pos, static_cast<std::uint32_t>(origin[i].second), deviation, errorRate);
        bool leap = pos.first != 0 && (static_cast<std::int64_t>(pos.first) < ctgRange.first ||
                                       static_cast<std::int64_t>(pos.first) >= ctgRange.second);
        if (leap) {
            auto dual = ctgMapper_.singleToDual(pos.first);
            if (dual.second > ctgMapper_.size(dual.first) * leapMin) continue;
        }
        if (!canLeap && leap) continue;
        if (check == Amazing || leap) amazing.push_back(i);
        else if (check == Excellent) excellent.push_back(i);
        else if (check == Good) great.push_back(i);
        else if (canLeap && check == Skip) skip.push_back(i);
    }
    auto &chosen = !amazing.empty() ? amazing : (!excellent.empty() ? excellent : (!great.empty() ? great : skip));
    for (auto idx : chosen) results.push_back(origin[idx]);
}

// PAlgorithm::walkStraight (PAlgorithm.tcc:93-170)
template <typename ParentFilter>
HostWalk::NodeStatus HostWalk::walkStraight(const std::pair<Vertex, int> &start, std::vector<std::pair<Vertex, int>> &path,
                                              int deviation, double errorRate, std::pair<std::int64_t, std::int64_t> ctgRange,
                                              std::size_t hasSize, std::size_t splitSize, double splitMin,
                                              ParentFilter parentFilter, std::size_t limitation) const {
    UniqueTable uniqueTable;
    PosTable ctgPosTable;
    resetCtgPosTable(ctgPosTable);

    const Vertex &v0 = start.first;
    std::size_t nowSize = static_cast<std::size_t>(start.second);
    path.emplace_back(v0, start.second);
    {
        std::uint32_t c = g_.position(v0).first;
        if (c != 0 && (static_cast<std::int64_t>(c) < ctgRange.first || static_cast<std::int64_t>(c) >= ctgRange.second))
            return Leap;
    }
    insertCtgPos(ctgPosTable, g_.position(path.front().first).first);
    uniqueTable.insert(g_.slot(v0));

    auto filter = [&](const Vertex &parent, const std::pair<Vertex, int> &succ) -> bool {
        DualPos sp = g_.position(succ.first);
        return parentFilter(parent, succ) && uniqueTable.count(g_.slot(succ.first)) == 0 &&
               (sp.first == 0 ||
                isEdgeSimilar(g_.position(parent), sp, succ.second, static_cast<std::size_t>(deviation), errorRate).first ||
                !existCtgPos(ctgPosTable, sp.first));
    };

    std::vector<std::pair<Vertex, int>> succ;
    for (;;) {
        succ.clear();
        classifySuccessors(succ, path.back().first, static_cast<std::uint32_t>(deviation), errorRate, ctgRange,
                           (hasSize + nowSize) >= splitSize, splitMin, filter);
        if (succ.empty()) return End;
        if (succ.size() > 1) return Branch;
        auto front = succ.front();
        uniqueTable.insert(g_.slot(front.first));
        insertCtgPos(ctgPosTable, g_.position(front.first).first);
        path.push_back(front);
        nowSize += static_cast<std::size_t>(front.second);
        std::uint32_t last = g_.position(path.back().first).first;
        if (last != 0 && (static_cast<std::int64_t>(last) < ctgRange.first || static_cast<std::int64_t>(last) >= ctgRange.second))
            return Leap;
        if (limitation > 0 && path.size() >= limitation) return Limit;
    }
}

// PAlgorithm::graphTravel (PAlgorithm.tcc:172-298)
template <typename ParentFilter>
TravelSequence HostWalk::graphTravel(const Vertex &start, int deviation, double errorRate,
                                      std::pair<std::int64_t, std::int64_t> ctgRange, std::size_t hasSize,
                                      std::size_t splitSize, double splitMin, ParentFilter parentFilter) const {
    PosTable ctgTravelPosTable;
    resetCtgPosTable(ctgTravelPosTable);
    UniqueTable travelUniqueTable;
    TravelSequence seq;
    std::size_t nowSize = g_.k;
    std::vector<std::pair<Vertex, int>> path;
    std::vector<std::vector<std::pair<Vertex, int>>> paths;

    std::pair<Vertex, int> chosenOne{start, static_cast<int>(g_.k)};
    insertCtgPos(ctgTravelPosTable, g_.position(start).first);

    auto filter = [&](const Vertex &parent, const std::pair<Vertex, int> &succ) -> bool {
        DualPos sp = g_.position(succ.first);
        return parentFilter(parent, succ) && travelUniqueTable.count(g_.slot(succ.first)) == 0 &&
               (sp.first == 0 ||
                isEdgeSimilar(g_.position(parent), sp, succ.second, static_cast<std::size_t>(deviation), errorRate).first ||
                !existCtgPos(ctgTravelPosTable, sp.first));
    };

    walkStraight(chosenOne, path, deviation, errorRate, ctgRange, hasSize + nowSize, splitSize, splitMin, filter);
    paths.push_back(path);
    std::size_t chosenIdx = 0;
    std::vector<std::pair<Vertex, int>> succ;

    for (;;) {
        auto &chosenPath = paths[chosenIdx];
        for (auto &p : chosenPath) {
            seq.push_back(p);
            travelUniqueTable.insert(g_.slot(p.first));
            nowSize += static_cast<std::size_t>(p.second);
        }
        for (auto &p : chosenPath) insertCtgPos(ctgTravelPosTable, g_.position(p.first).first);

        Vertex lastNode = seq.back().first;
        std::uint32_t lastCtgPos = g_.position(lastNode).first;
        if (lastCtgPos != 0 && (static_cast<std::int64_t>(lastCtgPos) < ctgRange.first ||
                                static_cast<std::int64_t>(lastCtgPos) >= ctgRange.second))
            break;

        succ.clear();
        classifySuccessors(succ, lastNode, static_cast<std::uint32_t>(deviation), errorRate, ctgRange,
                           (hasSize + nowSize) >= splitSize, splitMin, filter);

        std::vector<std::pair<std::size_t, std::size_t>> leap, branch, tips;
        paths.clear();
        for (std::size_t i = 0; i < succ.size(); ++i) {
            path.clear();
            NodeStatus status =
                walkStraight(succ[i], path, deviation, errorRate, ctgRange, hasSize + nowSize, splitSize, splitMin, filter);
            paths.push_back(path);
            if (status == Leap) leap.emplace_back(i, path.size());
            else if (status == End) tips.emplace_back(i, path.size());
            else branch.emplace_back(i, path.size());
        }
        if (leap.empty() && tips.empty() && branch.empty()) break;

        if (!leap.empty()) {
            chosenIdx = leap.front().first;
        } else if (!branch.empty()) {
            std::size_t chosen = 0;
            for (std::size_t i = 1; i < branch.size(); ++i)
                if (g_.abundance(succ[branch[i].first].first) > g_.abundance(succ[branch[chosen].first].first)) chosen = i;
            chosenIdx = branch[chosen].first;
        } else {
            std::size_t chosen = 0;
            for (std::size_t i = 1; i < tips.size(); ++i)
                if (tips[i].second > tips[chosen].second) chosen = i;
            chosenIdx = tips[chosen].first;
        }
    }
    return seq;
}

// PAlgorithm::appendSeq (PAlgorithm.cpp:110-142)
std::int64_t HostWalk::appendSeq(TravelSequence &base, const TravelSequence &tail) const {
    if (tail.empty()) return 0;
    std::int64_t dLen = 0;
    auto &head = tail.front();
    int dist = static_cast<int>(g_.k);
    std::uint32_t headCtg = g_.position(head.first).first;
    while (!base.empty() &&
           (g_.position(base.back().first).first == 0 || headCtg <= g_.position(base.back().first).first)) {
        dLen -= base.back().second;
        base.pop_back();
    }
    if (!base.empty()) dist = static_cast<int>(headCtg - g_.position(base.back().first).first);
    for (auto &n : tail) {
        dLen += n.second;
        base.push_back(n);
    }
    dLen -= base[base.size() - tail.size()].second - dist;
    base[base.size() - tail.size()].second = dist;
    return dLen;
}

// PAlgorithm::filterSequence (PAlgorithm.cpp:27-44)
void HostWalk::filterSequence(TravelSequence &seq) const {
    const std::size_t windowSize = 10;
    if (seq.size() < windowSize) return;
    std::size_t startIdx = seq.size() - seq.size() / 90;
    say(std::to_string(startIdx) + "\n");
    for (std::size_t i = startIdx; i < seq.size() - windowSize + 1; ++i) {
        std::uint32_t firstPos = g_.position(seq[i].first).first;
        std::uint32_t secondPos = g_.position(seq[std::min(seq.size(), i + windowSize) - 1].first).first;
        if (secondPos != 0 && firstPos != 0 && secondPos < firstPos) {
            seq.resize(i + 1);
            return;
        }
    }
}

TravelSequence HostWalk::travelSequence(std::size_t ctgIdx, bool forward, std::size_t deviation, double errorRate,
                                         double startSplit, std::size_t minLen) {
    const std::size_t topK = std::min(threadNum_, 8u);  // seed top-K is coupled to -t

    UniqueTable globalUniqueTable;
    PosTable ctgGlobalPosTable;
    resetCtgPosTable(ctgGlobalPosTable);

    const std::int64_t chosenOne = forward ? static_cast<std::int64_t>(ctgIdx) + 1 : -static_cast<std::int64_t>(ctgIdx) - 1;
    const std::string ctgStr = contigs_.toString(ctgIdx, forward);
    const std::size_t ctgLen = contigs_.length(ctgIdx);

    // PABruijnGraph::findAll (PABruijnGraph.cpp:339-353), restricted to k-mers that own vertices
    // (solid k-mers without positions contribute nothing to the seed searches)
    std::vector<std::pair<std::uint32_t, std::size_t>> aNodes;
    if (ctgLen >= g_.k) {
        std::uint64_t code = 0;
        const std::uint64_t mask = g_.k >= 32 ? ~0ull : ((1ull << (2 * g_.k)) - 1);
        auto acgt = [](char c) -> unsigned { return c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 0u; };
        for (std::size_t i = 0; i < ctgLen; ++i) {
            code = ((code << 2) | acgt(ctgStr[i])) & mask;
            if (i + 1 >= g_.k) {
                std::int64_t n = g_.findNode(static_cast<std::uint32_t>(code));
                if (n >= 0) aNodes.emplace_back(static_cast<std::uint32_t>(n), i + 1 - g_.k);
            }
        }
    }

    const std::size_t splitLen = static_cast<std::size_t>(ctgLen * startSplit);
    const double splitMin = 1 - startSplit;
    const std::uint32_t ctgLeft = static_cast<std::uint32_t>(ctgMapper_.dualToSingle(chosenOne, 0));
    const std::uint32_t ctgRight = static_cast<std::uint32_t>(ctgMapper_.dualToSingle(chosenOne, static_cast<std::int64_t>(ctgLen)));
    const std::uint32_t revCtgLeft = static_cast<std::uint32_t>(ctgMapper_.dualToSingle(-chosenOne, 0));
    const std::uint32_t revCtgRight = static_cast<std::uint32_t>(ctgMapper_.dualToSingle(-chosenOne, static_cast<std::int64_t>(ctgLen)));
    const std::pair<std::int64_t, std::int64_t> ctgRange{ctgLeft, ctgRight};

    auto filter = [&](const Vertex &parent, const std::pair<Vertex, int> &succ) -> bool {
        DualPos sp = g_.position(succ.first);
        return globalUniqueTable.count(g_.slot(succ.first)) == 0 &&
               (sp.first == 0 || isEdgeSimilar(g_.position(parent), sp, succ.second, deviation, errorRate).first ||
                !existCtgPos(ctgGlobalPosTable, sp.first)) &&
               (sp.first == 0 || (sp.first < revCtgLeft || sp.first >= revCtgRight));
    };

    // PAlgorithm::searchPANode / searchPANode2 (PAlgorithm.tcc:300-365)
    auto searchNodes = [&](std::vector<Vertex> &result, bool onlyFirst, bool windowed, std::size_t pos, auto f) {
        std::set<std::uint64_t> unique;
        std::size_t left = 0, right = 0;
        if (windowed) {
            left = pos - std::min(pos, 1000 * deviation);
            right = pos + 1000 * deviation;
        }
        for (auto &node : aNodes) {
            if (windowed) {
                if (node.second < left) continue;
                if (node.second > right) break;
            }
            std::size_t n = g_.nPositions(node.first);
            for (std::size_t i = 0; i < n; ++i) {
                Vertex v{node.first, static_cast<std::uint32_t>(i)};
                DualPos p = g_.position(v);
                auto d1 = ctgMapper_.singleToDual(p.first);
                std::uint64_t slot = g_.slot(v);
                if (unique.count(slot) == 0 && f(node.second, d1.first, static_cast<std::uint64_t>(d1.second))) {
                    result.push_back(v);
                    unique.insert(slot);
                }
            }
            if (!result.empty() && onlyFirst) break;
        }
    };

    std::vector<Vertex> paNodes;
    searchNodes(paNodes, true, false, 0, [&](std::size_t originCtgPos, std::int64_t curCtgIdx, std::uint64_t curCtgPos) {
        return curCtgIdx == chosenOne &&
               std::max<std::uint64_t>(curCtgPos, originCtgPos) - std::min<std::uint64_t>(curCtgPos, originCtgPos) <= deviation;
    });
    paNodes.resize(std::min(paNodes.size(), topK));

    std::int64_t ctgStart = 0;
    for (auto &n : paNodes) ctgStart = std::max(ctgStart, ctgMapper_.singleToDual(g_.position(n).first).second);
    say("Ctgstart=" + std::to_string(ctgStart) + "\n");

    TravelSequence travelSeq, longestSeq;
    std::int64_t varLen = 0;
    std::size_t countLen = 0;
    std::deque<std::uint32_t> ctgPosQue, refPosQue;
    const std::size_t maxQueSize = 4;
    bool finalLeap = false;

    while (!paNodes.empty()) {
        longestSeq.clear();
        std::size_t maxLen = 0, chooseCtgPos = 0, chooseRefPos = 0;
        bool leap = false;

        std::vector<TravelSequence> seqResult(paNodes.size());
        for (std::size_t ii = 0; ii < paNodes.size(); ++ii)
            seqResult[ii] = graphTravel(paNodes[ii], static_cast<int>(deviation), errorRate, ctgRange,
                                        static_cast<std::size_t>(varLen), splitLen, splitMin, filter);

        for (std::size_t i = 0; i < paNodes.size(); ++i) {
            auto &seq = seqResult[i];
            std::size_t len = SeqTools::seqSize(seq);
            std::uint32_t lastCtg = g_.position(seq.back().first).first;
            leap = lastCtg != 0 && ctgMapper_.singleToDual(lastCtg).first != chosenOne;
            if (!leap && i > 0 && minLen > 0 && SeqTools::seqSize(seq) < minLen) continue;
            if (len > maxLen || leap) {
                maxLen = len;
                longestSeq = seq;
                chooseCtgPos = static_cast<std::size_t>(ctgMapper_.singleToDual(g_.position(paNodes[i]).first).second);
                chooseRefPos = static_cast<std::size_t>(refMapper_.singleToDual(g_.position(paNodes[i]).second).second);
                if (leap) break;
            }
        }
        say("choose " + std::to_string(chooseCtgPos) + ", " + std::to_string(chooseRefPos) + "\n");

        varLen += appendSeq(travelSeq, longestSeq);
        countLen += maxLen;
        say("\tle " + std::to_string(varLen) + "\n");

        if (chooseCtgPos != 0) {
            ctgPosQue.push_back(static_cast<std::uint32_t>(chooseCtgPos));
            while (ctgPosQue.size() > maxQueSize) ctgPosQue.pop_front();
        }
        if (chooseRefPos != 0) {
            refPosQue.push_back(static_cast<std::uint32_t>(chooseRefPos));
            while (refPosQue.size() > maxQueSize) refPosQue.pop_front();
        }
        for (auto &n : longestSeq) globalUniqueTable.insert(g_.slot(n.first));
        for (auto &n : longestSeq) insertCtgPos(ctgGlobalPosTable, g_.position(n.first).first);

        bool ctgRepeat = false, refRepeat = false;
        if (ctgPosQue.size() >= maxQueSize) {
            auto mm = std::minmax_element(ctgPosQue.begin(), ctgPosQue.end());
            ctgRepeat = static_cast<std::size_t>(*mm.second - *mm.first) <= 2 * deviation;
        }
        if (refPosQue.size() >= maxQueSize) {
            auto mm = std::minmax_element(refPosQue.begin(), refPosQue.end());
            refRepeat = static_cast<std::size_t>(*mm.second - *mm.first) <= 2 * deviation;
        }
        if (ctgRepeat) say("REPEAT I\n");
        if (refRepeat) say("REPEAT II\n");
        if (ctgRepeat || refRepeat || leap) {
            if (leap) {
                say("leap\n");
                finalLeap = true;
            }
            break;
        }

        std::size_t lastCtgPos = 0;
        std::string lastCtgKmer;
        bool flag1 = false, flag2 = false;
        for (auto it = travelSeq.rbegin(); (!flag1 || !flag2) && it != travelSeq.rend(); ++it) {
            DualPos p = g_.position(it->first);
            if (!flag1 && p.first != 0) {
                auto dual = ctgMapper_.singleToDual(p.first);
                if (dual.first == chosenOne && dual.second >= 0) {
                    lastCtgPos = static_cast<std::size_t>(dual.second);
                    lastCtgKmer = g_.kmerString(it->first.node);
                    flag1 = true;
                }
            }
            if (!flag2 && p.second != 0) flag2 = true;
        }

        paNodes.clear();
        searchNodes(paNodes, false, true, lastCtgPos, [&](std::size_t, std::int64_t curCtgIdx, std::uint64_t curCtgPos) {
            return curCtgIdx == chosenOne &&
                   std::max<std::uint64_t>(curCtgPos, lastCtgPos) - std::min<std::uint64_t>(curCtgPos, lastCtgPos) <= deviation;
        });
        {
            std::size_t p = 0;
            for (std::size_t i = 0; i < paNodes.size(); ++i)
                if (globalUniqueTable.count(g_.slot(paNodes[i])) == 0) paNodes[p++] = paNodes[i];
            paNodes.resize(p);
        }
        const std::string &parentKmer = lastCtgKmer;
        // unstable std::sort on a key with ties: same comparator, same initial order, same libstdc++
        std::sort(paNodes.begin(), paNodes.end(), [&](const Vertex &l, const Vertex &r) {
            return editDistance(parentKmer, g_.kmerString(l.node)) < editDistance(parentKmer, g_.kmerString(r.node));
        });
        paNodes.resize(std::min(paNodes.size(), topK));
    }

    if (!finalLeap) filterSequence(travelSeq);
    if (finalLeap) {
        auto dual = ctgMapper_.singleToDual(g_.position(travelSeq.back().first).first);
        if (static_cast<std::size_t>(std::llabs(dual.first)) == ctgIdx + 1 ||
            dual.second >= contigs_.length(static_cast<std::size_t>(std::llabs(dual.first)) - 1) * (1 - startSplit)) {
            travelSeq.pop_back();
            say("Pump it\n");
        }
    }
    (void)countLen;
    return travelSeq;
}

// every (contig, orientation) of ctgSet, on a pool of host threads (contigs are independent; the reference uses
// max(1, t / 8) workers, PAssembly.cpp:30)
std::vector<TravelSequence> hostWalkAll(const HostGraph &graph, const SeqDb &contigs, const SeqDb &refs,
                                        const PositionMapper &ctgMapper, const PositionMapper &refMapper,
                                        const std::set<std::pair<std::string, bool>> &ctgSet, std::size_t deviation,
                                        double errorRate, double startSplit, std::size_t minLen, unsigned threadNum,
                                        unsigned hostThreads) {
    std::vector<TravelSequence> results(contigs.size() * 2);
    std::vector<std::pair<std::string, bool>> list(ctgSet.begin(), ctgSet.end());
    std::atomic<std::size_t> next{0};
    auto worker = [&]() {
        for (std::size_t li; (li = next.fetch_add(1)) < list.size();) {
            if (!contigs.contains(list[li].first)) continue;
            const std::size_t ctgIdx = contigs.id(list[li].first);
            HostWalk algo(graph, contigs, refs, ctgMapper, refMapper, threadNum, nullptr);
            results[2 * ctgIdx + (list[li].second ? 0 : 1)] =
                algo.travelSequence(ctgIdx, list[li].second, deviation, errorRate, startSplit, minLen);
        }
    };
    unsigned n = hostThreads ? hostThreads : std::max(1u, std::thread::hardware_concurrency());
    n = std::max(1u, std::min<unsigned>(n, (unsigned)list.size()));
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < n; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return results;
}

}  // namespace pagh
