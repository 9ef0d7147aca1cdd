// pa_cns — drop-in for the consensus step that follows pagraph in the pipeline (reference
// PAGraph/src/main/pa_cns.cpp:12-168): the backbone (first sequence of -i) is cut into parts of -l bases, the alignments of
// -a (3-line ALN) are sliced per part (tools/cns/AlignData.cpp:36-75), gap-normalised (tools/cns/Alignment.cpp:134-215),
// ordered by score, weighted (AlignData.cpp:77-104), threaded through a partial-order alignment graph per part
// (tools/cns/AlnGraphBoost.cpp: addAln / mergeNodes / bestPath / consensus) and the parts' consensus strings are written as
// one FASTA record, 70 columns.
//
// Where the graphs are built (PA_CNS_BACKEND): `hip` (default) — on the device, one thread per part (pag_cns_consensus,
// csrc/hip/k_cns.hip; the program fails without a gfx950 device, there is no silent fallback); `flat` — the device's code
// (csrc/hip/cns_graph.hpp: flat arrays, linked edge lists) compiled for the host, a verification aid; `host` — the restatement
// on std::vector / std::map below (AlnGraph), host code like the reference's, what the CPU tests pin on the goldens.  Reading,
// slicing, gap normalisation, the per-part sort and the weights are host code in every case.
//
// What has to be reproduced beyond the arithmetic, because it decides ties:
//   * the graph is boost::adjacency_list<vecS, vecS, bidirectionalS>: out- and in-edge lists are vectors in insertion order,
//     clear_vertex() erases entries in place (order of the others kept), edge(u, v) finds the first match in u's out list,
//     add_edge() appends.  Graph below keeps exactly that: per vertex two vectors of edge ids.
//   * `_bbMap` is a std::map read with operator[]: a vertex that was never entered (enter / exit vertex) maps to vertex 0.
//   * bestPath(): float scores, `>` keeps the FIRST best out-edge; nodeScore read with operator[] (0.0f when absent).
//   * the per-part std::sort by score is unstable: the same libstdc++ std::sort with the same comparator on a proxy array
//     in the same initial order gives the same permutation.
//   * header fields are read with the same stream extractions; a malformed header yields an all-zero record that is still
//     processed (AlignmentHelper.cpp:24-40) with size_t / uint32_t wrap-around as in the reference.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <iostream>
#include <map>
#include <queue>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../hip/cns_graph.hpp"
#include "host_threads.hpp"
#include "pagraph_hip.h"
#include "seq_db.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------------------
// dagcon::Alignment + normalizeGaps (tools/cns/Alignment.hpp, Alignment.cpp:134-215)
// ---------------------------------------------------------------------------------------------------------------------
struct Aln {
    std::uint32_t tlen = 0, start = 0, end = 0;
    std::string qstr, tstr;
};

Aln normalizeGaps(const Aln &aln) {
    std::size_t len = aln.qstr.length();
    std::string qNorm, tNorm;
    qNorm.reserve(len + 100);
    tNorm.reserve(len + 100);
    std::string qstr = aln.qstr, tstr = aln.tstr;
    for (std::size_t i = 0; i < len; i++) {  // dots to dashes
        if ('.' == qstr[i]) qstr[i] = '-';
        if ('.' == tstr[i]) tstr[i] = '-';
    }
    for (std::size_t i = 0; i < len; i++) {  // mismatches to indels
        char qb = qstr[i], tb = tstr[i];
        if (qb != tb && qb != '-' && tb != '-') {
            qNorm += '-';
            qNorm += qb;
            tNorm += tb;
            tNorm += '-';
        } else {
            qNorm += qb;
            tNorm += tb;
        }
    }
    len = qNorm.length();
    // push gaps to the right, but not past the end (len - 1 wraps for an empty string exactly as in the reference: size_t)
    for (std::size_t i = 0; i < len - 1 && len != 0; i++) {
        if (tNorm[i] == '-') {
            std::size_t j = i;
            while (++j < len) {
                char c = tNorm[j];
                if (c != '-') {
                    if (c == qNorm[i]) {
                        tNorm[i] = c;
                        tNorm[j] = '-';
                    }
                    break;
                }
            }
        }
        if (qNorm[i] == '-') {
            std::size_t j = i;
            while (++j < len) {
                char c = qNorm[j];
                if (c != '-') {
                    if (c == tNorm[i]) {
                        qNorm[i] = c;
                        qNorm[j] = '-';
                    }
                    break;
                }
            }
        }
    }
    Aln out;
    out.start = aln.start;
    out.tlen = aln.tlen;
    for (std::size_t i = 0; i < len; i++)
        if (qNorm[i] != '-' || tNorm[i] != '-') {
            out.qstr += qNorm[i];
            out.tstr += tNorm[i];
        }
    return out;
}

// AlignData::sliceHelper (AlignData.cpp:11-34)
std::pair<std::size_t, std::size_t> sliceHelper(const std::string &tstr, std::size_t originStart, std::size_t sliceStart, std::size_t sliceEnd) {
    std::size_t left, right, cnt = 0;
    for (left = 0; left < tstr.size(); ++left) {
        if (tstr[left] == '-') continue;
        if (originStart + cnt >= sliceStart) break;
        ++cnt;
    }
    for (right = left; right < tstr.size(); ++right) {
        if (tstr[right] == '-') continue;
        if (originStart + cnt >= sliceEnd) break;
        ++cnt;
    }
    return {left, right};
}

struct ScoredAln {
    Aln aln;
    unsigned score;
};

// AlignData::readFromRefFile (AlignData.cpp:36-75) over AlignmentHelper::loadFromRefFile (align/AlignmentHelper.cpp:10-48)
std::vector<std::vector<ScoredAln>> readFromRefFile(const std::string &path, const std::string &skeleton, std::size_t partLen) {
    if (skeleton.empty()) return {};
    const std::size_t partNum = (skeleton.size() + partLen - 1) / partLen;
    std::vector<std::vector<ScoredAln>> part(partNum);
    std::ifstream in(path);
    if (!in.is_open()) return part;
    std::stringstream ss;
    std::string queryName, refName, forward, score, line1, line2, line3;
    std::size_t queryBegin = 0, queryEnd = 0, querySize = 0, refBegin = 0, refEnd = 0, refSize = 0;
    std::size_t curBegin = 0, curEnd = 0;
    std::string curScore;
    for (std::size_t lineCount = 0;; ++lineCount) {
        if (lineCount % 3 == 0) {
            if (!std::getline(in, line1)) break;
            ss.clear();
            ss.str(line1);
            ss >> queryName >> refName >> forward >> score >> queryBegin >> queryEnd >> querySize >> refBegin >> refEnd >> refSize;
            if (!ss.fail()) {
                curBegin = refBegin;
                curEnd = refEnd;
                curScore = score;
            } else {
                curBegin = curEnd = 0;
                curScore.clear();
            }
        } else if (lineCount % 3 == 1) {
            if (!std::getline(in, line2)) break;
        } else {
            if (!std::getline(in, line3)) break;
            const std::string &qstr = line2, &tstr = line3;
            const std::size_t toStart = curBegin, toEnd = curEnd;
            const int sc = std::atoi(curScore.c_str());
            const std::size_t leftPart = toStart / partLen;
            const std::size_t rightPart = std::min((toEnd - 1) / partLen, partNum - 1);
            for (std::size_t i = leftPart; i <= rightPart; ++i) {
                Aln a;
                a.start = static_cast<std::uint32_t>(i == leftPart ? toStart - leftPart * partLen + 1 : 1);
                a.end = static_cast<std::uint32_t>(i == rightPart ? toEnd - rightPart * partLen + 1 : partLen);
                a.tlen = static_cast<std::uint32_t>(partLen);
                auto slice = sliceHelper(tstr, toStart, i * partLen, std::min((i + 1) * partLen, skeleton.size()));
                // (std::string::substr throws when the start lies beyond the string, as the reference's does)
                a.qstr = qstr.substr(slice.first, slice.second - slice.first);
                a.tstr = tstr.substr(slice.first, slice.second - slice.first);
                Aln n = normalizeGaps(a);
                n.end = a.end;
                part[i].push_back(ScoredAln{std::move(n), static_cast<unsigned>(sc)});
            }
        }
    }
    return part;
}

// AlignData::weightAln (AlignData.cpp:77-104)
std::vector<std::size_t> weightAln(const std::vector<ScoredAln> &alignment, std::size_t alpha) {
    if (alignment.empty()) return {};
    unsigned maxScore = alignment[0].score, minScore = alignment[0].score;
    for (auto &a : alignment) {
        maxScore = std::max(maxScore, a.score);
        minScore = std::min(minScore, a.score);
    }
    const unsigned scoreRange = maxScore - minScore;
    std::vector<std::size_t> weights;
    weights.reserve(alignment.size());
    for (auto &a : alignment)
        weights.push_back(std::max(static_cast<std::size_t>((a.score - minScore) * 1.0 / std::max(scoreRange, 1U) * alpha), std::size_t(1)));
    return weights;
}

// ---------------------------------------------------------------------------------------------------------------------
// AlnGraphBoost (tools/cns/AlnGraphBoost.{hpp,cpp}) on plain vectors with the adjacency_list<vecS, vecS, bidirectionalS>
// edge-list semantics
// ---------------------------------------------------------------------------------------------------------------------
struct Node {
    char base = 'N';
    int coverage = 0, weight = 0;
    bool backbone = false, deleted = false;
};
struct Edge {
    std::uint32_t src, dst;
    int count = 0;
    bool visited = false;
};

class AlnGraph {
public:
    explicit AlnGraph(const std::string &backbone) {
        const std::size_t blen = backbone.length();
        nodes_.resize(blen + 2);
        out_.resize(blen + 2);
        in_.resize(blen + 2);
        for (std::size_t i = 0; i < blen + 1; i++) addEdgeRaw(static_cast<std::uint32_t>(i), static_cast<std::uint32_t>(i + 1));
        enter_ = 0;
        nodes_[enter_].base = '^';
        nodes_[enter_].backbone = true;
        for (std::size_t i = 0; i < blen; i++) {
            Node &v = nodes_[i + 1];
            v.backbone = true;
            v.weight = 1;
            v.base = backbone[i];
            bbMap_[static_cast<std::uint32_t>(i + 1)] = static_cast<std::uint32_t>(i + 1);
        }
        exit_ = static_cast<std::uint32_t>(blen + 1);
        nodes_[exit_].base = '$';
        nodes_[exit_].backbone = true;
    }

    void addAln(const Aln &aln, int weight) {
        if (weight <= 0) return;
        std::uint32_t bbPos = aln.start;
        std::uint32_t prevVtx = enter_;
        for (std::size_t i = 0; i < aln.qstr.length(); i++) {
            const char queryBase = aln.qstr[i], targetBase = aln.tstr[i];
            const std::uint32_t currVtx = bbPos;
            if (queryBase == targetBase) {  // match
                checkVertex(currVtx);
                Node &bb = nodes_[bbMap_[currVtx]];
                bb.coverage += weight;
                bb.base = targetBase;
                nodes_[currVtx].weight += weight;
                addEdge(prevVtx, currVtx, weight);
                bbPos++;
                prevVtx = currVtx;
            } else if (queryBase == '-' && targetBase != '-') {  // query deletion
                checkVertex(currVtx);
                Node &bb = nodes_[bbMap_[currVtx]];
                bb.coverage += weight;
                bb.base = targetBase;
                bbPos++;
            } else if (queryBase != '-' && targetBase == '-') {  // query insertion
                const std::uint32_t newVtx = static_cast<std::uint32_t>(nodes_.size());
                nodes_.emplace_back();
                out_.emplace_back();
                in_.emplace_back();
                nodes_[newVtx].base = queryBase;
                nodes_[newVtx].weight += weight;
                bbMap_[newVtx] = bbPos;
                addEdge(prevVtx, newVtx, weight);
                prevVtx = newVtx;
            }
        }
        addEdge(prevVtx, exit_, weight);
    }

    void mergeNodes() {
        std::queue<std::uint32_t> seedNodes;
        seedNodes.push(enter_);
        while (!seedNodes.empty()) {
            const std::uint32_t u = seedNodes.front();
            seedNodes.pop();
            mergeInNodes(u);
            mergeOutNodes(u);
            for (std::size_t x = 0; x < out_[u].size(); ++x) {
                Edge &e = edges_[out_[u][x]];
                e.visited = true;
                const std::uint32_t v = e.dst;
                int notVisited = 0;
                for (std::uint32_t ie : in_[v])
                    if (!edges_[ie].visited) notVisited++;
                if (notVisited == 0) seedNodes.push(v);
            }
        }
    }

    std::string consensus(int minWeight = 0) {
        const std::vector<Node> path = bestPath();
        std::string cns;
        int offs = 0, bestOffs = 0, length = 0, idx = 0;
        bool metWeight = false;
        for (const Node &n : path) {
            if (n.base == nodes_[enter_].base || n.base == nodes_[exit_].base) continue;
            cns += n.base;
            if (!metWeight && n.weight >= minWeight) {
                offs = idx;
                metWeight = true;
            } else if (metWeight && n.weight < minWeight) {
                if ((idx - offs) > length) {
                    bestOffs = offs;
                    length = idx - offs;
                }
                metWeight = false;
            }
            idx++;
        }
        if (metWeight && (idx - offs) > length) {
            bestOffs = offs;
            length = idx - offs;
        }
        return cns.substr(bestOffs, length);
    }

private:
    void checkVertex(std::uint32_t v) const {
        // (the reference indexes its vertex vector without a check; an alignment that runs past its part is undefined
        // behaviour there — here it is an error)
        if (v >= nodes_.size()) throw std::runtime_error("pa_cns: an alignment runs past the end of its part");
    }
    std::uint32_t addEdgeRaw(std::uint32_t u, std::uint32_t v) {  // boost::add_edge
        const std::uint32_t id = static_cast<std::uint32_t>(edges_.size());
        edges_.push_back(Edge{u, v, 0, false});
        out_[u].push_back(id);
        in_[v].push_back(id);
        return id;
    }
    void addEdge(std::uint32_t u, std::uint32_t v, int weight) {  // AlnGraphBoost::addEdge (:114-133)
        checkVertex(v);
        bool edgeExists = false;
        for (std::uint32_t ie : in_[v])
            if (edges_[ie].src == u) {
                edges_[ie].count += weight;
                edgeExists = true;
            }
        if (!edgeExists) edges_[addEdgeRaw(u, v)].count += weight;
    }
    int findEdge(std::uint32_t u, std::uint32_t v) const {  // boost::edge(u, v, g): first match in u's out-edge list
        for (std::uint32_t oe : out_[u])
            if (edges_[oe].dst == v) return static_cast<int>(oe);
        return -1;
    }
    void clearVertex(std::uint32_t n) {  // boost::clear_vertex for a bidirectional vecS graph
        for (std::uint32_t oe : out_[n]) {
            auto &lst = in_[edges_[oe].dst];
            lst.erase(std::remove_if(lst.begin(), lst.end(), [&](std::uint32_t x) { return edges_[x].src == n; }), lst.end());
        }
        for (std::uint32_t ie : in_[n]) {
            auto &lst = out_[edges_[ie].src];
            lst.erase(std::remove_if(lst.begin(), lst.end(), [&](std::uint32_t x) { return edges_[x].dst == n; }), lst.end());
        }
        out_[n].clear();
        in_[n].clear();
    }
    void markForReaper(std::uint32_t n) {
        nodes_[n].deleted = true;
        clearVertex(n);
    }

    void mergeInNodes(std::uint32_t n) {
        std::map<char, std::vector<std::uint32_t>> nodeGroups;
        for (std::uint32_t ie : in_[n]) {
            const std::uint32_t inNode = edges_[ie].src;
            if (out_[inNode].size() == 1) nodeGroups[nodes_[inNode].base].push_back(inNode);
        }
        for (auto kvp = nodeGroups.cbegin(); kvp != nodeGroups.cend(); ++kvp) {
            const std::vector<std::uint32_t> nodes = kvp->second;
            if (nodes.size() <= 1) continue;
            const std::uint32_t an = nodes[0];
            for (std::size_t x = 1; x < nodes.size(); ++x) {  // accumulate out edge information
                edges_[out_[an].front()].count += edges_[out_[nodes[x]].front()].count;
                nodes_[an].weight += nodes_[nodes[x]].weight;
            }
            for (std::size_t x = 1; x < nodes.size(); ++x) {  // accumulate in edge information, merge nodes
                const std::uint32_t m = nodes[x];
                for (std::size_t y = 0; y < in_[m].size(); ++y) {
                    const std::uint32_t ie = in_[m][y];
                    const std::uint32_t n1 = edges_[ie].src;
                    const int e = findEdge(n1, an);
                    if (e >= 0) {
                        edges_[e].count += edges_[ie].count;
                    } else {
                        const int cnt = edges_[ie].count;
                        const bool vis = edges_[ie].visited;
                        const std::uint32_t ne = addEdgeRaw(n1, an);
                        edges_[ne].count = cnt;
                        edges_[ne].visited = vis;
                    }
                }
                markForReaper(m);
            }
            mergeInNodes(an);
        }
    }
    void mergeOutNodes(std::uint32_t n) {
        std::map<char, std::vector<std::uint32_t>> nodeGroups;
        for (std::uint32_t oe : out_[n]) {
            const std::uint32_t outNode = edges_[oe].dst;
            if (in_[outNode].size() == 1) nodeGroups[nodes_[outNode].base].push_back(outNode);
        }
        for (auto kvp = nodeGroups.cbegin(); kvp != nodeGroups.cend(); ++kvp) {
            const std::vector<std::uint32_t> nodes = kvp->second;
            if (nodes.size() <= 1) continue;
            const std::uint32_t an = nodes[0];
            for (std::size_t x = 1; x < nodes.size(); ++x) {  // accumulate inner edge information
                edges_[in_[an].front()].count += edges_[in_[nodes[x]].front()].count;
                nodes_[an].weight += nodes_[nodes[x]].weight;
            }
            for (std::size_t x = 1; x < nodes.size(); ++x) {  // accumulate and merge outer edge information
                const std::uint32_t m = nodes[x];
                for (std::size_t y = 0; y < out_[m].size(); ++y) {
                    const std::uint32_t oe = out_[m][y];
                    const std::uint32_t n2 = edges_[oe].dst;
                    const int e = findEdge(an, n2);
                    if (e >= 0) {
                        edges_[e].count += edges_[oe].count;
                    } else {
                        const int cnt = edges_[oe].count;
                        const bool vis = edges_[oe].visited;
                        const std::uint32_t ne = addEdgeRaw(an, n2);
                        edges_[ne].count = cnt;
                        edges_[ne].visited = vis;
                    }
                }
                markForReaper(m);
            }
        }
    }

    std::uint32_t bbOf(std::uint32_t v) {  // _bbMap[v] (std::map::operator[]: 0 for a vertex never entered)
        return bbMap_[v];
    }

    std::vector<Node> bestPath() {
        // (edges(_g) only holds the edges that still exist; the flags of erased ones do not matter)
        for (Edge &e : edges_) e.visited = false;
        std::vector<int> bestEdge(nodes_.size(), -1);
        std::vector<float> nodeScore(nodes_.size(), 0.0f);
        std::queue<std::uint32_t> seedNodes;
        seedNodes.push(exit_);
        while (!seedNodes.empty()) {
            const std::uint32_t n = seedNodes.front();
            seedNodes.pop();
            bool bestEdgeFound = false;
            float bestScore = -FLT_MAX;
            int bestEdgeD = -1;
            for (std::uint32_t oe : out_[n]) {
                const std::uint32_t outNodeD = edges_[oe].dst;
                const Node outNode = nodes_[outNodeD];
                float newScore;
                const float score = nodeScore[outNodeD];
                if (outNode.backbone && outNode.weight == 1) {
                    newScore = score - 10.0f;
                } else {
                    const Node bbNode = nodes_[bbOf(outNodeD)];
                    newScore = edges_[oe].count - bbNode.coverage * 0.5f + score;
                }
                if (newScore > bestScore) {
                    bestScore = newScore;
                    bestEdgeD = static_cast<int>(oe);
                    bestEdgeFound = true;
                }
            }
            if (bestEdgeFound) {
                nodeScore[n] = bestScore;
                bestEdge[n] = bestEdgeD;
            }
            for (std::size_t x = 0; x < in_[n].size(); ++x) {
                Edge &inEdge = edges_[in_[n][x]];
                inEdge.visited = true;
                const std::uint32_t inNode = inEdge.src;
                int notVisited = 0;
                for (std::uint32_t oe : out_[inNode])
                    if (!edges_[oe].visited) notVisited++;
                if (notVisited == 0) seedNodes.push(inNode);
            }
        }
        std::vector<Node> bpath;
        std::uint32_t prev = enter_;
        while (true) {
            bpath.push_back(nodes_[prev]);
            if (bestEdge[prev] < 0) break;
            prev = edges_[bestEdge[prev]].dst;
        }
        return bpath;
    }

    std::vector<Node> nodes_;
    std::vector<Edge> edges_;
    std::vector<std::vector<std::uint32_t>> out_, in_;
    std::map<std::uint32_t, std::uint32_t> bbMap_;
    std::uint32_t enter_ = 0, exit_ = 0;
};

// ---------------------------------------------------------------------------------------------------------------------
// command line (args.hxx surface of pa_cns.cpp:23-46)
// ---------------------------------------------------------------------------------------------------------------------
struct Options {
    unsigned threads = 16;
    std::size_t partLen = 5000, topK = 3000, alpha = 250;
    std::string in, out, align;
};
void usage(std::ostream &os) {
    os << "  pa_cns {OPTIONS}\n\n  OPTIONS:\n\n"
          "      -h, --help                        display this help menu\n"
          "      -t[thread_num], --thread=[thread_num]   number of thread\n"
          "      -l[size], --len=[size]            length of part\n"
          "      -k[k], --top_k=[k]                top n\n"
          "      --alpha=[alpha]                   alpha\n"
          "      -i[path], --in=[path]             input of backbone\n"
          "      -o[path], --out=[path]            output of file\n"
          "      -a[path], --align=[path]          alignments file\n";
}
struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HelpRequested {};
template <typename T>
T parseNumber(const std::string &flag, const std::string &v) {
    std::istringstream ss(v);
    T x{};
    ss >> x;
    if (ss.fail() || ss.rdbuf()->in_avail() != 0) throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}
Options parseCli(int argc, char **argv) {
    Options o;
    auto assign = [&](const std::string &name, const std::string &value) {
        if (name == "t" || name == "thread") o.threads = parseNumber<unsigned>(name, value);
        else if (name == "l" || name == "len") o.partLen = parseNumber<std::size_t>(name, value);
        else if (name == "k" || name == "top_k") o.topK = parseNumber<std::size_t>(name, value);
        else if (name == "alpha") o.alpha = parseNumber<std::size_t>(name, value);
        else if (name == "i" || name == "in") o.in = value;
        else if (name == "o" || name == "out") o.out = value;
        else if (name == "a" || name == "align") o.align = value;
        else throw ParseError("Flag could not be matched: " + name);
    };
    auto isLong = [](const std::string &n) {
        for (const char *x : {"thread", "len", "top_k", "alpha", "in", "out", "align"})
            if (n == x) return true;
        return false;
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") throw HelpRequested{};
        if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
            std::string body = a.substr(2), value;
            auto eq = body.find('=');
            if (eq != std::string::npos) {
                value = body.substr(eq + 1);
                body = body.substr(0, eq);
                if (!isLong(body)) throw ParseError("Flag could not be matched: " + body);
            } else {
                if (!isLong(body)) throw ParseError("Flag could not be matched: " + body);
                if (i + 1 >= argc) throw ParseError("Flag '" + body + "' requires an argument but received none");
                value = argv[++i];
            }
            assign(body, value);
        } else if (a.size() >= 2 && a[0] == '-') {
            std::string name(1, a[1]);
            if (std::strchr("tlkioa", a[1]) == nullptr) throw ParseError("Flag could not be matched: '" + name + "'");
            std::string value;
            if (a.size() > 2) value = a.substr(2);
            else {
                if (i + 1 >= argc) throw ParseError("Flag '" + name + "' requires an argument but received none");
                value = argv[++i];
            }
            assign(name, value);
        } else {
            throw ParseError("Passed in argument, but no positional arguments were ready to receive it: " + a);
        }
    }
    return o;
}

}  // namespace

// parts from which the default backend is the device (PA_CNS_BACKEND=auto)
constexpr std::size_t kDevicePartsMin = 4096;

// libpagraph_hip.so, loaded on demand from beside the executable (bin/../libpagraph_hip.so) or the loader's search path
struct HipLibrary {
    using ConsensusFn = int (*)(int, const char *, std::uint64_t, const pag_cns_part *, std::uint64_t, const pag_cns_aln *, std::uint64_t, const char *, const char *,
                                std::uint64_t, std::int32_t, char *, std::uint64_t, std::uint64_t *, std::uint32_t *, std::int32_t *);
    using ErrorFn = const char *(*)();
    void *handle = nullptr;
    ConsensusFn consensus = nullptr;
    ErrorFn last_error = nullptr;
    explicit HipLibrary(const char *argv0) {
        std::string dir = argv0 ? argv0 : "";
        const std::size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
        const std::string beside = dir + "/../libpagraph_hip.so";
        handle = dlopen(beside.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!handle) handle = dlopen("libpagraph_hip.so", RTLD_NOW | RTLD_LOCAL);
        if (!handle) throw std::runtime_error(std::string("the device backend needs libpagraph_hip.so: ") + dlerror());
        consensus = reinterpret_cast<ConsensusFn>(dlsym(handle, "pag_cns_consensus"));
        last_error = reinterpret_cast<ErrorFn>(dlsym(handle, "pag_last_error"));
        if (!consensus || !last_error) throw std::runtime_error("libpagraph_hip.so does not export pag_cns_consensus");
    }
    HipLibrary(const HipLibrary &) = delete;
    HipLibrary &operator=(const HipLibrary &) = delete;
    ~HipLibrary() {
        if (handle) dlclose(handle);
    }
};

int main(int argc, char **argv) {
    if (argc <= 1) {
        usage(std::cerr);
        return 0;
    }
    Options opt;
    try {
        opt = parseCli(argc, argv);
    } catch (const HelpRequested &) {
        usage(std::cerr);
        return 0;
    } catch (const ParseError &e) {
        std::cerr << e.what() << std::endl;
        usage(std::cerr);
        return 1;
    }
    try {
        pagh::SeqDb seqDatabase;
        try {
            seqDatabase = pagh::SeqDb(opt.in);
        } catch (const std::exception &) {  // (a file that cannot be opened is an empty database in the reference, AutoSeqDatabase.cpp:9-22)
        }
        if (seqDatabase.size() == 0) {
            std::cerr << "[Error] No backbone" << std::endl;
            return 2;
        }
        const std::string name = seqDatabase.name(0);
        const std::string backbone = seqDatabase.toString(0, true);
        const std::size_t partLen = opt.partLen;
        const std::size_t partNum = (backbone.size() + partLen - 1) / partLen;
        std::cout << "PartNum=" << partNum << std::endl;
        std::vector<std::string> consensusResults(partNum);
        auto alignments = readFromRefFile(opt.align, backbone, partLen);
        std::atomic<std::size_t> consensusLen(0), next(0);
        std::atomic<bool> failed(false);
        std::string failure;
        // Where the per-part graphs are built.  hip: one device thread per part (csrc/hip/k_cns.hip) — the library is loaded when, and
        // only when, this backend runs: the program starts on a machine without the ROCm runtime; flat: the device's code
        // (csrc/hip/cns_graph.hpp) on host threads; host: the std::vector / std::map restatement.  Default: by the number of parts —
        // a part is a serial chain of dependent accesses, the device wins by running thousands side by side, host threads win on
        // the few hundred parts of one contig of the pipeline (AlignGraph2.py:503 calls pa_cns once per new contig;
        // profiles/r06_pa_cns_timing.json).
        const char *backendEnv = std::getenv("PA_CNS_BACKEND");
        std::string backend = backendEnv ? backendEnv : "auto";
        if (backend == "auto") backend = partNum >= kDevicePartsMin ? "hip" : "flat";
        if (backend != "hip" && backend != "flat" && backend != "host") throw std::runtime_error("PA_CNS_BACKEND must be hip, flat, host or auto");
        const bool onHost = backend == "host";
        std::vector<std::vector<std::size_t>> partWeights(partNum);
        auto work = [&]() {
            for (std::size_t i; (i = next.fetch_add(1)) < partNum;) {
                try {
                    auto &alns = alignments[i];
                    {   // std::sort by score, descending: the same algorithm on a proxy array (the tie order is the library's)
                        struct Proxy {
                            unsigned score;
                            std::size_t idx;
                        };
                        std::vector<Proxy> proxy(alns.size());
                        for (std::size_t x = 0; x < alns.size(); ++x) proxy[x] = Proxy{alns[x].score, x};
                        std::sort(proxy.begin(), proxy.end(), [](const Proxy &l, const Proxy &r) { return l.score > r.score; });
                        std::vector<ScoredAln> sorted;
                        sorted.reserve(std::min(alns.size(), opt.topK));
                        for (std::size_t x = 0; x < proxy.size() && x < opt.topK; ++x) sorted.push_back(std::move(alns[proxy[x].idx]));
                        alns.swap(sorted);
                    }
                    const std::size_t left = i * partLen, right = std::min((i + 1) * partLen, backbone.size());
                    partWeights[i] = weightAln(alns, opt.alpha);
                    if (!onHost) continue;  // (the graphs of all parts are built together below)
                    const auto &weights = partWeights[i];
                    AlnGraph g(backbone.substr(left, right - left));
                    for (std::size_t x = 0; x < weights.size(); ++x) g.addAln(alns[x].aln, static_cast<int>(weights[x]));
                    g.mergeNodes();
                    consensusResults[i] = g.consensus();
                    consensusLen += consensusResults[i].size();
                    std::vector<ScoredAln>().swap(alns);
                } catch (const std::exception &e) {
                    if (!failed.exchange(true)) failure = e.what();
                }
            }
        };
        const unsigned nThreads = std::max(1u, std::min<unsigned>(std::max(1u, pagh::usableCpus()), static_cast<unsigned>(std::max<std::size_t>(1, partNum))));
        {
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
        }
        if (failed) throw std::runtime_error(failure);
        if (!onHost) {
            // the parts as flat arrays: every alignment's two rows in two pools, the regions of every part's graph sized from its
            // columns (include/pagraph_hip.h, pag_cns_consensus)
            std::vector<pag_cns_part> parts(partNum);
            std::vector<pag_cns_aln> flat;
            std::string qpool, tpool;
            std::uint64_t outBytes = 0;
            for (std::size_t i = 0; i < partNum; ++i) {
                const std::size_t left = i * partLen, right = std::min((i + 1) * partLen, backbone.size());
                pag_cns_part &P = parts[i];
                P.bb_off = left;
                P.bb_len = static_cast<std::uint32_t>(right - left);
                P.aln_first = flat.size();
                std::uint64_t nIns = 0, nEdgeCols = 0;
                const auto &alns = alignments[i];
                for (std::size_t x = 0; x < partWeights[i].size(); ++x) {
                    const Aln &a = alns[x].aln;
                    pag_cns_aln f{};
                    f.str_off = qpool.size();
                    f.len = static_cast<std::uint32_t>(a.qstr.size());
                    f.start = a.start;
                    f.weight = static_cast<std::int32_t>(partWeights[i][x]);
                    flat.push_back(f);
                    qpool += a.qstr;
                    tpool += a.tstr;
                    for (std::size_t c = 0; c < a.qstr.size(); ++c) {
                        const char qb = a.qstr[c], tb = a.tstr[c];
                        if (qb != '-' && tb == '-') ++nIns;
                        if (qb != '-') ++nEdgeCols;
                    }
                }
                std::vector<ScoredAln>().swap(alignments[i]);  // (its rows live in the pools now)
                P.n_aln = static_cast<std::uint32_t>(flat.size() - P.aln_first);
                const std::uint64_t nodeCap = P.bb_len + 2ull + nIns, edgeCap = P.bb_len + 1ull + nEdgeCols + P.n_aln + 4096ull;
                if (nodeCap > 0x7FFFFFFFull || edgeCap > 0x7FFFFFFFull) throw std::runtime_error("pa_cns: a part with more than 2^31 graph slots");
                P.node_cap = static_cast<std::uint32_t>(nodeCap);
                P.edge_cap = static_cast<std::uint32_t>(edgeCap);
                P.aux_cap = static_cast<std::uint32_t>(std::min<std::uint64_t>(4 * nodeCap + 2048, 0x7FFFFFFFull));
                P.out_cap = P.node_cap;
                outBytes += P.out_cap;
            }
            std::vector<char> outBuf(outBytes + 16);
            std::vector<std::uint64_t> outOff(partNum + 1, 0);
            std::vector<std::uint32_t> outLen(partNum, 0);
            std::vector<std::int32_t> partErr(partNum, 0);
            if (backend == "hip") {
                const int device = static_cast<int>(pagh::envInt("PAGRAPH_DEVICE", 0));
                const HipLibrary hip(argv[0]);  // (throws when the library or a gfx950 device is missing: there is no silent fallback)
                const int rc = hip.consensus(device, backbone.data(), backbone.size(), parts.data(), partNum, flat.data(), flat.size(), qpool.data(), tpool.data(),
                                             qpool.size(), 0, outBuf.data(), outBytes, outOff.data(), outLen.data(), partErr.data());
                if (rc != 0) throw std::runtime_error(std::string("pag_cns_consensus failed (") + std::to_string(rc) + "): " + hip.last_error());
            } else {  // the device's code on host threads
                std::uint64_t oo = 0;
                for (std::size_t i = 0; i < partNum; ++i) {
                    outOff[i] = oo;
                    oo += parts[i].out_cap;
                }
                outOff[partNum] = oo;
                std::atomic<std::size_t> nextPart(0);
                auto flatWork = [&]() {
                    for (std::size_t i; (i = nextPart.fetch_add(1)) < partNum;) {
                        const pag_cns_part &S = parts[i];
                        std::vector<std::uint8_t> nb(S.node_cap), nf(S.node_cap), ev(S.edge_cap);
                        std::vector<std::int32_t> ncov(S.node_cap), nw(S.node_cap), nbest(S.node_cap), ec(S.edge_cap);
                        std::vector<std::uint32_t> nu[7], eu[6], aux(S.aux_cap);
                        for (auto &v : nu) v.resize(S.node_cap);
                        for (auto &v : eu) v.resize(S.edge_cap);
                        std::vector<float> nscore(S.node_cap);
                        pagcns::Arrays A{nb.data(), nf.data(), ncov.data(), nw.data(), nu[0].data(), nu[1].data(), nu[2].data(), nu[3].data(), nu[4].data(), nu[5].data(),
                                         nu[6].data(), nscore.data(), nbest.data(), eu[0].data(), eu[1].data(), eu[2].data(), eu[3].data(), eu[4].data(), eu[5].data(),
                                         ec.data(), ev.data(), aux.data()};
                        pagcns::Part P{};
                        P.bb_off = S.bb_off;
                        P.bb_len = S.bb_len;
                        P.n_aln = S.n_aln;
                        P.aln_first = S.aln_first;
                        P.out_off = outOff[i];
                        P.node_cap = S.node_cap;
                        P.edge_cap = S.edge_cap;
                        P.aux_cap = S.aux_cap;
                        P.out_cap = S.out_cap;
                        static_assert(sizeof(pag_cns_aln) == sizeof(pagcns::Aln), "pag_cns_aln is pagcns::Aln");
                        std::uint32_t len = 0;
                        partErr[i] = pagcns::run_part(A, P, backbone.data(), reinterpret_cast<const pagcns::Aln *>(flat.data()), qpool.data(), tpool.data(), 0, outBuf.data(), &len);
                        outLen[i] = partErr[i] ? 0 : len;
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(flatWork);
                flatWork();
                for (auto &t : pool) t.join();
            }
            for (std::size_t i = 0; i < partNum; ++i) {
                if (partErr[i] == pagcns::CNS_E_OVERRUN) throw std::runtime_error("pa_cns: an alignment runs past the end of its part");
                if (partErr[i]) throw std::runtime_error("pa_cns: part " + std::to_string(i) + " ran out of its graph regions (code " + std::to_string(partErr[i]) + ")");
                consensusResults[i].assign(outBuf.data() + outOff[i], outLen[i]);
                consensusLen += outLen[i];
            }
        }
        std::cout << consensusLen << std::endl;
        std::cout << backbone.size() << std::endl;
        std::ofstream of(opt.out);
        std::string text;
        text.reserve(backbone.size() + backbone.size() / 70 + name.size() + 8);
        text += ">" + name + "\n";
        std::size_t cnt = 0;
        for (auto &seq : consensusResults)
            for (char ch : seq) {
                text += ch;
                if (++cnt % 70 == 0) {
                    text += '\n';
                    cnt = 0;
                }
            }
        if (cnt > 0) text += '\n';
        of << text;
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "pa_cns: " << e.what() << std::endl;
        return 1;
    }
}
