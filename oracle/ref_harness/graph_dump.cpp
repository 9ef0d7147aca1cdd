// oracle/ref_harness/graph_dump.cpp — TEST INFRASTRUCTURE ONLY (built into oracle/_ref/, never shipped).
//
// Drives the REFERENCE's own classes (compiled from /root/reference/PAGraph/src/tools, see
// oracle/Makefile) through the graph-build half of pagraph (the part of run2() that ends with
// PositionProcessor::process(), reference PAGraph/src/main/pagraph.cpp:129-239) and then dumps the
// complete positional A-Bruijn graph, which the reference itself never writes anywhere.  The dump is
// what pins oracle/pag_oracle.c (and through it the HIP build) at graph level: every k-mer node's
// clustered positions, their u16 counts and its de-duplicated out-edges.
//
// usage: graph_dump -t T -k kmer.bin -c ctg.fasta -R ref.fasta -p predir -a aln -o outdir
//                   [--epsilon E] [-v V]
// writes outdir/<P>.graph.txt per config block P:
//   S <merge_edge_1> <total_pos_1> <merge_pos_1> <merge_edge_2> <total_pos_2> <merge_pos_2>
//   K <code> <npos> <nchild>          (only nodes with npos+nchild > 0, ascending code)
//   P <ctgSingle> <refSingle> <count>
//   C <toCode> <step>
// with --succ 1 also outdir/<P>.succ.txt: what PABruijnGraph::successors (the reference's epsilon-join,
// PABruijnGraph.cpp:167-197, called as the traversal calls it: deviation = 2 * epsilon, error rate 0.15,
// pagraph.cpp:251, PAlgorithm.tcc:41) returns for EVERY vertex of the finished graph, in its own order:
//   V <code> <posIndex> <n>           (every vertex, nodes ascending by code, positions ascending)
//   T <toCode> <toPosIndex> <step> <grade> <ctgSimilar>     n lines; grade = checkPosition (:143-165: 1 Skip,
//                                     2 Good, 3 Excellent, 4 Amazing), ctgSimilar = isEdgeSimilar(...).first (:385-400)
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <regex>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#define private public
#include "graph/PABruijnGraph.hpp"
#include "position/PositionProcessor.hpp"
#undef private
#include "align/MecatAlignDatabase.hpp"
#include "align/MummerAlignDatabaseV2.hpp"
#include "kmer/FileKmerIterator.hpp"
#include "seq/AutoSeqDatabase.hpp"

namespace {

struct Block {
    std::string ref, reads, ctgAln, refAln;
    std::vector<std::pair<std::string, bool>> contigs;
};

// config.txt layout: reference pagraph.cpp:29-49
std::vector<Block> readConfig(const std::string &path) {
    std::vector<Block> out;
    std::ifstream in(path);
    std::string line;
    while (std::getline(in, line)) {
        Block b;
        b.ref = line;
        std::getline(in, b.reads);
        std::getline(in, b.ctgAln);
        std::getline(in, b.refAln);
        while (std::getline(in, line) && !line.empty()) {
            std::string name = line;
            std::getline(in, line);
            bool fwd = false;
            std::stringstream(line) >> fwd;
            b.contigs.emplace_back(name, fwd);
        }
        out.push_back(b);
    }
    return out;
}

const char *argOf(int argc, char **argv, const char *flag, const char *dflt) {
    for (int i = 1; i + 1 < argc; ++i)
        if (std::strcmp(argv[i], flag) == 0) return argv[i + 1];
    return dflt;
}

}  // namespace

int main(int argc, char **argv) {
    unsigned threads = static_cast<unsigned>(std::atoi(argOf(argc, argv, "-t", "1")));
    std::string kmerPath = argOf(argc, argv, "-k", "");
    std::string ctgPath = argOf(argc, argv, "-c", "");
    std::string refPath = argOf(argc, argv, "-R", "");
    std::string preDir = argOf(argc, argv, "-p", "");
    std::string alnPath = argOf(argc, argv, "-a", "");
    std::string outDir = argOf(argc, argv, "-o", ".");
    std::size_t eps = static_cast<std::size_t>(std::atoll(argOf(argc, argv, "--epsilon", "10")));
    std::size_t cov = static_cast<std::size_t>(std::atoll(argOf(argc, argv, "-v", "1")));
    bool dumpSucc = std::atoi(argOf(argc, argv, "--succ", "0")) != 0;

    auto blocks = readConfig(preDir + "/config.txt");
    auto kmerIt = std::make_shared<FileKmerIterator>(kmerPath);
    auto ctgDB = std::make_shared<AutoSeqDatabase>(ctgPath);
    auto refDB = std::make_shared<AutoSeqDatabase>(refPath);
    auto ctgToRef = std::make_shared<MummerAlignDatabaseV2>(alnPath);
    auto graph = std::make_shared<PABruijnGraph>(*kmerIt, threads);

    std::size_t blockNo = 0;
    for (auto &b : blocks) {
        graph->resetAllNodes(threads);
        auto readDB = std::make_shared<AutoSeqDatabase>(preDir + "/" + b.reads);
        auto readToCtg = std::make_shared<MecatAlignDatabase>(preDir + "/" + b.ctgAln);
        auto readToRef = std::make_shared<MecatAlignDatabase>(preDir + "/" + b.refAln);

        PositionProcessor pp(graph, readDB, ctgDB, refDB, readToCtg, readToRef, ctgToRef);
        // same knob values as the reference main (pagraph.cpp:110-125, 205-231)
        pp.setReadToCtgTopK(-1);
        pp.setReadToRefTopK(-1);
        pp.setCtgToRefTopK(-1);
        pp.setOuterSample(3);
        pp.setInnerSample(1);
        pp.setPositionError(eps);
        pp.setReadToCtgRatio(0.35);
        pp.setReadToRefRatio(0.10);
        pp.setCtgToRefRatio(0.00);
        pp.setCtgToRefTotalRatio(0.1);
        pp.setCtgToRefMinLen(50);
        pp.setCovFilter(cov);
        pp.setThreadNum(threads);
        pp.clearRefFilter(false);
        pp.clearCtgFilter(false);
        pp.setRefFilter(b.ref, true);
        for (auto &c : b.contigs) pp.setCtgFilter(c.first, c.second, true);
        pp.preProcess();

        // capture the six count lines process() prints
        std::stringstream captured;
        auto *old = std::cout.rdbuf(captured.rdbuf());
        pp.process();
        std::cout.rdbuf(old);
        std::vector<std::string> stats;
        {
            std::string line;
            while (std::getline(captured, line)) {
                auto eq = line.find(" = ");
                if (eq != std::string::npos && line[0] == '\t') stats.push_back(line.substr(eq + 3));
            }
        }

        std::ofstream out(outDir + "/" + std::to_string(blockNo) + ".graph.txt");
        out << "S";
        for (auto &s : stats) out << " " << s;
        out << "\n";
        auto &table = *graph->_pDenseHashTable;
        for (std::size_t i = 0; i < table.size(); ++i) {
            auto &node = table[i];
            auto &pos = node.getAllPositions();
            auto &cnt = node.getAllCount();
            auto &chd = node.getAllChild();
            if (pos.empty() && chd.empty()) continue;
            out << "K " << graph->_kmerIndexArr[i] << " " << pos.size() << " " << chd.size() << "\n";
            for (std::size_t j = 0; j < pos.size(); ++j)
                out << "P " << pos[j].first << " " << pos[j].second << " " << cnt[j] << "\n";
            for (auto &c : chd) out << "C " << graph->_kmerIndexArr[c.first] << " " << c.second << "\n";
        }
        if (dumpSucc) {
            std::ofstream so(outDir + "/" + std::to_string(blockNo) + ".succ.txt");
            const PABruijnGraph::PosType deviation = static_cast<PABruijnGraph::PosType>(2 * eps);
            const double errorRate = 0.15;
            std::vector<std::pair<PABruijnGraph::PANode, int>> res;
            for (std::size_t i = 0; i < table.size(); ++i) {
                auto &node = table[i];
                PABruijnGraph::ANode aNode(node, i);
                for (std::size_t j = 0; j < aNode.size(); ++j) {
                    PABruijnGraph::PANode pa(aNode, j);
                    res.clear();
                    graph->successors(res, pa, deviation, errorRate);
                    so << "V " << graph->_kmerIndexArr[i] << " " << j << " " << res.size() << "\n";
                    for (auto &r : res) {
                        auto &t = r.first;
                        auto grade = PABruijnGraph::checkPosition(pa.getPosition(), t.getPosition(), r.second, deviation, errorRate);
                        auto es = PABruijnGraph::isEdgeSimilar(pa.getPosition(), t.getPosition(), r.second, deviation, errorRate);
                        so << "T " << graph->_kmerIndexArr[t.getABruijnNode().getIndex()] << " " << t.getPosIndex() << " " << r.second
                           << " " << static_cast<int>(grade) << " " << (es.first ? 1 : 0) << "\n";
                    }
                }
            }
        }
        ++blockNo;
    }
    return 0;
}
