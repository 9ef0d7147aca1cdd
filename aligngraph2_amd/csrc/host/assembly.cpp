#include "assembly.hpp"
#include "host_threads.hpp"

#include <functional>

#include <atomic>
#include <charconv>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <thread>
#include <vector>

#include "traversal.hpp"

namespace pagh {

namespace {

// UnionSet (UnionSet.cpp:7-20)
struct UnionSet {
    std::vector<std::size_t> parent;
    explicit UnionSet(std::size_t n) : parent(n) {
        for (std::size_t i = 0; i < n; ++i) parent[i] = i;
    }
    std::size_t find(std::size_t p) { return p == parent[p] ? p : (parent[p] = find(parent[p])); }
    void unionTwo(std::size_t left, std::size_t right) { parent[find(right)] = find(left); }
};

// PAssembly::combatSeq (PAssembly.tcc:2-31): follow the chain of leaps from contig to contig
template <typename F>
void combatSeq(const std::vector<TravelSequence> &seqs, const HostGraph &g, const PositionMapper &mapper, std::size_t start,
               bool forward, F functor) {
    std::size_t nextIdx = start * 2 + (forward ? 0 : 1);
    std::size_t nextPos = 0;
    std::set<std::size_t> seen;
    seen.insert(nextIdx);
    while (true) {
        if (!functor(nextIdx / 2, nextIdx % 2 == 0, nextPos)) break;
        if (seqs[nextIdx].empty() || g.position(seqs[nextIdx].back().first).first == 0) break;
        auto last = mapper.singleToDual(g.position(seqs[nextIdx].back().first).first);
        nextPos = static_cast<std::size_t>(last.second);
        nextIdx = static_cast<std::size_t>(std::llabs(last.first) - 1) * 2 + (last.first > 0 ? 0 : 1);
        if (seen.count(nextIdx) > 0) break;
        seen.insert(nextIdx);
    }
}

}  // namespace

std::set<std::pair<std::string, bool>> assemble(const std::string &outDir, const std::string &prefix, const HostGraph &graph,
                                                const SeqDb &contigs, const SeqDb &refs, const PositionMapper &ctgMapper,
                                                const PositionMapper &refMapper,
                                                const std::set<std::pair<std::string, bool>> &ctgSet, std::size_t deviation,
                                                double errorRate, double startSplit, std::size_t minLen, unsigned threadNum,
                                                unsigned hostThreads, AssembleStats *stats, bool quiet,
                                                std::vector<TravelSequence> &travelled, std::ostream *logTo, const AssembleShare *share) {
    (void)minLen;     // (both only steer the walk itself, which has already happened: PAlgorithm::travelSequence on the device)
    (void)threadNum;
    const bool timing = pagh::envTiming();
    auto nowMs = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tLap = nowMs();
    auto lap = [&](const char *what) {
        if (!timing) return;
        double t = nowMs();
        std::fprintf(stderr, "[timing] assemble %s %.1f ms\n", what, t - tLap);
        tLap = t;
    };
    std::ostream nullOut(nullptr);
    std::ostream &out = quiet ? nullOut : logTo ? *logTo : std::cout;  // (logTo: a block assembled beside the next block's device work keeps its log in one piece)
    std::set<std::pair<std::string, bool>> success;
    std::vector<TravelSequence> results(contigs.size() * 2);
    std::vector<std::size_t> inDegrees(contigs.size() * 2);

    // per selected contig: traverse, dump the path, drop short paths, count leap targets
    // (PAssembly.cpp:30-79).  Contigs are independent, so the loop runs on a pool of host threads; the
    // in-degree bookkeeping and the log are replayed in set order afterwards.
    std::vector<std::pair<std::string, bool>> ctgList(ctgSet.begin(), ctgSet.end());
    std::vector<std::string> logs(ctgList.size());
    std::vector<std::int64_t> leapTarget(ctgList.size(), -1);
    // The path dumps are text, ~60 bytes per path vertex (a 1.5 Mb contig alone is 370 k lines): every contig's lines are
    // rendered in chunks of 16 Ki vertices by a pool of host threads, each into its own reusable buffer, and appended to
    // the contig's file in chunk order (a thread waits for its chunk's turn; the chunks are handed out in order, so the
    // wait is at most one chunk's rendering).
    const std::size_t CHUNK = 16384;
    struct DumpFile {
        std::size_t ctgIdx = 0, ctgOffset = 0, nChunks = 0;
        bool written = true;
        std::FILE *f = nullptr;
        std::atomic<std::size_t> turn{0};
    };
    std::vector<DumpFile> files(ctgList.size());
    std::vector<std::pair<std::size_t, std::size_t>> chunkOf;  // (file, chunk in file)
    for (std::size_t li = 0; li < ctgList.size(); ++li) {
        DumpFile &F = files[li];
        F.ctgIdx = contigs.id(ctgList[li].first);
        F.ctgOffset = ctgList[li].second ? 0 : 1;
        auto &res = results[2 * F.ctgIdx + F.ctgOffset];
        // PAlgorithm::travelSequence ran on the device (pag_travel); its result is taken over here and its storage
        // handed back at the end: no copy
        if (2 * F.ctgIdx + F.ctgOffset < travelled.size()) res.swap(travelled[2 * F.ctgIdx + F.ctgOffset]);
        F.nChunks = (res.size() + CHUNK - 1) / CHUNK;
        F.written = !(share && share->writesDump) || share->writesDump(F.ctgIdx);  // (a sharded run: the rank that walked the contig writes its dump)
        if (!F.written) F.nChunks = 0;
    }
    // chunk c of every file before chunk c + 1 of any: the threads then append to as many different files as there are
    // threads (writing new pages of a file is what the kernel serialises)
    for (std::size_t c = 0;; ++c) {
        bool any = false;
        for (std::size_t li = 0; li < files.size(); ++li)
            if (c < files[li].nChunks) {
                chunkOf.emplace_back(li, c);
                any = true;
            }
        if (!any) break;
    }
    auto runPool = [&](std::size_t nTasks, const std::function<void(std::size_t)> &task) {
        std::atomic<std::size_t> next{0};
        auto worker = [&]() {
            for (std::size_t x; (x = next.fetch_add(1)) < nTasks;) task(x);
        };
        unsigned nThreads = hostThreads ? hostThreads : std::min(64u, std::max(1u, usableCpus()));
        nThreads = static_cast<unsigned>(std::min<std::size_t>(nThreads, std::max<std::size_t>(1, nTasks)));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    };
    // (opening truncates: giving back the pages of a previous run's files is work the kernel does per file, so in parallel)
    std::atomic<bool> openFailed{false};
    runPool(files.size(), [&](std::size_t li) {
        DumpFile &F = files[li];
        if (!F.written) return;
        const std::string path = outDir + "/" + prefix + std::to_string(F.ctgIdx) + "_" + std::to_string(F.ctgOffset) + ".txt";
        F.f = std::fopen(path.c_str(), "wb");
        if (!F.f) {
            openFailed = true;
            return;
        }
        std::fprintf(F.f, "%s\t%zu\n", ctgList[li].first.c_str(), static_cast<std::size_t>(contigs.length(F.ctgIdx)));
    });
    if (openFailed) {
        for (auto &F : files)
            if (F.f) std::fclose(F.f);
        throw std::runtime_error("cannot write the path dumps into " + outDir);
    }
    runPool(chunkOf.size(), [&](std::size_t x) {
        DumpFile &F = files[chunkOf[x].first];
        const std::size_t c = chunkOf[x].second;
        const auto &res = results[2 * F.ctgIdx + F.ctgOffset];
        const std::size_t from = c * CHUNK, to = std::min(res.size(), from + CHUNK);
        SeqTools algo(graph, contigs, refs, ctgMapper, refMapper);
        static thread_local std::string buf;
        buf.clear();
        buf.reserve(CHUNK * 72);
        char num[24];
        auto putInt = [&](long long v) {
            auto r = std::to_chars(num, num + sizeof num, v);
            buf.append(num, static_cast<std::size_t>(r.ptr - num));
        };
        for (std::size_t q = from; q < to; ++q) {
            const auto &s = res[q];
            DualPos p = graph.position(s.first);
            auto d1 = ctgMapper.singleToDual(p.first);
            auto d2 = refMapper.singleToDual(p.second);
            buf.append(algo.vertexString(s.first));
            buf.push_back('\t');
            putInt(s.second);
            buf.push_back('\t');
            putInt(d1.first);
            buf.push_back(',');
            putInt(d1.second);
            buf.push_back('\t');
            putInt(d2.first);
            buf.push_back(',');
            putInt(d2.second);
            buf.push_back('\n');
        }
        while (F.turn.load(std::memory_order_acquire) != c) std::this_thread::yield();
        std::fwrite(buf.data(), 1, buf.size(), F.f);
        F.turn.store(c + 1, std::memory_order_release);
    });
    for (std::size_t li = 0; li < ctgList.size(); ++li) {
        DumpFile &F = files[li];
        if (F.f) std::fclose(F.f);
    }
    if (share && share->dumpsOnly) {
        lap("per-contig dumps of this rank's contigs");
        return {};
    }
    for (std::size_t li = 0; li < ctgList.size(); ++li) {
        DumpFile &F = files[li];
        std::stringstream log;
        log << "[Travel] " << F.ctgIdx << " - " << contigs.name(F.ctgIdx) << " - " << contigs.length(F.ctgIdx) << "\n";
        log << "[Travel] " << (F.ctgOffset == 0 ? "forward" : "reverse") << "\n";
        auto &res = results[2 * F.ctgIdx + F.ctgOffset];
        if (SeqTools::seqSize(res) < contigs.length(F.ctgIdx) * startSplit * 0.9) res.clear();
        if (!res.empty()) {
            std::uint32_t lastCtgPos = graph.position(res.back().first).first;
            if (lastCtgPos != 0) {
                auto dual = ctgMapper.singleToDual(lastCtgPos);
                std::size_t idx = static_cast<std::size_t>(std::llabs(dual.first) - 1);
                std::size_t fwd = dual.first > 0 ? 0 : 1;
                if (idx != F.ctgIdx || fwd != F.ctgOffset) leapTarget[li] = static_cast<std::int64_t>(2 * idx + fwd);
            }
        }
        log << "[Travel] End\n";
        logs[li] = log.str();
    }
    lap("per-contig paths + dumps");
    for (std::size_t li = 0; li < ctgList.size(); ++li) {
        out << logs[li];
        if (leapTarget[li] >= 0) ++inDegrees[static_cast<std::size_t>(leapTarget[li])];
    }

    // leaps into contigs whose own path was dropped are cut off again (PAssembly.cpp:129-150)
    for (auto &ctgName : ctgSet) {
        std::size_t ctgIdx = contigs.id(ctgName.first);
        std::size_t ctgOffset = ctgName.second ? 0 : 1;
        auto &res = results[2 * ctgIdx + ctgOffset];
        if (res.empty()) continue;
        std::uint32_t lastCtgPos = graph.position(res.back().first).first;
        if (lastCtgPos == 0) continue;
        auto dual = ctgMapper.singleToDual(lastCtgPos);
        std::size_t idx = static_cast<std::size_t>(std::llabs(dual.first) - 1);
        std::size_t fwd = dual.first > 0 ? 0 : 1;
        if ((idx != ctgIdx || fwd != ctgOffset) && results[2 * idx + fwd].empty()) {
            res.pop_back();
            --inDegrees[2 * idx + fwd];
        }
    }
    for (std::size_t i = 0; i < inDegrees.size(); ++i)
        if (inDegrees[i] > 0)
            out << "\t" << (i / 2) << " " << (i % 2 == 0 ? "forward" : "reverse") << " " << inDegrees[i] << std::endl;

    // union-find over chained contigs (PAssembly.cpp:160-198)
    out << "[Union] Start" << std::endl;
    std::map<std::pair<std::string, bool>, std::size_t> helper;
    std::vector<std::pair<std::string, bool>> table;
    for (auto &c : ctgSet) {
        helper[c] = table.size();
        table.push_back(c);
    }
    std::vector<bool> touched(helper.size(), false);
    UnionSet us(helper.size());
    for (auto &ctgName : ctgSet) {
        std::size_t ctgIdx = contigs.id(ctgName.first);
        std::size_t i = ctgIdx * 2 + (ctgName.second ? 0 : 1);
        if (inDegrees[i] > 0 || results[i].empty()) continue;
        std::size_t mainIdx = helper[ctgName];
        touched[mainIdx] = true;
        combatSeq(results, graph, ctgMapper, i / 2, i % 2 == 0, [&](std::size_t ctgId, bool forward, std::size_t) -> bool {
            if (ctgId != ctgIdx || forward != ctgName.second) {
                std::size_t h = helper[{contigs.name(ctgId), forward}];  // operator[]: inserts 0 for unknown pairs, as the reference
                us.unionTwo(h, mainIdx);
                if (touched[h]) return false;
                touched[h] = true;
                return true;
            }
            return true;
        });
    }
    std::vector<std::vector<std::size_t>> merged(helper.size());
    for (std::size_t i = 0; i < table.size(); ++i) merged[us.find(i)].push_back(i);

    // per group keep the start with the longest chain (PAssembly.cpp:207-232)
    std::set<std::pair<std::string, bool>> starts;
    for (auto &group : merged) {
        if (group.empty()) continue;
        std::size_t maxSize = 0, chosen = group.front();
        for (auto idx : group) {
            auto &ctgName = table[idx];
            std::size_t ctgIdx = contigs.id(ctgName.first);
            std::size_t i = ctgIdx * 2 + (ctgName.second ? 0 : 1);
            if (inDegrees[i] > 0 || results[i].empty()) continue;
            std::size_t len = 0;
            combatSeq(results, graph, ctgMapper, i / 2, i % 2 == 0, [&](std::size_t ctgId, bool, std::size_t) -> bool {
                len += contigs.length(ctgId);
                return true;
            });
            if (len > maxSize) {
                maxSize = len;
                chosen = idx;
            }
        }
        starts.insert(table[chosen]);
    }
    out << "[Union] End" << std::endl;
    out << "Start From:" << std::endl;
    for (auto &c : starts) out << "\t" << c.first << " " << c.second << std::endl;

    lap("union/starts");
    // emit chains (PAssembly.cpp:242-333)
    out << "[Assembly] Start" << std::endl;
    std::size_t nameCnt = 0;
    SeqTools algo(graph, contigs, refs, ctgMapper, refMapper);
    // Pass 1 (serial, cheap): which chains are written, under which names, from which pieces — and the log lines, in the
    // reference's order.  Pass 2: every consensus piece of every chain is rendered by a pool of host threads (a chain is
    // often ONE contig: a pool per chain would render the chains one after the other), then the chains' files are written
    // side by side.
    struct ChainOut {
        std::size_t i = 0;
        std::string name, base;
        std::vector<std::size_t> order;  // results[] indices of its pieces, in chain order
        std::vector<std::pair<std::pair<std::string, bool>, std::size_t>> conInf;
        std::size_t totalLen = 0, maxLen = 0, cmbLen = 0, firstPiece = 0;
    };
    std::vector<ChainOut> chains;
    std::size_t nPieces = 0;
    for (auto &ctgName : starts) {
        std::size_t ctgIdx = contigs.id(ctgName.first);
        std::size_t i = ctgIdx * 2 + (ctgName.second ? 0 : 1);
        if (inDegrees[i] > 0 || results[i].empty()) continue;
        std::string name = prefix + std::to_string(nameCnt++);

        std::set<std::pair<std::size_t, bool>> connected;
        std::size_t maxLen = 0, totalLen = 0;
        combatSeq(results, graph, ctgMapper, i / 2, i % 2 == 0, [&](std::size_t ctgId, bool forward, std::size_t) -> bool {
            connected.emplace(ctgId, forward);
            maxLen = std::max<std::size_t>(maxLen, contigs.length(ctgId));
            totalLen += SeqTools::seqSize(results[ctgId * 2 + (forward ? 0 : 1)]);
            return true;
        });
        bool isConnected = connected.size() > 1 && totalLen > maxLen * 1.05;
        bool isExtended = connected.size() == 1 && SeqTools::seqSize(results[i]) > contigs.length(ctgIdx) * 1.2;
        if (!(isConnected || isExtended)) {
            out << "Ignore output" << std::endl;
            continue;
        }
        ChainOut co;
        co.i = i;
        co.name = name;
        co.base = outDir + "/" + prefix + std::to_string(i / 2) + "_" + std::to_string(i % 2);
        co.totalLen = totalLen;
        co.maxLen = maxLen;
        co.firstPiece = nPieces;
        combatSeq(results, graph, ctgMapper, i / 2, i % 2 == 0, [&](std::size_t ctgId, bool forward, std::size_t) -> bool {
            out << i << "=" << ctgId << std::endl;
            co.conInf.push_back({{contigs.name(ctgId), forward}, contigs.length(ctgId)});
            co.order.push_back(ctgId * 2 + (forward ? 0 : 1));
            return true;
        });
        nPieces += co.order.size();
        out << "Out file: " << co.base << ".fasta" << std::endl;
        for (auto &s : connected) success.emplace(contigs.name(s.first), s.second);
        chains.push_back(std::move(co));
    }
    {
        std::vector<std::string> pieces(nPieces);
        std::vector<std::pair<std::size_t, std::size_t>> pieceOf(nPieces);  // (chain, index inside it)
        for (std::size_t c = 0; c < chains.size(); ++c)
            for (std::size_t x = 0; x < chains[c].order.size(); ++x) pieceOf[chains[c].firstPiece + x] = {c, x};
        auto runPool = [&](std::size_t n, const std::function<void(std::size_t)> &fn) {
            std::atomic<std::size_t> next{0};
            auto worker = [&]() {
                for (std::size_t x; (x = next.fetch_add(1)) < n;) fn(x);
            };
            unsigned nThreads = hostThreads ? hostThreads : std::max(1u, usableCpus());
            nThreads = static_cast<unsigned>(std::min<std::size_t>(std::min(nThreads, 64u), std::max<std::size_t>(1, n)));
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
        };
        runPool(nPieces, [&](std::size_t x) {
            const ChainOut &co = chains[pieceOf[x].first];
            pieces[x] = algo.seqToString(results[co.order[pieceOf[x].second]], deviation, errorRate);
        });
        runPool(chains.size(), [&](std::size_t c) {
            ChainOut &co = chains[c];
            const std::size_t lineSize = 70;
            {
                std::ofstream help(co.base + ".help");
                help << co.totalLen << "\n" << co.maxLen << "\n";
            }
            std::ofstream fasta(co.base + ".fasta");
            std::ofstream con(co.base + ".con");
            fasta << ">" << co.name << "\n";
            std::size_t cmbLen = 0;
            std::string line;
            line.reserve(lineSize + 1);
            std::string buf;
            buf.reserve(1 << 20);
            for (std::size_t x = 0; x < co.order.size(); ++x) {
                std::string &piece = pieces[co.firstPiece + x];
                cmbLen += piece.size();
                std::size_t at = 0;
                while (at < piece.size()) {
                    std::size_t take = std::min(lineSize - line.size(), piece.size() - at);
                    line.append(piece, at, take);
                    at += take;
                    if (line.size() == lineSize) {
                        buf.append(line);
                        buf.push_back('\n');
                        line.clear();
                        if (buf.size() >= (1 << 20) - lineSize - 1) {
                            fasta.write(buf.data(), static_cast<std::streamsize>(buf.size()));
                            buf.clear();
                        }
                    }
                }
                std::string().swap(piece);
            }
            if (!line.empty()) {
                buf.append(line);
                buf.push_back('\n');
            }
            fasta.write(buf.data(), static_cast<std::streamsize>(buf.size()));
            con << co.name << "\t" << cmbLen << "\n";
            for (auto &ci : co.conInf) con << ci.first.first << "\t" << (ci.first.second ? "FORWARD" : "REV") << "\t" << ci.second << "\n";
            co.cmbLen = cmbLen;
        });
    }
    if (stats)
        for (auto &co : chains) {
            stats->nChains++;
            stats->nFastaBases += co.cmbLen;
        }
    lap("emit chains");
    if (stats) {
        stats->nContigs = ctgSet.size();
        std::vector<std::uint64_t> hs(results.size(), 0), nb(results.size(), 0);
        std::atomic<std::size_t> nextRes{0};
        auto sumWorker = [&]() {
            for (std::size_t i; (i = nextRes.fetch_add(1)) < results.size();) {
                std::uint64_t h = 1469598103934665603ull ^ i;
                for (auto &n : results[i]) {
                    DualPos pp = graph.position(n.first);
                    h = (h ^ ((static_cast<std::uint64_t>(pp.first) << 32) | pp.second)) * 1099511628211ull;
                    h = (h ^ graph.nodeCode[n.first.node]) * 1099511628211ull;
                    h = (h ^ static_cast<std::uint64_t>(n.second)) * 1099511628211ull;
                }
                hs[i] = h;
                nb[i] = SeqTools::seqSize(results[i]);
            }
        };
        {
            unsigned nThreads = hostThreads ? hostThreads : std::max(1u, usableCpus());
            nThreads = static_cast<unsigned>(std::min<std::size_t>(nThreads, std::max<std::size_t>(1, results.size())));
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(sumWorker);
            sumWorker();
            for (auto &t : pool) t.join();
        }
        for (std::size_t i = 0; i < results.size(); ++i) {
            if (!results[i].empty()) stats->pathChecksum += hs[i];
            stats->nPathNodes += results[i].size();
            stats->nPathBases += nb[i];
        }
    }
    lap("stats");
    // the storage goes back to the caller's cache (the contents are spent)
    for (std::size_t i = 0; i < results.size() && i < travelled.size(); ++i)
        if (results[i].capacity() > travelled[i].capacity()) results[i].swap(travelled[i]);
    return success;
}

}  // namespace pagh
