#include "host_threads.hpp"
#include "pagraph_driver.hpp"

#include <sstream>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <iostream>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <exception>
#include <unordered_set>

#include "aln_db.hpp"
#include "assembly.hpp"
#include "config.hpp"
#include "raw_input.hpp"
#include "kmer_file.hpp"
#include "path_graph.hpp"
#include "seq_db.hpp"

namespace pagh {

namespace {

struct Options {
    unsigned threads = 16;
    std::string kmer, read, contig, ref, pre, aln, out;
    std::size_t minLen = 50, epsilon = 10, cov = 1;
};

void usage(std::ostream &os) {
    os << "  pagraph {OPTIONS}\n\n  OPTIONS:\n\n"
          "      -h, --help                        display this help menu\n"
          "      -t[thread_num], --thread=[thread_num]   number of thread\n"
          "      -k[path], --kmer=[path]           solid kmer set path\n"
          "      -r[path], --read=[path]           read path\n"
          "      -c[path], --contig=[path]         contig path\n"
          "      -R[path], --ref=[path]            reference path\n"
          "      -p[path], --pre_process=[path]    pre process directory\n"
          "      -a[path], --aln=[path]            alignment path of contig to reference\n"
          "      -o[path], --output=[path]         output directory\n"
          "      -l[len], --length=[len]           minimum path length\n"
          "      --epsilon=[dist]                  distance to join vertices\n"
          "      -v[cov]                           coverage to filter\n";
}

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HelpRequested {};

template <typename T>
T parseNumber(const std::string &flag, const std::string &v) {
    // args.hxx reads values with an istream extraction and rejects leftovers
    std::istringstream ss(v);
    T x{};
    ss >> x;
    if (ss.fail() || ss.rdbuf()->in_avail() != 0)
        throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}

// Same surface as the args.hxx parser the reference uses (pagraph.cpp:71-99): "-t 16", "-t16",
// "--thread 16", "--thread=16"; repeated flags are allowed and the last one wins (AlignGraph2.py passes
// -r twice); unknown flags and stray positionals are errors; -h / --help prints the usage.
Options parseCli(int argc, char **argv) {
    Options o;
    auto assign = [&](const std::string &name, const std::string &value) {
        if (name == "t" || name == "thread") o.threads = parseNumber<unsigned>(name, value);
        else if (name == "k" || name == "kmer") o.kmer = value;
        else if (name == "r" || name == "read") o.read = value;
        else if (name == "c" || name == "contig") o.contig = value;
        else if (name == "R" || name == "ref") o.ref = value;
        else if (name == "p" || name == "pre_process") o.pre = value;
        else if (name == "a" || name == "aln") o.aln = value;
        else if (name == "o" || name == "output") o.out = value;
        else if (name == "l" || name == "length") o.minLen = parseNumber<std::size_t>(name, value);
        else if (name == "epsilon") o.epsilon = parseNumber<std::size_t>(name, value);
        else if (name == "v") o.cov = parseNumber<std::size_t>(name, value);
        else throw ParseError("Flag could not be matched: " + name);
    };
    auto isLong = [](const std::string &n) {
        static const char *names[] = {"thread", "kmer", "read", "contig", "ref", "pre_process", "aln", "output", "length", "epsilon"};
        for (auto *x : names) if (n == x) return true;
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
            if (std::strchr("tkrcRpaolv", a[1]) == nullptr) throw ParseError("Flag could not be matched: '" + name + "'");
            std::string value;
            if (a.size() > 2) {
                value = a.substr(2);
            } else {
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

namespace {
double nowSec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int runPagraph(int argc, char **argv, GraphBackend &backend) {
    const bool timing = pagh::envTiming();
    double tPrev = nowSec();
    const double tStart = tPrev;
    auto lap = [&](const char *what) {
        double t = nowSec();
        if (timing) std::cerr << "[timing] " << what << " " << (t - tPrev) << " s" << std::endl;
        tPrev = t;
    };
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
        BuildParams params;
        params.threads = opt.threads;
        params.epsilon = opt.epsilon;
        params.covFilter = opt.cov;
        const double errorRate = 0.15, startSplit = 0.90;

        std::cout << "Loading config.txt" << std::endl;
        auto configs = loadConfig(opt.pre + "/config.txt");
        // Multi-GPU runs of ONE pagraph invocation (aligngraph2_amd/parallel.py: run_config_blocks): the config blocks are
        // independent (pagraph.cpp:181-182 resets the graph between them), so every GPU's process gets a list of block
        // numbers in PAGRAPH_BLOCKS ("0,3,5"), processes only those — under their ORIGINAL numbers, which are the output
        // file prefixes — and writes its share of contig.txt to contig.txt.part<PAGRAPH_PART>; the launcher merges the parts.
        std::set<std::size_t> onlyBlocks;
        const char *blocksEnv = std::getenv("PAGRAPH_BLOCKS");
        if (blocksEnv) {
            std::stringstream ss(blocksEnv);
            std::string tok;
            while (std::getline(ss, tok, ','))
                if (!tok.empty()) onlyBlocks.insert(static_cast<std::size_t>(std::stoull(tok)));
        }
        // the block's three text files (pagraph.cpp:186-199 reads them one after the other; they are independent): parsed side by
        // side, each by its own pool of threads
        struct BlockFiles {
            SeqDb reads;
            AlnDb readToCtg, readToRef;
            BlockFiles(const std::string &pre, const BlockConfig &cfg, unsigned shardRank, unsigned shardWorld, unsigned threads) {
                if (shardWorld > 1) {
                    // one of several ranks that build this block together: the columns of an alignment are parsed only for the
                    // reads of THIS rank's stretch of the emission order (pag_shard_extract: positions [n r / N, n (r + 1) / N) of the
                    // thread-major strided order, MultiThreadTools.tcc:8-14) — of the others the header is all that is used
                    // (and its bases are packed, and staged on the device, for that stretch only)
                    const SeqDb::PackWindow window{shardRank, shardWorld, threads ? threads : 1u};
                    reads = SeqDb(pre + "/" + cfg.readPath, &window);
                    const std::uint64_t n = reads.size(), T = threads ? threads : 1;
                    const std::uint64_t lo = n * shardRank / shardWorld, hi = n * (shardRank + 1ull) / shardWorld;
                    const SeqDb *rd = &reads;
                    const AlnRecordFilter mine = [rd, n, T, lo, hi](const char *name, std::size_t len) {
                        const std::string nm(name, len);
                        if (!rd->contains(nm)) return false;
                        const std::uint64_t i = rd->id(nm), t = i % T;
                        // reads before i in the order: those of the threads below t (thread t' handles ceil((n - t') / T) reads), then i / T
                        const std::uint64_t full = n / T, rem = n % T;
                        const std::uint64_t pos = t * full + (t < rem ? t : rem) + i / T;
                        return pos >= lo && pos < hi;
                    };
                    // (each database gets its own copy of the filter; `reads`, which it refers to, is a member and outlives both parses
                    // whichever way they end)
                    auto ctgAln = std::async(std::launch::async, [&] { return AlnDb(pre + "/" + cfg.ctgAlnPath, AlnDb::Flavor::Mecat, mine); });
                    auto refAln = std::async(std::launch::async, [&] { return AlnDb(pre + "/" + cfg.refAlnPath, AlnDb::Flavor::Mecat, mine); });
                    readToCtg = ctgAln.get();
                    readToRef = refAln.get();
                    return;
                }
                auto ctgAln = std::async(std::launch::async, [&] { return AlnDb(pre + "/" + cfg.ctgAlnPath, AlnDb::Flavor::Mecat); });
                auto refAln = std::async(std::launch::async, [&] { return AlnDb(pre + "/" + cfg.refAlnPath, AlnDb::Flavor::Mecat); });
                reads = SeqDb(pre + "/" + cfg.readPath);
                readToCtg = ctgAln.get();
                readToRef = refAln.get();
            }
        };
        const bool prefetch = envInt("PAGRAPH_PREFETCH", 1) != 0;
        auto mine = [&](std::size_t no) { return !blocksEnv || onlyBlocks.count(no) != 0; };
        auto loadBlock = [&opt, &configs, &backend](std::size_t no) {
            return std::make_unique<BlockFiles>(opt.pre, configs[no], backend.shardRank(), backend.shardWorld(), static_cast<unsigned>(opt.threads));
        };
        std::future<std::unique_ptr<BlockFiles>> ahead;
        std::size_t aheadNo = static_cast<std::size_t>(-1);
        // (the text of the first block this process works on is parsed beside the global inputs below: 0.3 s of a 2 s run)
        if (prefetch)
            for (std::size_t no = 0; no < configs.size(); ++no)
                if (mine(no)) {
                    aheadNo = no;
                    ahead = std::async(std::launch::async, loadBlock, no);
                    break;
                }
        std::cout << "Loading KMer" << std::endl;
        KmerFile kmers(opt.kmer);
        std::cout << "Done! k=" << kmers.k() << std::endl;
        lap("  solid k-mer file");
        // (the three files are independent: parsed side by side, reported in the reference's order; a file that fails throws
        // where the reference would have failed on it)
        auto refsAhead = std::async(std::launch::async, [&opt] { return SeqDb(opt.ref); });
        auto alnAhead = std::async(std::launch::async, [&opt] { return AlnDb(opt.aln, AlnDb::Flavor::MummerV2); });
        std::cout << "Loading Contigs" << std::endl;
        SeqDb contigs(opt.contig);
        std::cout << "Done! contigs number=" << contigs.size() << std::endl;
        std::cout << "Loading References" << std::endl;
        SeqDb refs = refsAhead.get();
        std::cout << "Done! reference number=" << refs.size() << std::endl;
        std::cout << "Loading ContigToRef" << std::endl;
        AlnDb ctgToRef = alnAhead.get();
        std::cout << "Done! number=" << ctgToRef.size() << std::endl;
        lap("  contigs, references, contig->reference alignments");
        std::cout << "Building original pa Graph [" << backend.name() << "]" << std::endl;
        backend.create(kmers.data(), kmers.size(), static_cast<unsigned>(kmers.k()));
        std::cout << "Done! kmer number=" << backend.solidCount() << std::endl;
        lap("  solid set -> device (create)");
        if (timing) std::cerr << "[timing] load global inputs + create " << (nowSec() - tStart) << " s" << std::endl;

        std::unordered_set<std::string> okCtg;  // same container as the reference: contig.txt order (quirk Q11)
        std::size_t blockNo = 0;
        HostGraph graphs[2];  // (storage reused from block to block; two sets: a block's host half runs beside the next block)
        std::vector<TravelSequence> precomputed[2];
        GraphBackend::TravelViews tviews[2];
        const bool halves = backend.travelsInHalves() && envInt("PAGRAPH_OVERLAP", 1) != 0;
        std::size_t nHalves = 0;
        // the host half of the block before: joined before the next walks (and before anything is thrown past it)
        struct HostHalf {
            std::thread th;
            std::exception_ptr err;
            std::set<std::pair<std::string, bool>> success;
            std::ostringstream log;
            void start(std::function<std::set<std::pair<std::string, bool>>()> work) {
                th = std::thread([this, work]() {
                    try {
                        success = work();
                    } catch (...) {
                        err = std::current_exception();
                    }
                });
            }
            void join(std::unordered_set<std::string> &ok) {
                if (!th.joinable()) return;
                th.join();
                std::cout << log.str() << std::flush;
                log.str(std::string());
                if (err) {
                    std::exception_ptr e = err;
                    err = nullptr;
                    std::rethrow_exception(e);
                }
                for (auto &s : success) ok.emplace(s.first);
                success.clear();
            }
            ~HostHalf() {
                if (th.joinable()) th.join();
            }
        } half;
        // The text files of the NEXT block this process will handle are parsed while the current block is on the device
        // (the parsers' threads are idle then; PAGRAPH_PREFETCH=0 parses every block when its turn comes, as the
        // reference does, and holds one block's inputs in host memory instead of two).
        for (auto &cfg : configs) {
            if (!mine(blockNo)) {
                ++blockNo;
                continue;
            }
            backend.reset();
            // (the device memory of the walks is asked for while the block's text files are parsed: the first large
            // allocation of a process can take seconds)
            std::uint64_t ctgBases = 0;
            for (auto &c : cfg.contigs)
                if (contigs.contains(c.first)) ctgBases += contigs.length(contigs.id(c.first));
            std::exception_ptr reserveError;
            std::thread reserver([&] {
                try {
                    backend.reserveForContigs(ctgBases);
                } catch (...) {
                    reserveError = std::current_exception();
                }
            });
            struct Joiner {
                std::thread &t;
                ~Joiner() {
                    if (t.joinable()) t.join();
                }
            } joiner{reserver};
            std::cout << "Use Ref: " << cfg.ref << std::endl;
            std::unique_ptr<BlockFiles> files = (ahead.valid() && aheadNo == blockNo) ? ahead.get() : loadBlock(blockNo);
            const SeqDb &reads = files->reads;
            const AlnDb &readToCtg = files->readToCtg, &readToRef = files->readToRef;
            std::cout << "Done! reads number=" << reads.size() << std::endl;
            std::cout << "Done! aln number=" << readToCtg.size() << std::endl;
            std::cout << "Done! aln number=" << readToRef.size() << std::endl;

            lap("load block inputs");
            reserver.join();
            if (reserveError) std::rethrow_exception(reserveError);
            lap("device memory of the walks reserved (wait)");
            if (prefetch) {
                std::size_t nextNo = blockNo + 1;
                while (nextNo < configs.size() && !mine(nextNo)) ++nextNo;
                if (nextNo < configs.size()) {
                    aheadNo = nextNo;
                    ahead = std::async(std::launch::async, loadBlock, nextNo);
                }
            }
            std::cout << "Pre Process" << std::endl;
            RawInput raw(reads, contigs, refs, readToCtg, readToRef, ctgToRef, cfg, params);
            pag_build_input input{};
            backend.prepare(raw, input);
            const PositionMapper ctgMapper(contigs), refMapper(refs);
            std::set<std::pair<std::string, bool>> usedCtg;
            for (auto &c : cfg.contigs) usedCtg.emplace(c);

            lap("prepare (lists, filters, contig->reference map)");
            std::cout << "[PositionProcessor] Running read to contig..." << std::endl;
            pag_build_stats st{};
            backend.process(input, st);
            std::cout << "\n\tmerge edge = " << st.merge_edge[0] << "\n\ttotal pos = " << st.total_pos[0]
                      << "\n\tmerge pos = " << st.merge_pos[0] << "\n\n\n\tmerge edge = " << st.merge_edge[1]
                      << "\n\ttotal pos = " << st.total_pos[1] << "\n\tmerge pos = " << st.merge_pos[1] << std::endl;
            std::cout << "[PositionProcessor] Done!" << std::endl;

            lap("graph build (process)");
            pag_travel_params tp{};
            tp.ref_threads = opt.threads;
            tp.deviation = opt.epsilon * 2;
            tp.error_rate = errorRate;
            tp.start_split = startSplit;
            tp.min_len = opt.minLen;
            const std::string prefix = std::to_string(blockNo) + "_";
            if (!halves) {
                GraphBackend::TravelContext ctx{contigs, refs, ctgMapper, refMapper, usedCtg, static_cast<unsigned>(kmers.k())};
                backend.travel(ctx, tp, graphs[0], precomputed[0]);
                lap("traversal");
                if (backend.shardRank() != 0) {  // (rank 0 has the travel sequences of all ranks and writes the block's outputs)
                    ++blockNo;
                    continue;
                }
                auto successCtg = assemble(opt.out, prefix, graphs[0], contigs, refs, ctgMapper, refMapper, usedCtg, opt.epsilon * 2, errorRate,
                                           startSplit, opt.minLen, opt.threads, 0, nullptr, false, precomputed[0]);
                lap("traverse + write");
                ++blockNo;
                for (auto &s : successCtg) okCtg.emplace(s.first);
                continue;
            }
            // The block's host half — path graph, chain selection, the output files — runs on a thread of its own beside the
            // next block's device work (the blocks are independent; pagraph.cpp:181-182 resets the graph between them).
            {
                GraphBackend::TravelContext ctx{contigs, refs, ctgMapper, refMapper, usedCtg, static_cast<unsigned>(kmers.k())};
                backend.travelPrepare(ctx, tp);  // (successor records: device work, the previous block's host half still runs)
                lap("successor records");
                half.join(okCtg);  // ... which reads travel sequences in memory the next walks reuse
                lap("wait for the previous block's host half");
                const std::size_t slot = nHalves++ & 1u;
                backend.travelWalks(ctx, tp, tviews[slot]);
                lap("walks");
                // ONE block by several ranks: every rank writes the path dumps of the contigs it walked (most of a block's
                // output bytes), rank 0 — which has the travel sequences of all ranks — selects the chains and writes the rest
                const bool rankDumps = backend.shardWorld() > 1;
                if (backend.shardRank() != 0) {
                    if (rankDumps) {
                        std::set<std::pair<std::string, bool>> own;
                        for (auto &c : usedCtg)
                            if (contigs.contains(c.first) && backend.walksContig(contigs.id(c.first))) own.emplace(c);
                        const PositionMapper cm(contigs), rm(refs);
                        buildPathGraph(tviews[slot].views, static_cast<unsigned>(kmers.k()), graphs[slot], precomputed[slot]);
                        AssembleShare share;
                        share.dumpsOnly = true;
                        assemble(opt.out, prefix, graphs[slot], contigs, refs, cm, rm, own, opt.epsilon * 2, errorRate, startSplit, opt.minLen, opt.threads, 0,
                                 nullptr, true, precomputed[slot], nullptr, &share);
                        lap("path dumps of this rank's contigs");
                    }
                    ++blockNo;
                    continue;
                }
                // (beside another block's device work the pool stays small: a burst of threads uses up a container's CPU
                // quota and stalls the device's control threads, traverse_api.cpp)
                std::size_t nextNo = blockNo + 1;
                while (nextNo < configs.size() && !mine(nextNo)) ++nextNo;
                unsigned poolThreads = 0;
                if (nextNo < configs.size())
                    poolThreads = pagh::envOverlapThreads() ? pagh::envOverlapThreads() : 20u;  // (traverse_api.cpp: the measurement)
                // (which contigs this rank walked: a copy — the backend's plan is the next block's by the time the thread reads it)
                std::vector<char> walkedHere;
                if (rankDumps) {
                    walkedHere.assign(contigs.size(), 0);
                    for (std::size_t id = 0; id < contigs.size(); ++id) walkedHere[id] = backend.walksContig(id) ? 1 : 0;
                }
                half.start([&, slot, prefix, usedCtg, poolThreads, tp, rankDumps, walkedHere]() {
                    const PositionMapper cm(contigs), rm(refs);
                    buildPathGraph(tviews[slot].views, static_cast<unsigned>(kmers.k()), graphs[slot], precomputed[slot]);
                    AssembleShare share;
                    if (rankDumps) share.writesDump = [&walkedHere](std::size_t id) { return id < walkedHere.size() && walkedHere[id] != 0; };
                    return assemble(opt.out, prefix, graphs[slot], contigs, refs, cm, rm, usedCtg, opt.epsilon * 2, errorRate, startSplit, opt.minLen,
                                    opt.threads, poolThreads, nullptr, false, precomputed[slot], &half.log, rankDumps ? &share : nullptr);
                });
                ++blockNo;
            }
        }
        half.join(okCtg);
        lap("last block's host half");
        if (backend.shardRank() != 0) return EXIT_SUCCESS;
        const char *part = std::getenv("PAGRAPH_PART");
        std::ofstream ctgList(opt.out + (blocksEnv ? std::string("/contig.txt.part") + (part ? part : "0") : std::string("/contig.txt")));
        for (auto &c : okCtg) ctgList << c << "\n";
        return EXIT_SUCCESS;
    } catch (const std::exception &e) {
        std::cerr << "pagraph: " << e.what() << std::endl;
        return 1;
    }
}

}  // namespace pagh
