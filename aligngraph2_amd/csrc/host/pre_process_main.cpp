// pre_process — drop-in for the reference program of the same name (PAGraph/src/main/pre_process.cpp:24-321, SURVEY
// §8f.3): assigns every contig to the reference sequence(s) it covers best, splits the read->contig / read->reference
// alignment files and the reads into one block per reference, and writes the config.txt that pagraph iterates.
//   pre_process -r reads.fastq -c ctg.fasta -x read_to_ctg.ref -y read_to_ref.ref -z ctg_to_ref.ref -o out_dir
//               [-k top_k] [-m min_ratio] [--test]
// Outputs (byte-identical to the reference): <out>/<b>.ctg.ref, <b>.ref.ref, <b>.new.fastq, config.txt.
//
// What is kept on purpose, because it is observable:
//  * block numbers follow the ITERATION ORDER of the reference's std::unordered_map<std::string, ...> of references
//    (pre_process.cpp:73-110,122-133,293-305) and contig->coverage follows that of its ctgMap (:62-85); the same
//    containers are filled here in the same order, so the same libstdc++ yields the same order;
//  * `ctgMap[align.queryName]` (:40) INSERTS unknown names with index 0: alignments of an unknown contig are booked on
//    the first contig's coverage, and the name then takes part in the assignment loop like a contig;
//  * the coverage vector of a (contig, reference, strand) is sized by the first alignment seen (:42-45);
//  * the unstable std::sort of (count, target) by count only (:77-80);
//  * read ids are std::stoll of the query name (:140,155): a non-numeric name terminates the program.
// What differs: the alignment files are indexed once and parsed / written by a pool of threads with large buffers
// instead of line-by-line streams flushed with std::endl (the reference's cost on a 1 Gbase data set is minutes).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "line_index.hpp"

namespace {

using pagh::FileLines;
using pagh::isSpaceC;
using pagh::parallelFor;

struct Header {  // AlignmentHelper::loadFromRefFile's SimpleAlign (AlignmentHelper.cpp:24-36); empty names if malformed
    std::string queryName, refName;
    bool forward = false;
    std::size_t queryBegin = 0, queryEnd = 0, querySize = 0, refBegin = 0, refEnd = 0, refSize = 0;
};

// one ALN header line: the strict fast path (ten fields, the last six plain decimal numbers) or the stream extraction
Header parseHeader(const char *p, std::size_t n) {
    Header h;
    std::pair<std::size_t, std::size_t> tk[10];
    std::size_t nt = 0, i = 0;
    while (nt < 10) {
        while (i < n && isSpaceC(p[i])) ++i;
        if (i >= n) break;
        const std::size_t a = i;
        while (i < n && !isSpaceC(p[i])) ++i;
        tk[nt++] = {a, i - a};
    }
    bool fastOk = nt == 10;
    std::size_t num[6] = {0, 0, 0, 0, 0, 0};
    for (int f = 0; f < 6 && fastOk; ++f) {
        const char *q = p + tk[4 + f].first;
        const std::size_t len = tk[4 + f].second;
        if (len == 0 || len > 18) fastOk = false;
        std::size_t v = 0;
        for (std::size_t c = 0; c < len && fastOk; ++c) {
            if (q[c] < '0' || q[c] > '9') fastOk = false;
            v = v * 10 + static_cast<std::size_t>(q[c] - '0');
        }
        num[f] = v;
    }
    if (fastOk) {
        h.queryName.assign(p + tk[0].first, tk[0].second);
        h.refName.assign(p + tk[1].first, tk[1].second);
        h.forward = tk[2].second == 1 && p[tk[2].first] == 'F';
        h.queryBegin = num[0];
        h.queryEnd = num[1];
        h.querySize = num[2];
        h.refBegin = num[3];
        h.refEnd = num[4];
        h.refSize = num[5];
        return h;
    }
    std::stringstream ss;
    ss.str(std::string(p, n));
    std::string queryName, refName, forward, score;
    std::size_t queryBegin = 0, queryEnd = 0, querySize = 0, refBegin = 0, refEnd = 0, refSize = 0;
    ss >> queryName >> refName >> forward >> score >> queryBegin >> queryEnd >> querySize >> refBegin >> refEnd >> refSize;
    if (!ss.fail()) {
        h.queryName = queryName;
        h.refName = refName;
        h.forward = forward == "F";
        h.queryBegin = queryBegin;
        h.queryEnd = queryEnd;
        h.querySize = querySize;
        h.refBegin = refBegin;
        h.refEnd = refEnd;
        h.refSize = refSize;
    }
    return h;
}

using RefWithCtg = std::unordered_map<std::string, std::set<std::pair<std::string, bool>>>;

// parseCtgToRef (pre_process.cpp:24-113)
RefWithCtg assignContigs(const std::string &ctgPath, const std::string &ctgToRefPath, std::size_t topK, double minRatio) {
    std::vector<std::map<std::pair<std::string, bool>, std::vector<bool>>> ctgCover;
    std::unordered_map<std::string, std::size_t> ctgMap;

    {   // contig headers in file order: key = header line minus its first character (the WHOLE rest of the line)
        FileLines fl;
        if (fl.load(ctgPath)) {
            auto addCtg = [&](const std::string &headerLine) {
                ctgMap[headerLine.substr(1)] = ctgCover.size();  // throws on an empty header line, like the reference
                ctgCover.emplace_back();
            };
            // SeqHelper::testFileType (SeqHelper.cpp:76-99): FASTA iff the first line starts with '>' or ';', everything
            // else (also an empty first line) is read as FASTQ
            const bool fasta = fl.size() && fl.length(0) && (fl.data(0)[0] == '>' || fl.data(0)[0] == ';');
            if (fasta) {  // only '>' starts a record (SeqHelper.cpp:33-57)
                for (std::size_t i = 0; i < fl.size(); ++i)
                    if (fl.length(i) && fl.data(i)[0] == '>') addCtg(fl.str(i));
            } else {      // complete 4-line records (SeqHelper.cpp:8-31)
                for (std::size_t r = 0; r < fl.size() / 4; ++r) addCtg(fl.str(4 * r));
            }
        }
    }
    {   // coverage of every contig by (reference, strand)
        FileLines fl;
        if (fl.load(ctgToRefPath)) {
            for (std::size_t r = 0; r < fl.size() / 3; ++r) {
                const Header al = parseHeader(fl.data(3 * r), fl.length(3 * r));
                const std::size_t idx = ctgMap[al.queryName];  // inserts 0 for an unknown name (see the file header)
                if (idx >= ctgCover.size()) continue;          // no contig at all: the reference indexes an empty vector here
                auto &cover = ctgCover[idx];
                auto it = cover.find({al.refName, al.forward});
                if (it == cover.end()) it = cover.emplace(std::make_pair(al.refName, al.forward), std::vector<bool>(al.querySize, false)).first;
                auto &arr = it->second;
                // inside the vector the reference fills [begin, end); past it its behaviour is undefined — clamped here
                const std::size_t b = std::min(al.queryBegin, arr.size()), e = std::min(std::max(al.queryEnd, b), arr.size());
                std::fill(arr.begin() + static_cast<std::ptrdiff_t>(b), arr.begin() + static_cast<std::ptrdiff_t>(e), true);
            }
        }
    }
    RefWithCtg refWithCtg;
    for (auto &ctg : ctgMap) {
        auto &cov = ctgCover[ctg.second];
        std::vector<std::pair<std::size_t, std::pair<std::string, bool>>> covToRef;
        for (auto &refCov : cov) {
            const auto &record = refCov.second;
            const std::size_t total = record.size();
            const std::size_t cnt = std::accumulate(record.begin(), record.end(), static_cast<std::size_t>(0));
            if (cnt * 1.0 / total >= minRatio) covToRef.emplace_back(cnt, refCov.first);
        }
        std::sort(covToRef.begin(), covToRef.end(),
                  [](const std::pair<std::size_t, std::pair<std::string, bool>> &l, const std::pair<std::size_t, std::pair<std::string, bool>> &r) {
                      return l.first > r.first;
                  });
        for (std::size_t i = 0; i < topK && i < covToRef.size(); ++i) refWithCtg[covToRef[i].second.first].insert({ctg.first, covToRef[i].second.second});
    }
    RefWithCtg filtered;  // references with at least two contigs; one orientation per contig (:97-110)
    for (auto &withCtg : refWithCtg) {
        if (withCtg.second.size() > 1) {
            std::unordered_set<std::string> seen;
            for (auto &ctg : withCtg.second)
                if (seen.count(ctg.first) == 0) {
                    filtered[withCtg.first].insert(ctg);
                    seen.insert(ctg.first);
                }
        }
    }
    return filtered;
}

// appends the chosen records (whole line triples) of an alignment file to the block files, one thread per block
void splitAlignments(const std::string &path, const std::unordered_map<std::string, std::vector<std::size_t>> &targetBlocks,
                     std::size_t nBlocks, const std::string &outDir, const char *suffix, std::vector<std::vector<std::size_t>> &readIds) {
    FileLines fl;
    const bool have = fl.load(path);
    const std::size_t nRec = have ? fl.size() / 3 : 0;
    // per record: the blocks of its target (none for an unknown target or a malformed header) and its read id
    std::vector<const std::vector<std::size_t> *> blocksOf(nRec, nullptr);
    std::vector<std::size_t> idOf(nRec, 0);
    std::atomic<bool> badId{false};
    parallelFor(nRec, 256, [&](std::size_t r) {
        const Header al = parseHeader(fl.data(3 * r), fl.length(3 * r));
        auto it = targetBlocks.find(al.refName);
        if (it == targetBlocks.end() || it->second.empty()) return;
        try {
            idOf[r] = static_cast<std::size_t>(std::stoll(al.queryName));
            blocksOf[r] = &it->second;
        } catch (...) {
            badId = true;
        }
    });
    if (badId) throw std::invalid_argument("stoll");  // the reference dies on the first non-numeric read name
    std::vector<std::vector<std::size_t>> recsOf(nBlocks);
    for (std::size_t r = 0; r < nRec; ++r)
        if (blocksOf[r])
            for (std::size_t b : *blocksOf[r]) recsOf[b].push_back(r);
    parallelFor(nBlocks, 1, [&](std::size_t b) {
        std::ofstream of(outDir + "/" + std::to_string(b) + suffix, std::ios::binary);
        std::string buf;
        buf.reserve(1 << 22);
        for (std::size_t r : recsOf[b]) {
            for (int l = 0; l < 3; ++l) {
                buf.append(fl.data(3 * r + l), fl.length(3 * r + l));
                buf.push_back('\n');
            }
            readIds[b].push_back(idOf[r]);
            if (buf.size() >= (1u << 22) - (1u << 16)) {
                of.write(buf.data(), static_cast<std::streamsize>(buf.size()));
                buf.clear();
            }
        }
        of.write(buf.data(), static_cast<std::streamsize>(buf.size()));
    });
}

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HelpRequested {};

void usage(std::ostream &os) {
    os << "  pre_process {OPTIONS}\n\n  OPTIONS:\n\n"
          "      -h, --help                        display this help menu\n"
          "      -r[path], --read=[path]           read path\n"
          "      -c[path], --contig=[path]         contig path\n"
          "      -x[path], --read_to_ctg=[path]    alignment path of read to contig\n"
          "      -y[path], --read_to_ref=[path]    alignment path of read to reference\n"
          "      -z[path], --ctg_to_ref=[path]     alignment path of contig to reference\n"
          "      -o[path], --output=[path]         output directory\n"
          "      -k[unsigned], --top_k=[unsigned]  top K of alignment\n"
          "      -m[double], --min=[double]        alignment min threshold\n"
          "      --test                            only do one step\n";
}

template <typename T>
T parseNumber(const std::string &flag, const std::string &v) {
    std::istringstream ss(v);
    T x{};
    ss >> x;
    if (ss.fail() || ss.rdbuf()->in_avail() != 0) throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}

struct Options {
    std::string read, contig, readToCtg, readToRef, ctgToRef, out;
    std::size_t topK = 1;
    double minRatio = 0.15;
    bool test = false;
};

Options parseCli(int argc, char **argv) {
    Options o;
    auto assign = [&](const std::string &name, const std::string &value) {
        if (name == "r" || name == "read") o.read = value;
        else if (name == "c" || name == "contig") o.contig = value;
        else if (name == "x" || name == "read_to_ctg") o.readToCtg = value;
        else if (name == "y" || name == "read_to_ref") o.readToRef = value;
        else if (name == "z" || name == "ctg_to_ref") o.ctgToRef = value;
        else if (name == "o" || name == "output") o.out = value;
        else if (name == "k" || name == "top_k") o.topK = parseNumber<std::size_t>(name, value);
        else if (name == "m" || name == "min") o.minRatio = parseNumber<double>(name, value);
        else throw ParseError("Flag could not be matched: " + name);
    };
    auto isLong = [](const std::string &n) {
        static const char *names[] = {"read", "contig", "read_to_ctg", "read_to_ref", "ctg_to_ref", "output", "top_k", "min"};
        for (auto *x : names)
            if (n == x) return true;
        return false;
    };
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") throw HelpRequested{};
        if (a == "--test") {
            o.test = true;
            continue;
        }
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
            if (std::strchr("rcxyzokm", a[1]) == nullptr) throw ParseError("Flag could not be matched: '" + name + "'");
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

int main(int argc, char **argv) {
    if (argc <= 1) {
        usage(std::cerr);
        return 0;
    }
    Options o;
    try {
        o = parseCli(argc, argv);
    } catch (const HelpRequested &) {
        usage(std::cerr);
        return 0;
    } catch (const ParseError &e) {
        std::cerr << e.what() << std::endl;
        usage(std::cerr);
        return 1;
    }
    const RefWithCtg refWithCtg = assignContigs(o.contig, o.ctgToRef, o.topK, o.minRatio);
    for (auto &withCtg : refWithCtg) {
        std::cerr << withCtg.first << std::endl;
        for (auto &ctg : withCtg.second) {
            std::cerr << "\t" << ctg.first << std::endl;
            std::cerr << "\t" << ctg.second << std::endl;
        }
    }
    if (o.test) return 0;

    // filterReadAndRef (pre_process.cpp:115-196): block b = b-th reference in the iteration order of the map
    std::unordered_map<std::string, std::vector<std::size_t>> ctgBlocks, refBlocks;
    std::size_t nBlocks = 0;
    for (auto &withCtg : refWithCtg) {
        refBlocks[withCtg.first].push_back(nBlocks);
        for (auto &ctg : withCtg.second) ctgBlocks[ctg.first].push_back(nBlocks);
        ++nBlocks;
    }
    for (std::size_t b = 0; b < nBlocks; ++b) {  // every block file exists even when it stays empty
        std::ofstream(o.out + "/" + std::to_string(b) + ".ctg.ref");
        std::ofstream(o.out + "/" + std::to_string(b) + ".ref.ref");
        std::ofstream(o.out + "/" + std::to_string(b) + ".new.fastq");
    }
    std::vector<std::vector<std::size_t>> idsCtg(nBlocks), idsRef(nBlocks);
    splitAlignments(o.readToCtg, ctgBlocks, nBlocks, o.out, ".ctg.ref", idsCtg);
    splitAlignments(o.readToRef, refBlocks, nBlocks, o.out, ".ref.ref", idsRef);

    // reads of every block: ids are 1-based record numbers of the FASTQ file (:171-188)
    FileLines reads;
    const std::size_t nReads = reads.load(o.read) ? reads.size() / 4 : 0;
    std::vector<std::size_t> distinct(nBlocks, 0);
    parallelFor(nBlocks, 1, [&](std::size_t b) {
        std::vector<bool> want(nReads + 1, false);
        std::set<std::size_t> all;  // the reference counts distinct ids, also those beyond the read file
        for (auto *ids : {&idsCtg[b], &idsRef[b]})
            for (std::size_t id : *ids) {
                all.insert(id);
                if (id >= 1 && id <= nReads) want[id] = true;
            }
        distinct[b] = all.size();
        std::ofstream of(o.out + "/" + std::to_string(b) + ".new.fastq", std::ios::binary);
        std::string buf;
        buf.reserve(1 << 22);
        for (std::size_t id = 1; id <= nReads; ++id) {
            if (!want[id]) continue;
            buf.push_back('@');
            buf.append(std::to_string(id));
            buf.push_back('\n');
            for (int l = 1; l < 4; ++l) {
                buf.append(reads.data(4 * (id - 1) + l), reads.length(4 * (id - 1) + l));
                buf.push_back('\n');
            }
            if (buf.size() >= (1u << 22) - (1u << 17)) {
                of.write(buf.data(), static_cast<std::streamsize>(buf.size()));
                buf.clear();
            }
        }
        of.write(buf.data(), static_cast<std::streamsize>(buf.size()));
    });
    for (std::size_t b = 0; b < nBlocks; ++b) std::cerr << distinct[b] << std::endl;
    std::cerr << "Total read=" << nReads << std::endl;
    std::cerr << "Read iterator Done" << std::endl;
    std::cerr << "Filter Done" << std::endl;

    std::ofstream of(o.out + "/config.txt");
    std::size_t cnt = 0;
    for (auto &withCtg : refWithCtg) {
        of << withCtg.first << "\n" << cnt << ".new.fastq\n" << cnt << ".ctg.ref\n" << cnt << ".ref.ref\n";
        for (auto &ctg : withCtg.second) of << ctg.first << "\n" << ctg.second << "\n";
        of << "\n";
        ++cnt;
    }
    of.close();
    std::cerr << "All Done" << std::endl;
    return 0;
}
