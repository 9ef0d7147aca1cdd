// kmer_counter — drop-in for the reference program of the same name (PAGraph/src/main/kmer_counter.cpp): same
// flags, same output file (k as u64, then the solid k-mer codes as u64 in the order the reference's worker threads
// emit them), the counting done by the HIP library (pag_kmer_count).  There is no CPU fallback.
//   kmer_counter -t <threads> -i <reads.fastq|fasta> -o <out> -k <k> -m <threshold> [-p parts] [-s batch]
// -p / -s only shape the reference's host tables and are accepted and ignored; -t decides the order of the codes in
// the file exactly as there (thread t emits the codes c with c % t_num == t, ascending; kmer_counter.cpp:79-95,
// MultiThreadTools.tcc:6-21).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "host_threads.hpp"
#include "pagraph_hip.h"
#include "seq_db.hpp"

namespace {

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct HelpRequested {};

void usage(std::ostream &os) {
    os << "  kmer_counter {OPTIONS}\n\n  OPTIONS:\n\n"
          "      -h, --help                        display this help menu\n"
          "      -t[thread_num], --thread=[thread_num]   number of thread\n"
          "      -i[path], --in=[path]             input of read\n"
          "      -o[path], --out=[path]            output of file\n"
          "      -k[k], --kmer=[k]                 kmer size\n"
          "      -m[threshold], --min=[threshold]  threshold for min abundance\n"
          "      -p[size], --part=[size]           number of hash table\n"
          "      -s[size], --size=[size]           size of each batch\n";
}

template <typename T>
T parseNumber(const std::string &flag, const std::string &v) {
    std::istringstream ss(v);
    T x{};
    ss >> x;
    if (ss.fail() || ss.rdbuf()->in_avail() != 0)
        throw ParseError("Argument '" + flag + "' received invalid value type '" + v + "'");
    return x;
}

struct Options {
    unsigned threads = 16;
    std::string in, out;
    std::size_t k = 14, parts = 4, batch = 10240;
    double threshold = 0.2;
};

Options parseCli(int argc, char **argv) {
    Options o;
    auto assign = [&](const std::string &name, const std::string &value) {
        if (name == "t" || name == "thread") o.threads = parseNumber<unsigned>(name, value);
        else if (name == "i" || name == "in") o.in = value;
        else if (name == "o" || name == "out") o.out = value;
        else if (name == "k" || name == "kmer") o.k = parseNumber<std::size_t>(name, value);
        else if (name == "m" || name == "min") o.threshold = parseNumber<double>(name, value);
        else if (name == "p" || name == "part") o.parts = parseNumber<std::size_t>(name, value);
        else if (name == "s" || name == "size") o.batch = parseNumber<std::size_t>(name, value);
        else throw ParseError("Flag could not be matched: " + name);
    };
    auto isLong = [](const std::string &n) {
        static const char *names[] = {"thread", "in", "out", "kmer", "min", "part", "size"};
        for (auto *x : names)
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
            if (std::strchr("tiokmps", a[1]) == nullptr) throw ParseError("Flag could not be matched: '" + name + "'");
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
    if (o.k < 1 || o.k > 15) {
        std::cerr << "kmer_counter: k must be 1..15 (the reference's table size overflows at 16)" << std::endl;
        return 1;
    }
    if (!pag_device_available()) {
        std::cerr << "kmer_counter: no gfx950 device (there is no CPU fallback): " << pag_last_error() << std::endl;
        return 1;
    }
    try {
        const char *devEnv = std::getenv("PAGRAPH_DEVICE");
        const int device = devEnv ? std::atoi(devEnv) : 0;
        // a read file that cannot be opened is an empty input for the reference (SeqHelper::testFileType -> "unknown",
        // nothing is loaded): every abundance is 0 and, with the default threshold, every k-mer is "solid"
        pagh::SeqDb reads;
        {
            std::ifstream probe(o.in);
            if (probe) reads = pagh::SeqDb(o.in);
            else reads.finish();
        }
        pag_seqs rs{reads.size(), reads.byteOff().data(), reads.lens().data(), reads.packed().data(), reads.packed().size()};
        const std::uint64_t nCodes = 1ull << (2 * o.k);
        std::vector<std::uint32_t> bitmap((nCodes + 31) / 32 + 1, 0);
        pag_kmer_count_result res{};
        int rc = pag_kmer_count(&rs, 0, static_cast<std::uint32_t>(o.k), o.threshold, device, bitmap.data(), 0, &res);
        if (rc != PAG_OK) {
            std::cerr << "kmer_counter: " << pag_last_error() << std::endl;
            return 1;
        }
        // the reference's writer (kmer_counter.cpp:79-95): k as size_t, then per worker thread its codes
        std::ofstream of(o.out, std::ios::binary);
        const std::uint64_t k64 = o.k;
        of.write(reinterpret_cast<const char *>(&k64), sizeof k64);
        // -t 0: the reference's multiTraversal starts no worker, its availKmer(0) is empty, and the file is the 8-byte
        // header alone (kmer_counter.cpp:79-95)
        const unsigned T = o.threads;
        std::vector<std::uint64_t> buf;
        buf.reserve(1 << 16);
        for (unsigned t = 0; t < T; ++t) {
            for (std::uint64_t c = t; c < nCodes; c += T) {
                if ((bitmap[c >> 5] >> (c & 31)) & 1u) {
                    buf.push_back(c);
                    if (buf.size() == (1u << 16)) {
                        of.write(reinterpret_cast<const char *>(buf.data()), static_cast<std::streamsize>(buf.size() * 8));
                        buf.clear();
                    }
                }
            }
        }
        of.write(reinterpret_cast<const char *>(buf.data()), static_cast<std::streamsize>(buf.size() * 8));
        if (pagh::envTiming())
            std::fprintf(stderr, "[timing] kmer_counter: %llu reads, min abundance %llu, %llu solid k-mers, count %.2f ms, select %.2f ms\n",
                         (unsigned long long)reads.size(), (unsigned long long)res.min_abundance, (unsigned long long)res.n_solid,
                         res.ms_count, res.ms_select);
    } catch (const std::exception &e) {
        std::cerr << "kmer_counter: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
