// paf2aln — contig->reference alignments in PAF (paftools.js delta2paf / minimap2, with a cg:Z: CIGAR in the 14th column)
// to the 3-line ALN text `pagraph -a` reads.  Native counterpart of the pipeline's helper script/paf2aln.py:19-95
// (AlignGraph2.py:353-355 calls paf2aln(ctg_path, ref_path, paf_path, out_path, thread_num)); SURVEY.md §8f.2.
//
//   paf2aln <contigs.fasta> <reference.fasta> <in.paf> <out.aln> [threads = 16]
//
// PARITY UNPINNED: the script imports Biopython (Bio.SeqIO), which this image does not have, so it cannot be run here to
// make golden vectors.  What follows restates its text; tests/test_paf2aln.py checks it against a second restatement
// written from the same text, on seeded inputs.  What the script does, quirks included:
//   * FASTA records are keyed by the first word of the header line; a sequence is its lines with trailing white space,
//     blanks and carriage returns removed (Bio.SeqIO.FastaIO); a repeated key is an error (SeqIO.to_dict);
//   * per PAF line: header  name, ref name, F|R (column 5 == '+'), "NULL", contig start, end, length, reference start,
//     end, length — the numbers as the text has them;
//   * the CIGAR is column 14 minus its first five characters ("cg:Z:") and minus its LAST character (the line's newline:
//     a CIGAR that is not the last column, or a last line without a newline, loses a real character; an operation letter
//     lost that way is never applied);
//   * M: one contig base and one reference base; D: '-' over a reference base; I: a contig base over '-'; any other
//     letter consumes its count and does nothing.  On the reverse strand the contig is read from end - 1 downwards through
//     rv(): A->T, C->G, G->T (sic, paf2aln.py:12-13), T->A, anything else -> N, case folded; forward bases are copied as
//     the file has them.  Indices follow Python: a negative one counts from the end, one past the end is an error;
//   * the script's `thread_num` threads take the lines round robin, each into its own part file, and the part files are
//     concatenated in thread order: the output ORDER depends on the thread count (line i goes before line j when
//     i % threads < j % threads, or they are equal and i < j).  This program writes that order directly; it does not
//     leave the part files (`<out>_<i>`) behind.
// A line the script would die on (fewer than 14 columns, an unknown sequence name, an index out of range) ends this
// program with an error instead of silently truncating one thread's share of the output.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

std::string rstrip(const std::string &s) {
    size_t n = s.size();
    while (n > 0 && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\n' || s[n - 1] == '\r' || s[n - 1] == '\f' || s[n - 1] == '\v')) --n;
    return s.substr(0, n);
}

// Bio.SeqIO.parse(path, "fasta") -> to_dict: id = first word of the title, sequence without blanks / carriage returns
std::unordered_map<std::string, std::string> readFasta(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::unordered_map<std::string, std::string> out;
    std::string line, id, seq;
    bool have = false;
    auto flush = [&]() {
        if (!have) return;
        if (!out.emplace(id, seq).second) throw std::runtime_error("Duplicate key '" + id + "' in " + path);
        seq.clear();
    };
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '>') {
            flush();
            have = true;
            const std::string title = rstrip(line.substr(1));
            size_t b = 0;
            while (b < title.size() && std::isspace(static_cast<unsigned char>(title[b]))) ++b;
            size_t e = b;
            while (e < title.size() && !std::isspace(static_cast<unsigned char>(title[e]))) ++e;
            id = title.substr(b, e - b);
            continue;
        }
        if (!have) continue;  // (text before the first record is skipped)
        for (char c : rstrip(line))
            if (c != ' ' && c != '\r') seq.push_back(c);
    }
    flush();
    return out;
}

char rv(char ch) {
    switch (std::toupper(static_cast<unsigned char>(ch))) {
        case 'A': return 'T';
        case 'C': return 'G';
        case 'G': return 'T';  // (as the script has it)
        case 'T': return 'A';
        default: return 'N';
    }
}

char at(const std::string &s, long long idx, const char *what) {  // Python indexing
    const long long n = static_cast<long long>(s.size());
    if (idx < 0) idx += n;
    if (idx < 0 || idx >= n) throw std::runtime_error(std::string(what) + " index out of range");
    return s[static_cast<size_t>(idx)];
}

std::vector<std::string> splitTabs(const std::string &line) {
    std::vector<std::string> f;
    size_t b = 0;
    for (;;) {
        const size_t e = line.find('\t', b);
        if (e == std::string::npos) {
            f.push_back(line.substr(b));
            return f;
        }
        f.push_back(line.substr(b, e - b));
        b = e + 1;
    }
}

long long pyInt(const std::string &s, const char *what) {
    const std::string t = rstrip(s);
    size_t b = 0;
    while (b < t.size() && std::isspace(static_cast<unsigned char>(t[b]))) ++b;
    size_t p = b;
    if (p < t.size() && (t[p] == '+' || t[p] == '-')) ++p;
    if (p == t.size()) throw std::runtime_error(std::string("not a number: ") + what);
    for (size_t x = p; x < t.size(); ++x)
        if (!std::isdigit(static_cast<unsigned char>(t[x]))) throw std::runtime_error(std::string("not a number: ") + what);
    return std::stoll(t.substr(b));
}

std::string convert(const std::string &line, const std::unordered_map<std::string, std::string> &ctgs,
                    const std::unordered_map<std::string, std::string> &refs) {
    const std::vector<std::string> sp = splitTabs(line);
    if (sp.size() < 14) throw std::runtime_error("a PAF line with fewer than 14 columns");
    const std::string &ctgName = sp[0], &refName = sp[5];
    const bool fwd = sp[4] == "+";
    std::string out = ctgName + "\t" + refName + "\t" + (fwd ? "F" : "R") + "\tNULL\t" + sp[2] + "\t" + sp[3] + "\t" + sp[1] + "\t" + sp[7] + "\t" + sp[8] +
                      "\t" + sp[6] + "\n";
    auto qi = ctgs.find(ctgName);
    auto ri = refs.find(refName);
    if (qi == ctgs.end()) throw std::runtime_error("unknown contig " + ctgName);
    if (ri == refs.end()) throw std::runtime_error("unknown reference " + refName);
    const std::string &query = qi->second, &ref = ri->second;
    long long cb = fwd ? pyInt(sp[2], "contig start") : pyInt(sp[3], "contig end") - 1;
    const long long cs = fwd ? 1 : -1;
    long long rb = pyInt(sp[7], "reference start");
    const std::string &cg = sp[13];
    // cigar_str[5:-1]
    const size_t from = std::min<size_t>(5, cg.size());
    const size_t to = cg.empty() ? 0 : cg.size() - 1;
    std::string s1, s2, nb;
    for (size_t x = from; x < to; ++x) {
        const char ch = cg[x];
        if (ch >= '0' && ch <= '9') {
            nb.push_back(ch);
            continue;
        }
        if (nb.empty()) throw std::runtime_error("a CIGAR operation without a count");
        const long long gap = std::stoll(nb);
        nb.clear();
        for (long long j = 0; j < gap; ++j) {
            if (ch == 'M') {
                const char q = at(query, cb, "contig");
                s1.push_back(fwd ? q : rv(q));
                cb += cs;
                s2.push_back(at(ref, rb, "reference"));
                rb += 1;
            } else if (ch == 'D') {
                s1.push_back('-');
                s2.push_back(at(ref, rb, "reference"));
                rb += 1;
            } else if (ch == 'I') {
                const char q = at(query, cb, "contig");
                s1.push_back(fwd ? q : rv(q));
                cb += cs;
                s2.push_back('-');
            } else {
                break;  // (no branch of the script's loop matches: the count is spent on nothing)
            }
        }
    }
    out += s1;
    out += "\n";
    out += s2;
    out += "\n";
    return out;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 5) {
        std::fprintf(stderr, "usage: paf2aln <contigs.fasta> <reference.fasta> <in.paf> <out.aln> [threads = 16]\n");
        return 1;
    }
    try {
        const long long tn = argc > 5 ? std::atoll(argv[5]) : 16;
        if (tn <= 0) throw std::runtime_error("threads must be positive");
        const size_t T = static_cast<size_t>(tn);
        const auto ctgs = readFasta(argv[1]);
        const auto refs = readFasta(argv[2]);
        std::vector<std::string> lines;
        {
            std::ifstream in(argv[3], std::ios::binary);
            if (!in) throw std::runtime_error(std::string("cannot open ") + argv[3]);
            std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            // readlines(): every line keeps its newline; a last line without one is a line too
            size_t b = 0;
            while (b < all.size()) {
                const size_t e = all.find('\n', b);
                if (e == std::string::npos) {
                    lines.push_back(all.substr(b));
                    break;
                }
                lines.push_back(all.substr(b, e - b + 1));
                b = e + 1;
            }
        }
        std::vector<std::string> recs(lines.size());
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        std::string error;
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < lines.size() && !failed.load();) {
                try {
                    recs[i] = convert(lines[i], ctgs, refs);
                } catch (const std::exception &e) {
                    if (!failed.exchange(true)) error = "line " + std::to_string(i + 1) + ": " + e.what();
                }
            }
        };
        const unsigned nThreads = std::max(1u, std::min<unsigned>(static_cast<unsigned>(std::min<size_t>(T, 64)), std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nThreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (failed.load()) throw std::runtime_error(error);
        std::ofstream out(argv[4], std::ios::binary);
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[4]);
        for (size_t start = 0; start < T; ++start)  // the script's part files, in thread order
            for (size_t i = start; i < lines.size(); i += T) out << recs[i];
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "paf2aln: %s\n", e.what());
        return 1;
    }
}
