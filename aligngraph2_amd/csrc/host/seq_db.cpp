#include "seq_db.hpp"

#include <fstream>
#include <sstream>
#include <stdexcept>

namespace pagh {

namespace {
inline unsigned encodeBase(char c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;  // A, a and everything else
    }
}
}  // namespace

void SeqDb::add(const std::string &comment, const std::string &seq) {
    // name: first whitespace-separated token of the header line, minus its leading '>' / '@'.
    // An empty/blank header leaves the previous token in place, as the reference's stream does.
    std::stringstream ss;
    ss << comment;
    ss >> lastName_;
    std::string name = lastName_.empty() ? std::string() : lastName_.substr(1);
    if (seq.size() > 0xFFFFFFFFull) throw std::runtime_error("sequence longer than 2^32 bases: " + name);

    nameToId_[name] = names_.size();  // later duplicates win
    names_.push_back(name);
    len_.push_back(static_cast<std::uint32_t>(seq.size()));
    byteOff_.push_back(packed_.size());
    totalBases_ += seq.size();

    std::size_t nBytes = (seq.size() + 3) / 4;
    std::size_t base = packed_.size();
    packed_.resize(base + ((nBytes + 3) & ~std::size_t(3)), 0);
    for (std::size_t i = 0; i < seq.size(); ++i) {
        packed_[base + (i >> 2)] |= static_cast<std::uint8_t>(encodeBase(seq[i]) << ((i & 3) * 2));
    }
}

void SeqDb::addPacked(const std::string &name, const std::uint8_t *packed, std::uint32_t len) {
    nameToId_[name] = names_.size();
    names_.push_back(name);
    len_.push_back(len);
    byteOff_.push_back(packed_.size());
    totalBases_ += len;
    std::size_t nBytes = (static_cast<std::size_t>(len) + 3) / 4;
    std::size_t base = packed_.size();
    packed_.resize(base + ((nBytes + 3) & ~std::size_t(3)), 0);
    for (std::size_t i = 0; i < nBytes; ++i) packed_[base + i] = packed[i];
    if (len & 3u) packed_[base + nBytes - 1] &= static_cast<std::uint8_t>((1u << ((len & 3u) * 2)) - 1u);
}

void SeqDb::finish() { packed_.resize(packed_.size() + 32, 0); }

SeqDb::SeqDb(const std::string &path) {
    std::ifstream probe(path);
    if (!probe) throw std::runtime_error("cannot open sequence file: " + path);
    std::string first;
    bool fasta = false;
    if (std::getline(probe, first) && !first.empty()) fasta = first.front() == '>' || first.front() == ';';
    probe.close();

    std::ifstream in(path);
    std::string line;
    if (fasta) {
        // multi-line records; only '>' starts a record; text before the first header is glued to
        // the first record (the reference never clears its buffer there, SeqHelper.cpp:39-49)
        std::string header, buffer;
        while (std::getline(in, line)) {
            if (!line.empty() && line[0] == '>') {
                if (!header.empty()) {
                    add(header, buffer);
                    buffer.clear();
                }
                header = line;
            } else {
                buffer += line;
            }
        }
        if (!header.empty()) add(header, buffer);
    } else {
        // 4-line FASTQ records; a trailing partial record is dropped (SeqHelper.cpp:13-26)
        std::string l1, l2, l3, l4;
        while (std::getline(in, l1) && std::getline(in, l2) && std::getline(in, l3) && std::getline(in, l4)) {
            add(l1, l2);
        }
    }
    finish();
}

std::string SeqDb::toString(std::size_t id, bool forward) const {
    const char *table = forward ? "ACGT" : "TGCA";
    std::size_t n = len_[id];
    std::string s(n, 'A');
    const std::uint8_t *p = packed_.data() + byteOff_[id];
    for (std::size_t i = 0; i < n; ++i) {
        unsigned code = (p[i >> 2] >> ((i & 3) * 2)) & 3u;
        s[forward ? i : n - 1 - i] = table[code];
    }
    return s;
}

char SeqDb::baseAt(std::size_t id, std::size_t idx, bool forward) const {
    std::size_t n = len_[id];
    if (idx >= n) return 'N';
    std::size_t src = forward ? idx : n - 1 - idx;
    unsigned code = (packed_[byteOff_[id] + (src >> 2)] >> ((src & 3) * 2)) & 3u;
    return (forward ? "ACGT" : "TGCA")[code];
}

}  // namespace pagh
