#include "seq_db.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "line_index.hpp"

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
#if defined(__x86_64__)
// 32 bases -> 8 packed bytes per turn (CompressedSeq's packing: 4 bases per byte, base i at bits 2 * (i & 3); C/c = 1, G/g = 2,
// T/t = 3, everything else 0).  Returns how many bases it packed (a multiple of 32).
__attribute__((target("avx2"))) std::size_t packBasesAvx2(const char *sq, std::size_t n, std::uint8_t *out) {
    const __m256i lower = _mm256_set1_epi8(0x20), cC = _mm256_set1_epi8('c'), cG = _mm256_set1_epi8('g'), cT = _mm256_set1_epi8('t');
    const __m256i one = _mm256_set1_epi8(1), two = _mm256_set1_epi8(2), three = _mm256_set1_epi8(3);
    const __m256i w8 = _mm256_set1_epi16(0x0401);       // bytes (c0, c1) -> c0 + 4 c1
    const __m256i w16 = _mm256_set1_epi32(0x00100001);  // words (x0, x1) -> x0 + 16 x1
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    std::size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_or_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(sq + i)), lower);
        const __m256i code = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(_mm256_cmpeq_epi8(v, cC), one), _mm256_and_si256(_mm256_cmpeq_epi8(v, cG), two)),
                                             _mm256_and_si256(_mm256_cmpeq_epi8(v, cT), three));
        const __m256i b = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(code, w8), w16), gather);
        const std::uint32_t lo = static_cast<std::uint32_t>(_mm256_extract_epi32(b, 0)), hi = static_cast<std::uint32_t>(_mm256_extract_epi32(b, 4));
        std::memcpy(out + (i >> 2), &lo, 4);
        std::memcpy(out + (i >> 2) + 4, &hi, 4);
    }
    return i;
}
#endif
// a sequence's bases packed to out (zeroed, (n + 3) / 4 bytes)
void packBases(const char *sq, std::size_t n, std::uint8_t *out) {
    std::size_t i = 0;
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2");
    if (wide) i = packBasesAvx2(sq, n, out);
#endif
    for (; i + 4 <= n; i += 4)
        out[i >> 2] = static_cast<std::uint8_t>(encodeBase(sq[i]) | (encodeBase(sq[i + 1]) << 2) | (encodeBase(sq[i + 2]) << 4) | (encodeBase(sq[i + 3]) << 6));
    for (; i < n; ++i) out[i >> 2] |= static_cast<std::uint8_t>(encodeBase(sq[i]) << ((i & 3) * 2));
}
}  // namespace

// test hook (tests/test_host_simd.py): a sequence packed by the production path or by the scalar loop alone (out zeroed, (n + 3) / 4 bytes)
extern "C" void pagh_debug_pack_bases(const char *sq, std::uint64_t n, std::uint8_t *out, int scalar_only) {
    if (!scalar_only) return packBases(sq, n, out);
    for (std::uint64_t i = 0; i < n; ++i) out[i >> 2] |= static_cast<std::uint8_t>(encodeBase(sq[i]) << ((i & 3) * 2));
}

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

SeqDb::SeqDb(const std::string &path, const PackWindow *window) {
    std::ifstream probe(path);
    if (!probe) {  // the reference's AutoSeqDatabase yields an empty database for a file it cannot open (SeqHelper::
        finish();  // autoLoadFromFile finds no record type, AutoSeqDatabase.cpp:9-22) and pagraph carries on
        return;
    }
    std::string first;
    bool fasta = false;
    if (std::getline(probe, first) && !first.empty()) fasta = first.front() == '>' || first.front() == ';';
    probe.close();

    std::ifstream in(path);
    std::string line;
    const bool sequential = std::getenv("PAGH_SEQUENTIAL_LOADERS") != nullptr;  // (tests: the plain loops below)
    if (fasta && !sequential && loadFastaParallel(path)) {
        // (done)
    } else if (fasta) {
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
    } else if (sequential || !loadFastqParallel(path, window)) {  // (the plain loop packs every record: a window only ever leaves work out)
        // 4-line FASTQ records; a trailing partial record is dropped (SeqHelper.cpp:13-26)
        std::string l1, l2, l3, l4;
        while (std::getline(in, l1) && std::getline(in, l2) && std::getline(in, l3) && std::getline(in, l4)) {
            add(l1, l2);
        }
    }
    finish();
}

// The same records as the sequential loop above, with the lines found and the reads packed by a pool of threads.
// Returns false (nothing loaded) when a header line has no token: the reference then re-uses the previous
// record's name (AutoSeqDatabase.cpp:12), a cross-record dependency that is left to the sequential path.
bool SeqDb::loadFastqParallel(const std::string &path, const PackWindow *window) {
    FileLines fl;
    if (!fl.load(path)) return false;
    const std::size_t nRec = fl.size() / 4;
    if (nRec == 0) return true;
    // token of every header line
    std::vector<std::pair<std::uint32_t, std::uint32_t>> tok(nRec);  // (offset in line, length)
    std::atomic<bool> plain{true};
    parallelFor(nRec, 4096, [&](std::size_t r) {
        const char *p = fl.data(4 * r);
        const std::size_t n = fl.length(4 * r);
        std::size_t a = 0;
        while (a < n && isSpaceC(p[a])) ++a;
        std::size_t b = a;
        while (b < n && !isSpaceC(p[b])) ++b;
        if (a == b) plain = false;
        tok[r] = {static_cast<std::uint32_t>(a), static_cast<std::uint32_t>(b - a)};
        if (fl.length(4 * r + 1) > 0xFFFFFFFFull) plain = false;
    });
    if (!plain) return false;
    const std::size_t first = names_.size();
    names_.resize(first + nRec);
    len_.resize(first + nRec);
    byteOff_.resize(first + nRec);
    std::size_t cursor = packed_.size();
    const bool windowed = window && window->world > 1 && first == 0;
    std::size_t shared = 0;  // (bytes of the zero stretch every record outside the window points at)
    if (windowed) {
        for (std::size_t r = 0; r < nRec; ++r)
            if (!window->wants(r, nRec)) shared = std::max(shared, ((fl.length(4 * r + 1) + 3) / 4 + 3) & ~std::size_t(3));
        cursor += shared;
    }
    const std::size_t sharedAt = cursor - shared;
    for (std::size_t r = 0; r < nRec; ++r) {
        const std::size_t n = fl.length(4 * r + 1);
        len_[first + r] = static_cast<std::uint32_t>(n);
        totalBases_ += n;
        if (windowed && !window->wants(r, nRec)) {
            byteOff_[first + r] = sharedAt;
            continue;
        }
        byteOff_[first + r] = cursor;
        cursor += (((n + 3) / 4) + 3) & ~std::size_t(3);
    }
    packed_.resize(cursor, 0);
    parallelFor(nRec, 256, [&](std::size_t r) {
        names_[first + r].assign(fl.data(4 * r) + tok[r].first + 1, tok[r].second - 1);
        if (windowed && !window->wants(r, nRec)) return;
        const char *sq = fl.data(4 * r + 1);
        const std::size_t n = len_[first + r];
        packBases(sq, n, packed_.data() + byteOff_[first + r]);
    });
    for (std::size_t r = 0; r < nRec; ++r) nameToId_[names_[first + r]] = first + r;  // later duplicates win
    lastName_.assign(fl.data(4 * (nRec - 1)) + tok[nRec - 1].first, tok[nRec - 1].second);
    return true;
}

// The records of the sequential FASTA loop above — only '>' starts a record, its lines are concatenated, whatever precedes the
// first header is glued to the first record — with the lines found, copied and packed by a pool of threads (a 50 Mb reference
// is ONE record: the work is split by base ranges, not by records).  Returns false (nothing loaded) for what is left to
// the sequential path: a record of 2^32 bases or more.
bool SeqDb::loadFastaParallel(const std::string &path) {
    FileLines fl;
    if (!fl.load(path)) return false;
    const std::size_t nLines = fl.size();
    std::vector<std::size_t> headers;  // line numbers that start a record
    {
        std::vector<std::uint8_t> isHeader(nLines, 0);
        parallelFor(nLines, 1 << 16, [&](std::size_t i) { isHeader[i] = fl.length(i) > 0 && fl.data(i)[0] == '>'; });
        for (std::size_t i = 0; i < nLines; ++i)
            if (isHeader[i]) headers.push_back(i);
    }
    if (headers.empty()) return true;  // no record at all
    const std::size_t nRec = headers.size();
    // sequence lines of record r: [headers[r] + 1, headers[r + 1]); the first record also owns the lines before its header
    std::vector<std::size_t> lineOff(nLines + 1, 0);  // bases before line i (header lines count as empty)
    for (std::size_t i = 0; i < nLines; ++i) lineOff[i + 1] = lineOff[i] + ((fl.length(i) > 0 && fl.data(i)[0] == '>') ? 0 : fl.length(i));
    auto recBegin = [&](std::size_t r) { return r == 0 ? lineOff[0] : lineOff[headers[r]]; };
    auto recEnd = [&](std::size_t r) { return r + 1 < nRec ? lineOff[headers[r + 1]] : lineOff[nLines]; };
    for (std::size_t r = 0; r < nRec; ++r)
        if (recEnd(r) - recBegin(r) > 0xFFFFFFFFull) return false;
    // all bases in one buffer, in file order (every line knows where it goes)
    std::vector<char> flat(lineOff[nLines] + 4, 'A');
    parallelFor(nLines, 1 << 12, [&](std::size_t i) {
        const std::size_t n = lineOff[i + 1] - lineOff[i];
        if (n) std::memcpy(flat.data() + lineOff[i], fl.data(i), n);
    });
    const std::size_t first = names_.size();
    for (std::size_t r = 0; r < nRec; ++r) {
        std::stringstream ss;  // (the name as add() takes it: first token of the header line, minus its first character)
        ss << fl.str(headers[r]);
        ss >> lastName_;
        const std::string name = lastName_.empty() ? std::string() : lastName_.substr(1);
        const std::size_t n = recEnd(r) - recBegin(r);
        nameToId_[name] = names_.size();  // later duplicates win
        names_.push_back(name);
        len_.push_back(static_cast<std::uint32_t>(n));
        byteOff_.push_back(packed_.size());
        totalBases_ += n;
        const std::size_t nBytes = (n + 3) / 4;
        packed_.resize(packed_.size() + ((nBytes + 3) & ~std::size_t(3)), 0);
    }
    // pack: chunks of 2^16 output bytes of every record
    struct Job {
        std::size_t rec, byteLo, byteHi;
    };
    std::vector<Job> jobs;
    for (std::size_t r = 0; r < nRec; ++r) {
        const std::size_t nBytes = (static_cast<std::size_t>(len_[first + r]) + 3) / 4;
        for (std::size_t b = 0; b < nBytes; b += 1 << 16) jobs.push_back(Job{r, b, std::min(nBytes, b + (1 << 16))});
    }
    parallelFor(jobs.size(), 1, [&](std::size_t j) {
        const Job &jb = jobs[j];
        const char *sq = flat.data() + recBegin(jb.rec);
        const std::size_t n = len_[first + jb.rec];
        std::uint8_t *out = packed_.data() + byteOff_[first + jb.rec];
        for (std::size_t b = jb.byteLo; b < jb.byteHi; ++b) {
            const std::size_t i = 4 * b;
            unsigned v = 0;
            for (std::size_t q = 0; q < 4 && i + q < n; ++q) v |= encodeBase(sq[i + q]) << (2 * q);
            out[b] = static_cast<std::uint8_t>(v);
        }
    });
    return true;
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
