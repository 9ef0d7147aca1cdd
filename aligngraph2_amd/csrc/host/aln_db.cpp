#include "host_threads.hpp"
#include "aln_db.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <unordered_map>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include "line_index.hpp"

namespace pagh {

namespace {
// parseDiff (ParseAlignTools.cpp:8-26) over one record: q=='-' -> class 1; r=='-' -> 2; mismatch -> 3; match -> 0, two bits per
// column, 16 columns per word (words zeroed by the caller); a reference row shorter than the query row reads as NUL.
void classifyScalar(const char *q, std::size_t n, const char *r, std::size_t rn, std::uint32_t *w, std::size_t from, std::uint32_t &nEmit, std::uint32_t &nRadv) {
    for (std::size_t i = from; i < n; ++i) {
        const char qc = q[i];
        const char rc = i < rn ? r[i] : '\0';
        unsigned cls;
        if (qc == '-') cls = 1;
        else if (rc == '-') cls = 2;
        else if (qc != rc) cls = 3;
        else cls = 0;
        w[i >> 4] |= cls << ((i & 15) * 2);
        nEmit += cls != 1;
        nRadv += cls != 2;
    }
}
#if defined(__x86_64__)
// 32 columns per turn: byte compares, the two class bits as movemasks, interleaved with pdep (the text of BASELINE configs[1]
// is 2.2 G columns per block: the one-character-at-a-time loop was most of the time a block's files take to load)
__attribute__((target("avx2,bmi2,popcnt"))) std::size_t classifyAvx2(const char *q, const char *r, std::size_t n32, std::uint32_t *w, std::uint32_t &nEmit,
                                                                   std::uint32_t &nRadv) {
    const __m256i dash = _mm256_set1_epi8('-');
    std::uint32_t e = 0, a = 0;
    for (std::size_t i = 0; i < n32; i += 32) {
        const __m256i qv = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(q + i));
        const __m256i rv = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(r + i));
        const std::uint32_t mq = static_cast<std::uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(qv, dash)));
        const std::uint32_t mr = static_cast<std::uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(rv, dash))) & ~mq;
        const std::uint32_t ne = ~static_cast<std::uint32_t>(_mm256_movemask_epi8(_mm256_cmpeq_epi8(qv, rv))) & ~mq & ~mr;
        const std::uint32_t b0 = mq | ne, b1 = mr | ne;  // class bits 0 and 1 of the 32 columns
        w[(i >> 4)] = static_cast<std::uint32_t>(_pdep_u32(b0 & 0xFFFFu, 0x55555555u) | _pdep_u32(b1 & 0xFFFFu, 0xAAAAAAAAu));
        w[(i >> 4) + 1] = static_cast<std::uint32_t>(_pdep_u32(b0 >> 16, 0x55555555u) | _pdep_u32(b1 >> 16, 0xAAAAAAAAu));
        e += 32u - static_cast<std::uint32_t>(_mm_popcnt_u32(mq));
        a += 32u - static_cast<std::uint32_t>(_mm_popcnt_u32(mr));
    }
    nEmit += e;
    nRadv += a;
    return n32;
}
#endif
void classifyColumns(const char *q, std::size_t n, const char *r, std::size_t rn, std::uint32_t *w, std::uint32_t &nEmit, std::uint32_t &nRadv) {
    std::size_t done = 0;
#if defined(__x86_64__)
    static const bool wide = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt");
    if (wide) done = classifyAvx2(q, r, std::min(n, rn) & ~static_cast<std::size_t>(31), w, nEmit, nRadv);
#endif
    classifyScalar(q, n, r, rn, w, done, nEmit, nRadv);
}
}  // namespace

// test hook (tests/test_host_simd.py): the column classes of one record, by the production path or by the scalar loop alone
extern "C" void pagh_debug_classify_columns(const char *q, std::uint64_t n, const char *r, std::uint64_t rn, std::uint32_t *words, std::uint32_t *n_emit,
                                            std::uint32_t *n_radv, int scalar_only) {
    std::uint32_t e = 0, a = 0;
    if (scalar_only) classifyScalar(q, n, r, rn, words, 0, e, a);
    else classifyColumns(q, n, r, rn, words, e, a);
    *n_emit = e;
    *n_radv = a;
}

void AlnDb::addRecord(AlnRecord rec, const std::string &qline, const std::string &rline) {
    // parseDiff: q=='-' -> (1,0); r=='-' -> (0,1); mismatch -> (1,1); match -> (0,0).  One column
    // per character of the query line; a shorter reference line reads as NUL (mismatch).
    std::size_t n = qline.size();
    if (n > 0xFFFFFFFFull) throw std::runtime_error("alignment longer than 2^32 columns");
    if (filter_ && *filter_ && !(*filter_)(rec.queryName.data(), rec.queryName.size())) {  // (header only, see AlnRecordFilter)
        rec.diffOff = diff_.size();
        recs_.push_back(std::move(rec));
        return;
    }
    rec.diffOff = diff_.size();
    rec.nCols = static_cast<std::uint32_t>(n);
    diff_.resize(diff_.size() + (n + 15) / 16, 0);
    std::uint32_t *w = diff_.data() + rec.diffOff;
    std::uint32_t nEmit = 0, nRadv = 0;
    classifyColumns(qline.data(), n, rline.data(), rline.size(), w, nEmit, nRadv);
    rec.nEmit = nEmit;
    rec.nRadv = nRadv;
    recs_.push_back(std::move(rec));
}

void AlnDb::sortByScore() {
    // std::sort(_alignments) with AlignInf::operator< = "score greater" (AlignInf.cpp:31-33).
    // The sort is unstable; to get the reference's exact permutation on ties we run the same
    // libstdc++ std::sort with the same comparator on a proxy array in the same initial order.
    struct Proxy {
        std::size_t score;
        std::size_t idx;
    };
    std::vector<Proxy> proxy(recs_.size());
    for (std::size_t i = 0; i < recs_.size(); ++i) proxy[i] = {recs_[i].score, i};
    std::sort(proxy.begin(), proxy.end(), [](const Proxy &a, const Proxy &b) { return a.score > b.score; });
    std::vector<AlnRecord> sorted;
    sorted.reserve(recs_.size());
    for (auto &p : proxy) sorted.push_back(std::move(recs_[p.idx]));
    recs_.swap(sorted);
}

// The Mecat-flavoured loop below, record by record, on a pool of threads: records are independent there (three
// lines each, no state carried from one header to the next), so the lines are indexed once, the header of record r
// is parsed by the strict fast path (ten whitespace-separated fields, the last six plain decimal numbers) or, when it
// does not fit that, by the very stream extraction of the sequential loop, and the column classes are written
// to their final place (offsets = running sum of the word counts in file order).
namespace {
double nowS() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
bool AlnDb::loadMecatParallel(const std::string &path, const AlnRecordFilter &filter) {
    const bool timing = pagh::envTiming();
    const double t0 = nowS();
    FileLines fl;
    if (!fl.load(path)) return false;
    const double t1 = nowS();
    const std::size_t nRec = fl.size() / 3;
    recs_.assign(nRec, AlnRecord{});
    // headers first (a filter decides by the query name which records get their columns)
    parallelFor(nRec, 256, [&](std::size_t r) {
        AlnRecord &rec = recs_[r];
        const char *p = fl.data(3 * r);
        const std::size_t n = fl.length(3 * r);
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
            rec.queryName.assign(p + tk[0].first, tk[0].second);
            rec.refName.assign(p + tk[1].first, tk[1].second);
            rec.forward = tk[2].second == 1 && p[tk[2].first] == 'F';
            rec.score = static_cast<std::size_t>(std::atoll(std::string(p + tk[3].first, tk[3].second).c_str()));
            rec.queryBegin = num[0];
            rec.queryEnd = num[1];
            rec.refBegin = num[3];
            rec.refEnd = num[4];
        } else {  // anything unusual: the stream extraction itself
            std::stringstream ss;
            ss.str(std::string(p, n));
            std::string queryName, refName, forward, score;
            std::size_t queryBegin = 0, queryEnd = 0, querySize = 0, refBegin = 0, refEnd = 0, refSize = 0;
            ss >> queryName >> refName >> forward >> score >> queryBegin >> queryEnd >> querySize >> refBegin >> refEnd >> refSize;
            if (!ss.fail()) {
                rec.queryName = queryName;
                rec.refName = refName;
                rec.forward = forward == "F";
                rec.score = static_cast<std::size_t>(std::atoll(score.c_str()));
                rec.queryBegin = queryBegin;
                rec.queryEnd = queryEnd;
                rec.refBegin = refBegin;
                rec.refEnd = refEnd;
            }
        }
    });
    std::vector<std::uint8_t> wanted(nRec, 1);
    std::size_t nWanted = nRec;
    if (filter) {
        parallelFor(nRec, 1024, [&](std::size_t r) { wanted[r] = filter(recs_[r].queryName.data(), recs_[r].queryName.size()) ? 1 : 0; });
        nWanted = 0;
        for (std::size_t r = 0; r < nRec; ++r) nWanted += wanted[r];
    }
    std::vector<std::uint64_t> off(nRec + 1, 0);
    for (std::size_t r = 0; r < nRec; ++r) {
        const std::size_t n = fl.length(3 * r + 1);
        if (n > 0xFFFFFFFFull) throw std::runtime_error("alignment longer than 2^32 columns");
        off[r + 1] = off[r] + (wanted[r] ? (n + 15) / 16 : 0);
    }
    diff_.assign(off[nRec], 0);
    const double t2 = nowS();
    parallelFor(nRec, 64, [&](std::size_t r) {
        AlnRecord &rec = recs_[r];
        // parseDiff, as in addRecord
        const char *ql = fl.data(3 * r + 1), *rl = fl.data(3 * r + 2);
        const std::size_t n = fl.length(3 * r + 1), rn = fl.length(3 * r + 2);
        rec.diffOff = off[r];
        if (!wanted[r]) return;  // (header only: nCols = nEmit = nRadv = 0)
        rec.nCols = static_cast<std::uint32_t>(n);
        std::uint32_t *w = diff_.data() + off[r];
        std::uint32_t nEmit = 0, nRadv = 0;
        classifyColumns(ql, n, rl, rn, w, nEmit, nRadv);
        rec.nEmit = nEmit;
        rec.nRadv = nRadv;
    });
    if (timing && filter) std::fprintf(stderr, "[timing]   ALN %s: columns of %zu of %zu records (this rank's reads)\n", path.c_str(), nWanted, nRec);
    if (timing)
        std::fprintf(stderr, "[timing]   ALN %s: %zu records, lines found %.3f s, offsets %.3f s, headers + columns %.3f s\n", path.c_str(), nRec, t1 - t0, t2 - t1,
                     nowS() - t2);
    return true;
}

// ------------------------------------------------------------------------------------------------ packed sidecar
namespace {
struct PackedHeader {
    char magic[8];  // "PAGALN1\0"
    std::uint32_t version, flavor;
    std::uint64_t nRecs, nDiffWords, nameBytes;
    std::uint64_t srcSize;
    std::int64_t srcMtimeSec, srcMtimeNsec;
};
struct PackedRec {
    std::uint64_t score, queryBegin, queryEnd, refBegin, refEnd, diffOff;
    std::uint32_t nCols, nEmit, nRadv, queryName, refName;  // names: offsets of NUL-terminated strings in the blob
    std::uint32_t forward;
};
const char kMagic[8] = {'P', 'A', 'G', 'A', 'L', 'N', '1', '\0'};

bool statOf(const std::string &path, std::uint64_t *size, std::int64_t *sec, std::int64_t *nsec) {
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return false;
    *size = static_cast<std::uint64_t>(st.st_size);
    *sec = static_cast<std::int64_t>(st.st_mtim.tv_sec);
    *nsec = static_cast<std::int64_t>(st.st_mtim.tv_nsec);
    return true;
}
}  // namespace

bool AlnDb::loadPacked(const std::string &alnPath, Flavor flavor) {
    std::uint64_t size = 0;
    std::int64_t sec = 0, nsec = 0;
    if (!statOf(alnPath, &size, &sec, &nsec)) return false;
    std::FILE *f = std::fopen(sidecarPath(alnPath).c_str(), "rb");
    if (!f) return false;
    PackedHeader h;
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, kMagic, 8) == 0 && h.version == 1 &&
              h.flavor == static_cast<std::uint32_t>(flavor) && h.srcSize == size && h.srcMtimeSec == sec && h.srcMtimeNsec == nsec;
    std::vector<PackedRec> pr;
    std::vector<char> names;
    std::vector<std::uint32_t> diff;
    if (ok) {
        pr.resize(h.nRecs);
        names.resize(h.nameBytes);
        diff.resize(h.nDiffWords);
        ok = (h.nRecs == 0 || std::fread(pr.data(), sizeof(PackedRec), h.nRecs, f) == h.nRecs) &&
             (h.nameBytes == 0 || std::fread(names.data(), 1, h.nameBytes, f) == h.nameBytes) &&
             (h.nDiffWords == 0 || std::fread(diff.data(), 4, h.nDiffWords, f) == h.nDiffWords) &&
             (names.empty() || names.back() == '\0');
    }
    std::fclose(f);
    if (!ok) return false;
    for (const PackedRec &r : pr)  // a damaged file must not send anyone out of bounds
        if (r.queryName >= std::max<std::size_t>(names.size(), 1) || r.refName >= std::max<std::size_t>(names.size(), 1) ||
            r.diffOff + (static_cast<std::uint64_t>(r.nCols) + 15) / 16 > diff.size())
            return false;
    recs_.assign(pr.size(), AlnRecord{});
    parallelFor(pr.size(), 4096, [&](std::size_t i) {
        const PackedRec &r = pr[i];
        AlnRecord &o = recs_[i];
        if (!names.empty()) {
            o.queryName = names.data() + r.queryName;
            o.refName = names.data() + r.refName;
        }
        o.score = r.score;
        o.queryBegin = r.queryBegin;
        o.queryEnd = r.queryEnd;
        o.refBegin = r.refBegin;
        o.refEnd = r.refEnd;
        o.forward = r.forward != 0;
        o.diffOff = r.diffOff;
        o.nCols = r.nCols;
        o.nEmit = r.nEmit;
        o.nRadv = r.nRadv;
    });
    diff_.swap(diff);
    fromSidecar_ = true;
    return true;
}

bool AlnDb::savePacked(const std::string &alnPath, Flavor flavor) const {
    PackedHeader h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.flavor = static_cast<std::uint32_t>(flavor);
    if (!statOf(alnPath, &h.srcSize, &h.srcMtimeSec, &h.srcMtimeNsec)) return false;
    std::vector<PackedRec> pr(recs_.size());
    std::vector<char> names;
    std::unordered_map<std::string, std::uint32_t> seen;
    auto intern = [&](const std::string &s) -> std::uint32_t {
        auto it = seen.find(s);
        if (it != seen.end()) return it->second;
        if (names.size() + s.size() + 1 > 0xFFFFFFFFull) throw std::runtime_error("sidecar: name blob over 4 GiB");
        const std::uint32_t off = static_cast<std::uint32_t>(names.size());
        names.insert(names.end(), s.begin(), s.end());
        names.push_back('\0');
        seen.emplace(s, off);
        return off;
    };
    for (std::size_t i = 0; i < recs_.size(); ++i) {
        const AlnRecord &r = recs_[i];
        if (r.queryName.find('\0') != std::string::npos || r.refName.find('\0') != std::string::npos) return false;
        pr[i] = PackedRec{r.score, r.queryBegin, r.queryEnd, r.refBegin, r.refEnd, r.diffOff, r.nCols, r.nEmit, r.nRadv,
                          intern(r.queryName), intern(r.refName), r.forward ? 1u : 0u};
    }
    h.nRecs = pr.size();
    h.nDiffWords = diff_.size();
    h.nameBytes = names.size();
    const std::string out = sidecarPath(alnPath), tmp = out + ".tmp." + std::to_string(static_cast<long>(::getpid()));
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && (pr.empty() || std::fwrite(pr.data(), sizeof(PackedRec), pr.size(), f) == pr.size()) &&
              (names.empty() || std::fwrite(names.data(), 1, names.size(), f) == names.size()) &&
              (diff_.empty() || std::fwrite(diff_.data(), 4, diff_.size(), f) == diff_.size());
    ok = (std::fclose(f) == 0) && ok;
    if (ok) ok = std::rename(tmp.c_str(), out.c_str()) == 0;
    if (!ok) std::remove(tmp.c_str());
    return ok;
}

AlnDb::AlnDb(const std::string &path, Flavor flavor, AlnRecordFilter filter) {
    std::ifstream in(path);
    if (!in.is_open()) return;  // the reference silently yields an empty database
    if (loadPacked(path, flavor)) return;

    std::stringstream ss;
    if (flavor == Flavor::Mecat && loadMecatParallel(path, filter)) {
        // done by the thread pool
    } else if (flavor == Flavor::Mecat) {
        filter_ = &filter;
        std::string queryName, refName, forward, score;
        std::size_t queryBegin = 0, queryEnd = 0, querySize = 0, refBegin = 0, refEnd = 0, refSize = 0;
        std::string l1, l2, l3;
        while (std::getline(in, l1)) {
            ss.clear();
            ss.str(l1);
            ss >> queryName >> refName >> forward >> score >> queryBegin >> queryEnd >> querySize >> refBegin >>
                refEnd >> refSize;
            AlnRecord rec;  // a malformed header yields an all-zero record with empty names
            if (!ss.fail()) {
                rec.queryName = queryName;
                rec.refName = refName;
                rec.forward = forward == "F";
                rec.score = static_cast<std::size_t>(std::atoll(score.c_str()));
                rec.queryBegin = queryBegin;
                rec.queryEnd = queryEnd;
                rec.refBegin = refBegin;
                rec.refEnd = refEnd;
            }
            if (!std::getline(in, l2)) break;
            if (!std::getline(in, l3)) break;
            addRecord(std::move(rec), l2, l3);
        }
    } else {
        std::string line, queryName, refName, forward, readDiffStr, ignored;
        std::size_t queryBegin = 0, queryEnd = 0, refBegin = 0, refEnd = 0;
        bool bad = false;
        for (std::size_t lineCount = 0; std::getline(in, line); ++lineCount) {
            if (lineCount % 3 == 0) {
                ss.clear();
                ss.str(line);
                ss >> queryName >> refName >> forward >> ignored >> queryBegin >> queryEnd >> ignored >> refBegin >>
                    refEnd;
                if (ss.fail()) bad = true;
            } else if (lineCount % 3 == 1) {
                if (!bad) readDiffStr = line;
            } else {
                if (!bad) {
                    AlnRecord rec;
                    rec.queryName = queryName;
                    rec.refName = refName;
                    rec.forward = forward == "F";
                    rec.score = queryEnd - queryBegin;
                    rec.queryBegin = queryBegin;
                    rec.queryEnd = queryEnd;
                    rec.refBegin = refBegin;
                    rec.refEnd = refEnd;
                    addRecord(std::move(rec), readDiffStr, line);
                }
                bad = false;
            }
        }
    }
    filter_ = nullptr;
    sortByScore();
    diff_.resize(diff_.size() + 4, 0);  // kernels may read one word past an alignment
    if (const char *e = std::getenv("PAGRAPH_ALN_SIDECAR"))
        if (e[0] == '1' && !(flavor == Flavor::Mecat && filter)) savePacked(path, flavor);  // (never a sidecar of a filtered parse)
}

}  // namespace pagh
