#include "aln_db.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "line_index.hpp"

namespace pagh {

void AlnDb::addRecord(AlnRecord rec, const std::string &qline, const std::string &rline) {
    // parseDiff: q=='-' -> (1,0); r=='-' -> (0,1); mismatch -> (1,1); match -> (0,0).  One column
    // per character of the query line; a shorter reference line reads as NUL (mismatch).
    std::size_t n = qline.size();
    if (n > 0xFFFFFFFFull) throw std::runtime_error("alignment longer than 2^32 columns");
    rec.diffOff = diff_.size();
    rec.nCols = static_cast<std::uint32_t>(n);
    diff_.resize(diff_.size() + (n + 15) / 16, 0);
    std::uint32_t *w = diff_.data() + rec.diffOff;
    std::uint32_t nEmit = 0, nRadv = 0;
    for (std::size_t i = 0; i < n; ++i) {
        char q = qline[i];
        char r = i < rline.size() ? rline[i] : '\0';
        unsigned cls;
        if (q == '-') cls = 1;
        else if (r == '-') cls = 2;
        else if (q != r) cls = 3;
        else cls = 0;
        w[i >> 4] |= cls << ((i & 15) * 2);
        nEmit += cls != 1;
        nRadv += cls != 2;
    }
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
bool AlnDb::loadMecatParallel(const std::string &path) {
    FileLines fl;
    if (!fl.load(path)) return false;
    const std::size_t nRec = fl.size() / 3;
    recs_.assign(nRec, AlnRecord{});
    std::vector<std::uint64_t> off(nRec + 1, 0);
    for (std::size_t r = 0; r < nRec; ++r) {
        const std::size_t n = fl.length(3 * r + 1);
        if (n > 0xFFFFFFFFull) throw std::runtime_error("alignment longer than 2^32 columns");
        off[r + 1] = off[r] + (n + 15) / 16;
    }
    diff_.assign(off[nRec], 0);
    parallelFor(nRec, 64, [&](std::size_t r) {
        AlnRecord &rec = recs_[r];
        {   // header
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
        }
        // parseDiff, as in addRecord
        const char *ql = fl.data(3 * r + 1), *rl = fl.data(3 * r + 2);
        const std::size_t n = fl.length(3 * r + 1), rn = fl.length(3 * r + 2);
        rec.diffOff = off[r];
        rec.nCols = static_cast<std::uint32_t>(n);
        std::uint32_t *w = diff_.data() + off[r];
        std::uint32_t nEmit = 0, nRadv = 0;
        for (std::size_t i = 0; i < n; ++i) {
            const char q = ql[i];
            const char rr = i < rn ? rl[i] : '\0';
            unsigned cls;
            if (q == '-') cls = 1;
            else if (rr == '-') cls = 2;
            else if (q != rr) cls = 3;
            else cls = 0;
            w[i >> 4] |= cls << ((i & 15) * 2);
            nEmit += cls != 1;
            nRadv += cls != 2;
        }
        rec.nEmit = nEmit;
        rec.nRadv = nRadv;
    });
    return true;
}

AlnDb::AlnDb(const std::string &path, Flavor flavor) {
    std::ifstream in(path);
    if (!in.is_open()) return;  // the reference silently yields an empty database

    std::stringstream ss;
    if (flavor == Flavor::Mecat && loadMecatParallel(path)) {
        // done by the thread pool
    } else if (flavor == Flavor::Mecat) {
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
    sortByScore();
    diff_.resize(diff_.size() + 4, 0);  // kernels may read one word past an alignment
}

}  // namespace pagh
