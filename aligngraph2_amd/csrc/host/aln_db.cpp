#include "aln_db.hpp"

#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

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

AlnDb::AlnDb(const std::string &path, Flavor flavor) {
    std::ifstream in(path);
    if (!in.is_open()) return;  // the reference silently yields an empty database

    std::stringstream ss;
    if (flavor == Flavor::Mecat) {
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
