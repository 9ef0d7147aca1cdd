// Host-side alignment database: the 3-line ALN (".ref") parser and the column diff bit-vectors.
// Restates the reference's MecatAlignDatabase / MummerAlignDatabaseV2 / AlignmentHelper /
// ParseAlignTools::parseDiff (PAGraph/src/tools/align/MecatAlignDatabase.cpp:8-19,
// MummerAlignDatabaseV2.cpp:7-49, AlignmentHelper.cpp:11-48, ParseAlignTools.cpp:8-26).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace pagh {

struct AlnRecord {
    std::string queryName, refName;
    std::size_t score = 0;
    std::size_t queryBegin = 0, queryEnd = 0, refBegin = 0, refEnd = 0;
    bool forward = false;
    std::uint64_t diffOff = 0;  // first 32-bit word in AlnDb::diff
    std::uint32_t nCols = 0;
    std::uint32_t nEmit = 0;  // columns that emit a query base (classes 00, 11, 10)
    std::uint32_t nRadv = 0;  // columns that advance the target (classes 00, 11, 01)
};

// A rank of a sharded build (PAGRAPH_SHARD, SURVEY 8e level 2) extracts the reads of ITS emission range only: of every other
// read's alignments it needs the header (per-query lists, eligibility, the coverage filter: header fields), never the columns.
// A database constructed with a filter: the records whose query name it rejects keep their header and get no column classes
// (nCols = nEmit = nRadv = 0): the bulk of the parse — 2.2 B of text per aligned base — and of the staged bytes is then the
// rank's own share.  Applies to Flavor::Mecat text parses; a packed sidecar is loaded as it is, and a filtered parse writes none.
// The filter belongs to the database it was handed to (it is called from the parser's threads while the constructor runs and
// dropped when it returns): nothing outlives the objects it refers to, whichever way the constructor ends.
using AlnRecordFilter = std::function<bool(const char *queryName, std::size_t len)>;

class AlnDb {
public:
    enum class Flavor {
        Mecat,   // read->contig / read->ref: score = atoll(col 4), 10 header fields
        MummerV2 // contig->ref: score = qEnd - qBegin, 9 header fields read
    };
    AlnDb() = default;
    AlnDb(const std::string &path, Flavor flavor, AlnRecordFilter filter = nullptr);

    std::size_t size() const { return recs_.size(); }
    const AlnRecord &operator[](std::size_t i) const { return recs_[i]; }
    const std::vector<std::uint32_t> &diff() const { return diff_; }

    // column class of column c of record r: bit0 = queryDiff, bit1 = refDiff
    unsigned colClass(const AlnRecord &r, std::size_t c) const {
        return (diff_[r.diffOff + (c >> 4)] >> ((c & 15) * 2)) & 3u;
    }
    // ParseAlignTools::exactAlign (ParseAlignTools.tcc:44-70)
    template <typename F>
    void exactAlign(const AlnRecord &r, std::size_t queryBegin, std::size_t refBegin, bool forward, F f) const {
        std::size_t q = queryBegin, t = refBegin;
        for (std::size_t jj = 0; jj < r.nCols; ++jj) {
            unsigned cls = colClass(r, forward ? jj : r.nCols - jj - 1);
            if (cls == 0 || cls == 3) {
                f(q, t);
                ++t;
                ++q;
            } else if (cls == 1) {
                ++t;
            } else {
                f(q, t);
                ++q;
            }
        }
    }

    void addRecord(AlnRecord rec, const std::string &qline, const std::string &rline);
    void sortByScore();

    // Packed sidecar "<aln path>.pagaln" (SURVEY.md 8f.2): the database exactly as the constructor leaves it (records in
    // their sorted order, names, 2 diff bits per column), stamped with the size and modification time of the text file
    // it was made from.  The constructor loads a valid sidecar instead of parsing the text (the text parse is ~2.2 B of
    // text per aligned base; the sidecar is 0.25 B); it WRITES one only when PAGRAPH_ALN_SIDECAR=1 is set.
    static std::string sidecarPath(const std::string &alnPath) { return alnPath + ".pagaln"; }
    bool loadPacked(const std::string &alnPath, Flavor flavor);
    bool savePacked(const std::string &alnPath, Flavor flavor) const;
    bool fromSidecar() const { return fromSidecar_; }

private:
    bool loadMecatParallel(const std::string &path, const AlnRecordFilter &filter);
    bool fromSidecar_ = false;
    const AlnRecordFilter *filter_ = nullptr;  // (set while the sequential text parse of a filtered read database runs: addRecord asks it)
    std::vector<AlnRecord> recs_;
    std::vector<std::uint32_t> diff_;
};

}  // namespace pagh
