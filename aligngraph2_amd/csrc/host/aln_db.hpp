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

// The bulk half of an ALN parse — the column classes of every record's two rows (parseDiff) and the two counts per record — as
// a hook: the HIP backend installs pag_classify_columns_host (k_ingest.hip) when PAGRAPH_DEVICE_INGEST=1; without a hook, or
// when it returns false, the host loop below does the work.  Arguments as pag_classify_columns_host takes them.
using ColumnClassifier = bool (*)(const char *text, std::uint64_t textBytes, const std::uint64_t *qOff, const std::uint32_t *qLen,
                                  const std::uint64_t *rOff, const std::uint32_t *rLen, const std::uint64_t *diffOff, std::uint64_t nRecs,
                                  std::uint32_t *diff, std::uint64_t nDiffWords, std::uint32_t *nEmit, std::uint32_t *nRadv);
void setColumnClassifier(ColumnClassifier f);

// A rank of a sharded build (PAGRAPH_SHARD, SURVEY 8e level 2) extracts the reads of ITS emission range only: of every other
// read's alignments it needs the header (per-query lists, eligibility, the coverage filter: header fields), never the columns.
// With a filter set, a read database's records whose query name it rejects keep their header and get no column classes
// (nCols = nEmit = nRadv = 0): the bulk of the parse — 2.2 B of text per aligned base — and of the staged bytes is then the
// rank's own share.  Applies to Flavor::Mecat text parses; a packed sidecar is loaded as it is.
using AlnRecordFilter = std::function<bool(const char *queryName, std::size_t len)>;
void setAlnRecordFilter(AlnRecordFilter f);  // (empty: none)

class AlnDb {
public:
    enum class Flavor {
        Mecat,   // read->contig / read->ref: score = atoll(col 4), 10 header fields
        MummerV2 // contig->ref: score = qEnd - qBegin, 9 header fields read
    };
    AlnDb() = default;
    AlnDb(const std::string &path, Flavor flavor);

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
    bool loadMecatParallel(const std::string &path);
    bool fromSidecar_ = false;
    bool filterOn_ = false;  // (the sequential text parse of a read database: addRecord asks the record filter)
    std::vector<AlnRecord> recs_;
    std::vector<std::uint32_t> diff_;
};

}  // namespace pagh
