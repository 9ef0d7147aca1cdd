// Host-side sequence database: FASTA/FASTQ loader + 2-bit packed storage.
// Restates (does not copy) the behaviour of the reference's AutoSeqDatabase / SeqHelper /
// CompressedSeq (PAGraph/src/tools/seq/AutoSeqDatabase.cpp:9-22, SeqHelper.cpp:8-99,
// CompressedSeq.cpp:8-87): file type sniffed from the first byte, name = first whitespace token of
// the header minus its first character, every non-ACGT base becomes A, 4 bases per byte LSB first.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace pagh {

class SeqDb {
public:
    static constexpr std::size_t kNotFound = static_cast<std::size_t>(-1);

    // One of several ranks that share a block packs the bases of ITS reads only: the records whose position in the emission
    // order (thread-major strided over `threads` workers, MultiThreadTools.tcc:8-14) lies in [n rank / world, n (rank + 1) / world).
    // Every other record keeps its name and length and points at one shared stretch of zero bytes (it reads as A's).
    struct PackWindow {
        unsigned rank = 0, world = 1, threads = 1;
        bool wants(std::uint64_t i, std::uint64_t n) const {
            const std::uint64_t T = threads ? threads : 1, t = i % T, full = n / T, rem = n % T;
            const std::uint64_t pos = t * full + (t < rem ? t : rem) + i / T;
            return pos >= n * rank / world && pos < n * (rank + 1ull) / world;
        }
    };
    // throws std::runtime_error when the file cannot be opened
    explicit SeqDb(const std::string &path, const PackWindow *window = nullptr);
    SeqDb() = default;

    std::size_t size() const { return len_.size(); }
    std::uint32_t length(std::size_t id) const { return len_[id]; }
    const std::string &name(std::size_t id) const { return names_[id]; }
    bool contains(const std::string &name) const { return nameToId_.count(name) > 0; }
    std::size_t id(const std::string &name) const {
        auto it = nameToId_.find(name);
        return it == nameToId_.end() ? kNotFound : it->second;
    }
    // CompressedSeq::toString: forward, or reverse complement read back to front
    std::string toString(std::size_t id, bool forward) const;
    // CompressedSeq::baseAt: 'N' past the end; reverse strand = complement of base len-1-idx
    char baseAt(std::size_t id, std::size_t idx, bool forward) const;
    std::uint64_t totalBases() const { return totalBases_; }

    // flat view for the C ABI (pag_seqs)
    const std::vector<std::uint64_t> &byteOff() const { return byteOff_; }
    const std::vector<std::uint32_t> &lens() const { return len_; }
    const std::vector<std::uint8_t> &packed() const { return packed_; }

    void add(const std::string &comment, const std::string &seq);
    // adopt an already 2-bit packed sequence (pag_seqs layout)
    void addPacked(const std::string &name, const std::uint8_t *packed, std::uint32_t len);
    void finish();  // pads the packed buffer

private:
    bool loadFastqParallel(const std::string &path, const PackWindow *window);
    bool loadFastaParallel(const std::string &path);
    std::vector<std::string> names_;
    std::vector<std::uint32_t> len_;
    std::vector<std::uint64_t> byteOff_;
    std::vector<std::uint8_t> packed_;
    std::unordered_map<std::string, std::size_t> nameToId_;
    std::uint64_t totalBases_ = 0;
    std::string lastName_;  // the reference keeps `name` across records (AutoSeqDatabase.cpp:12)
};

}  // namespace pagh
