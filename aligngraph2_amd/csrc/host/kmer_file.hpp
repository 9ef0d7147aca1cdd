// Solid k-mer set file: u64 k, then u64 codes, native endian (written by the reference's
// kmer_counter.cpp:87-95).  The reference reads k from the first word (FileKmerIterator.cpp:11-14) and
// then iterates EVERY word of the file, the header included, as a k-mer code (:16-44) — quirk Q1 —
// so words() returns the whole file.
#pragma once
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pagh {

class KmerFile {
public:
    explicit KmerFile(const std::string &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("cannot open solid k-mer file: " + path);
        in.seekg(0, std::ios::end);
        std::streamoff bytes = in.tellg();
        in.seekg(0);
        words_.resize(static_cast<std::size_t>(bytes) / 8);  // a trailing partial word is never read
        in.read(reinterpret_cast<char *>(words_.data()), static_cast<std::streamsize>(words_.size() * 8));
        k_ = words_.empty() ? 0 : words_[0];
    }
    std::uint64_t k() const { return k_; }
    const std::vector<std::uint64_t> &words() const { return words_; }

private:
    std::uint64_t k_ = 0;
    std::vector<std::uint64_t> words_;
};

}  // namespace pagh
