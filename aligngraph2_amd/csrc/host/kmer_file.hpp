// Solid k-mer set file: u64 k, then u64 codes, native endian (written by the reference's
// kmer_counter.cpp:87-95).  The reference reads k from the first word (FileKmerIterator.cpp:11-14) and
// then iterates EVERY word of the file, the header included, as a k-mer code (:16-44) — quirk Q1 —
// so data() / size() cover the whole file.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdint>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pagh {

class KmerFile {
public:
    // The file is mapped (a 0.7 GB solid set read through an ifstream into a zero-filled vector cost 0.2 s before a single
    // word was used); a file that cannot be mapped is read.
    explicit KmerFile(const std::string &path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open solid k-mer file: " + path);
        struct stat st {};
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size >= 8) {
            void *m = ::mmap(nullptr, static_cast<std::size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = m;
                mapBytes_ = static_cast<std::size_t>(st.st_size);
                data_ = static_cast<const std::uint64_t *>(m);
                n_ = mapBytes_ / 8;  // a trailing partial word is never read
            }
        }
        ::close(fd);
        if (!map_) {
            std::ifstream in(path, std::ios::binary);
            if (!in) throw std::runtime_error("cannot open solid k-mer file: " + path);
            in.seekg(0, std::ios::end);
            std::streamoff bytes = in.tellg();
            in.seekg(0);
            words_.resize(static_cast<std::size_t>(bytes > 0 ? bytes : 0) / 8);
            in.read(reinterpret_cast<char *>(words_.data()), static_cast<std::streamsize>(words_.size() * 8));
            data_ = words_.data();
            n_ = words_.size();
        }
        k_ = n_ == 0 ? 0 : data_[0];
    }
    KmerFile(const KmerFile &) = delete;
    KmerFile &operator=(const KmerFile &) = delete;
    ~KmerFile() {
        if (map_) ::munmap(map_, mapBytes_);
    }
    std::uint64_t k() const { return k_; }
    // every 64-bit word of the file, the header word included
    const std::uint64_t *data() const { return data_; }
    std::size_t size() const { return n_; }

private:
    std::uint64_t k_ = 0;
    const std::uint64_t *data_ = nullptr;
    std::size_t n_ = 0;
    void *map_ = nullptr;
    std::size_t mapBytes_ = 0;
    std::vector<std::uint64_t> words_;  // (only when the file could not be mapped)
};

}  // namespace pagh
