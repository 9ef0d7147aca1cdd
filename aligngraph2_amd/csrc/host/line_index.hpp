// Whole-file line index for the parallel text loaders: the file is read once, the '\n' positions are found by a
// pool of threads, and line i is (begin, length) exactly as the i-th successful std::getline would return it
// (no '\n', a '\r' stays, a last line without '\n' counts, a trailing '\n' does not open an empty last line).
#pragma once
#include <algorithm>
#include "host_threads.hpp"
#include <atomic>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pagh {

inline unsigned hostThreads(std::size_t work_items, unsigned cap = 64) {
    unsigned hw = std::max(1u, usableCpus());
    return static_cast<unsigned>(std::max<std::size_t>(1, std::min<std::size_t>(std::min(hw, cap), work_items)));
}

// run f(i) for i in [0, n) on a pool of threads (dynamic chunks of `grain`)
template <typename F>
inline void parallelFor(std::size_t n, std::size_t grain, F f) {
    if (n == 0) return;
    const std::size_t chunks = (n + grain - 1) / grain;
    const unsigned T = hostThreads(chunks);
    std::atomic<std::size_t> next{0};
    auto worker = [&]() {
        for (std::size_t c; (c = next.fetch_add(1)) < chunks;) {
            const std::size_t lo = c * grain, hi = std::min(n, lo + grain);
            for (std::size_t i = lo; i < hi; ++i) f(i);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
}

class FileLines {
public:
    FileLines() = default;
    FileLines(const FileLines &) = delete;
    FileLines &operator=(const FileLines &) = delete;
    ~FileLines() { release(); }
    // returns false if the file cannot be opened.  The file is MAPPED (a 2 GB input read through an ifstream into a
    // zero-initialised vector cost ~0.7 s on one thread before the first line was parsed); a file that cannot be mapped
    // (a pipe, a special file) is read.
    bool load(const std::string &path) {
        release();
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st {};
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
            if (st.st_size == 0) {
                ::close(fd);
                index();
                return true;
            }
            void *m = ::mmap(nullptr, static_cast<std::size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = m;
                mapBytes_ = static_cast<std::size_t>(st.st_size);
                base_ = static_cast<const char *>(m);
                n_ = mapBytes_;
                ::close(fd);
                index();
                return true;
            }
        }
        ::close(fd);
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        in.seekg(0, std::ios::end);
        const std::streamoff bytes = in.tellg();
        in.seekg(0);
        buf_.resize(static_cast<std::size_t>(bytes > 0 ? bytes : 0));
        if (bytes > 0) in.read(buf_.data(), bytes);
        buf_.resize(static_cast<std::size_t>(in.gcount()));
        base_ = buf_.data();
        n_ = buf_.size();
        index();
        return true;
    }
    std::size_t size() const { return start_.size(); }
    const char *base() const { return base_; }       // the whole text
    std::size_t bytes() const { return n_; }
    std::size_t offset(std::size_t i) const { return start_[i]; }  // of line i in the text
    const char *data(std::size_t i) const { return base_ + start_[i]; }
    std::size_t length(std::size_t i) const {
        const std::size_t end = i + 1 < start_.size() ? start_[i + 1] - 1 : (endsWithNewline_ ? n_ - 1 : n_);
        return end - start_[i];
    }
    std::string str(std::size_t i) const { return std::string(data(i), length(i)); }

private:
    void release() {
        if (map_) ::munmap(map_, mapBytes_);
        map_ = nullptr;
        mapBytes_ = 0;
        base_ = nullptr;
        n_ = 0;
        buf_.clear();
        start_.clear();
    }
    void index() {
        const std::size_t n = n_;
        start_.clear();
        endsWithNewline_ = n && base_[n - 1] == '\n';
        if (n == 0) return;
        const std::size_t chunk = 1 << 22;
        const std::size_t chunks = (n + chunk - 1) / chunk;
        std::vector<std::vector<std::size_t>> found(chunks);
        parallelFor(chunks, 1, [&](std::size_t c) {
            const char *p = base_ + c * chunk, *e = base_ + std::min(n, (c + 1) * chunk);
            auto &v = found[c];
            while (p < e) {
                const char *q = static_cast<const char *>(std::memchr(p, '\n', static_cast<std::size_t>(e - p)));
                if (!q) break;
                v.push_back(static_cast<std::size_t>(q - base_));
                p = q + 1;
            }
        });
        std::size_t total = 1;
        for (auto &v : found) total += v.size();
        start_.reserve(total);
        start_.push_back(0);
        for (auto &v : found)
            for (std::size_t nl : v)
                if (nl + 1 < n) start_.push_back(nl + 1);
    }
    std::vector<char> buf_;  // (only when the file could not be mapped)
    void *map_ = nullptr;
    std::size_t mapBytes_ = 0;
    const char *base_ = nullptr;
    std::size_t n_ = 0;
    std::vector<std::size_t> start_;
    bool endsWithNewline_ = false;
};

inline bool isSpaceC(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

}  // namespace pagh
