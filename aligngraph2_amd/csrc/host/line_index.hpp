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

namespace pagh {

inline unsigned hostThreads(std::size_t work_items, unsigned cap = 32) {
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
    // returns false if the file cannot be opened
    bool load(const std::string &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return false;
        in.seekg(0, std::ios::end);
        const std::streamoff bytes = in.tellg();
        in.seekg(0);
        buf_.resize(static_cast<std::size_t>(bytes));
        if (bytes) in.read(buf_.data(), bytes);
        buf_.resize(static_cast<std::size_t>(in.gcount()));
        index();
        return true;
    }
    std::size_t size() const { return start_.size(); }
    const char *data(std::size_t i) const { return buf_.data() + start_[i]; }
    std::size_t length(std::size_t i) const {
        const std::size_t end = i + 1 < start_.size() ? start_[i + 1] - 1 : (endsWithNewline_ ? buf_.size() - 1 : buf_.size());
        return end - start_[i];
    }
    std::string str(std::size_t i) const { return std::string(data(i), length(i)); }

private:
    void index() {
        const std::size_t n = buf_.size();
        start_.clear();
        endsWithNewline_ = n && buf_[n - 1] == '\n';
        if (n == 0) return;
        const std::size_t chunk = 1 << 22;
        const std::size_t chunks = (n + chunk - 1) / chunk;
        std::vector<std::vector<std::size_t>> found(chunks);
        parallelFor(chunks, 1, [&](std::size_t c) {
            const char *p = buf_.data() + c * chunk, *e = buf_.data() + std::min(n, (c + 1) * chunk);
            auto &v = found[c];
            while (p < e) {
                const char *q = static_cast<const char *>(std::memchr(p, '\n', static_cast<std::size_t>(e - p)));
                if (!q) break;
                v.push_back(static_cast<std::size_t>(q - buf_.data()));
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
    std::vector<char> buf_;
    std::vector<std::size_t> start_;
    bool endsWithNewline_ = false;
};

inline bool isSpaceC(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

}  // namespace pagh
