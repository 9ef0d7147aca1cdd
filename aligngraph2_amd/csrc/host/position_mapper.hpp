// Single-coordinate address space of a sequence database.
// Restates PositionMapper (PAGraph/src/tools/position/PositionMapper.cpp:16-70): forward strand of
// sequence i lives at [start[i], start[i]+len[i]), its reverse strand at start[i]+2*len[i]+pos;
// 0 means "no coordinate".
#pragma once
#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "seq_db.hpp"

namespace pagh {

class PositionMapper {
public:
    PositionMapper() = default;
    explicit PositionMapper(const SeqDb &db) {
        for (std::size_t i = 0; i < db.size(); ++i) sizes_.push_back(db.length(i));
        if (sizes_.empty()) return;
        start_.push_back(sizes_[0]);
        for (std::size_t i = 1; i < sizes_.size(); ++i)
            start_.push_back(start_.back() + 3 * sizes_[i - 1] + std::max(sizes_[i - 1], sizes_[i]));
        start_.push_back(start_.back() + 4 * sizes_.back());
    }
    std::size_t extraStart() const { return start_.empty() ? 0 : start_.back(); }
    std::size_t dualToSingle(std::int64_t idx, std::int64_t pos) const {
        if (idx == 0) return 0;
        std::size_t i = static_cast<std::size_t>(idx > 0 ? idx - 1 : -idx - 1);
        std::size_t off = idx > 0 ? 0 : 2 * sizes_[i];
        return start_[i] + off + static_cast<std::size_t>(pos);
    }
    std::pair<std::int64_t, std::int64_t> singleToDual(std::size_t single) const {
        if (single == 0) return {0, 0};
        auto it = std::upper_bound(start_.begin(), start_.end(), single);
        if (it != start_.begin()) it = std::prev(it);
        std::int64_t idx = it - start_.begin();
        // the reference's arithmetic is unsigned here; offset can only be compared after the subtraction
        std::size_t offset = single - *it;
        if (offset >= 2 * sizes_[static_cast<std::size_t>(idx)]) {
            offset -= 2 * sizes_[static_cast<std::size_t>(idx)];
            idx = -(idx + 1);
        } else {
            ++idx;
        }
        return {idx, static_cast<std::int64_t>(offset)};
    }
    std::size_t size(std::int64_t idx) const {
        if (idx == 0) return 0;
        return sizes_[static_cast<std::size_t>(idx > 0 ? idx - 1 : -idx - 1)];
    }
    std::size_t start(std::size_t i) const { return start_[i]; }

private:
    std::vector<std::size_t> sizes_;
    std::vector<std::size_t> start_;
};

}  // namespace pagh
