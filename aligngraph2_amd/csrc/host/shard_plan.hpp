// Who traverses which contigs, and which part of the finished graph that takes, when ONE config block is built by several
// GPUs (include/pagraph_hip.h: pag_region / pag_shard_select / pag_shard_run; SURVEY.md 8e level 2).
//
// Contigs are traversed independently (reference PAGraph/src/tools/graph/PAssembly.cpp:30-79).  They are dealt out in
// contiguous runs along the reference, balanced by length, so that the reference bands of a rank's contigs merge into one
// stretch.  A rank's region: the traversed strands of its contigs; the landing zone (first 1 - startSplit) of every contig
// strand — the only place a vertex with a contig coordinate outside the own strand survives a classification
// (PAlgorithm.tcc:60-67); the coordinate-free vertices of the reference bands its contigs map to, plus a halo.
#pragma once
#include <cstdint>
#include <vector>

#include "pagraph_hip.h"
#include "raw_input.hpp"

namespace pagh {

struct ShardPlan {
    std::vector<std::vector<std::size_t>> deal;  // per rank: contig indices
    std::vector<int> ownerOf;                    // per contig: rank, -1 = not selected in this block
    std::vector<std::vector<std::uint32_t>> ctgIv, refIv;  // per rank: [lo, hi) pairs
    std::vector<std::vector<std::uint8_t>> refOpen;
    std::vector<pag_region> regions;             // views of the above
};

// orient[c]: PAG_ORIENT_* of contig c as the block lists it
ShardPlan planShards(const RawInput &raw, const std::vector<std::int32_t> &orient, unsigned world, std::uint64_t halo, double startSplit);

}  // namespace pagh
