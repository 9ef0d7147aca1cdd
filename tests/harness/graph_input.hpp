// tests/harness/graph_input.hpp — TEST HARNESS (never part of the product).
// The HOST restatement of the preparation stage: builds the flat description of one config block (pag_build_input) that
// the product's device stage (pag_prepare, csrc/hip/k_prepare.hip) produces in HBM.  The tests compare the two array by
// array; the oracle-backed harness programs (no GPU) run on this one.
//
// This is the host half of what the reference does around its two hot loops: per-query alignment
// lists in score order (Aligner::mergeAlignInfHelper, PAGraph/src/tools/align/Aligner.cpp:32-56),
// the contig->reference per-base map (Aligner::simpleAlign :97-202, AlignReference::insert /
// addExtraPosition, AlignReference.cpp:41-79), the per-alignment eligibility tests and coordinate
// flips at the top of parseToCtg / parseToRef (Aligner.tcc:40-71, :121-152), and the coordinate
// mapping of PositionProcessor::transformPosition (PositionProcessor.cpp:37-55).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "aln_db.hpp"
#include "pagraph_hip.h"
#include "position_mapper.hpp"
#include "raw_input.hpp"
#include "seq_db.hpp"

namespace pagh {

class GraphInput {
public:
    GraphInput(const SeqDb &reads, const SeqDb &ctgs, const SeqDb &refs, const AlnDb &readToCtg,
               const AlnDb &readToRef, const AlnDb &ctgToRef, const BlockConfig &cfg, const BuildParams &p);

    const pag_build_input &view() const { return in_; }
    const PositionMapper &ctgMapper() const { return ctgMapper_; }
    const PositionMapper &refMapper() const { return refMapper_; }

private:
    struct ListEntry {
        std::size_t score;
        std::size_t rec;
        std::size_t tgt;
    };
    static std::vector<std::vector<ListEntry>> mergeLists(const AlnDb &db, const SeqDb &queries, const SeqDb &targets);
    void buildCtgMap(const SeqDb &ctgs, const SeqDb &refs, const AlnDb &ctgToRef);
    void buildPass1(const SeqDb &reads, const SeqDb &ctgs, const AlnDb &db, const BuildParams &p);
    void buildPass2(const SeqDb &reads, const SeqDb &refs, const AlnDb &db, const BuildParams &p);

    PositionMapper ctgMapper_, refMapper_;
    std::vector<bool> refAccepted_, ctgSelected_, ctgForward_;

    std::vector<pag_ctg> ctgTab_;
    std::vector<pag_ref> refTab_;
    std::vector<std::uint32_t> ctgEntOff_, ctgEnt_;
    std::vector<pag_aln> aln1_, aln2_;
    std::vector<std::uint64_t> qoff1_, qoff2_;
    std::vector<std::uint32_t> emitOrder_;
    pag_build_input in_{};
};

}  // namespace pagh
