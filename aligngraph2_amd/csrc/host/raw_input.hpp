// One config block (one reference sequence) as the C ABI's device-side preparation stage takes it (include/pagraph_hip.h:
// pag_raw_input / pag_prepare): the parsed alignment records with their names resolved to sequence indices, the sequence
// tables, and the filter settings of the reference main.  Everything the reference does with these records before and at
// the top of its two hot loops — per-query lists in score order (Aligner::mergeAlignInfHelper, PAGraph/src/tools/align/
// Aligner.cpp:32-56), the contig->reference per-base map (Aligner::simpleAlign :97-202, AlignReference.cpp:41-79), the
// per-alignment eligibility tests and coordinate flips of parseToCtg / parseToRef (Aligner.tcc:40-71, :121-152) — happens
// on the device (csrc/hip/k_prepare.hip).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "aln_db.hpp"
#include "pagraph_hip.h"
#include "seq_db.hpp"

namespace pagh {

struct BlockConfig {  // one block of config.txt (reference pagraph.cpp:21-49)
    std::string ref;
    std::vector<std::pair<std::string, bool>> contigs;  // (name, forward)
    std::string readPath, ctgAlnPath, refAlnPath;
};

struct BuildParams {  // hard-coded in the reference main (pagraph.cpp:110-125)
    unsigned threads = 16;
    std::size_t epsilon = 10;
    std::size_t covFilter = 1;
    int outerSample = 3;
    int readToCtgTopK = -1;
    int readToRefTopK = -1;
    double readToCtgRatio = 0.35;
    double readToRefRatio = 0.10;
};

class RawInput {
public:
    RawInput(const SeqDb &reads, const SeqDb &ctgs, const SeqDb &refs, const AlnDb &readToCtg, const AlnDb &readToRef,
             const AlnDb &ctgToRef, const BlockConfig &cfg, const BuildParams &p);
    const pag_raw_input &view() const { return in_; }

    // the sources (the test harness builds its host restatement of the preparation stage from them)
    const SeqDb &reads, &ctgs, &refs;
    const AlnDb &readToCtg, &readToRef, &ctgToRef;
    const BlockConfig &cfg;
    const BuildParams &params;

private:
    static std::vector<pag_raw_aln> resolve(const AlnDb &db, const SeqDb &queries, const SeqDb &targets);
    std::vector<pag_raw_aln> rec1_, rec2_, rec3_;
    std::vector<std::uint32_t> ctgLen_, refLen_;
    std::vector<std::uint8_t> ctgSelected_, ctgForward_, refAccepted_;
    pag_raw_input in_{};
};

}  // namespace pagh
