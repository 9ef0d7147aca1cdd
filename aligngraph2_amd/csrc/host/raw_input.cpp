#include "raw_input.hpp"

#include "line_index.hpp"

#include <stdexcept>

namespace pagh {

// the database's records in database order, names -> indices (PAG_NONE: the name is not in the sequence file; such a
// record is in nobody's list, Aligner.cpp:42-45, but still counts for the read->reference coverage when its target is
// known, Aligner.cpp:70-82)
std::vector<pag_raw_aln> RawInput::resolve(const AlnDb &db, const SeqDb &queries, const SeqDb &targets) {
    std::vector<pag_raw_aln> out(db.size());
    parallelFor(db.size(), 4096, [&](std::size_t i) {
        const AlnRecord &r = db[i];
        pag_raw_aln o{};
        o.query = queries.contains(r.queryName) ? static_cast<std::uint32_t>(queries.id(r.queryName)) : PAG_NONE;
        o.target = targets.contains(r.refName) ? static_cast<std::uint32_t>(targets.id(r.refName)) : PAG_NONE;
        o.score = r.score;
        o.q_begin = r.queryBegin;
        o.q_end = r.queryEnd;
        o.t_begin = r.refBegin;
        o.t_end = r.refEnd;
        o.diff_off = r.diffOff;
        o.n_cols = r.nCols;
        o.n_emit = r.nEmit;
        o.n_radv = r.nRadv;
        o.forward = r.forward ? 1u : 0u;
        out[i] = o;
    });
    return out;
}

RawInput::RawInput(const SeqDb &reads_, const SeqDb &ctgs_, const SeqDb &refs_, const AlnDb &readToCtg_, const AlnDb &readToRef_,
                   const AlnDb &ctgToRef_, const BlockConfig &cfg_, const BuildParams &p)
    : reads(reads_), ctgs(ctgs_), refs(refs_), readToCtg(readToCtg_), readToRef(readToRef_), ctgToRef(ctgToRef_), cfg(cfg_), params(p) {
    if (reads.size() >= 0xFFFFFFFFull) throw std::runtime_error("too many reads");
    // filters exactly as the reference main sets them (pagraph.cpp:218-231)
    refAccepted_.assign(refs.size(), 0);
    if (refs.contains(cfg.ref)) refAccepted_[refs.id(cfg.ref)] = 1;
    ctgSelected_.assign(ctgs.size(), 0);
    ctgForward_.assign(ctgs.size(), 1);
    for (auto &c : cfg.contigs) {
        if (!ctgs.contains(c.first)) continue;
        ctgSelected_[ctgs.id(c.first)] = 1;
        ctgForward_[ctgs.id(c.first)] = c.second ? 1 : 0;
    }
    for (std::size_t i = 0; i < ctgs.size(); ++i) ctgLen_.push_back(static_cast<std::uint32_t>(ctgs.length(i)));
    for (std::size_t i = 0; i < refs.size(); ++i) refLen_.push_back(static_cast<std::uint32_t>(refs.length(i)));
    rec1_ = resolve(readToCtg, reads, ctgs);
    rec2_ = resolve(readToRef, reads, refs);
    rec3_ = resolve(ctgToRef, ctgs, refs);

    in_ = pag_raw_input{};
    in_.bulk_on_device = 0;
    in_.n_threads = p.threads;
    in_.reads.n_seqs = reads.size();
    in_.reads.byte_off = reads.byteOff().data();
    in_.reads.len = reads.lens().data();
    in_.reads.packed = reads.packed().data();
    in_.reads.packed_bytes = reads.packed().size();
    in_.read_to_ctg = pag_raw_db{rec1_.size(), rec1_.data(), readToCtg.diff().data(), readToCtg.diff().size()};
    in_.read_to_ref = pag_raw_db{rec2_.size(), rec2_.data(), readToRef.diff().data(), readToRef.diff().size()};
    in_.ctg_to_ref = pag_raw_db{rec3_.size(), rec3_.data(), ctgToRef.diff().data(), ctgToRef.diff().size()};
    in_.n_ctgs = ctgs.size();
    in_.ctg_len = ctgLen_.data();
    in_.ctg_selected = ctgSelected_.data();
    in_.ctg_forward = ctgForward_.data();
    in_.n_refs = refs.size();
    in_.ref_len = refLen_.data();
    in_.ref_accepted = refAccepted_.data();
    in_.read_to_ctg_ratio = p.readToCtgRatio;
    in_.read_to_ref_ratio = p.readToRefRatio;
    if (p.epsilon > 0xFFFFFFFFull || p.covFilter > 0xFFFFFFFFull) throw std::runtime_error("value does not fit 32 bits: epsilon / coverage filter");
    in_.eps = static_cast<std::uint32_t>(p.epsilon);
    in_.cov_filter = static_cast<std::uint32_t>(p.covFilter);
    in_.outer_sample = static_cast<std::uint32_t>(p.outerSample);
    in_.topk_ctg = p.readToCtgTopK;
    in_.topk_ref = p.readToRefTopK;
}

}  // namespace pagh
