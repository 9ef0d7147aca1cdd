#include "graph_input.hpp"
#include "line_index.hpp"

#include <algorithm>
#include <stdexcept>

namespace pagh {

namespace {

std::uint32_t fitU32(std::size_t v, const char *what) {
    if (v > 0xFFFFFFFFull) throw std::runtime_error(std::string("value does not fit 32 bits: ") + what);
    return static_cast<std::uint32_t>(v);
}

// Aligner::flipPosition (Aligner.cpp:235-239), unsigned arithmetic like the reference
void flip(std::size_t &left, std::size_t &right, std::size_t length) {
    std::size_t tmp = left;
    left = length - right;
    right = length - tmp;
}

}  // namespace

std::vector<std::vector<GraphInput::ListEntry>> GraphInput::mergeLists(const AlnDb &db, const SeqDb &queries,
                                                                       const SeqDb &targets) {
    // Aligner::mergeAlignInfHelper (Aligner.cpp:32-56): per-query lists in database order, then an
    // (unstable) std::sort by score, descending — same comparator, same initial order, same libstdc++.
    std::vector<std::vector<ListEntry>> lists(queries.size());
    for (std::size_t i = 0; i < db.size(); ++i) {
        const AlnRecord &r = db[i];
        if (queries.contains(r.queryName) && targets.contains(r.refName))
            lists[queries.id(r.queryName)].push_back({r.score, i, targets.id(r.refName)});
    }
    for (auto &l : lists)
        std::sort(l.begin(), l.end(), [](const ListEntry &a, const ListEntry &b) { return a.score > b.score; });
    return lists;
}

void GraphInput::buildCtgMap(const SeqDb &ctgs, const SeqDb &refs, const AlnDb &ctgToRef) {
    auto lists = mergeLists(ctgToRef, ctgs, refs);
    ctgTab_.assign(ctgs.size(), pag_ctg{});
    ctgEntOff_.clear();
    ctgEnt_.clear();
    const std::size_t nCtg = ctgs.size();

    // Aligner::simpleAlign (Aligner.cpp:152-186): alignments of a contig, list order, only the accepted reference and only
    // the configured orientation; two sweeps (count, fill).  The contigs are independent: both sweeps run on the thread pool,
    // the offsets in between are a serial prefix sum over the contigs.
    auto sweep = [&](std::size_t c, bool fill, std::vector<std::uint32_t> &cnt, std::uint32_t *cursor) {
        const std::size_t len = ctgs.length(c);
        for (auto &e : lists[c]) {
            if (!refAccepted_[e.tgt]) continue;
            const AlnRecord &r = ctgToRef[e.rec];
            if (r.forward != ctgForward_[c]) continue;
            std::size_t cb = r.queryBegin, ce = r.queryEnd;
            if (!r.forward) flip(cb, ce, len);
            std::size_t span = ce > cb ? ce - cb : 0;
            std::size_t k = 0;
            ctgToRef.exactAlign(r, cb, r.refBegin, true, [&](std::size_t, std::size_t refCur) {
                // AlignReference::insert: base cb + k receives (refIdx + 1, refPos[k]) for k < ce - cb
                std::size_t b = cb + k;
                if (k < span && b < len) {
                    if (!fill) {
                        ++cnt[b];
                    } else {
                        ctgEnt_[cursor[b]++] = static_cast<std::uint32_t>(
                            refMapper_.dualToSingle(static_cast<std::int64_t>(e.tgt) + 1, static_cast<std::int64_t>(refCur)));
                    }
                }
                ++k;
            });
        }
    };
    std::vector<std::vector<std::uint32_t>> cnt(nCtg);
    std::vector<std::size_t> runOf(nCtg, 0);  // entries of the contig, AlignReference::addExtraPosition included
    parallelFor(nCtg, 1, [&](std::size_t c) {
        if (!ctgSelected_[c]) return;
        const std::size_t len = ctgs.length(c);
        cnt[c].assign(len, 0);
        sweep(c, false, cnt[c], nullptr);
        std::size_t run = 0;
        std::uint32_t multi = 0;
        for (std::size_t b = 0; b < len; ++b) {
            run += cnt[c][b] ? cnt[c][b] : 1;  // (AlignReference.cpp:69-79: empty lists get (0, 0) -> 0)
            if (cnt[c][b] > 1) multi = 1;
        }
        runOf[c] = run;
        ctgTab_[c].multi = multi;
    });
    std::vector<std::size_t> entBase(nCtg, 0);
    std::size_t offTotal = 0, entTotal = 0;
    for (std::size_t c = 0; c < nCtg; ++c) {
        pag_ctg &t = ctgTab_[c];
        const std::size_t len = ctgs.length(c);
        t.len = static_cast<std::uint32_t>(len);
        t.selected = ctgSelected_[c] ? 1u : 0u;
        std::int64_t dual = ctgForward_[c] ? static_cast<std::int64_t>(c) + 1 : -static_cast<std::int64_t>(c) - 1;
        t.single_base = static_cast<std::uint32_t>(ctgMapper_.dualToSingle(dual, 0));
        t.map_off = offTotal;
        entBase[c] = entTotal;
        if (!ctgSelected_[c]) continue;
        offTotal += len + 1;
        entTotal += runOf[c];
        fitU32(entTotal, "contig map offset");
    }
    ctgEntOff_.assign(offTotal, 0);
    ctgEnt_.assign(entTotal, 0);
    parallelFor(nCtg, 1, [&](std::size_t c) {
        if (!ctgSelected_[c]) return;
        const std::size_t len = ctgs.length(c);
        std::vector<std::uint32_t> cursor(len);
        std::size_t run = entBase[c];
        std::uint32_t *off = ctgEntOff_.data() + ctgTab_[c].map_off;
        for (std::size_t b = 0; b < len; ++b) {
            off[b] = static_cast<std::uint32_t>(run);
            cursor[b] = static_cast<std::uint32_t>(run);
            run += cnt[c][b] ? cnt[c][b] : 1;
        }
        off[len] = static_cast<std::uint32_t>(run);
        sweep(c, true, cnt[c], cursor.data());
        std::vector<std::uint32_t>().swap(cnt[c]);
    });
    if (ctgEntOff_.empty()) ctgEntOff_.push_back(0);
    ctgEnt_.push_back(0);  // never-empty buffers
}

void GraphInput::buildPass1(const SeqDb &reads, const SeqDb &ctgs, const AlnDb &db, const BuildParams &p) {
    auto lists = mergeLists(db, reads, ctgs);
    aln1_.clear();
    qoff1_.assign(reads.size() + 1, 0);
    for (std::size_t r = 0; r < reads.size(); ++r) {
        qoff1_[r] = aln1_.size();
        std::size_t readLen = reads.length(r);
        for (auto &e : lists[r]) {
            const AlnRecord &a = db[e.rec];
            std::size_t c = e.tgt;
            // static filters of parseToCtg (Aligner.tcc:44-64)
            if (!ctgSelected_[c]) continue;
            std::size_t readBegin = a.queryBegin, readEnd = a.queryEnd;
            if ((readEnd - readBegin) * 1.0 / readLen < p.readToCtgRatio) continue;
            std::size_t ctgBegin = a.refBegin, ctgEnd = a.refEnd;
            std::size_t ctgLen = ctgs.length(c);
            if (ctgEnd >= ctgLen || ctgBegin >= ctgLen) continue;  // note: >= on the end (quirk Q14)

            bool isForward = a.forward;
            if (!isForward) flip(readBegin, readEnd, readLen);
            bool walkBack = false;
            if (!ctgForward_[c]) {  // the ii == 1 iteration (Aligner.tcc:73-96)
                isForward = !isForward;
                flip(readBegin, readEnd, readLen);
                flip(ctgBegin, ctgEnd, ctgLen);
                walkBack = true;
            }

            pag_aln o{};
            o.query = static_cast<std::uint32_t>(r);
            o.target = static_cast<std::uint32_t>(c);
            o.t_begin = fitU32(a.refBegin, "contig begin");
            o.t_end = fitU32(a.refEnd, "contig end");
            o.n_cols = a.nCols;
            o.diff_off = a.diffOff;
            o.flags = PAG_ALN_ELIGIBLE | (isForward ? 0u : PAG_ALN_REV_STRAND) | (walkBack ? PAG_ALN_WALK_BACK : 0u);
            std::size_t nValid = 0;
            o.q_start = PAG_NONE;  // (flipped) read begin outside the read: nothing is emitted
            if (readBegin < readLen) {
                nValid = std::min<std::size_t>(a.nEmit, readLen - readBegin);
                // bases whose contig coordinate falls off the contig have an empty list
                // (AlignReference::query, AlignReference.cpp:60-67): clip exactly, by walking
                if (ctgBegin + a.nRadv >= ctgLen) {
                    std::size_t k = 0, firstBad = nValid;
                    bool found = false;
                    db.exactAlign(a, readBegin, ctgBegin, !walkBack, [&](std::size_t, std::size_t t) {
                        if (!found && t >= ctgLen) {
                            firstBad = k;
                            found = true;
                        }
                        ++k;
                    });
                    nValid = std::min(nValid, firstBad);
                }
                o.q_start = fitU32(readBegin, "read begin");
                o.t_start = fitU32(ctgBegin, "contig begin");
            }
            o.n_valid = static_cast<std::uint32_t>(nValid);
            aln1_.push_back(o);
        }
    }
    qoff1_[reads.size()] = aln1_.size();
}

void GraphInput::buildPass2(const SeqDb &reads, const SeqDb &refs, const AlnDb &db, const BuildParams &p) {
    auto lists = mergeLists(db, reads, refs);
    aln2_.clear();
    qoff2_.assign(reads.size() + 1, 0);
    std::vector<bool> inList(db.size(), false);

    auto clampCov = [&](const AlnRecord &a, std::size_t refIdx, pag_aln &o) {
        // coverage loops run for (j = begin; j < end; ++j) { if (j >= size) break; ... }
        // (Aligner.cpp:76-81, Aligner.tcc:142-145): the effective interval is clamped to the sequence
        std::size_t size = refs.length(refIdx);
        std::size_t b = std::min(a.refBegin, size), e = std::min(a.refEnd, size);
        if (e < b) e = b;
        o.t_begin = static_cast<std::uint32_t>(b);
        o.t_end = static_cast<std::uint32_t>(e);
    };

    for (std::size_t r = 0; r < reads.size(); ++r) {
        qoff2_[r] = aln2_.size();
        std::size_t readLen = reads.length(r);
        for (auto &e : lists[r]) {
            const AlnRecord &a = db[e.rec];
            // static filters of parseToRef (Aligner.tcc:121-131); the coverage filter is dynamic
            if (!refAccepted_[e.tgt]) continue;
            std::size_t readBegin = a.queryBegin, readEnd = a.queryEnd;
            if ((readEnd - readBegin) * 1.0 / readLen < p.readToRefRatio) continue;
            bool isForward = a.forward;
            if (!isForward) flip(readBegin, readEnd, readLen);

            pag_aln o{};
            o.query = static_cast<std::uint32_t>(r);
            o.target = static_cast<std::uint32_t>(e.tgt);
            clampCov(a, e.tgt, o);
            o.n_cols = a.nCols;
            o.diff_off = a.diffOff;
            o.flags = PAG_ALN_ELIGIBLE | (isForward ? 0u : PAG_ALN_REV_STRAND);
            o.q_start = PAG_NONE;
            if (readBegin < readLen) {
                o.n_valid = static_cast<std::uint32_t>(std::min<std::size_t>(a.nEmit, readLen - readBegin));
                o.q_start = fitU32(readBegin, "read begin");
                o.t_start = fitU32(a.refBegin, "reference begin");
            }
            aln2_.push_back(o);
            inList[e.rec] = true;
        }
    }
    qoff2_[reads.size()] = aln2_.size();
    // every other record whose reference name is known still counts for coverage
    // (Aligner::covInfHelper ignores the query, Aligner.cpp:70-82)
    for (std::size_t i = 0; i < db.size(); ++i) {
        if (inList[i]) continue;
        const AlnRecord &a = db[i];
        if (!refs.contains(a.refName)) continue;
        pag_aln o{};
        o.query = PAG_NONE;
        o.target = static_cast<std::uint32_t>(refs.id(a.refName));
        clampCov(a, o.target, o);
        aln2_.push_back(o);
    }
}

GraphInput::GraphInput(const SeqDb &reads, const SeqDb &ctgs, const SeqDb &refs, const AlnDb &readToCtg,
                       const AlnDb &readToRef, const AlnDb &ctgToRef, const BlockConfig &cfg, const BuildParams &p)
    : ctgMapper_(ctgs), refMapper_(refs) {
    if (ctgMapper_.extraStart() >= 0xFFFFFFFFull || refMapper_.extraStart() >= 0xFFFFFFFFull)
        throw std::runtime_error(
            "coordinate space exceeds 32 bits (the reference truncates silently, PositionProcessor.cpp:48-51); "
            "split the input per reference sequence");
    if (reads.size() >= 0xFFFFFFFFull) throw std::runtime_error("too many reads");

    // filters exactly as the reference main sets them (pagraph.cpp:218-231)
    refAccepted_.assign(refs.size(), false);
    if (refs.contains(cfg.ref)) refAccepted_[refs.id(cfg.ref)] = true;
    ctgSelected_.assign(ctgs.size(), false);
    ctgForward_.assign(ctgs.size(), true);
    for (auto &c : cfg.contigs) {
        if (!ctgs.contains(c.first)) continue;
        ctgSelected_[ctgs.id(c.first)] = true;
        ctgForward_[ctgs.id(c.first)] = c.second;
    }

    buildCtgMap(ctgs, refs, ctgToRef);
    buildPass1(reads, ctgs, readToCtg, p);
    buildPass2(reads, refs, readToRef, p);

    refTab_.assign(refs.size(), pag_ref{});
    for (std::size_t i = 0; i < refs.size(); ++i) {
        refTab_[i].len = refs.length(i);
        refTab_[i].accepted = refAccepted_[i] ? 1u : 0u;
        refTab_[i].single_base = static_cast<std::uint32_t>(refMapper_.dualToSingle(static_cast<std::int64_t>(i) + 1, 0));
    }

    // canonical emission order: MultiThreadTools strided loops run thread-major when serialised
    // (MultiThreadTools.tcc:8-14): t = 0: 0, T, 2T, ...; t = 1: 1, T + 1, ...
    unsigned T = std::max(1u, p.threads);
    emitOrder_.clear();
    emitOrder_.reserve(reads.size());
    for (unsigned t = 0; t < T; ++t)
        for (std::size_t i = t; i < reads.size(); i += T) emitOrder_.push_back(static_cast<std::uint32_t>(i));

    in_ = pag_build_input{};
    in_.on_device = 0;
    in_.n_threads = p.threads;
    in_.reads.n_seqs = reads.size();
    in_.reads.byte_off = reads.byteOff().data();
    in_.reads.len = reads.lens().data();
    in_.reads.packed = reads.packed().data();
    in_.reads.packed_bytes = reads.packed().size();
    in_.emit_order = emitOrder_.data();
    in_.read_to_ctg = pag_aln_db{aln1_.size(), aln1_.data(), qoff1_.data(), readToCtg.diff().data(), readToCtg.diff().size()};
    in_.read_to_ref = pag_aln_db{aln2_.size(), aln2_.data(), qoff2_.data(), readToRef.diff().data(), readToRef.diff().size()};
    in_.n_ctgs = ctgTab_.size();
    in_.ctgs = ctgTab_.data();
    in_.ctg_ent_off = ctgEntOff_.data();
    in_.n_ctg_ent_off = ctgEntOff_.size();
    in_.ctg_ent = ctgEnt_.data();
    in_.n_ctg_ent = ctgEnt_.size();
    in_.n_refs = refTab_.size();
    in_.refs = refTab_.data();
    in_.eps = fitU32(p.epsilon, "epsilon");
    in_.cov_filter = fitU32(p.covFilter, "coverage filter");
    in_.outer_sample = static_cast<std::uint32_t>(p.outerSample);
    in_.topk_ctg = p.readToCtgTopK;
    in_.topk_ref = p.readToRefTopK;
}

}  // namespace pagh
