// WalkSession, second layer: a round of a contig — where its walk can be cut (checkpoints, segments, the leaping zone's pieces),
// the jobs it posts, resumed walks; delivery of a finished contig's path; the set-up steps of pag_travel (contig tables, first
// seeds, rings, arena, first rounds).
#pragma once

struct WalkRounds : WalkJobs {
    using WalkJobs::WalkJobs;


    // ---- start of a round of contig i: its seeds are in cs.seeds.  Decides where the walk can be cut, finds the checkpoint
    //      vertices and posts the seed jobs and the segment jobs.
    // stop coordinate of a job that walks up to segment q of the round (its checkpoint + the overlap)
    uint32_t first_stop(const stitch::RoundState &R) const { return stitch::stop_for(R, 0, seg_ov); }
    // The rounds of several contigs are prepared together: their checkpoint vertices come from ONE launch of k_checkpoints and
    // the id ranges around their segments from ONE launch of k_id_bounds (two synchronisations per call; contig by contig
    // the 48 first rounds of configs[1] were ~100 small launches and synchronisations, ~10 ms before the first job).
    struct RoundPlan {
        std::vector<uint32_t> ck_x;
        size_t n_spec_ck = 0;
        uint32_t x0 = 0xFFFFFFFFu, seed_lo = 0, seed_hi = 0;
        size_t req_off = 0, co_off = 0;
        bool has_co = false;
        bool kept = false;        // the round adopts the segments of an earlier round (RoundState::kept): none are planned
        uint32_t kept_stop = 0;   // ... and its seeds walk up to this coordinate (0: to the end)
    };
    // ---- start of a round of contigs `which` (their seeds are in cs.seeds): four steps
    // (1) per contig: the round's state, the checkpoint coordinates of its segments (reqs: the checkpoint searches)
    void plan_rounds(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, std::vector<TravSeedReq> &reqs) {
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            std::vector<uint32_t> &ck_x = P.ck_x;
            size_t &n_spec_ck = P.n_spec_ck;
            uint32_t &x0 = P.x0;
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            const bool keep = R.kept && !R.segs.empty();
            P.kept = keep;
            R.round += 1;
            R.active = true;
            if (!keep) {
                R.segs.clear();
                R.n_spec = 0;
                R.zone_end = 0;
                R.live_jobs = 0;
                R.slot_bufs = false;
            }
            R.chains.assign(cs.seeds.size(), Chain{});
            R.has_size = (uint64_t)cs.varLen;  // int64 -> size_t conversion as in the reference call
            rounds = std::max<uint64_t>(rounds, R.round);
            const uint64_t split = (uint64_t)(cs.len * startSplit);
            // where leaping becomes possible: hasSize + nowSize >= split, nowSize = k + the steps walked.  The steps follow the
            // contig coordinate closely but not exactly, so the zone is left with a margin; WHERE the walk is cut only decides how
            // much of it runs in parallel, every adoption is checked against the true sizes (try_merge).
            for (auto &sd : cs.seeds) x0 = std::min(x0, sd.ctg);
            if (keep) {
                // where the seeds' own walks stop: a little into the first kept segment ahead of them, of the kind a chain
                // at their coordinate adopts (advance_chain) — or of the other kind when none of that kind lies ahead
                const bool can = R.has_size + k >= split;
                const int n_spec = (int)R.n_spec, n_all = (int)R.segs.size();
                auto first_ahead = [&](int lo, int hi) -> int {
                    for (int q = lo; q < hi; ++q)
                        if (R.segs[(size_t)q].x > x0 && (R.segs[(size_t)q].leap || x0 < R.zone_end)) return q;
                    return -1;
                };
                int q = can ? first_ahead(n_spec, n_all) : first_ahead(0, n_spec);
                if (q < 0) q = can ? first_ahead(0, n_spec) : first_ahead(n_spec, n_all);
                P.kept_stop = q >= 0 ? stitch::stop_for(R, (size_t)q, seg_ov) : 0u;
            } else if (use_pieces && cs.varLen >= 0 && x0 >= cs.ctgLeft && x0 < cs.ctgRight) {
                const uint64_t H = (uint64_t)cs.varLen + k;
                // (measured at BASELINE configs[1]: segments of 10-20 kb with 1.5 kb of overlap are the optimum, a few thousand
                // jobs; shorter ones pay more overlap and job start-up, longer ones lengthen the first piece of every chain)
                const uint64_t seg_len = seg_len_env ? seg_len_env : 12000;
                // The two kinds of segments OVERLAP around the coordinate where leaping becomes possible (x0 + split - H if the steps
                // followed the coordinate exactly; they do not quite: `margin` on either side).  A chain adopts segments that
                // cannot leap up to where its true size allows (try_merge cuts the adoption there), crosses the point with a short
                // exact walk (TRAV_MODE_UNTIL_LEAP) and goes on with the pieces of the leaping zone that were started before the
                // point.  (Until round 3 the kinds were kept apart by the margins and every contig walked the ~10 kb between them
                // exactly, 30-48 ms at the end of its round.)  Only decides how much is walked in parallel: every adoption is
                // checked against the true sizes.
                const uint64_t margin = cfg.seg_safety_set ? cfg.seg_safety : cs.len / 400 + 200;
                if (split > H + seg_len) {
                    const uint64_t zone = std::min<uint64_t>((uint64_t)x0 + (split - H) + margin, (uint64_t)cs.ctgRight - 1);
                    for (uint64_t x = (uint64_t)x0 + seg_len; x + seg_ov + seg_len / 4 < zone; x += seg_len) ck_x.push_back((uint32_t)x);
                    if (!ck_x.empty()) R.zone_end = (uint32_t)zone;
                }
                n_spec_ck = ck_x.size();
                if (use_leap_pieces) {
                    // the leaping zone gets segments of its own (TRAV_MODE_LEAP), from where the real walk has certainly begun to
                    // leap (the steps follow the coordinate closely, not exactly: a margin; every adoption is checked with the true
                    // size) to the end of the strand
                    // (a walk there makes three times the classifications per vertex: shorter pieces for the same job length)
                    const uint64_t lseg = std::max<uint64_t>(seg_len / 2, seg_ov * 2);
                    // (how far before x0 + split - H the first piece starts: at configs[1] the steps of a path add up to 0.6 % more than
                    // the coordinates it covers — leaping begins ~7 kb earlier than the coordinate says on a 1.2 Mb contig; pieces
                    // started too early cost a few jobs, pieces started too late an exact walk on the contig's critical path)
                    const uint64_t left = cfg.seg_safety_set ? cfg.seg_safety : cs.len / 64 + 500;
                    const uint64_t first = (uint64_t)x0 + (split > H + left + lseg ? split - H - left : lseg);
                    // (the last stretch of the strand in shorter pieces still: the job that reaches the end of the strand is the
                    // last one of its round, and a contig that needs a second round waits for it twice)
                    const uint64_t end_div = 2;
                    const uint64_t end_zone = (uint64_t)cs.ctgRight > 2 * lseg ? (uint64_t)cs.ctgRight - 2 * lseg : 0;
                    for (uint64_t x = std::max<uint64_t>(first, (uint64_t)x0 + lseg); x + lseg / 4 / end_div < (uint64_t)cs.ctgRight - 1; x += (x >= end_zone ? std::max<uint64_t>(lseg / end_div, seg_ov) : lseg))
                        if (ck_x.size() == n_spec_ck || x > (uint64_t)ck_x.back() + lseg / 4 / end_div) ck_x.push_back((uint32_t)x);
                }
            }
            P.req_off = reqs.size();
            for (size_t q = 0; q < ck_x.size(); ++q) {
                const uint64_t off = ck_x[q] - cs.ctgLeft;
                TravSeedReq rq;
                rq.ctg = i;
                rq.pad = 0;
                rq.pos = off;
                rq.left = off - std::min<uint64_t>(off, 64);
                rq.right = off + 64;
                reqs.push_back(rq);
            }
        }
    }
    // (2) the checkpoint vertices of all of them: one launch, one round trip
    int find_checkpoints(const std::vector<TravSeedReq> &reqs, std::vector<uint32_t> &out) {
        out.assign(reqs.size() * 3, 0u);
        int r;
        if (!reqs.empty()) {
            if ((r = b_ckreq.alloc(reqs.size() * sizeof(TravSeedReq))) || (r = b_ckout.alloc(reqs.size() * 12))) return r;
            if ((r = upload_contigs())) return r;
            PAG_HIP_TRY(hipMemcpyAsync(b_ckreq.p, reqs.data(), reqs.size() * sizeof(TravSeedReq), hipMemcpyHostToDevice, s));
            trav_launch_checkpoints(G, b_tc.as<TravContig>(), b_ckreq.as<TravSeedReq>(), (uint32_t)reqs.size(), deviation, b_ckout.as<uint32_t>(), s);
            PAG_HIP_TRY(hipMemcpyAsync(out.data(), b_ckout.p, out.size() * 4, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
        return PAG_OK;
    }
    // (3) per contig: its segments, and the contig coordinates their id ranges are asked for (co)
    void make_segments(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, const std::vector<uint32_t> &out, std::vector<uint32_t> &co) {
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            const std::vector<uint32_t> &ck_x = P.ck_x;
            const size_t n_spec_ck = P.n_spec_ck;
            const uint32_t x0 = P.x0;
            if (P.kept) {  // (the id range around the seeds' own first piece: [lowest seed - 2000, its stop + 3000])
                if (P.kept_stop != 0u) {
                    P.has_co = true;
                    P.co_off = co.size();
                    co.push_back((uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)x0 - std::min<uint64_t>(x0, 2000)));
                    co.push_back((uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)P.kept_stop + 3000));
                }
                continue;
            }
            if (ck_x.empty()) continue;
            const uint32_t *out_c = out.data() + 3 * P.req_off;
            for (size_t q = 0; q < ck_x.size(); ++q) {
                if (out_c[3 * q] == PAG_NONE) continue;
                Seg sg;
                sg.x = out_c[3 * q + 1];
                sg.vid = out_c[3 * q];
                sg.leap = q >= n_spec_ck;
                sg.win_low = x0;
                sg.round = R.round;
                if (!R.segs.empty() && R.segs.back().leap == sg.leap && sg.x <= R.segs.back().x) continue;  // (increasing within a kind)
                R.segs.push_back(std::move(sg));
            }
            R.n_spec = 0;
            for (auto &sg : R.segs) R.n_spec += sg.leap ? 0 : 1;
            for (size_t q = 0; q < R.segs.size(); ++q) {
                const bool more = q + 1 < R.segs.size();
                if (R.segs[q].leap) R.segs[q].stop = more ? (uint32_t)std::min<uint64_t>((uint64_t)R.segs[q + 1].x + seg_ov, 0xFFFFFFFFull) : 0u;  // 0: to the end
                else R.segs[q].stop = more && !R.segs[q + 1].leap ? (uint32_t)std::min<uint64_t>((uint64_t)R.segs[q + 1].x + seg_ov, R.zone_end) : R.zone_end;
            }
            {
                bool any_spec = false;
                for (auto &sg : R.segs) any_spec = any_spec || !sg.leap;
                if (!any_spec) R.zone_end = 0;
            }
            if (!R.segs.empty()) {  // id ranges around the segments: [checkpoint - 2000, stop + 3000] in contig coordinates
                const size_t nq = R.segs.size();
                P.has_co = true;
                P.co_off = co.size();
                co.resize(co.size() + 2 * nq + 2);
                uint32_t *cc = co.data() + P.co_off;
                for (size_t q = 0; q < nq; ++q) {
                    cc[2 * q] = (uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)R.segs[q].x - std::min<uint64_t>(R.segs[q].x, 2000));
                    cc[2 * q + 1] = R.segs[q].stop ? (uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)R.segs[q].stop + 3000) : cs.ctgRight;
                }
                // ... and around the seeds' own first piece: [lowest seed - 2000, first stop + 3000]
                cc[2 * nq] = (uint32_t)std::max<uint64_t>(cs.ctgLeft, (uint64_t)x0 - std::min<uint64_t>(x0, 2000));
                cc[2 * nq + 1] = (uint32_t)std::min<uint64_t>(cs.ctgRight, (uint64_t)first_stop(R) + 3000);
            }
        }
    }
    int find_id_bounds(const std::vector<uint32_t> &co, std::vector<uint32_t> &ids) {
        ids.assign(co.size(), 0u);
        int r;
        if (!co.empty()) {
            if ((r = b_ckreq.alloc(co.size() * 4)) || (r = b_ckout.alloc(co.size() * 4))) return r;
            PAG_HIP_TRY(hipMemcpyAsync(b_ckreq.p, co.data(), co.size() * 4, hipMemcpyHostToDevice, s));
            trav_launch_id_bounds(G, b_ckreq.as<uint32_t>(), (uint32_t)co.size(), b_ckout.as<uint32_t>(), s);
            PAG_HIP_TRY(hipMemcpyAsync(ids.data(), b_ckout.p, ids.size() * 4, hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
        return PAG_OK;
    }
    // (4) per contig: the id ranges, the jobs
    int post_round_jobs(const std::vector<uint32_t> &which, std::vector<RoundPlan> &RP, const std::vector<uint32_t> &ids) {
        int r;
        for (size_t w = 0; w < which.size(); ++w) {
            const uint32_t i = which[w];
            RoundPlan &P = RP[w];
            CtgState &cs = st[i];
            RoundState &R = RS[i];
            uint32_t seed_lo = 0, seed_hi = 0;  // id range for the walks of the seeds up to the first checkpoint (0, 0: the strand)
            if (P.has_co) {
                const size_t nq = P.kept ? 0 : R.segs.size();
                const uint32_t *idc = ids.data() + P.co_off;
                auto window = [&](size_t q, uint32_t *wlo, uint32_t *whi) {
                    uint32_t lo = std::max(idc[2 * q], cs.inLo), hi = std::min(idc[2 * q + 1], cs.inHi);
                    lo = cs.inLo + ((lo - cs.inLo) & ~31u);  // (the strand's global-visited bitmap is read word-wise from here)
                    if (hi <= lo) hi = std::min<uint32_t>(cs.inHi, lo + 64);
                    *wlo = lo;
                    *whi = hi;
                };
                for (size_t q = 0; q < nq; ++q) window(q, &R.segs[q].win_lo, &R.segs[q].win_hi);
                if (P.kept || first_stop(R) != 0u) window(nq, &seed_lo, &seed_hi);
            }
            std::vector<JobPlan> plans;
            const uint64_t cap_full = cs.seqCap;
            for (size_t sd = 0; sd < cs.seeds.size(); ++sd) {
                const uint32_t stop = P.kept ? P.kept_stop : (R.segs.empty() ? 0u : first_stop(R));
                // (a seed's walk that stops at the first checkpoint is a piece like the segments: direct-mapped marks around it,
                // a sequence buffer for its stretch; the full-strand arrays, 130 MB per job at configs[1], are for resumed walks)
                JobPlan pl{0, (int)sd, cap_full, cs.seeds[sd].vid, 0u, stop, nullptr, false};
                if (stop != 0u && seed_hi != 0u && cs.seeds[sd].ctg <= stop) {
                    pl.win_lo = seed_lo;
                    pl.win_hi = seed_hi;
                    pl.cap = std::min<uint64_t>(cap_full, ((uint64_t)stop - cs.seeds[sd].ctg) / 2 + 8192);
                }
                plans.push_back(pl);
            }
            // (the segments of the leaping zone first: they are the slowest, three times the classifications per vertex)
            // ... and of those the piece that runs to the end of the strand FIRST: it walks on from there until it leaps (at
            // configs[1] ~5 000 vertices and 20 000 classifications where the other pieces have 1 700 and 6 000: 55-70 ms, the
            // longest job of its contig by far and the one its round waits for; tests/walk_trace.py showed a fifth of them
            // starting 12-14 ms into the walks)
            std::vector<size_t> seg_order;
            for (size_t q = R.segs.size(); q-- > 0;)
                if (R.segs[q].leap && R.segs[q].stop == 0u) {
                    seg_order.push_back(q);
                    break;
                }
            for (size_t q = 0; q < R.segs.size(); ++q)
                if (seg_order.empty() || q != seg_order[0]) seg_order.push_back(q);
            for (int pass = 0; pass < 2 && !P.kept; ++pass)  // (kept segments have their jobs, or their paths, already)
                for (size_t q : seg_order) {
                    if (R.segs[q].leap != (pass == 0)) continue;
                    const uint64_t spanc = (R.segs[q].stop ? (uint64_t)R.segs[q].stop : (uint64_t)cs.ctgRight) - R.segs[q].x;
                    const uint64_t cap = std::min<uint64_t>(cap_full, spanc / 2 + 8192);
                    JobPlan pl{1, (int)q, cap, R.segs[q].vid, (uint32_t)(R.segs[q].leap ? TRAV_MODE_LEAP : TRAV_MODE_SPEC), R.segs[q].stop, nullptr, false, R.segs[q].win_lo, R.segs[q].win_hi};
                    pl.win_low = R.segs[q].leap ? R.segs[q].win_low : 0u;
                    plans.push_back(pl);
                }
            if (wdebug)
                std::fprintf(stderr, "[walk] t=%.1f ms contig %u round %u: %zu seeds, %zu segments%s, cut zone ends at %u (strand %u..%u)\n", now_ms() - t_walk0, i,
                             R.round, cs.seeds.size(), R.segs.size(), P.kept ? " kept from an earlier round" : "", R.zone_end, cs.ctgLeft, cs.ctgRight);
            if ((r = post_batch(i, GRP_ROUND, plans))) return r;
        }
        return PAG_OK;
    }
    int start_rounds(const std::vector<uint32_t> &which) {
        std::vector<RoundPlan> RP(which.size());
        std::vector<TravSeedReq> reqs;
        std::vector<uint32_t> out, co, ids;
        int r;
        plan_rounds(which, RP, reqs);
        if ((r = find_checkpoints(reqs, out))) return r;
        make_segments(which, RP, out, co);
        if ((r = find_id_bounds(co, ids))) return r;
        if ((r = post_round_jobs(which, RP, ids))) return r;
        return flush_clears();  // (one launch for the buffers of all these rounds, under way while this thread goes on)
    }

    // continue chain c of contig i exactly: the path so far goes to the walker as a RESUME job
    // (until_leap: only as far as the first iteration boundary from which the walk can leap, TRAV_MODE_UNTIL_LEAP)
    int post_resume(uint32_t i, int c, uint32_t stop, bool until_leap = false) {
        CtgState &cs = st[i];
        Chain &ch = RS[i].chains[(size_t)c];
        const uint64_t cap = std::max<uint64_t>(cs.seqCap * ch.grow, ch.len + cs.seqCap / 4 + 4096);
        std::vector<JobPlan> plans{JobPlan{0, c, cap, cs.seeds[(size_t)c].vid, (uint32_t)(TRAV_MODE_RESUME | (until_leap ? TRAV_MODE_UNTIL_LEAP : 0)), stop, &ch, ch.exact}};
        return post_batch(i, GRP_CHAIN0 + c, plans);  // (its buffers are cleared with those of the other resumed walks of this turn: stitch_finished)
    }

    // adoption of a finished segment by a chain (conditions and their justification: walk_stitch.hpp)
    stitch::MergeCtx merge_ctx(uint32_t i) {
        MergeCtx M;
        M.k = k;
        M.deviation = deviation;
        M.split = (uint64_t)(st[i].len * startSplit);
        M.has_size = RS[i].has_size;
        M.round = RS[i].round;
        if (st[i].committed) {
            M.g_lo = st[i].gwinLo;
            M.g_hi = st[i].gwinHi;
            M.g_free_hi = st[i].gFreeHi;
        }
        return M;
    }
    // totals of what the chains adopted (walk_stitch.hpp advance_chain, called from stitch_finished)
    stitch::AdvanceStats adv_stats;

    int fail(int rc2) {
        shutdown_walker();
        g->defer_free = false;
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
        // deliveries made while the walks ran (gather kernels writing pinned chunks the next call reuses) must have landed,
        // and nothing of a failed call may be handed out as a path
        if (g->deliver_stream) hipStreamSynchronize(g->deliver_stream);
        hipStreamSynchronize(s);
        std::fill(g->path_valid.begin(), g->path_valid.end(), (uint8_t)0);
        std::fill(g->path_ptr.begin(), g->path_ptr.end(), nullptr);
        return rc2;
    }

    // filterSequence / "Pump it" of a finished contig (PAlgorithm.cpp:409-423)
    bool pumped(const CtgState &cs, uint32_t last_ctg) {  // the last vertex of a path that ends in a leap is dropped?
        auto d = mapper.singleToDual(last_ctg);
        uint64_t a = (uint64_t)std::llabs(d.first);
        return a == (uint64_t)cs.ci + 1 || (a >= 1 && a <= mapper.sizes.size() && (double)d.second >= (double)mapper.sizes[a - 1] * (1 - startSplit));
    }
    void filter_travel(CtgState &cs) {
        auto &seq = cs.travel;
        if (!cs.finalLeap) {
            const size_t windowSize = 10;
            if (seq.size() >= windowSize) {
                size_t startIdx = seq.size() - seq.size() / 90;
                for (size_t i = startIdx; i < seq.size() - windowSize + 1; ++i) {
                    uint32_t firstPos = seq[i].ctg;
                    uint32_t secondPos = seq[std::min(seq.size(), i + windowSize) - 1].ctg;
                    if (secondPos != 0 && firstPos != 0 && secondPos < firstPos) {
                        seq.resize(i + 1);
                        break;
                    }
                }
            }
        } else if (!seq.empty()) {
            if (pumped(cs, seq.back().ctg)) seq.pop_back();
        }
    }
    // A contig whose traversal is over is DELIVERED while the others still walk: its sequence is filtered, the full records of
    // its vertices are gathered on the device and copied (asynchronously, stream s) into pinned memory that lives until the
    // next call — at configs[1] the one gather + 380 MB copy for all contigs used to follow the last walk (15 ms).
    // Device buffers from the walk arena; without room there the contig is left to the epilogue.
    // (a delivery issued while walk jobs are live runs on 24 blocks: its thousands of waves, each with stores to host memory in
    // flight, slowed every walker wave beside them — 2.5 -> 3.2-5 us per classification in the last 40 ms of a block, round 5)
    static constexpr unsigned DELIVER_BLOCKS = 24;
    int deliver_contig(uint32_t i) {
        CtgState &cs = st[i];
        if (cs.delivered || !cs.done) return PAG_OK;
        if (cs.tail.on) {  // (a path that ends in a leap: finalLeap, nothing but the last vertex to filter)
            const CtgState::DevTail &T = cs.tail;
            const size_t m0 = T.m0, m = m0 + T.n - (pumped(cs, T.last_ctg) ? 1 : 0);
            const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
            cs.delivered = true;
            g->path_off[slot2] = 0;
            g->path_len[slot2] = m;
            g->path_valid[slot2] = 1;
            if (m == 0) return PAG_OK;
            pag_path_node *dst = (pag_path_node *)fetch_alloc(m * sizeof(pag_path_node));
            if (!dst) return PAG_ENOMEM;
            if (m0) {
                uint32_t *hp = (uint32_t *)fetch_alloc(m0 * 8);
                if (!hp) return PAG_ENOMEM;
                for (size_t x = 0; x < m0; ++x) {
                    hp[x] = cs.travel[x].u;
                    hp[m0 + x] = (uint32_t)cs.travel[x].step;
                }
                PAG_HIP_TRY(hipMemcpyAsync(T.d_ids, hp, m0 * 4, hipMemcpyHostToDevice, g->deliver_stream));
                PAG_HIP_TRY(hipMemcpyAsync(T.d_ids + T.cap, hp + m0, m0 * 4, hipMemcpyHostToDevice, g->deliver_stream));
            }
            trav_launch_gather_path(G, T.d_ids, T.d_ids + T.cap, m, dst, g->deliver_stream, n_live ? DELIVER_BLOCKS : 0u);
            g->path_ptr[slot2] = dst;
            return PAG_OK;
        }
        const size_t n = cs.travel.size();
        const size_t need = ((n * 8 + 255) & ~(size_t)255) + 512;
        if (!g->walk_arena || g->walk_arena_used + need > g->walk_arena_cap) return PAG_OK;
        filter_travel(cs);
        const size_t m = cs.travel.size();
        const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
        cs.delivered = true;
        g->path_off[slot2] = 0;
        g->path_len[slot2] = m;
        g->path_valid[slot2] = 1;
        if (m == 0) return PAG_OK;
        uint32_t *hp = (uint32_t *)fetch_alloc(m * 8);
        pag_path_node *dst = (pag_path_node *)fetch_alloc(m * sizeof(pag_path_node));
        if (!hp || !dst) return PAG_ENOMEM;
        for (size_t x = 0; x < m; ++x) {
            hp[x] = cs.travel[x].u;
            hp[m + x] = (uint32_t)cs.travel[x].step;
        }
        uint32_t *d_ids = (uint32_t *)((char *)g->walk_arena + g->walk_arena_used);
        g->walk_arena_used += (m * 8 + 255) & ~(size_t)255;
        // A stream of its own (behind this work on stream s the fetches of finished jobs would wait), and the gather kernel
        // writes the records straight into the pinned host array: a device-to-host copy of 32 bytes per vertex would
        // occupy the copy engine the fetches need (measured: their lap 9 -> 24 ms per step).
        if (!g->deliver_stream) PAG_HIP_TRY(hipStreamCreateWithFlags(&g->deliver_stream, hipStreamNonBlocking));
        PAG_HIP_TRY(hipMemcpyAsync(d_ids, hp, m * 8, hipMemcpyHostToDevice, g->deliver_stream));
        trav_launch_gather_path(G, d_ids, d_ids + m, m, dst, g->deliver_stream, n_live ? DELIVER_BLOCKS : 0u);
        g->path_ptr[slot2] = dst;
        return PAG_OK;
    }

    // ---- the event loop
    double t_progress = 0, t_first_fin = 0;
    // Waiting for the walker: a busy wait (pause instructions), not a sleep — on a loaded host a 20 us sleep comes back after
    // a millisecond or more, and every finished job that waits for this thread holds up the jobs that depend on it.  Only
    // after 5 ms without any news does the thread start yielding its time slice.
    double t_last_news = 0;
    void idle_wait(double us) {
        const double t0w = now_ms();
        if (t0w - t_last_news > 5.0) {
            std::this_thread::sleep_for(std::chrono::microseconds((long)us));
            return;
        }
        while ((now_ms() - t0w) * 1000.0 < us) {
            for (int q = 0; q < 32; ++q) __builtin_ia32_pause();
        }
    }
    // Contigs whose round is decided and not yet chosen / spliced / re-seeded.  While jobs are in flight they are taken a few at
    // a time, those that go on to another round first: the copy of a finished contig's walk (hundreds of thousands of vertices
    // out of pinned memory) keeps this thread — the one every chain waits for — away from the jobs that finish meanwhile; in
    // the last third of the walks, when the contigs that leapt finish in batches of a dozen, a job of a contig still walking
    // used to wait 10 - 15 ms for its turn.
    std::vector<uint32_t> over_queue;

    // ---- the steps of a call, in the order run() takes them
    // the traversal view: compact CSR, coordinate order, successor records (once per built graph)
    int begin() {
        PAG_HIP_TRY(hipSetDevice(g->device));
        s = g->stream;
        t_begin = now_ms();
        timing = cfg.timing;
        wdebug = cfg.walk_debug;
        wtrace = cfg.walk_trace;
        lap_t = t_begin;
        k = g->k;
        deviation = prm->deviation;
        errorRate = prm->error_rate;
        startSplit = prm->start_split;
        topK = std::min<uint32_t>(prm->ref_threads, 8u);
        int rc;
        if ((rc = trav_prepare_graph(g, ctgs->len, ctgs->n_seqs, ref_len, n_refs, deviation, errorRate, &G, &t_compact, orient, startSplit))) return rc;
        slot += TRAV_GRAPH_SLOTS + TRAV_EXTRA_SLOTS;
        lap("compact");
        return PAG_OK;
    }
    // contigs: packed bases, mapper tables, per-strand node tables, id ranges, global visited structures
    int setup_contigs() {
        int rc;
        n_ctgs = (uint32_t)ctgs->n_seqs;
        g->path_off.assign(2 * (size_t)n_ctgs, 0);
        g->path_len.assign(2 * (size_t)n_ctgs, 0);
        g->path_valid.assign(2 * (size_t)n_ctgs, 0);
        g->path_ptr.assign(2 * (size_t)n_ctgs, nullptr);
        // one entry per (contig, orientation): a contig selected with both orientations is two independent traversals
        // (PAssembly.cpp:28-36 walks every (name, forward) pair of its set)
        for (uint32_t c2 = 0; c2 < 2 * n_ctgs; ++c2) {
            const uint32_t c = c2 >> 1;
            const bool fwd = (c2 & 1u) == 0;
            const int32_t o = orient[c];
            if (!(o == PAG_ORIENT_BOTH || (fwd && o == PAG_ORIENT_FORWARD) || (!fwd && o == PAG_ORIENT_REVERSE))) continue;
            CtgState cs;
            cs.ci = c;
            cs.forward = fwd;
            cs.chosenOne = cs.forward ? (int64_t)c + 1 : -(int64_t)c - 1;
            cs.len = ctgs->len[c];
            cs.ctgLeft = (uint32_t)mapper.dualToSingle(cs.chosenOne, 0);
            cs.ctgRight = (uint32_t)mapper.dualToSingle(cs.chosenOne, cs.len);
            cs.revLeft = (uint32_t)mapper.dualToSingle(-cs.chosenOne, 0);
            cs.revRight = (uint32_t)mapper.dualToSingle(-cs.chosenOne, cs.len);
            cs.nodesOff = nodes_total;
            cs.seqCap = (uint64_t)cs.len / 2 + 8192;
            if (cfg.debug_seqcap) cs.seqCap = (uint64_t)cfg.debug_seqcap;  // tests: force the overflow / regrow path
            nodes_total += cs.len >= k ? cs.len - k + 1 : 0;
            st.push_back(std::move(cs));
        }
        n_sel = (uint32_t)st.size();
        if (n_sel == 0) return PAG_OK;

        b_packed = buf(), b_nodes = buf(), b_starts = buf(), b_sizes = buf(), b_tc = buf(), b_seedout = buf(), b_req = buf();
        b_gset = buf(), b_gather = buf(), b_vids = buf(), b_gbits = buf();
        if ((rc = b_packed.alloc(ctgs->packed_bytes + 64)) || (rc = b_nodes.alloc((nodes_total + 1) * 4)) ||
            (rc = b_starts.alloc(mapper.starts.size() * 8 + 8)) || (rc = b_sizes.alloc(mapper.sizes.size() * 8 + 8)) ||
            (rc = b_tc.alloc(n_sel * sizeof(TravContig))))
            return rc;
        PAG_HIP_TRY(hipMemcpyAsync(b_packed.p, ctgs->packed, ctgs->packed_bytes, hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(b_starts.p, mapper.starts.data(), mapper.starts.size() * 8, hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(b_sizes.p, mapper.sizes.data(), mapper.sizes.size() * 8, hipMemcpyHostToDevice, s));
        {   // the strands' node tables: one launch
            std::vector<TravCtgNodesJob> cj;
            uint32_t max_len = 0;
            for (auto &cs : st) {
                cj.push_back(TravCtgNodesJob{ctgs->byte_off[cs.ci], cs.nodesOff, (uint32_t)cs.len, cs.forward ? 1 : 0});
                max_len = std::max<uint32_t>(max_len, (uint32_t)cs.len);
            }
            DevBuf b_cj = buf();
            if ((rc = b_cj.alloc(cj.size() * sizeof(TravCtgNodesJob)))) return rc;
            PAG_HIP_TRY(hipMemcpyAsync(b_cj.p, cj.data(), cj.size() * sizeof(TravCtgNodesJob), hipMemcpyHostToDevice, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));  // (cj is a local)
            trav_launch_ctg_nodes(b_packed.as<uint8_t>(), b_cj.as<TravCtgNodesJob>(), (uint32_t)cj.size(), max_len, k, G, b_nodes.as<uint32_t>(), s);
        }

        tc.assign(n_sel, TravContig{});
        // id ranges of the strands, then the per-contig global visited structures
        {
            for (auto &cs : st) cs.gcap = 1024;  // placeholder so that gmask is well formed
            if ((rc = upload_contigs())) return rc;
            trav_launch_ranges(G, b_tc.as<TravContig>(), n_sel, s);
            PAG_HIP_TRY(hipMemcpyAsync(tc.data(), b_tc.p, n_sel * sizeof(TravContig), hipMemcpyDeviceToHost, s));
            PAG_HIP_TRY(hipStreamSynchronize(s));
            uint64_t tot_set = 0, tot_bits = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                CtgState &cs = st[i];
                cs.inLo = tc[i].in_lo;
                cs.inHi = tc[i].in_hi;
                cs.gcap = (uint32_t)pow2_at_least(cs.seqCap / 2 + 8192);
                tot_set += cs.gcap;
                tot_bits += ((uint64_t)(cs.inHi - cs.inLo) + 31) / 32 + 1;
            }
            if ((rc = b_gset.alloc(tot_set * 4)) || (rc = b_gbits.alloc(tot_bits * 4))) return rc;
            PAG_HIP_TRY(hipMemsetAsync(b_gset.p, 0xFF, tot_set * 4, s));
            PAG_HIP_TRY(hipMemsetAsync(b_gbits.p, 0, tot_bits * 4, s));
            uint64_t o1 = 0, o2 = 0;
            for (auto &cs : st) {
                cs.gset = b_gset.as<uint32_t>() + o1;
                o1 += cs.gcap;
                cs.gbits = b_gbits.as<uint32_t>() + o2;
                o2 += ((uint64_t)(cs.inHi - cs.inLo) + 31) / 32 + 1;
            }
        }

        lap("contig tables");
        return PAG_OK;
    }
    // round 0 seeds: searchPANode(onlyFirst) then top-K
    int first_seeds() {
        int rc;
        if ((rc = b_seedout.alloc((uint64_t)n_sel * SEED_STRIDE * 4))) return rc;
        if ((rc = upload_contigs())) return rc;
        trav_launch_seed_first(G, b_tc.as<TravContig>(), n_sel, deviation, b_seedout.as<uint32_t>(), SEED_STRIDE, s);
        std::vector<uint32_t> seedbuf((size_t)n_sel * SEED_STRIDE);
        PAG_HIP_TRY(hipMemcpyAsync(seedbuf.data(), b_seedout.p, seedbuf.size() * 4, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        {
            std::vector<uint32_t> vids;
            std::vector<size_t> cnt(n_sel);
            for (uint32_t i = 0; i < n_sel; ++i) {
                const uint32_t *o = &seedbuf[(size_t)i * SEED_STRIDE];
                size_t n = std::min<size_t>(std::min<size_t>(o[0], (SEED_STRIDE - 2) / 2), topK);
                cnt[i] = n;
                for (size_t j = 0; j < n; ++j) vids.push_back(o[1 + 2 * j]);
            }
            std::vector<pag_path_node> attrs;
            if ((rc = fetch_vertices(vids, attrs))) return rc;
            size_t at = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                st[i].seeds.assign(attrs.begin() + at, attrs.begin() + at + cnt[i]);
                at += cnt[i];
                if (st[i].seeds.empty()) st[i].done = true;
            }
        }

        lap("first seeds");
        return PAG_OK;
    }
    // the rings of job records (host memory the walker reads), the switches of the pieces
    int setup_rings() {
        if (g->cpool.size() < (size_t)n_sel * GROUPS * CB_N) g->cpool.resize((size_t)n_sel * GROUPS * CB_N);
        {
            const uint64_t sl = std::max<uint64_t>(128, cfg.seg_len ? cfg.seg_len : 12000);
            const uint64_t ll = std::max<uint64_t>(128, sl / 2);
            uint64_t est = 0;
            for (uint32_t i = 0; i < n_sel; ++i) est += (uint64_t)st[i].len / sl + (uint64_t)st[i].len / ll + 32;  // (every strand as if all of it were both zones)
            while (QCAP < 2 * est && QCAP < (1u << 24)) QCAP *= 2;
            if (cfg.debug_ring) QCAP = (uint32_t)cfg.debug_ring;  // tests: a ring far smaller than a round
        }
        const size_t q_need = 256 + NR * (size_t)QCAP * (sizeof(TravPosted) + sizeof(TravJobOut) + sizeof(uint32_t)) + 256;
        if (g->wq_bytes < q_need) {
            if (g->wq_host) hipHostFree(g->wq_host);
            g->wq_host = nullptr;
            g->wq_bytes = 0;
            PAG_HIP_TRY(hipHostMalloc(&g->wq_host, q_need, hipHostMallocCoherent | hipHostMallocMapped));
            g->wq_bytes = q_need;
        }
        if (!g->wq_next) PAG_HIP_TRY(hipMalloc((void **)&g->wq_next, 256));
        hq = (TravQueue *)g->wq_host;
        hjobs = (TravPosted *)((char *)g->wq_host + 256);
        houts = (TravJobOut *)(hjobs + NR * (size_t)QCAP);
        hdone = (uint32_t *)(houts + NR * (size_t)QCAP);
        std::memset(g->wq_host, 0, 256);
        std::memset(hdone, 0, NR * (size_t)QCAP * sizeof(uint32_t));
        PAG_HIP_TRY(hipMemsetAsync(g->wq_next, 0, 256, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        t_walk0 = now_ms();
        use_pieces = cfg.pieces;
        seg_len_env = cfg.seg_len;
        seg_ov = cfg.seg_overlap;
        force_exact = cfg.force_exact;
        RS.clear();
        RS.resize(n_sel);
        jref.assign(NR * (size_t)QCAP, JobRef{});
        for (auto &x : n_leap_refused) x = 0;
        use_leap_pieces = cfg.leap_pieces;
        deferred.clear();
        deferred.resize(n_sel);
        b_ckreq = buf(), b_ckout = buf();
        lap("rings");
        return PAG_OK;
    }
    // pinned staging + the walk arena
    int reserve_arena() {
        if (!pinned(64u << 20)) return PAG_ENOMEM;  // (grown later if a batch needs more)
        {   // the walk arena: sized for the first round of every contig (chain buffers over the whole strand + segment buffers)
            // plus half again for resumed walks and later rounds; at most 40 % of the free device memory; kept by the handle
            size_t want = 0;
            for (uint32_t i = 0; i < n_sel; ++i) {
                const CtgState &cs = st[i];
                const size_t span = (size_t)(cs.inHi - cs.inLo) + 8, cap = cs.seqCap, oc = pow2_at_least(cap / 4 + 4096);
                const size_t chain = cap * 8 + cap * 8 * TRAV_PROBE_GROUPS + oc * 8 * (1 + TRAV_PROBE_GROUPS) + span * 4 * (1 + TRAV_PROBE_GROUPS);
                const size_t n_seg = cs.len / 12000 + 1, scap = 8192 + 8192, soc = pow2_at_least(scap / 4 + 4096), sspan = span / (n_seg ? n_seg : 1) * 2 + 4096;
                const size_t seg = scap * 8 + scap * 8 * TRAV_PROBE_GROUPS + soc * 8 * (1 + TRAV_PROBE_GROUPS) + sspan * 4 * (1 + TRAV_PROBE_GROUPS);
                // segments of the leaping zone (the last tenth of the strand + margin, half as long, far larger hash sets, a log)
                const size_t n_lseg = cs.len / 8 / 6000 + 2, lcap = 3000 + 8192, loc = pow2_at_least(lcap + 8192), lspan = sspan;
                const size_t lseg = lcap * 8 + lcap * 8 * TRAV_PROBE_GROUPS + lcap * 8 + loc * 8 * (1 + TRAV_PROBE_GROUPS) + lspan * 4 * (1 + TRAV_PROBE_GROUPS);
                // (full-strand buffers: the resumed walks — the seeds' own first pieces are sized like segments)
                want += chain * 3 / 2 + (seg * (n_seg + 8) + lseg * n_lseg) * 3 / 2;
            }
            size_t free_b = 0, total_b = 0;
            const size_t sharers = env_device_sharers();
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) want = std::min(want, (free_b + g->walk_arena_cap) * 2 / 5 / sharers);
            if (g->walk_arena_cap < want / 10 * 7) {  // (an arena that is there — pag_reserve_walk_arena, an earlier call — is kept
                                                      // unless it is much too small: what does not fit goes to the slots)
                if (g->walk_arena) hipFree(g->walk_arena);
                g->walk_arena = nullptr;
                g->walk_arena_cap = 0;
                if (hipMalloc(&g->walk_arena, want) == hipSuccess) g->walk_arena_cap = want;
                else g->walk_arena = nullptr;  // (the slots do all the work then)
            }
            g->walk_arena_used = 0;
        }
        lap("arena");
        return PAG_OK;
    }
    // the first round of every contig is posted, the walker grid launched
    int post_first_rounds() {
        int rc;
        tw0 = now_ms();
        t_walk0 = tw0;
        g->defer_free = true;
        {   // longest contigs first: their exact tails (the leaping zone is a tenth of the contig) are the longest, so their
            // segments should be through the queue first
            std::vector<uint32_t> order(n_sel);
            for (uint32_t i = 0; i < n_sel; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a2, uint32_t b2) { return st[a2].len > st[b2].len; });
            // First rounds: the contigs' segment jobs enter the ring interleaved, a few per contig and turn (a contig's leap
            // segments first).  Posted contig by contig, the last contigs of the list finish their first round when the grid
            // runs empty — and those of them that need a second round (a re-seed after a walk that ended early) start it then:
            // every contig's first round now ends at about the same time, earlier than the last ones did.
            const uint32_t interleave = 16;  // (the share of the contig with the fewest jobs)
            defer_ring2 = true;
            {
                std::vector<uint32_t> first_rounds;
                for (uint32_t i : order)
                    if (!st[i].done) first_rounds.push_back(i);
                if ((rc = start_rounds(first_rounds))) return fail(rc);
            }
            lap("first rounds planned");
            defer_ring2 = false;
            {
                std::vector<size_t> at(n_sel, 0);
                // a contig's share of a turn (round 5): in proportion to the jobs it has, so that every contig's first round runs
                // out of the ring in the same turn.  With equal shares the contigs with the most segments — the longest ones, whose
                // chains also take the longest to stitch — saw their last segments START when the grid was already running empty
                // (configs[1]: at 58 of 105 ms), and the ones among them that need a second round started it last of all.
                std::vector<uint32_t> share(n_sel, interleave);
                {
                    size_t least = 0;
                    for (uint32_t i : order)
                        if (!deferred[i].empty() && (least == 0 || deferred[i].size() < least)) least = deferred[i].size();
                    const size_t turns = least ? (least + interleave - 1) / interleave : 1;
                    for (uint32_t i : order)
                        if (!deferred[i].empty()) share[i] = (uint32_t)std::max<double>(1.0, std::ceil((double)deferred[i].size() / (double)turns));
                }
                // (a turn of its own for the contigs' longest jobs — the piece that runs to the end of the strand, first in every
                // contig's list: they all start with the first wave of the grid)
                for (uint32_t i : order) {
                    auto &dq = deferred[i];
                    if (at[i] < dq.size() && (dq[0].P.J.mode & TRAV_MODE_LEAP) && dq[0].P.J.stop_pc == 0u) {
                        if ((rc = commit_job(2u, dq[0].P, dq[0].jr, dq[0].P.J.mode, dq[0].P.J.stop_pc))) return fail(rc);
                        at[i] = 1;
                    }
                }
                for (bool more = true; more;) {
                    more = false;
                    for (uint32_t i : order) {
                        auto &dq = deferred[i];
                        for (uint32_t c = 0; c < share[i] && at[i] < dq.size(); ++c, ++at[i])
                            if ((rc = commit_job(2u, dq[at[i]].P, dq[at[i]].jr, dq[at[i]].P.J.mode, dq[at[i]].P.J.stop_pc))) return fail(rc);
                        more = more || at[i] < dq.size();
                    }
                }
                for (auto &dq : deferred) std::vector<Deferred>().swap(dq);
            }
        }
        lap("ring order");
        if (n_live) {
            walkers.init(g, G, hjobs, houts, hdone, hq, QCAP, k);
            walkers.trace = wtrace;
            if ((rc = publish())) return fail(rc);  // (the jobs' buffers are ready, the rings are visible)
            lap("marks cleared, rings published");
            if ((rc = walkers.ensure(n_live))) {
                g->defer_free = false;
                return rc;
            }
            if (wdebug) std::fprintf(stderr, "[walk] %u walker waves launched (at most %u), %u + %u + %u jobs posted\n", walkers.launched, walkers.max_waves, n_posted[0], n_posted[1], n_posted[2]);
        } else {
            g->defer_free = false;
        }
        lap("round prep");
        return PAG_OK;
    }

    // ---- the event loop, step by step
};

