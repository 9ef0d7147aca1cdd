// WalkSession, third layer: the event loop of pag_travel — finished jobs, their paths, the chains that move on (walk_stitch.hpp),
// decided rounds (choose, splice, re-seed or deliver), the epilogue; run() is the whole traversal.
#pragma once

struct WalkSession : WalkRounds {
    using WalkRounds::WalkRounds;

    // jobs that have finished since the last look (again: nothing to do yet, look again)
    int poll_finished(std::vector<uint32_t> &fin, bool &again) {
        int rc;
        again = false;
        for (uint32_t ring = 0; ring < NR; ++ring) {
            // (at most QCAP jobs of a ring are in flight: a slot is taken again only when the job QCAP numbers earlier has been
            // handled — without this clamp a ring far smaller than a round would be scanned around more than once)
            if (n_posted[ring] > QCAP && scan_from[ring] < n_posted[ring] - QCAP) scan_from[ring] = n_posted[ring] - QCAP;
            while (scan_from[ring] < n_posted[ring] && !jref[ring * QCAP + scan_from[ring] % QCAP].live) ++scan_from[ring];
            for (uint32_t jn = scan_from[ring]; jn < n_posted[ring]; ++jn) {
                const uint32_t slot = ring * QCAP + jn % QCAP;
                if (jref[slot].live && __atomic_load_n(&hdone[slot], __ATOMIC_ACQUIRE) != 0) fin.push_back(slot);
            }
        }
        // A batch costs a kernel launch, a copy and a stream synchronisation (~0.1 ms of this thread): finished SEGMENT jobs
        // that no chain is waiting for are left to accumulate (up to 64 of them or 1 ms); a finished chain job, or a
        // segment some chain of its contig waits for, is fetched at once.
        if (!fin.empty()) {
            bool urgent = fin.size() >= 64 || (t_first_fin > 0 && now_ms() - t_first_fin > 1.0);
            for (size_t x = 0; x < fin.size() && !urgent; ++x) {
                const JobRef &jr = jref[fin[x]];
                if (jr.kind == 0) urgent = true;
                else
                    for (const Chain &ch : RS[jr.ctg].chains)
                        if (ch.waiting_seg == jr.idx) urgent = true;
            }
            if (t_first_fin == 0) t_first_fin = now_ms();
            if (!urgent) {
                idle_wait(20.0);
                again = true;
                return PAG_OK;
            }
            t_first_fin = 0;
        }
        if (fin.empty() && over_queue.empty()) {
            // waves that found nothing to do have left (k_walk_persistent): jobs that are outstanding get new ones
            if ((rc = walkers.ensure(n_live))) return fail(rc);
            const double idle_limit_ms = cfg.idle_limit_ms;
            if (now_ms() - t_progress > idle_limit_ms) {  // no job finished for a minute: give up instead of hanging
                uint32_t ticket[TRAV_RINGS] = {0, 0, 0};
                hipMemcpyAsync(ticket, g->wq_next, sizeof(ticket), hipMemcpyDeviceToHost, s);
                hipStreamSynchronize(s);
                set_error("pag_travel: no walk job finished within %.0f s (posted %u + %u + %u, claimed %u + %u + %u, jobs outstanding %u, walker waves started %u / left %u of %u launched)",
                          idle_limit_ms / 1000.0, n_posted[0], n_posted[1], n_posted[2], ticket[0], ticket[1], ticket[2], n_live, walkers.started(), walkers.exited(), walkers.launched);
                return fail(PAG_EFAULT);
            }
            idle_wait(30.0);
            again = true;
            return PAG_OK;
        }
        t_progress = now_ms();
        t_last_news = t_progress;
        lap("walk");
        return PAG_OK;
    }
    // ---- their paths: vertices, steps and contig coordinates to the host (one round trip for the batch)
    struct Got {
        uint32_t jn;
        uint64_t from, len, off;       // the part of the sequence that is new; word offset of its packed words (trav_pack_words)
        const uint32_t *v, *s, *pc;     // ... in pinned memory that lives as long as this call (fetch_alloc)
        const uint32_t *xl = nullptr, *xh = nullptr;  // TRAV_MODE_LEAP: low / high words of the iteration log
        const uint32_t *agg = nullptr, *xagg = nullptr;  // block tables of those arrays (walk_stitch.hpp; written by k_pack_paths)
    };
    int fetch_paths(const std::vector<uint32_t> &fin, std::vector<Got> &got) {
        static_assert(AGG_BLOCK == 64 && AGG_WORDS == 5 && AGG_XWORDS == 2, "k_pack_paths writes these tables");
        got.assign(fin.size(), Got{});
        {
            uint64_t tot = 0, max_len = 0;
            std::vector<TravPackDesc> descs(fin.size());
            for (size_t x = 0; x < fin.size(); ++x) {
                const uint32_t slot = fin[x];
                const TravJobOut &o = houts[slot];
                const TravJob &J = hjobs[slot].J;
                Got &G2 = got[x];
                G2.jn = fin[x];
                G2.from = std::min<uint64_t>(jref[slot].init_len, o.seq_len);
                G2.len = is_orphan(jref[slot]) ? 0 : o.seq_len - G2.from;  // (nobody reads an orphan's path)
                G2.off = tot;
                tot += trav_pack_words(G2.len, J.seq_x != nullptr);
                max_len = std::max(max_len, G2.len);
                descs[x] = TravPackDesc{J.seq_v + G2.from, J.seq_s + G2.from, G2.len, G2.off, J.seq_x ? J.seq_x + G2.from : nullptr};
            }
            uint32_t *hp = (uint32_t *)fetch_alloc(tot * 4 + fin.size() * sizeof(TravPackDesc) + 256);
            if (!hp) return fail(PAG_ENOMEM);
            TravPackDesc *hd = (TravPackDesc *)(hp + ((tot + 3) & ~3ull));
            std::memcpy(hd, descs.data(), descs.size() * sizeof(TravPackDesc));
            // the pack kernel reads its descriptors from, and writes the packed paths to, the pinned host memory directly: one
            // launch + one synchronisation per batch instead of copy + launch + copy + synchronisation (every call of this
            // thread is on the critical path of some chain)
            trav_launch_pack_paths(G, hd, (uint32_t)descs.size(), max_len, hp, s);
            if (hipStreamSynchronize(s) != hipSuccess) {
                set_error("pag_travel: stream failure while fetching paths");
                return fail(PAG_EFAULT);
            }
            const bool check_aggs = cfg.check_aggs;
            for (Got &G2 : got) {
                G2.v = hp + G2.off;
                G2.s = G2.v + G2.len;
                G2.pc = G2.s + G2.len;
                G2.agg = G2.pc + G2.len;
                if (hjobs[G2.jn].J.seq_x) {
                    G2.xl = G2.pc + G2.len;
                    G2.xh = G2.xl + G2.len;
                    G2.agg = G2.xh + G2.len;
                    G2.xagg = G2.agg + agg_blocks((size_t)G2.len) * AGG_WORDS;
                }
                if (check_aggs) {  // (tests: the device's block tables against the host's definition of them)
                    std::vector<uint32_t> want(agg_blocks((size_t)G2.len) * AGG_WORDS), wantx(agg_blocks((size_t)G2.len) * AGG_XWORDS);
                    build_block_aggs(G2.v, G2.s, G2.pc, (size_t)G2.len, want.data());
                    bool same = std::memcmp(want.data(), G2.agg, want.size() * 4) == 0;
                    if (G2.xagg) {
                        build_block_xaggs(G2.xl, G2.xh, (size_t)G2.len, wantx.data());
                        same = same && std::memcmp(wantx.data(), G2.xagg, wantx.size() * 4) == 0;
                    }
                    if (!same) {
                        set_error("pag_travel: block tables of a fetched path differ from their definition (job %u, %llu entries)", G2.jn, (unsigned long long)G2.len);
                        return fail(PAG_EFAULT);
                    }
                }
            }
        }
        lap("fetch");
        return PAG_OK;
    }
    // bookkeeping of the finished jobs, their paths into segments and chains, the chains move on; touched: the contigs with news
    int stitch_finished(std::vector<Got> &got, std::vector<uint32_t> &touched) {
        int rc;
        const double ts0 = now_ms();
        // serial part: bookkeeping, and the (rare) jobs that have to be posted again
        std::vector<size_t> heavy;  // items of `got` whose path has to be copied / indexed
        for (size_t gx = 0; gx < got.size(); ++gx) {
            Got &G2 = got[gx];
            const uint32_t slot = G2.jn;
            JobRef &jr = jref[slot];
            const TravJobOut o = houts[slot];
            const uint32_t i = jr.ctg;
            RoundState &R = RS[i];
            if (is_orphan(jr)) {  // a segment job of a round that is over: its slot is free again, nothing else
                jr.live = false;
                n_live -= 1;
                continue;
            }
            jr.live = false;
            n_live -= 1;
            R.live_jobs -= 1;
            steps_total += o.seq_len - G2.from;
            classify_total += o.n_classify;
            probe_total += o.n_probe;
            record_total += o.n_records;
            const bool overflow = (o.overflow & 3) != 0, misspec = (o.overflow & 4) != 0;
            if (o.poison) {
                // (a regional graph, pag_shard_select: the walk reached a vertex whose successors another rank holds)
                set_error(g->regional ? "pag_travel: a walk of contig %u left the region of the graph this rank holds (reference band halo too small: raise PAG_SHARD_HALO)"
                                      : "pag_travel: a walk of contig %u left the view built for this handle's traversals (PAG_VIEW_HALO / PAG_VIEW_MARGIN)",
                          st[i].ci);
                return fail(PAG_ERANGE);
            }
            touched.push_back(i);
            if (jr.kind == 1) {  // a segment
                Seg &sg = R.segs[(size_t)jr.idx];
                sg.usable = !overflow && !misspec && G2.len >= 8;
                sg.stopped = o.stopped != 0;
                if (sg.usable) heavy.push_back(gx);
                else sg.done = true;
                continue;
            }
            Chain &ch = R.chains[(size_t)jr.idx];
            ch.job = -1;
            if (misspec && !overflow) {  // a zombie probe leapt: the job is walked again, every probe to its end
                ch.exact = true;
                ++respeculated;
                if (wdebug) std::fprintf(stderr, "[walk] contig %u chain %d: speculation failed, exact walk\n", i, jr.idx);
            } else if (overflow) {
                if (misspec) ch.exact = true;
                if (ch.grow >= (1u << 24)) {  // (the buffers double until the walk fits; a device allocation that fails reports itself)
                    set_error("pag_travel: walker buffers overflow at %u times their first size", ch.grow);
                    return fail(PAG_ENOMEM);
                }
                ch.grow *= 2;
            }
            if (misspec || overflow) {  // the same job again (its path so far, if any, is still on the host)
                std::vector<JobPlan> plans;
                const uint64_t cap = std::max<uint64_t>(st[i].seqCap * ch.grow, ch.len + st[i].seqCap / 4 + 4096);
                plans.push_back(JobPlan{0, jr.idx, cap, st[i].seeds[(size_t)jr.idx].vid, ch.job_mode, ch.job_stop, (ch.job_mode & TRAV_MODE_RESUME) ? &ch : nullptr, ch.exact});
                if ((rc = post_batch(i, GRP_CHAIN0 + jr.idx, plans))) return fail(rc);
                continue;
            }
            if (!o.stopped) ch.final = true;
            heavy.push_back(gx);
        }
        const double ts1 = now_ms();
        // parallel part: the paths of the finished jobs (a job belongs to one segment or one chain: the items are independent)
        for (size_t hx = 0; hx < heavy.size(); ++hx) {
            Got &G2 = got[heavy[hx]];
            const uint32_t slot = G2.jn;
            const JobRef &jr = jref[slot];
            const TravJobOut o = houts[slot];
            RoundState &R = RS[jr.ctg];
            if (jr.kind != 1) {  // the new part of a chain's path
                extend_chain(R.chains[(size_t)jr.idx], G2.v, G2.s, G2.pc, (size_t)G2.len, nullptr, G2.agg, 0, hjobs[slot].J.seq_v + G2.from, hjobs[slot].J.seq_s + G2.from);
                continue;
            }
            Seg &sg = R.segs[(size_t)jr.idx];
            sg.P.v = G2.v;
            sg.P.s = G2.s;
            sg.P.pc = G2.pc;
            sg.P.xl = G2.xl;
            sg.P.xh = G2.xh;
            sg.P.n = (size_t)G2.len;
            sg.P.agg = G2.agg;
            sg.P.xagg = G2.xagg;
            sg.P.dv = hjobs[slot].J.seq_v + G2.from;
            sg.P.ds = hjobs[slot].J.seq_s + G2.from;
            // (a coordinate-free vertex cannot happen while leaping is off; never adopt such a path)
            if (!sg.leap && range_agg(sg.P.v, sg.P.s, sg.P.pc, sg.P.agg, 0, sg.P.n).lo_all == 0u) sg.usable = false;
            if (sg.leap) {
                sg.usable = sg.usable && G2.xl != nullptr;
                sg.wd_below_max = o.wd_below_max;
                sg.wd_forced_min = o.wd_forced_min;
            }
            sg.max_back = o.max_back;
            sg.max_chosen = std::max<uint32_t>(o.max_chosen, 1u);
            sg.max_probe = o.max_probe;
            sg.done = true;
        }
        const double ts2 = now_ms();
        if (wtrace)
            for (Got &G2 : got) {
                const JobRef &jr = jref[G2.jn];
                const TravJobOut &o = houts[G2.jn];
                trace.push_back(TraceEv{ts2 - tw0, 0u, jr.ctg, (int32_t)jr.kind, (int32_t)jr.idx, o.t_begin, o.t_end, G2.len, o.n_classify});
            }
        if (wdebug)
            for (Got &G2 : got) {
                const uint32_t slot = G2.jn;
                const JobRef &jr = jref[slot];
                const TravJobOut o = houts[slot];
                const uint32_t i = jr.ctg;
                if (jr.kind == 1) {
                    const Seg &sg = RS[i].segs[(size_t)jr.idx];
                    std::fprintf(stderr, "[walk] t=%.1f ms dev %.3f..%.3f contig %u segment %d done: %llu vertices, %s, %s (flags %d, outside %llu, classify %llu, back %u, chosen %u, probe %llu)\n", now_ms() - tw0,
                                 (double)(o.t_begin % 100000000000ull) * 1e-5, (double)(o.t_end % 100000000000ull) * 1e-5, i, jr.idx, (unsigned long long)o.seq_len, sg.stopped ? "stopped" : "ended", sg.usable ? "usable" : "NOT usable", o.overflow, (unsigned long long)o.n_out, (unsigned long long)o.n_classify,
                                 o.max_back, o.max_chosen, (unsigned long long)o.max_probe);
                } else if ((o.overflow & 7) == 0) {
                    std::fprintf(stderr, "[walk] t=%.1f ms dev %.3f..%.3f contig %u chain %d job done: +%llu vertices (%zu), %s (classify %llu)\n", now_ms() - tw0,
                                 (double)(o.t_begin % 100000000000ull) * 1e-5, (double)(o.t_end % 100000000000ull) * 1e-5, i, jr.idx, (unsigned long long)G2.len,
                                 RS[i].chains[(size_t)jr.idx].len, o.stopped ? "stopped" : "ended", (unsigned long long)o.n_classify);
#ifdef PAG_WALK_PROF
                    // (make WALK_PROF=1: 100 MHz ticks and counts per section of the job — 0 append, 1 classification, 2 wait for a
                    // free slot, 3 slot setup, 4 slot steps, 5 choice, 6 window refills, 7 whole-wave probes, 8-11 inside a slot step:
                    // window, evaluation, class minimum + shuffles, state update, 12 job setup: filters + the contig's global set)
                    std::fprintf(stderr, "[walk]    prof mode %u main %llu fills %llu probes %llu:", hjobs[slot].J.mode, (unsigned long long)o.n_main,
                                 (unsigned long long)(o.n_fill & 0xFFFFFFFFull), (unsigned long long)o.n_probe);
                    for (int q = 0; q < 14; ++q) std::fprintf(stderr, " [%d] %.2f ms / %u", q, (double)o.prof_t[q] * 1e-5, o.prof_c[q]);
                    std::fprintf(stderr, "\n");
#endif
                }
            }
        std::sort(touched.begin(), touched.end());
        touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
        // the chains of the touched contigs move on: adoptions (advance_chain reads the round's segments and writes its chain
        // only: the chains are independent, those of one contig too), then the resumed walks are posted by this thread.  Host
        // threads since round 5: when the device runs out of first-round work, ~900 segments finish within 15 ms and their
        // adoptions — 5 us each, 24 ms per block — were what the control thread was busy with while finished chain jobs waited
        // (tests/walk_trace.py: laps of 3-4 ms per loop iteration in the last 30 ms of the walks).
        {
            struct AdvTask {
                uint32_t i;
                int c;
                Next nx;
            };
            std::vector<AdvTask> tasks;
            for (uint32_t i : touched) {
                RoundState &R = RS[i];
                for (size_t c = 0; c < R.chains.size(); ++c) {
                    const Chain &ch = R.chains[c];
                    if (ch.final || ch.job >= 0) continue;
                    if (ch.waiting_seg >= 0 && !R.segs[(size_t)ch.waiting_seg].done) continue;
                    tasks.push_back(AdvTask{i, (int)c, Next{}});
                }
            }
            const unsigned nthr = (unsigned)std::min<size_t>(tasks.size() / 2, cfg.stitch_threads);
            const uint64_t fails_before = adv_stats.merge_fail;
            auto run = [&](std::atomic<size_t> &next, stitch::AdvanceStats &S) {
                for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1))
                    tasks[t].nx = advance_chain(RS[tasks[t].i], RS[tasks[t].i].chains[(size_t)tasks[t].c], merge_ctx(tasks[t].i), seg_ov, S);
            };
            std::atomic<size_t> next{0};
            if (nthr <= 1) {
                run(next, adv_stats);
            } else {
                std::vector<stitch::AdvanceStats> part(nthr);
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < nthr; ++t) pool.emplace_back([&, t]() { run(next, part[t]); });
                run(next, part[0]);
                for (auto &th : pool) th.join();
                for (const stitch::AdvanceStats &S : part) {
                    adv_stats.adopted += S.adopted;
                    adv_stats.leap_adopted += S.leap_adopted;
                    adv_stats.merge_fail += S.merge_fail;
                    for (int w = 0; w < 8; ++w) adv_stats.leap_refused[w] += S.leap_refused[w];
                }
            }
            n_adopted = adv_stats.adopted;
            n_leap_adopted = adv_stats.leap_adopted;
            n_merge_fail = adv_stats.merge_fail;
            for (int w = 0; w < 8; ++w) n_leap_refused[w] = adv_stats.leap_refused[w];
            if (wdebug && adv_stats.merge_fail != fails_before)
                std::fprintf(stderr, "[walk] %llu segments were not adoptable, walking on exactly\n", (unsigned long long)(adv_stats.merge_fail - fails_before));
            for (const AdvTask &T : tasks)
                if (T.nx.what == Next::Resume && (rc = post_resume(T.i, T.c, T.nx.stop, T.nx.until_leap))) return fail(rc);
            if ((rc = flush_clears())) return fail(rc);
        }
        t_st[0] += ts1 - ts0;
        t_st[1] += ts2 - ts1;
        t_st[2] += now_ms() - ts2;
        lap("stitch");
        return PAG_OK;
    }
    // contigs whose chains are all final: the round is decided; batch: those taken now
    void decide_rounds(const std::vector<uint32_t> &touched, std::vector<uint32_t> &batch) {
        for (uint32_t i : touched) {
            RoundState &R = RS[i];
            if (!R.active) continue;
            bool all = true;
            for (auto &ch : R.chains) all = all && ch.final;
            // (segment jobs still waiting or walking stay with the contig or become orphans — unless the round's buffers are
            // per-contig slots, which the next round takes over: such a round waits for them)
            if (all && (R.live_jobs == 0 || !R.slot_bufs) && std::find(over_queue.begin(), over_queue.end(), i) == over_queue.end())
                over_queue.push_back(i);
        }
        {
            auto leaps = [&](uint32_t i) {  // the round ended on another contig: the contig is finished (PAlgorithm.cpp:254-262, 322-328)
                for (const Chain &ch : RS[i].chains) {
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    if (last_ctg != 0 && mapper.singleToDual(last_ctg).first != st[i].chosenOne) return true;
                }
                return false;
            };
            const size_t take = over_queue.size();
            std::stable_partition(over_queue.begin(), over_queue.end(), [&](uint32_t i) { return !leaps(i); });
            batch.assign(over_queue.begin(), over_queue.begin() + (long)take);
            over_queue.erase(over_queue.begin(), over_queue.begin() + (long)take);
        }
    }
    // ---- per contig: choose (PAlgorithm.cpp:238-262); the chosen walks are uploaded, gathered and committed on the
    //      device back to back, copied out, and spliced by a pool of host threads (contigs are independent)
    struct Pick {
        int chosen = -1;
        bool leap = false;
        size_t chooseCtgPos = 0, chooseRefPos = 0;
        uint64_t off = 0, len = 0;
    };
    int take_walks(const std::vector<uint32_t> &batch, std::vector<Pick> &picks) {
        int rc;
        {
            uint64_t tot = 0;
            for (uint32_t i : batch) {
                CtgState &cs = st[i];
                RoundState &R = RS[i];
                R.active = false;
                Pick &P = picks[i];
                size_t maxLen = 0;
                for (size_t sd = 0; sd < R.chains.size(); ++sd) {
                    const Chain &ch = R.chains[sd];
                    const size_t len = ch.size;
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    P.leap = last_ctg != 0 && mapper.singleToDual(last_ctg).first != cs.chosenOne;
                    if (!P.leap && sd > 0 && prm->min_len > 0 && len < prm->min_len) continue;
                    if (len > maxLen || P.leap) {
                        maxLen = len;
                        P.chosen = (int)sd;
                        P.chooseCtgPos = (size_t)mapper.singleToDual(cs.seeds[sd].ctg).second;
                        P.chooseRefPos = (size_t)refMapper.singleToDual(cs.seeds[sd].ref).second;
                        if (P.leap) break;
                    }
                }
                if (P.chosen >= 0) {
                    P.len = R.chains[(size_t)P.chosen].len;
                    tot += P.len;  // (an upper bound: walks that stay on the device take no room, see below)
                }
                if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 2u, i, 0, P.chosen, R.round, P.leap ? 1u : 0u, P.chosen >= 0 ? R.chains[(size_t)P.chosen].len : 0, 0});
                if (wdebug && P.chosen >= 0) {
                    const Chain &ch = R.chains[(size_t)P.chosen];
                    const uint32_t last_ctg = ch.len == 0 ? 0u : ch.parts.back().pc[ch.parts.back().n - 1];
                    std::fprintf(stderr, "[walk] t=%.1f ms contig %u round %llu over: chain %d of %zu chosen, %llu vertices, size %llu, from offset %lld, ends at offset %lld (strand %u), mx %u%s\n",
                                 now_ms() - t_begin, i, (unsigned long long)R.round, P.chosen, R.chains.size(), (unsigned long long)ch.len, (unsigned long long)ch.size,
                                 (long long)cs.seeds[(size_t)P.chosen].ctg - (long long)cs.ctgLeft, last_ctg ? (long long)last_ctg - (long long)cs.ctgLeft : -1ll, cs.len,
                                 ch.mx_all >= cs.ctgLeft ? ch.mx_all - cs.ctgLeft : 0u, P.leap ? ", leap" : "");
                }
            }
            // the chosen walks go to the device through the pinned staging area (vertex ids only: the commit kernels need
            // nothing else), one copy for the batch
            uint32_t *hp = (uint32_t *)pinned(tot * 4 + 256);
            if (!hp) return fail(PAG_ENOMEM);
            if ((rc = b_gather.alloc(tot * 4 + 64))) return fail(rc);
            // ONE pass over the chosen walk of every contig of the batch (its parts lie where the fetches put them): the
            // vertex ids for the device, the walk appended to the contig's running path (appendSeq, PAlgorithm.cpp:110-142),
            // the coordinate window of the global table, the vertices outside the strand's id range.  (Five passes and two
            // copies of the walk before: 14 M path vertices per block at configs[1], on the thread every contig waits for.)
            // The chosen chains' parts are copied to the contigs' paths (cs.travel) and to the id list of the commit in chunks,
            // by a small pool of threads: a round of a long contig is millions of vertices in a handful of parts, and the
            // entries are in pinned memory the device wrote (first read = DRAM latency).  What the loop used to add up on
            // the way — the steps, the coordinate window — the chain knows already.
            struct CopyChunk {
                uint32_t i;            // contig
                const Chain::Part *pt;
                size_t x0, x1;         // entries of the part
                LNode *dst;            // of the part's first entry
                uint32_t *ids;
                std::vector<uint32_t> outside;  // vertices outside the strand's id range, in order
            };
            std::vector<CopyChunk> chunks;
            const size_t CHUNK = 1u << 17;
            uint64_t used = 0;
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0) continue;
                const Chain &ch = RS[i].chains[(size_t)P.chosen];
                CtgState &cs = st[i];
                std::vector<LNode> &base = cs.travel;
                int64_t dLen = 0;
                const uint32_t head_ctg = ch.parts.front().pc[0];
                int32_t dist = (int32_t)k;
                while (!base.empty() && (base.back().ctg == 0 || head_ctg <= base.back().ctg)) {
                    dLen -= base.back().step;
                    base.pop_back();
                }
                if (!base.empty()) dist = (int32_t)(head_ctg - base.back().ctg);
                const size_t at0 = base.size();
                if (P.leap && !RS[i].slot_bufs && g->walk_arena) {
                    // the contig is finished by this walk (splice below): nothing of it is needed on the host
                    bool on_dev = true;
                    for (const Chain::Part &pt : ch.parts) on_dev = on_dev && pt.dv && pt.ds;
                    const size_t cap = at0 + P.len, need = (cap * 8 + 255) & ~(size_t)255;
                    TravConcatPart *cp = on_dev && g->walk_arena_used + need <= g->walk_arena_cap ? (TravConcatPart *)fetch_alloc(ch.parts.size() * sizeof(TravConcatPart)) : nullptr;
                    if (cp) {
                        CtgState::DevTail &T = cs.tail;
                        T.on = true;
                        T.d_ids = (uint32_t *)((char *)g->walk_arena + g->walk_arena_used);
                        g->walk_arena_used += need;
                        T.cap = cap;
                        T.m0 = at0;
                        T.n = P.len;
                        T.last_ctg = ch.parts.back().pc[ch.parts.back().n - 1];
                        for (size_t x = 0; x < ch.parts.size(); ++x) cp[x] = TravConcatPart{ch.parts[x].dv, ch.parts[x].ds, ch.parts[x].start, ch.parts[x].n};
                        if (!g->deliver_stream && hipStreamCreateWithFlags(&g->deliver_stream, hipStreamNonBlocking) != hipSuccess) return fail(PAG_EFAULT);
                        trav_launch_concat_parts(cp, (uint32_t)ch.parts.size(), T.d_ids + at0, T.d_ids + cap + at0, (uint32_t)dist, g->deliver_stream);
                        continue;
                    }
                }
                base.resize(at0 + P.len);
                LNode *dst = base.data() + at0;
                picks[i].off = used;
                used += P.len;
                uint32_t *ids = hp + P.off;
                for (const Chain::Part &pt : ch.parts)
                    for (size_t x0 = 0; x0 < pt.n; x0 += CHUNK)
                        chunks.push_back(CopyChunk{i, &pt, x0, std::min(pt.n, x0 + CHUNK), dst + pt.start, ids + pt.start, {}});
                dLen += (int64_t)ch.size;
                if (ch.low_nz != 0xFFFFFFFFu) {  // (some vertex has a coordinate)
                    cs.gwinLo = std::min(cs.gwinLo, ch.low_nz);
                    cs.gwinHi = std::max(cs.gwinHi, ch.mx_all);
                }
                cs.gFreeHi = std::max(cs.gFreeHi, ch.m0_all);
                // the first vertex of the round's path: its step is the distance to the path so far (set after the copy)
                cs.varLen += dLen - ((int64_t)ch.parts.front().s[0] - dist);
                cs.pendingFirst = at0;
                cs.pendingFirstStep = dist;
            }
            {
                std::atomic<size_t> nxt{0};
                auto worker = [&]() {
                    for (size_t c; (c = nxt.fetch_add(1)) < chunks.size();) {
                        CopyChunk &C = chunks[c];
                        const Chain::Part &pt = *C.pt;
                        const uint32_t in_lo = st[C.i].inLo, in_hi = st[C.i].inHi;
                        for (size_t x = C.x0; x < C.x1; ++x) {
                            const uint32_t v = pt.v[x];
                            C.dst[x] = LNode(v, (int32_t)pt.s[x], pt.pc[x]);
                            C.ids[x] = v;
                            if (v < in_lo || v >= in_hi) C.outside.push_back(v);
                        }
                    }
                };
                // (one thread: measured at configs[1] on the GPU box, 16-CPU quota, the previous block's host half running beside — 430 ms
                // per block with one thread, 436 with six; the copy is not what the round waits for)
                worker();
            }
            for (CopyChunk &C : chunks)  // (in the order of the path)
                if (!C.outside.empty()) st[C.i].outsideU.insert(st[C.i].outsideU.end(), C.outside.begin(), C.outside.end());
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0 || st[i].tail.on) continue;
                st[i].travel[st[i].pendingFirst].step = st[i].pendingFirstStep;
            }
            if (used) hipMemcpyAsync(b_gather.p, hp, used * 4, hipMemcpyHostToDevice, s);
            for (uint32_t i : batch) {
                const Pick &P = picks[i];
                if (P.chosen < 0 || P.len == 0 || st[i].tail.on) continue;
                CtgState &cs = st[i];
                // globalUniqueTable on the device: its hash set (vertices outside the strand's id range) grows as needed
                if ((uint64_t)cs.outsideU.size() * 2 > cs.gcap) {
                    uint32_t ncap = cs.gcap;
                    while ((uint64_t)cs.outsideU.size() * 4 > ncap) ncap *= 2;
                    DevBuf b_ng = cbuf(i, GRP_FINAL, CB_TSET), b_ou = cbuf(i, GRP_FINAL, CB_PSET);
                    if ((rc = b_ng.alloc((size_t)ncap * 4)) || (rc = b_ou.alloc(cs.outsideU.size() * 4))) return fail(rc);
                    hipMemsetAsync(b_ng.p, 0xFF, (size_t)ncap * 4, s);
                    // (the vertices of this round's path are inserted by the commit below; the earlier ones here)
                    hipMemcpyAsync(b_ou.p, cs.outsideU.data(), cs.outsideU.size() * 4, hipMemcpyHostToDevice, s);
                    trav_launch_commit(b_ou.as<uint32_t>(), cs.outsideU.size(), 0u, 0u, nullptr, b_ng.as<uint32_t>(), ncap - 1, s);
                    cs.gset = b_ng.as<uint32_t>();
                    cs.gcap = ncap;
                }
                // record the walk in the device-side global visited set of this contig
                trav_launch_commit(b_gather.as<uint32_t>() + P.off, P.len, cs.inLo, cs.inHi, cs.gbits, cs.gset, cs.gcap - 1, s);
                cs.committed = true;
            }
        }
        for (uint32_t i : batch) RS[i].chains.clear();  // (the host copies of the round's chains are spent; its segments: below)
        lap("choose+gather");
        return PAG_OK;
    }
    // splice + stop rules (PAlgorithm.cpp:264-360); reqs / req_cs: the seed searches of the contigs that go on
    void splice_batch(const std::vector<uint32_t> &batch, const std::vector<Pick> &picks, std::vector<TravSeedReq> &reqs, std::vector<uint32_t> &req_cs) {
        std::vector<TravSeedReq> slot_req(n_sel);
        std::vector<uint8_t> slot_has(n_sel, 0);
        auto splice = [&](uint32_t i) {
            CtgState &cs = st[i];
            const Pick &P = picks[i];
            const bool leap = P.leap;
            // (the walk was appended to cs.travel by take_walk above)
            if (P.chooseCtgPos != 0) {
                cs.ctgQ.push_back((uint32_t)P.chooseCtgPos);
                while (cs.ctgQ.size() > 4) cs.ctgQ.pop_front();
            }
            if (P.chooseRefPos != 0) {
                cs.refQ.push_back((uint32_t)P.chooseRefPos);
                while (cs.refQ.size() > 4) cs.refQ.pop_front();
            }
            bool ctgRepeat = false, refRepeat = false;
            if (cs.ctgQ.size() >= 4) {
                auto mm = std::minmax_element(cs.ctgQ.begin(), cs.ctgQ.end());
                ctgRepeat = (uint64_t)(*mm.second - *mm.first) <= 2 * deviation;
            }
            if (cs.refQ.size() >= 4) {
                auto mm = std::minmax_element(cs.refQ.begin(), cs.refQ.end());
                refRepeat = (uint64_t)(*mm.second - *mm.first) <= 2 * deviation;
            }
            if (ctgRepeat || refRepeat || leap) {
                if (leap) cs.finalLeap = true;
                cs.done = true;
                return;
            }
            // last contig-consistent vertex of the running path (PAlgorithm.cpp:332-360)
            uint64_t lastCtgPos = 0;
            uint32_t lastU = 0;  // (its k-mer is looked up on the device together with the next seeds)
            bool haveKmer = false;
            for (auto it = cs.travel.rbegin(); it != cs.travel.rend(); ++it) {
                if (it->ctg != 0) {
                    auto d = mapper.singleToDual(it->ctg);
                    if (d.first == cs.chosenOne && d.second >= 0) {
                        lastCtgPos = (uint64_t)d.second;
                        lastU = it->u;
                        haveKmer = true;
                        break;
                    }
                }
            }
            TravSeedReq r{};
            r.ctg = i;
            r.pos = lastCtgPos;
            r.left = lastCtgPos - std::min<uint64_t>(lastCtgPos, 1000 * deviation);
            r.right = lastCtgPos + 1000 * deviation;
            slot_req[i] = r;
            slot_has[i] = 1;
            cs.seeds.clear();
            cs.parentU = lastU;
            cs.haveParent = haveKmer;
        };
        {
            unsigned nthr = std::min<unsigned>((unsigned)batch.size(), std::max(1u, std::min(32u, std::thread::hardware_concurrency())));
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (size_t x; (x = next.fetch_add(1)) < batch.size();) splice(batch[x]);
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(worker);
            worker();
            for (auto &t : pool) t.join();
        }
        for (uint32_t i : batch)
            if (slot_has[i]) {
                reqs.push_back(slot_req[i]);
                req_cs.push_back(i);
            }
        // The round's SEGMENTS: a contig that goes on re-seeds behind the path it has just committed and walks the rest of the
        // strand — through the very checkpoints this round's segments were started from.  They are kept (those still waiting
        // or walking included): the next round's chains adopt them under the conditions of walk_stitch.hpp, which count the
        // marks committed since (Seg::round, MergeCtx::g_*).  A finished contig gives them up.
        for (uint32_t i : batch) {
            if (st[i].done) give_up_segments(i);
            else RS[i].kept = !RS[i].segs.empty();
        }
        lap("splice");
    }
    // next seeds: searchPANode2 + filterPANodes + sort by edit distance + top-K
    int reseed(std::vector<TravSeedReq> &reqs, const std::vector<uint32_t> &req_cs, std::vector<uint32_t> &next_round) {
        int rc;
        if (!reqs.empty()) {
            const uint32_t PARTS = TRAV_SEED_PARTS;
            uint32_t WSTRIDE = 2048;  // words per part of a request
            std::vector<uint32_t> wb;
            for (;;) {  // (a part with more candidates than the stride is searched again with a larger one)
                const size_t words = (size_t)reqs.size() * PARTS * WSTRIDE;
                if ((rc = b_req.alloc(reqs.size() * sizeof(TravSeedReq))) || (rc = b_seedout.alloc(words * 4))) return fail(rc);
                uint32_t *hp = (uint32_t *)pinned(words * 4 + 256);
                if (!hp) return fail(PAG_ENOMEM);
                hipError_t he = hipMemcpyAsync(b_req.p, reqs.data(), reqs.size() * sizeof(TravSeedReq), hipMemcpyHostToDevice, s);
                if (upload_contigs() != PAG_OK) he = hipErrorUnknown;
                trav_launch_seed_window(G, b_tc.as<TravContig>(), b_req.as<TravSeedReq>(), (uint32_t)reqs.size(), deviation,
                                        b_seedout.as<uint32_t>(), WSTRIDE, s);
                if (he == hipSuccess) he = hipMemcpyAsync(hp, b_seedout.p, words * 4, hipMemcpyDeviceToHost, s);
                if (he == hipSuccess) he = hipStreamSynchronize(s);
                if (he != hipSuccess) {
                    set_error("pag_travel: seed search failed: %s", hipGetErrorString(he));
                    return fail(PAG_EFAULT);
                }
                uint32_t most = 0;
                for (size_t q = 0; q < reqs.size() * PARTS; ++q) most = std::max(most, hp[q * WSTRIDE]);
                if (most <= WSTRIDE - 1) {
                    wb.assign(hp, hp + words);
                    break;
                }
                if (most > (1u << 28)) {
                    set_error("pag_travel: seed window with %u candidates", most);
                    return fail(PAG_ENOMEM);
                }
                WSTRIDE = (uint32_t)pow2_at_least((uint64_t)most + 2);
            }
            std::vector<uint32_t> vids;
            std::vector<size_t> cnt(reqs.size());
            for (size_t q = 0; q < reqs.size(); ++q) {
                std::unordered_set<uint32_t> seen;
                size_t n = 0;
                for (uint32_t part = 0; part < PARTS; ++part) {  // (the parts of the window, in offset order)
                    const uint32_t *o = &wb[(q * PARTS + part) * WSTRIDE];
                    for (uint32_t x = 0; x < o[0]; ++x) {
                        uint32_t v = o[1 + x];
                        if (!seen.insert(v).second) continue;         // std::set `unique` in searchPANode2
                        vids.push_back(v);                            // (filterPANodes was applied by the kernel)
                        ++n;
                    }
                }
                cnt[q] = n;
            }
            std::vector<pag_path_node> attrs;
            if ((rc = fetch_vertices(vids, attrs))) return fail(rc);
            {   // k-mers of the parents (last contig-consistent vertex of each running path)
                std::vector<uint32_t> pu(reqs.size());
                for (size_t q = 0; q < reqs.size(); ++q) pu[q] = st[req_cs[q]].parentU;
                std::vector<pag_path_node> pa(reqs.size());
                if ((rc = b_vids.alloc(pu.size() * 8)) || (rc = b_gather.alloc(pu.size() * sizeof(pag_path_node) + 64))) return fail(rc);
                hipMemcpyAsync(b_vids.p, pu.data(), pu.size() * 4, hipMemcpyHostToDevice, s);
                hipMemsetAsync(b_vids.as<uint32_t>() + pu.size(), 0, pu.size() * 4, s);
                trav_launch_gather_path(G, b_vids.as<uint32_t>(), b_vids.as<uint32_t>() + pu.size(), pu.size(), b_gather.as<pag_path_node>(), s);
                hipMemcpyAsync(pa.data(), b_gather.p, pa.size() * sizeof(pag_path_node), hipMemcpyDeviceToHost, s);
                if (hipStreamSynchronize(s) != hipSuccess) {
                    set_error("pag_travel: parent k-mer lookup failed");
                    return fail(PAG_EFAULT);
                }
                for (size_t q = 0; q < reqs.size(); ++q) st[req_cs[q]].parentCode = pa[q].code;
            }
            size_t at = 0;
            for (size_t q = 0; q < reqs.size(); ++q) {
                CtgState &cs = st[req_cs[q]];
                const std::string parent = cs.haveParent ? code2kmer(cs.parentCode, k) : std::string();
                std::vector<pag_path_node> cand(attrs.begin() + at, attrs.begin() + at + cnt[q]);
                at += cnt[q];
                // std::sort with the reference's comparator (edit distance to the parent k-mer), unstable:
                // precomputed keys give the same comparison outcomes, hence the same permutation
                struct Keyed {
                    size_t d;
                    pag_path_node n;
                };
                std::vector<Keyed> keyed;
                keyed.reserve(cand.size());
                for (auto &c : cand) keyed.push_back({edit_distance(parent, code2kmer(c.code, k)), c});
                std::sort(keyed.begin(), keyed.end(), [](const Keyed &a, const Keyed &b) { return a.d < b.d; });
                cs.seeds.clear();
                for (size_t x = 0; x < keyed.size() && x < topK; ++x) cs.seeds.push_back(keyed[x].n);
                if (cs.seeds.empty()) {
                    cs.done = true;
                    give_up_segments(req_cs[q]);
                }
                else next_round.push_back(req_cs[q]);
            }
        }
        return PAG_OK;
    }
    int event_loop() {
        int rc;
        t_progress = now_ms();
        t_first_fin = 0;
        t_last_news = now_ms();
        while (n_live || !over_queue.empty()) {
            std::vector<uint32_t> fin, touched, batch, req_cs, next_round;
            bool again = false;
            if ((rc = poll_finished(fin, again))) return rc;
            if (again) continue;
            auto mark = [&](const char *what) {  // (PAG_WALK_DEBUG: where this thread's time goes, iteration by iteration)
                if (wdebug) std::fprintf(stderr, "[walk] t=%.1f ms loop: %s (%zu jobs, %zu contigs decided, %u live)\n", now_ms() - tw0, what, fin.size(), batch.size(), n_live);
                if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 4u, n_live, 0, 0, (uint64_t)(uintptr_t)what, 0, fin.size(), batch.size()});
            };
            mark("polled");
            std::vector<Got> got;
            if ((rc = fetch_paths(fin, got))) return rc;
            mark("fetched");
            if ((rc = stitch_finished(got, touched))) return rc;
            mark("stitched");
            decide_rounds(touched, batch);
            flush_backlog();
            if (batch.empty()) {
                if ((rc = publish())) return fail(rc);
                mark("published");
                lap("round prep");
                continue;
            }
            // per contig: choose (PAlgorithm.cpp:238-262), commit, splice, stop or re-seed
            std::vector<Pick> picks(n_sel);
            if ((rc = take_walks(batch, picks))) return rc;
            mark("walks taken");
            std::vector<TravSeedReq> reqs;
            splice_batch(batch, picks, reqs, req_cs);
            mark("spliced");
            if ((rc = reseed(reqs, req_cs, next_round))) return rc;
            mark("re-seeded");
            for (uint32_t i : batch)
                if (st[i].done && (rc = deliver_contig(i))) return fail(rc);
            mark("delivered");
            lap("reseed");
            // the follow-up rounds
            if (!next_round.empty() && (rc = start_rounds(next_round))) return fail(rc);
            mark("next rounds posted");
            flush_backlog();
            if ((rc = publish())) return fail(rc);
            mark("published");
            lap("round prep");
        }
        return PAG_OK;
    }
    // the walker is sent home; whatever has not been delivered while the walks ran; statistics
    int finish() {
        int rc;
        shutdown_walker();
        g->defer_free = false;
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
        t_walk = now_ms() - tw0;
        lap("walk");
        if (wtrace) {
            uint64_t d0 = ~0ull;
            for (const TraceEv &e : trace)
                if (e.what == 0u) d0 = std::min(d0, e.a);
            std::fprintf(stderr, "[trace] walks %.2f ms, %zu events; device times relative to the first job's begin\n", t_walk, trace.size());
            for (const TraceEv &e : trace) {
                if (e.what == 0u)
                    std::fprintf(stderr, "[trace] done t=%.2f ctg %u %s %d dev %.2f..%.2f len %llu classify %llu\n", e.t, e.ctg, e.kind ? "seg" : "chain", e.idx, (double)(e.a - d0) * 1e-5,
                                 (double)(e.b - d0) * 1e-5, (unsigned long long)e.len, (unsigned long long)e.classify);
                else if (e.what == 4u)
                    std::fprintf(stderr, "[trace] loop t=%.2f %s jobs %llu decided %llu live %u\n", e.t, (const char *)(uintptr_t)e.a, (unsigned long long)e.len, (unsigned long long)e.classify, e.ctg);
                else if (e.what == 1u)
                    std::fprintf(stderr, "[trace] post t=%.2f ctg %u %s %d ring %llu mode %llu init %llu\n", e.t, e.ctg, e.kind ? "seg" : "chain", e.idx, (unsigned long long)e.a, (unsigned long long)e.b,
                                 (unsigned long long)e.len);
                else
                    std::fprintf(stderr, "[trace] over t=%.2f ctg %u round %llu chain %d len %llu%s\n", e.t, e.ctg, (unsigned long long)e.a, e.idx, (unsigned long long)e.len, e.b ? " leap" : "");
            }
            trace.clear();
        }
        if (timing || wdebug) {
            std::fprintf(stderr, "[timing] stitch: bookkeeping %.1f ms, paths %.1f ms, chains %.1f ms; posting jobs (all callers) %.1f ms; fetch memory: chunk %zu of %zu; walk arena: %.2f of %.2f GB used\n", t_st[0], t_st[1], t_st[2], t_st[3],
                         fetch_chunk + 1, g->fetch_chunks.size(), g->walk_arena_used / 1e9, g->walk_arena_cap / 1e9);
            std::fprintf(stderr, "[timing] leaping zone: %llu segment jobs, %llu adopted, refused by reason (unusable, no junction, not a boundary, cannot leap yet, window top, window bottom, contig-following record, coordinate-free record): %llu %llu %llu %llu %llu %llu %llu %llu\n",
                         (unsigned long long)n_leap_jobs, (unsigned long long)n_leap_adopted.load(), (unsigned long long)n_leap_refused[0].load(), (unsigned long long)n_leap_refused[1].load(), (unsigned long long)n_leap_refused[2].load(),
                         (unsigned long long)n_leap_refused[3].load(), (unsigned long long)n_leap_refused[4].load(), (unsigned long long)n_leap_refused[5].load(), (unsigned long long)n_leap_refused[6].load(), (unsigned long long)n_leap_refused[7].load());
            {
                size_t n_tail = 0, n_tail_behind = 0;
                for (const CtgState &cs : st) n_tail += cs.tail.on, n_tail_behind += cs.tail.on && cs.tail.m0;
                std::fprintf(stderr, "[timing] last rounds put together on the device: %zu of %u contigs (%zu behind an earlier round's path)\n", n_tail, n_sel, n_tail_behind);
            }
            std::fprintf(stderr, "[timing] pieces: %llu segment jobs, %llu resume jobs, %llu vertices adopted, %llu segments not adoptable\n", (unsigned long long)n_seg_jobs,
                         (unsigned long long)n_resume_jobs, (unsigned long long)n_adopted.load(), (unsigned long long)n_merge_fail.load());
        }

        // ---- epilogue: whatever has not been delivered while the walks ran (see deliver_contig)
        for (uint32_t i = 0; i < n_sel; ++i)
            if (!st[i].delivered) filter_travel(st[i]);
        {   // the full records of those sequences: one gather, results straight into the pinned array the handle keeps for
            // pag_travel_path()
            uint64_t tot = 0;
            for (auto &cs : st)
                if (!cs.delivered) tot += cs.travel.size();
            if (g->path_cap < tot + 1) {
                if (g->path_store) hipHostFree(g->path_store);
                g->path_store = nullptr;
                g->path_cap = 0;
                const size_t want = tot + tot / 8 + 1024;
                if (hipHostMalloc((void **)&g->path_store, want * sizeof(pag_path_node), hipHostMallocDefault) != hipSuccess) {
                    set_error("pag_travel: hipHostMalloc for %zu path records failed", want);
                    return fail(PAG_ENOMEM);
                }
                g->path_cap = want;
            }
            uint32_t *hp = (uint32_t *)pinned(tot * 8 + 256);
            if (!hp) return fail(PAG_ENOMEM);
            uint64_t at = 0;
            for (auto &cs : st) {
                if (cs.delivered) continue;
                const size_t slot2 = 2 * (size_t)cs.ci + (cs.forward ? 0 : 1);
                g->path_off[slot2] = at;
                g->path_len[slot2] = cs.travel.size();
                g->path_valid[slot2] = 1;
                for (size_t x = 0; x < cs.travel.size(); ++x) {
                    hp[at + x] = cs.travel[x].u;
                    hp[tot + at + x] = (uint32_t)cs.travel[x].step;
                }
                at += cs.travel.size();
            }
            DevBuf b_fin = buf();
            if ((rc = b_fin.alloc(tot * 8 + 64)) || (rc = b_gather.alloc(tot * sizeof(pag_path_node) + 64))) return fail(rc);
            if (tot) {
                PAG_HIP_TRY(hipMemcpyAsync(b_fin.p, hp, tot * 8, hipMemcpyHostToDevice, s));
                trav_launch_gather_path(G, b_fin.as<uint32_t>(), b_fin.as<uint32_t>() + tot, tot, b_gather.as<pag_path_node>(), s);
                PAG_HIP_TRY(hipMemcpyAsync(g->path_store, b_gather.p, tot * sizeof(pag_path_node), hipMemcpyDeviceToHost, s));
            }
            PAG_HIP_TRY(hipStreamSynchronize(s));
            if (g->deliver_stream) PAG_HIP_TRY(hipStreamSynchronize(g->deliver_stream));  // (the deliveries made during the walks)
        }
        lap("epilogue");
        if (timing) {
            std::fprintf(stderr, "[timing] walks redone without speculation: %u\n", respeculated);
            std::fprintf(stderr, "[timing] segment jobs left behind by rounds that were decided without them: %llu\n", (unsigned long long)n_orphans);
            std::fprintf(stderr, "[timing] pag_travel laps:");
            for (auto &l : laps) std::fprintf(stderr, " %s %.1f ms;", l.first, l.second);
            std::fprintf(stderr, "\n");
        }
        if (stats) {
            stats->ms_compact = t_compact;
            stats->ms_walk = t_walk;
            stats->ms_total = now_ms() - t_begin;
            stats->rounds = rounds;
            stats->jobs = jobs_total;
            stats->walk_steps = steps_total;
            stats->classify_calls = classify_total;
            stats->probes = probe_total;
            stats->records = record_total;
        }
        return PAG_OK;
    }
    int run() {
        int rc;
        if ((rc = begin()) || (rc = setup_contigs())) return rc;
        if (n_sel == 0) return PAG_OK;
        if ((rc = first_seeds()) || (rc = setup_rings()) || (rc = reserve_arena()) || (rc = post_first_rounds()) || (rc = event_loop())) return rc;
        return finish();
    }
};
