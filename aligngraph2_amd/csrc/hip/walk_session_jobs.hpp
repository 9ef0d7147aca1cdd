// WalkSession, first layer (included by k5_travel_host.hip): the call, its configuration and timing, the buffers a traversal
// takes from the handle, and how walk jobs are prepared, posted to the rings of the persistent walker and published.
#pragma once

using namespace stitch;  // Piece, View, Seg, Chain, RoundState, PartAgg, extend_chain .. try_merge_leap (walk_stitch.hpp)

// One call of pag_travel (one graph, the contigs of one block): the state of the traversal and the steps it goes through.
// run() is the whole of it — the traversal view, the contigs' tables and first seeds, the job rings, the first rounds, then the
// event loop (finished jobs -> their paths -> the chains move on -> decided rounds are chosen from, spliced, re-seeded or
// delivered) and the epilogue; the members are what those steps share.
struct WalkJobs {
    // ---- the call
    pag_graph *g;
    const pag_seqs *ctgs;
    const int32_t *orient;
    const uint32_t *ref_len;
    uint64_t n_refs;
    const pag_travel_params *prm;
    pag_travel_stats *stats;
    WalkJobs(pag_graph *g_, const pag_seqs *ctgs_, const int32_t *orient_, const uint32_t *ref_len_, uint64_t n_refs_, const pag_travel_params *prm_,
                pag_travel_stats *stats_)
        : g(g_), ctgs(ctgs_), orient(orient_), ref_len(ref_len_), n_refs(n_refs_), prm(prm_), stats(stats_), cfg(WalkConfig::from_env()),
          mapper(ctgs_->len, ctgs_->n_seqs), refMapper(ref_len_, n_refs_) {}

    // ---- configuration, timing
    hipStream_t s = nullptr;
    double t_begin = 0;
    const WalkConfig cfg;
    bool timing = false, wdebug = false, wtrace = false;
    // PAG_WALK_TRACE: what = 0 job done (a, b = device begin / end in 10 ns ticks), 1 job posted (a = ring, b = mode), 2 round over
    // (a = round, b = leap), 3 round started (a = round, b = seeds)
    struct TraceEv {
        double t;
        uint32_t what, ctg;
        int32_t kind, idx;
        uint64_t a, b, len, classify;
    };
    std::vector<TraceEv> trace;
    double lap_t = 0;
    std::vector<std::pair<const char *, double>> laps;
    uint32_t k = 0;
    uint64_t deviation = 0;
    double errorRate = 0, startSplit = 0;
    size_t topK = 0;
    int slot = TRAV_SLOT0;  // pool slots of the handle are handed out in the order of the buf() calls
    void lap(const char *what) {
        if (!timing) return;
        const double t = now_ms();
        for (auto &l : laps)
            if (l.first == what) {
                l.second += t - lap_t;
                lap_t = t;
                return;
            }
        laps.emplace_back(what, t - lap_t);
        lap_t = t;
    }
    DevBuf buf() { return DevBuf(g, slot++); }
    // pinned host staging area (grown, kept in the handle): packed job results on their way in, uploads on their way out
    std::vector<void *> pinned_parked;  // (freeing host memory synchronises the device: never while the walker grid is resident)
    void *pinned(size_t bytes) {
        if (g->pin_bytes < bytes) {
            if (g->pin_host) {
                if (g->defer_free) pinned_parked.push_back(g->pin_host);
                else hipHostFree(g->pin_host);
            }
            g->pin_host = nullptr;
            g->pin_bytes = 0;
            const size_t want = bytes + bytes / 4 + (1u << 20);
            if (hipHostMalloc(&g->pin_host, want, hipHostMallocDefault) != hipSuccess) {
                set_error("pag_travel: hipHostMalloc(%zu) failed", want);
                return nullptr;
            }
            g->pin_bytes = want;
        }
        return g->pin_host;
    }

    // Pinned memory that keeps what it is given for the whole call: the fetched paths of finished jobs stay where the copy
    // from the device put them (segments and chains refer to them by pointer).  64 MB chunks kept by the handle.
    size_t fetch_chunk = 0, fetch_used = 0;
    static constexpr size_t FETCH_CHUNK = 64u << 20;
    void *fetch_alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        for (; fetch_chunk < g->fetch_chunks.size(); ++fetch_chunk, fetch_used = 0)
            if (fetch_used + bytes <= g->fetch_chunk_bytes[fetch_chunk]) {
                void *q = (char *)g->fetch_chunks[fetch_chunk] + fetch_used;
                fetch_used += bytes;
                return q;
            }
        void *q = nullptr;
        const size_t want = std::max(FETCH_CHUNK, bytes);
        if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) {
            set_error("pag_travel: hipHostMalloc(%zu) failed", want);
            return nullptr;
        }
        g->fetch_chunks.push_back(q);
        g->fetch_chunk_bytes.push_back(want);
        fetch_used = bytes;  // (fetch_chunk is the index of the new chunk)
        return q;
    }

    // ---- the traversal view, the contigs
    TravGraph G{};
    double t_compact = 0;
    Mapper mapper, refMapper;
    uint32_t n_ctgs = 0, n_sel = 0;
    std::vector<CtgState> st;  // one entry per (contig, orientation) that is walked
    uint64_t nodes_total = 0;
    DevBuf b_packed, b_nodes, b_starts, b_sizes, b_tc, b_seedout, b_req, b_gset, b_gather, b_vids, b_gbits, b_ckreq, b_ckout;
    std::vector<TravContig> tc;
    static constexpr uint32_t SEED_STRIDE = 4096;
    void fill_contigs() {
        for (uint32_t i = 0; i < n_sel; ++i) {
            CtgState &cs = st[i];
            TravContig &t = tc[i];
            t.nodes = b_nodes.as<uint32_t>() + cs.nodesOff;
            t.n_kmers = cs.len >= k ? cs.len - k + 1 : 0;
            t.ctg_left = cs.ctgLeft;
            t.ctg_right = cs.ctgRight;
            t.rev_left = cs.revLeft;
            t.rev_right = cs.revRight;
            t.split_size = (uint64_t)(cs.len * startSplit);
            t.leap_min = 1 - startSplit;
            t.starts = b_starts.as<uint64_t>();
            t.sizes = b_sizes.as<uint64_t>();
            t.n_ctgs = n_ctgs;
            t.in_lo = cs.inLo;
            t.in_hi = cs.inHi;
            t.g_lo = cs.inLo;
            t.g_hi = cs.inHi;
            t.gbits = cs.committed ? cs.gbits : nullptr;
            t.gset = cs.committed ? cs.gset : nullptr;
            t.gmask = cs.gcap - 1;
            t.gwin_lo = cs.gwinLo;
            t.gwin_hi = cs.gwinHi;
        }
    }
    int upload_contigs() {
        fill_contigs();
        PAG_HIP_TRY(hipMemcpyAsync(b_tc.p, tc.data(), n_sel * sizeof(TravContig), hipMemcpyHostToDevice, s));
        return PAG_OK;
    }
    // vertex attributes for a list of vertex ids
    int fetch_vertices(const std::vector<uint32_t> &vids, std::vector<pag_path_node> &out) {
        out.resize(vids.size());
        if (vids.empty()) return PAG_OK;
        int r;
        if ((r = b_vids.alloc(vids.size() * 4)) || (r = b_gather.alloc(vids.size() * sizeof(pag_path_node)))) return r;
        PAG_HIP_TRY(hipMemcpyAsync(b_vids.p, vids.data(), vids.size() * 4, hipMemcpyHostToDevice, s));
        trav_launch_gather_vertices(G, b_vids.as<uint32_t>(), (uint32_t)vids.size(), b_gather.as<pag_path_node>(), s);
        PAG_HIP_TRY(hipMemcpyAsync(out.data(), b_gather.p, vids.size() * sizeof(pag_path_node), hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        return PAG_OK;
    }

    // ---- statistics of the call
    uint64_t rounds = 0, jobs_total = 0, steps_total = 0, classify_total = 0, probe_total = 0, record_total = 0;
    double t_walk = 0;

    // ---- the walks.
    // The contigs are independent state machines (walk the seeds of the round, choose, splice, stop or re-seed); a persistent
    // walker grid executes whatever jobs are posted, and this loop posts the next piece of work of a contig as soon as what it
    // depends on is done.
    //
    // PIECES.  A graphTravel (PAlgorithm.tcc:172-298) is one chain of dependent steps: a quarter of a million path vertices on
    // a 1 Mb contig, walked by one wavefront at ~1.7 us per step, while the other 255 compute units idle.  The chain is cut
    // along the contig coordinate:
    //   * the walk of a seed (a CHAIN) runs as a job with a stop coordinate: it ends at the first iteration boundary of
    //     graphTravel whose last vertex lies at or beyond it;
    //   * ahead of it, SEGMENT jobs start at checkpoint vertices (the most abundant on-contig vertex near x0 + j * seg_len)
    //     and walk as if they were graphTravels of their own that can never leap (TRAV_MODE_SPEC), each up to the next
    //     checkpoint plus an overlap.  They only exist up to the coordinate at which the real walk could start leaping;
    //   * when a chain has reached the start of a finished segment and its tail COINCIDES, vertex for vertex, with a stretch of
    //     that segment's path, the rest of the segment's path is adopted (see `try_merge` for the condition under which that is
    //     exactly what the real walk would have done) and the chain goes on to the next segment;
    //   * where no segment can be adopted the chain continues as a RESUME job (the path so far is handed to the walker, which
    //     marks it visited and goes on exactly as graphTravel would), with the next checkpoint as its stop coordinate, or
    //     without one from the zone where leaping becomes possible to the end of the walk.
    // The result is vertex-for-vertex the path of the un-cut walk (PAG_WALK_PIECES=0 runs that, tests compare both with the
    // host restatement of the reference), and the critical path of a contig shrinks from the whole contig to one segment
    // plus the leaping zone.
    enum { CB_SEQV = 0, CB_SEQS, CB_ARV, CB_ARS, CB_TSET, CB_PSET, CB_STAMP, CB_TBITS, CB_SEQX, CB_N };
    enum { GRP_ROUND = 0, GRP_CHAIN0 = 1, GRP_FINAL = 9, GROUPS = 10 };  // buffer groups per contig (chains: top-K <= 8)
    DevBuf cbuf(uint32_t i, int grp, int b) { return DevBuf(g, &g->cpool[((size_t)i * GROUPS + grp) * CB_N + b]); }
    // rings of job records, served in order: 0 chain jobs (what a contig's progress waits for), 1 segment jobs of contigs
    // in a later round (they are further along their critical path), 2 segment jobs of first rounds.  slot = ring * QCAP +
    // number mod QCAP
    // A ring holds the jobs of a round that are in flight; a slot is reused QCAP postings later.  Sized by what the contigs
    // of this call can post in one round (segments every few kb of every strand, top-K <= 8 chains each), twice over.
    uint32_t QCAP = 32768;
    static constexpr uint32_t NR = TRAV_RINGS;
    TravQueue *hq = nullptr;
    TravPosted *hjobs = nullptr;
    TravJobOut *houts = nullptr;
    uint32_t *hdone = nullptr;
    double t_walk0 = 0, tw0 = 0;  // (debug time stamps count from the launch of the walker)
    bool use_pieces = true, use_leap_pieces = true, force_exact = false;
    uint64_t seg_len_env = 0, seg_ov = 0;

    double t_st[4] = {0, 0, 0, 0};  // stitch: bookkeeping / paths of finished jobs / chains moving on; posting (inside the others)
    // (The stitch below is serial on purpose.  Worker threads — spinning, polling or sleeping on a condition variable, 4 to
    // 12 of them — cut the path copies from 33 to 10 ms on the GPU box (16-CPU cgroup quota, busy host), but every HIP call
    // of this thread (posting, fetching, re-seeding) got several times slower while they were active and the walks took
    // 180-450 ms instead of 165-180 ms.)
    std::vector<stitch::RoundState> RS;
    struct JobRef {
        uint32_t ctg = 0;
        int kind = 0;  // 0: chain job (seed or resume), 1: segment
        int idx = 0;   // chain / segment number
        uint64_t init_len = 0;
        bool live = false;
        uint32_t epoch = 0;  // RoundState::seg_epoch of its contig when the job was posted (see is_orphan)
    };
    std::vector<JobRef> jref;
    // A round of a contig is decided when all its chains are final.  Segment jobs of the round that are still waiting or
    // walking then are ORPHANS: nobody will look at their paths (a chain that dead-ends at a fifth of its contig leaves four
    // fifths of the round's segments behind — at BASELINE configs[1] the next round of such a contig used to start when the
    // last of them had been walked, ~85 ms into the walks, and its own walk was the tail everything waited for).  An orphan
    // that no wave has taken yet is cancelled (the wave that takes it reports it done at once); one that is walking finishes
    // into its own buffers — a round's buffers come from the walk arena, which is never handed out twice within one
    // pag_travel; a round that had to fall back on the per-contig slots waits for its jobs as before (RoundState::slot_bufs).
    uint64_t n_orphans = 0;
    uint32_t n_posted[TRAV_RINGS] = {0, 0, 0}, n_live = 0, respeculated = 0;
    uint32_t scan_from[TRAV_RINGS] = {0, 0, 0};  // per ring: every job number below it has been handled
    std::atomic<uint64_t> n_adopted{0}, n_merge_fail{0}, n_leap_adopted{0}, n_leap_refused[8];
    uint64_t n_seg_jobs = 0, n_resume_jobs = 0, n_leap_jobs = 0;
    WalkerGrid walkers;
    void shutdown_walker() {
        if (!walkers.up) return;
        walkers.shutdown();
        g->defer_free = false;
        for (void *q : g->deferred) hipFree(q);
        g->deferred.clear();
        for (void *q : pinned_parked) hipHostFree(q);
        pinned_parked.clear();
    }
    struct JobPlan {
        int kind, idx;
        uint64_t cap;       // sequence capacity (vertices)
        uint32_t start_vid; // old id of the start vertex
        uint32_t mode, stop_pc;
        const Chain *init;  // RESUME: the chain whose path so far the job continues
        bool exact;
        uint32_t win_lo = 0, win_hi = 0;  // id range of the job's direct-mapped marks (0, 0: the whole strand)
        uint32_t win_low = 0;             // TRAV_MODE_LEAP: forced lower end of the travel coordinate window
    };
    bool need_publish = false;
    // what the posted batches want cleared before their jobs become visible (hash sets, stamps, travel bits): collected, and
    // cleared by ONE launch when the batch is published (trav_clear_ranges) — five hipMemsetAsync per batch, 673 fill kernels of
    // ~14 us per block at configs[1], ran one after the other on the stream in front of the first job
    std::vector<TravClear> clears;
    void want_clear(void *p, size_t bytes, uint32_t byte_value) {
        if (bytes) clears.push_back(TravClear{p, (uint64_t)bytes, byte_value * 0x01010101u, 0u});
    }
    int flush_clears() {  // (asynchronous: the list is read from pinned memory that lives as long as the walks; publish() waits for the stream)
        if (clears.empty()) return PAG_OK;
        TravClear *d = (TravClear *)fetch_alloc(clears.size() * sizeof(TravClear));
        if (!d) return PAG_ENOMEM;
        std::memcpy(d, clears.data(), clears.size() * sizeof(TravClear));
        const int r = trav_clear_ranges(d, clears.size(), s);
        clears.clear();
        return r;
    }
    // a prepared job enters its ring (in posting order; the walker takes the rings' jobs in that order)
    struct Deferred {
        TravPosted P;
        JobRef jr;
    };
    std::vector<std::vector<Deferred>> deferred;
    bool defer_ring2 = false;
    // A job enters its ring when the slot it takes (its number mod QCAP) is free again; until then it waits in the ring's
    // backlog, in posting order (a ring smaller than the jobs of a round is a matter of flow control, not an error).
    struct Backlogged {
        TravPosted P;
        JobRef jr;
    };
    std::deque<Backlogged> backlog[TRAV_RINGS];
    bool place_job(uint32_t ring, const TravPosted &P, const JobRef &jr2) {
        const uint32_t jn = n_posted[ring], slot = ring * QCAP + jn % QCAP;
        if (jref[slot].live) return false;
        hjobs[slot] = P;
        hdone[slot] = 0;
        jref[slot] = jr2;
        if (jr2.kind == 0) RS[jr2.ctg].chains[(size_t)jr2.idx].job = (int)slot;
        n_posted[ring] += 1;
        return true;
    }
    int commit_job(uint32_t ring, const TravPosted &P, const JobRef &jr2, uint32_t mode, uint32_t stop_pc) {
        if (jr2.kind == 0) {
            Chain &ch = RS[jr2.ctg].chains[(size_t)jr2.idx];
            ch.job = 0x7FFFFFFF;  // (outstanding; the slot number follows when the job enters the ring)
            ch.job_mode = mode;
            ch.job_stop = stop_pc;
        }
        n_live += 1;
        RS[jr2.ctg].live_jobs += 1;
        jobs_total += 1;
        if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 1u, jr2.ctg, (int32_t)jr2.kind, (int32_t)jr2.idx, ring, mode, jr2.init_len, 0});
        if (!backlog[ring].empty() || !place_job(ring, P, jr2)) backlog[ring].push_back(Backlogged{P, jr2});
        return PAG_OK;
    }
    bool is_orphan(const JobRef &jr) const { return jr.kind == 1 && jr.epoch != RS[jr.ctg].seg_epoch; }
    void flush_backlog() {
        for (uint32_t ring = 0; ring < TRAV_RINGS; ++ring)
            while (!backlog[ring].empty()) {
                if (is_orphan(backlog[ring].front().jr)) {  // (never entered a ring: gone)
                    backlog[ring].pop_front();
                    n_live -= 1;
                    continue;
                }
                if (!place_job(ring, backlog[ring].front().P, backlog[ring].front().jr)) break;
                backlog[ring].pop_front();
                need_publish = true;
            }
    }
    // The segment list of contig i is given up (the contig is finished, or its next round plans its own): the jobs of the list
    // that no wave has taken are cancelled, those that are walking finish as orphans.
    void give_up_segments(uint32_t i) {
        RoundState &R = RS[i];
        if (R.live_jobs != 0)  // (nothing of the contig is in a ring otherwise: most contigs finish that way)
            for (uint32_t ring = 0; ring < TRAV_RINGS; ++ring) {
                // the live jobs of a ring are among its last QCAP postings, none below scan_from
                const uint32_t hi = n_posted[ring], lo = std::max(scan_from[ring], hi > QCAP ? hi - QCAP : 0u);
                for (uint32_t jn = lo; jn < hi; ++jn) {
                    const uint32_t q = jn % QCAP;
                    JobRef &jr = jref[ring * QCAP + q];
                    if (jr.live && jr.ctg == i && jr.kind == 1 && jr.epoch == R.seg_epoch) {
                        __atomic_fetch_or(&hjobs[ring * QCAP + q].J.mode, (uint32_t)TRAV_MODE_CANCELLED, __ATOMIC_RELEASE);
                        ++n_orphans;
                    }
                }
            }
        R.seg_epoch += 1;
        R.segs.clear();
        R.n_spec = 0;
        R.zone_end = 0;
        R.live_jobs = 0;
        R.kept = false;
    }
    // buffers + job records of a batch of jobs of contig i (memsets and uploads go to stream s; the records become visible to
    // the walker only by publish())
    int post_batch(uint32_t i, int grp, const std::vector<JobPlan> &plans) {
        if (plans.empty()) return PAG_OK;
        const double tp0 = now_ms();
        struct PostTimer {
            double t0, *acc;
            ~PostTimer() { *acc += now_ms() - t0; }
        } post_timer{tp0, &t_st[3]};
        CtgState &cs = st[i];
        RoundState &R = RS[i];
        const uint64_t PG = TRAV_PROBE_GROUPS;
        // one travel epoch / probe stamp per vertex of the job's id range (the whole strand, or the surroundings of a
        // segment); padded to a multiple of four so that the walker's window refills can use 16-byte loads
        const size_t nj = plans.size();
        std::vector<uint64_t> o_seq(nj + 1, 0), o_oc(nj + 1, 0), o_st(nj + 1, 0), o_tb(nj + 1, 0), spans(nj, 0), o_x(nj + 1, 0);
        for (size_t j = 0; j < nj; ++j) {
            o_x[j + 1] = o_x[j] + ((plans[j].mode & TRAV_MODE_LEAP) ? plans[j].cap : 0);
            const uint32_t lo = plans[j].win_hi ? plans[j].win_lo : cs.inLo, hi = plans[j].win_hi ? plans[j].win_hi : cs.inHi;
            spans[j] = ((uint64_t)(hi - lo) + 1 + 3) & ~3ull;
            o_seq[j + 1] = o_seq[j] + plans[j].cap;
            // (a walk in the leaping zone visits vertices without a contig coordinate all the time: they live in the hash sets)
            o_oc[j + 1] = o_oc[j] + pow2_at_least((plans[j].mode & TRAV_MODE_LEAP) ? plans[j].cap + 8192 : plans[j].cap / 4 + 4096);
            o_st[j + 1] = o_st[j] + PG * spans[j];
            o_tb[j + 1] = o_tb[j] + spans[j] + 4;
        }
        // the batch's buffers come out of the walk arena (one allocation of the handle, bump pointer, reset per pag_travel:
        // a cold process otherwise spends seconds in thousands of hipMalloc calls); the per-(contig, group) slots take over
        // when the arena is used up
        DevBuf b_sv = cbuf(i, grp, CB_SEQV), b_ss = cbuf(i, grp, CB_SEQS), b_av = cbuf(i, grp, CB_ARV), b_as = cbuf(i, grp, CB_ARS),
               b_ts = cbuf(i, grp, CB_TSET), b_ps = cbuf(i, grp, CB_PSET), b_st = cbuf(i, grp, CB_STAMP), b_tb = cbuf(i, grp, CB_TBITS),
               b_sx = cbuf(i, grp, CB_SEQX);
        int r;
        {
            const size_t need[9] = {(size_t)o_seq[nj] * 4, (size_t)o_seq[nj] * 4, (size_t)o_seq[nj] * PG * 4, (size_t)o_seq[nj] * PG * 4,
                                    (size_t)o_oc[nj] * 8, (size_t)o_oc[nj] * PG * 8, (size_t)o_st[nj] * 4, (size_t)o_tb[nj] * 4, (size_t)o_x[nj] * 8};
            DevBuf *bufs[9] = {&b_sv, &b_ss, &b_av, &b_as, &b_ts, &b_ps, &b_st, &b_tb, &b_sx};
            size_t tot = 0;
            for (size_t q = 0; q < 9; ++q) tot += (need[q] + 16 + 255) & ~(size_t)255;
            if (g->walk_arena && g->walk_arena_used + tot <= g->walk_arena_cap) {
                for (size_t q = 0; q < 9; ++q) {
                    bufs[q]->p = (char *)g->walk_arena + g->walk_arena_used;
                    g->walk_arena_used += (need[q] + 16 + 255) & ~(size_t)255;
                }
            } else {
                R.slot_bufs = true;  // (per-contig slots are handed out again by the next batch of the group)
                for (size_t q = 0; q < 9; ++q)
                    if ((r = bufs[q]->alloc(need[q]))) return r;
            }
        }
        want_clear(b_sx.p, o_x[nj] * 8, 0u);
        want_clear(b_ts.p, o_oc[nj] * 8, 0xFFu);
        want_clear(b_ps.p, o_oc[nj] * PG * 8, 0u);
        want_clear(b_st.p, o_st[nj] * 4, 0u);
        want_clear(b_tb.p, o_tb[nj] * 4, 0u);
        fill_contigs();
        for (size_t j = 0; j < nj; ++j) {
            const JobPlan &pl = plans[j];
            const uint32_t ring = pl.kind == 0 ? 0u : (R.round > 1 ? 1u : 2u);
            const uint64_t cap = pl.cap, oc = o_oc[j + 1] - o_oc[j];
            TravPosted P{};
            TravJob &J = P.J;
            J.ctg = i;
            J.start = pl.start_vid;
            J.has_size = R.has_size;
            J.seq_v = b_sv.as<uint32_t>() + o_seq[j];
            J.seq_s = b_ss.as<uint32_t>() + o_seq[j];
            J.seq_cap = cap;
            J.arena_v = b_av.as<uint32_t>() + o_seq[j] * PG;
            J.arena_s = b_as.as<uint32_t>() + o_seq[j] * PG;
            J.arena_cap = PG * cap;
            J.stamp = b_st.as<uint32_t>() + o_st[j];
            J.stamp_stride = (uint32_t)spans[j];
            J.tbits = b_tb.as<uint32_t>() + o_tb[j];
            J.tset = b_ts.as<uint64_t>() + o_oc[j];
            J.tmask = (uint32_t)oc - 1;
            J.pset = b_ps.as<uint64_t>() + o_oc[j] * PG;
            J.pmask = (uint32_t)oc - 1;
            J.exact = (pl.exact || force_exact) ? 1u : 0u;
            J.mode = pl.mode;
            J.stop_pc = pl.stop_pc;
            J.init_len = 0;
            J.win_low = pl.win_low;
            J.seq_x = (pl.mode & TRAV_MODE_LEAP) ? b_sx.as<uint64_t>() + o_x[j] : nullptr;
            if (pl.mode & TRAV_MODE_RESUME) {
                const uint64_t n0 = pl.init->len;
                if (n0 == 0 || n0 > cap) {
                    set_error("pag_travel: resume job with a %llu-vertex path in a %llu-vertex buffer", (unsigned long long)n0, (unsigned long long)cap);
                    return PAG_EFAULT;
                }
                J.init_len = n0;
                // (put together in pinned memory: the copies below are asynchronous for real)
                uint32_t *flat = (uint32_t *)fetch_alloc(n0 * 8);
                if (!flat) return PAG_ENOMEM;
                flatten_chain(*pl.init, flat, flat + n0, nullptr);
                PAG_HIP_TRY(hipMemcpyAsync(J.seq_v, flat, n0 * 4, hipMemcpyHostToDevice, s));
                PAG_HIP_TRY(hipMemcpyAsync(J.seq_s, flat + n0, n0 * 4, hipMemcpyHostToDevice, s));
            }
            P.C = tc[i];
            if (pl.win_hi) {  // a segment job: direct-mapped marks only around the segment
                P.C.in_lo = pl.win_lo;
                P.C.in_hi = pl.win_hi;
            }
            JobRef jr2;
            jr2.ctg = i;
            jr2.kind = pl.kind;
            jr2.idx = pl.idx;
            jr2.init_len = J.init_len;
            jr2.live = true;
            jr2.epoch = R.seg_epoch;
            if (pl.kind == 0) {
                if (pl.mode & TRAV_MODE_RESUME) ++n_resume_jobs;
            } else {
                ++n_seg_jobs;
                if (pl.mode & TRAV_MODE_LEAP) ++n_leap_jobs;
            }
            if (defer_ring2 && ring == 2u) {  // (first rounds before the walker starts: the ring order is decided later)
                deferred[i].push_back(Deferred{P, jr2});
                continue;
            }
            int r2;
            if ((r2 = commit_job(ring, P, jr2, pl.mode, pl.stop_pc))) return r2;
        }
        need_publish = true;
        return PAG_OK;
    }
    int publish() {  // after the prepared buffers are ready on the device
        if (!need_publish) return PAG_OK;
        auto tmark = [&](const char *what) {
            if (wtrace) trace.push_back(TraceEv{now_ms() - tw0, 4u, n_live, 0, 0, (uint64_t)(uintptr_t)what, 0, clears.size(), 0});
        };
        tmark("publish: begin");
        int rcl;
        if ((rcl = flush_clears())) return rcl;
        tmark("publish: clears launched");
        PAG_HIP_TRY(hipStreamSynchronize(s));
        tmark("publish: stream idle");
        for (uint32_t r = NR; r-- > 0;) __atomic_store_n(&hq->posted[r], n_posted[r], __ATOMIC_RELEASE);
        need_publish = false;
        const int rcw = walkers.g ? walkers.ensure(n_live) : PAG_OK;  // (before the first launch: pag_travel starts the waves itself)
        tmark("publish: waves");
        return rcw;
    }
};

