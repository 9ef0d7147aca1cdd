// The switches of the traversal ("Environment switches"), read from the environment ONCE per call into one
// struct: none is needed in normal use; tests and measurements set them.  Host code only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace pagdev {

struct WalkConfig {
    // ---- the view (trav_view_region)
    bool view_whole = false;          // PAG_TRAVEL_VIEW=whole: never cut the view to the walked orientations
    bool nodes_by_place = false;      // PAG_NODE_ORDER=place: the traversal graph's nodes numbered by place instead of by k-mer code (measured slower)
    uint64_t view_halo = 100000;      // PAG_VIEW_HALO
    bool view_margin_set = false;     // PAG_VIEW_MARGIN given: view_margin bases instead of max(4000, 3 % of the contig)
    uint64_t view_margin = 4000;
    // ---- successor records
    std::string succ_mode;            // PAG_SUCC_MODE=bound|twopass ("": by size); PAG_SUCC_TWO_PASS=1 = twopass
    // ---- diagnostics
    bool timing = false;              // PAGRAPH_TIMING
    bool walk_debug = false;          // PAG_WALK_DEBUG
    unsigned stitch_threads = 8;      // PAG_STITCH_THREADS: host threads for the adoptions of finished segments (1: the control thread alone)
    bool walk_trace = false;          // PAG_WALK_TRACE: jobs (device begin / end, when the host saw them), postings and decisions kept in memory, printed when the walks are over
    double idle_limit_ms = 60000.0;   // PAG_WALK_IDLE_S: watchdog of the event loop
    bool check_aggs = false;          // PAG_DEBUG_CHECK_AGGS
    int debug_seqcap = 0;             // PAG_DEBUG_SEQCAP (> 0: tiny initial walk buffers)
    int debug_ring = 0;               // PAG_DEBUG_RING (> 0: job rings of that many entries)
    // ---- pieces
    bool pieces = true;               // PAG_WALK_PIECES
    bool leap_pieces = true;          // PAG_LEAP_PIECES
    bool leap_first = true;           // PAG_LEAP_FIRST
    bool last_piece_first = true;     // PAG_LAST_PIECE_FIRST=0: the piece that runs to the end of the strand in its place among the others
    bool force_exact = false;         // PAG_WALK_EXACT
    bool orphaning = true;            // PAG_WALK_ORPHANS
    bool keep_segments = true;        // PAG_WALK_KEEP_SEGMENTS
    uint64_t seg_len = 0;             // PAG_SEG_LEN (0: 12000)
    uint64_t seg_overlap = 1500;      // PAG_SEG_OVERLAP
    double seg_tail_frac = 0.0;       // PAG_SEG_TAIL_FRAC: the last fraction of a strand's segment stretch in segments of half the length
    bool seg_safety_set = false;      // PAG_SEG_SAFETY
    uint64_t seg_safety = 0;
    uint64_t leap_seg_len = 0;        // PAG_LEAP_SEG_LEN (0: derived from the segment length)
    bool leap_left_set = false;       // PAG_LEAP_LEFT
    uint64_t leap_left = 0;
    uint64_t leap_end_div = 2;        // PAG_LEAP_END_DIV
    uint32_t post_interleave = 16;    // PAG_POST_INTERLEAVE (the share of the contig with the fewest jobs; 8 with equal shares until round 5)
    bool post_proportional = true;    // PAG_POST_PROPORTIONAL=0: equal shares of a turn for every contig (until round 5)
    double post_spread = 1.0;         // PAG_POST_SPREAD=<0..1>: the longest contig's jobs are through the ring at this fraction of the turns
    // ---- delivery of results
    bool self_clear = false;          // PAG_WALK_SELFCLEAR=1: every job clears its own marks when a wave takes it, instead of one launch per batch before the
                                      // jobs are published (2.7 ms at the head of a block's walks).  Measured slower: a lone wave clears its 4 MB in ~0.6 ms,
                                      // walks 88.2 against 85.0 ms at configs[1] (round 5)
    bool deliver_early = true;        // PAG_DELIVER_EARLY
    unsigned deliver_blocks = 24;     // PAG_GATHER_BLOCKS: grid of a delivery that runs while walk jobs are live (0: no bound).  The delivery's
                                      // thousands of waves, each with stores to host memory in flight, slowed every walker wave beside them:
                                      // 2.5 -> 3.2-5 us per classification in the last 40 ms of a block (round 5, tests/tail_clock_probe.sh)
    uint32_t succ_heavy = 64;         // PAG_SUCC_HEAVY: successor records of a vertex with more candidate pairs than this: by a whole wave (0: never)
    bool device_tail = true;          // PAG_DEVICE_TAIL: the last round of a contig that leaps is put together on the device
    unsigned pace = 0;                // PAG_WALK_PACE: decided rounds taken per look at the rings while jobs are live (0: all; measured: no gain)
    bool fetch_direct = true;         // PAG_FETCH_DIRECT
    bool fetch_tables = true;         // PAG_FETCH_TABLES
    unsigned take_threads = 1;        // PAG_TAKE_THREADS

    static bool off(const char *name) {  // set and 0
        const char *e = std::getenv(name);
        return e && std::atoi(e) == 0;
    }
    static bool u64(const char *name, uint64_t *out) {
        const char *e = std::getenv(name);
        if (!e) return false;
        *out = std::strtoull(e, nullptr, 10);
        return true;
    }
    static WalkConfig from_env() {
        WalkConfig c;
        if (const char *e = std::getenv("PAG_TRAVEL_VIEW")) c.view_whole = std::strcmp(e, "whole") == 0;
        if (const char *e = std::getenv("PAG_NODE_ORDER")) c.nodes_by_place = std::strcmp(e, "place") == 0;
        if (const char *e = std::getenv("PAG_VIEW_HALO")) c.view_halo = (uint64_t)std::max(0ll, std::atoll(e));
        if (const char *e = std::getenv("PAG_VIEW_MARGIN")) {
            c.view_margin_set = true;
            c.view_margin = (uint64_t)std::max(0ll, std::atoll(e));
        }
        if (const char *e = std::getenv("PAG_SUCC_MODE")) c.succ_mode = e;
        else if (std::getenv("PAG_SUCC_TWO_PASS")) c.succ_mode = "twopass";
        c.timing = std::getenv("PAGRAPH_TIMING") != nullptr;
        c.walk_debug = std::getenv("PAG_WALK_DEBUG") != nullptr;
        c.walk_trace = std::getenv("PAG_WALK_TRACE") != nullptr;
        if (const char *e = std::getenv("PAG_STITCH_THREADS")) c.stitch_threads = (unsigned)std::max(1, std::atoi(e));
        c.stitch_threads = std::min(c.stitch_threads, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = std::getenv("PAG_WALK_IDLE_S")) c.idle_limit_ms = std::atof(e) * 1000.0;
        c.check_aggs = std::getenv("PAG_DEBUG_CHECK_AGGS") != nullptr;
        if (const char *e = std::getenv("PAG_DEBUG_SEQCAP")) c.debug_seqcap = std::max(16, std::atoi(e));
        if (const char *e = std::getenv("PAG_DEBUG_RING")) c.debug_ring = std::max(4, std::atoi(e));
        c.pieces = !off("PAG_WALK_PIECES");
        c.leap_pieces = !off("PAG_LEAP_PIECES");
        c.leap_first = !off("PAG_LEAP_FIRST");
        c.last_piece_first = !off("PAG_LAST_PIECE_FIRST");
        c.force_exact = std::getenv("PAG_WALK_EXACT") != nullptr;
        c.orphaning = !off("PAG_WALK_ORPHANS");
        c.keep_segments = !off("PAG_WALK_KEEP_SEGMENTS");
        u64("PAG_SEG_LEN", &c.seg_len);
        u64("PAG_SEG_OVERLAP", &c.seg_overlap);
        if (const char *e = std::getenv("PAG_SEG_TAIL_FRAC")) c.seg_tail_frac = std::min(1.0, std::max(0.0, std::atof(e)));
        c.seg_safety_set = u64("PAG_SEG_SAFETY", &c.seg_safety);
        u64("PAG_LEAP_SEG_LEN", &c.leap_seg_len);
        c.leap_left_set = u64("PAG_LEAP_LEFT", &c.leap_left);
        if (u64("PAG_LEAP_END_DIV", &c.leap_end_div)) c.leap_end_div = std::max<uint64_t>(1, c.leap_end_div);
        if (const char *e = std::getenv("PAG_POST_INTERLEAVE")) c.post_interleave = (uint32_t)std::atoi(e);
        c.post_proportional = !off("PAG_POST_PROPORTIONAL");
        if (const char *e = std::getenv("PAG_POST_SPREAD")) c.post_spread = std::min(1.0, std::max(0.05, std::atof(e)));
        c.self_clear = std::getenv("PAG_WALK_SELFCLEAR") && std::atoi(std::getenv("PAG_WALK_SELFCLEAR")) != 0;
        c.deliver_early = !off("PAG_DELIVER_EARLY");
        if (const char *e = std::getenv("PAG_GATHER_BLOCKS")) c.deliver_blocks = (unsigned)std::max(0, std::atoi(e));
        if (const char *e = std::getenv("PAG_SUCC_HEAVY")) c.succ_heavy = (uint32_t)std::min(64, std::max(0, std::atoi(e)));
        c.device_tail = !off("PAG_DEVICE_TAIL");
        if (const char *e = std::getenv("PAG_WALK_PACE")) c.pace = (unsigned)std::max(0, std::atoi(e));
        c.fetch_direct = !off("PAG_FETCH_DIRECT");
        c.fetch_tables = !off("PAG_FETCH_TABLES");
        if (const char *e = std::getenv("PAG_TAKE_THREADS")) c.take_threads = (unsigned)std::max(1, std::atoi(e));
        return c;
    }
};

}  // namespace pagdev
