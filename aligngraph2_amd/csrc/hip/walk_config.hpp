// The switches of the traversal ("Environment switches"), read from the environment ONCE per call into one
// struct: none is needed in normal use; tests and measurements set them.  Host code only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "pag_device.hpp"
namespace pagdev {

struct WalkConfig {
    // ---- the view (trav_view_region)
    bool view_whole = false;          // PAG_TRAVEL_VIEW=whole: never cut the view to the walked orientations
    uint64_t view_halo = 100000;      // PAG_VIEW_HALO
    bool view_margin_set = false;     // PAG_VIEW_MARGIN given: view_margin bases instead of max(4000, 3 % of the contig)
    uint64_t view_margin = 4000;
    // ---- successor records
    uint32_t succ_heavy = 1024;       // PAG_SUCC_HEAVY: a vertex with more candidate pairs than this is done by a whole wave (0: never; tests: 4)
    // ---- diagnostics
    bool timing = false;              // PAGRAPH_TIMING
    bool walk_debug = false;          // PAG_WALK_DEBUG
    unsigned stitch_threads = 8;      // host threads for the adoptions of finished segments
    bool walk_trace = false;          // PAG_WALK_TRACE: jobs (device begin / end, when the host saw them), postings and decisions kept in memory, printed when the walks are over
    double idle_limit_ms = 60000.0;   // PAG_WALK_IDLE_S: watchdog of the event loop
    bool check_aggs = false;          // PAG_DEBUG_CHECK_AGGS
    int debug_seqcap = 0;             // PAG_DEBUG_SEQCAP (> 0: tiny initial walk buffers)
    int debug_ring = 0;               // PAG_DEBUG_RING (> 0: job rings of that many entries)
    uint64_t debug_emit_cap = 0;      // PAG_DEBUG_EMIT_CAP (> 0: the first emission stream of the successor records has that many slots: the grow-and-repeat path)
    // ---- pieces
    bool pieces = true;               // PAG_WALK_PIECES
    bool leap_pieces = true;          // PAG_LEAP_PIECES
    bool force_exact = false;         // PAG_WALK_EXACT
    uint64_t seg_len = 0;             // PAG_SEG_LEN (0: 12000)
    uint64_t seg_overlap = 1500;      // PAG_SEG_OVERLAP
    bool seg_safety_set = false;      // PAG_SEG_SAFETY
    uint64_t seg_safety = 0;
    static bool u64(const char *name, uint64_t *out) {  // the one place the traversal's switches are read from the environment
        const char *e = std::getenv(name);
        if (!e) return false;
        *out = std::strtoull(e, nullptr, 10);
        return true;
    }
    static bool given(const char *name) {
        uint64_t x;
        return u64(name, &x);
    }
    static bool off(const char *name) {  // set and 0
        uint64_t x = 1;
        return u64(name, &x) && x == 0;
    }
    static WalkConfig from_env() {
        WalkConfig c;
        if (const char *e = std::getenv("PAG_TRAVEL_VIEW")) c.view_whole = std::strcmp(e, "whole") == 0;
        u64("PAG_VIEW_HALO", &c.view_halo);
        c.view_margin_set = u64("PAG_VIEW_MARGIN", &c.view_margin);
        c.timing = env_timing();
        c.walk_debug = given("PAG_WALK_DEBUG");
        c.walk_trace = given("PAG_WALK_TRACE");
        c.stitch_threads = std::min(c.stitch_threads, std::max(1u, std::thread::hardware_concurrency()));
        uint64_t x = 0;
        if (u64("PAG_WALK_IDLE_S", &x)) c.idle_limit_ms = (double)x * 1000.0;
        c.check_aggs = given("PAG_DEBUG_CHECK_AGGS");
        if (u64("PAG_DEBUG_SEQCAP", &x)) c.debug_seqcap = (int)std::max<uint64_t>(16, std::min<uint64_t>(x, 1u << 30));
        if (u64("PAG_DEBUG_RING", &x)) c.debug_ring = (int)std::max<uint64_t>(4, std::min<uint64_t>(x, 1u << 30));
        u64("PAG_DEBUG_EMIT_CAP", &c.debug_emit_cap);
        c.pieces = !off("PAG_WALK_PIECES");
        c.leap_pieces = !off("PAG_LEAP_PIECES");
        c.force_exact = given("PAG_WALK_EXACT");
        u64("PAG_SEG_LEN", &c.seg_len);
        u64("PAG_SEG_OVERLAP", &c.seg_overlap);
        c.seg_safety_set = u64("PAG_SEG_SAFETY", &c.seg_safety);
        if (u64("PAG_SUCC_HEAVY", &x)) c.succ_heavy = (uint32_t)std::min<uint64_t>(1u << 24, x);
        return c;
    }
};

}  // namespace pagdev
