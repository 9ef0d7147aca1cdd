// Internal types of the device traversal (k5_travel.hip kernels <-> k5_travel_host.hip orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pagraph_hip.h"

namespace pagdev {

// probe slots of a walker wave (lane groups); sizes the per-job stamp arrays, outside sets and arena shares
constexpr int TRAV_PROBE_GROUPS = 8;

// compact CSR of the finished graph, dense ids: node = k-mer with >= 1 vertex, by ascending code; vertex = clustered position
// (node-major, inside a node ascending (ctg, ref))
struct TravGraph {
    uint64_t n_nodes, n_pos, n_edges;
    uint32_t *ncode;      // [n_nodes]
    uint32_t *npos_off;   // [n_nodes + 1] first vertex of each node
    uint32_t *nedge_off;  // [n_nodes + 1]
    uint64_t *vpos;       // [n_pos] ctg << 32 | ref
    uint16_t *vcnt;       // [n_pos]
    uint32_t *vnode;      // [n_pos]
    uint32_t *eto;        // [n_edges] first position (k-mer-major vertex id) of the child node; PAG_NONE: the child has no node
    uint32_t *estep;      // [n_edges] step | min(number of the child's positions, 255) << 24 (k5_travel.hip edge_target)
    uint64_t *bitmap;     // 4^k bits: k-mer code owns a node
    uint32_t *rank;       // per 64-bit bitmap word: nodes before it
    // coordinate order ("new ids" u): [ctg == 0 vertices] ++ [ctg != 0 ascending]
    uint32_t *uold;       // [n_pos] new id -> vertex id
    uint32_t *newid;      // [n_pos] vertex id -> new id
    uint64_t *upos;       // [n_pos] positions in new order
    uint32_t *ucnt;       // [n_pos] abundance in new order
    uint32_t *succ_off;   // [n_pos + 1] successor records of new id u
    struct SuccRec *succ; // [n_succ]
    uint64_t n_succ;
    // a graph that holds only a REGION of the block's vertices (pag_shard_select, one rank of a sharded build): bit u set =
    // successors of vertex u may lie outside the region (its reference coordinate is near an open band end, or — a vertex
    // selected by its contig coordinate — in no band of this rank at all).  Such a vertex has ONE record of grade
    // GRADE_POISON instead of its successors; a walk that examines it reports a fault (TravJobOut::poison).
    const uint32_t *incomplete;  // null: the whole graph is here
    uint32_t n_zero;
    // What k_mark_incomplete tested, kept so that a kernel whose threads run over the vertices in ANOTHER order (the successor
    // kernels: k-mer-major) can repeat the test from the vertex's own position instead of gathering the bit at a random place:
    // the reference bands [inc_iv[2 i], inc_iv[2 i + 1]) of the region, ascending, whether either end of a band is open
    // (inc_open[2 i], [2 i + 1]) and the margin a successor's coordinate can lie beyond its source's.  inc_n = 0 with
    // `incomplete` set: every vertex with a reference coordinate is marked.  (device pointers)
    const uint32_t *inc_iv;
    const uint8_t *inc_open;
    uint32_t inc_n, inc_margin;
};
constexpr uint32_t GRADE_POISON = 7u, GRADE_POISON_IF_LEAP = 6u;  // (marker records, see k_succ / walk_note_record in k5_travel.hip)

// one graded successor of a vertex (searchSuccessors + checkPosition != Oops), in reference order
struct alignas(16) SuccRec {
    uint32_t tgt;   // new id
    uint32_t pc;    // contig coordinate of the target
    uint32_t meta;  // step (24 bits) | grade << 24 | isEdgeSimilar().first << 27 | min(#successors of tgt, 15) << 28
    uint32_t toff;  // first successor record of the target (so a walk never has to read the offset table)
};

struct TravContig {
    const uint32_t *nodes;  // [n_kmers] node id of every k-mer of the traversed contig strand, PAG_NONE if absent
    uint32_t n_kmers;
    uint32_t ctg_left, ctg_right;  // single-coordinate range of the traversed strand
    uint32_t rev_left, rev_right;  // ... of the opposite strand (excluded, PAlgorithm.cpp:176-178)
    uint64_t split_size;           // ctgLen * startSplit
    double leap_min;               // 1 - startSplit
    const uint64_t *starts;        // contig PositionMapper start table [n_ctgs + 1]
    const uint64_t *sizes;         // [n_ctgs]
    uint32_t n_ctgs;
    uint32_t in_lo, in_hi;  // new-id range covered by the per-job arrays (stamps, travel epochs) and the LDS window: the whole
                            // strand (filled by k_ranges) or, for a segment job, the part of it around the segment; a vertex
                            // outside is kept in the job's hash sets, exactly like a vertex off the strand
    uint32_t g_lo, g_hi;    // new-id range of the vertices on the traversed strand (in_lo - g_lo is a multiple of 32)
    uint32_t *gbits;        // globalUniqueTable: bitmap over [g_lo, g_hi) (null before the first commit)
    uint32_t *gset;         // ... hash set for vertices outside that range (null before the first commit)
    uint32_t gmask;
    uint32_t gwin_lo, gwin_hi;  // ctgGlobalPosTable
};

struct TravJob {
    uint32_t ctg;    // index into the TravContig array
    uint32_t start;  // seed vertex
    uint64_t has_size;
    uint32_t *seq_v, *seq_s;
    uint64_t seq_cap;
    uint32_t *arena_v, *arena_s;
    uint64_t arena_cap;
    uint32_t *stamp;  // zeroed: TRAV_PROBE_GROUPS arrays of [stamp_stride] generation stamps (walkStraight marks, one array per probe slot)
    uint32_t stamp_stride;
    uint32_t *tbits;  // zeroed: travel-visited EPOCH per strand vertex (0 = not visited; epoch = graphTravel iteration of the append)
    uint64_t *tset;   // all-ones: (vertex | epoch << 32) hash set for vertices outside the strand's id range
    uint32_t tmask;
    uint64_t *pset;
    uint32_t pmask;
    uint32_t exact;  // 1: no speculation (probes of a branch all run to their end before the choice), see Slot
    // ---- segment-parallel walking (k5_travel_host.hip, "pieces"): a graphTravel is cut into pieces along the contig
    // coordinate; every piece is one job.  mode bit 0 (TRAV_MODE_SPEC): a piece started ahead of the walk at a
    // checkpoint vertex, walked as if leaping were impossible (split size = infinity); bit 1 (TRAV_MODE_RESUME): the
    // walk continues a path whose first init_len vertices are already in seq_v / seq_s (the job marks them visited, sums
    // their steps, then classifies the last one as graphTravel would at that point).  stop_pc != 0: the job ends at the
    // first graphTravel iteration boundary whose last vertex has a contig coordinate >= stop_pc (on the own strand).
    // bit 2 (TRAV_MODE_LEAP): a piece started ahead of the walk INSIDE the leaping zone, walked as if leaping were possible
    // from its first vertex on (split size = 0), its travel window forced down to win_low (the walk it will be spliced
    // into has everything from its seed on in that window), and with a log of its iterations in seq_x.
    uint32_t mode;
    uint32_t stop_pc;
    uint64_t init_len;
    uint32_t win_low;   // TRAV_MODE_LEAP: lower end forced on the travel coordinate window
    uint32_t pad0;
    // TRAV_MODE_LEAP: [seq_cap] zeroed; entry i != 0 <=> seq_v[i] is the last vertex of a chosen path, i.e. graphTravel
    // classified it at the top level (an iteration boundary).  Entry = 1 << 63 | elow << 32 | m0 about the iteration that
    // STARTS there: elow = lowest contig coordinate (31 bits, saturated) of any successor record that follows the contig
    // (isEdgeSimilar().first) examined until the next boundary, m0 = lowest new id of any examined record whose target has
    // no contig coordinate (all ones: none); records examined by probes of earlier iterations that are still walking count.
    uint64_t *seq_x;
};
enum { TRAV_MODE_SPEC = 1, TRAV_MODE_RESUME = 2, TRAV_MODE_LEAP = 4, TRAV_MODE_UNTIL_LEAP = 8,
       TRAV_MODE_CANCELLED = 16 };  // (16: set by the host on a job nobody waits for any more: the wave that takes it reports it done at once)  // (8: with RESUME, see k_walk's stop rules)

struct TravJobOut {
    uint64_t seq_len, seq_size;
    uint64_t n_classify, n_probe, n_records;
    uint32_t last_ctg;
    int overflow;
    uint64_t n_fill, n_out, n_main;
    // what a later stitch needs to know about this piece (see k5_travel_host.hip):
    uint32_t stopped;       // 1: ended by the stop_pc rule (the graphTravel is not finished), 0: ended by itself
    uint32_t max_back;      // max over iterations of (coordinate of the branch vertex - lowest coordinate any probe of that
                            // iteration visited); the initial walkStraight counts as an iteration
    uint32_t max_chosen;    // longest chosen path appended in one iteration (vertices)
    uint32_t wd_below_max;  // window-dependent records (a coordinate, not following the contig): the highest coordinate below
                            // TravJob::win_low (0: none) ...
    uint64_t max_probe;     // largest size (sum of steps) of any probe, zombies included
    uint32_t wd_forced_min; // ... and the lowest one at or above it (all ones: none)
    uint32_t poison;        // 1: the walk examined the records of a vertex whose successors this rank does not hold (TravGraph::incomplete)
    uint64_t t_begin, t_end;  // 100 MHz device clock when the wave took the job / finished it (PAG_WALK_DEBUG timeline)
#ifdef PAG_WALK_PROF
    uint64_t prof_t[14];  // cycles per section of the walk (development aid, make WALK_PROF=1)
    uint32_t prof_c[14];
#endif
};

// queue of the persistent walker (fine-grained host memory)
struct TravPosted {
    TravJob J;
    TravContig C;  // snapshot of the contig record this job runs with
};
constexpr int TRAV_RINGS = 3;
struct TravQueue {
    uint32_t posted[TRAV_RINGS];  // per ring: job numbers below it are valid (host writes, release); ring 0 is served first
    uint32_t exit;                // host: no further jobs will be posted
    uint32_t started, exited;     // device: walker waves that have begun / left (system-scope atomics), see k_walk_persistent
};

struct TravPackDesc {  // one finished job of a fetch batch (k_pack_paths)
    const uint32_t *seq_v, *seq_s;  // first new vertex / step of the job's sequence
    uint64_t len;                   // how many
    uint64_t off;                   // word offset of the job's words in the packed buffer: 3 * len (5 * len with seq_x) + its block tables
                                    // (trav_pack_words)
    const uint64_t *seq_x;          // iteration log of a TRAV_MODE_LEAP job (null otherwise)
};

// words a job of `len` entries takes in the packed buffer (k_pack_paths): its arrays, then 5 (+ 2 for a leap job) per 64 entries
inline uint64_t trav_pack_words(uint64_t len, bool leap) { return (leap ? 5 : 3) * len + ((len + 63) / 64) * (leap ? 7 : 5); }

constexpr uint32_t TRAV_SEED_PARTS = 16;  // waves per re-seed window request (k_seed_window)

struct TravSeedReq {
    uint32_t ctg;
    uint32_t pad;
    uint64_t left, right, pos;
};

// what of a finished graph a traversal view takes (device arrays of sorted, disjoint [lo, hi) pairs): vertices with a contig
// coordinate inside civ, vertices without one whose reference coordinate lies inside riv
struct TravView {
    const uint32_t *civ, *riv;
    uint32_t n_civ, n_riv;
};
int trav_compact(const uint32_t *tkey, const uint64_t *tval, const uint32_t *tseg, const uint16_t *tcnt, uint64_t T,
                 const uint32_t *ekey, const uint64_t *eval, const uint32_t *eseg, uint64_t E, uint32_t k, uint64_t n_nodes,
                 uint64_t n_pos, uint64_t n_edges, TravGraph G, void *tmp, size_t tmp_bytes, hipStream_t s, const TravView *view = nullptr,
                 uint64_t *counts_out = nullptr);
int trav_zone_bands(const uint64_t *tval, uint64_t T, const uint32_t *zones_dev, uint32_t n_z, uint32_t *lo_dev, uint32_t *hi_dev, hipStream_t s);
size_t trav_compact_tmp_bytes(uint64_t T, uint64_t E, uint32_t k, uint64_t n_nodes);
struct TravCtgNodesJob {  // one contig strand of a k_ctg_nodes launch
    uint64_t byte_off;  // of its packed bases
    uint64_t out_off;   // of its node ids in the output array
    uint32_t len;
    int32_t forward;
};
// (jobs: device array; one launch for all strands — 49 launches of ~30 us one behind the other until round 5)
void trav_launch_ctg_nodes(const uint8_t *packed, const TravCtgNodesJob *jobs, uint32_t n_jobs, uint32_t max_len, uint32_t k, TravGraph G, uint32_t *out,
                           hipStream_t s);
void trav_launch_seed_first(TravGraph G, const TravContig *ctgs, uint32_t n, uint64_t dev, uint32_t *out, uint32_t stride,
                            hipStream_t s);
void trav_launch_seed_window(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev,
                             uint32_t *out, uint32_t stride, hipStream_t s);
void trav_launch_checkpoints(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev, uint32_t *out,
                             hipStream_t s);  // out: 3 x u32 per request (old vertex id | PAG_NONE, contig coordinate, abundance)
void trav_launch_id_bounds(TravGraph G, const uint32_t *coords, uint32_t n, uint32_t *out, hipStream_t s);
void trav_launch_pack_paths(TravGraph G, const TravPackDesc *descs, uint32_t n, uint64_t max_len, uint32_t *out, hipStream_t s);
void trav_launch_gather_pc(TravGraph G, const uint32_t *seq_v, uint64_t len, uint32_t *out, hipStream_t s);
void trav_launch_walk(TravGraph G, const TravContig *ctgs, const TravJob *jobs, TravJobOut *outs, uint32_t n, uint32_t k,
                      hipStream_t s);
int trav_walk_waves_per_cu();
void trav_launch_walk_persistent(TravGraph G, const TravPosted *jobs, TravJobOut *outs, uint32_t *done, TravQueue *q,
                                 uint32_t *next, uint32_t cap, uint32_t k, uint32_t n_waves, uint64_t idle_ticks, hipStream_t s);
void trav_launch_commit(const uint32_t *seq_v, uint64_t len, uint32_t in_lo, uint32_t in_hi, uint32_t *gbits, uint32_t *gset,
                        uint32_t gmask, hipStream_t s);
void trav_launch_ranges(TravGraph G, TravContig *ctgs, uint32_t n, hipStream_t s);
int trav_mark_incomplete(TravGraph &G, uint32_t n_zero, const uint32_t *iv_host, const uint8_t *open_host, uint32_t n_iv, uint32_t dev, double err,
                         uint32_t *bits, void *tmp, hipStream_t s);  // (sets G.incomplete and G.inc_*)
size_t trav_mark_incomplete_tmp_bytes(uint32_t n_iv);

int trav_order(TravGraph G, uint32_t *key, uint64_t *val, uint32_t *key2, uint64_t *val2, void *sort_tmp, uint64_t *n_zero, int ctg_bits,
               int ref_bits, hipStream_t s);
// device ranges cleared by one launch (the buffers of the walk jobs of a batch): `bytes` bytes at p set to the bytes of `word`
struct TravClear {
    void *p;
    uint64_t bytes;
    uint32_t word, pad;
};
int trav_clear_ranges(const TravClear *ranges_dev, size_t n, hipStream_t s);
// the records of ONE vertex named by (code, position), translated back to (code, position) of their targets (pag_successors):
// out[0] = number of records or ~0 (no such vertex) / ~1 (its list is a marker), written to recs (32 bytes each) up to cap
int trav_successors_of(TravGraph G, uint32_t code, uint64_t pos, void *recs, uint64_t cap, unsigned long long *out, hipStream_t s);
// geometry of the emission stream (k5_travel.hip, k_succ_emit): slots a wave takes at a time, and how many waves can hold a
// partly used chunk at the end (both launches) — the slack of the stream
constexpr uint32_t EMIT_CHUNK_SLOTS = 2048u, EMIT_GRID_THREADS = 2048u, EMIT_GRID_WAVES = 1024u;
constexpr uint64_t EMIT_SLACK_SLOTS = (uint64_t)(EMIT_GRID_THREADS + EMIT_GRID_WAVES) * 4u * EMIT_CHUNK_SLOTS;
// the successor records of every vertex (k5_travel.hip, k_succ_emit): one evaluation of the candidate pairs into an emission
// stream sorted by source, then the records a walk reads
int trav_succ_emit(TravGraph G, uint32_t dev, double err, uint32_t *key0, uint64_t *val0, uint32_t *key1, uint64_t *val1, uint64_t cap, void *sort_tmp,
                   unsigned long long *counters, uint32_t *heavy_list, uint32_t heavy_limit, uint64_t *n_slots, uint64_t *n_rec, uint64_t *n_heavy,
                   const uint32_t **sorted_key, const uint64_t **sorted_val, hipStream_t s);
int trav_succ_finish(TravGraph G, const uint32_t *key, const uint64_t *val, uint64_t n_rec, hipStream_t s);
// (max_blocks: 0 = as many as the path has work for; a delivery that runs beside the walks of other contigs is kept small)
void trav_launch_gather_path(TravGraph G, const uint32_t *seq_v, const uint32_t *seq_s, uint64_t len, pag_path_node *out,
                             hipStream_t s, unsigned max_blocks = 0);
void trav_launch_gather_vertices(TravGraph G, const uint32_t *vids, uint32_t n, pag_path_node *out, hipStream_t s);
// the parts of a chosen walk, lying in the sequence buffers of the jobs that walked them, put one behind the other
// (out_v / out_s + TravConcatPart::start); the step of the very first vertex becomes first_step.  `parts` may be pinned
// host memory.
struct TravConcatPart {
    const uint32_t *v, *s;
    uint64_t start, n;
};
void trav_launch_concat_parts(const TravConcatPart *parts, uint32_t n_parts, uint32_t *out_v, uint32_t *out_s, uint32_t first_step,
                              hipStream_t s);

}  // namespace pagdev
