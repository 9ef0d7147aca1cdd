/* include/pagraph_hip.h — C ABI of the MI355X-native PAGraph hot path (libpagraph_hip.so).
 *
 * The reference (Godotcoffee/AlignGraph2, PAGraph/) has no library/FFI boundary for this path: the
 * boundary its caller sees is the `pagraph` process (AlignGraph2.py:414-427).  This header is the thin
 * C ABI *inside* our drop-in `pagraph` executable, between the host C++ (file parsing, alignment
 * bookkeeping, traversal control, FASTA writers) and the hand-written HIP kernels.  Its entry points
 * mirror the public surface of the reference's graph class that PositionProcessor / PAlgorithm call
 * (PAGraph/src/tools/graph/PABruijnGraph.hpp:90-131); each one cites what it replaces.
 *
 * Conventions: opaque handle owned by the caller; int return (0 = ok, negative = PAG_E*); plain
 * pointers and sizes only; no exceptions cross the ABI; one HIP stream per handle; a handle is not
 * thread-safe.  Every input array may live in host memory or already in device memory (HBM) —
 * `pag_build_input.on_device` says which; device-resident inputs are used in place, never copied.
 * There is NO CPU fallback: every entry point fails with PAG_ENODEV when no gfx950 device is usable.
 */
#ifndef PAGRAPH_HIP_H
#define PAGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PAG_OK 0
#define PAG_EINVAL (-22)  /* bad argument / unsupported parameter (k > 16, outer_sample > 7, ...) */
#define PAG_ENOMEM (-12)  /* device or host allocation failed */
#define PAG_ENODEV (-19)  /* no usable HIP device */
#define PAG_ERANGE (-34)  /* caller buffer too small; required sizes are reported */
#define PAG_EFAULT (-14)  /* a HIP runtime call or kernel failed; see pag_last_error() */

#define PAG_NONE 0xFFFFFFFFu

/* ---- sequences -------------------------------------------------------------------------------
 * 2-bit packed exactly like the reference's CompressedSeq (PAGraph/src/tools/seq/CompressedSeq.cpp
 * :8-38): A/other = 0, C = 1, G = 2, T = 3; base i of a sequence sits at bits 2*(i&3) of byte i>>2.
 * Every sequence starts on a 4-byte boundary of `packed`, and `packed` is padded with >= 16 readable
 * bytes after the last sequence. */
typedef struct pag_seqs {
    uint64_t n_seqs;
    const uint64_t *byte_off; /* [n_seqs] byte offset of each sequence in packed (multiple of 4) */
    const uint32_t *len;      /* [n_seqs] length in bases */
    const uint8_t *packed;
    uint64_t packed_bytes;
} pag_seqs;

/* ---- alignments ------------------------------------------------------------------------------
 * One record of a 3-line ALN file (reference AlignmentHelper.cpp:11-48) after the bookkeeping the
 * reference does per alignment at the top of its two hot loops (Aligner.tcc:40-71 / :121-152):
 * static eligibility, strand choice, coordinate flips.  The alignment columns are kept as the two
 * "diff" bit-vectors of ParseAlignTools::parseDiff (ParseAlignTools.cpp:8-26), 2 bits per column,
 * 16 columns per 32-bit word, column c at bits 2*(c&15): bit0 = queryDiff, bit1 = refDiff.
 *   00 / 11  both advance, query base is emitted        (exactAlign, ParseAlignTools.tcc:58-61)
 *   01       (queryDiff only) target advances            (:63-64)
 *   10       (refDiff only)  query base emitted, query advances (:65-67) */
typedef struct pag_aln {
    uint32_t query;    /* index of the read in `reads`, PAG_NONE if the name is unknown */
    uint32_t target;   /* contig index (pass 1) / reference index (pass 2), PAG_NONE if unknown */
    uint32_t t_begin;  /* header target interval (used by the coverage filter, Aligner.tcc:140-147) */
    uint32_t t_end;
    uint32_t q_start;  /* read-strand coordinate of the first emitted base (after flipPosition);
                          PAG_NONE when it lies outside the read (then n_valid == 0) */
    uint32_t t_start;  /* target-strand coordinate the walk starts from (after flipPosition) */
    uint32_t n_cols;   /* alignment columns */
    uint32_t n_valid;  /* leading emitted columns (walk order) whose per-base list is non-empty */
    uint64_t diff_off; /* index of the first 32-bit diff word of this alignment in pag_aln_db.diff */
    uint32_t flags;    /* PAG_ALN_* */
    uint32_t reserved;
} pag_aln;

#define PAG_ALN_REV_STRAND 1u /* positions go to the read's reverse-complement strand */
#define PAG_ALN_WALK_BACK 2u  /* exactAlign(forward = false): columns are walked last -> first */
#define PAG_ALN_ELIGIBLE 4u   /* passed the static filters (selected target, ratio, range) */

typedef struct pag_aln_db {
    uint64_t n_aln;
    const pag_aln *aln;        /* grouped by query, inside a group in the reference's list order
                                  (score-descending std::sort order, Aligner.cpp:53-55); records with
                                  query == PAG_NONE come last */
    const uint64_t *query_off; /* [n_reads + 1] range of each read's records in aln */
    const uint32_t *diff;
    uint64_t n_diff_words;
} pag_aln_db;

/* contig table for pass 1.  single_base already contains the strand offset of the selected
 * orientation: ctgSingle = single_base + ctgPos (PositionMapper::dualToSingle, PositionMapper.cpp
 * :37-42, truncated to u32 as in PositionProcessor.cpp:48-51). */
typedef struct pag_ctg {
    uint32_t len;
    uint32_t selected; /* Aligner::_ctgFilterFlag */
    uint32_t single_base;
    uint32_t multi;   /* 1 if some base of the selected orientation has more than one entry */
    uint64_t map_off; /* index of base 0 of this contig in ctg_ent_off (selected contigs only) */
} pag_ctg;

typedef struct pag_ref {
    uint32_t len;
    uint32_t accepted;    /* Aligner::_refFilterFlag */
    uint32_t single_base; /* refSingle = single_base + refPos */
    uint32_t reserved;
} pag_ref;

typedef struct pag_build_input {
    uint32_t on_device; /* 0: every pointer below is host memory; 1: every pointer is device memory */
    uint32_t n_threads; /* the reference's -t (only recorded; emission order comes from emit_order) */
    pag_seqs reads;
    const uint32_t *emit_order; /* [n_reads] read indices in canonical emission order */
    pag_aln_db read_to_ctg;     /* pass 1 */
    pag_aln_db read_to_ref;     /* pass 2 */
    uint64_t n_ctgs;
    const pag_ctg *ctgs;
    /* per selected-contig base: list of refSingle coordinates = AlignReference::_fPositions /
     * _rPositions after addExtraPosition (AlignReference.cpp:41-57, 69-79), already run through
     * PositionMapper.  Entries of base b of contig c: ctg_ent[ctg_ent_off[m + b] .. ctg_ent_off[m + b + 1])
     * with m = ctgs[c].map_off; each selected contig owns len + 1 consecutive offsets. */
    const uint32_t *ctg_ent_off;
    uint64_t n_ctg_ent_off;
    const uint32_t *ctg_ent;
    uint64_t n_ctg_ent;
    uint64_t n_refs;
    const pag_ref *refs;
    uint32_t eps;          /* --epsilon: cluster radius (PositionProcessor.cpp:128) */
    uint32_t cov_filter;   /* -v (Aligner.tcc:149) */
    uint32_t outer_sample; /* 3 (pagraph.cpp:113); 1..7 supported */
    int32_t topk_ctg;      /* -1 = all (pagraph.cpp:110) */
    int32_t topk_ref;      /* -1 = all (pagraph.cpp:112) */
    uint32_t reserved;
} pag_build_input;

/* the six numbers PositionProcessor::process prints (PositionProcessor.cpp:126-142) + sizes */
typedef struct pag_build_stats {
    uint64_t merge_edge[2];
    uint64_t total_pos[2];
    uint64_t merge_pos[2];
    uint64_t n_tuples[2]; /* position tuples emitted by pass 1 / pass 2 */
    uint64_t n_edges[2];  /* edge tuples emitted by pass 1 / pass 2 */
    uint64_t n_nodes;     /* k-mer nodes with >= 1 position */
    uint64_t n_pos;       /* vertices = clustered positions */
    uint64_t n_uniq_edges;
    double ms_extract, ms_sort, ms_cluster, ms_edges, ms_total; /* device time, HIP events */
    double ms_sort_kernel; /* average duration of one launch of the dominant sort kernel */
    uint64_t sort_records; /* records one such launch moves */
} pag_build_stats;

/* the finished graph, compact CSR, ascending k-mer code (types KMerAdjNode.hpp:19-23) */
typedef struct pag_csr {
    uint64_t n_nodes, n_pos, n_edges; /* in: capacities; out: sizes */
    uint32_t *node_code;              /* [n_nodes] */
    uint64_t *pos_off;                /* [n_nodes + 1] */
    uint32_t *pos_ctg;                /* [n_pos] DualPos.first  */
    uint32_t *pos_ref;                /* [n_pos] DualPos.second */
    uint16_t *pos_cnt;                /* [n_pos] u16 abundance (wraps, PABruijnGraph.hpp:28) */
    uint64_t *edge_off;               /* [n_nodes + 1] */
    uint32_t *edge_to;                /* [n_edges] child k-mer code */
    int32_t *edge_step;               /* [n_edges] */
} pag_csr;

typedef struct pag_graph pag_graph;

/* PABruijnGraph::PABruijnGraph (PABruijnGraph.cpp:10-45): `codes` are ALL 64-bit words of the
 * solid-set file after the first one, PLUS the first one (the header word k is ingested as a code,
 * SURVEY quirk Q1) — the caller passes the file's words verbatim; sort+unique happens here.
 * k <= 16.  Builds the 4^k-bit solid bitmap in HBM. */
pag_graph *pag_create(const uint64_t *codes, uint64_t n_codes, uint32_t k, int device_ordinal, int *err);
/* Same graph object from the solid set given as its 4^k-bit membership bitmap (bit c of word c >> 5),
 * host or device resident; n_solid = number of set bits (only reported).  Used when the set is produced
 * on the device (k-mer counting), so that it never makes a round trip through a sorted host array. */
pag_graph *pag_create_from_bitmap(const uint32_t *bits, uint64_t n_solid, uint32_t k, int bits_on_device,
                                  int device_ordinal, int *err);
void pag_destroy(pag_graph *g);
/* PABruijnGraph::availableKmerNumber (:371-373) */
uint64_t pag_solid_count(const pag_graph *g);
/* PABruijnGraph::resetAllNodes (:310-318) */
int pag_reset(pag_graph *g);
/* PositionProcessor::process (PositionProcessor.cpp:79-151): both extraction passes, mergeEdge,
 * mergeKmerPosition, sortKmerPosition — as one device pipeline. */
int pag_process(pag_graph *g, const pag_build_input *in, pag_build_stats *stats);
/* ---- the bookkeeping around the two passes, on the device ---------------------------------------
 * Aligner::mergeAlignInfHelper (PAGraph/src/tools/align/Aligner.cpp:32-56: per-query lists, std::sort by score), the static
 * filters and flipPosition at the top of parseToCtg / parseToRef (Aligner.tcc:40-71, 121-152), Aligner::simpleAlign with
 * AlignReference::insert / addExtraPosition (Aligner.cpp:97-202, AlignReference.cpp:41-79), the PositionMapper tables
 * (PositionMapper.cpp:16-42) and the emission order — from alignment records AS THE PARSER LEAVES THEM (names resolved to
 * sequence indices).  pag_prepare fills a DEVICE-resident pag_build_input whose arrays belong to the handle (valid until the
 * next pag_prepare / pag_destroy); pag_process takes it as it is. */
typedef struct pag_raw_aln {
    uint32_t query;  /* index of the query sequence (a read; a contig in the contig->reference database), PAG_NONE: unknown name */
    uint32_t target; /* index of the target sequence (contig / reference), PAG_NONE: unknown name */
    uint64_t score;  /* read databases: header column 4; contig->reference: qEnd - qBegin (MummerAlignDatabaseV2.cpp:38) */
    uint64_t q_begin, q_end, t_begin, t_end; /* header intervals (half-open, forward strand of the query / target) */
    uint64_t diff_off;                       /* first 32-bit word of the record's column classes in pag_raw_db.diff */
    uint32_t n_cols, n_emit, n_radv;         /* columns; columns that emit a query base; columns that advance the target */
    uint32_t forward;                        /* header strand column: 1 = F */
} pag_raw_aln;
typedef struct pag_raw_db {
    uint64_t n;
    const pag_raw_aln *rec; /* HOST memory, database order (AlnDb: sorted by score as the reference's databases are) */
    const uint32_t *diff;   /* column classes, 2 bits per column (host or device, pag_raw_input.bulk_on_device) */
    uint64_t n_diff_words;
} pag_raw_db;
typedef struct pag_raw_input {
    uint32_t bulk_on_device; /* where the bulk arrays live: reads.byte_off / len / packed and the three diff arrays */
    uint32_t n_threads;      /* the reference's -t: emission order (thread-major strided) */
    pag_seqs reads;
    pag_raw_db read_to_ctg, read_to_ref, ctg_to_ref;
    uint64_t n_ctgs;             /* ALL contigs of the -c file (the coordinate space is laid out over all of them) */
    const uint32_t *ctg_len;     /* host */
    const uint8_t *ctg_selected; /* host: listed in the config block (Aligner::_ctgFilterFlag) */
    const uint8_t *ctg_forward;  /* host: orientation the block lists it with */
    uint64_t n_refs;
    const uint32_t *ref_len;     /* host */
    const uint8_t *ref_accepted; /* host: the block's reference sequence (Aligner::_refFilterFlag) */
    double read_to_ctg_ratio, read_to_ref_ratio; /* 0.35 / 0.10 (pagraph.cpp:118-121) */
    uint32_t eps, cov_filter, outer_sample;
    int32_t topk_ctg, topk_ref;
    uint32_t reserved;
} pag_raw_input;
int pag_prepare(pag_graph *g, const pag_raw_input *raw, pag_build_input *out);

/* ---- one graph built by several GPUs (SURVEY §8e level 2) ------------------------------------------------------
 * The reference keeps all graph state per k-mer (node/KMerAdjNode.hpp:19-23; PABruijnGraph::mergeKmerPosition /
 * mergeEdge loop over the k-mers, PABruijnGraph.cpp:259-297) and extracts read by read (PositionProcessor.cpp:86-124),
 * so one config block splits as: reads over the shards for the extraction, k-mer ranges over the shards for everything
 * after it, ONE exchange of tuples in between.  With n_shards a power of two (<= 8):
 *   shard s extracts the reads at emission positions [s * n / S, (s + 1) * n / S) of in->emit_order;
 *   owner(code) = code >> (2k - log2 S): contiguous, ascending k-mer ranges;
 *   the canonical order of a k-mer's tuples is [pass 1, emission order] ++ [pass 2, emission order]; the shards' read
 *     ranges are contiguous in emission order, so the owner restores it by concatenating what it receives as
 *     [pass 1 from shard 0] .. [pass 1 from shard S-1] [pass 2 from shard 0] .. and sorting stably by k-mer — no sequence
 *     numbers travel;
 *   the finished graph is the concatenation of the owners' slices in owner order (the count lines are sums over
 *     k-mers, hence over owners).
 * pag_shard_extract: both extraction passes for this shard's reads, then a stable partition by owner.  counts[o * 4 + ..]
 *   = {tuples pass 1, tuples pass 2, edges pass 1, edges pass 2} destined for owner o; in the partitioned streams the
 *   part of owner o is [its pass-1 records][its pass-2 records], owners ascending.
 * pag_shard_take: copies the partitioned streams into caller (device) buffers of the sizes just reported.
 * pag_shard_take_part: ... a stretch of them only — records [t_off, t_off + t_n) of the tuple stream, [e_off, e_off + e_n) of
 *   the edge stream (e.g. ONE owner's pass-1 records: the offsets follow from `counts`) — for a caller that gathers one
 *   owner's records from several extractions without holding the others' (aligngraph2_amd/rank_serial.py).
 * pag_shard_build: the received records (device pointers, already in the order described above; t1 / e1 = how many
 *   of them are pass-1 records) -> K2 sort, K3, K4; the handle then holds the slice of this owner.
 * pag_shard_export / pag_shard_take_slice / pag_shard_import: the slice as device arrays; a whole graph from the slices of all owners (device
 *   pointers, in owner order): the handle then is indistinguishable from one that ran pag_process on the whole input
 *   (pag_export_csr, pag_travel ...). */
typedef struct pag_shard_slice {
    uint64_t n_t, n_e;      /* records in the tuple / edge stream arrays */
    const uint32_t *tkey;   /* [n_t] k-mer code, ascending */
    const uint64_t *tval;   /* [n_t] in place: the clustered, sorted positions of the segment at its head */
    const uint32_t *tseg;   /* [n_t] leaders of the segment starting here (0 off a segment head) */
    const uint16_t *tcnt;   /* [n_t] abundance of the leader stored here */
    const uint32_t *ekey;   /* [n_e] */
    const uint64_t *eval;   /* [n_e] */
    const uint32_t *eseg;   /* [n_e] */
    pag_build_stats stats;  /* this owner's share of every count */
} pag_shard_slice;
int pag_shard_extract(pag_graph *g, const pag_build_input *in, uint32_t shard, uint32_t n_shards, uint64_t *counts);
/* ... for the reads at emission positions [emit_lo, emit_hi) — a CHUNK of a shard's range: pag_shard_run extracts its reads in
 * chunks and sends chunk c to the owners while chunk c + 1 is extracted (SURVEY.md 8e "overlap with extraction by chunking"); the
 * chunks of a shard, one behind the other, are the shard's range, so the owner's layout argument above holds chunk by chunk. */
int pag_shard_extract_range(pag_graph *g, const pag_build_input *in, uint64_t emit_lo, uint64_t emit_hi, uint32_t n_shards, uint64_t *counts);
int pag_shard_take(pag_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *ekey, uint64_t *eval);
int pag_shard_take_part(pag_graph *g, uint64_t t_off, uint64_t t_n, uint32_t *tkey, uint64_t *tval, uint64_t e_off, uint64_t e_n, uint32_t *ekey,
                        uint64_t *eval);
int pag_shard_build(pag_graph *g, const uint32_t *tkey, const uint64_t *tval, uint64_t n_t, uint64_t t1, const uint32_t *ekey,
                    const uint64_t *eval, uint64_t n_e, uint64_t e1, uint32_t eps, pag_build_stats *stats);
int pag_shard_export(const pag_graph *g, pag_shard_slice *out);
/* ... copied into caller (device) buffers of the sizes pag_shard_export reports (they outlive the handle's own storage,
 * e.g. through an all-gather) */
int pag_shard_take_slice(pag_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *tseg, uint16_t *tcnt, uint32_t *ekey, uint64_t *eval,
                         uint32_t *eseg);
int pag_shard_import(pag_graph *g, const pag_shard_slice *parts, uint32_t n_parts, pag_build_stats *total);
/* The traversal side partitioned too (a block larger than one GPU's memory: BASELINE configs[2]).  Contigs are traversed
 * independently (PAssembly.cpp:30-79), so a rank only needs the vertices the traversals it was dealt can examine:
 *   - every vertex whose contig coordinate lies on the traversed strands of its contigs, or in the landing zone (the first
 *     1 - startSplit of a strand) of ANY contig strand — a vertex with a contig coordinate anywhere else is dropped by
 *     every classification that meets it (leap rule, PAlgorithm.tcc:60-67), present or not;
 *   - the vertices WITHOUT a contig coordinate whose reference coordinate lies in the bands its contigs map to, plus a halo.
 * pag_shard_select (on the owner, after pag_shard_build): the part of the owner's slice inside a region, in slice layout
 *   (valid until the next select on the handle; stats = the owner's share of the count lines + the selected sizes).
 * pag_shard_import takes the selections of all owners (owner order) exactly like whole slices;
 * pag_shard_set_region then tells the handle which reference bands it holds: a walk that comes within one successor's
 *   reach (max step x (1 + error rate) + deviation) of an OPEN band end makes pag_travel fail with PAG_ERANGE — never a
 *   silently different path.
 * Intervals: [lo, hi) pairs of single coordinates, sorted, disjoint; host memory. */
typedef struct pag_region {
    uint64_t n_ctg_iv;
    const uint32_t *ctg_iv;  /* [2 * n_ctg_iv] contig single coordinates (PositionMapper over the contigs) */
    uint64_t n_ref_iv;
    const uint32_t *ref_iv;  /* [2 * n_ref_iv] reference single coordinates of coordinate-free vertices */
    const uint8_t *ref_open; /* [2 * n_ref_iv] 1: the block goes on beyond that end of the band (0: end of the reference) */
} pag_region;
int pag_shard_select(pag_graph *g, const pag_region *region, pag_shard_slice *out);
int pag_shard_set_region(pag_graph *g, const pag_region *region);
/* hands back the device memory of the build stages (inputs, streams, scratch, the owner's slice and selections) once the
 * rank has imported what it traverses; the imported graph stays.  The solid set stays too. */
int pag_shard_release_build(pag_graph *g);
/* (internal to the library's own exchange: the graph arrays were received into the import buffers) */
int pag_shard_adopt(pag_graph *g, uint64_t n_t, uint64_t n_e, const pag_build_stats *stats);

/* ---- the communicator of a sharded run and the whole sharded build behind one call --------------------------------
 * One process per GPU of ONE node.  rendezvous_dir: a directory all ranks see (e.g. under /dev/shm); small host tables and
 * the RCCL unique id go through files in it.  The directory need not be fresh: the ranks of a job agree on a job nonce when
 * the communicator is created (only LIVE processes count as ranks) and every file of the job carries it in its name, so
 * what a crashed or concurrent job left there is never read.  A rank that fails calls pag_comm_abort (pag_shard_run does
 * it itself): its peers return PAG_EFAULT at once, with that rank's message in pag_last_error(), instead of waiting for
 * PAG_COMM_TIMEOUT_S; a bulk exchange over RCCL is only entered by a complete set of ranks that met through the directory.  transport "rccl" (NULL = default): the bulk all-to-all(v)s of
 * device buffers are grouped ncclSend / ncclRecv over xGMI (librccl loaded at run time); "host": through files of the
 * rendezvous directory — for ranks that share ONE device, where RCCL refuses to work (single-GPU test boxes).
 * Every rank calls the collectives in the same order. */
typedef struct pag_comm pag_comm;
pag_comm *pag_comm_create(int rank, int world, const char *rendezvous_dir, int device_ordinal, const char *transport, int *err);
void pag_comm_destroy(pag_comm *c);
int pag_comm_rank(const pag_comm *c);
int pag_comm_world(const pag_comm *c);
uint64_t pag_comm_bytes_sent(const pag_comm *c); /* payload of the bulk exchanges that left this rank so far */
void pag_comm_abort(pag_comm *c, const char *message);
int pag_comm_barrier(pag_comm *c);
int pag_comm_all_gather(pag_comm *c, const void *mine_host, uint64_t bytes, void *all_host);
int pag_comm_gather_v(pag_comm *c, const void *mine_host, uint64_t bytes, int root, void *out_host, uint64_t out_cap, uint64_t *sizes,
                      uint64_t *need);
int pag_comm_all_to_all_v(pag_comm *c, const void *send_dev, const uint64_t *send_bytes, void *recv_dev, const uint64_t *recv_bytes);
/* pag_process for ONE block over all ranks of `c` (every rank passes the same prepared input): own read range extracted,
 * tuples to their k-mer owners, K2-K4 on the owned range, every rank's region (regions[world], the same array on all
 * ranks: pag_shard_select) sent to it and received straight into the handle's graph; the handle then holds what THIS
 * rank's traversals need (pag_shard_set_region done, the build's memory released).  total: the block's count lines. */
int pag_shard_run(pag_graph *g, pag_comm *c, const pag_build_input *in, const pag_region *regions, pag_build_stats *total);

/* sizes of the finished graph, then the graph itself into caller buffers */
int pag_csr_sizes(const pag_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges);
int pag_export_csr(const pag_graph *g, pag_csr *out);
/* ---- traversal ----------------------------------------------------------------------------------
 * PAlgorithm::travelSequence (PAGraph/src/tools/graph/PAlgorithm.cpp:144-426) for every selected contig of
 * the block at once: seed search, graphTravel / walkStraight / classifySuccessors with the epsilon-join
 * predicates run on the device (one wavefront per (contig, seed) walk); the outer per-round bookkeeping
 * (longest / leaping pick, appendSeq, repeat detection, re-seeding order, filterSequence) is host code
 * inside the library.  The graph never leaves HBM; only the chosen paths come back. */
typedef struct pag_path_node {
    uint32_t code;  /* k-mer code of the vertex' node */
    uint32_t ctg;   /* DualPos.first  */
    uint32_t ref;   /* DualPos.second */
    uint16_t cnt;   /* abundance */
    uint16_t reserved;
    int32_t step;   /* distance from the previous path vertex (k for the first) */
    uint32_t vid;   /* dense vertex id (stable for the lifetime of the built graph) */
} pag_path_node;

typedef struct pag_travel_params {
    uint32_t ref_threads; /* the reference's -t: seed top-K = min(t, 8) (PAlgorithm.cpp:146) */
    uint32_t reserved;
    uint64_t deviation;   /* 2 * epsilon (pagraph.cpp:251) */
    double error_rate;    /* 0.15 */
    double start_split;   /* 0.90 */
    uint64_t min_len;     /* -l */
} pag_travel_params;

typedef struct pag_travel_stats {
    double ms_compact, ms_walk, ms_total; /* device + host wall, ms */
    uint64_t rounds, jobs, walk_steps;
    uint64_t classify_calls, probes, records; /* successor evaluations, walkStraight calls, records read */
} pag_travel_stats;

/* orientations in which a contig is traversed (config.txt lists (name, 1|0) pairs; the reference walks every pair of
 * its std::set, so a contig listed with both orientations is walked twice: PAssembly.cpp:28-36) */
#define PAG_ORIENT_NONE (-1)
#define PAG_ORIENT_REVERSE 0
#define PAG_ORIENT_FORWARD 1
#define PAG_ORIENT_BOTH 2

/* ctgs: HOST memory (2-bit packed); orient[i]: PAG_ORIENT_*.
 * ref_len[n_refs]: lengths of the reference sequences (their PositionMapper is needed for the repeat check).
 * After success, pag_travel_path_oriented(g, i, forward, &len) returns the path of contig i in that orientation
 * (library-owned PINNED memory, valid until the next pag_travel or pag_destroy on the handle — NOT invalidated by
 * pag_reset / pag_prepare / pag_process / pag_travel_prepare / pag_reserve_walk_arena, which the drivers run for the next
 * block while the previous block's host half still reads these paths; NULL / 0 if that orientation was not traversed); pag_travel_path(g, i, &len) = the forward path if there is one, else the reverse. */
/* The first part of pag_travel on its own: the traversal's view of the graph (compact CSR, coordinate order, successor
 * records), kept in the handle until the graph changes; a following pag_travel with the same parameters goes straight to the
 * walks.  *ms (may be NULL) receives its wall time. */
int pag_travel_prepare(pag_graph *g, const pag_seqs *ctgs, const uint32_t *ref_len, uint64_t n_refs, const pag_travel_params *prm,
                       double *ms);
/* ... for the traversals pag_travel will then be asked for (orient[i]: PAG_ORIENT_* as pag_travel takes it).  The view is
 * built from what THOSE traversals can examine — the traversed strand of every contig, the landing zones of all strands
 * (PAlgorithm.tcc:60-67), the vertices without a contig coordinate along the reference stretch the last tenth of every
 * traversed strand maps to, where a walk can take a Skip grade (PAlgorithm.tcc:69-86) — and the successor stage runs over
 * that alone.  Outputs are those of the whole graph: what is left out cannot change a classification, and a walk that comes
 * within a successor's reach of something left out is detected and pag_travel walks again on the whole graph's view.  A
 * pag_travel whose orientations the view does not cover rebuilds it.  PAG_TRAVEL_VIEW=whole switches the cut off. */
int pag_travel_prepare_for(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
                           const pag_travel_params *prm, double *ms);
/* sizes of the prepared view (after pag_travel_prepare* / pag_travel): nodes, vertices, edges, successor records; *cut = 1 if
 * it was built for given orientations only; *fallbacks = walks-again on the whole graph since the handle was created */
int pag_travel_view_sizes(const pag_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges, uint64_t *n_succ, int *cut,
                          uint64_t *fallbacks);
int pag_travel(pag_graph *g, const pag_seqs *ctgs, const int32_t *orient, const uint32_t *ref_len, uint64_t n_refs,
               const pag_travel_params *params, pag_travel_stats *stats);
/* One vertex's graded successors as the reference's graph returns them — PABruijnGraph::successors (PABruijnGraph.cpp:370-373 ->
 * searchSuccessors :167-197: every position of every child with checkPosition != Oops; the grade and isEdgeSimilar().first
 * its caller computes per entry, PAlgorithm.tcc:45-58, come with it) with the deviation / error rate the view was prepared
 * with (pag_travel_prepare*), in the reference's order.  A vertex is named the way the reference's PANode names it, by value:
 * its k-mer code and its clustered position (contig coordinate << 32 | reference coordinate).  Returns the number of
 * successors (the first `cap` are written); PAG_EINVAL when the view is not prepared or holds no such vertex; PAG_ERANGE when
 * the view was cut for given traversals (pag_travel_prepare_for, a regional graph) and left this vertex's successors out. */
typedef struct pag_succ {
    uint32_t code;        /* target k-mer */
    uint32_t step;        /* edge step (KMerAdjEdge) */
    uint64_t pos;         /* target position: contig coordinate << 32 | reference coordinate */
    uint32_t grade;       /* checkPosition's grade as the reference numbers it (PABruijnGraph.hpp PositionGrade) */
    uint32_t ctg_similar; /* isEdgeSimilar().first */
} pag_succ;
int64_t pag_successors(const pag_graph *g, uint32_t code, uint64_t pos, pag_succ *out, uint64_t cap);
const pag_path_node *pag_travel_path(const pag_graph *g, uint64_t ctg_index, uint64_t *len);
const pag_path_node *pag_travel_path_oriented(const pag_graph *g, uint64_t ctg_index, int forward, uint64_t *len);
/* Optional: have the device memory the walks of pag_travel take their job buffers from (one arena per handle, kept between
 * calls) allocated NOW, for contigs of `contig_bases` bases in total.  May be called from another thread while the caller
 * parses its inputs (no other call on the handle may run at the same time): a first large allocation of a process can take
 * seconds.  pag_travel sizes the arena itself when this was not called or asked for too little. */
int pag_reserve_walk_arena(pag_graph *g, uint64_t contig_bases);
/* ---- kmer_counter on the device (replaces PAGraph/src/main/kmer_counter.cpp:19-96) ----------------
 * Counts every k-mer of the forward strand of every read (KmerHelper::kmer2Code, KmerHelper.cpp:7-25) in a dense 4^k
 * table, derives the minimum abundance by the reference's rule (the first occurring abundance a, ascending, with
 * 1 - #{codes with abundance <= a} / 4^k <= threshold; 0 if none) and returns the solid set {abundance >= minimum}
 * as a 4^k-bit bitmap (bit c of word c >> 5) — the form pag_create_from_bitmap() takes, and what the kmer_counter
 * executable of this package writes out as the reference's `-k` file.
 * reads: pag_seqs in host memory (reads_on_device = 0) or device memory (1; every read 4-byte aligned and followed by
 * at least 8 readable bytes).  bitmap: 4^k / 8 bytes (at least 4), host or device (bitmap_on_device).  k = 1..16
 * (the reference's own table size overflows at k = 16, quirk Q13; here k = 16 means 2^32 codes). */
/* ---- a block's TEXT to the packed forms the build takes, on the device (first half) --------------------
 * text: the bytes of a FASTA / FASTQ / ALN file as they lie in the file, in device memory (text_on_device = 1) or in host
 * memory (0: streamed to the device through pinned staging buffers).  The caller has found the lines (and, for ALN
 * records, parsed the header fields): these calls do the bulk work.
 * pag_pack_text_seqs: sequence i = seq_len[i] characters at text + seq_off[i] -> 2-bit packed at packed_dev + byte_off[i]
 * (byte_off a multiple of 4; ceil(len / 4) bytes rounded up to a multiple of 4 are written, unused bits zero) exactly as
 * CompressedSeq does (CompressedSeq.cpp:8-38): 4 bases per byte, base i at bits 2 * (i & 3), C/c = 1, G/g = 2, T/t = 3,
 * every other character 0.
 * pag_classify_columns: record i = a query row (q_len[i] characters at q_off[i]) over a reference row (r_len / r_off) ->
 * parseDiff (ParseAlignTools.cpp:8-26) as 2 bits per column of the query row, 16 columns per u32, at diff_dev +
 * diff_off[i] (ceil(q_len / 16) words): 1 = gap in the query row, 2 = gap in the reference row, 3 = the characters differ
 * (a reference row that is shorter reads as NUL there), 0 = equal; n_emit_dev[i] / n_radv_dev[i] = columns of class != 1 /
 * != 2.  The offset arrays are host memory; packed_dev, diff_dev, n_emit_dev, n_radv_dev device memory. */
int pag_pack_text_seqs(const char *text, int text_on_device, uint64_t text_bytes, const uint64_t *seq_off, const uint32_t *seq_len,
                       uint64_t n_seqs, const uint64_t *byte_off, uint8_t *packed_dev, uint64_t packed_bytes, int device);
int pag_classify_columns(const char *text, int text_on_device, uint64_t text_bytes, const uint64_t *q_off, const uint32_t *q_len,
                         const uint64_t *r_off, const uint32_t *r_len, const uint64_t *diff_off, uint64_t n_recs, uint32_t *diff_dev,
                         uint64_t n_diff_words, uint32_t *n_emit_dev, uint32_t *n_radv_dev, int device);
/* ... text and results in HOST memory (device buffers are the call's own) */
int pag_classify_columns_host(const char *text, uint64_t text_bytes, const uint64_t *q_off, const uint32_t *q_len, const uint64_t *r_off,
                              const uint32_t *r_len, const uint64_t *diff_off, uint64_t n_recs, uint32_t *diff_host, uint64_t n_diff_words,
                              uint32_t *n_emit_host, uint32_t *n_radv_host, int device);

/* ---- pa_cns on the device -----------------------------------------------------------------------------
 * The consensus step after pagraph (reference PAGraph/src/main/pa_cns.cpp:98-124 + tools/cns/AlnGraphBoost.cpp): the backbone
 * is cut into parts, every part gets a partial-order alignment graph — AlnGraphBoost(backbone) :16-39, addAln :64-113 for its
 * alignments in the caller's order (score order, AlignData::weightAln weights), mergeNodes :137-275, bestPath :383-467,
 * consensus :293-333 — and yields its consensus string.  One device thread per part; a part's graph lives in regions the
 * caller sizes (slots, not bytes): node_cap >= bb_len + 2 + insertion columns, edge_cap >= bb_len + 1 + columns that are not
 * query deletions + n_aln (+ slack: a merge creates the surviving node's edges before it clears the merged one), aux_cap =
 * queue + stack words, out_cap >= node_cap.  The alignments are gap-normalised rows (dagcon normalizeGaps,
 * cns/Alignment.cpp:134-215) in two pools at the same offsets.  Results: out[out_off[p] .. + out_len[p]) = part p's
 * consensus (out_off[n_parts] = bytes used), part_err[p] != 0: the part ran out of one of its regions (1 nodes, 2 edges,
 * 3 queue, 4 stack, 6 output), met an alignment that runs past its part (5) or an empty edge list where the reference reads
 * .front() (7) — nothing is written for it.  Returns PAG_ENODEV without a device: there is no host fallback behind this call. */
typedef struct pag_cns_aln {
    uint64_t str_off; /* offset of its rows in qpool / tpool */
    uint32_t len;     /* columns */
    uint32_t start;   /* 1-based backbone position of its first column inside the part */
    int32_t weight;
    uint32_t reserved;
} pag_cns_aln;
typedef struct pag_cns_part {
    uint64_t bb_off;    /* the part's slice of `backbone` */
    uint32_t bb_len;
    uint32_t n_aln;
    uint64_t aln_first; /* its alignments: alns[aln_first .. + n_aln) */
    uint32_t node_cap, edge_cap, aux_cap, out_cap;
} pag_cns_part;
int pag_cns_consensus(int device, const char *backbone, uint64_t backbone_len, const pag_cns_part *parts, uint64_t n_parts, const pag_cns_aln *alns,
                      uint64_t n_alns, const char *qpool, const char *tpool, uint64_t pool_bytes, int32_t min_weight, char *out, uint64_t out_bytes,
                      uint64_t *out_off, uint32_t *out_len, int32_t *part_err);

typedef struct pag_kmer_count_result {
    uint64_t min_abundance;
    uint64_t n_solid;
    uint64_t n_kmers_counted; /* k-mer occurrences (0 if some abundance exceeded the histogram range) */
    double ms_count, ms_select; /* device time: count + histogram; bitmap */
} pag_kmer_count_result;
int pag_kmer_count(const pag_seqs *reads, int reads_on_device, uint32_t k, double threshold, int device, uint32_t *bitmap,
                   int bitmap_on_device, pag_kmer_count_result *res);

const char *pag_last_error(void);
/* 1 if a gfx950 device is present and the code object loads */
int pag_device_available(void);
/* Brings up the HIP runtime on the device (context, first allocation, this library's code object) — the third of a second a
 * cold process otherwise spends inside its first pag_create().  A caller with input files to parse calls it on a thread of its
 * own at start-up (bin/pagraph does).  Thread-safe with respect to the other entry points; PAG_OK or an error code. */
int pag_device_warm(int device_ordinal);

#ifdef __cplusplus
}
#endif
#endif /* PAGRAPH_HIP_H */
