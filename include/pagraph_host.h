/* include/pagraph_host.h — C ABI of the host half of the PAGraph hot path (libpagraph_host.so):
 * traversal control, chain selection and the output writers, on top of a graph built by
 * libpagraph_hip.so.  Mirrors PAssembly::testTravel5 (reference PAGraph/src/tools/graph/PAssembly.cpp
 * :11-336), which the reference main calls right after PositionProcessor::process (pagraph.cpp:241-256).
 * The `pagraph` executable links the same code statically; this library exists so that non-C++ callers
 * (bench.py, tests) can drive the identical path. */
#ifndef PAGRAPH_HOST_H
#define PAGRAPH_HOST_H

#include "pagraph_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pagh_traverse_stats {
    uint64_t n_contigs;        /* contig orientations traversed */
    uint64_t n_path_nodes;     /* vertices on all kept paths */
    uint64_t n_path_bases;     /* sum of steps over all kept paths (PAlgorithm::seqSize) */
    uint64_t n_chains_emitted; /* FASTA records written */
    uint64_t n_fasta_bases;
    uint64_t path_checksum;    /* order-independent hash of (contig, path vertices): cheap cross-checks */
    double ms_export, ms_traverse, ms_total; /* host wall clock */
    /* device traversal (copied from pag_travel_stats; zero for the host walk) */
    double ms_successors, ms_walk;           /* successor records; walk event loop */
    uint64_t walk_rounds, walk_jobs;         /* longest chain of rounds over the contigs; (contig, seed) walks */
    uint64_t walk_steps, walk_classifications; /* path vertices produced; successor classifications evaluated */
} pagh_traverse_stats;

/* ctgs / refs: HOST memory, 2-bit packed (pag_seqs).  names may be NULL ("ctg<i>" / "ref<i+1>").
 * ctg_orient[i]: PAG_ORIENT_FORWARD / _REVERSE / _BOTH / _NONE (config.txt; a contig listed with both orientations is
 * traversed twice, PAssembly.cpp:28-36).
 * Writes <prefix><C>_<O>.txt/.fasta/.con/.help into out_dir exactly like the reference.
 * host_threads: worker threads for the per-contig loop (0 = hardware concurrency); results do not
 * depend on it.  ref_threads = the reference's -t (seed top-K = min(t, 8), quirk Q10). */
int pagh_traverse(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names,
                  const pag_seqs *refs, const char *const *ref_names, const int32_t *ctg_orient, uint32_t ref_threads,
                  uint64_t epsilon, uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads,
                  pagh_traverse_stats *stats);
/* pagh_traverse in two halves, so that the host half of block n (path graph, chain selection, the output files: host threads
 * only) runs beside the device work of block n + 1 (the reference's config blocks are independent, pagraph.cpp:181-182).
 * begin(): the device traversal (pag_travel); the host half then starts on a thread of its own and begin() returns.
 * end(): waits for it and hands out its statistics / error.  A begin() on a handle whose previous host half is still
 * running waits for it first (the travel sequences it reads live in pinned memory the next pag_travel reuses); pag_prepare
 * and pag_process of the next block need not wait.  The arrays passed to begin() must stay valid until end(). */
int pagh_traverse_begin(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names, const pag_seqs *refs,
                        const char *const *ref_names, const int32_t *ctg_orient, uint32_t ref_threads, uint64_t epsilon,
                        uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads);
int pagh_traverse_end(pag_graph *g, pagh_traverse_stats *stats);
/* The second half of pagh_traverse on travel sequences obtained elsewhere — e.g. walked by several GPUs, each for a part of
 * the contigs (pag_travel with the other contigs PAG_ORIENT_NONE), and gathered: paths[2 * c + (reverse ? 1 : 0)] /
 * path_len[...] = the sequence of contig c in that orientation (NULL / 0: none).  ctg_orient lists ALL contigs of the
 * block, as config.txt does.  cache_key: any pointer under which host storage is kept between calls (NULL: nothing is kept). */
int pagh_assemble_paths(pag_graph *cache_key, uint32_t k, const pag_seqs *ctgs, const char *const *ctg_names,
                        const pag_seqs *refs, const char *const *ref_names, const int32_t *ctg_orient,
                        const pag_path_node *const *paths, const uint64_t *path_len, uint32_t ref_threads, uint64_t epsilon,
                        uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads,
                        pagh_traverse_stats *stats);
/* Drops the host storage kept for a graph handle between pagh_traverse calls; waits for a host half still running.  Call it
 * before pag_destroy of a handle that was traversed. */
void pagh_release(pag_graph *g);
const char *pagh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
