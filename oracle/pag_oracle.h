/* oracle/pag_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's graph build for the PAGraph hot path, taking the same
 * flat input and producing the same flat output as the HIP library (include/pagraph_hip.h), so parity
 * tests can hand both the same bytes and compare results bit for bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.  It is
 * never linked into the product (`pagraph`, libpagraph_hip.so).
 *
 * Pinning: oracle/pag_oracle.c is checked against the compiled reference itself — the complete graph
 * dumped by oracle/_ref/graph_dump (the reference's own PositionProcessor / PABruijnGraph classes) on
 * every fixture under tests/golden/, see tests/test_oracle_golden.py.
 */
#ifndef PAG_ORACLE_H
#define PAG_ORACLE_H

#include "pagraph_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pago_graph pago_graph;

pago_graph *pago_create(const uint64_t *codes, uint64_t n_codes, uint32_t k);
void pago_destroy(pago_graph *g);
uint64_t pago_solid_count(const pago_graph *g);
int pago_reset(pago_graph *g);
int pago_process(pago_graph *g, const pag_build_input *in, pag_build_stats *stats);
int pago_csr_sizes(const pago_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges);
int pago_export_csr(const pago_graph *g, pag_csr *out);

/* emitted streams in canonical order (for stage-by-stage comparison with the HIP library's
 * pag_debug_streams): tval = ctgSingle << 32 | refSingle; eval = toCode << 32 | step << 1 | pass */
void pago_debug_enable(pago_graph *g, int on);
int pago_debug_stream_sizes(const pago_graph *g, uint64_t *n_tuples, uint64_t *n_edges);
int pago_debug_streams(const pago_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *ekey, uint64_t *eval);
int pago_debug_stream_reads(const pago_graph *g, uint32_t *tread, uint32_t *eread);

/* function-level seams, exposed for known-answer tests */
/* KmerHelper::kmer2Code (KmerHelper.cpp:7-25): writes len-k+1 codes, returns the count */
uint64_t pago_kmer_codes(const char *seq, uint64_t len, uint32_t k, uint64_t *out);
/* PABruijnGraph::isPosSimilar + the zero rule of mergeKmerPosition (PABruijnGraph.cpp:259-274, 379-383) */
int pago_cluster_similar(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, uint64_t eps);
/* PABruijnGraph::checkPosition (PABruijnGraph.cpp:143-165): 0 Oops 1 Skip 2 Good 3 Excellent 4 Amazing */
int pago_check_position(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, uint32_t dist,
                        uint32_t deviation, double error_rate);
/* PABruijnGraph::isEdgeSimilar (:385-400): bit0 = contig side, bit1 = reference side */
int pago_edge_similar(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, int dist, uint64_t deviation,
                      double error_rate);

/* ---- kmer_counter (PAGraph/src/main/kmer_counter.cpp:19-96), for the device k-mer counter ----
 * reads: host pag_seqs (2-bit packed).  Counts every forward-strand k-mer (kmer2Code), applies the abundance rule
 * (:59-77) and fills the solid bitmap (4^k bits, bit c of word c >> 5).  Pinned against oracle/_ref/kmer_counter. */
int pago_kmer_count(const pag_seqs *reads, uint32_t k, double threshold, uint64_t *min_abundance, uint32_t *bitmap);
/* the file the reference writes (:79-95): k as u64, then the solid codes as u64, thread t of `threads` contributing
 * the codes c with c % threads == t in ascending order, t = 0 .. threads-1.  Returns the number of u64 words; writes
 * them to out when out != NULL (capacity cap words). */
uint64_t pago_kmer_file_words(const uint32_t *bitmap, uint32_t k, uint32_t threads, uint64_t *out, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
