// tests/harness/pagh_walk_test.cpp — TEST SUPPORT (never part of the product).
// pagt_traverse_hostwalk: the outputs of pagh_traverse (include/pagraph_host.h) with the WALK done on the host over an
// exported copy of the device graph (pag_export_csr + tests/harness/host_walk.cpp, the restatement of the reference's
// PAlgorithm).  Slow; tests use it to cross-check the device walkers (tests/walk_check.py, test_gpu_big.py).
#include <chrono>
#include <cstdio>
#include <exception>
#include <set>
#include <string>

#include "assembly.hpp"
#include "host_graph.hpp"
#include "host_walk.hpp"
#include "pagraph_host.h"
#include "position_mapper.hpp"
#include "seq_db.hpp"

namespace {
pagh::SeqDb fromPacked(const pag_seqs *s, const char *stem, unsigned firstNo) {
    pagh::SeqDb db;
    for (std::uint64_t i = 0; i < s->n_seqs; ++i) db.addPacked(stem + std::to_string(i + firstNo), s->packed + s->byte_off[i], s->len[i]);
    db.finish();
    return db;
}
}  // namespace

extern "C" int pagt_traverse_hostwalk(pag_graph *g, uint32_t k, const pag_seqs *ctgs, const char *const *, const pag_seqs *refs,
                                      const char *const *, const int32_t *ctg_orient, uint32_t ref_threads, uint64_t epsilon,
                                      uint64_t min_len, const char *out_dir, const char *prefix, uint32_t host_threads,
                                      pagh_traverse_stats *stats) {
    try {
        pagh::SeqDb contigDb = fromPacked(ctgs, "ctg", 0), refDb = fromPacked(refs, "ref", 1);
        pagh::PositionMapper ctgMapper(contigDb), refMapper(refDb);
        std::set<std::pair<std::string, bool>> ctgSet;
        for (std::uint64_t i = 0; i < ctgs->n_seqs; ++i) {
            const int32_t o = ctg_orient[i];
            if (o == PAG_ORIENT_FORWARD || o == PAG_ORIENT_BOTH) ctgSet.emplace(contigDb.name(i), true);
            if (o == PAG_ORIENT_REVERSE || o == PAG_ORIENT_BOTH) ctgSet.emplace(contigDb.name(i), false);
        }
        pagh::HostGraph graph;
        std::uint64_t nn = 0, np = 0, ne = 0;
        int rc = pag_csr_sizes(g, &nn, &np, &ne);
        if (rc != PAG_OK) return rc;
        graph.resize(nn, np, ne);
        pag_csr csr = graph.view();
        if ((rc = pag_export_csr(g, &csr)) != PAG_OK) return rc;
        graph.k = k;
        auto travelled = pagh::hostWalkAll(graph, contigDb, refDb, ctgMapper, refMapper, ctgSet, epsilon * 2, 0.15, 0.90, min_len,
                                           ref_threads, host_threads);
        pagh::AssembleStats as;
        pagh::assemble(out_dir, prefix ? prefix : "0_", graph, contigDb, refDb, ctgMapper, refMapper, ctgSet, epsilon * 2, 0.15, 0.90,
                       min_len, ref_threads, host_threads, &as, true, travelled);
        if (stats) {
            *stats = pagh_traverse_stats{};
            stats->n_contigs = as.nContigs;
            stats->n_path_nodes = as.nPathNodes;
            stats->n_path_bases = as.nPathBases;
            stats->n_chains_emitted = as.nChains;
            stats->n_fasta_bases = as.nFastaBases;
            stats->path_checksum = as.pathChecksum;
        }
        return PAG_OK;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "pagt_traverse_hostwalk: %s\n", e.what());
        return PAG_EFAULT;
    }
}
