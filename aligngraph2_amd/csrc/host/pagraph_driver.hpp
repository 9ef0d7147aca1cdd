// The `pagraph` program: command line, per-config-block loop, output files.
// Restates run2() of the reference main (PAGraph/src/main/pagraph.cpp:69-272) on top of a graph
// backend.  The product links exactly one backend: the HIP library (hip_backend.cpp).  The test
// harness under tests/harness/ plugs the C oracle in instead, to check parsers + traversal + writers
// against the reference's golden outputs on machines without a GPU.
#pragma once
#include <cstdint>
#include <vector>

#include <set>
#include <string>
#include <utility>

#include "host_graph.hpp"
#include "pagraph_hip.h"
#include "position_mapper.hpp"
#include "raw_input.hpp"
#include "seq_db.hpp"
#include "traversal.hpp"

namespace pagh {

class GraphBackend {
public:
    virtual ~GraphBackend() = default;
    virtual const char *name() const = 0;
    // PABruijnGraph::PABruijnGraph: every word of the solid-set file (header word included)
    virtual void create(const std::uint64_t *kmerWords, std::size_t nWords, unsigned k) = 0;
    virtual std::uint64_t solidCount() = 0;
    virtual void reset() = 0;
    // optional: get ready for traversals over contigs of that many bases in total (may run beside the input parsing)
    virtual void reserveForContigs(std::uint64_t /*bases*/) {}
    // the bookkeeping around the two passes (per-query lists, filters, flips, contig->reference map): the product backend
    // runs it on the device (pag_prepare) and returns a device-resident input; valid until the next prepare()
    virtual void prepare(const RawInput &raw, pag_build_input &out) = 0;
    virtual void process(const pag_build_input &in, pag_build_stats &stats) = 0;
    // ONE block built by several processes, one per GPU (PAGRAPH_SHARD=r/N): every rank runs the driver; rank 0 writes the
    // block's outputs.  shardRank() / shardWorld() = 0 / 1 for an ordinary run.
    virtual unsigned shardRank() const { return 0; }
    virtual unsigned shardWorld() const { return 1; }
    // a sharded run: did THIS rank walk contig `id` of the current block (it then writes that contig's path dump)
    virtual bool walksContig(std::size_t) const { return true; }
    virtual void exportCsr(HostGraph &out) = 0;
    // PAlgorithm::travelSequence for every (contig, orientation) of ctgSet (PAssembly.cpp:30-36): fills `graph` with (at
    // least) the vertices on the travel sequences and travelled[2 * contig + (reverse ? 1 : 0)].  The product backend
    // walks on the device (pag_travel); there is no host walk in the product.
    struct TravelContext {
        const SeqDb &contigs;
        const SeqDb &refs;
        const PositionMapper &ctgMapper;
        const PositionMapper &refMapper;
        const std::set<std::pair<std::string, bool>> &ctgSet;
        unsigned k;
    };
    virtual void travel(const TravelContext &ctx, const pag_travel_params &params, HostGraph &graph,
                        std::vector<TravelSequence> &travelled) = 0;
    // travel() in halves, for a backend whose walks leave the host free: travelPrepare() = the traversal's view of the
    // graph just built (successor records: device work that may run while the PREVIOUS block's travel sequences are still
    // being read), travelWalks() = the walks; they leave the travel sequences as views into memory that stays valid
    // until the next travelWalks().  The driver then runs the block's host half (buildPathGraph, assemble) on a thread of
    // its own beside the next block's device work.
    struct TravelViews {
        std::vector<std::pair<const pag_path_node *, std::uint64_t>> views;  // 2 * contig + (reverse ? 1 : 0)
        std::vector<char> gathered;                                          // (a sharded run: what the other ranks sent)
    };
    virtual bool travelsInHalves() const { return false; }
    virtual void travelPrepare(const TravelContext &, const pag_travel_params &) {}
    virtual void travelWalks(const TravelContext &, const pag_travel_params &, TravelViews &) {}
};

int runPagraph(int argc, char **argv, GraphBackend &backend);

}  // namespace pagh
