// The `pagraph` program: command line, per-config-block loop, output files.
// Restates run2() of the reference main (PAGraph/src/main/pagraph.cpp:69-272) on top of a graph
// backend.  The product links exactly one backend: the HIP library (hip_backend.cpp).  The test
// harness under tests/harness/ plugs the C oracle in instead, to check parsers + traversal + writers
// against the reference's golden outputs on machines without a GPU.
#pragma once
#include <cstdint>
#include <vector>

#include "host_graph.hpp"
#include "pagraph_hip.h"

namespace pagh {

class GraphBackend {
public:
    virtual ~GraphBackend() = default;
    virtual const char *name() const = 0;
    // PABruijnGraph::PABruijnGraph: every word of the solid-set file (header word included)
    virtual void create(const std::vector<std::uint64_t> &kmerWords, unsigned k) = 0;
    virtual std::uint64_t solidCount() = 0;
    virtual void reset() = 0;
    virtual void process(const pag_build_input &in, pag_build_stats &stats) = 0;
    virtual void exportCsr(HostGraph &out) = 0;
    // device traversal (PAlgorithm::travelSequence for every selected contig); backends without one
    // return false and the driver walks the exported graph on the host instead
    virtual bool travel(const pag_seqs & /*ctgs*/, const std::vector<int> & /*orient*/, const std::vector<std::uint32_t> & /*refLen*/,
                        const pag_travel_params & /*params*/, std::vector<std::vector<pag_path_node>> & /*paths*/) {
        return false;
    }
};

int runPagraph(int argc, char **argv, GraphBackend &backend);

}  // namespace pagh
