// tests/harness/pagraph_oracle.cpp — TEST HARNESS (never part of the product).
// The product's `pagraph` driver (parsers, GraphInput, traversal, writers) with the graph build routed
// through the C ORACLE instead of the HIP library, so the host half of the program can be checked
// against the reference's golden outputs on a machine without a GPU.
#include <stdexcept>

#include <memory>

#include "graph_input.hpp"
#include "host_walk.hpp"
#include "pag_oracle.h"
#include "pagraph_driver.hpp"

namespace {
class OracleBackend final : public pagh::GraphBackend {
public:
    ~OracleBackend() override { pago_destroy(g_); }
    const char *name() const override { return "C oracle (test harness)"; }
    void create(const std::uint64_t *words, std::size_t nWords, unsigned k) override { g_ = pago_create(words, nWords, k); }
    std::uint64_t solidCount() override { return pago_solid_count(g_); }
    void reset() override { pago_reset(g_); }
    // the host restatement of the preparation stage (tests/harness/graph_input.cpp)
    void prepare(const pagh::RawInput &raw, pag_build_input &out) override {
        input_ = std::make_unique<pagh::GraphInput>(raw.reads, raw.ctgs, raw.refs, raw.readToCtg, raw.readToRef, raw.ctgToRef, raw.cfg, raw.params);
        out = input_->view();
    }
    void process(const pag_build_input &in, pag_build_stats &stats) override {
        if (pago_process(g_, &in, &stats) != PAG_OK) throw std::runtime_error("pago_process failed");
    }
    void exportCsr(pagh::HostGraph &out) override {
        std::uint64_t nn, np, ne;
        pago_csr_sizes(g_, &nn, &np, &ne);
        out.resize(nn, np, ne);
        pag_csr csr = out.view();
        if (pago_export_csr(g_, &csr) != PAG_OK) throw std::runtime_error("pago_export_csr failed");
    }

    // the walk of the harness: host restatement of PAlgorithm over the exported graph
    void travel(const TravelContext &ctx, const pag_travel_params &p, pagh::HostGraph &graph,
                std::vector<pagh::TravelSequence> &travelled) override {
        exportCsr(graph);
        graph.k = ctx.k;
        travelled = pagh::hostWalkAll(graph, ctx.contigs, ctx.refs, ctx.ctgMapper, ctx.refMapper, ctx.ctgSet, p.deviation,
                                      p.error_rate, p.start_split, p.min_len, p.ref_threads, 0);
    }

private:
    pago_graph *g_ = nullptr;
    std::unique_ptr<pagh::GraphInput> input_;
};
}  // namespace

int main(int argc, char **argv) {
    OracleBackend backend;
    return pagh::runPagraph(argc, argv, backend);
}
