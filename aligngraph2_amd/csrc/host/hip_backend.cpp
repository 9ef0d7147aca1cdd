#include "hip_backend.hpp"

#include "path_graph.hpp"

#include <stdexcept>
#include <string>

namespace pagh {

namespace {

class HipBackend final : public GraphBackend {
public:
    explicit HipBackend(int device) : device_(device) {}
    ~HipBackend() override { pag_destroy(g_); }
    const char *name() const override { return "HIP gfx950"; }
    void create(const std::uint64_t *words, std::size_t nWords, unsigned k) override {
        int err = 0;
        g_ = pag_create(words, nWords, k, device_, &err);
        if (!g_) throw std::runtime_error(std::string("pag_create failed (") + std::to_string(err) + "): " + pag_last_error());
    }
    std::uint64_t solidCount() override { return pag_solid_count(g_); }
    void reset() override { check(pag_reset(g_), "pag_reset"); }
    void reserveForContigs(std::uint64_t bases) override { check(pag_reserve_walk_arena(g_, bases), "pag_reserve_walk_arena"); }
    void prepare(const RawInput &raw, pag_build_input &out) override { check(pag_prepare(g_, &raw.view(), &out), "pag_prepare"); }
    void process(const pag_build_input &in, pag_build_stats &stats) override { check(pag_process(g_, &in, &stats), "pag_process"); }
    void exportCsr(HostGraph &out) override {
        std::uint64_t nn = 0, np = 0, ne = 0;
        check(pag_csr_sizes(g_, &nn, &np, &ne), "pag_csr_sizes");
        out.resize(nn, np, ne);
        pag_csr csr = out.view();
        check(pag_export_csr(g_, &csr), "pag_export_csr");
    }

    void travel(const TravelContext &ctx, const pag_travel_params &params, HostGraph &graph,
                std::vector<TravelSequence> &travelled) override {
        const SeqDb &contigs = ctx.contigs;
        // orientation(s) per contig as PAssembly::testTravel5 walks its ctgSet (PAssembly.cpp:28-36): a contig listed with
        // both orientations is traversed twice, as two independent entries
        std::vector<std::int32_t> orient(contigs.size(), PAG_ORIENT_NONE);
        for (auto &c : ctx.ctgSet) {
            if (!contigs.contains(c.first)) continue;
            std::int32_t &o = orient[contigs.id(c.first)];
            const std::int32_t mine = c.second ? PAG_ORIENT_FORWARD : PAG_ORIENT_REVERSE;
            o = (o == PAG_ORIENT_NONE || o == mine) ? mine : PAG_ORIENT_BOTH;
        }
        std::vector<std::uint32_t> refLen;
        for (std::size_t i = 0; i < ctx.refs.size(); ++i) refLen.push_back(ctx.refs.length(i));
        pag_seqs cs{contigs.size(), contigs.byteOff().data(), contigs.lens().data(), contigs.packed().data(), contigs.packed().size()};
        check(pag_travel(g_, &cs, orient.data(), refLen.data(), refLen.size(), &params, nullptr), "pag_travel");
        std::vector<std::pair<const pag_path_node *, std::uint64_t>> views(2 * contigs.size(), {nullptr, 0});
        for (std::uint64_t c = 0; c < contigs.size(); ++c)
            for (int rev = 0; rev < 2; ++rev) {
                std::uint64_t len = 0;
                const pag_path_node *p = pag_travel_path_oriented(g_, c, rev == 0, &len);
                if (p && len) views[2 * c + rev] = {p, len};
            }
        buildPathGraph(views, ctx.k, graph, travelled);
    }

private:
    void check(int rc, const char *what) {
        if (rc != PAG_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + pag_last_error());
    }
    int device_;
    pag_graph *g_ = nullptr;
};

}  // namespace

std::unique_ptr<GraphBackend> makeHipBackend(int deviceOrdinal) { return std::unique_ptr<GraphBackend>(new HipBackend(deviceOrdinal)); }

}  // namespace pagh
