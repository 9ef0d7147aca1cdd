#include "hip_backend.hpp"

#include <stdexcept>
#include <string>

namespace pagh {

namespace {

class HipBackend final : public GraphBackend {
public:
    explicit HipBackend(int device) : device_(device) {}
    ~HipBackend() override { pag_destroy(g_); }
    const char *name() const override { return "HIP gfx950"; }
    void create(const std::vector<std::uint64_t> &words, unsigned k) override {
        int err = 0;
        g_ = pag_create(words.data(), words.size(), k, device_, &err);
        if (!g_) throw std::runtime_error(std::string("pag_create failed (") + std::to_string(err) + "): " + pag_last_error());
    }
    std::uint64_t solidCount() override { return pag_solid_count(g_); }
    void reset() override { check(pag_reset(g_), "pag_reset"); }
    void process(const pag_build_input &in, pag_build_stats &stats) override { check(pag_process(g_, &in, &stats), "pag_process"); }
    void exportCsr(HostGraph &out) override {
        std::uint64_t nn = 0, np = 0, ne = 0;
        check(pag_csr_sizes(g_, &nn, &np, &ne), "pag_csr_sizes");
        out.resize(nn, np, ne);
        pag_csr csr = out.view();
        check(pag_export_csr(g_, &csr), "pag_export_csr");
    }

    bool travel(const pag_seqs &ctgs, const std::vector<int> &orient, const std::vector<std::uint32_t> &refLen,
                const pag_travel_params &params, std::vector<std::vector<pag_path_node>> &paths) override {
        std::vector<std::int32_t> o(orient.begin(), orient.end());
        check(pag_travel(g_, &ctgs, o.data(), refLen.data(), refLen.size(), &params, nullptr), "pag_travel");
        paths.assign(ctgs.n_seqs, {});
        for (std::uint64_t c = 0; c < ctgs.n_seqs; ++c) {
            std::uint64_t len = 0;
            const pag_path_node *p = pag_travel_path(g_, c, &len);
            if (p && len) paths[c].assign(p, p + len);
        }
        return true;
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
