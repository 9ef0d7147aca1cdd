#include "hip_backend.hpp"

#include "aln_db.hpp"
#include "host_threads.hpp"
#include "path_graph.hpp"
#include "shard_plan.hpp"

#include <thread>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

namespace pagh {

namespace {

class HipBackend final : public GraphBackend {
public:
    explicit HipBackend(int device) : device_(device) {
        // (the HIP runtime comes up while the driver parses its first input files)
        warm_ = std::thread([device] { (void)pag_device_warm(device); });
        try {
            configure();
        } catch (...) {  // (a joinable std::thread member in a constructor that throws would end the process without the message)
            if (warm_.joinable()) warm_.join();
            throw;
        }
    }
    void configure() {
        // PAGRAPH_SHARD=r/N (or "env": RANK / WORLD_SIZE as torchrun sets them): this process is rank r of N that build every
        // config block TOGETHER, one process per GPU of the node; PAGRAPH_SHARD_DIR = a fresh directory all of them see;
        // PAGRAPH_SHARD_TRANSPORT = rccl (default) | host (ranks sharing one device: test boxes)
        const char *sh = std::getenv("PAGRAPH_SHARD");
        if (!sh || !*sh) return;
        int r = 0, n = 1;
        if (std::strcmp(sh, "env") == 0) {
            r = static_cast<int>(envInt("RANK", 0));
            n = static_cast<int>(envInt("WORLD_SIZE", 1));
        } else if (std::sscanf(sh, "%d/%d", &r, &n) != 2) {
            throw std::runtime_error("PAGRAPH_SHARD must be r/N or env");
        }
        if (n < 1 || (n == 1 && !std::getenv("PAG_COMM_FORCE_RCCL"))) return;  // (one rank takes the sharded path only as a test of it: RCCL with itself)
        const char *dir = std::getenv("PAGRAPH_SHARD_DIR");
        if (!dir) throw std::runtime_error("PAGRAPH_SHARD needs PAGRAPH_SHARD_DIR (a fresh directory every rank sees)");
        if (n != 1 && n != 2 && n != 4 && n != 8) throw std::runtime_error("PAGRAPH_SHARD: 1, 2, 4 or 8 ranks");  // (1: the sharded path with itself, a test aid)
        int err = 0;
        comm_ = pag_comm_create(r, n, dir, device_, std::getenv("PAGRAPH_SHARD_TRANSPORT"), &err);
        if (!comm_) throw std::runtime_error(std::string("pag_comm_create failed (") + std::to_string(err) + "): " + pag_last_error());
        rank_ = static_cast<unsigned>(r);
        world_ = static_cast<unsigned>(n);
    }
    ~HipBackend() override {
        if (warm_.joinable()) warm_.join();
        pag_destroy(g_);
        pag_comm_destroy(comm_);
    }
    unsigned shardRank() const override { return rank_; }
    unsigned shardWorld() const override { return world_; }
    bool walksContig(std::size_t id) const override { return !comm_ || (id < plan_.ownerOf.size() && plan_.ownerOf[id] == static_cast<int>(rank_)); }
    const char *name() const override { return "HIP gfx950"; }
    void create(const std::uint64_t *words, std::size_t nWords, unsigned k) override {
        if (warm_.joinable()) warm_.join();
        int err = 0;
        g_ = pag_create(words, nWords, k, device_, &err);
        if (!g_) throw std::runtime_error(std::string("pag_create failed (") + std::to_string(err) + "): " + pag_last_error());
    }
    std::uint64_t solidCount() override { return pag_solid_count(g_); }
    void reset() override { check(pag_reset(g_), "pag_reset"); }
    void reserveForContigs(std::uint64_t bases) override {
        // (a rank of a sharded build walks the contigs it is dealt: about 1 / N of the block's, ShardPlan balances them by length)
        if (comm_ && world_ > 1) bases = bases / world_ + bases / (4 * world_);
        check(pag_reserve_walk_arena(g_, bases), "pag_reserve_walk_arena");
    }
    void prepare(const RawInput &raw, pag_build_input &out) override {
        check(pag_prepare(g_, &raw.view(), &out), "pag_prepare");
        if (!comm_) return;
        // who traverses which contigs of this block, and what of the graph that takes (shard_plan.hpp)
        std::vector<std::int32_t> orient(raw.ctgs.size(), PAG_ORIENT_NONE);
        for (auto &c : raw.cfg.contigs) {
            if (!raw.ctgs.contains(c.first)) continue;
            std::int32_t &o = orient[raw.ctgs.id(c.first)];
            const std::int32_t mine = c.second ? PAG_ORIENT_FORWARD : PAG_ORIENT_REVERSE;
            o = (o == PAG_ORIENT_NONE || o == mine) ? mine : PAG_ORIENT_BOTH;
        }
        const std::uint64_t halo = static_cast<std::uint64_t>(envInt("PAG_SHARD_HALO", 200000));
        plan_ = planShards(raw, orient, world_, halo, 0.90);
    }
    void process(const pag_build_input &in, pag_build_stats &stats) override {
        if (!comm_) check(pag_process(g_, &in, &stats), "pag_process");
        else check(pag_shard_run(g_, comm_, &in, plan_.regions.data(), &stats), "pag_shard_run");
    }
    void exportCsr(HostGraph &out) override {
        std::uint64_t nn = 0, np = 0, ne = 0;
        check(pag_csr_sizes(g_, &nn, &np, &ne), "pag_csr_sizes");
        out.resize(nn, np, ne);
        pag_csr csr = out.view();
        check(pag_export_csr(g_, &csr), "pag_export_csr");
    }

    void travel(const TravelContext &ctx, const pag_travel_params &params, HostGraph &graph,
                std::vector<TravelSequence> &travelled) override {
        TravelViews tv;
        travelWalks(ctx, params, tv);  // (pag_travel prepares the graph's traversal view itself when nobody has)
        if (comm_ && rank_ != 0) {
            graph = HostGraph{};
            travelled.clear();
            return;
        }
        buildPathGraph(tv.views, ctx.k, graph, travelled);
        gathered_.swap(tv.gathered);  // (the path graph refers to it)
    }

    bool travelsInHalves() const override { return true; }

    void travelPrepare(const TravelContext &ctx, const pag_travel_params &params) override {
        std::vector<std::uint32_t> refLen;
        for (std::size_t i = 0; i < ctx.refs.size(); ++i) refLen.push_back(ctx.refs.length(i));
        const SeqDb &contigs = ctx.contigs;
        pag_seqs cs{contigs.size(), contigs.byteOff().data(), contigs.lens().data(), contigs.packed().data(), contigs.packed().size()};
        const std::vector<std::int32_t> orient = orientations(ctx);
        check(pag_travel_prepare_for(g_, &cs, orient.data(), refLen.data(), refLen.size(), &params, nullptr), "pag_travel_prepare_for");
    }

    // orientation(s) per contig as PAssembly::testTravel5 walks its ctgSet (PAssembly.cpp:28-36): a contig listed with
    // both orientations is traversed twice, as two independent entries; in a sharded run only the contigs this rank was dealt
    std::vector<std::int32_t> orientations(const TravelContext &ctx) const {
        const SeqDb &contigs = ctx.contigs;
        std::vector<std::int32_t> orient(contigs.size(), PAG_ORIENT_NONE);
        for (auto &c : ctx.ctgSet) {
            if (!contigs.contains(c.first)) continue;
            const std::size_t id = contigs.id(c.first);
            if (comm_ && (id >= plan_.ownerOf.size() || plan_.ownerOf[id] != static_cast<int>(rank_))) continue;  // (another rank's)
            std::int32_t &o = orient[id];
            const std::int32_t mine = c.second ? PAG_ORIENT_FORWARD : PAG_ORIENT_REVERSE;
            o = (o == PAG_ORIENT_NONE || o == mine) ? mine : PAG_ORIENT_BOTH;
        }
        return orient;
    }

    void travelWalks(const TravelContext &ctx, const pag_travel_params &params, TravelViews &out) override {
        const SeqDb &contigs = ctx.contigs;
        const std::vector<std::int32_t> orient = orientations(ctx);
        std::vector<std::uint32_t> refLen;
        for (std::size_t i = 0; i < ctx.refs.size(); ++i) refLen.push_back(ctx.refs.length(i));
        pag_seqs cs{contigs.size(), contigs.byteOff().data(), contigs.lens().data(), contigs.packed().data(), contigs.packed().size()};
        check(pag_travel(g_, &cs, orient.data(), refLen.data(), refLen.size(), &params, nullptr), "pag_travel");
        std::vector<std::pair<const pag_path_node *, std::uint64_t>> &views = out.views;
        std::vector<char> &gathered = out.gathered;
        views.assign(2 * contigs.size(), {nullptr, 0});
        for (std::uint64_t c = 0; c < contigs.size(); ++c)
            for (int rev = 0; rev < 2; ++rev) {
                std::uint64_t len = 0;
                const pag_path_node *p = pag_travel_path_oriented(g_, c, rev == 0, &len);
                if (p && len) views[2 * c + rev] = {p, len};
            }
        if (comm_) {
            // the travel sequences of all ranks to rank 0 (which selects the chains and writes the block's outputs):
            // per rank [n][(slot, length) x n][records ...]
            std::vector<char> blob;
            std::uint64_t n = 0;
            for (auto &v : views) n += v.second ? 1 : 0;
            blob.resize(8 + n * 16);
            std::memcpy(blob.data(), &n, 8);
            std::size_t at = 8;
            for (std::uint64_t sl = 0; sl < views.size(); ++sl)
                if (views[sl].second) {
                    std::memcpy(blob.data() + at, &sl, 8);
                    std::memcpy(blob.data() + at + 8, &views[sl].second, 8);
                    at += 16;
                }
            for (auto &v : views)
                if (v.second) {
                    const std::size_t bytes = v.second * sizeof(pag_path_node);
                    blob.resize(blob.size() + bytes);
                    std::memcpy(blob.data() + blob.size() - bytes, v.first, bytes);
                }
            std::vector<std::uint64_t> sizes(world_), got(world_);
            const std::uint64_t mineBytes = blob.size();
            check(pag_comm_all_gather(comm_, &mineBytes, 8, sizes.data()), "pag_comm_all_gather");
            std::uint64_t totalBytes = 0;
            for (auto x : sizes) totalBytes += x;
            if (rank_ == 0) gathered.resize(totalBytes);
            check(pag_comm_gather_v(comm_, blob.data(), blob.size(), 0, rank_ == 0 ? gathered.data() : nullptr, rank_ == 0 ? gathered.size() : 0, got.data(),
                                    nullptr),
                  "pag_comm_gather_v");
            if (rank_ != 0) return;
            std::fill(views.begin(), views.end(), std::pair<const pag_path_node *, std::uint64_t>{nullptr, 0});
            std::size_t base = 0;
            for (unsigned r = 0; r < world_; ++r) {
                const char *b = gathered.data() + base;
                std::uint64_t cnt = 0;
                std::memcpy(&cnt, b, 8);
                const char *rec = b + 8 + cnt * 16;
                for (std::uint64_t e = 0; e < cnt; ++e) {
                    std::uint64_t sl = 0, len = 0;
                    std::memcpy(&sl, b + 8 + e * 16, 8);
                    std::memcpy(&len, b + 8 + e * 16 + 8, 8);
                    if (sl < views.size()) views[sl] = {reinterpret_cast<const pag_path_node *>(rec), len};
                    rec += len * sizeof(pag_path_node);
                }
                base += sizes[r];
            }
        }
    }

private:
    void check(int rc, const char *what) {
        if (rc == PAG_OK) return;
        const std::string msg = std::string(what) + " failed (" + std::to_string(rc) + "): " + pag_last_error();
        if (comm_) pag_comm_abort(comm_, msg.c_str());  // the other ranks of a sharded run stop with this message
        throw std::runtime_error(msg);
    }
    int device_;
    std::thread warm_;  // pag_device_warm() beside the first file parses
    pag_graph *g_ = nullptr;
    pag_comm *comm_ = nullptr;
    unsigned rank_ = 0, world_ = 1;
    ShardPlan plan_;
    std::vector<char> gathered_;  // rank 0: the travel sequences of all ranks (the path graph refers to them)
};

}  // namespace

std::unique_ptr<GraphBackend> makeHipBackend(int deviceOrdinal) { return std::unique_ptr<GraphBackend>(new HipBackend(deviceOrdinal)); }

}  // namespace pagh
