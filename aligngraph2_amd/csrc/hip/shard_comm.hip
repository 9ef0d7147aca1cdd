// shard_comm.hip — ONE config block built by the GPUs of a node, natively (SURVEY.md 8e level 2): the communicator and
// pag_shard_run, the whole sharded build behind one call.
//
// One process per GPU.  The two bulk exchanges — 12-byte tuples to their k-mer owners, and every rank's region of the
// finished graph from the owners (pag_shard_select) — are all-to-all(v)s of DEVICE buffers: RCCL over xGMI (grouped
// ncclSend / ncclRecv, the library loaded at run time), or, where RCCL cannot be used (two ranks on one device: the
// single-GPU test box), a transport through files of the rendezvous directory.  Small host-side tables (counts, sizes,
// statistics, the RCCL unique id, gathered travel sequences) always go through the rendezvous directory, which every rank of
// the node sees (/dev/shm by default): no second launcher, no sockets.
//
// Order argument for bit-identity with one GPU: see include/pagraph_hip.h (pag_shard_*) — rank r extracts the r-th contiguous
// range of the emission order, owners lay the received records out as [pass 1 from rank 0] .. [pass 1 from rank N-1]
// [pass 2 from rank 0] .. and sort stably by k-mer.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pag_graph_impl.hpp"

namespace {

// the few RCCL entry points used (rccl.h): resolved with dlsym so that the library only depends on RCCL when it is asked for
struct Rccl {
    void *lib = nullptr;
    struct UniqueId {
        char internal[128];
    };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
        CommInitRank = (int (*)(void **, int, UniqueId, int))sym("ncclCommInitRank");
        CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
        GroupStart = (int (*)())sym("ncclGroupStart");
        GroupEnd = (int (*)())sym("ncclGroupEnd");
        Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))sym("ncclSend");
        Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))sym("ncclRecv");
        GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
        return GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
constexpr int NCCL_UINT8 = 1;  // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

struct pag_comm {
    int rank = 0, world = 1, device = 0;
    std::string dir;
    bool use_rccl = false;
    Rccl rccl;
    void *comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t seq = 0;       // every collective of the job takes the next number (all ranks call them in the same order)
    double timeout_s = 600;
    uint64_t bytes_sent = 0;  // payload of the bulk exchanges that left this rank (wire volume)

    std::string path(const char *tag, uint64_t n, int a, int b = -1) const {
        char buf[96];
        if (b >= 0) std::snprintf(buf, sizeof buf, "/%s%llu_%d_%d", tag, (unsigned long long)n, a, b);
        else std::snprintf(buf, sizeof buf, "/%s%llu_%d", tag, (unsigned long long)n, a);
        return dir + buf;
    }
    // a file that appears complete or not at all (written under another name, then renamed)
    int put_file(const std::string &p, const void *data, size_t bytes) const {
        const std::string tmp = p + ".part";
        FILE *f = std::fopen(tmp.c_str(), "wb");
        if (!f) {
            pagdev::set_error("pag_comm: cannot write %s", tmp.c_str());
            return PAG_EFAULT;
        }
        const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
        if (std::fclose(f) != 0 || !ok || std::rename(tmp.c_str(), p.c_str()) != 0) {
            pagdev::set_error("pag_comm: cannot write %s", p.c_str());
            return PAG_EFAULT;
        }
        return PAG_OK;
    }
    int get_file(const std::string &p, std::vector<char> &out, bool remove_after) const {
        const double t0 = now_s();
        struct stat st;
        while (stat(p.c_str(), &st) != 0) {
            if (now_s() - t0 > timeout_s) {
                pagdev::set_error("pag_comm: rank %d waited %.0f s for %s (a peer has failed or never started)", rank, timeout_s, p.c_str());
                return PAG_EFAULT;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        out.resize((size_t)st.st_size);
        FILE *f = std::fopen(p.c_str(), "rb");
        if (!f || (out.size() && std::fread(out.data(), 1, out.size(), f) != out.size())) {
            if (f) std::fclose(f);
            pagdev::set_error("pag_comm: cannot read %s", p.c_str());
            return PAG_EFAULT;
        }
        std::fclose(f);
        if (remove_after) std::remove(p.c_str());
        return PAG_OK;
    }
};

extern "C" {

pag_comm *pag_comm_create(int rank, int world, const char *rendezvous_dir, int device, const char *transport, int *err) {
    auto fail = [&](int code) -> pag_comm * {
        if (err) *err = code;
        return nullptr;
    };
    if (rank < 0 || world < 1 || rank >= world || !rendezvous_dir) return fail(PAG_EINVAL);
    pag_comm *c = new pag_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    c->dir = rendezvous_dir;
    if (const char *e = std::getenv("PAG_COMM_TIMEOUT_S")) c->timeout_s = std::max(1.0, std::atof(e));
    mkdir(c->dir.c_str(), 0700);  // (every rank tries; the directory may exist)
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        pagdev::set_error("pag_comm_create: device %d unusable", device);
        delete c;
        return fail(PAG_ENODEV);
    }
    const bool want_rccl = !transport || std::strcmp(transport, "rccl") == 0;
    // (PAG_COMM_FORCE_RCCL=1: also for a world of one — exercises the RCCL calls on a single-GPU box)
    if (want_rccl && (world > 1 || (std::getenv("PAG_COMM_FORCE_RCCL") && transport && std::strcmp(transport, "rccl") == 0))) {
        if (!c->rccl.load()) {
            pagdev::set_error("pag_comm_create: librccl.so not found (transport \"host\" goes through the rendezvous directory)");
            delete c;
            return fail(PAG_ENODEV);
        }
        Rccl::UniqueId id{};
        std::vector<char> blob;
        int rc = PAG_OK;
        if (rank == 0) {
            if (c->rccl.GetUniqueId(&id) != 0) rc = PAG_EFAULT;
            else rc = c->put_file(c->dir + "/rccl_id", &id, sizeof id);
        } else {
            rc = c->get_file(c->dir + "/rccl_id", blob, false);
            if (rc == PAG_OK && blob.size() == sizeof id) std::memcpy(&id, blob.data(), sizeof id);
            else if (rc == PAG_OK) rc = PAG_EFAULT;
        }
        int nrc = rc == PAG_OK ? c->rccl.CommInitRank(&c->comm, world, id, rank) : -1;
        if (rc != PAG_OK || nrc != 0) {
            pagdev::set_error("pag_comm_create: RCCL communicator of %d ranks failed (%s)", world,
                              nrc > 0 && c->rccl.GetErrorString ? c->rccl.GetErrorString(nrc) : "rendezvous");
            delete c;
            return fail(PAG_EFAULT);
        }
        c->use_rccl = true;
    }
    if (err) *err = PAG_OK;
    return c;
}

void pag_comm_destroy(pag_comm *c) {
    if (!c) return;
    if (c->comm) c->rccl.CommDestroy(c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
int pag_comm_rank(const pag_comm *c) { return c ? c->rank : -1; }
int pag_comm_world(const pag_comm *c) { return c ? c->world : 0; }
uint64_t pag_comm_bytes_sent(const pag_comm *c) { return c ? c->bytes_sent : 0; }

// every rank contributes `bytes` of host memory; all[r * bytes ..] = rank r's
int pag_comm_all_gather(pag_comm *c, const void *mine, uint64_t bytes, void *all) {
    if (!c || (!mine && bytes) || !all) return PAG_EINVAL;
    const uint64_t n = c->seq++;
    if (c->world == 1) {
        std::memcpy(all, mine, bytes);
        return PAG_OK;
    }
    int rc = c->put_file(c->path("g", n, c->rank), mine, bytes);
    if (rc) return rc;
    std::vector<char> blob;
    for (int r = 0; r < c->world; ++r) {
        if ((rc = c->get_file(c->path("g", n, r), blob, false))) return rc;
        if (blob.size() != bytes) {
            pagdev::set_error("pag_comm_all_gather: rank %d sent %zu bytes, %llu expected", r, blob.size(), (unsigned long long)bytes);
            return PAG_EFAULT;
        }
        std::memcpy((char *)all + (size_t)r * bytes, blob.data(), bytes);
    }
    return PAG_OK;
}
int pag_comm_barrier(pag_comm *c) {
    char x = 0;
    std::vector<char> all(c ? (size_t)c->world : 1);
    return pag_comm_all_gather(c, &x, 1, all.data());
}
// host blobs of any size to `root`: sizes[r] / offsets into `out` (capacity out_cap, required size returned in *need)
int pag_comm_gather_v(pag_comm *c, const void *mine, uint64_t bytes, int root, void *out, uint64_t out_cap, uint64_t *sizes, uint64_t *need) {
    if (!c || (!mine && bytes)) return PAG_EINVAL;
    const uint64_t n = c->seq++;
    int rc;
    if (c->rank != root) return c->put_file(c->path("v", n, c->rank), mine, bytes);
    uint64_t at = 0;
    std::vector<char> blob;
    for (int r = 0; r < c->world; ++r) {
        const void *src = mine;
        uint64_t sz = bytes;
        if (r != root) {
            if ((rc = c->get_file(c->path("v", n, r), blob, true))) return rc;
            src = blob.data();
            sz = blob.size();
        }
        if (sizes) sizes[r] = sz;
        if (out && at + sz <= out_cap && sz) std::memcpy((char *)out + at, src, sz);
        at += sz;
    }
    if (need) *need = at;
    return at <= out_cap ? PAG_OK : PAG_ERANGE;
}

// all-to-all(v) of DEVICE memory: send_bytes[d] consecutive bytes of `send` go to rank d, recv_bytes[s] bytes arrive from
// rank s, in rank order on both sides
int pag_comm_all_to_all_v(pag_comm *c, const void *send, const uint64_t *send_bytes, void *recv, const uint64_t *recv_bytes) {
    if (!c || !send_bytes || !recv_bytes) return PAG_EINVAL;
    const uint64_t n = c->seq++;
    PAG_HIP_TRY(hipSetDevice(c->device));
    std::vector<uint64_t> so(c->world + 1, 0), ro(c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) {
        so[r + 1] = so[r] + send_bytes[r];
        ro[r + 1] = ro[r] + recv_bytes[r];
        if (r != c->rank) c->bytes_sent += send_bytes[r];
    }
    if (send_bytes[c->rank] != recv_bytes[c->rank]) return PAG_EINVAL;
    if (c->use_rccl) {
        int rc = c->rccl.GroupStart();
        for (int r = 0; r < c->world && rc == 0; ++r) {
            if (send_bytes[r]) rc = c->rccl.Send((const char *)send + so[r], send_bytes[r], NCCL_UINT8, r, c->comm, c->stream);
            if (rc == 0 && recv_bytes[r]) rc = c->rccl.Recv((char *)recv + ro[r], recv_bytes[r], NCCL_UINT8, r, c->comm, c->stream);
        }
        const int rc2 = c->rccl.GroupEnd();
        if (rc != 0 || rc2 != 0) {
            pagdev::set_error("pag_comm_all_to_all_v: RCCL error %s", c->rccl.GetErrorString ? c->rccl.GetErrorString(rc ? rc : rc2) : "?");
            return PAG_EFAULT;
        }
        PAG_HIP_TRY(hipStreamSynchronize(c->stream));
        return PAG_OK;
    }
    // through the rendezvous directory (single-device test boxes)
    std::vector<char> host;
    int rc;
    for (int d = 0; d < c->world; ++d) {
        if (d == c->rank) continue;
        host.resize(send_bytes[d]);
        if (send_bytes[d]) PAG_HIP_TRY(hipMemcpy(host.data(), (const char *)send + so[d], send_bytes[d], hipMemcpyDeviceToHost));
        if ((rc = c->put_file(c->path("a", n, c->rank, d), host.data(), send_bytes[d]))) return rc;
    }
    if (send_bytes[c->rank]) PAG_HIP_TRY(hipMemcpy((char *)recv + ro[c->rank], (const char *)send + so[c->rank], send_bytes[c->rank], hipMemcpyDeviceToDevice));
    for (int s = 0; s < c->world; ++s) {
        if (s == c->rank) continue;
        if ((rc = c->get_file(c->path("a", n, s, c->rank), host, true))) return rc;
        if (host.size() != recv_bytes[s]) {
            pagdev::set_error("pag_comm_all_to_all_v: %zu bytes from rank %d, %llu expected", host.size(), s, (unsigned long long)recv_bytes[s]);
            return PAG_EFAULT;
        }
        if (recv_bytes[s]) PAG_HIP_TRY(hipMemcpy((char *)recv + ro[s], host.data(), recv_bytes[s], hipMemcpyHostToDevice));
    }
    return PAG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// pag_shard_run: the sharded build of one block behind one call (every rank calls it with the same prepared input).
//   extract own read range -> all-to-all(v) of the tuple / edge streams -> K2-K4 on the owned k-mer range -> every rank's
//   region selected from the slice -> all-to-all(v) of the selections, received straight into the handle's graph buffers ->
//   region set, the build's memory released.
// regions[world]: what every rank traverses (the same on all ranks); total: the block's count lines.
// ---------------------------------------------------------------------------------------------------------------------
int pag_shard_run(pag_graph *g, pag_comm *c, const pag_build_input *in, const pag_region *regions, pag_build_stats *total) {
    using namespace pagdev;
    if (!g || !c || !in || !regions) return PAG_EINVAL;
    const int W = c->world, me = c->rank;
    int rc;
    std::vector<uint64_t> counts(4 * (size_t)W), allc(4 * (size_t)W * W);
    if ((rc = pag_shard_extract(g, in, (uint32_t)me, (uint32_t)W, counts.data()))) return rc;
    if ((rc = pag_comm_all_gather(c, counts.data(), counts.size() * 8, allc.data()))) return rc;
    auto cnt = [&](int src, int dst, int q) { return allc[((size_t)src * W + dst) * 4 + q]; };
    hipStream_t s = g->stream;
    // ---- the partitioned streams to their owners.  Received: [from rank 0: pass 1, pass 2][from rank 1: ..] per array;
    //      laid out for the build as [pass 1 from all ranks][pass 2 from all ranks]
    const int ts = g->shard_in0[0] ? 30 : 32, es = g->shard_in0[1] ? 34 : 36;
    uint64_t nT = 0, nE = 0, t1 = 0, e1 = 0;
    for (int r = 0; r < W; ++r) {
        nT += cnt(r, me, 0) + cnt(r, me, 1);
        nE += cnt(r, me, 2) + cnt(r, me, 3);
        t1 += cnt(r, me, 0);
        e1 += cnt(r, me, 2);
    }
    DevBuf b_rk(g, 206), b_rv(g, 207), b_lk(g, 208), b_lv(g, 209);
    const uint64_t nmax = std::max(nT, nE);
    if ((rc = b_rk.alloc((nmax + 1) * 4)) || (rc = b_rv.alloc((nmax + 1) * 8)) || (rc = b_lk.alloc((nT + 1) * 4)) || (rc = b_lv.alloc((nT + 1) * 8))) return rc;
    DevBuf b_lek(g, 210), b_lev(g, 211);
    if ((rc = b_lek.alloc((nE + 1) * 4)) || (rc = b_lev.alloc((nE + 1) * 8))) return rc;
    for (int stream_no = 0; stream_no < 2; ++stream_no) {
        const int q0 = stream_no * 2;
        const void *sk = g->pool[stream_no == 0 ? ts : es].p, *sv = g->pool[(stream_no == 0 ? ts : es) + 1].p;
        void *lk = stream_no == 0 ? b_lk.p : b_lek.p, *lv = stream_no == 0 ? b_lv.p : b_lev.p;
        std::vector<uint64_t> sb(W), rb(W);
        for (int esz : {4, 8}) {
            for (int r = 0; r < W; ++r) {
                sb[r] = (cnt(me, r, q0) + cnt(me, r, q0 + 1)) * esz;
                rb[r] = (cnt(r, me, q0) + cnt(r, me, q0 + 1)) * esz;
            }
            void *recv = esz == 4 ? b_rk.p : b_rv.p;
            if ((rc = pag_comm_all_to_all_v(c, esz == 4 ? sk : sv, sb.data(), recv, rb.data()))) return rc;
            // [pass 1 from rank 0] .. [pass 1 from rank W-1] [pass 2 from rank 0] ..
            uint64_t src = 0, d1 = 0, d2 = 0;
            for (int r = 0; r < W; ++r) d2 += cnt(r, me, q0);
            for (int r = 0; r < W; ++r) {
                const uint64_t a = cnt(r, me, q0), b = cnt(r, me, q0 + 1);
                char *dst = (char *)(esz == 4 ? lk : lv);
                if (a) PAG_HIP_TRY(hipMemcpyAsync(dst + d1 * esz, (char *)recv + src * esz, a * esz, hipMemcpyDeviceToDevice, s));
                if (b) PAG_HIP_TRY(hipMemcpyAsync(dst + d2 * esz, (char *)recv + (src + a) * esz, b * esz, hipMemcpyDeviceToDevice, s));
                src += a + b;
                d1 += a;
                d2 += b;
            }
            PAG_HIP_TRY(hipStreamSynchronize(s));
        }
    }
    pag_build_stats mine{};
    if ((rc = pag_shard_build(g, b_lk.as<uint32_t>(), b_lv.as<uint64_t>(), nT, t1, b_lek.as<uint32_t>(), b_lev.as<uint64_t>(), nE, e1, in->eps, &mine))) return rc;
    // ---- every rank's region of this owner's slice
    std::vector<pag_shard_slice> sel(W);
    std::vector<uint64_t> my_sizes(2 * (size_t)W), all_sizes(2 * (size_t)W * W);
    // (a selection lives in the handle until the next one: copied behind the previous ones into the send buffers)
    struct Arr {
        int esz;
        int slot;
    };
    const Arr arrs[7] = {{4, 212}, {8, 213}, {4, 214}, {2, 215}, {4, 216}, {8, 217}, {4, 218}};  // tkey tval tseg tcnt | ekey eval eseg
    std::vector<uint64_t> at(7, 0);
    std::vector<pag_build_stats> stats_to(W);
    for (int d = 0; d < W; ++d) {
        if ((rc = pag_shard_select(g, &regions[d], &sel[d]))) return rc;
        my_sizes[2 * d] = sel[d].n_t;
        my_sizes[2 * d + 1] = sel[d].n_e;
        stats_to[d] = sel[d].stats;
        const void *src[7] = {sel[d].tkey, sel[d].tval, sel[d].tseg, sel[d].tcnt, sel[d].ekey, sel[d].eval, sel[d].eseg};
        for (int a = 0; a < 7; ++a) {
            const uint64_t n = a < 4 ? sel[d].n_t : sel[d].n_e;
            DevBuf b(g, arrs[a].slot);
            // (grown with the old contents kept: DevBuf::alloc replaces the allocation)
            const uint64_t need = (at[a] + n + 1) * arrs[a].esz;
            if (b.sl->cap < need) {
                void *np = nullptr;
                const size_t want = need + need / 2 + 256;
                PAG_HIP_TRY(hipMalloc(&np, want));
                if (b.sl->p && at[a]) PAG_HIP_TRY(hipMemcpy(np, b.sl->p, at[a] * arrs[a].esz, hipMemcpyDeviceToDevice));
                if (b.sl->p) hipFree(b.sl->p);
                b.sl->p = np;
                b.sl->cap = want;
            }
            if (n) PAG_HIP_TRY(hipMemcpyAsync((char *)b.sl->p + at[a] * arrs[a].esz, src[a], n * arrs[a].esz, hipMemcpyDeviceToDevice, s));
            at[a] += n;
        }
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }
    if ((rc = pag_comm_all_gather(c, my_sizes.data(), my_sizes.size() * 8, all_sizes.data()))) return rc;
    std::vector<pag_build_stats> all_stats((size_t)W * W);
    if ((rc = pag_comm_all_gather(c, stats_to.data(), stats_to.size() * sizeof(pag_build_stats), all_stats.data()))) return rc;
    auto size_of = [&](int owner, int dst, int which) { return all_sizes[((size_t)owner * W + dst) * 2 + which]; };
    uint64_t T = 0, E = 0;
    for (int o = 0; o < W; ++o) {
        T += size_of(o, me, 0);
        E += size_of(o, me, 1);
    }
    // received straight into the buffers pag_shard_import fills (slots 52 .. 58): owner order = ascending k-mer ranges
    DevBuf imp[7] = {DevBuf(g, 52), DevBuf(g, 53), DevBuf(g, 54), DevBuf(g, 55), DevBuf(g, 56), DevBuf(g, 57), DevBuf(g, 58)};
    for (int a = 0; a < 7; ++a) {
        const uint64_t n = a < 4 ? T : E;
        if ((rc = imp[a].alloc((n + 1) * arrs[a].esz))) return rc;
        std::vector<uint64_t> sb(W), rb(W);
        for (int r = 0; r < W; ++r) {
            sb[r] = size_of(me, r, a < 4 ? 0 : 1) * arrs[a].esz;
            rb[r] = size_of(r, me, a < 4 ? 0 : 1) * arrs[a].esz;
        }
        if ((rc = pag_comm_all_to_all_v(c, g->pool[arrs[a].slot].p, sb.data(), imp[a].p, rb.data()))) return rc;
    }
    pag_build_stats st{};
    for (int o = 0; o < W; ++o) {
        const pag_build_stats &P = all_stats[(size_t)o * W + me];
        for (int q = 0; q < 2; ++q) {
            st.merge_edge[q] += P.merge_edge[q];
            st.total_pos[q] += P.total_pos[q];
            st.merge_pos[q] += P.merge_pos[q];
            st.n_tuples[q] += P.n_tuples[q];
            st.n_edges[q] += P.n_edges[q];
        }
        st.n_nodes += P.n_nodes;
        st.n_pos += P.n_pos;
        st.n_uniq_edges += P.n_uniq_edges;
    }
    if ((rc = pag_shard_adopt(g, T, E, &st))) return rc;
    if ((rc = pag_shard_set_region(g, &regions[me]))) return rc;
    if ((rc = pag_shard_release_build(g))) return rc;
    if (total) *total = st;
    return PAG_OK;
}

}  // extern "C"
