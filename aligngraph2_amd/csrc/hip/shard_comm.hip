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
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>  // types and prototypes only: the entry points are resolved with dlsym (no link-time dependency)
#include <signal.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "pag_graph_impl.hpp"

namespace {

// the few RCCL entry points used, with the signatures rccl.h declares them with (decltype of the prototypes), resolved
// with dlsym so that the library only depends on RCCL when it is asked for
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load() {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        auto sym = [&](const char *n) { return dlsym(lib, n); };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommAbort = (decltype(CommAbort))sym("ncclCommAbort");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        return GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat), 0 if it does not exist: (pid, start
// time) names ONE process for as long as the machine is up
unsigned long long proc_start(long pid) {
    char path[64];
    std::snprintf(path, sizeof path, "/proc/%ld/stat", pid);
    FILE *f = std::fopen(path, "r");
    if (!f) return 0;
    char buf[1024];
    const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[n] = 0;
    const char *p = std::strrchr(buf, ')');  // (the command name may hold blanks and parentheses)
    if (!p) return 0;
    unsigned long long v = 0;
    int field = 2;
    for (++p; *p && field < 22;) {
        while (*p == ' ') ++p;
        ++field;
        if (field == 22) {
            v = std::strtoull(p, nullptr, 10);
            break;
        }
        while (*p && *p != ' ') ++p;
    }
    return v;
}

}  // namespace

struct pag_comm {
    int rank = 0, world = 1, device = 0;
    std::string dir;
    bool use_rccl = false;
    Rccl rccl;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t seq = 0;       // every collective of the job takes the next number (all ranks call them in the same order)
    double timeout_s = 600;
    uint64_t bytes_sent = 0;  // payload of the bulk exchanges that left this rank (wire volume)
    // Every file of the job carries the job's nonce in its name (agreed at creation, see handshake()): what an earlier
    // job — crashed, or still running in the same directory — left or leaves there is never read.
    unsigned long long job = 0;
    bool aborted = false;
    std::vector<std::pair<uint64_t, std::string>> kept;  // all-gather files of mine that peers may still read: (collective, path)

    std::string path(const char *tag, uint64_t n, int a, int b = -1) const {
        char buf[128];
        if (b >= 0) std::snprintf(buf, sizeof buf, "/j%016llx_%s%llu_%d_%d", job, tag, (unsigned long long)n, a, b);
        else std::snprintf(buf, sizeof buf, "/j%016llx_%s%llu_%d", job, tag, (unsigned long long)n, a);
        return dir + buf;
    }
    std::string abort_path(int r) const {
        char buf[64];
        std::snprintf(buf, sizeof buf, "/j%016llx_abort_%d", job, r);
        return dir + buf;
    }
    // a file that appears complete or not at all (written under another name, then renamed)
    int put_file(const std::string &p, const void *data, size_t bytes) const {
        char suffix[48];
        std::snprintf(suffix, sizeof suffix, ".part%d_%ld", rank, (long)getpid());
        const std::string tmp = p + suffix;
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
    static bool slurp(const std::string &p, std::vector<char> &out) {
        FILE *f = std::fopen(p.c_str(), "rb");
        if (!f) return false;
        out.clear();
        char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
        std::fclose(f);
        return true;
    }
    // a peer that failed says so (pag_comm_abort): the waiting ranks report ITS message at once instead of timing out
    bool peer_aborted() const {
        if (!job) return false;
        std::vector<char> msg;
        for (int r = 0; r < world; ++r) {
            if (r == rank) continue;
            struct stat st;
            const std::string ap = abort_path(r);
            if (stat(ap.c_str(), &st) != 0) continue;
            slurp(ap, msg);
            msg.push_back(0);
            pagdev::set_error("pag_comm: rank %d has failed: %s", r, msg.data());
            return true;
        }
        return false;
    }
    int get_file(const std::string &p, std::vector<char> &out, bool remove_after) const {
        const double t0 = now_s();
        struct stat st;
        for (unsigned spin = 0; stat(p.c_str(), &st) != 0; ++spin) {
            if ((spin & 63u) == 63u && peer_aborted()) return PAG_EFAULT;
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
    // all-gather files of collectives before `done` have been read by every peer (each of them has since written a file
    // of a later collective, which it only does after leaving the earlier one)
    void drop_kept(uint64_t done) {
        size_t w = 0;
        for (size_t i = 0; i < kept.size(); ++i) {
            if (kept[i].first < done) std::remove(kept[i].second.c_str());
            else kept[w++] = kept[i];
        }
        kept.resize(w);
    }

    // The ranks of ONE job find each other in a directory that may hold the files of other jobs.  Every rank publishes
    // hello_<rank> = (pid, process start time, a random word); rank 0 waits until every rank's hello names a LIVE process
    // (what a crashed job left names a dead one and is ignored until its successor overwrites it), draws the job's nonce and
    // answers every rank with job_<rank> = (nonce, the random word it saw).  A rank takes the answer that echoes ITS word —
    // an answer left by an earlier job echoes another.  Two live jobs in one directory: the second hello of a rank
    // overwrites the first, one of the two jobs never gets its echo and times out with a message that says so.
    int handshake() {
        if (world == 1) {
            job = std::random_device{}() | ((unsigned long long)std::random_device{}() << 32) | 1ull;
            return PAG_OK;
        }
        struct Hello {
            long long pid;
            unsigned long long start, word;
        } me{(long long)getpid(), proc_start(getpid()), std::random_device{}() | ((unsigned long long)std::random_device{}() << 32)};
        struct Job {
            unsigned long long nonce, word;
        };
        char name[64];
        std::snprintf(name, sizeof name, "/hello_%d", rank);
        int rc = put_file(dir + name, &me, sizeof me);
        if (rc) return rc;
        const double t0 = now_s();
        std::vector<char> blob;
        if (rank == 0) {
            std::vector<Hello> seen(world);
            seen[0] = me;
            for (int r = 1; r < world; ++r) {
                std::snprintf(name, sizeof name, "/hello_%d", r);
                for (;;) {
                    Hello h{};
                    if (slurp(dir + name, blob) && blob.size() == sizeof h) {
                        std::memcpy(&h, blob.data(), sizeof h);
                        // (PAG_COMM_ANY_NAMESPACE=1: the ranks run in containers with PID namespaces of their own — one per GPU
                        // sharing the rendezvous directory — where /proc/<pid> of a peer is not this process's to see: a hello
                        // is then taken when the directory is the job's own fresh one, which is the launcher's to guarantee)
                        static const bool any_ns = pagdev::env_int("PAG_COMM_ANY_NAMESPACE", 0) != 0;
                        if (h.start != 0 && (any_ns || proc_start((long)h.pid) == h.start)) {
                            seen[r] = h;
                            break;
                        }
                    }
                    if (now_s() - t0 > timeout_s) {
                        pagdev::set_error("pag_comm_create: rank 0 waited %.0f s for a live rank %d in %s", timeout_s, r, dir.c_str());
                        return PAG_EFAULT;
                    }
                    std::this_thread::sleep_for(std::chrono::microseconds(500));
                }
            }
            job = std::random_device{}() | ((unsigned long long)std::random_device{}() << 32) | 1ull;
            for (int r = 1; r < world; ++r) {
                Job j{job, seen[r].word};
                std::snprintf(name, sizeof name, "/job_%d", r);
                if ((rc = put_file(dir + name, &j, sizeof j))) return rc;
            }
            return PAG_OK;
        }
        std::snprintf(name, sizeof name, "/job_%d", rank);
        for (;;) {
            Job j{};
            if (slurp(dir + name, blob) && blob.size() == sizeof j) {
                std::memcpy(&j, blob.data(), sizeof j);
                if (j.word == me.word) {
                    job = j.nonce;
                    std::remove((dir + name).c_str());
                    return PAG_OK;
                }
            }
            if (now_s() - t0 > timeout_s) {
                pagdev::set_error("pag_comm_create: rank %d waited %.0f s for rank 0's answer in %s (no rank 0, or another job is using the directory)",
                                  rank, timeout_s, dir.c_str());
                return PAG_EFAULT;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(500));
        }
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
    if (c->handshake() != PAG_OK) {
        hipStreamDestroy(c->stream);
        delete c;
        return fail(PAG_EFAULT);
    }
    const bool want_rccl = !transport || std::strcmp(transport, "rccl") == 0;
    // (PAG_COMM_FORCE_RCCL=1: also for a world of one — exercises the RCCL calls on a single-GPU box)
    if (want_rccl && (world > 1 || (std::getenv("PAG_COMM_FORCE_RCCL") && transport && std::strcmp(transport, "rccl") == 0))) {
        if (!c->rccl.load()) {
            pagdev::set_error("pag_comm_create: librccl.so not found (transport \"host\" goes through the rendezvous directory)");
            pag_comm_destroy(c);
            return fail(PAG_ENODEV);
        }
        ncclUniqueId id{};
        std::vector<char> blob;
        int rc = PAG_OK;
        const std::string idp = c->path("rcclid", 0, 0);
        if (rank == 0) {
            if (c->rccl.GetUniqueId(&id) != ncclSuccess) rc = PAG_EFAULT;
            else rc = c->put_file(idp, &id, sizeof id);
        } else {
            rc = c->get_file(idp, blob, false);
            if (rc == PAG_OK && blob.size() == sizeof id) std::memcpy(&id, blob.data(), sizeof id);
            else if (rc == PAG_OK) rc = PAG_EFAULT;
        }
        const ncclResult_t nrc = rc == PAG_OK ? c->rccl.CommInitRank(&c->comm, world, id, rank) : ncclSystemError;
        if (rank == 0 && rc == PAG_OK) c->kept.push_back({0, idp});  // (read by every peer before its CommInitRank returns)
        if (rc != PAG_OK || nrc != ncclSuccess) {
            pagdev::set_error("pag_comm_create: RCCL communicator of %d ranks failed (%s)", world,
                              rc == PAG_OK && c->rccl.GetErrorString ? c->rccl.GetErrorString(nrc) : "rendezvous");
            c->comm = nullptr;
            pag_comm_destroy(c);
            return fail(PAG_EFAULT);
        }
        c->use_rccl = true;
    }
    if (err) *err = PAG_OK;
    return c;
}

void pag_comm_destroy(pag_comm *c) {
    if (!c) return;
    if (c->comm) {
        // a communicator whose job failed may have peers that never arrive: abort instead of the collective destroy
        if (c->aborted && c->rccl.CommAbort) c->rccl.CommAbort(c->comm);
        else c->rccl.CommDestroy(c->comm);
    }
    if (c->stream) hipStreamDestroy(c->stream);
    // (the files of this rank's LAST all-gather may still be read by a peer; whatever stays carries the job's nonce and is
    // never taken for another job's)
    if (!c->kept.empty()) c->drop_kept(c->kept.back().first);
    char name[64];
    std::snprintf(name, sizeof name, "/hello_%d", c->rank);
    std::remove((c->dir + name).c_str());
    delete c;
}
int pag_comm_rank(const pag_comm *c) { return c ? c->rank : -1; }
int pag_comm_world(const pag_comm *c) { return c ? c->world : 0; }
uint64_t pag_comm_bytes_sent(const pag_comm *c) { return c ? c->bytes_sent : 0; }

// This rank cannot go on (message: why): every peer that waits for it — now or later — fails at once with that message
// instead of waiting for its timeout.
void pag_comm_abort(pag_comm *c, const char *message) {
    if (!c || c->aborted) return;
    c->aborted = true;
    if (c->world == 1 || !c->job) return;
    const char *m = message && *message ? message : "(no message)";
    const std::string keep(m);  // (put_file may overwrite the error buffer `message` points into)
    c->put_file(c->abort_path(c->rank), keep.data(), keep.size());
}

// every rank contributes `bytes` of host memory; all[r * bytes ..] = rank r's
int pag_comm_all_gather(pag_comm *c, const void *mine, uint64_t bytes, void *all) {
    if (!c || (!mine && bytes) || !all) return PAG_EINVAL;
    const uint64_t n = c->seq++;
    if (c->world == 1) {
        std::memcpy(all, mine, bytes);
        return PAG_OK;
    }
    if (c->aborted) {
        pagdev::set_error("pag_comm: this rank has aborted the job");
        return PAG_EFAULT;
    }
    const std::string minep = c->path("g", n, c->rank);
    int rc = c->put_file(minep, mine, bytes);
    if (rc) return rc;
    c->kept.push_back({n, minep});
    std::vector<char> blob;
    for (int r = 0; r < c->world; ++r) {
        if ((rc = c->get_file(c->path("g", n, r), blob, false))) return rc;
        if (blob.size() != bytes) {
            pagdev::set_error("pag_comm_all_gather: rank %d sent %zu bytes, %llu expected", r, blob.size(), (unsigned long long)bytes);
            return PAG_EFAULT;
        }
        std::memcpy((char *)all + (size_t)r * bytes, blob.data(), bytes);
    }
    c->drop_kept(n);  // every peer has written its file of collective n: it has left every earlier collective
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
        // a grouped Send / Recv whose peer never arrives waits for ever: the ranks first meet through the rendezvous directory
        // (where a failed peer's abort marker is seen), and only a complete set of ranks enters RCCL
        if (c->world > 1) {
            const int brc = pag_comm_barrier(c);
            if (brc) return brc;
        }
        ncclResult_t rc = c->rccl.GroupStart();
        for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
            if (send_bytes[r]) rc = c->rccl.Send((const char *)send + so[r], send_bytes[r], ncclUint8, r, c->comm, c->stream);
            if (rc == ncclSuccess && recv_bytes[r]) rc = c->rccl.Recv((char *)recv + ro[r], recv_bytes[r], ncclUint8, r, c->comm, c->stream);
        }
        const ncclResult_t rc2 = c->rccl.GroupEnd();
        if (rc != ncclSuccess || rc2 != ncclSuccess) {
            pagdev::set_error("pag_comm_all_to_all_v: RCCL error %s", c->rccl.GetErrorString ? c->rccl.GetErrorString(rc != ncclSuccess ? rc : rc2) : "?");
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

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// A bulk exchange that runs IN THE BACKGROUND (round 5): several device arrays, each an all-to-all(v) as above, begun by
// xchg_begin and waited for by xchg_end, so that the next piece of work of the call's own stream — the next chunk's
// extraction, the next destination's selection — runs beside it.  RCCL: the grouped sends / receives of all arrays are
// enqueued on the communicator's stream (a stream of its own, non-blocking) and xchg_end waits for that stream.  Files of the
// rendezvous directory (ranks that share a device): a helper thread does what pag_comm_all_to_all_v does, with copies on the
// communicator's stream.  One exchange at a time per communicator.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct Xchg {
    pag_comm *c = nullptr;
    bool active = false;
    std::thread th;
    int rc = PAG_OK;
    std::string err;
    double t_begin = 0, t_done = 0;  // (host clock; t_done is set when the transfer is known to be over)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float dev_ms = 0.f;
};
struct XchgArr {
    const void *send;
    void *recv;
    std::vector<uint64_t> sb, rb;  // bytes to / from every rank
};

int xchg_files(pag_comm *c, uint64_t n, const XchgArr &A, std::string &err) {
    std::vector<uint64_t> so(c->world + 1, 0), ro(c->world + 1, 0);
    for (int r = 0; r < c->world; ++r) {
        so[r + 1] = so[r] + A.sb[r];
        ro[r + 1] = ro[r] + A.rb[r];
    }
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    };
    std::vector<char> host;
    int rc;
    for (int d = 0; d < c->world; ++d) {
        if (d == c->rank) continue;
        host.resize(A.sb[d]);
        if (A.sb[d] && !(hip_ok(hipMemcpyAsync(host.data(), (const char *)A.send + so[d], A.sb[d], hipMemcpyDeviceToHost, c->stream), "copy to host") &&
                         hip_ok(hipStreamSynchronize(c->stream), "copy to host")))
            return PAG_EFAULT;
        if ((rc = c->put_file(c->path("a", n, c->rank, d), host.data(), A.sb[d]))) {
            err = pagdev::last_error();
            return rc;
        }
    }
    if (A.sb[c->rank] && !(hip_ok(hipMemcpyAsync((char *)A.recv + ro[c->rank], (const char *)A.send + so[c->rank], A.sb[c->rank], hipMemcpyDeviceToDevice, c->stream), "own part") &&
                           hip_ok(hipStreamSynchronize(c->stream), "own part")))
        return PAG_EFAULT;
    for (int s2 = 0; s2 < c->world; ++s2) {
        if (s2 == c->rank) continue;
        if ((rc = c->get_file(c->path("a", n, s2, c->rank), host, true))) {
            err = pagdev::last_error();
            return rc;
        }
        if (host.size() != A.rb[s2]) {
            err = "exchange: " + std::to_string(host.size()) + " bytes from rank " + std::to_string(s2) + ", " + std::to_string(A.rb[s2]) + " expected";
            return PAG_EFAULT;
        }
        if (A.rb[s2] && !(hip_ok(hipMemcpyAsync((char *)A.recv + ro[s2], host.data(), A.rb[s2], hipMemcpyHostToDevice, c->stream), "copy to device") &&
                          hip_ok(hipStreamSynchronize(c->stream), "copy to device")))
            return PAG_EFAULT;
    }
    return PAG_OK;
}

// (the caller's data — send arrays written on another stream — must be complete: the caller synchronises its stream first)
int xchg_begin(pag_comm *c, Xchg &X, std::vector<XchgArr> arrs) {
    X.c = c;
    X.rc = PAG_OK;
    X.err.clear();
    X.dev_ms = 0.f;
    PAG_HIP_TRY(hipSetDevice(c->device));
    for (const XchgArr &A : arrs) {
        if (A.sb[c->rank] != A.rb[c->rank]) return PAG_EINVAL;
        for (int r = 0; r < c->world; ++r)
            if (r != c->rank) c->bytes_sent += A.sb[r];
    }
    X.t_begin = now_s();
    if (c->use_rccl) {
        if (c->world > 1) {  // (only a complete set of ranks enters RCCL: see pag_comm_all_to_all_v)
            const int brc = pag_comm_barrier(c);
            if (brc) return brc;
        }
        if (!X.ev0) {
            PAG_HIP_TRY(hipEventCreate(&X.ev0));
            PAG_HIP_TRY(hipEventCreate(&X.ev1));
        }
        PAG_HIP_TRY(hipEventRecord(X.ev0, c->stream));
        ncclResult_t rc = c->rccl.GroupStart();
        for (const XchgArr &A : arrs) {
            uint64_t so = 0, ro = 0;
            for (int r = 0; r < c->world && rc == ncclSuccess; ++r) {
                if (A.sb[r]) rc = c->rccl.Send((const char *)A.send + so, A.sb[r], ncclUint8, r, c->comm, c->stream);
                if (rc == ncclSuccess && A.rb[r]) rc = c->rccl.Recv((char *)A.recv + ro, A.rb[r], ncclUint8, r, c->comm, c->stream);
                so += A.sb[r];
                ro += A.rb[r];
            }
        }
        const ncclResult_t rc2 = c->rccl.GroupEnd();
        if (rc != ncclSuccess || rc2 != ncclSuccess) {
            pagdev::set_error("exchange: RCCL error %s", c->rccl.GetErrorString ? c->rccl.GetErrorString(rc != ncclSuccess ? rc : rc2) : "?");
            return PAG_EFAULT;
        }
        PAG_HIP_TRY(hipEventRecord(X.ev1, c->stream));
        X.active = true;
        return PAG_OK;
    }
    std::vector<uint64_t> seqs;
    for (size_t a = 0; a < arrs.size(); ++a) seqs.push_back(c->seq++);
    try {
        X.th = std::thread([c, &X, arrs = std::move(arrs), seqs]() {
            if (hipSetDevice(c->device) != hipSuccess) {
                X.rc = PAG_EFAULT;
                X.err = "hipSetDevice in the exchange thread";
            }
            for (size_t a = 0; a < arrs.size() && X.rc == PAG_OK; ++a) X.rc = xchg_files(c, seqs[a], arrs[a], X.err);
            X.t_done = now_s();
        });
    } catch (const std::exception &e) {  // (no thread to be had: nothing was started)
        pagdev::set_error("exchange: cannot start the helper thread: %s", e.what());
        return PAG_EFAULT;
    }
    X.active = true;
    return PAG_OK;
}
int xchg_end(Xchg &X) {
    if (!X.active) return PAG_OK;
    X.active = false;
    if (X.c->use_rccl) {
        PAG_HIP_TRY(hipStreamSynchronize(X.c->stream));
        PAG_HIP_TRY(hipEventElapsedTime(&X.dev_ms, X.ev0, X.ev1));
        X.t_done = X.t_begin + X.dev_ms * 1e-3;
        return PAG_OK;
    }
    X.th.join();
    if (X.rc != PAG_OK) pagdev::set_error("%s", X.err.c_str());
    return X.rc;
}
struct XchgGuard {  // (an error return between begin and end must not leave a joinable thread behind)
    Xchg &X;
    ~XchgGuard() {
        if (X.active && !X.c->use_rccl && X.th.joinable()) X.th.join();
        // (RCCL: the grouped sends / receives of an exchange that was begun and never ended read and write the chunk buffers the
        // caller is about to free — they are waited for first; a peer that has died makes this wait end with the communicator's
        // own error, which the caller is returning anyway)
        if (X.active && X.c->use_rccl) (void)hipStreamSynchronize(X.c->stream);
        if (X.ev0) hipEventDestroy(X.ev0);
        if (X.ev1) hipEventDestroy(X.ev1);
    }
};
struct DevTemp {  // device arrays of one call, freed when it returns
    std::vector<void *> ptrs;
    void *get(size_t bytes) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
            pagdev::set_error("hipMalloc(%zu) failed in the sharded build", bytes);
            return nullptr;
        }
        ptrs.push_back(p);
        return p;
    }
    void release() {
        for (void *p : ptrs) hipFree(p);
        ptrs.clear();
    }
    ~DevTemp() { release(); }
};
}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------------------------
// pag_shard_run: the sharded build of one block behind one call (every rank calls it with the same prepared input).
//   extract own read range -> all-to-all(v) of the tuple / edge streams -> K2-K4 on the owned k-mer range -> every rank's
//   region selected from the slice -> all-to-all(v) of the selections, received straight into the handle's graph buffers ->
//   region set, the build's memory released.
// regions[world]: what every rank traverses (the same on all ranks); total: the block's count lines.
// ---------------------------------------------------------------------------------------------------------------------
static int shard_run(pag_graph *g, pag_comm *c, const pag_build_input *in, const pag_region *regions, pag_build_stats *total);
int pag_shard_run(pag_graph *g, pag_comm *c, const pag_build_input *in, const pag_region *regions, pag_build_stats *total) {
    if (!g || !c || !in || !regions) return PAG_EINVAL;
    const int rc = shard_run(g, c, in, regions, total);
    if (rc != PAG_OK) pag_comm_abort(c, pag_last_error());  // the peers learn why instead of waiting for this rank
    return rc;
}
static int shard_run(pag_graph *g, pag_comm *c, const pag_build_input *in, const pag_region *regions, pag_build_stats *total) {
    using namespace pagdev;
    const int W = c->world, me = c->rank;
    int rc;
    // PAG_SHARD_TIMING=1: one line per rank and block on stderr — seconds per stage of this call (the device idle at every
    // boundary) and the payload that left the rank in each of the two bulk exchanges
    const bool timing = env_timing();
    double lap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t wire[2] = {0, 0};
    auto now = [&]() {
        if (timing) hipDeviceSynchronize();
        return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    double t_prev = now();
    auto mark = [&](int i) {
        if (!timing) return;
        const double t = now();
        lap[i] += t - t_prev;
        t_prev = t;
    };
    // ---- extraction in CHUNKS, every chunk's partitioned streams on their way to the owners while the next chunk is extracted
    //      (SURVEY.md 8e).  PAG_SHARD_CHUNKS (default 4; 1: the whole range at once).  A chunk's streams are copied out of the
    //      extraction's slots (the next chunk overwrites them) and received into arrays of their own; when the last chunk has
    //      arrived the owner lays the pieces out as [pass 1: rank 0 chunk 0, chunk 1 .. rank 1 ..][pass 2: ..] — the ranks' read
    //      ranges and, inside a rank, its chunks are contiguous in emission order, so that is the canonical order again.
    hipStream_t s = g->stream;
    const uint64_t n_reads = in->reads.n_seqs, r_lo = n_reads * (uint64_t)me / (uint64_t)W, r_hi = n_reads * ((uint64_t)me + 1) / (uint64_t)W;
    int C = 4;
    if (const char *e = std::getenv("PAG_SHARD_CHUNKS")) C = std::max(1, std::min(64, std::atoi(e)));
    std::vector<std::vector<uint64_t>> allc((size_t)C, std::vector<uint64_t>(4 * (size_t)W * W));
    auto cnt = [&](int ch, int src, int dst, int q) { return allc[(size_t)ch][((size_t)src * W + dst) * 4 + q]; };
    struct ChunkBufs {
        void *send[4] = {nullptr, nullptr, nullptr, nullptr};  // tkey tval ekey eval of this rank's chunk, partitioned by owner
        void *recv[4] = {nullptr, nullptr, nullptr, nullptr};  // what the ranks sent of theirs
    };
    std::vector<ChunkBufs> cb((size_t)C);
    DevTemp tmp;
    Xchg X;
    XchgGuard xguard{X};
    double t_extract = 0, t_xfer = 0, t_loop0 = 0, t_hidden = 0;
    const uint64_t sent0 = pag_comm_bytes_sent(c);
    {
        const double tl0 = now_s();
        double prev_begin = 0;
        for (int ch = 0; ch < C; ++ch) {
            const uint64_t lo = r_lo + (r_hi - r_lo) * (uint64_t)ch / (uint64_t)C, hi = r_lo + (r_hi - r_lo) * ((uint64_t)ch + 1) / (uint64_t)C;
            std::vector<uint64_t> counts(4 * (size_t)W);
            const double te0 = now_s();
            if ((rc = pag_shard_extract_range(g, in, lo, hi, (uint32_t)W, counts.data()))) return rc;
            const int ts = g->shard_in0[0] ? 30 : 32, es = g->shard_in0[1] ? 34 : 36;
            const uint64_t Tc = g->shard_x[0], Ec = g->shard_x[1];
            const size_t esz[4] = {4, 8, 4, 8};
            const void *src[4] = {g->pool[ts].p, g->pool[ts + 1].p, g->pool[es].p, g->pool[es + 1].p};
            for (int a = 0; a < 4; ++a) {
                const uint64_t n = a < 2 ? Tc : Ec;
                if (!(cb[ch].send[a] = tmp.get((n + 1) * esz[a]))) return PAG_ENOMEM;
                if (n) PAG_HIP_TRY(hipMemcpyAsync(cb[ch].send[a], src[a], n * esz[a], hipMemcpyDeviceToDevice, s));
            }
            PAG_HIP_TRY(hipStreamSynchronize(s));
            const double te1 = now_s();
            t_extract += te1 - te0;
            if ((rc = pag_comm_all_gather(c, counts.data(), counts.size() * 8, allc[(size_t)ch].data()))) return rc;
            // the chunk before this one has been travelling beside this extraction
            const bool was_active = X.active;
            if ((rc = xchg_end(X))) return rc;
            if (was_active) {
                t_xfer += X.t_done - prev_begin;
                t_hidden += std::max(0.0, std::min(X.t_done, te1) - std::max(prev_begin, te0));
            }
            std::vector<XchgArr> arrs(4);
            for (int a = 0; a < 4; ++a) {
                const int q0 = a < 2 ? 0 : 2;
                uint64_t tot = 0;
                arrs[a].sb.resize(W);
                arrs[a].rb.resize(W);
                for (int r = 0; r < W; ++r) {
                    arrs[a].sb[r] = (cnt(ch, me, r, q0) + cnt(ch, me, r, q0 + 1)) * esz[a];
                    arrs[a].rb[r] = (cnt(ch, r, me, q0) + cnt(ch, r, me, q0 + 1)) * esz[a];
                    tot += arrs[a].rb[r];
                }
                if (!(cb[ch].recv[a] = tmp.get(tot + esz[a]))) return PAG_ENOMEM;
                arrs[a].send = cb[ch].send[a];
                arrs[a].recv = cb[ch].recv[a];
            }
            if ((rc = xchg_begin(c, X, std::move(arrs)))) return rc;
            prev_begin = X.t_begin;
        }
        if ((rc = xchg_end(X))) return rc;
        t_xfer += X.t_done - prev_begin;
        t_loop0 = now_s() - tl0;
    }
    mark(0);
    uint64_t nT = 0, nE = 0, t1 = 0, e1 = 0;
    for (int ch = 0; ch < C; ++ch)
        for (int r = 0; r < W; ++r) {
            nT += cnt(ch, r, me, 0) + cnt(ch, r, me, 1);
            nE += cnt(ch, r, me, 2) + cnt(ch, r, me, 3);
            t1 += cnt(ch, r, me, 0);
            e1 += cnt(ch, r, me, 2);
        }
    for (int ch = 0; ch < C; ++ch)  // (the chunks' send copies are done with)
        for (int a = 0; a < 4; ++a) {
            auto it = std::find(tmp.ptrs.begin(), tmp.ptrs.end(), cb[ch].send[a]);
            if (it != tmp.ptrs.end()) {
                hipFree(*it);
                tmp.ptrs.erase(it);
            }
        }
    DevBuf b_lk(g, 208), b_lv(g, 209), b_lek(g, 210), b_lev(g, 211);
    if ((rc = b_lk.alloc((nT + 1) * 4)) || (rc = b_lv.alloc((nT + 1) * 8)) || (rc = b_lek.alloc((nE + 1) * 4)) || (rc = b_lev.alloc((nE + 1) * 8))) return rc;
    mark(1);
    {
        // [pass 1 from rank 0: chunk 0, chunk 1 ..] .. [pass 1 from rank W-1 ..] [pass 2 from rank 0 ..] ..; a chunk's received array
        // is [from rank 0: pass 1, pass 2][from rank 1: ..]
        void *dst_of[4] = {b_lk.p, b_lv.p, b_lek.p, b_lev.p};
        const size_t esz[4] = {4, 8, 4, 8};
        for (int a = 0; a < 4; ++a) {
            const int q0 = a < 2 ? 0 : 2;
            std::vector<uint64_t> src_at((size_t)C, 0);  // read position in every chunk's received array
            uint64_t d1 = 0, d2 = a < 2 ? t1 : e1;
            for (int r = 0; r < W; ++r)
                for (int ch = 0; ch < C; ++ch) {
                    const uint64_t p1 = cnt(ch, r, me, q0), p2 = cnt(ch, r, me, q0 + 1);
                    const char *from = (const char *)cb[ch].recv[a] + src_at[(size_t)ch] * esz[a];
                    if (p1) PAG_HIP_TRY(hipMemcpyAsync((char *)dst_of[a] + d1 * esz[a], from, p1 * esz[a], hipMemcpyDeviceToDevice, s));
                    if (p2) PAG_HIP_TRY(hipMemcpyAsync((char *)dst_of[a] + d2 * esz[a], from + p1 * esz[a], p2 * esz[a], hipMemcpyDeviceToDevice, s));
                    src_at[(size_t)ch] += p1 + p2;
                    d1 += p1;
                    d2 += p2;
                }
        }
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }
    tmp.release();
    mark(2);
    wire[0] = pag_comm_bytes_sent(c) - sent0;
    pag_build_stats mine{};
    if ((rc = pag_shard_build(g, b_lk.as<uint32_t>(), b_lv.as<uint64_t>(), nT, t1, b_lek.as<uint32_t>(), b_lev.as<uint64_t>(), nE, e1, in->eps, &mine))) return rc;
    mark(3);
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
    // the send buffers are sized once for what the selections of all destinations usually add up to — the slice itself plus
    // the landing zones and halos that several ranks take (1.2x at N = 4, measured) — so that the loop below neither
    // allocates nor copies what it has already gathered; a block that needs more grows them as before
    for (int a = 0; a < 7; ++a) {
        const uint64_t n_slice = a < 4 ? g->n_t : g->n_e;
        DevBuf b(g, arrs[a].slot);
        if ((rc = b.alloc((n_slice + n_slice / 2 + 1024) * arrs[a].esz))) return rc;
    }
    // (PAG_SHARD_PIPELINE=0: all selections first, then seven whole all-to-all(v)s, as until round 5)
    const bool pipeline = env_int("PAG_SHARD_PIPELINE", 1) != 0;
    uint64_t T = 0, E = 0;
    std::vector<pag_build_stats> from_owner(W);  // the statistics of what owner o selected for this rank
    DevBuf imp[7] = {DevBuf(g, 52), DevBuf(g, 53), DevBuf(g, 54), DevBuf(g, 55), DevBuf(g, 56), DevBuf(g, 57), DevBuf(g, 58)};
    double t_select = 0, t_rxfer = 0, t_rhidden = 0, t_loop1 = 0;
    const uint64_t sent1 = pag_comm_bytes_sent(c);
    if (pipeline) {
        // Step i: this owner selects the region of rank (me + i) mod W and receives, from rank (me - i) mod W, that owner's
        // selection for this rank; the selection of step i travels while the one of step i + 1 is made.  What arrives is kept
        // per owner and put in owner order (ascending k-mer ranges) when the last piece is there.
        struct StepMsg {
            uint64_t n_t, n_e;
            pag_build_stats st;
        };
        std::vector<std::vector<void *>> piece((size_t)W, std::vector<void *>(7, nullptr));  // [owner][array]
        std::vector<uint64_t> piece_t(W, 0), piece_e(W, 0);
        DevTemp rtmp;
        Xchg X;
        XchgGuard xguard{X};
        const double tl0 = now_s();
        double prev_begin = 0;
        for (int i = 0; i < W; ++i) {
            const int dst = (me + i) % W, src = (me - i + W) % W;
            const double ts0 = now_s();
            if ((rc = pag_shard_select(g, &regions[dst], &sel[dst]))) return rc;
            const void *from[7] = {sel[dst].tkey, sel[dst].tval, sel[dst].tseg, sel[dst].tcnt, sel[dst].ekey, sel[dst].eval, sel[dst].eseg};
            std::vector<uint64_t> at0 = at;
            for (int a = 0; a < 7; ++a) {
                const uint64_t n = a < 4 ? sel[dst].n_t : sel[dst].n_e;
                DevBuf b(g, arrs[a].slot);
                const uint64_t need = (at[a] + n + 1) * arrs[a].esz;
                if (b.sl->cap < need) {
                    // (grown with the old contents kept; an exchange that still reads the old array is waited for first)
                    if ((rc = xchg_end(X))) return rc;
                    void *np = nullptr;
                    const size_t want = need + need / 2 + 256;
                    PAG_HIP_TRY(hipMalloc(&np, want));
                    if (b.sl->p && at[a]) PAG_HIP_TRY(hipMemcpy(np, b.sl->p, at[a] * arrs[a].esz, hipMemcpyDeviceToDevice));
                    if (b.sl->p) {  // (as DevBuf::alloc: nothing is freed under a resident walker grid)
                        if (g->defer_free) g->deferred.push_back(b.sl->p);
                        else hipFree(b.sl->p);
                    }
                    b.sl->p = np;
                    b.sl->cap = want;
                    if (piece[(size_t)me][a]) piece[(size_t)me][a] = b.sl->p;  // (this rank's own piece lies at the front of these arrays)
                }
                if (n) PAG_HIP_TRY(hipMemcpyAsync((char *)b.sl->p + at[a] * arrs[a].esz, from[a], n * arrs[a].esz, hipMemcpyDeviceToDevice, s));
                at[a] += n;
            }
            PAG_HIP_TRY(hipStreamSynchronize(s));
            const double ts1 = now_s();
            t_select += ts1 - ts0;
            std::vector<StepMsg> msg(W);
            StepMsg mine_msg{sel[dst].n_t, sel[dst].n_e, sel[dst].stats};
            if ((rc = pag_comm_all_gather(c, &mine_msg, sizeof mine_msg, msg.data()))) return rc;
            const bool was_active = X.active;
            if ((rc = xchg_end(X))) return rc;
            if (was_active) {
                t_rxfer += X.t_done - prev_begin;
                t_rhidden += std::max(0.0, std::min(X.t_done, ts1) - std::max(prev_begin, ts0));
            }
            piece_t[(size_t)src] = msg[(size_t)src].n_t;
            piece_e[(size_t)src] = msg[(size_t)src].n_e;
            from_owner[(size_t)src] = msg[(size_t)src].st;
            if (i == 0) {  // this rank's own selection stays where it is
                for (int a = 0; a < 7; ++a) piece[(size_t)me][a] = (char *)g->pool[arrs[a].slot].p + at0[a] * arrs[a].esz;
                continue;
            }
            std::vector<XchgArr> xa(7);
            for (int a = 0; a < 7; ++a) {
                const uint64_t n_out = a < 4 ? sel[dst].n_t : sel[dst].n_e, n_in = a < 4 ? piece_t[(size_t)src] : piece_e[(size_t)src];
                if (!(piece[(size_t)src][a] = rtmp.get((n_in + 1) * arrs[a].esz))) return PAG_ENOMEM;
                xa[a].sb.assign(W, 0);
                xa[a].rb.assign(W, 0);
                xa[a].sb[(size_t)dst] = n_out * arrs[a].esz;
                xa[a].rb[(size_t)src] = n_in * arrs[a].esz;
                xa[a].send = (const char *)g->pool[arrs[a].slot].p + at0[a] * arrs[a].esz;
                xa[a].recv = piece[(size_t)src][a];
            }
            if ((rc = xchg_begin(c, X, std::move(xa)))) return rc;
            prev_begin = X.t_begin;
        }
        const bool was_active = X.active;
        if ((rc = xchg_end(X))) return rc;
        if (was_active) t_rxfer += X.t_done - prev_begin;
        t_loop1 = now_s() - tl0;
        mark(4);
        for (int o = 0; o < W; ++o) {
            T += piece_t[(size_t)o];
            E += piece_e[(size_t)o];
        }
        for (int a = 0; a < 7; ++a) {
            if ((rc = imp[a].alloc(((a < 4 ? T : E) + 1) * arrs[a].esz))) return rc;
            uint64_t off = 0;
            for (int o = 0; o < W; ++o) {  // owner order = ascending k-mer ranges
                const uint64_t n = a < 4 ? piece_t[(size_t)o] : piece_e[(size_t)o];
                if (n) PAG_HIP_TRY(hipMemcpyAsync((char *)imp[a].p + off * arrs[a].esz, piece[(size_t)o][a], n * arrs[a].esz, hipMemcpyDeviceToDevice, s));
                off += n;
            }
        }
        PAG_HIP_TRY(hipStreamSynchronize(s));
    } else {
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
                if (b.sl->p) {
                    if (g->defer_free) g->deferred.push_back(b.sl->p);
                    else hipFree(b.sl->p);
                }
                b.sl->p = np;
                b.sl->cap = want;
            }
            if (n) PAG_HIP_TRY(hipMemcpyAsync((char *)b.sl->p + at[a] * arrs[a].esz, src[a], n * arrs[a].esz, hipMemcpyDeviceToDevice, s));
            at[a] += n;
        }
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }
    mark(4);
    if ((rc = pag_comm_all_gather(c, my_sizes.data(), my_sizes.size() * 8, all_sizes.data()))) return rc;
    std::vector<pag_build_stats> all_stats((size_t)W * W);
    if ((rc = pag_comm_all_gather(c, stats_to.data(), stats_to.size() * sizeof(pag_build_stats), all_stats.data()))) return rc;
    auto size_of = [&](int owner, int dst, int which) { return all_sizes[((size_t)owner * W + dst) * 2 + which]; };
    for (int o = 0; o < W; ++o) {
        T += size_of(o, me, 0);
        E += size_of(o, me, 1);
    }
    // received straight into the buffers pag_shard_import fills (slots 52 .. 58): owner order = ascending k-mer ranges
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
        for (int o = 0; o < W; ++o) from_owner[(size_t)o] = all_stats[(size_t)o * W + me];
    }
    mark(5);
    wire[1] = pag_comm_bytes_sent(c) - sent1;
    pag_build_stats st{};
    for (int o = 0; o < W; ++o) {
        const pag_build_stats &P = from_owner[(size_t)o];
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
    mark(6);
    if (timing)
        std::fprintf(stderr,
                     "[shard timing] rank %d/%d tuples in %d chunks: extraction + owner partition %.4f s, exchange %.4f s of which %.4f s beside the next chunk's "
                     "extraction (loop %.4f s, %llu B out), owner layout %.4f s, K2-K4 %.4f s; regions %s: selections for %d ranks %.4f s, exchange %.4f s of which "
                     "%.4f s beside the next selection (loop %.4f s, %llu B out), owner order %.4f s, import+region+release %.4f s\n",
                     me, W, C, t_extract, t_xfer, t_hidden, t_loop0, (unsigned long long)wire[0], lap[1] + lap[2], lap[3], pipeline ? "pipelined" : "whole", W,
                     pipeline ? t_select : lap[4], pipeline ? t_rxfer : lap[5], t_rhidden, t_loop1, (unsigned long long)wire[1], pipeline ? lap[5] : 0.0, lap[6]);
    if (total) *total = st;
    return PAG_OK;
}

}  // extern "C"
