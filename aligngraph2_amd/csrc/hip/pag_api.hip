// pag_api.hip — the C ABI of libpagraph_hip.so (include/pagraph_hip.h) and the device pipeline of
// PositionProcessor::process (reference PAGraph/src/tools/position/PositionProcessor.cpp:79-151):
//
//   coverage filter -> column index -> K1 count (pass 1, pass 2) -> scans -> K1 emit (pass 1, pass 2)
//   -> K2 stable sort of position tuples by k-mer, K2 sort of edge tuples by `from`
//   -> K3 cluster + sort positions per k-mer, K4 sort + unique edges per k-mer.
//
// The reference runs {extract, mergeEdge, cluster} twice (pass 1, then pass 2 on top of pass 1's
// leaders).  Both greedy procedures are prefix-closed, so ONE sort + ONE cluster over the
// concatenation [pass-1 stream] ++ [pass-2 stream] yields the identical graph (k34_segments.hip), and
// the reference's six count lines are recovered from the stream lengths and two counters.
//
// No CPU fallback: without a usable gfx950 device every entry point returns PAG_ENODEV.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <chrono>
#include <vector>

#include "pag_device.hpp"

using namespace pagdev;

#include "pag_graph_impl.hpp"
#include "walker_grid.hpp"

namespace {

// device view of an input array: uploads when the caller's array is in host memory
template <typename T>
int stage(const T *src, uint64_t n, bool on_device, DevBuf &own, const T **out, hipStream_t s) {
    if (on_device) {
        *out = src;
        return PAG_OK;
    }
    int rc = own.alloc((size_t)n * sizeof(T) + 64);
    if (rc != PAG_OK) return rc;
    if (n) PAG_HIP_TRY(hipMemcpyAsync(own.p, src, (size_t)n * sizeof(T), hipMemcpyHostToDevice, s));
    *out = own.as<T>();
    return PAG_OK;
}

int pick_device(int ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device (hipGetDeviceCount)");
        return PAG_ENODEV;
    }
    if (ordinal < 0 || ordinal >= n) {
        set_error("device ordinal %d out of range (%d devices)", ordinal, n);
        return PAG_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return PAG_ENODEV;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 only", ordinal, prop.gcnArchName);
        return PAG_ENODEV;
    }
    if (hipSetDevice(ordinal) != hipSuccess) return PAG_ENODEV;
    return PAG_OK;
}

void free_graph_results(pag_graph *g) {  // the memory stays in the pool
    g->tkey = nullptr;
    g->tval = nullptr;
    g->tseg = nullptr;
    g->tcnt = nullptr;
    g->ekey = nullptr;
    g->eval = nullptr;
    g->eseg = nullptr;
    g->n_t = g->n_e = 0;
    g->tg_ready = false;
    g->view_pruned = g->view_off = false;
    g->view_orient.clear();
    g->regional = false;
    g->region_ref_iv.clear();
    g->region_ref_open.clear();
    // CONTRACT (pagraph_hip.h, pag_travel_path): the pinned storage behind the paths (path_store, fetch_chunks) is NOT
    // released or re-pinned here, nor in pag_reserve_walk_arena / pag_prepare / pag_process: the previous block's host half
    // (pagh_traverse_begin, pagraph_driver's HostHalf) reads the raw pointers it was handed while this runs for the next
    // block.  Only the next pag_travel (which waits for that host half) and pag_destroy may touch that memory.
    g->path_off.clear();
    g->path_len.clear();
    g->path_valid.clear();
    g->path_ptr.clear();
}

__global__ void chunk_counts(const pag_aln *__restrict__ aln, uint64_t n, uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pag_aln a = aln[i];
    bool used = (a.flags & PAG_ALN_ELIGIBLE) && a.query != PAG_NONE;
    out[i] = used ? (a.n_cols + 1023u) / 1024u : 0u;
}

// records per (owner, pass) of a stream in emission order: the first n1 records are pass-1 records
__global__ void owner_counts(const uint32_t *__restrict__ key, uint64_t n, uint64_t n1, int shift, unsigned long long *__restrict__ out /*[8][2]*/) {
    __shared__ unsigned int h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[((key[i] >> shift) & 7u) * 2u + (i < n1 ? 0u : 1u)], 1u);
    __syncthreads();
    if (threadIdx.x < 16 && h[threadIdx.x]) atomicAdd(&out[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

__global__ void edge_counts(const uint32_t *__restrict__ samples, uint64_t n, uint32_t *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = samples[i] ? samples[i] - 1u : 0u;
}

}  // namespace

extern "C" {

const char *pag_last_error(void) { return last_error(); }

int pag_device_available(void) { return pick_device(0) == PAG_OK ? 1 : 0; }

namespace {
__global__ void k_warm(uint32_t *p) {
    if (p) *p = 1u;
}
}  // namespace
int pag_device_warm(int device_ordinal) {
    int rc = pick_device(device_ordinal);
    if (rc != PAG_OK) return rc;
    // the context, the first allocation, the code object of this library: a third of a second of a cold process
    uint32_t *p = nullptr;
    if (hipMalloc((void **)&p, 256) != hipSuccess) return PAG_ENOMEM;
    k_warm<<<dim3(1), dim3(64)>>>(p);
    hipError_t e = hipDeviceSynchronize();
    hipFree(p);
    if (e != hipSuccess) {
        set_error("pag_device_warm: %s", hipGetErrorString(e));
        return PAG_EFAULT;
    }
    return PAG_OK;
}

pag_graph *pag_create(const uint64_t *codes, uint64_t n_codes, uint32_t k, int device_ordinal, int *err) {
    int rc = PAG_OK;
    pag_graph *g = nullptr;
    if (k == 0 || k > 16 || (!codes && n_codes)) {
        set_error("pag_create: k must be in 1..16 (got %u)", k);
        rc = PAG_EINVAL;
    }
    if (rc == PAG_OK) rc = pick_device(device_ordinal);
    if (rc == PAG_OK) {
        g = new (std::nothrow) pag_graph();
        if (!g) rc = PAG_ENOMEM;
    }
    if (rc == PAG_OK) {
        g->device = device_ordinal;
        g->k = k;
        // PABruijnGraph::PABruijnGraph: sort + unique of every word (PABruijnGraph.cpp:32-34).  The set itself is all
        // that is needed: a bitmap over the 4^k code space (read k-mers are masked to 2k bits, larger words can never
        // match) and the number of distinct words; the bits are set by a pool of threads, no copy and no sort.
        const uint64_t space = 1ull << (2 * k);
        const uint64_t words = (space + 31) / 32;
        std::vector<uint32_t> bits((size_t)words, 0u);
        std::vector<uint64_t> outside;
        {
            unsigned T = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
            if (n_codes < (1u << 20)) T = 1;
            std::vector<std::vector<uint64_t>> out_t(T);
            auto work = [&](unsigned t) {
                const uint64_t lo = n_codes * t / T, hi = n_codes * (t + 1) / T;
                for (uint64_t i = lo; i < hi; ++i) {
                    const uint64_t c = codes[i];
                    if (c < space) __atomic_fetch_or(&bits[(size_t)(c >> 5)], 1u << (c & 31), __ATOMIC_RELAXED);
                    else out_t[t].push_back(c);
                }
            };
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
            work(0);
            for (auto &th : pool) th.join();
            for (auto &v : out_t) outside.insert(outside.end(), v.begin(), v.end());
        }
        std::sort(outside.begin(), outside.end());
        outside.erase(std::unique(outside.begin(), outside.end()), outside.end());
        uint64_t in_space = 0;
        for (uint32_t wbits : bits) in_space += (uint64_t)__builtin_popcount(wbits);
        g->n_solid = in_space + outside.size();
        g->all_solid = in_space == space;
        hipError_t e = hipStreamCreate(&g->stream);
        if (e == hipSuccess) e = hipMalloc((void **)&g->solid_bits, (size_t)words * 4 + 64);
        if (e == hipSuccess) e = hipMemcpy(g->solid_bits, bits.data(), (size_t)words * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("pag_create: %s", hipGetErrorString(e));
            rc = PAG_EFAULT;
        }
    }
    if (rc != PAG_OK && g) {
        pag_destroy(g);
        g = nullptr;
    }
    if (err) *err = rc;
    return g;
}

pag_graph *pag_create_from_bitmap(const uint32_t *bits, uint64_t n_solid, uint32_t k, int bits_on_device,
                                  int device_ordinal, int *err) {
    int rc = PAG_OK;
    pag_graph *g = nullptr;
    if (k == 0 || k > 16 || !bits) {
        set_error("pag_create_from_bitmap: k must be in 1..16 and bits non-null");
        rc = PAG_EINVAL;
    }
    if (rc == PAG_OK) rc = pick_device(device_ordinal);
    if (rc == PAG_OK) {
        g = new (std::nothrow) pag_graph();
        if (!g) rc = PAG_ENOMEM;
    }
    if (rc == PAG_OK) {
        g->device = device_ordinal;
        g->k = k;
        g->n_solid = n_solid;
        const uint64_t space = 1ull << (2 * k);
        const uint64_t words = (space + 31) / 32;
        g->all_solid = n_solid == space;
        hipError_t e = hipStreamCreate(&g->stream);
        if (e == hipSuccess) e = hipMalloc((void **)&g->solid_bits, (size_t)words * 4 + 64);
        if (e == hipSuccess)
            e = hipMemcpy(g->solid_bits, bits, (size_t)words * 4, bits_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            set_error("pag_create_from_bitmap: %s", hipGetErrorString(e));
            rc = PAG_EFAULT;
        }
    }
    if (rc != PAG_OK && g) {
        pag_destroy(g);
        g = nullptr;
    }
    if (err) *err = rc;
    return g;
}

void pag_destroy(pag_graph *g) {
    if (!g) return;
    hipSetDevice(g->device);
    free_graph_results(g);
    for (auto &sl : g->pool)
        if (sl.p) hipFree(sl.p);
    for (auto &sl : g->cpool)
        if (sl.p) hipFree(sl.p);
    for (void *q : g->deferred) hipFree(q);
    if (g->wq_host) hipHostFree(g->wq_host);
    if (g->path_store) hipHostFree(g->path_store);
    if (g->deliver_stream) hipStreamDestroy(g->deliver_stream);
    if (g->pin_host) hipHostFree(g->pin_host);
    for (void *q : g->fetch_chunks) hipHostFree(q);
    g->fetch_chunks.clear();
    g->fetch_chunk_bytes.clear();
    if (g->walk_arena) hipFree(g->walk_arena);
    if (g->wq_next) hipFree(g->wq_next);
    for (hipStream_t ws : g->walk_streams) hipStreamDestroy(ws);
    if (g->solid_bits) hipFree(g->solid_bits);
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
}

uint64_t pag_solid_count(const pag_graph *g) { return g ? g->n_solid : 0; }

int pag_reset(pag_graph *g) {
    if (!g) return PAG_EINVAL;
    hipSetDevice(g->device);
    free_graph_results(g);
    g->stats = pag_build_stats{};
    return PAG_OK;
}

}  // extern "C"

namespace {

struct Extracted {  // what extract_stage leaves in the pool slots 30 / 31 / 34 / 35
    uint64_t T = 0, E = 0, T1 = 0, E1 = 0;  // tuples / edges emitted, of which by pass 1 (they come first)
};

// Both extraction passes of PositionProcessor::process (PositionProcessor.cpp:86-124) for the reads at emission positions
// [emit_lo, emit_hi) of in->emit_order: coverage filter, column index, K1 count, scans, K1 emit.  The streams
// [pass-1 tuples] ++ [pass-2 tuples] (and edges likewise) are left in canonical order in slots 30/31 (tuples) and 34/35 (edges).
int extract_stage(pag_graph *g, const pag_build_input *in, uint64_t emit_lo, uint64_t emit_hi, Extracted *out, hipEvent_t ev_begin,
                  hipEvent_t ev_end) {
    hipStream_t s = g->stream;
    const bool dev = in->on_device != 0;
    const uint32_t n_reads_all = (uint32_t)in->reads.n_seqs;
    const uint32_t n_reads = (uint32_t)(emit_hi - emit_lo);  // reads of this call
    const uint64_t n_jobs = 4ull * n_reads;                   // 2 passes x 2 strands
    int rc;
    if (ev_begin) PAG_HIP_TRY(hipEventRecord(ev_begin, s));

    // ---- inputs -> device
    DevBuf b_roff(g, 0), b_rlen(g, 1), b_packed(g, 2), b_order(g, 3), b_aln1(g, 4), b_q1(g, 5), b_d1(g, 6), b_aln2(g, 7), b_q2(g, 8), b_d2(g, 9), b_ctg(g, 10), b_eoff(g, 11), b_ent(g, 12), b_ref(g, 13);
    const uint64_t *d_roff;
    const uint32_t *d_rlen, *d_order, *d_d1, *d_d2, *d_eoff, *d_ent;
    const uint8_t *d_packed;
    const pag_aln *d_aln1, *d_aln2;
    const uint64_t *d_q1, *d_q2;
    const pag_ctg *d_ctg;
    const pag_ref *d_ref;
    if ((rc = stage(in->reads.byte_off, n_reads_all, dev, b_roff, &d_roff, s))) return rc;
    if ((rc = stage(in->reads.len, n_reads_all, dev, b_rlen, &d_rlen, s))) return rc;
    if ((rc = stage(in->reads.packed, in->reads.packed_bytes, dev, b_packed, &d_packed, s))) return rc;
    if ((rc = stage(in->emit_order, n_reads_all, dev, b_order, &d_order, s))) return rc;
    if ((rc = stage(in->read_to_ctg.aln, in->read_to_ctg.n_aln, dev, b_aln1, &d_aln1, s))) return rc;
    if ((rc = stage(in->read_to_ctg.query_off, (uint64_t)n_reads_all + 1, dev, b_q1, &d_q1, s))) return rc;
    if ((rc = stage(in->read_to_ctg.diff, in->read_to_ctg.n_diff_words, dev, b_d1, &d_d1, s))) return rc;
    if ((rc = stage(in->read_to_ref.aln, in->read_to_ref.n_aln, dev, b_aln2, &d_aln2, s))) return rc;
    if ((rc = stage(in->read_to_ref.query_off, (uint64_t)n_reads_all + 1, dev, b_q2, &d_q2, s))) return rc;
    if ((rc = stage(in->read_to_ref.diff, in->read_to_ref.n_diff_words, dev, b_d2, &d_d2, s))) return rc;
    if ((rc = stage(in->ctgs, in->n_ctgs, dev, b_ctg, &d_ctg, s))) return rc;
    if ((rc = stage(in->ctg_ent_off, in->n_ctg_ent_off, dev, b_eoff, &d_eoff, s))) return rc;
    if ((rc = stage(in->ctg_ent, in->n_ctg_ent, dev, b_ent, &d_ent, s))) return rc;
    if ((rc = stage(in->refs, in->n_refs, dev, b_ref, &d_ref, s))) return rc;
    d_order += emit_lo;

    // the reference table is tiny and needed on the host for scratch sizing
    std::vector<pag_ref> refs_host((size_t)in->n_refs);
    if (in->n_refs) {
        if (dev) PAG_HIP_TRY(hipMemcpy(refs_host.data(), in->refs, in->n_refs * sizeof(pag_ref), hipMemcpyDeviceToHost));
        else std::memcpy(refs_host.data(), in->refs, in->n_refs * sizeof(pag_ref));
    }

    // ---- coverage filter of pass 2 (over ALL read->reference alignments, whichever reads this call extracts)
    const uint64_t n_aln1 = in->read_to_ctg.n_aln, n_aln2 = in->read_to_ref.n_aln;
    DevBuf b_covok(g, 14), b_covtmp(g, 15);
    if ((rc = b_covok.alloc(n_aln2 + 16))) return rc;
    size_t cov_bytes = cov_tmp_bytes(refs_host.data(), in->n_refs);
    if ((rc = b_covtmp.alloc(cov_bytes))) return rc;
    if ((rc = launch_cov_filter(d_aln2, n_aln2, d_ref, refs_host.data(), in->n_refs, in->cov_filter,
                                b_covok.as<uint8_t>(), b_covtmp.p, cov_bytes, s)))
        return rc;

    // ---- column index of both alignment databases
    DevBuf b_cc(g, 16), b_cio1(g, 17), b_cio2(g, 18), b_ci1(g, 19), b_ci2(g, 20), b_scan(g, 21), b_tot(g, 22);
    uint64_t n_alnmax = std::max(n_aln1, n_aln2);
    if ((rc = b_cc.alloc((n_alnmax + 1) * 4))) return rc;
    if ((rc = b_cio1.alloc((n_aln1 + 1) * 8))) return rc;
    if ((rc = b_cio2.alloc((n_aln2 + 1) * 8))) return rc;
    if ((rc = b_scan.alloc(scan_tmp_bytes(std::max<uint64_t>(n_alnmax, n_jobs) + 1)))) return rc;
    if ((rc = b_tot.alloc(64))) return rc;
    uint64_t *d_tot = b_tot.as<uint64_t>();
    uint64_t n_ci[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        const pag_aln *al = pass == 0 ? d_aln1 : d_aln2;
        uint64_t na = pass == 0 ? n_aln1 : n_aln2;
        DevBuf &off = pass == 0 ? b_cio1 : b_cio2;
        if (na) {
            chunk_counts<<<dim3((unsigned)((na + 255) / 256)), dim3(256), 0, s>>>(al, na, b_cc.as<uint32_t>());
            if ((rc = scan_u32_to_u64(b_cc.as<uint32_t>(), off.as<uint64_t>(), na, d_tot + pass, b_scan.p, s))) return rc;
            PAG_HIP_TRY(hipMemcpyAsync(&n_ci[pass], d_tot + pass, 8, hipMemcpyDeviceToHost, s));
        }
    }
    PAG_HIP_TRY(hipStreamSynchronize(s));  // (one round trip for both sizes)
    for (int pass = 0; pass < 2; ++pass) {
        const pag_aln *al = pass == 0 ? d_aln1 : d_aln2;
        uint64_t na = pass == 0 ? n_aln1 : n_aln2;
        DevBuf &off = pass == 0 ? b_cio1 : b_cio2;
        DevBuf &ci = pass == 0 ? b_ci1 : b_ci2;
        if ((rc = ci.alloc((n_ci[pass] + 1) * sizeof(uint2)))) return rc;
        if ((rc = launch_colidx(al, na, pass == 0 ? d_d1 : d_d2, off.as<uint64_t>(), ci.as<uint2>(), s))) return rc;
    }

    // ---- K1 count
    DevBuf b_js(g, 25), b_jt(g, 26), b_je(g, 27), b_toff(g, 28), b_eoff2(g, 29);
    if ((rc = b_js.alloc((n_jobs + 1) * 4))) return rc;
    if ((rc = b_jt.alloc((n_jobs + 1) * 4))) return rc;
    if ((rc = b_je.alloc((n_jobs + 1) * 4))) return rc;
    if ((rc = b_toff.alloc((n_jobs + 1) * 8))) return rc;
    if ((rc = b_eoff2.alloc((n_jobs + 1) * 8))) return rc;
    DevBuf b_smask(g, 51);
    if (!g->all_solid) {
        // mask entries: 2 strands x one u16 per 16 bases (= per packed 32-bit word of the reads)
        if ((rc = b_smask.alloc((in->reads.packed_bytes / 4 + 16) * 2 * sizeof(uint16_t) * solid_mask_slices()))) return rc;
    }
    ExtractArgs xa[2];
    for (int pass = 0; pass < 2; ++pass) {
        ExtractArgs &a = xa[pass];
        a = ExtractArgs{};
        a.read_off = d_roff;
        a.read_len = d_rlen;
        a.packed = d_packed;
        a.emit_order = d_order;
        a.n_reads = n_reads;
        a.aln = pass == 0 ? d_aln1 : d_aln2;
        a.query_off = pass == 0 ? d_q1 : d_q2;
        a.diff = pass == 0 ? d_d1 : d_d2;
        a.colidx_off = (pass == 0 ? b_cio1 : b_cio2).as<uint64_t>();
        a.colidx = (pass == 0 ? b_ci1 : b_ci2).as<uint2>();
        a.cov_ok = pass == 0 ? nullptr : b_covok.as<uint8_t>();
        a.pass = pass;
        a.topk = pass == 0 ? in->topk_ctg : in->topk_ref;
        a.ctgs = d_ctg;
        a.ctg_ent_off = d_eoff;
        a.ctg_ent = d_ent;
        a.refs = d_ref;
        a.solid_bits = g->solid_bits;
        a.solid_mask = g->all_solid ? nullptr : b_smask.as<uint16_t>();
        a.solid_mask_stride = (in->reads.packed_bytes / 4 + 16) * 2;
        a.all_solid = g->all_solid;
        a.k = g->k;
        a.outer = in->outer_sample;
        a.job_samples = b_js.as<uint32_t>();
        a.job_tuples = b_jt.as<uint32_t>();
        a.job_base = (uint32_t)(pass * 2ull * n_reads);
    }
    // the order pass 0's jobs run in
    DevBuf b_pk0(g, 39), b_pv0(g, 40), b_pk1(g, 41), b_pv1(g, 42), b_ptmp(g, 23), b_perm(g, 24);
    {
        const uint64_t nj0 = 2ull * n_reads;
        if (nj0 >= 4096) {
            if ((rc = b_pk0.alloc(nj0 * 4)) || (rc = b_pv0.alloc(nj0 * 8)) || (rc = b_pk1.alloc(nj0 * 4)) || (rc = b_pv1.alloc(nj0 * 8)) ||
                (rc = b_ptmp.alloc(sort_tmp_bytes(nj0))) || (rc = b_perm.alloc(nj0 * 4)))
                return rc;
            if ((rc = launch_exec_perm(xa[0], b_pk0.as<uint32_t>(), b_pv0.as<uint64_t>(), b_pk1.as<uint32_t>(), b_pv1.as<uint64_t>(), b_ptmp.p,
                                       b_perm.as<uint32_t>(), s)))
                return rc;
            xa[0].exec_perm = b_perm.as<uint32_t>();
        }
    }
    if (!g->all_solid && (rc = launch_solid_mask(xa[0], d_aln2, d_q2, b_smask.as<uint16_t>(), s))) return rc;
    for (int pass = 0; pass < 2; ++pass)
        if ((rc = launch_extract(xa[pass], false, s))) return rc;
    uint64_t T = 0, E = 0, T1 = 0, E1 = 0;
    if (n_jobs) {
        edge_counts<<<dim3((unsigned)((n_jobs + 255) / 256)), dim3(256), 0, s>>>(b_js.as<uint32_t>(), n_jobs,
                                                                                 b_je.as<uint32_t>());
        if ((rc = scan_u32_to_u64(b_jt.as<uint32_t>(), b_toff.as<uint64_t>(), n_jobs, d_tot + 2, b_scan.p, s))) return rc;
        if ((rc = scan_u32_to_u64(b_je.as<uint32_t>(), b_eoff2.as<uint64_t>(), n_jobs, d_tot + 3, b_scan.p, s))) return rc;
        PAG_HIP_TRY(hipMemcpyAsync(&T, d_tot + 2, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipMemcpyAsync(&E, d_tot + 3, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipMemcpyAsync(&T1, b_toff.as<uint64_t>() + 2ull * n_reads, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipMemcpyAsync(&E1, b_eoff2.as<uint64_t>() + 2ull * n_reads, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }

    // ---- K1 emit
    DevBuf b_tk0(g, 30), b_tv0(g, 31), b_ek0(g, 34), b_ev0(g, 35);
    if ((rc = b_tk0.alloc((T + 1) * 4))) return rc;
    if ((rc = b_tv0.alloc((T + 1) * 8))) return rc;
    if ((rc = b_ek0.alloc((E + 1) * 4))) return rc;
    if ((rc = b_ev0.alloc((E + 1) * 8))) return rc;
    for (int pass = 0; pass < 2; ++pass) {
        ExtractArgs &a = xa[pass];
        a.tuple_off = b_toff.as<uint64_t>();
        a.edge_off = b_eoff2.as<uint64_t>();
        a.tkey = b_tk0.as<uint32_t>();
        a.tval = b_tv0.as<uint64_t>();
        a.ekey = b_ek0.as<uint32_t>();
        a.eval = b_ev0.as<uint64_t>();
        if ((rc = launch_extract(a, true, s))) return rc;
    }
    if (ev_end) PAG_HIP_TRY(hipEventRecord(ev_end, s));
    out->T = T;
    out->E = E;
    out->T1 = T1;
    out->E1 = E1;
    return PAG_OK;
}

// mergeEdge / mergeKmerPosition / sortKmerPosition for both passes at once (see the head of this file) over the streams in
// slots 30/31 (tuples: T records, the first T1 from pass 1) and 34/35 (edges): K2 sorts, K3, K4; leaves the finished graph
// in the handle.  The reference's count lines are additive over k-mers, so a handle that owns only a k-mer range of the
// graph reports its share of each line.
int build_stage(pag_graph *g, uint32_t eps, const Extracted &x, pag_build_stats *st_out, hipEvent_t *ev /*[4]: sort begin, sort end, cluster end, edges end*/) {
    hipStream_t s = g->stream;
    const uint64_t T = x.T, E = x.E, T1 = x.T1, E1 = x.E1;
    int rc;
    DevBuf b_tk0(g, 30), b_tv0(g, 31), b_tk1(g, 32), b_tv1(g, 33), b_ek0(g, 34), b_ev0(g, 35), b_ek1(g, 36), b_ev1(g, 37);
    // (the first buffers were filled by the caller; alloc() only hands their pointers back)
    if ((rc = b_tk0.alloc((T + 1) * 4)) || (rc = b_tv0.alloc((T + 1) * 8)) || (rc = b_ek0.alloc((E + 1) * 4)) || (rc = b_ev0.alloc((E + 1) * 8))) return rc;
    // the second value buffer of each stream doubles as u64[n] + u32[n] scratch for long segments
    if ((rc = b_tk1.alloc((T + 1) * 4))) return rc;
    if ((rc = b_tv1.alloc((T + 1) * 12))) return rc;
    if ((rc = b_ek1.alloc((E + 1) * 4))) return rc;
    if ((rc = b_ev1.alloc((E + 1) * 12))) return rc;
    PAG_HIP_TRY(hipEventRecord(ev[0], s));

    // ---- K2 sorts.  (Measured in round 4 and not kept: the edge stream's sort on a stream and a host thread of its own, so that
    // its histogram / scan launches run beside the other stream's scatter pass — the sort stage took the same 24.6 ms, two
    // persistent scatter kernels never share a CU, and the step was 30 ms slower.)
    DevBuf b_sorttmp(g, 38);
    if ((rc = b_sorttmp.alloc(sort_tmp_bytes(std::max(T, E))))) return rc;
    int t_in0 = 1, e_in0 = 1, passes = 0;
    float ms_scatter_t = 0.f, ms_scatter_e = 0.f;
    if ((rc = sort_pairs(b_tk0.as<uint32_t>(), b_tv0.as<uint64_t>(), b_tk1.as<uint32_t>(), b_tv1.as<uint64_t>(), T,
                         2 * (int)g->k, b_sorttmp.p, &t_in0, s, &ms_scatter_t, &passes)))
        return rc;
    if ((rc = sort_pairs(b_ek0.as<uint32_t>(), b_ev0.as<uint64_t>(), b_ek1.as<uint32_t>(), b_ev1.as<uint64_t>(), E,
                         2 * (int)g->k, b_sorttmp.p, &e_in0, s, &ms_scatter_e, nullptr)))
        return rc;
    PAG_HIP_TRY(hipEventRecord(ev[1], s));
    // make the sorted data live in the (k0, v0)-sized buffers, the spare (v1: 12 B/record) is scratch
    DevBuf *tk = t_in0 ? &b_tk0 : &b_tk1, *tv = t_in0 ? &b_tv0 : &b_tv1;
    DevBuf *ek = e_in0 ? &b_ek0 : &b_ek1, *evb = e_in0 ? &b_ev0 : &b_ev1;
    DevBuf b_tscr(g, 43), b_escr(g, 44);
    uint64_t *t_scratch, *e_scratch;
    if (t_in0) {
        t_scratch = b_tv1.as<uint64_t>();
    } else {  // sorted values sit in the big buffer: give the segment kernels a fresh scratch
        if ((rc = b_tscr.alloc((T + 1) * 12))) return rc;
        t_scratch = b_tscr.as<uint64_t>();
    }
    if (e_in0) {
        e_scratch = b_ev1.as<uint64_t>();
    } else {
        if ((rc = b_escr.alloc((E + 1) * 12))) return rc;
        e_scratch = b_escr.as<uint64_t>();
    }

    // ---- K3 / K4
    DevBuf b_tseg(g, 45), b_tcnt(g, 46), b_eseg(g, 47), b_long(g, 48), b_lcnt(g, 49), b_ctr(g, 50);
    if ((rc = b_tseg.alloc((T + 1) * 4))) return rc;
    if ((rc = b_tcnt.alloc((T + 1) * 2))) return rc;
    if ((rc = b_eseg.alloc((E + 1) * 4))) return rc;
    if ((rc = b_long.alloc((std::max(T, E) / 32 + 2) * 8))) return rc;
    if ((rc = b_lcnt.alloc(64))) return rc;
    if ((rc = b_ctr.alloc(128))) return rc;
    uint64_t *ctr_dev = b_ctr.as<uint64_t>();  // [0..3] cluster counters, [4..7] edge counters
    ClusterOut co{b_tseg.as<uint32_t>(), b_tcnt.as<uint16_t>(), ctr_dev};
    // (the short path's width: 64-record masks where the average k-mer segment is long — 30x coverage and more; results do not depend on it)
    const bool wide = g->n_solid != 0 && T > 10 * g->n_solid;  // (records per solid k-mer: 7.7 at 20x — the 32-record form is the faster one —, 11.5 at 30x, 15.4 at 40x)
    if ((rc = launch_cluster(tk->as<uint32_t>(), tv->as<uint64_t>(), t_scratch, T, eps, co, b_long.as<uint64_t>(),
                             b_lcnt.as<uint32_t>(), s, wide)))
        return rc;
    PAG_HIP_TRY(hipEventRecord(ev[2], s));
    EdgeOut eo{b_eseg.as<uint32_t>(), ctr_dev + 4};
    if ((rc = launch_edges(ek->as<uint32_t>(), evb->as<uint64_t>(), e_scratch, E, eo, b_long.as<uint64_t>(),
                           b_lcnt.as<uint32_t>(), s, wide)))
        return rc;
    uint64_t ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    PAG_HIP_TRY(hipMemcpyAsync(ctr, b_ctr.p, 64, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipEventRecord(ev[3], s));
    PAG_HIP_TRY(hipStreamSynchronize(s));  // (the only host round trip of the stage)

    // ---- the reference's count lines (PositionProcessor.cpp:126-142)
    pag_build_stats st{};
    const uint64_t T2 = T - T1, E2 = E - E1;
    const uint64_t L_ctg = ctr[0], L_all = ctr[1], U_all = ctr[4], U_1 = ctr[5];
    st.n_tuples[0] = T1;
    st.n_tuples[1] = T2;
    st.n_edges[0] = E1;
    st.n_edges[1] = E2;
    st.merge_edge[0] = E1 - U_1;
    st.total_pos[0] = T1;
    st.merge_pos[0] = T1 - L_ctg;
    st.merge_edge[1] = U_1 + E2 - U_all;
    st.total_pos[1] = L_ctg + T2;
    st.merge_pos[1] = L_ctg + T2 - L_all;
    st.n_nodes = ctr[2];
    st.n_pos = L_all;
    st.n_uniq_edges = U_all;
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev[0], ev[1]);
    st.ms_sort = ms;
    hipEventElapsedTime(&ms, ev[1], ev[2]);
    st.ms_cluster = ms;
    hipEventElapsedTime(&ms, ev[2], ev[3]);
    st.ms_edges = ms;
    st.ms_sort_kernel = ms_scatter_t;
    st.sort_records = T;
    (void)passes;
    (void)ms_scatter_e;

    // ---- keep the finished graph
    g->n_t = T;
    g->n_e = E;
    g->tkey = tk->as<uint32_t>();
    g->tval = tv->as<uint64_t>();
    g->tseg = b_tseg.as<uint32_t>();
    g->tcnt = b_tcnt.as<uint16_t>();
    g->ekey = ek->as<uint32_t>();
    g->eval = evb->as<uint64_t>();
    g->eseg = b_eseg.as<uint32_t>();
    g->stats = st;
    *st_out = st;
    return PAG_OK;
}

struct EventSet {
    hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int create() {
        for (auto &x : e) PAG_HIP_TRY(hipEventCreate(&x));
        return PAG_OK;
    }
    ~EventSet() {
        for (auto x : e)
            if (x) hipEventDestroy(x);
    }
};

void keep_debug_streams(pag_graph *g, uint64_t T, uint64_t E) {
    const char *keep = std::getenv("PAG_DEBUG_KEEP_STREAMS");
    g->dbg_tkey.clear();
    g->dbg_tval.clear();
    g->dbg_ekey.clear();
    g->dbg_eval.clear();
    if (!(keep && keep[0] == '1')) return;
    g->dbg_tkey.resize((size_t)T);
    g->dbg_tval.resize((size_t)T);
    g->dbg_ekey.resize((size_t)E);
    g->dbg_eval.resize((size_t)E);
    hipStreamSynchronize(g->stream);
    if (T) hipMemcpy(g->dbg_tkey.data(), g->pool[30].p, T * 4, hipMemcpyDeviceToHost);
    if (T) hipMemcpy(g->dbg_tval.data(), g->pool[31].p, T * 8, hipMemcpyDeviceToHost);
    if (E) hipMemcpy(g->dbg_ekey.data(), g->pool[34].p, E * 4, hipMemcpyDeviceToHost);
    if (E) hipMemcpy(g->dbg_eval.data(), g->pool[35].p, E * 8, hipMemcpyDeviceToHost);
}

int check_process_args(const pag_graph *g, const pag_build_input *in) {
    if (!g || !in) return PAG_EINVAL;
    if (in->outer_sample < 1 || in->outer_sample > 7) {
        set_error("outer_sample must be 1..7");
        return PAG_EINVAL;
    }
    if (in->reads.n_seqs >= 0x3FFFFFFFull) {
        set_error("too many reads for one launch");
        return PAG_EINVAL;
    }
    return PAG_OK;
}

}  // namespace

extern "C" {

int pag_process(pag_graph *g, const pag_build_input *in, pag_build_stats *stats) {
    int rc = check_process_args(g, in);
    if (rc != PAG_OK) return rc;
    const bool timing = env_timing();
    const auto wall0 = std::chrono::steady_clock::now();
    auto wall_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count(); };
    PAG_HIP_TRY(hipSetDevice(g->device));
    free_graph_results(g);
    EventSet ev;
    if ((rc = ev.create())) return rc;
    Extracted x;
    if ((rc = extract_stage(g, in, 0, in->reads.n_seqs, &x, ev.e[0], ev.e[1]))) return rc;
    keep_debug_streams(g, x.T, x.E);
    pag_build_stats st{};
    if ((rc = build_stage(g, in->eps, x, &st, ev.e + 2))) return rc;
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev.e[0], ev.e[1]);
    st.ms_extract = ms;
    hipEventElapsedTime(&ms, ev.e[0], ev.e[5]);
    st.ms_total = ms;
    if (timing) fprintf(stderr, "[timing] pag_process wall %.1f ms (device %.1f ms)\n", wall_ms(), (double)ms);
    g->stats = st;
    if (stats) *stats = st;
    return PAG_OK;
}

extern "C" int pag_reserve_walk_arena(pag_graph *g, uint64_t contig_bases) {
    if (!g) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    // (~1.2 KB per contig base is what pag_travel's own estimate comes to at sequencing coverage; it caps itself likewise)
    size_t want = (size_t)contig_bases * 1200 + (64u << 20);
    size_t free_b = 0, total_b = 0;
    // (the same cap as pag_travel's own estimate: an arena reserved here must not be thrown away there as too small)
    // (processes that share the device — PAG_DEVICE_SHARERS, the one-GPU test box — share that cap)
    const size_t sharers = env_device_sharers();
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) want = std::min(want, (free_b + g->walk_arena_cap) * 2 / 5 / sharers);
    // the pinned memory the fetched paths of the walks' jobs land in (64 MB chunks kept by the handle, pag_travel's
    // fetch_alloc): ~14 bytes per contig base at sequencing coverage; a cold process otherwise pins them one by one between
    // the walks' first fetches (5 ms each)
    {
        const size_t FETCH_CHUNK = 64u << 20;
        size_t have = 0;
        for (size_t b : g->fetch_chunk_bytes) have += b;
        const size_t want_pinned = std::min<size_t>((size_t)contig_bases * 14, (size_t)4 << 30);
        while (have < want_pinned) {
            void *q = nullptr;
            if (hipHostMalloc(&q, FETCH_CHUNK, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                break;  // (pag_travel asks again when it needs the memory, and reports a failure there)
            }
            g->fetch_chunks.push_back(q);
            g->fetch_chunk_bytes.push_back(FETCH_CHUNK);
            have += FETCH_CHUNK;
        }
    }
    // (and the streams the walker waves are launched on: ~6 ms each to create, here beside the caller's other work)
    pagdev::WalkerGrid::prepare_streams(g, 4);
    if (g->walk_arena_cap >= want) return PAG_OK;
    if (g->walk_arena) hipFree(g->walk_arena);
    g->walk_arena = nullptr;
    g->walk_arena_cap = 0;
    if (hipMalloc(&g->walk_arena, want) != hipSuccess) {
        g->walk_arena = nullptr;
        (void)hipGetLastError();
        return PAG_OK;  // (pag_travel tries again, or works without the arena)
    }
    g->walk_arena_cap = want;
    return PAG_OK;
}

// ---- one graph built by several GPUs (see include/pagraph_hip.h)
static int log2_shards(uint32_t n) { return n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : n == 8 ? 3 : -1; }

int pag_shard_extract(pag_graph *g, const pag_build_input *in, uint32_t shard, uint32_t n_shards, uint64_t *counts) {
    int rc = check_process_args(g, in);
    if (rc != PAG_OK) return rc;
    if (log2_shards(n_shards) < 0 || shard >= n_shards) {
        set_error("pag_shard_extract: n_shards must be 1, 2, 4 or 8 (got %u) and shard below it", n_shards);
        return PAG_EINVAL;
    }
    const uint64_t n = in->reads.n_seqs;
    return pag_shard_extract_range(g, in, n * shard / n_shards, n * (shard + 1ull) / n_shards, n_shards, counts);
}

int pag_shard_extract_range(pag_graph *g, const pag_build_input *in, uint64_t lo, uint64_t hi, uint32_t n_shards, uint64_t *counts) {
    int rc = check_process_args(g, in);
    if (rc != PAG_OK) return rc;
    const int lg = log2_shards(n_shards);
    if (lg < 0 || !counts || 2 * (int)g->k < lg || lo > hi || hi > in->reads.n_seqs) {
        set_error("pag_shard_extract_range: n_shards must be 1, 2, 4 or 8 (got %u), the reads [%llu, %llu) of %llu", n_shards, (unsigned long long)lo,
                  (unsigned long long)hi, (unsigned long long)in->reads.n_seqs);
        return PAG_EINVAL;
    }
    PAG_HIP_TRY(hipSetDevice(g->device));
    free_graph_results(g);
    hipStream_t s = g->stream;
    Extracted x;
    if ((rc = extract_stage(g, in, lo, hi, &x, nullptr, nullptr))) return rc;
    keep_debug_streams(g, x.T, x.E);
    g->shard_x[0] = x.T;
    g->shard_x[1] = x.E;
    g->shard_in0[0] = g->shard_in0[1] = 1;
    for (uint32_t i = 0; i < 4 * n_shards; ++i) counts[i] = 0;
    if (n_shards == 1) {
        counts[0] = x.T1;
        counts[1] = x.T - x.T1;
        counts[2] = x.E1;
        counts[3] = x.E - x.E1;
        return PAG_OK;
    }
    // records per (owner, pass), then ONE stable radix pass on the owner bits of the k-mer code
    const int shift = 2 * (int)g->k - lg;
    DevBuf b_tk0(g, 30), b_tv0(g, 31), b_tk1(g, 32), b_tv1(g, 33), b_ek0(g, 34), b_ev0(g, 35), b_ek1(g, 36), b_ev1(g, 37), b_sorttmp(g, 38), b_ctr(g, 50);
    if ((rc = b_tk0.alloc((x.T + 1) * 4)) || (rc = b_tv0.alloc((x.T + 1) * 8)) || (rc = b_ek0.alloc((x.E + 1) * 4)) || (rc = b_ev0.alloc((x.E + 1) * 8)) ||
        (rc = b_tk1.alloc((x.T + 1) * 4)) || (rc = b_tv1.alloc((x.T + 1) * 12)) || (rc = b_ek1.alloc((x.E + 1) * 4)) || (rc = b_ev1.alloc((x.E + 1) * 12)) ||
        (rc = b_sorttmp.alloc(sort_tmp_bytes(std::max(x.T, x.E)))) || (rc = b_ctr.alloc(512)))
        return rc;
    unsigned long long *ctr = b_ctr.as<unsigned long long>();
    PAG_HIP_TRY(hipMemsetAsync(ctr, 0, 256, s));
    if (x.T) owner_counts<<<dim3(1024), dim3(256), 0, s>>>(b_tk0.as<uint32_t>(), x.T, x.T1, shift, ctr);
    if (x.E) owner_counts<<<dim3(1024), dim3(256), 0, s>>>(b_ek0.as<uint32_t>(), x.E, x.E1, shift, ctr + 16);
    int t_in0 = 1, e_in0 = 1;
    if ((rc = sort_pairs(b_tk0.as<uint32_t>(), b_tv0.as<uint64_t>(), b_tk1.as<uint32_t>(), b_tv1.as<uint64_t>(), x.T, lg, b_sorttmp.p, &t_in0, s, nullptr,
                         nullptr, shift)))
        return rc;
    if ((rc = sort_pairs(b_ek0.as<uint32_t>(), b_ev0.as<uint64_t>(), b_ek1.as<uint32_t>(), b_ev1.as<uint64_t>(), x.E, lg, b_sorttmp.p, &e_in0, s, nullptr,
                         nullptr, shift)))
        return rc;
    unsigned long long h[32];
    PAG_HIP_TRY(hipMemcpyAsync(h, ctr, 256, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    for (uint32_t o = 0; o < n_shards; ++o) {
        counts[4 * o + 0] = h[2 * o];
        counts[4 * o + 1] = h[2 * o + 1];
        counts[4 * o + 2] = h[16 + 2 * o];
        counts[4 * o + 3] = h[16 + 2 * o + 1];
    }
    g->shard_in0[0] = t_in0;
    g->shard_in0[1] = e_in0;
    return PAG_OK;
}

int pag_shard_take(pag_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *ekey, uint64_t *eval) {
    if (!g) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const uint64_t T = g->shard_x[0], E = g->shard_x[1];
    const int ts = g->shard_in0[0] ? 30 : 32, es = g->shard_in0[1] ? 34 : 36;
    if (T) PAG_HIP_TRY(hipMemcpyAsync(tkey, g->pool[ts].p, T * 4, hipMemcpyDeviceToDevice, s));
    if (T) PAG_HIP_TRY(hipMemcpyAsync(tval, g->pool[ts + 1].p, T * 8, hipMemcpyDeviceToDevice, s));
    if (E) PAG_HIP_TRY(hipMemcpyAsync(ekey, g->pool[es].p, E * 4, hipMemcpyDeviceToDevice, s));
    if (E) PAG_HIP_TRY(hipMemcpyAsync(eval, g->pool[es + 1].p, E * 8, hipMemcpyDeviceToDevice, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    return PAG_OK;
}

int pag_shard_take_part(pag_graph *g, uint64_t t_off, uint64_t t_n, uint32_t *tkey, uint64_t *tval, uint64_t e_off, uint64_t e_n, uint32_t *ekey,
                        uint64_t *eval) {
    if (!g) return PAG_EINVAL;
    const uint64_t T = g->shard_x[0], E = g->shard_x[1];
    if (t_off > T || t_n > T - t_off || e_off > E || e_n > E - e_off || (t_n && (!tkey || !tval)) || (e_n && (!ekey || !eval))) {
        set_error("pag_shard_take_part: [%llu, +%llu) of %llu tuples, [%llu, +%llu) of %llu edges", (unsigned long long)t_off, (unsigned long long)t_n,
                  (unsigned long long)T, (unsigned long long)e_off, (unsigned long long)e_n, (unsigned long long)E);
        return PAG_EINVAL;
    }
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const int ts = g->shard_in0[0] ? 30 : 32, es = g->shard_in0[1] ? 34 : 36;
    if (t_n) PAG_HIP_TRY(hipMemcpyAsync(tkey, (const uint32_t *)g->pool[ts].p + t_off, t_n * 4, hipMemcpyDeviceToDevice, s));
    if (t_n) PAG_HIP_TRY(hipMemcpyAsync(tval, (const uint64_t *)g->pool[ts + 1].p + t_off, t_n * 8, hipMemcpyDeviceToDevice, s));
    if (e_n) PAG_HIP_TRY(hipMemcpyAsync(ekey, (const uint32_t *)g->pool[es].p + e_off, e_n * 4, hipMemcpyDeviceToDevice, s));
    if (e_n) PAG_HIP_TRY(hipMemcpyAsync(eval, (const uint64_t *)g->pool[es + 1].p + e_off, e_n * 8, hipMemcpyDeviceToDevice, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    return PAG_OK;
}

int pag_shard_build(pag_graph *g, const uint32_t *tkey, const uint64_t *tval, uint64_t n_t, uint64_t t1, const uint32_t *ekey,
                    const uint64_t *eval, uint64_t n_e, uint64_t e1, uint32_t eps, pag_build_stats *stats) {
    if (!g || t1 > n_t || e1 > n_e || (n_t && (!tkey || !tval)) || (n_e && (!ekey || !eval))) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    free_graph_results(g);
    hipStream_t s = g->stream;
    int rc;
    DevBuf b_tk0(g, 30), b_tv0(g, 31), b_ek0(g, 34), b_ev0(g, 35);
    if ((rc = b_tk0.alloc((n_t + 1) * 4)) || (rc = b_tv0.alloc((n_t + 1) * 8)) || (rc = b_ek0.alloc((n_e + 1) * 4)) || (rc = b_ev0.alloc((n_e + 1) * 8))) return rc;
    if (n_t) PAG_HIP_TRY(hipMemcpyAsync(b_tk0.p, tkey, n_t * 4, hipMemcpyDeviceToDevice, s));
    if (n_t) PAG_HIP_TRY(hipMemcpyAsync(b_tv0.p, tval, n_t * 8, hipMemcpyDeviceToDevice, s));
    if (n_e) PAG_HIP_TRY(hipMemcpyAsync(b_ek0.p, ekey, n_e * 4, hipMemcpyDeviceToDevice, s));
    if (n_e) PAG_HIP_TRY(hipMemcpyAsync(b_ev0.p, eval, n_e * 8, hipMemcpyDeviceToDevice, s));
    EventSet ev;
    if ((rc = ev.create())) return rc;
    Extracted x;
    x.T = n_t;
    x.E = n_e;
    x.T1 = t1;
    x.E1 = e1;
    pag_build_stats st{};
    if ((rc = build_stage(g, eps, x, &st, ev.e + 2))) return rc;
    g->stats = st;
    if (stats) *stats = st;
    return PAG_OK;
}

int pag_shard_export(const pag_graph *g, pag_shard_slice *out) {
    if (!g || !out) return PAG_EINVAL;
    out->n_t = g->n_t;
    out->n_e = g->n_e;
    out->tkey = g->tkey;
    out->tval = g->tval;
    out->tseg = g->tseg;
    out->tcnt = g->tcnt;
    out->ekey = g->ekey;
    out->eval = g->eval;
    out->eseg = g->eseg;
    out->stats = g->stats;
    return PAG_OK;
}

int pag_shard_take_slice(pag_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *tseg, uint16_t *tcnt, uint32_t *ekey, uint64_t *eval,
                         uint32_t *eseg) {
    if (!g) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const uint64_t T = g->n_t, E = g->n_e;
    if (T) {
        PAG_HIP_TRY(hipMemcpyAsync(tkey, g->tkey, T * 4, hipMemcpyDeviceToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(tval, g->tval, T * 8, hipMemcpyDeviceToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(tseg, g->tseg, T * 4, hipMemcpyDeviceToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(tcnt, g->tcnt, T * 2, hipMemcpyDeviceToDevice, s));
    }
    if (E) {
        PAG_HIP_TRY(hipMemcpyAsync(ekey, g->ekey, E * 4, hipMemcpyDeviceToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(eval, g->eval, E * 8, hipMemcpyDeviceToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(eseg, g->eseg, E * 4, hipMemcpyDeviceToDevice, s));
    }
    PAG_HIP_TRY(hipStreamSynchronize(s));
    return PAG_OK;
}

int pag_shard_import(pag_graph *g, const pag_shard_slice *parts, uint32_t n_parts, pag_build_stats *total) {
    if (!g || (!parts && n_parts)) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    uint64_t T = 0, E = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        T += parts[p].n_t;
        E += parts[p].n_e;
    }
    // the imported streams live in slots of their own: the parts may be the handle's own slice (slots 30..47)
    int rc;
    DevBuf b_tk(g, 52), b_tv(g, 53), b_ts(g, 54), b_tc(g, 55), b_ek(g, 56), b_ev(g, 57), b_es(g, 58);
    if ((rc = b_tk.alloc((T + 1) * 4)) || (rc = b_tv.alloc((T + 1) * 8)) || (rc = b_ts.alloc((T + 1) * 4)) || (rc = b_tc.alloc((T + 1) * 2)) ||
        (rc = b_ek.alloc((E + 1) * 4)) || (rc = b_ev.alloc((E + 1) * 8)) || (rc = b_es.alloc((E + 1) * 4)))
        return rc;
    pag_build_stats st{};
    uint64_t at = 0, ae = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const pag_shard_slice &P = parts[p];
        if (P.n_t) {
            PAG_HIP_TRY(hipMemcpyAsync(b_tk.as<uint32_t>() + at, P.tkey, P.n_t * 4, hipMemcpyDeviceToDevice, s));
            PAG_HIP_TRY(hipMemcpyAsync(b_tv.as<uint64_t>() + at, P.tval, P.n_t * 8, hipMemcpyDeviceToDevice, s));
            PAG_HIP_TRY(hipMemcpyAsync(b_ts.as<uint32_t>() + at, P.tseg, P.n_t * 4, hipMemcpyDeviceToDevice, s));
            PAG_HIP_TRY(hipMemcpyAsync(b_tc.as<uint16_t>() + at, P.tcnt, P.n_t * 2, hipMemcpyDeviceToDevice, s));
        }
        if (P.n_e) {
            PAG_HIP_TRY(hipMemcpyAsync(b_ek.as<uint32_t>() + ae, P.ekey, P.n_e * 4, hipMemcpyDeviceToDevice, s));
            PAG_HIP_TRY(hipMemcpyAsync(b_ev.as<uint64_t>() + ae, P.eval, P.n_e * 8, hipMemcpyDeviceToDevice, s));
            PAG_HIP_TRY(hipMemcpyAsync(b_es.as<uint32_t>() + ae, P.eseg, P.n_e * 4, hipMemcpyDeviceToDevice, s));
        }
        at += P.n_t;
        ae += P.n_e;
        for (int q = 0; q < 2; ++q) {
            st.merge_edge[q] += P.stats.merge_edge[q];
            st.total_pos[q] += P.stats.total_pos[q];
            st.merge_pos[q] += P.stats.merge_pos[q];
            st.n_tuples[q] += P.stats.n_tuples[q];
            st.n_edges[q] += P.stats.n_edges[q];
        }
        st.n_nodes += P.stats.n_nodes;
        st.n_pos += P.stats.n_pos;
        st.n_uniq_edges += P.stats.n_uniq_edges;
    }
    PAG_HIP_TRY(hipStreamSynchronize(s));
    free_graph_results(g);
    g->n_t = T;
    g->n_e = E;
    g->tkey = b_tk.as<uint32_t>();
    g->tval = b_tv.as<uint64_t>();
    g->tseg = b_ts.as<uint32_t>();
    g->tcnt = b_tc.as<uint16_t>();
    g->ekey = b_ek.as<uint32_t>();
    g->eval = b_ev.as<uint64_t>();
    g->eseg = b_es.as<uint32_t>();
    g->stats = st;
    if (total) *total = st;
    return PAG_OK;
}

// the graph arrays are already in the import slots (52 .. 58: received there, shard_comm.hip): the handle takes them over
int pag_shard_adopt(pag_graph *g, uint64_t T, uint64_t E, const pag_build_stats *st) {
    if (!g || !st) return PAG_EINVAL;
    free_graph_results(g);
    g->n_t = T;
    g->n_e = E;
    g->tkey = (uint32_t *)g->pool[52].p;
    g->tval = (uint64_t *)g->pool[53].p;
    g->tseg = (uint32_t *)g->pool[54].p;
    g->tcnt = (uint16_t *)g->pool[55].p;
    g->ekey = (uint32_t *)g->pool[56].p;
    g->eval = (uint64_t *)g->pool[57].p;
    g->eseg = (uint32_t *)g->pool[58].p;
    g->stats = *st;
    return PAG_OK;
}

int pag_csr_sizes(const pag_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges) {
    if (!g) return PAG_EINVAL;
    if (n_nodes) *n_nodes = g->stats.n_nodes;
    if (n_pos) *n_pos = g->stats.n_pos;
    if (n_edges) *n_edges = g->stats.n_uniq_edges;
    return PAG_OK;
}

// Compaction of the in-place segment results into a CSR.  Test/host-traversal path, not timed: the
// streams are copied back and compacted on the host.
int pag_export_csr(const pag_graph *g, pag_csr *out) {
    if (!g || !out) return PAG_EINVAL;
    const uint64_t nn = g->stats.n_nodes, np = g->stats.n_pos, ne = g->stats.n_uniq_edges;
    if (out->n_nodes < nn || out->n_pos < np || out->n_edges < ne) {
        out->n_nodes = nn;
        out->n_pos = np;
        out->n_edges = ne;
        return PAG_ERANGE;
    }
    PAG_HIP_TRY(hipSetDevice(g->device));
    const uint64_t T = g->n_t, E = g->n_e;
    std::vector<uint32_t> tkey((size_t)T), tseg((size_t)T), ekey((size_t)E), eseg((size_t)E);
    std::vector<uint64_t> tval((size_t)T), eval((size_t)E);
    std::vector<uint16_t> tcnt((size_t)T);
    if (T) {
        PAG_HIP_TRY(hipMemcpy(tkey.data(), g->tkey, T * 4, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(tseg.data(), g->tseg, T * 4, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(tval.data(), g->tval, T * 8, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(tcnt.data(), g->tcnt, T * 2, hipMemcpyDeviceToHost));
    }
    if (E) {
        PAG_HIP_TRY(hipMemcpy(ekey.data(), g->ekey, E * 4, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(eseg.data(), g->eseg, E * 4, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(eval.data(), g->eval, E * 8, hipMemcpyDeviceToHost));
    }
    uint64_t in = 0, ip = 0, ie = 0, ecur = 0;
    for (uint64_t i = 0; i < T; ++i) {
        if (i != 0 && tkey[i] == tkey[i - 1]) continue;  // not a segment head
        uint32_t code = tkey[i];
        if (in >= nn) {
            set_error("export: more nodes than counted");
            return PAG_EFAULT;
        }
        out->node_code[in] = code;
        out->pos_off[in] = ip;
        out->edge_off[in] = ie;
        for (uint32_t l = 0; l < tseg[i]; ++l) {
            if (ip >= np) {
                set_error("export: more positions than counted");
                return PAG_EFAULT;
            }
            out->pos_ctg[ip] = (uint32_t)(tval[i + l] >> 32);
            out->pos_ref[ip] = (uint32_t)tval[i + l];
            out->pos_cnt[ip] = tcnt[i + l];
            ++ip;
        }
        // edge segment of the same k-mer, if any (every `from` k-mer also owns positions)
        while (ecur < E && ekey[ecur] < code) ++ecur;
        if (ecur < E && ekey[ecur] == code) {
            for (uint32_t l = 0; l < eseg[ecur]; ++l) {
                if (ie >= ne) {
                    set_error("export: more edges than counted");
                    return PAG_EFAULT;
                }
                out->edge_to[ie] = (uint32_t)(eval[ecur + l] >> 32);
                out->edge_step[ie] = (int32_t)(((uint32_t)eval[ecur + l]) >> 1);
                ++ie;
            }
            while (ecur < E && ekey[ecur] == code) ++ecur;
        }
        ++in;
    }
    out->pos_off[in] = ip;
    out->edge_off[in] = ie;
    out->n_nodes = in;
    out->n_pos = ip;
    out->n_edges = ie;
    if (in != nn || ip != np || ie != ne) {
        set_error("export: size mismatch nodes %llu/%llu pos %llu/%llu edges %llu/%llu", (unsigned long long)in,
                  (unsigned long long)nn, (unsigned long long)ip, (unsigned long long)np, (unsigned long long)ie,
                  (unsigned long long)ne);
        return PAG_EFAULT;
    }
    return PAG_OK;
}

// test hook: raw emitted streams of the last pag_process (needs PAG_DEBUG_KEEP_STREAMS=1)
int pag_debug_stream_sizes(const pag_graph *g, uint64_t *n_tuples, uint64_t *n_edges) {
    if (!g) return PAG_EINVAL;
    *n_tuples = g->dbg_tkey.size();
    *n_edges = g->dbg_ekey.size();
    return PAG_OK;
}
int pag_debug_streams(const pag_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *ekey, uint64_t *eval) {
    if (!g) return PAG_EINVAL;
    std::memcpy(tkey, g->dbg_tkey.data(), g->dbg_tkey.size() * 4);
    std::memcpy(tval, g->dbg_tval.data(), g->dbg_tval.size() * 8);
    std::memcpy(ekey, g->dbg_ekey.data(), g->dbg_ekey.size() * 4);
    std::memcpy(eval, g->dbg_eval.data(), g->dbg_eval.size() * 8);
    return PAG_OK;
}

}  // extern "C"
