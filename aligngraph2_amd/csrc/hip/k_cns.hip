// k_cns.hip — pa_cns on the device: the partial-order alignment graphs of a backbone's parts, ONE WAVEFRONT PER
// PART.  The parts are independent (pa_cns.cpp:98-124 hands them to threads the same way); inside a part the algorithm is a
// chain of order-dependent list operations (AlnGraphBoost.cpp: addAln / mergeNodes / bestPath — the order of a vertex's edge
// lists decides every tie), so a part is one serial instruction stream: cns_graph.hpp, the same code the host build of
// bin/pa_cns can run (PA_CNS_BACKEND=flat).  What the device adds is many such streams side by side, each a chain of dependent
// gathers the memory system overlaps across waves.  One LANE of a wave runs a part: with a part per lane (round 5) the 64
// parts of a wave diverged at every branch and the wave executed them one after the other — 64 s for the 200 parts of a 1 Mb
// backbone at 190x against 4 s on 16 host threads (profiles/r06_pa_cns_timing.json).
//
// Memory: a part's node / edge / scratch regions come out of arrays allocated per batch; batches are cut so that a batch fits the
// byte budget (free device memory x 0.8).
#include <algorithm>
#include <vector>

#include "cns_graph.hpp"
#include "pag_device.hpp"

namespace pagdev {

__global__ __launch_bounds__(64) void cns_parts_kernel(pagcns::Arrays A, const pagcns::Part *__restrict__ parts, uint32_t n_parts, const char *__restrict__ backbone,
                                                        const pagcns::Aln *__restrict__ alns, const char *__restrict__ qpool, const char *__restrict__ tpool,
                                                        int min_weight, char *__restrict__ out, uint32_t *__restrict__ out_len, int32_t *__restrict__ part_err) {
    const uint32_t p = blockIdx.x;
    if (p >= n_parts || threadIdx.x != 0) return;
    uint32_t len = 0;
    const int err = pagcns::run_part(A, parts[p], backbone, alns, qpool, tpool, min_weight, out, &len);
    out_len[p] = err ? 0u : len;
    part_err[p] = err;
}

namespace {
struct Dev {  // device allocations of one call, freed on every way out
    std::vector<void *> ptrs;
    ~Dev() {
        for (void *p : ptrs) (void)hipFree(p);
    }
    template <typename T>
    int alloc(T **out, size_t n) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            set_error("pag_cns_consensus: out of device memory (%zu bytes)", n * sizeof(T));
            return PAG_ENOMEM;
        }
        ptrs.push_back(p);
        *out = (T *)p;
        return PAG_OK;
    }
};
}  // namespace

}  // namespace pagdev

using namespace pagdev;

extern "C" int pag_cns_consensus(int device, const char *backbone, uint64_t backbone_len, const pag_cns_part *parts, uint64_t n_parts, const pag_cns_aln *alns,
                                 uint64_t n_alns, const char *qpool, const char *tpool, uint64_t pool_bytes, int32_t min_weight, char *out, uint64_t out_bytes,
                                 uint64_t *out_off, uint32_t *out_len, int32_t *part_err) {
    static_assert(sizeof(pag_cns_aln) == sizeof(pagcns::Aln), "pag_cns_aln is pagcns::Aln");
    if (!backbone || (!parts && n_parts) || (!alns && n_alns) || !out || !out_off || !out_len || !part_err) return PAG_EINVAL;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) {
        (void)hipGetLastError();
        set_error("pag_cns_consensus: no gfx950 device %d (the consensus graphs are built on the device; PA_CNS_BACKEND=host is the host restatement)", device);
        return PAG_ENODEV;
    }
    PAG_HIP_TRY(hipSetDevice(device));
    // inputs
    Dev D;
    char *d_bb, *d_q, *d_t, *d_out;
    pagcns::Aln *d_alns;
    int rc;
    if ((rc = D.alloc(&d_bb, backbone_len + 16)) || (rc = D.alloc(&d_q, pool_bytes + 16)) || (rc = D.alloc(&d_t, pool_bytes + 16)) || (rc = D.alloc(&d_alns, n_alns + 1)) ||
        (rc = D.alloc(&d_out, out_bytes + 16)))
        return rc;
    PAG_HIP_TRY(hipMemcpy(d_bb, backbone, backbone_len, hipMemcpyHostToDevice));
    if (pool_bytes) {
        PAG_HIP_TRY(hipMemcpy(d_q, qpool, pool_bytes, hipMemcpyHostToDevice));
        PAG_HIP_TRY(hipMemcpy(d_t, tpool, pool_bytes, hipMemcpyHostToDevice));
    }
    if (n_alns) PAG_HIP_TRY(hipMemcpy(d_alns, alns, n_alns * sizeof(pagcns::Aln), hipMemcpyHostToDevice));
    // output places
    uint64_t oo = 0;
    for (uint64_t p = 0; p < n_parts; ++p) {
        out_off[p] = oo;
        oo += parts[p].out_cap;
    }
    out_off[n_parts] = oo;
    if (oo > out_bytes) {
        set_error("pag_cns_consensus: the output buffer holds %llu bytes, the parts may need %llu", (unsigned long long)out_bytes, (unsigned long long)oo);
        return PAG_EINVAL;
    }
    // batches of parts whose regions fit the budget
    size_t free_b = 0, total_b = 0;
    PAG_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    const uint64_t budget = (uint64_t)((double)free_b * 0.8);
    constexpr uint64_t NODE_B = 1 + 1 + 4 + 4 + 4 * 7 + 4 + 4, EDGE_B = 4 * 6 + 4 + 1, AUX_B = 4;
    for (uint64_t p0 = 0; p0 < n_parts;) {
        uint64_t nn = 0, ne = 0, na = 0, p1 = p0;
        std::vector<pagcns::Part> hp;
        while (p1 < n_parts) {
            const pag_cns_part &s = parts[p1];
            const uint64_t need = (nn + s.node_cap) * NODE_B + (ne + s.edge_cap) * EDGE_B + (na + s.aux_cap) * AUX_B;
            if (need > budget && p1 > p0) break;
            if (need > budget) {
                set_error("pag_cns_consensus: part %llu alone needs %llu bytes of device memory, %llu are free", (unsigned long long)p1, (unsigned long long)need,
                          (unsigned long long)free_b);
                return PAG_ENOMEM;
            }
            pagcns::Part q{};
            q.bb_off = s.bb_off;
            q.bb_len = s.bb_len;
            q.n_aln = s.n_aln;
            q.aln_first = s.aln_first;
            q.node_base = nn;
            q.edge_base = ne;
            q.aux_base = na;
            q.out_off = out_off[p1];
            q.node_cap = s.node_cap;
            q.edge_cap = s.edge_cap;
            q.aux_cap = s.aux_cap;
            q.out_cap = s.out_cap;
            hp.push_back(q);
            nn += s.node_cap;
            ne += s.edge_cap;
            na += s.aux_cap;
            ++p1;
        }
        Dev B;
        pagcns::Arrays A{};
        pagcns::Part *d_parts;
        uint32_t *d_len;
        int32_t *d_err;
        if ((rc = B.alloc(&A.n_base, nn)) || (rc = B.alloc(&A.n_flags, nn)) || (rc = B.alloc(&A.n_cov, nn)) || (rc = B.alloc(&A.n_weight, nn)) || (rc = B.alloc(&A.n_bb, nn)) ||
            (rc = B.alloc(&A.n_oh, nn)) || (rc = B.alloc(&A.n_ot, nn)) || (rc = B.alloc(&A.n_ih, nn)) || (rc = B.alloc(&A.n_it, nn)) || (rc = B.alloc(&A.n_oc, nn)) ||
            (rc = B.alloc(&A.n_ic, nn)) || (rc = B.alloc(&A.n_score, nn)) || (rc = B.alloc(&A.n_best, nn)) || (rc = B.alloc(&A.e_src, ne)) || (rc = B.alloc(&A.e_dst, ne)) ||
            (rc = B.alloc(&A.e_on, ne)) || (rc = B.alloc(&A.e_op, ne)) || (rc = B.alloc(&A.e_in, ne)) || (rc = B.alloc(&A.e_ip, ne)) || (rc = B.alloc(&A.e_count, ne)) ||
            (rc = B.alloc(&A.e_vis, ne)) || (rc = B.alloc(&A.aux, na)) || (rc = B.alloc(&d_parts, hp.size())) || (rc = B.alloc(&d_len, hp.size())) ||
            (rc = B.alloc(&d_err, hp.size())))
            return rc;
        PAG_HIP_TRY(hipMemcpy(d_parts, hp.data(), hp.size() * sizeof(pagcns::Part), hipMemcpyHostToDevice));
        const uint32_t n = (uint32_t)hp.size();
        cns_parts_kernel<<<dim3(n), dim3(64), 0, 0>>>(A, d_parts, n, d_bb, d_alns, d_q, d_t, min_weight, d_out, d_len, d_err);
        PAG_HIP_TRY(hipGetLastError());
        PAG_HIP_TRY(hipDeviceSynchronize());
        PAG_HIP_TRY(hipMemcpy(out_len + p0, d_len, (size_t)n * 4, hipMemcpyDeviceToHost));
        PAG_HIP_TRY(hipMemcpy(part_err + p0, d_err, (size_t)n * 4, hipMemcpyDeviceToHost));
        p0 = p1;
    }
    if (oo) PAG_HIP_TRY(hipMemcpy(out, d_out, oo, hipMemcpyDeviceToHost));
    return PAG_OK;
}
