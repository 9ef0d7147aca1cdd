// Kernel-level check of K3/K4 (cluster_short/long, edges_short/long) against a sequential restatement of
// KMerAdjNode::cluster / removeDuplicate on random segmented streams.  Test infrastructure (GPU only).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../aligngraph2_amd/csrc/hip/k34_segments.hip"

namespace pagdev {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}
const char *last_error() { return ""; }
}  // namespace pagdev

static bool csim(uint32_t a, uint32_t b, uint32_t eps) {
    if (a == 0 || b == 0) return a == 0 && b == 0;
    uint32_t d = a > b ? a - b : b - a;
    return d <= eps;
}
static bool psim(uint64_t x, uint64_t y, uint32_t eps) {
    return csim((uint32_t)(x >> 32), (uint32_t)(y >> 32), eps) && csim((uint32_t)x, (uint32_t)y, eps);
}
#define CK(e)                                                                   \
    do {                                                                        \
        hipError_t r__ = (e);                                                   \
        if (r__ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r__));            \
            return 2;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char **argv) {
    const uint64_t n_seg_target = argc > 1 ? strtoull(argv[1], nullptr, 10) : 20000;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    const bool wide = argc > 3 ? atoi(argv[3]) != 0 : false;  // (the short path with 64-record masks instead of 32)
    const uint32_t eps = 10;
    std::mt19937_64 rng(seed);
    std::vector<uint32_t> key;
    std::vector<uint64_t> val, eval;
    uint32_t code = 0;
    for (uint64_t sgi = 0; sgi < n_seg_target; ++sgi) {
        code += 1 + (uint32_t)(rng() % 3);
        uint32_t r = (uint32_t)(rng() % 100);
        // (the short path takes segments of up to 64 records, the long path keeps up to 512 leaders / 1 024 edges on chip: lengths on
        // both sides of each limit; "spread": positions far apart, so that nearly every item becomes a leader)
        const bool spread = r >= 98;
        uint32_t len = r < 60 ? 1 + (uint32_t)(rng() % 8) : r < 85 ? 1 + (uint32_t)(rng() % 33) : r < 92 ? 30 + (uint32_t)(rng() % 200) : r < 96 ? 60 + (uint32_t)(rng() % 10)
                       : r < 98 ? 400 + (uint32_t)(rng() % 900) : 300 + (uint32_t)(rng() % 1200);
        uint32_t centers = 1 + (uint32_t)(rng() % 4);
        uint32_t c0[4], r0[4];
        for (int c = 0; c < 4; ++c) {
            c0[c] = 1000 + (uint32_t)(rng() % 300);
            r0[c] = 5000 + (uint32_t)(rng() % 300);
        }
        // (every 40th segment at the ends of the coordinate space: coordinates of 1 .. eps next to "no coordinate" (0), and coordinates
        // within 2 eps of 2^32, which cluster_short's two-subtraction form of the similarity test must leave to the plain predicate)
        if (sgi % 40 == 7)
            for (int c = 0; c < 4; ++c) {
                const bool top = rng() % 2 == 0;
                c0[c] = top ? 0xFFFFFFFFu - 14u - (uint32_t)(rng() % 30) : 1u + (uint32_t)(rng() % 12);
                r0[c] = rng() % 2 ? 0xFFFFFFFFu - 14u - (uint32_t)(rng() % 30) : 1u + (uint32_t)(rng() % 12);
            }
        for (uint32_t j = 0; j < len; ++j) {
            uint32_t c = (uint32_t)(rng() % centers);
            bool pass2 = j >= len / 2;
            uint32_t ctg = pass2 ? 0 : c0[c] + (uint32_t)(rng() % 15);
            uint32_t ref = (rng() % 5 == 0) ? 0 : r0[c] + (uint32_t)(rng() % 15);
            if (spread) {
                ctg = pass2 ? 0 : 1000 + (uint32_t)(rng() % 40000);
                ref = 5000 + (uint32_t)(rng() % 40000);
            }
            key.push_back(code);
            val.push_back((uint64_t)ctg << 32 | ref);
            uint32_t to = (uint32_t)(rng() % (spread ? 400 : 6)), step = (uint32_t)(rng() % 4);
            eval.push_back((uint64_t)to << 32 | step << 1 | (pass2 ? 1 : 0));
        }
    }
    const uint64_t n = key.size();
    // ---- sequential expectation
    std::vector<uint32_t> xlen(n, 0), xelen(n, 0);
    std::vector<uint64_t> xval(n, 0), xeval(n, 0);
    std::vector<uint16_t> xcnt(n, 0);
    uint64_t x_ctg = 0, x_all = 0, x_seg = 0, x_grp = 0, x_grp1 = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i;
        while (j < n && key[j] == key[i]) ++j;
        std::vector<std::pair<uint64_t, uint16_t>> lead;
        for (uint64_t t = i; t < j; ++t) {
            bool hit = false;
            for (auto &l : lead)
                if (psim(val[t], l.first, eps)) {
                    l.second = (uint16_t)(l.second + 1);
                    hit = true;
                    break;
                }
            if (!hit) lead.push_back({val[t], 1});
        }
        std::sort(lead.begin(), lead.end());
        xlen[i] = (uint32_t)lead.size();
        for (size_t l = 1; l < lead.size(); ++l) xlen[i + l] = pagdev::SEG_LEADER | (uint32_t)l;  // (the other leader slots are marked)
        for (size_t l = 0; l < lead.size(); ++l) {
            xval[i + l] = lead[l].first;
            xcnt[i + l] = lead[l].second;
            x_ctg += (lead[l].first >> 32) != 0;
        }
        x_all += lead.size();
        x_seg += 1;
        std::vector<uint64_t> e(eval.begin() + i, eval.begin() + j);
        std::stable_sort(e.begin(), e.end());
        uint32_t p = 0;
        for (size_t a = 0; a < e.size(); ++a)
            if (a == 0 || (e[a] >> 1) != (e[a - 1] >> 1)) {
                xeval[i + p++] = e[a];
                x_grp1 += (e[a] & 1) == 0;
            }
        xelen[i] = p;
        x_grp += p;
        i = j;
    }
    // ---- device
    uint32_t *d_key, *d_seg, *d_lc;
    uint64_t *d_val, *d_scr, *d_ll, *d_ctr;
    uint16_t *d_cnt;
    CK(hipMalloc(&d_key, n * 4));
    CK(hipMalloc(&d_seg, n * 4));
    CK(hipMalloc(&d_val, n * 8));
    CK(hipMalloc(&d_scr, n * 12 + 64));
    CK(hipMalloc(&d_ll, n * 8));
    CK(hipMalloc(&d_cnt, n * 2));
    CK(hipMalloc(&d_lc, 4));
    CK(hipMalloc(&d_ctr, 32));
    int bad = 0;
    {
        CK(hipMemcpy(d_key, key.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_val, val.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemset(d_seg, 0xEE, n * 4));
        pagdev::ClusterOut co{d_seg, d_cnt, d_ctr};
        if (pagdev::launch_cluster(d_key, d_val, d_scr, n, eps, co, d_ll, d_lc, 0, wide) != 0) return 2;
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> seg(n);
        std::vector<uint64_t> v(n), ctr(4);
        std::vector<uint16_t> c(n);
        CK(hipMemcpy(seg.data(), d_seg, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v.data(), d_val, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), d_cnt, n * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ctr.data(), d_ctr, 32, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n && bad < 10; ++i) {
            if (seg[i] != xlen[i]) {
                printf("cluster seg_len[%llu] = %u want %u\n", (unsigned long long)i, seg[i], xlen[i]);
                ++bad;
                continue;
            }
            for (uint32_t l = 0; !(xlen[i] & pagdev::SEG_LEADER) && l < xlen[i]; ++l)
                if (v[i + l] != xval[i + l] || c[i + l] != xcnt[i + l]) {
                    printf("cluster seg %llu slot %u: (%llx,%u) want (%llx,%u)\n", (unsigned long long)i, l,
                           (unsigned long long)v[i + l], c[i + l], (unsigned long long)xval[i + l], xcnt[i + l]);
                    ++bad;
                    break;
                }
        }
        if (ctr[0] != x_ctg || ctr[1] != x_all || ctr[2] != x_seg) {
            printf("cluster counters %llu %llu %llu want %llu %llu %llu\n", (unsigned long long)ctr[0],
                   (unsigned long long)ctr[1], (unsigned long long)ctr[2], (unsigned long long)x_ctg,
                   (unsigned long long)x_all, (unsigned long long)x_seg);
            ++bad;
        }
    }
    {
        CK(hipMemcpy(d_val, eval.data(), n * 8, hipMemcpyHostToDevice));
        CK(hipMemset(d_seg, 0xEE, n * 4));
        pagdev::EdgeOut eo{d_seg, d_ctr};
        if (pagdev::launch_edges(d_key, d_val, d_scr, n, eo, d_ll, d_lc, 0, wide) != 0) return 2;
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> seg(n);
        std::vector<uint64_t> v(n), ctr(4);
        CK(hipMemcpy(seg.data(), d_seg, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(v.data(), d_val, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ctr.data(), d_ctr, 32, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n && bad < 20; ++i) {
            if (seg[i] != xelen[i]) {
                printf("edges seg_len[%llu] = %u want %u\n", (unsigned long long)i, seg[i], xelen[i]);
                ++bad;
                continue;
            }
            for (uint32_t l = 0; l < xelen[i]; ++l)
                if (v[i + l] != xeval[i + l]) {
                    printf("edges seg %llu slot %u: %llx want %llx\n", (unsigned long long)i, l,
                           (unsigned long long)v[i + l], (unsigned long long)xeval[i + l]);
                    ++bad;
                    break;
                }
        }
        if (ctr[0] != x_grp || ctr[1] != x_grp1) {
            printf("edges counters %llu %llu want %llu %llu\n", (unsigned long long)ctr[0], (unsigned long long)ctr[1],
                   (unsigned long long)x_grp, (unsigned long long)x_grp1);
            ++bad;
        }
    }
    printf("%s: %llu records, %llu segments\n", bad ? "FAIL" : "OK", (unsigned long long)n, (unsigned long long)x_seg);
    return bad ? 1 : 0;
}
