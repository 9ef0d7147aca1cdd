// Development aid (not a test, not shipped): times the radix-sort passes of k2_sort.hip on random records and checks
// the result against std::stable_sort on a sample size.  make tests/harness/bin/sort_bench; run on a GPU box:
//   sort_bench [n_records] [key_bits]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../aligngraph2_amd/csrc/hip/util.hip"
#include "../../aligngraph2_amd/csrc/hip/k2_sort.hip"

using namespace pagdev;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));            \
            return 2;                                                              \
        }                                                                          \
    } while (0)

__global__ void fill_random(uint32_t *k, uint64_t *v, uint64_t n, uint32_t mask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        k[i] = (uint32_t)x & mask;
        v[i] = i;
    }
}

// sort_bench scan <n> <offset>: scan_u32_to_u64 of n random values read from an array view that starts `offset` elements
// into its allocation (misaligned for the 16-byte path when offset % 4 != 0), against a sequential sum; timed
static int scan_mode(uint64_t n, uint64_t offset) {
    uint32_t *in;
    uint64_t *out, *total;
    void *tmp;
    CK(hipMalloc(&in, (n + offset + 8) * 4));
    CK(hipMalloc(&out, (n + offset + 8) * 8));
    CK(hipMalloc(&total, 8));
    CK(hipMalloc(&tmp, scan_tmp_bytes(n + 1) + 64));
    std::vector<uint32_t> h(n);
    std::mt19937_64 rng(n * 31 + offset);
    for (auto &x : h) x = (rng() % 16 == 0) ? (uint32_t)rng() : (uint32_t)(rng() % 7);  // (large values too: the sums pass 2^32)
    CK(hipMemcpy(in + offset, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0xEE, (n + offset + 8) * 8));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        CK(hipEventRecord(a, 0));
        if (scan_u32_to_u64(in + offset, out + offset, n, total, tmp, 0) != PAG_OK) return 2;
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, ms);
    }
    std::vector<uint64_t> got(n + 1);
    uint64_t tot = 0;
    CK(hipMemcpy(got.data(), out + offset, (n + 1) * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&tot, total, 8, hipMemcpyDeviceToHost));
    uint64_t acc = 0, bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        if (got[i] != acc) ++bad;
        acc += h[i];
    }
    if (tot != acc) ++bad;
    if (got[n] != 0xEEEEEEEEEEEEEEEEull) ++bad;  // (nothing written past the end)
    std::printf("scan n=%llu offset=%llu: %.3f ms = %.0f GB/s (12 B per element), check: %llu mismatches\n", (unsigned long long)n,
                (unsigned long long)offset, best, n * 12.0 / best / 1e6, (unsigned long long)bad);
    return bad ? 1 : 0;
}

int main(int argc, char **argv) {
    if (argc > 2 && std::string(argv[1]) == "scan")
        return scan_mode(std::strtoull(argv[2], nullptr, 10), argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0);
    const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 448810094ull;
    const int bits = argc > 2 ? std::atoi(argv[2]) : 28;
    const bool check = n <= (1ull << 24);
    uint32_t *k0, *k1;
    uint64_t *v0, *v1;
    void *tmp;
    CK(hipMalloc(&k0, n * 4));
    CK(hipMalloc(&k1, n * 4));
    CK(hipMalloc(&v0, n * 8));
    CK(hipMalloc(&v1, n * 8));
    CK(hipMalloc(&tmp, sort_tmp_bytes(n)));
    const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    for (int rep = 0; rep < 3; ++rep) {
        fill_random<<<4096, 256>>>(k0, v0, n, mask);
        CK(hipDeviceSynchronize());
        int in0 = 0, passes = 0;
        float ms = 0;
        auto t0 = std::chrono::steady_clock::now();
        if (sort_pairs(k0, v0, k1, v1, n, bits, tmp, &in0, 0, &ms, &passes) != PAG_OK) {
            std::fprintf(stderr, "sort failed: %s\n", last_error());
            return 2;
        }
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("n=%llu bits=%d passes=%d scatter %.3f ms/pass = %.0f GB/s (%.1f %% of 8 TB/s), whole sort %.2f ms\n",
                    (unsigned long long)n, bits, passes, ms, 24.0 * n / ms * 1e-6, 24.0 * n / ms * 1e-6 / 80.0, wall);
        if (check && rep == 0) {
            std::vector<uint32_t> hk(n), rk(n);
            std::vector<uint64_t> hv(n), rv(n);
            CK(hipMemcpy(rk.data(), in0 ? k0 : k1, n * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(rv.data(), in0 ? v0 : v1, n * 8, hipMemcpyDeviceToHost));
            for (uint64_t i = 0; i < n; ++i) {
                uint64_t x = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
                x ^= x >> 29;
                x *= 0xBF58476D1CE4E5B9ull;
                x ^= x >> 32;
                hk[i] = (uint32_t)x & mask;
                hv[i] = i;
            }
            std::vector<uint64_t> idx(n);
            for (uint64_t i = 0; i < n; ++i) idx[i] = i;
            std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) { return hk[a] < hk[b]; });
            uint64_t bad = 0;
            for (uint64_t i = 0; i < n; ++i) {
                const bool b = rk[i] != hk[idx[i]] || rv[i] != hv[idx[i]];
                if (b && bad < 8)
                    std::printf("  [%llu] got key %u val %llu, expected key %u val %llu\n", (unsigned long long)i, rk[i],
                                (unsigned long long)rv[i], hk[idx[i]], (unsigned long long)hv[idx[i]]);
                bad += b ? 1 : 0;
            }
            std::printf("check: %llu mismatches\n", (unsigned long long)bad);
            if (bad) return 1;
        }
    }
    return 0;
}
