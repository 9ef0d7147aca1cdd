// k0_kmer_count.hip — the solid k-mer set: kmer_counter on the device.
//
// Reference semantics (PAGraph/src/main/kmer_counter.cpp:19-96, KmerHelper.cpp:7-25): every k-mer of the FORWARD
// strand of every read is counted in a dense 4^k table (reads shorter than k contribute nothing, a non-ACGT base
// is an A); the abundances that occur are visited in ascending order, summing how many codes have each, and the
// first abundance a with 1 - sum/4^k <= threshold becomes the minimum abundance (0 if none does); the solid set
// is {code : abundance >= minimum}.  The reference keeps 4 x 4^k size_t counters on the host (8.6 GB at k = 14);
// here the table is 4^k u32 in HBM (1 GB at k = 14, 16 GB at k = 16), filled with one atomic per k-mer.
//
//   kc_count   one wavefront per read: 64 lanes x 16 consecutive k-mer starts per tile, codes from a 64-bit
//              window of the 2-bit packed read (the same bit tricks as the extraction kernel), atomicAdd per code
//   kc_hist    histogram of the abundances 0 .. KC_BINS-2, everything larger in the last bin (the rule almost
//              always stops at a single-digit abundance; the tail is resolved exactly on the host if it does not)
//   kc_select  32 counters -> one bitmap word, population count of the set
#include <algorithm>
#include <cstdlib>
#include <map>
#include <vector>

#include "pag_device.hpp"

namespace pagdev {

constexpr uint32_t KC_BINS = 4096;

__global__ __launch_bounds__(64) void kc_count(const uint64_t *__restrict__ read_off, const uint32_t *__restrict__ read_len,
                                               const uint8_t *__restrict__ packed, uint32_t n_reads, uint32_t k,
                                               uint32_t *__restrict__ table, uint32_t slice, uint32_t slice_shift) {
    const uint32_t lane = lane_id();
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    for (uint32_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint32_t len = read_len[r];
        if (len < k) continue;
        const uint32_t n_pos = len - k + 1;
        const uint32_t *__restrict__ words = (const uint32_t *)(packed + read_off[r]);
        for (uint32_t t0 = 0; t0 < n_pos; t0 += 1024u) {
            const uint32_t p0 = t0 + lane * 16u;
            if (p0 >= n_pos) continue;
            const uint32_t n_mine = n_pos - p0 > 16u ? 16u : n_pos - p0;
            // bases p0 .. p0+30 in one 64-bit window (base i = bits 2i..2i+1 of the packed stream, first base of a
            // k-mer most significant in its code: reverse the 2-bit groups)
            const uint32_t w0 = p0 >> 4;
            const uint64_t W = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
            uint32_t codes[16];
            fwd_codes16(W, k, kmask, codes);
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                if (j < n_mine) {
                    const uint32_t code = codes[j];
                    // (one launch per slice of the code range: the slice's counters stay in the Infinity Cache)
                    if (slice_shift >= 32u || (code >> slice_shift) == slice) atomicAdd(&table[code], 1u);
                }
            }
        }
    }
}

__global__ void kc_hist(const uint32_t *__restrict__ table, uint64_t n, unsigned long long *__restrict__ hist) {
    __shared__ uint32_t h[KC_BINS];
    for (uint32_t i = threadIdx.x; i < KC_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = table[i];
        atomicAdd(&h[a < KC_BINS - 1 ? a : KC_BINS - 1], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < KC_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ void kc_select(const uint32_t *__restrict__ table, uint64_t n_words, uint32_t min_abundance, uint32_t *__restrict__ bitmap,
                          unsigned long long *__restrict__ n_solid) {
    unsigned long long mine = 0;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t bits = 0;
        const uint4 *src = (const uint4 *)(table + w * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 c = src[q];
            bits |= (uint32_t)(c.x >= min_abundance) << (4 * q);
            bits |= (uint32_t)(c.y >= min_abundance) << (4 * q + 1);
            bits |= (uint32_t)(c.z >= min_abundance) << (4 * q + 2);
            bits |= (uint32_t)(c.w >= min_abundance) << (4 * q + 3);
        }
        bitmap[w] = bits;
        mine += (unsigned)__popc(bits);
    }
    uint64_t tot;
    wave_excl_sum64(mine, &tot);
    if (lane_id() == 63 && tot) atomicAdd(n_solid, (unsigned long long)tot);
}

}  // namespace pagdev

using namespace pagdev;

extern "C" int pag_kmer_count(const pag_seqs *reads, int reads_on_device, uint32_t k, double threshold, int device,
                              uint32_t *bitmap, int bitmap_on_device, pag_kmer_count_result *res) {
    if (!reads || !bitmap || k < 1 || k > 16) {
        set_error("pag_kmer_count: bad arguments (k must be 1..16)");
        return PAG_EINVAL;
    }
    PAG_HIP_TRY(hipSetDevice(device));
    const uint64_t n_codes = 1ull << (2 * k);
    const uint64_t n_words = (n_codes + 31) / 32;
    hipStream_t s = nullptr;
    uint32_t *table = nullptr, *d_bitmap = nullptr, *d_len = nullptr;
    uint64_t *d_off = nullptr;
    uint8_t *d_packed = nullptr;
    unsigned long long *d_hist = nullptr;
    std::vector<void *> owned;
    auto cleanup = [&]() {
        for (void *p : owned) hipFree(p);
        if (s) hipStreamDestroy(s);
    };
#define KC_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            cleanup();                                                                            \
            return PAG_EFAULT;                                                                    \
        }                                                                                         \
    } while (0)
    KC_TRY(hipStreamCreate(&s));
    KC_TRY(hipMalloc((void **)&table, std::max<uint64_t>(n_codes, 32) * 4));
    owned.push_back(table);
    KC_TRY(hipMalloc((void **)&d_hist, (KC_BINS + 8) * 8));
    owned.push_back(d_hist);
    KC_TRY(hipMemsetAsync(table, 0, std::max<uint64_t>(n_codes, 32) * 4, s));
    KC_TRY(hipMemsetAsync(d_hist, 0, (KC_BINS + 8) * 8, s));
    const uint64_t n_reads = reads->n_seqs;
    if (reads_on_device) {
        d_off = (uint64_t *)reads->byte_off;
        d_len = (uint32_t *)reads->len;
        d_packed = (uint8_t *)reads->packed;  // the caller guarantees 8 readable bytes past every read (pag_seqs contract)
    } else if (n_reads) {
        KC_TRY(hipMalloc((void **)&d_off, n_reads * 8));
        owned.push_back(d_off);
        KC_TRY(hipMalloc((void **)&d_len, n_reads * 4));
        owned.push_back(d_len);
        KC_TRY(hipMalloc((void **)&d_packed, reads->packed_bytes + 64));
        owned.push_back(d_packed);
        KC_TRY(hipMemcpyAsync(d_off, reads->byte_off, n_reads * 8, hipMemcpyHostToDevice, s));
        KC_TRY(hipMemcpyAsync(d_len, reads->len, n_reads * 4, hipMemcpyHostToDevice, s));
        KC_TRY(hipMemsetAsync(d_packed + reads->packed_bytes, 0, 64, s));
        KC_TRY(hipMemcpyAsync(d_packed, reads->packed, reads->packed_bytes, hipMemcpyHostToDevice, s));
    }
    hipEvent_t ev[3];
    for (auto &e : ev) KC_TRY(hipEventCreate(&e));
    KC_TRY(hipEventRecord(ev[0], s));
    if (n_reads) {
        const unsigned grid = (unsigned)std::min<uint64_t>(n_reads, 1u << 20);
        // The table is filled one slice of the code range per launch once it is larger than the Infinity Cache can hold beside
        // the reads (k >= 13): with a quarter of the counters live at a time the atomics stay on chip — 52 -> 39 ms at
        // BASELINE configs[1], k = 14, the same with 8 or 16 slices (tests/kc_probe.sh).  Tests: PAG_KC_SLICES=<2^n> runs the sliced form at small k.
        const uint32_t slices = (uint32_t)std::max<long long>(1, env_int("PAG_KC_SLICES", n_codes * 4 > (128ull << 20) ? 4 : 1));
        uint32_t lg = 0;
        while ((1u << (lg + 1)) <= slices) ++lg;
        if (2 * k <= lg) lg = 0;
        const uint32_t shift = lg ? 2 * k - lg : 32u;
        for (uint32_t sl = 0; sl < (1u << lg); ++sl)
            kc_count<<<dim3(grid), dim3(64), 0, s>>>(d_off, d_len, d_packed, (uint32_t)n_reads, k, table, sl, shift);
    }
    kc_hist<<<dim3(4096), dim3(256), 0, s>>>(table, n_codes, d_hist);
    KC_TRY(hipEventRecord(ev[1], s));
    std::vector<unsigned long long> hist(KC_BINS);
    KC_TRY(hipMemcpyAsync(hist.data(), d_hist, KC_BINS * 8, hipMemcpyDeviceToHost, s));
    KC_TRY(hipStreamSynchronize(s));

    // the reference's rule (kmer_counter.cpp:59-77), same expression in double
    uint64_t min_abundance = 0;
    {
        uint64_t sum = 0;
        bool found = false;
        for (uint32_t a = 0; a + 1 < KC_BINS && !found; ++a) {
            if (!hist[a]) continue;
            sum += hist[a];
            if (1 - sum * 1.0 / (double)n_codes <= threshold) {
                min_abundance = a;
                found = true;
            }
        }
        if (!found && hist[KC_BINS - 1]) {  // the tail: exact abundances from the table itself
            std::vector<uint32_t> t(n_codes);
            KC_TRY(hipMemcpy(t.data(), table, n_codes * 4, hipMemcpyDeviceToHost));
            std::map<uint64_t, uint64_t> tail;
            for (uint64_t i = 0; i < n_codes; ++i)
                if (t[i] >= KC_BINS - 1) ++tail[t[i]];
            for (auto &kv : tail) {
                sum += kv.second;
                if (1 - sum * 1.0 / (double)n_codes <= threshold) {
                    min_abundance = kv.first;
                    break;
                }
            }
        }
    }
    if (bitmap_on_device) {
        d_bitmap = bitmap;
    } else {
        KC_TRY(hipMalloc((void **)&d_bitmap, n_words * 4));
        owned.push_back(d_bitmap);
    }
    unsigned long long *d_ns = d_hist + KC_BINS + 1;
    if (n_codes >= 32) {
        kc_select<<<dim3((unsigned)std::min<uint64_t>((n_words + 255) / 256, 65536)), dim3(256), 0, s>>>(
            table, n_words, (uint32_t)std::min<uint64_t>(min_abundance, 0xFFFFFFFFull), d_bitmap, d_ns);
    } else {  // k <= 2: fewer than 32 codes, one word assembled on the host
        std::vector<uint32_t> t(32, 0);
        KC_TRY(hipMemcpy(t.data(), table, n_codes * 4, hipMemcpyDeviceToHost));
        uint32_t bits = 0;
        unsigned long long ns = 0;
        for (uint64_t i = 0; i < n_codes; ++i)
            if (t[i] >= min_abundance) {
                bits |= 1u << i;
                ++ns;
            }
        KC_TRY(hipMemcpy(d_bitmap, &bits, 4, hipMemcpyHostToDevice));
        KC_TRY(hipMemcpy(d_ns, &ns, 8, hipMemcpyHostToDevice));
    }
    KC_TRY(hipEventRecord(ev[2], s));
    unsigned long long ns = 0;
    KC_TRY(hipMemcpyAsync(&ns, d_ns, 8, hipMemcpyDeviceToHost, s));
    if (!bitmap_on_device) KC_TRY(hipMemcpyAsync(bitmap, d_bitmap, n_words * 4, hipMemcpyDeviceToHost, s));
    KC_TRY(hipStreamSynchronize(s));
    KC_TRY(hipGetLastError());
    if (res) {
        float ms = 0;
        res->min_abundance = min_abundance;
        res->n_solid = ns;
        uint64_t nk = 0;
        for (uint32_t a = 0; a < KC_BINS; ++a) nk += (uint64_t)hist[a] * (a < KC_BINS - 1 ? a : 0);
        res->n_kmers_counted = hist[KC_BINS - 1] ? 0 : nk;  // exact only when no abundance reached the last bin
        hipEventElapsedTime(&ms, ev[0], ev[1]);
        res->ms_count = ms;
        hipEventElapsedTime(&ms, ev[1], ev[2]);
        res->ms_select = ms;
    }
    for (auto &e : ev) hipEventDestroy(e);
    cleanup();
    return PAG_OK;
#undef KC_TRY
}
