// k2_sort.hip — K2: stable radix sort of (u32 k-mer code, u64 payload) records.
//
// The reference has no sort here: its per-k-mer vectors ARE the bucketed form (node/KMerAdjNode.hpp
// :19-23) and stability is the insertion order into those vectors.  On the device the records are
// grouped by k-mer with a least-significant-digit radix sort, which is stable by construction, so the
// emission order inside every k-mer bucket survives.
//
// One digit pass = three launches:
//   sort_hist     per 4096-record tile: digit histogram in LDS -> hist[digit][tile]
//   scan          exclusive prefix over hist (digit-major), giving every (digit, tile) its output base
//   sort_scatter  per tile: wave-synchronous stable ranking (ballot match on the digit bits, per-wave
//                 LDS counters), records staged through LDS in digit order, then written out in runs
//                 so that the HBM stores are coalesced.
// HBM traffic per pass: keys 4 B (hist) + 12 B read + 12 B write per record.
#include "pag_device.hpp"

namespace pagdev {

constexpr int ST = 1024;          // threads per block
constexpr int SW = ST / 64;       // waves per block
constexpr int SROUNDS = 5;       // records per thread
constexpr int STILE = ST * SROUNDS;
constexpr int SMAXR = 256;        // max radix (8 bits)

__global__ __launch_bounds__(ST) void sort_hist(const uint32_t *__restrict__ keys, uint64_t n, int shift, uint32_t rmask,
                                               uint32_t *__restrict__ hist, uint32_t n_tiles) {
    __shared__ uint32_t h[SMAXR];
    if (threadIdx.x < SMAXR) h[threadIdx.x] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * STILE;
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        uint64_t i = base + (uint64_t)r * ST + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & rmask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= rmask) hist[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

struct ScatterLds {
    uint32_t wcnt[SW][SMAXR];   // per-wave running digit counters, then their exclusive prefix over waves
    uint32_t dstart[SMAXR];     // exclusive prefix of the tile's digit totals over digits
    uint64_t gbase[SMAXR];      // output base of (digit, this tile)
    uint32_t skey[STILE];
    uint64_t sval[STILE];
    uint64_t red[8];
};

__global__ __launch_bounds__(ST) void sort_scatter(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                  uint32_t *__restrict__ okeys, uint64_t *__restrict__ ovals, uint64_t n,
                                                  int shift, int bits, const uint64_t *__restrict__ hist_scan,
                                                  uint32_t n_tiles) {
    __shared__ ScatterLds L;
    const uint32_t rmask = (1u << bits) - 1u;
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < SW * SMAXR; i += ST) (&L.wcnt[0][0])[i] = 0;
    __syncthreads();

    const uint64_t tile_base = (uint64_t)blockIdx.x * STILE;
    const uint64_t wave_base = tile_base + (uint64_t)w * (64 * SROUNDS);
    uint32_t key[SROUNDS];
    uint32_t rk[SROUNDS];  // rank of the record among same-digit records of this wave
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        uint64_t i = wave_base + (uint64_t)r * 64 + lane;
        bool valid = i < n;
        key[r] = valid ? keys[i] : 0u;
        uint32_t d = (key[r] >> shift) & rmask;
        // peers = lanes of this wave holding the same digit in this round
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            uint64_t vote = __ballot(valid && ((d >> b) & 1u));
            peers &= ((d >> b) & 1u) ? vote : ~vote;
        }
        uint32_t before = __popcll(peers & lanemask_lt());
        // per-wave counters: the read (all lanes) and the leader's write are LDS operations of ONE wave,
        // which the LDS executes in program order — no barrier needed between rounds
        uint32_t base = valid ? L.wcnt[w][d] : 0u;
        if (valid && before == 0) L.wcnt[w][d] = base + (uint32_t)__popcll(peers);
        rk[r] = base + before;
    }

    __syncthreads();
    // per digit: prefix over waves, tile totals, prefix over digits, global base
    if (threadIdx.x < SMAXR) {
        uint32_t d = threadIdx.x;
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < SW; ++ww) {
            uint32_t c = L.wcnt[ww][d];
            L.wcnt[ww][d] = run;
            run += c;
        }
        uint64_t tot;
        // block exclusive scan of `run` over the 256 digits
        uint64_t wtot;
        uint64_t ex = wave_excl_sum64(run, &wtot);
        if (lane == 63) L.red[w] = wtot;
        L.dstart[d] = (uint32_t)ex;  // completed with the preceding waves' totals below
        L.gbase[d] = d <= rmask ? hist_scan[(uint64_t)d * n_tiles + blockIdx.x] : 0;
        (void)tot;
    }
    __syncthreads();
    if (threadIdx.x < SMAXR) {
        uint32_t pre = 0;
        for (int i = 0; i < (int)w; ++i) pre += (uint32_t)L.red[i];  // waves 0..3 hold the 256 digits
        L.dstart[threadIdx.x] += pre;
    }
    __syncthreads();

    // stage the tile in digit order
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
        uint64_t i = wave_base + (uint64_t)r * 64 + lane;
        if (i < n) {
            uint32_t d = (key[r] >> shift) & rmask;
            uint32_t p = L.dstart[d] + L.wcnt[w][d] + rk[r];
            L.skey[p] = key[r];
            L.sval[p] = vals[i];
        }
    }
    __syncthreads();
    const uint32_t count = (uint32_t)((n - tile_base) < (uint64_t)STILE ? (n - tile_base) : (uint64_t)STILE);
    for (uint32_t p = threadIdx.x; p < count; p += ST) {
        uint32_t kx = L.skey[p];
        uint32_t d = (kx >> shift) & rmask;
        uint64_t dst = L.gbase[d] + (p - L.dstart[d]);
        okeys[dst] = kx;
        ovals[dst] = L.sval[p];
    }
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// tmp layout: hist u32[R * n_tiles] | hist_scan u64[R * n_tiles] | scan tmp
size_t sort_tmp_bytes(uint64_t n) {
    uint64_t n_tiles = (n + STILE - 1) / STILE;
    if (n_tiles == 0) n_tiles = 1;
    uint64_t cells = (uint64_t)SMAXR * n_tiles;
    return align256(cells * 4) + align256(cells * 8) + align256(scan_tmp_bytes(cells)) + 256;
}

int sort_pairs(uint32_t *k0, uint64_t *v0, uint32_t *k1, uint64_t *v1, uint64_t n, int key_bits, void *tmp,
               int *result_in_0, hipStream_t s, float *ms_dominant_kernel, int *n_passes) {
    *result_in_0 = 1;
    if (ms_dominant_kernel) *ms_dominant_kernel = 0.f;
    if (n_passes) *n_passes = 0;
    if (n == 0 || key_bits <= 0) return PAG_OK;
    uint64_t n_tiles64 = (n + STILE - 1) / STILE;
    if (n_tiles64 > 0x7FFFFFFFull) {
        set_error("sort: too many tiles");
        return PAG_EINVAL;
    }
    uint32_t n_tiles = (uint32_t)n_tiles64;
    int passes = (key_bits + 7) / 8;
    int bits = (key_bits + passes - 1) / passes;
    char *p = (char *)tmp;
    uint32_t *hist = (uint32_t *)p;
    uint64_t cells = (uint64_t)SMAXR * n_tiles;
    p += align256(cells * 4);
    uint64_t *hist_scan = (uint64_t *)p;
    p += align256(cells * 8);
    void *scan_tmp = p;

    hipEvent_t ev[2 * 8];
    for (int i = 0; i < 2 * passes; ++i) PAG_HIP_TRY(hipEventCreate(&ev[i]));
    uint32_t *ka = k0, *kb = k1;
    uint64_t *va = v0, *vb = v1;
    int in0 = 1;
    for (int pass = 0; pass < passes; ++pass) {
        int shift = pass * bits;
        int b = key_bits - shift < bits ? key_bits - shift : bits;
        uint32_t rmask = (1u << b) - 1u;
        uint64_t used = (uint64_t)(rmask + 1) * n_tiles;
        sort_hist<<<dim3(n_tiles), dim3(ST), 0, s>>>(ka, n, shift, rmask, hist, n_tiles);
        int rc = scan_u32_to_u64(hist, hist_scan, used, nullptr, scan_tmp, s);
        if (rc != PAG_OK) return rc;
        PAG_HIP_TRY(hipEventRecord(ev[2 * pass], s));
        sort_scatter<<<dim3(n_tiles), dim3(ST), 0, s>>>(ka, va, kb, vb, n, shift, b, hist_scan, n_tiles);
        PAG_HIP_TRY(hipEventRecord(ev[2 * pass + 1], s));
        uint32_t *tk = ka;
        ka = kb;
        kb = tk;
        uint64_t *tv = va;
        va = vb;
        vb = tv;
        in0 = !in0;
    }
    PAG_HIP_TRY(hipGetLastError());
    PAG_HIP_TRY(hipStreamSynchronize(s));
    float tot = 0.f;
    for (int pass = 0; pass < passes; ++pass) {
        float ms = 0.f;
        PAG_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * pass], ev[2 * pass + 1]));
        tot += ms;
    }
    for (int i = 0; i < 2 * passes; ++i) hipEventDestroy(ev[i]);
    if (ms_dominant_kernel) *ms_dominant_kernel = tot / passes;
    if (n_passes) *n_passes = passes;
    *result_in_0 = in0;
    return PAG_OK;
}

}  // namespace pagdev
