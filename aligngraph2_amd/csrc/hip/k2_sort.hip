// k2_sort.hip — K2: stable radix sort of (u32 k-mer code, u64 payload) records.
//
// The reference has no sort here: its per-k-mer vectors ARE the bucketed form (node/KMerAdjNode.hpp
// :19-23) and stability is the insertion order into those vectors.  On the device the records are
// grouped by k-mer with a least-significant-digit radix sort, which is stable by construction, so the
// emission order inside every k-mer bucket survives.
//
// One digit pass = three launches:
//   sort_hist     per tile: digit histogram in LDS -> hist[digit][tile]
//   scan          exclusive prefix over hist (digit-major), giving every (digit, tile) its output base
//   sort_scatter  per tile: wave-synchronous stable ranking (ballot match on the digit bits, per-wave
//                 LDS counters), records staged through LDS in digit order, then written out in runs
//                 so that the HBM stores are coalesced.
// HBM traffic per pass: keys 4 B (hist) + 12 B read + 12 B write per record.
#include <stdlib.h>

#include "pag_device.hpp"

namespace pagdev {

#ifndef SORT_ST
#define SORT_ST 1024
#endif
#ifndef SORT_ROUNDS
#define SORT_ROUNDS 5
#endif
constexpr int ST = SORT_ST;       // threads per block
constexpr int SW = ST / 64;       // waves per block
constexpr int SROUNDS = SORT_ROUNDS;  // records per thread
constexpr int STILE = ST * SROUNDS;
constexpr int SMAXR = 256;        // max radix (8 bits)

// Digit histogram of every tile: the keys are read once, 16 bytes per lane, by persistent blocks of four waves; every wave
// counts into its own LDS table (a quarter of the collisions), the tables are summed when the tile's column of
// hist[digit][tile] is written.  The loads of the next tile are in flight while a tile is counted.
constexpr int HT = 256;                    // threads per histogram block
constexpr int HW = HT / 64;                // waves
constexpr int HV = STILE / (4 * HT);       // uint4 loads per thread and tile
static_assert(STILE % (4 * HT) == 0, "a tile is a whole number of 16-byte loads per thread");

__global__ __launch_bounds__(HT) void sort_hist(const uint32_t *__restrict__ keys, uint64_t n, int shift, uint32_t rmask,
                                                uint32_t *__restrict__ hist, uint32_t n_tiles, int aligned) {
    __shared__ uint32_t h[2][HW][SMAXR];
    const uint32_t w = threadIdx.x >> 6;
    uint4 nx[HV];
    auto request = [&](uint32_t tile) {
        const uint64_t base = (uint64_t)tile * STILE;
        if (aligned && tile < n_tiles && base + STILE <= n) {
            const uint4 *src = (const uint4 *)(keys + base);
#pragma unroll
            for (int r = 0; r < HV; ++r) nx[r] = src[r * HT + threadIdx.x];
        }
    };
    for (uint32_t i = threadIdx.x; i < 2 * HW * SMAXR; i += HT) (&h[0][0][0])[i] = 0;
    request(blockIdx.x);
    __syncthreads();
    uint32_t buf = 0;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1u) {
        const uint64_t base = (uint64_t)tile * STILE;
        uint32_t *hw = h[buf][w];
        if (aligned && base + STILE <= n) {
            uint4 cur[HV];
#pragma unroll
            for (int r = 0; r < HV; ++r) cur[r] = nx[r];
            request(tile + gridDim.x);
#pragma unroll
            for (int r = 0; r < HV; ++r) {
                atomicAdd(&hw[(cur[r].x >> shift) & rmask], 1u);
                atomicAdd(&hw[(cur[r].y >> shift) & rmask], 1u);
                atomicAdd(&hw[(cur[r].z >> shift) & rmask], 1u);
                atomicAdd(&hw[(cur[r].w >> shift) & rmask], 1u);
            }
        } else {  // the last tile, or a key array that is not 16-byte aligned
            request(tile + gridDim.x);
            for (uint32_t j = threadIdx.x; j < (uint32_t)STILE; j += HT)
                if (base + j < n) atomicAdd(&hw[(keys[base + j] >> shift) & rmask], 1u);
        }
        __syncthreads();
        // the column of this tile; the table is cleared for the tile after next (the next one counts into the other table)
        for (uint32_t d = threadIdx.x; d <= rmask; d += HT) {
            uint32_t c = 0;
#pragma unroll
            for (int ww = 0; ww < HW; ++ww) {
                c += h[buf][ww][d];
                h[buf][ww][d] = 0;
            }
            hist[(uint64_t)d * n_tiles + tile] = c;
        }
    }
}

struct ScatterStage {       // one tile staged in digit order, waiting to be written out
    uint64_t goff[SMAXR];   // output base of (digit, tile) minus the digit's first staged position: destination =
                            // goff[d] + staged position (its first half holds the tile's digit totals until the
                            // digit prefix is done)
    uint32_t skey[STILE];
#ifdef SORT_TWO_PLANE  // (measurement variant, tests/harness/sort_bench: the payload staged as two u32 planes instead of one u64 array)
    uint32_t sval_lo[STILE], sval_hi[STILE];
#else
    uint64_t sval[STILE];
#endif
};
struct ScatterLds {
    uint32_t wcnt[SW][SMAXR + 1];  // per-wave running digit counters, then their exclusive prefix over waves (+1: bank spread)
    uint32_t dstart[SMAXR];        // exclusive prefix of the tile's digit totals over digits
    ScatterStage stage[2];         // tile i is staged into stage[i & 1] while tile i - 1 is written out of the other
};
static_assert(SW == 16 || SW == 8, "the wave prefix below scans groups of SW lanes inside a DPP row");
static_assert(sizeof(ScatterLds) <= 160 * 1024, "LDS");
static_assert(STILE >= SW * SMAXR, "match words alias the payload staging buffer");

// inclusive prefix sum inside each group of SW (8 or 16) consecutive lanes (DPP row shifts, zero fill at the row start)
__device__ __forceinline__ uint32_t group_incl_sum(uint32_t v, uint32_t pos) {
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (SW == 16 || pos >= 1u) ? t : 0u;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (SW == 16 || pos >= 2u) ? t : 0u;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (SW == 16 || pos >= 4u) ? t : 0u;
    if (SW == 16) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    return v;
}

// Persistent blocks (one per CU: 140 KB of LDS), each looping over tiles as a three-stage pipeline:
//   iteration i:  request the keys / payloads / output bases of tile i + 1 into registers,
//                 write tile i - 1 out of its LDS staging buffer (stores only),
//                 rank tile i, prefix its digit counts, stage it into the other LDS buffer.
// So the loads of the next tile and the stores of the previous one are in flight while a tile is ranked and staged,
// and the wait for the requested data at the top of an iteration finds loads AND stores issued most of an iteration
// ago (gfx950 has one counter for both: waiting for loads right after issuing stores would expose the store latency
// once per tile).  Tile order: the blocks resident on one XCD (block id mod 8 = XCD) work on CONSECUTIVE tiles at a
// time, so the runs they append to a digit's output region are adjacent and the partial cache lines at run
// boundaries are combined in that XCD's L2 instead of being written back by two L2s.
__device__ __forceinline__ uint64_t scatter_tile(uint32_t block, uint32_t n_blocks, uint32_t it) {
    // a grid that does not fill the XCDs evenly falls back to a plain stride
    if (n_blocks % 8u != 0u) return (uint64_t)it * n_blocks + block;
    const uint32_t per_xcd = n_blocks / 8u, x = block & 7u, slot = block >> 3;
    return ((uint64_t)it * 8u + x) * per_xcd + slot;
}

#ifdef SORT_WAVES_PER_EU
#define SORT_BOUNDS __launch_bounds__(ST, SORT_WAVES_PER_EU)
#else
#define SORT_BOUNDS __launch_bounds__(ST)
#endif
template <int BITS>
__global__ SORT_BOUNDS void sort_scatter(const uint32_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                  uint32_t *__restrict__ okeys, uint64_t *__restrict__ ovals, uint64_t n,
                                                  int shift, const uint64_t *__restrict__ hist_scan, uint32_t n_tiles) {
    __shared__ ScatterLds L;
    constexpr uint32_t rmask = (1u << BITS) - 1u;
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t n_iter = (n_tiles + gridDim.x - 1) / gridDim.x;  // (both tile orders enumerate [it * grid, (it + 1) * grid) in iteration it)

    uint32_t nkey[SROUNDS];
    uint64_t nval[SROUNDS];
    uint64_t ngbase = 0;  // threads 0 .. 2^BITS - 1: output base of (digit = thread, tile)
    auto request = [&](uint64_t tile) {
        const uint64_t wb = tile * STILE + (uint64_t)w * (64 * SROUNDS);
        if (threadIdx.x <= rmask && tile < n_tiles) ngbase = hist_scan[(uint64_t)threadIdx.x * n_tiles + tile];
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            const uint64_t i = wb + (uint64_t)r * 64 + lane;
            const bool valid = tile < n_tiles && i < n;
            nkey[r] = valid ? keys[i] : 0u;
            nval[r] = valid ? vals[i] : 0ull;
        }
    };
    request(scatter_tile(blockIdx.x, gridDim.x, 0));

    const uint32_t last_count = (uint32_t)(n - (uint64_t)(n_tiles - 1u) * STILE);  // records of the last tile (1 .. STILE)
    auto write_out = [&](const ScatterStage &B, uint32_t count) {
        if (count == (uint32_t)STILE) {
#pragma unroll
            for (int r = 0; r < SROUNDS; ++r) {
                const uint32_t p = (uint32_t)r * ST + threadIdx.x;
                const uint32_t kx = B.skey[p];
                const uint64_t dst = B.goff[(kx >> shift) & rmask] + p;
                okeys[dst] = kx;
#ifdef SORT_TWO_PLANE
                ovals[dst] = (uint64_t)B.sval_lo[p] | ((uint64_t)B.sval_hi[p] << 32);
#else
                ovals[dst] = B.sval[p];
#endif
            }
        } else {
            for (uint32_t p = threadIdx.x; p < count; p += ST) {
                const uint32_t kx = B.skey[p];
                const uint64_t dst = B.goff[(kx >> shift) & rmask] + p;
                okeys[dst] = kx;
#ifdef SORT_TWO_PLANE
                ovals[dst] = (uint64_t)B.sval_lo[p] | ((uint64_t)B.sval_hi[p] << 32);
#else
                ovals[dst] = B.sval[p];
#endif
            }
        }
    };

    bool pending = false;  // a staged tile waits to be written out
    uint32_t pending_count = 0;
    uint32_t buf = 0;
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint64_t tile = scatter_tile(blockIdx.x, gridDim.x, it);
        uint32_t key[SROUNDS];
        uint64_t val[SROUNDS];
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            key[r] = nkey[r];
            val[r] = nval[r];
        }
        const uint64_t gbase = ngbase;
        if (it + 1 < n_iter) request(scatter_tile(blockIdx.x, gridDim.x, it + 1));
        if (pending) write_out(L.stage[buf ^ 1u], pending_count);
        pending = false;
        if (tile >= n_tiles) continue;  // (uniform per block)

        ScatterStage &B = L.stage[buf];
        const uint64_t tile_base = tile * STILE;
        const uint64_t wave_base = tile_base + (uint64_t)w * (64 * SROUNDS);
        // every wave clears its own counters and match words: LDS operations of one wave execute in program order, no
        // barrier needed
#ifdef SORT_TWO_PLANE
        unsigned long long *match = (unsigned long long *)&B.sval_lo[0] + (size_t)w * (rmask + 1u);
#else
        unsigned long long *match = (unsigned long long *)&B.sval[0] + (size_t)w * (rmask + 1u);  // (this staging buffer is free now)
#endif
#pragma unroll
        for (int i = 0; i < SMAXR / 64; ++i) L.wcnt[w][i * 64 + lane] = 0;
#pragma unroll
        for (uint32_t i = lane; i <= rmask; i += 64) match[i] = 0ull;
        uint32_t rk[SROUNDS];  // rank of the record among same-digit records of this wave
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            const uint64_t i = wave_base + (uint64_t)r * 64 + lane;
            const bool valid = i < n;
            const uint32_t d = (key[r] >> shift) & rmask;
            // peers = lanes of this wave holding the same digit in this round: every lane ORs its bit into the digit's
            // match word (one LDS atomic), reads the word back and clears it for the next round — three LDS
            // instructions instead of BITS ballots with per-lane 64-bit selects
            if (valid) atomicOr(&match[d], 1ull << lane);
            const uint64_t peers = valid ? match[d] : 0ull;
            if (valid) match[d] = 0ull;
            const uint32_t before = __popcll(peers & lanemask_lt());
            const uint32_t base = valid ? L.wcnt[w][d] : 0u;
            if (valid && before == 0) L.wcnt[w][d] = base + (uint32_t)__popcll(peers);
            rk[r] = base + before;
        }

        __syncthreads();
        // per digit: exclusive prefix over the waves (a group of SW lanes per digit), tile totals
        uint32_t *dtot = (uint32_t *)&B.goff[0];
#pragma unroll
        for (int j = 0; j < SMAXR * SW / ST; ++j) {
            const uint32_t d = (uint32_t)j * (ST / SW) + threadIdx.x / SW, ww = threadIdx.x % SW;
            const uint32_t c = L.wcnt[ww][d];
            const uint32_t incl = group_incl_sum(c, ww);
            L.wcnt[ww][d] = incl - c;
            if (ww == SW - 1u) dtot[d] = incl;
        }
        __syncthreads();
        // prefix over digits by one wave (four digits per lane)
        if (w == 0) {
            const uint32_t t0 = dtot[4 * lane], t1 = dtot[4 * lane + 1], t2 = dtot[4 * lane + 2], t3 = dtot[4 * lane + 3];
            uint32_t tot;
            const uint32_t ex = wave_excl_sum(t0 + t1 + t2 + t3, &tot);
            L.dstart[4 * lane] = ex;
            L.dstart[4 * lane + 1] = ex + t0;
            L.dstart[4 * lane + 2] = ex + t0 + t1;
            L.dstart[4 * lane + 3] = ex + t0 + t1 + t2;
        }
        __syncthreads();
#ifdef SORT_LOOKBACK_PROBE
        // Measurement variant (tests/harness/sort_bench only): what a decoupled look-back would add to a tile AT THE LEAST — the
        // tile's digit counts published (one word per digit) and the rows of SORT_LOOKBACK_PROBE predecessor tiles read back
        // with device-scope loads by the digit's thread, here, where the tile's output bases are needed.  No waiting for
        // flags, no retries: a lower bound of the real thing.  The words live in the (unused, digit-major) tail of the
        // histogram scan array; their values do not matter.
        if (threadIdx.x <= rmask) {
            uint32_t *rows = (uint32_t *)(hist_scan + (uint64_t)(rmask + 1u) * n_tiles);
            __hip_atomic_store(&rows[tile * (rmask + 1u) + threadIdx.x], dtot[threadIdx.x] | 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t acc = 0;
            for (uint32_t b = 1; b <= (uint32_t)SORT_LOOKBACK_PROBE; ++b) {
                const uint64_t t2 = tile >= b ? tile - b : 0;
                acc += __hip_atomic_load(&rows[t2 * (rmask + 1u) + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (acc == 0xFFFFFFFFu) L.dstart[threadIdx.x] += 1u;  // (never: keeps the loads alive)
        }
        __syncthreads();
#endif
        if (threadIdx.x <= rmask) B.goff[threadIdx.x] = gbase - L.dstart[threadIdx.x];

        // stage the tile in digit order
#pragma unroll
        for (int r = 0; r < SROUNDS; ++r) {
            const uint64_t i = wave_base + (uint64_t)r * 64 + lane;
            if (i < n) {
                const uint32_t d = (key[r] >> shift) & rmask;
                const uint32_t p = L.dstart[d] + L.wcnt[w][d] + rk[r];
                B.skey[p] = key[r];
#ifdef SORT_TWO_PLANE
                B.sval_lo[p] = (uint32_t)val[r];
                B.sval_hi[p] = (uint32_t)(val[r] >> 32);
#else
                B.sval[p] = val[r];
#endif
            }
        }
        __syncthreads();  // staged: the tile is written out in the next iteration (or after the loop)
        pending = true;
        pending_count = tile == (uint64_t)n_tiles - 1u ? last_count : (uint32_t)STILE;
        buf ^= 1u;
    }
    if (pending) write_out(L.stage[buf ^ 1u], pending_count);
}

static int scatter_blocks_cus(int dev) {
    static int cus_of[64] = {0};
    int &c = cus_of[dev & 63];
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 256;
    return c > 0 ? c : 256;
}

static size_t sort_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// tmp layout: hist u32[R * n_tiles] | hist_scan u64[R * n_tiles] | scan tmp
size_t sort_tmp_bytes(uint64_t n) {
    uint64_t n_tiles = (n + STILE - 1) / STILE;
    if (n_tiles == 0) n_tiles = 1;
    uint64_t cells = (uint64_t)SMAXR * n_tiles;
#ifdef SORT_LOOKBACK_PROBE
    return sort_align256(cells * 4) + 2 * sort_align256(cells * 8) + sort_align256(scan_tmp_bytes(cells)) + 256;  // (+ the probe's rows)
#else
    return sort_align256(cells * 4) + sort_align256(cells * 8) + sort_align256(scan_tmp_bytes(cells)) + 256;
#endif
}

int sort_pairs(uint32_t *k0, uint64_t *v0, uint32_t *k1, uint64_t *v1, uint64_t n, int key_bits, void *tmp,
               int *result_in_0, hipStream_t s, float *ms_dominant_kernel, int *n_passes, int first_bit) {
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
    p += sort_align256(cells * 4);
    uint64_t *hist_scan = (uint64_t *)p;
    p += sort_align256(cells * 8);
    void *scan_tmp = p;

    // persistent scatter blocks: as many as are resident at once (LDS would allow two per CU, registers decide); asked
    // once per device of the process (handles on different devices must not share the answer)
    static int scatter_blocks_of[64] = {0};
    int dev = 0;
    PAG_HIP_TRY(hipGetDevice(&dev));
    int &scatter_blocks = scatter_blocks_of[dev & 63];
    if (scatter_blocks == 0) {
        int cus = 0, per_cu = 0;
        PAG_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        PAG_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sort_scatter<8>, ST, 0));
        scatter_blocks = cus * (per_cu > 0 ? per_cu : 1);
    }
    const uint32_t scatter_grid = n_tiles < (uint32_t)scatter_blocks ? n_tiles : (uint32_t)scatter_blocks;
    const int hist_per_cu = 8;
    const int hist_blocks = scatter_blocks_cus(dev) * hist_per_cu;
    // timing (the k-mer sort of pag_process asks for it): events out of a pool created once per device, and the one
    // synchronisation their reading takes.  A caller that does not ask gets neither — until round 5 every call created and
    // destroyed 2 x passes events and synchronised the stream.
    const bool timed = ms_dominant_kernel != nullptr;
    static thread_local hipEvent_t ev_pool[64][2 * 8];  // (per host thread: two handles of one device may sort at the same time)
    static thread_local bool ev_ready[64] = {false};
    hipEvent_t *ev = ev_pool[dev & 63];
    if (timed && !ev_ready[dev & 63]) {
        for (int i = 0; i < 2 * 8; ++i) PAG_HIP_TRY(hipEventCreate(&ev[i]));
        ev_ready[dev & 63] = true;
    }
    uint32_t *ka = k0, *kb = k1;
    uint64_t *va = v0, *vb = v1;
    int in0 = 1;
    for (int pass = 0; pass < passes; ++pass) {
        int shift = first_bit + pass * bits;
        int b = key_bits - pass * bits < bits ? key_bits - pass * bits : bits;
        uint32_t rmask = (1u << b) - 1u;
        uint64_t used = (uint64_t)(rmask + 1) * n_tiles;
        const uint32_t hist_grid = n_tiles < (uint32_t)hist_blocks ? n_tiles : (uint32_t)hist_blocks;
        sort_hist<<<dim3(hist_grid), dim3(HT), 0, s>>>(ka, n, shift, rmask, hist, n_tiles, ((uintptr_t)ka & 15u) == 0 ? 1 : 0);
        int rc = scan_u32_to_u64(hist, hist_scan, used, nullptr, scan_tmp, s);
        if (rc != PAG_OK) return rc;
        if (timed) PAG_HIP_TRY(hipEventRecord(ev[2 * pass], s));
        switch (b) {
#define PAG_SCATTER(B)                                                                                              \
    case B:                                                                                                         \
        sort_scatter<B><<<dim3(scatter_grid), dim3(ST), 0, s>>>(ka, va, kb, vb, n, shift, hist_scan, n_tiles);      \
        break;
            PAG_SCATTER(1) PAG_SCATTER(2) PAG_SCATTER(3) PAG_SCATTER(4) PAG_SCATTER(5) PAG_SCATTER(6) PAG_SCATTER(7) PAG_SCATTER(8)
#undef PAG_SCATTER
        }
        if (timed) PAG_HIP_TRY(hipEventRecord(ev[2 * pass + 1], s));
        uint32_t *tk = ka;
        ka = kb;
        kb = tk;
        uint64_t *tv = va;
        va = vb;
        vb = tv;
        in0 = !in0;
    }
    PAG_HIP_TRY(hipGetLastError());
    if (timed) {
        PAG_HIP_TRY(hipStreamSynchronize(s));
        float tot = 0.f;
        for (int pass = 0; pass < passes; ++pass) {
            float ms = 0.f;
            PAG_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * pass], ev[2 * pass + 1]));
            tot += ms;
        }
        *ms_dominant_kernel = tot / passes;
    }
    if (n_passes) *n_passes = passes;
    *result_in_0 = in0;
    return PAG_OK;
}

}  // namespace pagdev
