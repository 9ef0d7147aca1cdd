// util.hip — device prefix sums and the read->reference coverage filter.
//
// Coverage filter = Aligner::covInfHelper + the max-over-sorted-coverage test of parseToRef
// (reference PAGraph/src/tools/align/Aligner.cpp:58-88, Aligner.tcc:140-149):
// coverage is counted per reference base, the array is then SORTED ascending, and an alignment passes
// when max(sorted[t_begin .. t_end)) >= covFilter.  Because the array is sorted, that maximum is
// sorted[t_end - 1], and sorted[i] >= F  <=>  i >= #{bases with coverage < F}.  So the device only
// needs nLow = the number of bases with coverage below F: difference array (atomics) -> prefix sum ->
// count, all streaming.
#include <stdarg.h>
#include <stdio.h>

#include <algorithm>
#include <cstdlib>

#include "pag_device.hpp"

namespace pagdev {

static thread_local char g_err[512] = "";
bool env_timing() { return std::getenv("PAGRAPH_TIMING") != nullptr; }
long long env_int(const char *name, long long otherwise) {
    const char *e = std::getenv(name);
    return e ? std::atoll(e) : otherwise;
}
size_t env_device_sharers() {
    const char *e = std::getenv("PAG_DEVICE_SHARERS");
    return e ? (size_t)std::max(1, std::atoi(e)) : 1;
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

// ------------------------------------------------------------------------------------------------
// generic exclusive scan u32 -> u64, three kernels (tile sums, scan of tile sums, apply)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t block_excl_sum64(uint64_t v, uint64_t *block_total, uint64_t *lds /*[5]*/) {
    uint64_t wtot;
    uint64_t ex = wave_excl_sum64(v, &wtot);
    int w = threadIdx.x >> 6;
    if (lane_id() == 63) lds[w] = wtot;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_THREADS / 64; ++i) {
        uint64_t x = lds[i];
        if (i < w) base += x;
        tot += x;
    }
    __syncthreads();
    *block_total = tot;
    return ex + base;
}

// Element j of a tile belongs to wave j / 1024, round (j % 1024) / 256, lane (j % 256) / 4: a wave's load instruction reads
// 1 KB of consecutive input (16 bytes per lane), its store instructions 2 KB of consecutive output.
constexpr int SCAN_WAVES = SCAN_THREADS / 64;
constexpr int SCAN_WAVE_ITEMS = SCAN_TILE / SCAN_WAVES;  // 1024
constexpr int SCAN_ROUNDS = SCAN_WAVE_ITEMS / 256;       // 4 rounds of 64 lanes x 4 items
static_assert(SCAN_WAVE_ITEMS % 256 == 0, "a wave's share of a tile is a whole number of 1 KB rounds");

__device__ __forceinline__ uint4 scan_load4(const uint32_t *__restrict__ in, uint64_t i, uint64_t n, bool vec) {
    if (vec && i + 4 <= n) return *(const uint4 *)(in + i);
    uint4 v;
    v.x = i < n ? in[i] : 0u;
    v.y = i + 1 < n ? in[i + 1] : 0u;
    v.z = i + 2 < n ? in[i + 2] : 0u;
    v.w = i + 3 < n ? in[i + 3] : 0u;
    return v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                              uint64_t *__restrict__ tile_sums, int vec) {
    __shared__ uint64_t lds[8];
    const uint64_t wave_base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)(threadIdx.x >> 6) * SCAN_WAVE_ITEMS;
    uint64_t s = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const uint4 v = scan_load4(in, wave_base + (uint64_t)r * 256 + lane_id() * 4u, n, vec != 0);
        s += (uint64_t)v.x + v.y + v.z + v.w;
    }
    uint64_t tot;
    block_excl_sum64(s, &tot, lds);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// single block: exclusive scan of the tile sums in place; total -> *total
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums_scan(uint64_t *__restrict__ tile_sums, uint64_t n_tiles,
                                                                   uint64_t *__restrict__ total) {
    __shared__ uint64_t lds[8];
    uint64_t carry = 0;
    for (uint64_t start = 0; start < n_tiles; start += SCAN_THREADS) {
        uint64_t i = start + threadIdx.x;
        uint64_t v = i < n_tiles ? tile_sums[i] : 0;
        uint64_t tot;
        uint64_t ex = block_excl_sum64(v, &tot, lds);
        if (i < n_tiles) tile_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const uint32_t *__restrict__ in, uint64_t n,
                                                          const uint64_t *__restrict__ tile_sums,
                                                          uint64_t *__restrict__ out, int vec) {
    __shared__ uint64_t lds[8];
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    const uint64_t wave_base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)w * SCAN_WAVE_ITEMS;
    uint4 v[SCAN_ROUNDS];
    uint64_t ex[SCAN_ROUNDS];  // exclusive prefix of the lane's four items inside the wave's share
    uint64_t carry = 0;
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        v[r] = scan_load4(in, wave_base + (uint64_t)r * 256 + lane * 4u, n, vec != 0);
        uint64_t tot;
        ex[r] = carry + wave_excl_sum64((uint64_t)v[r].x + v[r].y + v[r].z + v[r].w, &tot);
        carry += tot;
    }
    // the waves' totals -> every wave's base inside the tile
    if (lane == 0) lds[w] = carry;
    __syncthreads();
    uint64_t base = tile_sums[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_WAVES; ++i)
        if ((uint32_t)i < w) base += lds[i];
#pragma unroll
    for (int r = 0; r < SCAN_ROUNDS; ++r) {
        const uint64_t i = wave_base + (uint64_t)r * 256 + lane * 4u;
        const uint64_t e0 = base + ex[r], e1 = e0 + v[r].x, e2 = e1 + v[r].y, e3 = e2 + v[r].z;
        if (vec && i + 4 <= n) {
            ulonglong2 a, b;
            a.x = e0;
            a.y = e1;
            b.x = e2;
            b.y = e3;
            *(ulonglong2 *)(out + i) = a;
            *(ulonglong2 *)(out + i + 2) = b;
        } else {
            if (i < n) out[i] = e0;
            if (i + 1 < n) out[i + 1] = e1;
            if (i + 2 < n) out[i + 2] = e2;
            if (i + 3 < n) out[i + 3] = e3;
        }
    }
}

size_t scan_tmp_bytes(uint64_t n) { return ((n + SCAN_TILE - 1) / SCAN_TILE + 1) * sizeof(uint64_t); }

int scan_u32_to_u64(const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *total_dev, void *tmp, hipStream_t s) {
    if (n == 0) {
        if (total_dev) PAG_HIP_TRY(hipMemsetAsync(total_dev, 0, sizeof(uint64_t), s));
        return PAG_OK;
    }
    uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    uint64_t *tile_sums = (uint64_t *)tmp;
    // 16-byte loads and stores when both arrays allow them (pool slots do; a caller's offset view may not)
    const int vec = (((uintptr_t)in | (uintptr_t)out) & 15u) == 0 ? 1 : 0;
    scan_tile_sums<<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s>>>(in, n, tile_sums, vec);
    scan_tile_sums_scan<<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(tile_sums, n_tiles, total_dev);
    scan_apply<<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s>>>(in, n, tile_sums, out, vec);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

// ------------------------------------------------------------------------------------------------
// coverage filter
// ------------------------------------------------------------------------------------------------
__global__ void cov_mark(const pag_aln *__restrict__ aln, uint64_t n_aln, const pag_ref *__restrict__ refs,
                         const uint64_t *__restrict__ ref_base, uint32_t *__restrict__ diff) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_aln) return;
    pag_aln a = aln[i];
    if (a.target == PAG_NONE || !refs[a.target].accepted || a.t_begin >= a.t_end) return;
    uint64_t b = ref_base[a.target];
    atomicAdd(&diff[b + a.t_begin], 1u);
    atomicAdd(&diff[b + a.t_end], 0xFFFFFFFFu);  // -1 (the region of a reference has len + 1 slots)
}

// running coverage = inclusive prefix sum of the difference array; count bases with coverage < F
__global__ __launch_bounds__(SCAN_THREADS) void cov_count_low(const uint32_t *__restrict__ diff, uint64_t len,
                                                             const uint64_t *__restrict__ tile_sums, uint32_t F,
                                                             unsigned long long *__restrict__ n_low) {
    __shared__ uint64_t lds[8];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = base + i < len ? diff[base + i] : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_excl_sum64(s, &tot, lds) + tile_sums[blockIdx.x];
    uint32_t run = (uint32_t)ex;  // coverage fits 32 bits; the u32 wrap of "-1" entries cancels exactly
    uint32_t low = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        run += v[i];
        if (base + i < len && run < F) ++low;
    }
    low = wave_sum(low);
    if (lane_id() == 0 && low) atomicAdd(n_low, (unsigned long long)low);
}

__global__ void cov_flag(const pag_aln *__restrict__ aln, uint64_t n_aln, const unsigned long long *__restrict__ n_low,
                         uint32_t F, uint8_t *__restrict__ ok) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_aln) return;
    pag_aln a = aln[i];
    uint8_t r = 1;
    if (F > 0) {
        r = 0;
        if (a.target != PAG_NONE && a.t_begin < a.t_end) r = (uint64_t)(a.t_end - 1) >= n_low[a.target];
    }
    ok[i] = r;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// tmp layout: ref_base[n_refs] u64 | n_low[n_refs] u64 | tile_sums | diff[sum(len + 1)] u32
size_t cov_tmp_bytes(const pag_ref *refs_host, uint64_t n_refs) {
    uint64_t tot = 0, maxlen = 0;
    for (uint64_t r = 0; r < n_refs; ++r) {
        tot += (uint64_t)refs_host[r].len + 1;
        if (refs_host[r].len > maxlen) maxlen = refs_host[r].len;
    }
    return align256(n_refs * 8) * 2 + align256(scan_tmp_bytes(maxlen + 1)) + align256(tot * 4) + 256;
}

int launch_cov_filter(const pag_aln *aln, uint64_t n_aln, const pag_ref *refs_dev, const pag_ref *refs_host,
                      uint64_t n_refs, uint32_t cov_filter, uint8_t *cov_ok, void *tmp, size_t tmp_bytes, hipStream_t s) {
    if (n_aln == 0) return PAG_OK;
    const unsigned T = 256;
    unsigned grid_aln = (unsigned)((n_aln + T - 1) / T);
    char *p = (char *)tmp;
    uint64_t *ref_base = (uint64_t *)p;
    p += align256(n_refs * 8);
    unsigned long long *n_low = (unsigned long long *)p;
    p += align256(n_refs * 8);
    if (cov_filter == 0) {
        cov_flag<<<dim3(grid_aln), dim3(T), 0, s>>>(aln, n_aln, n_low, 0, cov_ok);
        PAG_HIP_TRY(hipGetLastError());
        return PAG_OK;
    }
    uint64_t tot = 0, maxlen = 0;
    uint64_t *base_host = (uint64_t *)malloc(n_refs * 8 + 8);
    for (uint64_t r = 0; r < n_refs; ++r) {
        base_host[r] = tot;
        tot += (uint64_t)refs_host[r].len + 1;
        if (refs_host[r].len > maxlen) maxlen = refs_host[r].len;
    }
    uint64_t *tile_sums = (uint64_t *)p;
    p += align256(scan_tmp_bytes(maxlen + 1));
    uint32_t *diff = (uint32_t *)p;
    if ((size_t)(p - (char *)tmp) + tot * 4 > tmp_bytes) {
        free(base_host);
        set_error("coverage scratch too small");
        return PAG_EINVAL;
    }
    hipError_t e = hipMemcpyAsync(ref_base, base_host, n_refs * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // base_host is freed below
    if (e == hipSuccess) e = hipMemsetAsync(n_low, 0, n_refs * 8, s);
    if (e == hipSuccess) e = hipMemsetAsync(diff, 0, tot * 4, s);
    if (e != hipSuccess) {
        free(base_host);
        set_error("coverage setup failed: %s", hipGetErrorString(e));
        return PAG_EFAULT;
    }
    cov_mark<<<dim3(grid_aln), dim3(T), 0, s>>>(aln, n_aln, refs_dev, ref_base, diff);
    for (uint64_t r = 0; r < n_refs; ++r) {
        if (!refs_host[r].accepted || refs_host[r].len == 0) continue;
        uint64_t len = refs_host[r].len;
        uint64_t n_tiles = (len + SCAN_TILE - 1) / SCAN_TILE;
        const uint32_t *d = diff + base_host[r];
        scan_tile_sums<<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s>>>(d, len, tile_sums, ((uintptr_t)d & 15u) == 0 ? 1 : 0);
        scan_tile_sums_scan<<<dim3(1), dim3(SCAN_THREADS), 0, s>>>(tile_sums, n_tiles, nullptr);
        cov_count_low<<<dim3((unsigned)n_tiles), dim3(SCAN_THREADS), 0, s>>>(d, len, tile_sums, cov_filter, n_low + r);
    }
    free(base_host);
    cov_flag<<<dim3(grid_aln), dim3(T), 0, s>>>(aln, n_aln, n_low, cov_filter, cov_ok);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

}  // namespace pagdev
