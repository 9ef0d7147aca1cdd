// k34_segments.hip — K3 (epsilon-windowed vertex merge) and K4 (edge de-duplication), both run over
// the k-mer-sorted tuple streams, one k-mer segment at a time, in place.
//
// K3 = PABruijnGraph::mergeKmerPosition + KMerAdjNode::cluster + sortKmerPosition
//      (reference graph/PABruijnGraph.cpp:259-283, node/KMerAdjNode.tcc:73-137):
//      greedy leader clustering IN INSERTION ORDER — an item joins the FIRST existing leader it is
//      similar to (both coordinates within eps, or both zero), adding to its u16 count (wraps);
//      otherwise it becomes a leader.  Leaders keep the coordinates of their first member and are
//      finally sorted by (ctg, ref).  The reference clusters after pass 1 and again after pass 2 on
//      [pass-1 leaders] ++ [pass-2 items]; one greedy scan over [pass-1 items] ++ [pass-2 items] gives
//      the identical result (leaders are pairwise dissimilar, so re-scanning them reproduces them),
//      which is what the stable sort hands us.  Pass-1 items always have ctg != 0, pass-2 items ctg == 0,
//      so "leaders created by pass 1" = leaders with ctg != 0 (needed for the reference's count lines).
// K4 = PABruijnGraph::mergeEdge + KMerAdjNode::removeDuplicate (graph/PABruijnGraph.cpp:285-297,
//      node/KMerAdjNode.tcc:45-71): per k-mer sort children by (to, step), drop exact duplicates.
//      Payload = to << 32 | step << 1 | pass, so one numeric sort orders by (to, step, pass) and the
//      head of every (to, step) group tells whether pass 1 already had that edge.
//
// Short segments (<= SHORT_MAX records): one thread per segment, sequential.  Long segments (repeats,
// low-complexity k-mers) are queued and handled by one wavefront each: 64 leaders compared per step,
// ballot -> first hit; rank sort through the idle sort ping-pong buffers.
#include <algorithm>

#include "pag_device.hpp"

namespace pagdev {

constexpr uint32_t SHORT_MAX = 32;

__device__ __forceinline__ bool coord_sim(uint32_t a, uint32_t b, uint32_t eps) {
    if (a == 0 || b == 0) return a == 0 && b == 0;
    uint32_t d = a > b ? a - b : b - a;
    return d <= eps;
}
__device__ __forceinline__ bool pos_sim(uint64_t x, uint64_t y, uint32_t eps) {
    return coord_sim((uint32_t)(x >> 32), (uint32_t)(y >> 32), eps) && coord_sim((uint32_t)x, (uint32_t)y, eps);
}

// sum three per-thread counters over the block and add them to counters[0..2] with one atomic each
__device__ __forceinline__ void block_flush3(uint64_t a, uint64_t b, uint64_t c, uint64_t *counters) {
    __shared__ unsigned long long red[3];
    if (threadIdx.x < 3) red[threadIdx.x] = 0;
    __syncthreads();
    uint64_t t;
    wave_excl_sum64(a, &t);
    a = t;
    wave_excl_sum64(b, &t);
    b = t;
    wave_excl_sum64(c, &t);
    c = t;
    if (lane_id() == 0) {
        if (a) atomicAdd(&red[0], (unsigned long long)a);
        if (b) atomicAdd(&red[1], (unsigned long long)b);
        if (c) atomicAdd(&red[2], (unsigned long long)c);
    }
    __syncthreads();
    if (threadIdx.x < 3 && red[threadIdx.x]) atomicAdd((unsigned long long *)&counters[threadIdx.x], red[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------ K3
// Tiles of 256 consecutive records (+ a look-ahead halo of SHORT_MAX) are staged through LDS with coalesced
// loads; the per-segment greedy scans then run on LDS copies (a head thread's chain of dependent reads costs
// ~64 cycles each instead of an HBM/L2 round trip each), and only the leaders go back to HBM.
constexpr int SEG_TILE = 256;
struct SegLds {
    uint32_t key[SEG_TILE + SHORT_MAX + 1];  // key[0] = record before the tile
    uint64_t val[SEG_TILE + SHORT_MAX];
    uint16_t cnt[SEG_TILE + SHORT_MAX];
};

__global__ __launch_bounds__(SEG_TILE) void cluster_short(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                         uint64_t n, uint32_t eps, ClusterOut out,
                                                         uint64_t *__restrict__ long_list, uint32_t *__restrict__ long_count) {
    __shared__ SegLds S;
    uint64_t n_ctg = 0, n_all = 0, n_seg = 0;
    const uint32_t t = threadIdx.x;
    const uint64_t n_tiles = (n + SEG_TILE - 1) / SEG_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * SEG_TILE;
        __syncthreads();
        for (uint32_t x = t; x < SEG_TILE + SHORT_MAX; x += SEG_TILE) {
            uint64_t gi = base + x;
            S.key[x + 1] = gi < n ? key[gi] : 0xFFFFFFFFu;
            S.val[x] = gi < n ? val[gi] : 0;
        }
        if (t == 0) S.key[0] = base ? key[base - 1] : 0xFFFFFFFFu;
        __syncthreads();
        const uint64_t i = base + t;
        if (i < n) {
            const uint32_t kx = S.key[t + 1];
            const bool head = i == 0 || S.key[t] != kx;
            uint32_t seglen = 0;
            if (head) {
                uint32_t len = 1;
                while (len <= SHORT_MAX && i + len < n && S.key[t + 1 + len] == kx) ++len;
                n_seg += 1;
                if (len > SHORT_MAX) {
                    uint32_t slot = atomicAdd(long_count, 1u);
                    long_list[slot] = i;
                    seglen = 0xFFFFFFFFu;  // filled in by cluster_long
                } else {
                    uint32_t p = 0;
                    for (uint32_t it = 0; it < len; ++it) {
                        const uint64_t item = S.val[t + it];
                        bool hit = false;
                        for (uint32_t l = 0; l < p; ++l) {
                            if (pos_sim(item, S.val[t + l], eps)) {
                                S.cnt[t + l] = (uint16_t)(S.cnt[t + l] + 1);
                                hit = true;
                                break;
                            }
                        }
                        if (!hit) {
                            S.val[t + p] = item;
                            S.cnt[t + p] = 1;
                            ++p;
                        }
                    }
                    // sortWithCount: insertion sort by position (distinct keys)
                    for (uint32_t a = 1; a < p; ++a) {
                        const uint64_t v = S.val[t + a];
                        const uint16_t c = S.cnt[t + a];
                        uint32_t b2 = a;
                        while (b2 > 0 && S.val[t + b2 - 1] > v) {
                            S.val[t + b2] = S.val[t + b2 - 1];
                            S.cnt[t + b2] = S.cnt[t + b2 - 1];
                            --b2;
                        }
                        S.val[t + b2] = v;
                        S.cnt[t + b2] = c;
                    }
                    for (uint32_t l = 0; l < p; ++l) {
                        const uint64_t v = S.val[t + l];
                        val[i + l] = v;
                        out.cnt[i + l] = S.cnt[t + l];
                        n_ctg += (v >> 32) != 0;
                    }
                    seglen = p;
                    n_all += p;
                }
            }
            out.seg_len[i] = seglen;
        }
    }
    block_flush3(n_ctg, n_all, n_seg, out.counters);
}

// wave-cooperative: length of the run of `kx` starting at i
__device__ __forceinline__ uint64_t run_end(const uint32_t *__restrict__ key, uint64_t i, uint64_t n, uint32_t kx) {
    uint64_t j = i;
    for (;;) {
        uint64_t x = j + lane_id();
        bool same = x < n && key[x] == kx;
        uint64_t m = __ballot(same);
        if (m != ~0ull) {
            j += (uint64_t)__ffsll((long long)~m) - 1;
            return j;
        }
        j += 64;
    }
}

// rank sort of m records val[0..m) (with companion u16 cnt) by (val, index) through scratch buffers
__device__ __forceinline__ void wave_rank_sort(uint64_t *__restrict__ val, uint16_t *__restrict__ cnt, uint32_t m,
                                               uint64_t *__restrict__ s64, uint32_t *__restrict__ s32) {
    __syncthreads();
    for (uint32_t t = lane_id(); t < m; t += 64) {
        uint64_t v = val[t];
        uint32_t rank = 0;
        for (uint32_t x = 0; x < m; ++x) {
            uint64_t u = val[x];
            rank += (u < v) || (u == v && x < t);
        }
        s64[rank] = v;
        if (cnt) s32[rank] = cnt[t];
    }
    __syncthreads();
    for (uint32_t t = lane_id(); t < m; t += 64) {
        val[t] = s64[t];
        if (cnt) cnt[t] = (uint16_t)s32[t];
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void cluster_long(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                  uint64_t *__restrict__ s64, uint32_t *__restrict__ s32, uint64_t n,
                                                  uint32_t eps, ClusterOut out, const uint64_t *__restrict__ long_list,
                                                  const uint32_t *__restrict__ long_count) {
    const uint32_t lane = lane_id();
    for (uint32_t li = blockIdx.x; li < *long_count; li += gridDim.x) {
        const uint64_t i = long_list[li];
        const uint64_t j = run_end(key, i, n, key[i]);
        uint32_t p = 0;
        for (uint64_t it = i; it < j; ++it) {
            uint64_t item = val[it];
            int hit = -1;
            for (uint32_t c0 = 0; c0 < p && hit < 0; c0 += 64) {
                uint32_t l = c0 + lane;
                bool sim = l < p && pos_sim(item, val[i + l], eps);
                uint64_t m = __ballot(sim);
                if (m) hit = (int)(c0 + __ffsll((long long)m) - 1);
            }
            if (lane == 0) {
                if (hit >= 0) {
                    out.cnt[i + hit] = (uint16_t)(out.cnt[i + hit] + 1);
                } else {
                    val[i + p] = item;
                    out.cnt[i + p] = 1;
                }
            }
            if (hit < 0) ++p;
            __syncthreads();  // the new leader must be visible to every lane before the next item
        }
        wave_rank_sort(val + i, out.cnt + i, p, s64 + i, s32 + i);
        uint32_t n_ctg = 0;
        for (uint32_t l = lane; l < p; l += 64) n_ctg += (val[i + l] >> 32) != 0;
        n_ctg = wave_sum(n_ctg);
        if (lane == 0) {
            out.seg_len[i] = p;
            if (n_ctg) atomicAdd((unsigned long long *)&out.counters[0], (unsigned long long)n_ctg);
            atomicAdd((unsigned long long *)&out.counters[1], (unsigned long long)p);
        }
    }
}

int launch_cluster(const uint32_t *key, uint64_t *val, uint64_t *scratch, uint64_t n, uint32_t eps, ClusterOut out,
                   uint64_t *long_list, uint32_t *long_count, hipStream_t s) {
    PAG_HIP_TRY(hipMemsetAsync(long_count, 0, sizeof(uint32_t), s));
    PAG_HIP_TRY(hipMemsetAsync(out.counters, 0, 4 * sizeof(uint64_t), s));
    if (n == 0) return PAG_OK;
    unsigned grid = (unsigned)std::min<uint64_t>((n + SEG_TILE - 1) / SEG_TILE, 256 * 32);
    cluster_short<<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, eps, out, long_list, long_count);
    // scratch: u64[n] followed by u32[n] (the idle sort ping-pong buffers)
    cluster_long<<<dim3(1024), dim3(64), 0, s>>>(key, val, scratch, (uint32_t *)(scratch + n), n, eps, out,
                                                 long_list, long_count);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

// ------------------------------------------------------------------------------------------------ K4
__global__ __launch_bounds__(SEG_TILE) void edges_short(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                       uint64_t n, EdgeOut out, uint64_t *__restrict__ long_list,
                                                       uint32_t *__restrict__ long_count) {
    __shared__ SegLds S;
    uint64_t n_grp = 0, n_grp1 = 0;
    const uint32_t t = threadIdx.x;
    const uint64_t n_tiles = (n + SEG_TILE - 1) / SEG_TILE;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * SEG_TILE;
        __syncthreads();
        for (uint32_t x = t; x < SEG_TILE + SHORT_MAX; x += SEG_TILE) {
            uint64_t gi = base + x;
            S.key[x + 1] = gi < n ? key[gi] : 0xFFFFFFFFu;
            S.val[x] = gi < n ? val[gi] : 0;
        }
        if (t == 0) S.key[0] = base ? key[base - 1] : 0xFFFFFFFFu;
        __syncthreads();
        const uint64_t i = base + t;
        if (i < n) {
            const uint32_t kx = S.key[t + 1];
            const bool head = i == 0 || S.key[t] != kx;
            uint32_t seglen = 0;
            if (head) {
                uint32_t len = 1;
                while (len <= SHORT_MAX && i + len < n && S.key[t + 1 + len] == kx) ++len;
                if (len > SHORT_MAX) {
                    uint32_t slot = atomicAdd(long_count, 1u);
                    long_list[slot] = i;
                    seglen = 0xFFFFFFFFu;
                } else {
                    for (uint32_t a = 1; a < len; ++a) {
                        const uint64_t v = S.val[t + a];
                        uint32_t b2 = a;
                        while (b2 > 0 && S.val[t + b2 - 1] > v) {
                            S.val[t + b2] = S.val[t + b2 - 1];
                            --b2;
                        }
                        S.val[t + b2] = v;
                    }
                    uint32_t p = 0;
                    uint64_t prev = 0;
                    for (uint32_t a = 0; a < len; ++a) {
                        const uint64_t v = S.val[t + a];
                        if (p == 0 || (prev >> 1) != (v >> 1)) {
                            val[i + p] = v;
                            prev = v;
                            ++p;
                            n_grp1 += (v & 1ull) == 0;
                        }
                    }
                    seglen = p;
                    n_grp += p;
                }
            }
            out.seg_len[i] = seglen;
        }
    }
    block_flush3(n_grp, n_grp1, 0, out.counters);
}

__global__ __launch_bounds__(64) void edges_long(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                uint64_t *__restrict__ s64, uint64_t n, EdgeOut out,
                                                const uint64_t *__restrict__ long_list,
                                                const uint32_t *__restrict__ long_count) {
    const uint32_t lane = lane_id();
    for (uint32_t li = blockIdx.x; li < *long_count; li += gridDim.x) {
        const uint64_t i = long_list[li];
        const uint64_t j = run_end(key, i, n, key[i]);
        const uint32_t m = (uint32_t)(j - i);
        wave_rank_sort(val + i, nullptr, m, s64 + i, nullptr);
        // unique by (to, step): compact group heads through the scratch buffer
        uint32_t p = 0, g1 = 0;
        for (uint32_t c0 = 0; c0 < m; c0 += 64) {
            uint32_t x = c0 + lane;
            uint64_t v = x < m ? val[i + x] : 0;
            bool headg = x < m && (x == 0 || (val[i + x - 1] >> 1) != (v >> 1));
            uint64_t hm = __ballot(headg);
            if (headg) {
                s64[i + p + __popcll(hm & lanemask_lt())] = v;
                g1 += (v & 1ull) == 0;
            }
            p += (uint32_t)__popcll(hm);
        }
        __syncthreads();
        for (uint32_t x = lane; x < p; x += 64) val[i + x] = s64[i + x];
        g1 = wave_sum(g1);
        if (lane == 0) {
            out.seg_len[i] = p;
            atomicAdd((unsigned long long *)&out.counters[0], (unsigned long long)p);
            if (g1) atomicAdd((unsigned long long *)&out.counters[1], (unsigned long long)g1);
        }
        __syncthreads();
    }
}

int launch_edges(const uint32_t *key, uint64_t *val, uint64_t *scratch, uint64_t n, EdgeOut out, uint64_t *long_list,
                 uint32_t *long_count, hipStream_t s) {
    PAG_HIP_TRY(hipMemsetAsync(long_count, 0, sizeof(uint32_t), s));
    PAG_HIP_TRY(hipMemsetAsync(out.counters, 0, 4 * sizeof(uint64_t), s));
    if (n == 0) return PAG_OK;
    unsigned grid = (unsigned)std::min<uint64_t>((n + SEG_TILE - 1) / SEG_TILE, 256 * 32);
    edges_short<<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, out, long_list, long_count);
    edges_long<<<dim3(1024), dim3(64), 0, s>>>(key, val, scratch, n, out, long_list, long_count);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

}  // namespace pagdev
