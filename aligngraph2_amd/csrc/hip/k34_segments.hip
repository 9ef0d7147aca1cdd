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
// Short segments (<= SHORT_MAX = 64 records): one thread per record, bit-mask formulation (see K3 below).  Longer ones are queued
// and handled by one wavefront each with the segment's state ON CHIP: up to 512 leaders in registers (a lane holds leaders l,
// l + 64, ...), the items broadcast from a coalesced load of 64, ballot -> first hit; up to 1 024 edges sorted in LDS.  Beyond
// that (low-complexity k-mers with thousands of positions) the same wavefront works in place through memory, as every long
// segment did until round 6 — one global round trip per item then, 106 + 94 ms for K3 + K4 on a 62.5 Mb block at 40x coverage,
// where most k-mer segments are longer than 32 records (the short path's limit until then).
#include <algorithm>
#include <type_traits>

#include "pag_device.hpp"

namespace pagdev {

// The short path in two widths, chosen per launch from the stream's average segment length (launch_cluster / launch_edges): 32-bit
// masks and 480 owned records per 512-thread tile where most segments are short (20x coverage: ~6 records), 64-bit masks and
// 448 owned records where many are longer than 32 (30x and more).  Same results either way: a segment longer than the width
// goes to the long path.
template <int W>
struct ShortW {
    using Mask = typename std::conditional<W == 32, uint32_t, uint64_t>::type;
    static constexpr uint32_t MAX = W;
    static constexpr int OWN = 512 - W;
};

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
// One thread per RECORD, not per segment.  A block stages SEG_OWN consecutive records plus a look-ahead
// halo of SHORT_MAX through LDS with coalesced loads and owns the segments whose head lies in the first
// SEG_OWN records.  Every record finds its offset `o` in its segment and the segment length by scanning
// the staged keys, then the greedy scan is evaluated without any serial per-segment thread:
//   M[x]  = bit j set iff record x is similar to the EARLIER record j of its segment     (o compares; 64-bit masks)
//   L     = leader mask: record j is a leader iff M[j] & L == 0, folded for j = 0..len-1   (len steps)
//   a non-leader adds 1 to leader ctz(M & L) — the first leader it is similar to, exactly the reference's
//   scan order; a leader's output slot is its rank by (ctg, ref) among the leaders.
// All lanes of a wave are busy and the trip counts are the segment length, instead of one lane in ~5
// running length^2/2 dependent compares (the earlier layout was VALU-issue bound at 45 ms for the C2 set).
constexpr int SEG_TILE = 512;  // threads per block = staged records = owned records + the look-ahead halo of the short path's width
template <int W>
struct SegLds {
    uint64_t flag[SEG_TILE / 64 + 1];  // bit x = record x starts a k-mer segment; word 8 covers record SEG_TILE
    uint64_t val[SEG_TILE];
    typename ShortW<W>::Mask m[SEG_TILE];
    uint32_t cnt[SEG_TILE];
};
// K3 only: a record's two coordinates in the form the similarity masks compare (SimForm below), and whether some coordinate of
// the tile lies too close to 2^32 for that form
struct SimLds {
    uint2 t[SEG_TILE];
    uint32_t slow;
};
// pos_sim in two subtractions and two compares per pair.  A coordinate c becomes c' = c ? c + eps + 1 : 0: "both zero" is
// distance 0, "one zero" a distance of more than eps, two real coordinates keep their distance — so coord_sim(a, b) is
// |a' - b'| <= eps, i.e. (a' + eps) - b' <= 2 eps in unsigned arithmetic (a difference below zero wraps to more than 2 eps).
// Exact as long as no c' + eps wraps: coordinates up to 2^32 - 2 eps - 3; a tile that holds a larger one (SimLds::slow), or an
// eps of 2^30 and more, takes the plain predicate.  The masks were 240 of cluster_short's 570 vector instructions per 64
// records, and the kernel is bound by their number (81 % of the issue slots, profiles/r05_pmc_kernel_mix_end.json).
struct SimForm {
    uint32_t eps, two, limit;
    bool usable;
    __device__ __forceinline__ explicit SimForm(uint32_t e) : eps(e), two(2u * e), limit(0xFFFFFFFFu - 2u * e - 3u), usable(e < (1u << 30)) {}
    __device__ __forceinline__ uint2 of(uint64_t v) const {
        const uint32_t c = (uint32_t)(v >> 32), r = (uint32_t)v;
        return make_uint2(c ? c + eps + 1u : 0u, r ? r + eps + 1u : 0u);
    }
    __device__ __forceinline__ bool fits(uint64_t v) const { return (uint32_t)(v >> 32) <= limit && (uint32_t)v <= limit; }
};

struct SegPos {
    uint32_t o, s, len;
    bool head;    // record is the head of a segment (any length) that starts inside the owned range
    bool active;  // record belongs to a short segment owned by this tile
    bool is_long; // head of a long segment
};

// Stage one tile: values into LDS, segment-start flags as one ballot word per wave.  A record that continues
// the segment of the record before the tile gets no flag, so the leading part of a segment headed in the
// previous tile has no start inside this tile and is left to that tile.
// The tile's words come from registers that were loaded while the tile BEFORE it was worked on (SegRegs, seg_request): staged
// straight from memory, every tile of a block began with a global load round trip in front of its first barrier.
struct SegRegs {
    uint32_t k = 0, kprev = 0, kend0 = 0, kend1 = 0;
    uint64_t v = 0;
};
__device__ __forceinline__ void seg_request(SegRegs &R, const uint32_t *__restrict__ key, const uint64_t *__restrict__ val, uint64_t base, uint64_t n) {
    const uint32_t t = threadIdx.x;
    const uint64_t gi = base + t;
    if (gi < n) {
        R.k = key[gi];
        R.kprev = gi ? key[gi - 1] : 0u;
        R.v = val[gi];
    }
    if (t == 0) {
        const uint64_t ge = base + SEG_TILE;
        if (ge < n) {
            R.kend0 = key[ge - 1];
            R.kend1 = key[ge];
        }
    }
}
template <int W>
__device__ __forceinline__ void seg_stage(SegLds<W> &S, const SegRegs &R, uint64_t base, uint64_t n) {
    const uint32_t t = threadIdx.x;
    const uint64_t gi = base + t;
    bool f = true;  // records past the end close the last segment
    if (gi < n) {
        f = gi == 0 || R.kprev != R.k;
        S.val[t] = R.v;
    }
    const uint64_t w = __ballot(f);
    if ((t & 63u) == 0) S.flag[t >> 6] = w;
    if (t == 0) {
        const uint64_t ge = base + SEG_TILE;
        const bool fe = ge >= n || R.kend0 != R.kend1;
        S.flag[SEG_TILE / 64] = ~1ull | (fe ? 1ull : 0ull);
    }
}

// locate record x of the staged tile inside its k-mer segment: nearest start flag at or before x and the
// next one after it, each at most two flag words away for a short segment
template <int W>
__device__ __forceinline__ SegPos seg_locate(const SegLds<W> &S, uint32_t x, uint64_t base, uint64_t n) {
    constexpr uint32_t SHORT_MAX = ShortW<W>::MAX;
    constexpr int SEG_OWN = ShortW<W>::OWN;
    SegPos r;
    r.o = 0;
    r.s = x;
    r.len = 0;
    r.head = r.active = r.is_long = false;
    const uint32_t w = x >> 6, b = x & 63u;
    const uint64_t cur = S.flag[w];
    const uint64_t prev = w ? S.flag[w - 1] : 0ull;
    const uint64_t next = S.flag[w + 1];
    const uint64_t le = cur & (~0ull >> (63u - b));
    uint32_t s = 0xFFFFFFFFu;
    if (le)
        s = w * 64u + 63u - (uint32_t)__builtin_clzll(le);
    else if (prev)
        s = (w - 1u) * 64u + 63u - (uint32_t)__builtin_clzll(prev);
    const uint64_t gt = cur & (~1ull << b);
    uint32_t e = 0xFFFFFFFFu;  // no start flag within reach: the segment is long
    if (gt)
        e = w * 64u + (uint32_t)__builtin_ctzll(gt);
    else if (next)
        e = (w + 1u) * 64u + (uint32_t)__builtin_ctzll(next);
    if (base + x >= n || s == 0xFFFFFFFFu || s >= (uint32_t)SEG_OWN || x - s > SHORT_MAX) return r;
    r.o = x - s;
    r.s = s;
    r.len = e == 0xFFFFFFFFu ? SHORT_MAX + 1 : e - s;
    if (r.len > SHORT_MAX) r.len = SHORT_MAX + 1;
    r.head = r.o == 0;
    r.is_long = r.len > SHORT_MAX;
    r.active = !r.is_long;
    return r;
}

template <int W>
__global__ __launch_bounds__(SEG_TILE) void cluster_short(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                         uint64_t n, uint32_t eps, ClusterOut out,
                                                         uint64_t *__restrict__ long_list, uint32_t *__restrict__ long_count) {
    using Mask = typename ShortW<W>::Mask;
    constexpr int SEG_OWN = ShortW<W>::OWN;
    __shared__ SegLds<W> S;
    __shared__ SimLds Q;
    uint64_t n_ctg = 0, n_all = 0, n_seg = 0;
    const uint32_t t = threadIdx.x;
    const uint64_t n_tiles = (n + SEG_OWN - 1) / SEG_OWN;
    const SimForm F(eps);
    if (t == 0) Q.slow = 0u;
    SegRegs R;
    if (blockIdx.x < n_tiles) seg_request(R, key, val, (uint64_t)blockIdx.x * SEG_OWN, n);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * SEG_OWN;
        __syncthreads();
        seg_stage(S, R, base, n);
        if (base + t < n) {
            Q.t[t] = F.of(R.v);
            if (!F.fits(R.v)) Q.slow = 1u;
        }
        if (tile + gridDim.x < n_tiles) seg_request(R, key, val, (tile + gridDim.x) * SEG_OWN, n);  // (in flight while this tile is worked on)
        __syncthreads();
        const SegPos P = seg_locate(S, t, base, n);
        const uint64_t v = S.val[t];
        const bool fast = F.usable && Q.slow == 0u;  // (the same answer for every thread of the tile)
        Mask M = 0;
        if (P.active) {
            if (fast) {
                const uint2 a = Q.t[t];
                const uint32_t ac = a.x + F.eps, ar = a.y + F.eps;
                for (uint32_t j = 0; j < P.o; ++j) {
                    const uint2 b = Q.t[P.s + j];
                    M |= (Mask)((ac - b.x <= F.two) & (ar - b.y <= F.two)) << j;
                }
            } else {
                for (uint32_t j = 0; j < P.o; ++j) M |= (Mask)pos_sim(v, S.val[P.s + j], eps) << j;
            }
        }
        S.m[t] = M;
        S.cnt[t] = 1;
        __syncthreads();
        if (t == 0) Q.slow = 0u;  // (every thread has read it; the next tile's writers are behind the barrier at the top of the loop)
        Mask L = 0;
        bool leader = false;
        uint32_t rank = 0;
        if (P.active) {
            for (uint32_t j = 0; j < P.len; ++j) L |= (Mask)((S.m[P.s + j] & L) == 0) << j;
            leader = (L >> P.o) & (Mask)1;
            if (!leader) {
                atomicAdd(&S.cnt[P.s + (uint32_t)__builtin_ctzll((uint64_t)(M & L))], 1u);
            } else {
                // sortWithCount: slot = rank by (ctg, ref) among the leaders (distinct keys)
                for (uint32_t j = 0; j < P.len; ++j) rank += (uint32_t)((L >> j) & 1u) & (uint32_t)(S.val[P.s + j] < v);  // (no short circuit: see edges_short)
            }
        }
        __syncthreads();
        if (leader) {
            const uint64_t dst = base + P.s + rank;
            val[dst] = v;
            out.cnt[dst] = (uint16_t)S.cnt[t];
            n_ctg += (v >> 32) != 0;
        }
        // seg_len: the head of a segment says how many leaders it has; the other LEADER slots say that they are one and how
        // far behind the head they lie (SEG_LEADER | offset), every other slot 0 — so that later passes can run one thread per
        // slot (the traversal's compaction, k5_travel.hip).  Every slot has exactly one writer: the tile that owns the head of
        // its segment (slots of a short segment may lie in that tile's look-ahead halo), cluster_long for a long segment.
        if (base + t < n) {
            if (P.head) {  // (t < SEG_OWN)
                uint32_t seglen;
                n_seg += 1;
                if (P.is_long) {
                    uint32_t slot = atomicAdd(long_count, 1u);
                    long_list[slot] = base + t;
                    seglen = 0xFFFFFFFFu;  // filled in by cluster_long
                } else {
                    seglen = (uint32_t)__popcll((uint64_t)L);
                    n_all += seglen;
                }
                out.seg_len[base + t] = seglen;
            } else if (P.active) {
                out.seg_len[base + t] = P.o < (uint32_t)__popcll((uint64_t)L) ? (SEG_LEADER | P.o) : 0u;
            }
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

// A long segment with its leaders in registers: lane l holds leaders l, l + 64, ... (values and counts), at most LONG_REGS * 64
// of them.  The items come in coalesced loads of 64 and are handed round by shuffles; an item is compared with 64 leaders per
// step, the ballot's first set bit is the first leader it is similar to — the reference's scan order.  Nothing is written until the
// segment is through (the leaders' slots are the segment's own first slots, which hold items until then).  Returns false when the
// segment has more leaders than the registers hold: nothing was written, the caller does it in place.
constexpr int LONG_REGS = 8;
__device__ __forceinline__ bool cluster_long_on_chip(uint64_t *__restrict__ val, uint64_t i, uint64_t j, uint32_t eps, ClusterOut out, uint64_t *lds_leaders) {
    const uint32_t lane = lane_id();
    uint64_t lv[LONG_REGS];
    uint32_t lc[LONG_REGS];
#pragma unroll
    for (int r = 0; r < LONG_REGS; ++r) lv[r] = 0ull, lc[r] = 0u;
    uint32_t p = 0;
    for (uint64_t c0 = i; c0 < j; c0 += 64) {
        const uint64_t mine = c0 + lane < j ? val[c0 + lane] : 0ull;
        const uint32_t nk = (uint32_t)(j - c0 < 64 ? j - c0 : 64);
        for (uint32_t kk = 0; kk < nk; ++kk) {
            const uint64_t item = __shfl(mine, (int)kk);
            int hit = -1;
#pragma unroll
            for (int r = 0; r < LONG_REGS; ++r) {
                if (hit < 0 && (uint32_t)r * 64u < p) {  // (wave-uniform)
                    const bool sim = (uint32_t)r * 64u + lane < p && pos_sim(item, lv[r], eps);
                    const uint64_t m = __ballot(sim);
                    if (m) hit = r * 64 + (int)__ffsll((long long)m) - 1;
                }
            }
            if (hit < 0) {
                if (p >= (uint32_t)LONG_REGS * 64u) return false;
#pragma unroll
                for (int r = 0; r < LONG_REGS; ++r)
                    if ((p >> 6) == (uint32_t)r && (p & 63u) == lane) lv[r] = item, lc[r] = 1u;
                ++p;
            } else {
#pragma unroll
                for (int r = 0; r < LONG_REGS; ++r)
                    if (((uint32_t)hit >> 6) == (uint32_t)r && ((uint32_t)hit & 63u) == lane) lc[r] += 1u;
            }
        }
    }
    // sortWithCount: a leader's slot = its rank by (ctg, ref) among the leaders (distinct keys); the leaders staged in LDS, read back
    // by every lane at the same address (broadcast)
#pragma unroll
    for (int r = 0; r < LONG_REGS; ++r)
        if ((uint32_t)r * 64u + lane < p) lds_leaders[r * 64 + lane] = lv[r];
    __syncthreads();
    uint32_t rank[LONG_REGS];
#pragma unroll
    for (int r = 0; r < LONG_REGS; ++r) rank[r] = 0u;
    for (uint32_t x = 0; x < p; ++x) {
        const uint64_t u = lds_leaders[x];
#pragma unroll
        for (int r = 0; r < LONG_REGS; ++r) rank[r] += u < lv[r];
    }
    uint32_t n_ctg = 0;
#pragma unroll
    for (int r = 0; r < LONG_REGS; ++r)
        if ((uint32_t)r * 64u + lane < p) {
            val[i + rank[r]] = lv[r];
            out.cnt[i + rank[r]] = (uint16_t)lc[r];
            n_ctg += (lv[r] >> 32) != 0;
        }
    n_ctg = wave_sum(n_ctg);
    for (uint64_t l = lane; l < j - i; l += 64) out.seg_len[i + l] = l == 0 ? p : (l < p ? (SEG_LEADER | (uint32_t)l) : 0u);
    if (lane == 0) {
        if (n_ctg) atomicAdd((unsigned long long *)&out.counters[0], (unsigned long long)n_ctg);
        atomicAdd((unsigned long long *)&out.counters[1], (unsigned long long)p);
    }
    __syncthreads();  // (the staging array is written again by the next segment)
    return true;
}

__global__ __launch_bounds__(64) void cluster_long(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                  uint64_t *__restrict__ s64, uint32_t *__restrict__ s32, uint64_t n,
                                                  uint32_t eps, ClusterOut out, const uint64_t *__restrict__ long_list,
                                                  const uint32_t *__restrict__ long_count) {
    __shared__ uint64_t lds_leaders[LONG_REGS * 64];
    const uint32_t lane = lane_id();
    for (uint32_t li = blockIdx.x; li < *long_count; li += gridDim.x) {
        const uint64_t i = long_list[li];
        const uint64_t j = run_end(key, i, n, key[i]);
        if (cluster_long_on_chip(val, i, j, eps, out, lds_leaders)) continue;
        // more leaders than the registers hold: in place, through memory
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
        for (uint64_t l = lane; l < j - i; l += 64) out.seg_len[i + l] = l == 0 ? p : (l < p ? (SEG_LEADER | (uint32_t)l) : 0u);
        if (lane == 0) {
            if (n_ctg) atomicAdd((unsigned long long *)&out.counters[0], (unsigned long long)n_ctg);
            atomicAdd((unsigned long long *)&out.counters[1], (unsigned long long)p);
        }
    }
}

int launch_cluster(const uint32_t *key, uint64_t *val, uint64_t *scratch, uint64_t n, uint32_t eps, ClusterOut out,
                   uint64_t *long_list, uint32_t *long_count, hipStream_t s, bool wide) {
    PAG_HIP_TRY(hipMemsetAsync(long_count, 0, sizeof(uint32_t), s));
    PAG_HIP_TRY(hipMemsetAsync(out.counters, 0, 4 * sizeof(uint64_t), s));
    if (n == 0) return PAG_OK;
    if (wide) {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + ShortW<64>::OWN - 1) / ShortW<64>::OWN, 256 * 16);
        cluster_short<64><<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, eps, out, long_list, long_count);
    } else {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + ShortW<32>::OWN - 1) / ShortW<32>::OWN, 256 * 16);
        cluster_short<32><<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, eps, out, long_list, long_count);
    }
    // scratch: u64[n] followed by u32[n] (the idle sort ping-pong buffers)
    cluster_long<<<dim3(1024), dim3(64), 0, s>>>(key, val, scratch, (uint32_t *)(scratch + n), n, eps, out,
                                                 long_list, long_count);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

// ------------------------------------------------------------------------------------------------ K4
// Same record-per-thread layout as cluster_short: a record survives iff it is the smallest (value, index) of its
// (to, step) group, and its output slot is the number of surviving records with a smaller value.
template <int W>
__global__ __launch_bounds__(SEG_TILE) void edges_short(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                       uint64_t n, EdgeOut out, uint64_t *__restrict__ long_list,
                                                       uint32_t *__restrict__ long_count) {
    using Mask = typename ShortW<W>::Mask;
    constexpr int SEG_OWN = ShortW<W>::OWN;
    __shared__ SegLds<W> S;
    uint64_t n_grp = 0, n_grp1 = 0;
    const uint32_t t = threadIdx.x;
    const uint64_t n_tiles = (n + SEG_OWN - 1) / SEG_OWN;
    SegRegs R;
    if (blockIdx.x < n_tiles) seg_request(R, key, val, (uint64_t)blockIdx.x * SEG_OWN, n);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t base = tile * SEG_OWN;
        __syncthreads();
        seg_stage(S, R, base, n);
        if (tile + gridDim.x < n_tiles) seg_request(R, key, val, (tile + gridDim.x) * SEG_OWN, n);  // (in flight while this tile is worked on)
        __syncthreads();
        const SegPos P = seg_locate(S, t, base, n);
        const uint64_t v = S.val[t];
        bool kept = false;
        if (P.active) {
            // (no short circuit: written as `kept = kept && ..` every turn waited for its own LDS read before it knew whether the
            // lane goes on — one exposed round trip and a pair of exec-mask branches per record of the segment)
            bool dead = false;
            for (uint32_t j = 0; j < P.len; ++j) {
                const uint64_t u = S.val[P.s + j];
                dead |= ((u ^ v) < 2ull) & ((u < v) | ((u == v) & (j < P.o)));
            }
            kept = !dead;
        }
        S.m[t] = kept ? (Mask)1 : (Mask)0;
        __syncthreads();
        uint32_t p = 0;
        if (P.active && (kept || P.head)) {
            uint32_t rank = 0;
            for (uint32_t j = 0; j < P.len; ++j) {
                const bool kj = S.m[P.s + j] != 0;
                p += kj;
                rank += (uint32_t)kj & (uint32_t)(S.val[P.s + j] < v);
            }
            if (kept) {
                val[base + P.s + rank] = v;
                n_grp1 += (v & 1ull) == 0;
            }
            if (P.head) n_grp += p;
        }
        if (t < (uint32_t)SEG_OWN && base + t < n) {
            uint32_t seglen = 0;
            if (P.head) {
                if (P.is_long) {
                    uint32_t slot = atomicAdd(long_count, 1u);
                    long_list[slot] = base + t;
                    seglen = 0xFFFFFFFFu;
                } else {
                    seglen = p;
                }
            }
            out.seg_len[base + t] = seglen;
        }
    }
    block_flush3(n_grp, n_grp1, 0, out.counters);
}

constexpr uint32_t EDGE_LDS = 1024;  // edges of a long segment sorted in LDS (more: through memory)
__global__ __launch_bounds__(64) void edges_long(const uint32_t *__restrict__ key, uint64_t *__restrict__ val,
                                                uint64_t *__restrict__ s64, uint64_t n, EdgeOut out,
                                                const uint64_t *__restrict__ long_list,
                                                const uint32_t *__restrict__ long_count) {
    __shared__ uint64_t lds_in[EDGE_LDS], lds_sorted[EDGE_LDS];
    const uint32_t lane = lane_id();
    for (uint32_t li = blockIdx.x; li < *long_count; li += gridDim.x) {
        const uint64_t i = long_list[li];
        const uint64_t j = run_end(key, i, n, key[i]);
        const uint32_t m = (uint32_t)(j - i);
        if (m <= EDGE_LDS) {
            // the segment in LDS: rank sort by (value, index) with broadcast reads, then the heads of the (to, step) groups compacted
            // in order
            for (uint32_t x = lane; x < m; x += 64) lds_in[x] = val[i + x];
            __syncthreads();
            for (uint32_t t = lane; t < m; t += 64) {
                const uint64_t v = lds_in[t];
                uint32_t rank = 0;
                for (uint32_t x = 0; x < m; ++x) {
                    const uint64_t u = lds_in[x];
                    rank += (u < v) || (u == v && x < t);
                }
                lds_sorted[rank] = v;
            }
            __syncthreads();
            uint32_t p = 0, g1 = 0;
            for (uint32_t c0 = 0; c0 < m; c0 += 64) {
                const uint32_t x = c0 + lane;
                const uint64_t v = x < m ? lds_sorted[x] : 0;
                const bool headg = x < m && (x == 0 || (lds_sorted[x - 1] >> 1) != (v >> 1));
                const uint64_t hm = __ballot(headg);
                if (headg) {
                    val[i + p + (uint32_t)__popcll(hm & lanemask_lt())] = v;
                    g1 += (v & 1ull) == 0;
                }
                p += (uint32_t)__popcll(hm);
            }
            g1 = wave_sum(g1);
            if (lane == 0) {
                out.seg_len[i] = p;
                atomicAdd((unsigned long long *)&out.counters[0], (unsigned long long)p);
                if (g1) atomicAdd((unsigned long long *)&out.counters[1], (unsigned long long)g1);
            }
            __syncthreads();
            continue;
        }
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
                 uint32_t *long_count, hipStream_t s, bool wide) {
    PAG_HIP_TRY(hipMemsetAsync(long_count, 0, sizeof(uint32_t), s));
    PAG_HIP_TRY(hipMemsetAsync(out.counters, 0, 4 * sizeof(uint64_t), s));
    if (n == 0) return PAG_OK;
    if (wide) {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + ShortW<64>::OWN - 1) / ShortW<64>::OWN, 256 * 16);
        edges_short<64><<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, out, long_list, long_count);
    } else {
        const unsigned grid = (unsigned)std::min<uint64_t>((n + ShortW<32>::OWN - 1) / ShortW<32>::OWN, 256 * 16);
        edges_short<32><<<dim3(grid), dim3(SEG_TILE), 0, s>>>(key, val, n, out, long_list, long_count);
    }
    edges_long<<<dim3(1024), dim3(64), 0, s>>>(key, val, scratch, n, out, long_list, long_count);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

}  // namespace pagdev
