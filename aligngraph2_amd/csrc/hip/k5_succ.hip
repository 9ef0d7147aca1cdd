// k5_succ.hip — the successor records of the traversal graph: PABruijnGraph::searchSuccessors + checkPosition + isEdgeSimilar
// (PAGraph/src/tools/graph/PABruijnGraph.cpp:143-197, 385-400) evaluated once for every vertex of the view, stored in
// coordinate order for the walker (k5_travel.hip); pag_successors (one vertex's list in the caller's terms); the test hook of
// the match predicates.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pag_device.hpp"
#include "pag_travel.hpp"
#include "trav_device.hpp"

namespace pagdev {

// =================================================================================================
// successor records (round 6): ONE evaluation of the candidate pairs -> emission stream -> sort by source -> finish
// =================================================================================================
// searchSuccessors + checkPosition + isEdgeSimilar (PABruijnGraph.cpp:143-197, 385-400) for every vertex of the view.
// Threads run over the vertices in k-mer-major order (the order of the CSR): neighbouring threads belong to the same k-mer node,
// so the node's edge list and the position lists of its target nodes are shared through the caches.  Every candidate pair is
// evaluated ONCE; an accepted one is appended, finished but for what only the coordinate order knows, to an EMISSION STREAM
// (key = the source's new id, value = the target's new id | step | grade | edge similarity): the waves take room in the stream
// 2 048 slots at a time and the lanes that accept a candidate at the same moment share one allocation (emit_append).  The stream
// is then sorted by its keys with the k-mer sort's own kernels (k2_sort.hip, stable: a vertex's records stay in the order its
// thread appended them, which is the reference's — edge order, then position order), the keys' run boundaries are the offset
// table, and one pass over the sorted stream, in coordinate order, adds what a walk wants to find in a record (the target's
// contig coordinate and its own record range: neighbours on the strand, cache hits).
// Until round 5 the records were built by two walks over the candidate pairs (count with an acceptance mask, scan, fill through
// the mask to coordinate-ordered places, link): the counts and the 16-byte records went to random places from k-mer-major
// threads — partial-line writes, 137 GB for 8.4 GB of records at BASELINE configs[1].
struct __attribute__((packed, aligned(4))) U32x4 { uint32_t a[4]; };
struct __attribute__((packed, aligned(8))) U64x4 { uint64_t a[4]; };

constexpr uint32_t EMIT_CHUNK = EMIT_CHUNK_SLOTS;  // slots a wave takes from the stream at a time (its last chunk and up to 63 slots at the end
                                        // of every chunk stay fillers)
struct EmitStream {
    uint32_t *key;               // [cap] preset to all ones — a filler: behind every vertex id in the sorted bits
    uint64_t *val;               // [cap] target's new id | meta << 32 (meta: step (24 bits) | grade << 24 | isEdgeSimilar().first << 27)
    uint64_t cap;
    unsigned long long *cursor;  // slots handed out; beyond cap: the stream was too small (nothing is written beyond it)
    unsigned long long *n_real;  // records appended
};
struct EmitWave {  // (LDS, one per wave; read and written by the leader of an appending group only)
    uint32_t base_lo, base_hi, used, n;
};
// (a pointer that SAYS it points into LDS: through a plain `volatile EmitWave *` every access became a flat load / store with
// system-scope cache flags and a full wait behind it — address-space inference leaves volatile accesses alone)
typedef volatile __attribute__((address_space(3))) EmitWave *EmitWaveLds;
// The lanes that are HERE together — any subset of the wave: the callers sit in divergent loops — append one record each.  One
// of them takes the room (from the wave's chunk, or a new chunk from the stream's cursor: one global atomic per 2 048 slots);
// later appends of a lane get higher slots than its earlier ones, so a vertex's records stay in order.
__device__ __forceinline__ void emit_append(const EmitStream &S, EmitWaveLds W, uint32_t key, uint64_t val) {
    const uint64_t m = __ballot(1);
    const uint32_t n = (uint32_t)__popcll(m), rank = (uint32_t)__popcll(m & lanemask_lt());
    uint32_t lo = 0, hi = 0;
    if (rank == 0u) {
        uint32_t used = W->used;
        uint64_t base = (uint64_t)W->base_lo | ((uint64_t)W->base_hi << 32);
        if (used + n > EMIT_CHUNK) {
            base = atomicAdd(S.cursor, (unsigned long long)EMIT_CHUNK);
            used = 0u;
            W->base_lo = (uint32_t)base;
            W->base_hi = (uint32_t)(base >> 32);
        }
        W->used = used + n;
        W->n = W->n + n;
        const uint64_t at = base + used;
        lo = (uint32_t)at;
        hi = (uint32_t)(at >> 32);
    }
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);  // (the first active lane is the one with rank 0)
    hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
    const uint64_t slot = ((uint64_t)lo | ((uint64_t)hi << 32)) + rank;
    if (slot < S.cap) {
        S.key[slot] = key;
        S.val[slot] = val;
    }
}
// ... ALL lanes of the wave are here: room for n records of every lane, in lane order (n <= EMIT_CHUNK / 64); returns the lane's
// first slot
__device__ __forceinline__ uint64_t emit_alloc_wave(const EmitStream &S, EmitWaveLds W, uint32_t n) {
    uint32_t tot;
    const uint32_t ex = wave_excl_sum(n, &tot);
    uint32_t lo = 0, hi = 0;
    if (lane_id() == 0u && tot) {
        uint32_t used = W->used;
        uint64_t base = (uint64_t)W->base_lo | ((uint64_t)W->base_hi << 32);
        if (used + tot > EMIT_CHUNK) {
            base = atomicAdd(S.cursor, (unsigned long long)EMIT_CHUNK);
            used = 0u;
            W->base_lo = (uint32_t)base;
            W->base_hi = (uint32_t)(base >> 32);
        }
        W->used = used + tot;
        W->n = W->n + tot;
        const uint64_t at = base + used;
        lo = (uint32_t)at;
        hi = (uint32_t)(at >> 32);
    }
    lo = (uint32_t)__shfl((int)lo, 0);
    hi = (uint32_t)__shfl((int)hi, 0);
    return ((uint64_t)lo | ((uint64_t)hi << 32)) + ex;
}
__device__ __forceinline__ uint64_t emit_value(uint32_t tgt, uint32_t step, uint32_t grade, uint32_t esim) {
    return (uint64_t)tgt | ((uint64_t)((step & 0xFFFFFFu) | (grade << 24) | ((esim & 1u) << 27)) << 32);
}
__device__ __forceinline__ void emit_wave_init(EmitWaveLds W) {  // (lane 0 of the wave; a barrier or wave-level sync after it)
    W->base_lo = 0u;
    W->base_hi = 0u;
    W->used = EMIT_CHUNK;  // (the first append takes a chunk)
    W->n = 0u;
}

// MODE 0: a thread per vertex; a vertex whose k-mer node has more than heavy_limit candidate pairs (the positions of all target
// nodes: the same number for every vertex of the node) is appended to heavy_list instead and done by MODE 1: a wave per listed
// vertex, 64 candidates of a position list per coalesced load (the lanes of a wave in MODE 0 run as long as the one with the
// longest lists).
// (launch bounds: 93 registers and none spilled — five waves per SIMD; kept to 64 or 80 registers for eight or six the kernel
// spills 53 / 13 of them and takes 35.0 / 35.8 ms at BASELINE configs[1] against 31.0, profiles/r06_emit_probe.txt)
template <int MODE, int BLOCKS>
__global__ void __launch_bounds__(256, BLOCKS) k_succ_emit(TravGraph G, uint32_t dev, double err, EmitStream S, uint32_t *__restrict__ heavy_list,
                                                           unsigned long long *__restrict__ heavy_n, uint32_t heavy_limit) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    __shared__ IncLds inc_bands;
    __shared__ EmitWave emit_wave[4];
    d_ratio_table_fill(ratio_tab, err);
    const bool inc_lds = inc_lds_fill(inc_bands, G);
    EmitWaveLds W = (EmitWaveLds)&emit_wave[threadIdx.x >> 6];
    if (lane_id() == 0u) emit_wave_init(W);
    __syncthreads();
    const uint32_t marker_meta = 1u | (GRADE_POISON_IF_LEAP << 24), poison_meta = 1u | (GRADE_POISON << 24);
    if (MODE == 0) {
        if (!heavy_list) heavy_limit = 0xFFFFFFFFu;
        const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
        for (uint64_t v0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); v0 < G.n_pos; v0 += stride) {  // (wave-uniform trip count)
            const uint64_t v = v0 + lane_id();
            bool heavy = false, marker = false;
            uint32_t nb = 0, u_self = 0;
            uint32_t bp0 = 0, bp1 = 0, bp2 = 0, bp3 = 0, bm0 = 0, bm1 = 0, bm2 = 0, bm3 = 0;
            if (v < G.n_pos) {
                const uint32_t u = G.newid[v];
                u_self = u;
                // successors of it may lie outside the region this graph holds (k_mark_incomplete).  A coordinate-free vertex gets one
                // poison record IN PLACE of its successors; a vertex on a contig keeps its successors — those that follow the contig
                // are all here — and gets one marker record behind them that only counts where a walk could take a Skip grade
                const bool inc = vertex_incomplete(G, inc_bands, inc_lds, v, u);
                if (inc && u < G.n_zero) {
                    emit_append(S, W, u, (uint64_t)u | ((uint64_t)poison_meta << 32));
                } else {
                    const uint64_t rootp = G.vpos[v];
                    const uint32_t rc = (uint32_t)(rootp >> 32), rr = (uint32_t)rootp;
                    const uint32_t node = G.vnode[v];
                    const uint32_t e_lo = G.nedge_off[node], e_hi = G.nedge_off[node + 1];
                    if (heavy_limit != 0xFFFFFFFFu) {  // candidate pairs of the node (the edges carry their targets' list lengths)
                        uint32_t total = 0;
                        for (uint32_t eb = e_lo; eb < e_hi && total <= heavy_limit; eb += 4u) {
                            const U32x4 toL = *(const U32x4 *)(G.eto + eb), stL = *(const U32x4 *)(G.estep + eb);  // (padded by four entries)
#pragma unroll
                            for (uint32_t t = 0; t < 4u; ++t) {
                                if (eb + t >= e_hi) break;
                                uint32_t step, p0, q;
                                edge_target(G, toL.a[t], stL.a[t], &step, &p0, &q);
                                total += q;
                            }
                        }
                        heavy = total > heavy_limit;
                    }
                    if (!heavy) {
                        for (uint32_t eb = e_lo; eb < e_hi; eb += 4u) {
                            // (targets and position ranges of up to four edges requested together: two round trips for the four)
                            uint32_t st4[4], p04[4], q4[4];
                            const U32x4 toL = *(const U32x4 *)(G.eto + eb), stL = *(const U32x4 *)(G.estep + eb);
#pragma unroll
                            for (uint32_t t = 0; t < 4u; ++t) {
                                const bool have = eb + t < e_hi;
                                edge_target(G, have ? toL.a[t] : PAG_NONE, have ? stL.a[t] : 0u, &st4[t], &p04[t], &q4[t]);
                            }
#pragma unroll
                            for (uint32_t t4 = 0; t4 < 4u; ++t4) {
                                const uint32_t step = st4[t4], p0 = p04[t4], q = q4[t4];
                                if (q == 0u) continue;
                                const uint32_t entry = step < RATIO_TAB_N ? ratio_tab[step] : RATIO_TAB_NONE;  // (the ratio tests of this edge, see d_ratio_entry)
                                // four candidates per turn, their positions requested together: one at a time, every candidate cost a full
                                // memory round trip (load -> tests -> next load)
                                for (uint32_t jb = 0; jb < q; jb += 4u) {
                                    const U64x4 pqL = *(const U64x4 *)(G.vpos + p0 + jb);  // (padded by four entries)
                                    // The four candidates are only GRADED here — a byte each, 0 = rejected — and the accepted ones taken
                                    // in order by the loop behind: one copy of the bookkeeping, run as often as the wave's busiest lane
                                    // accepted (one in ten candidates is: once or twice), where it stood behind every one of the sixteen
                                    // evaluation sites of an edge group and ran at each as soon as ANY lane accepted (a third of the
                                    // kernel's instructions, 55 KB of code).
                                    uint32_t am = 0;
#pragma unroll
                                    for (uint32_t t = 0; t < 4u; ++t) {
                                        if (jb + t >= q) break;
                                        const uint32_t pc = (uint32_t)(pqL.a[t] >> 32), pr = (uint32_t)pqL.a[t];
                                        uint32_t esim;
                                        const int grade = d_check_position_any(rc, rr, pc, pr, step, dev, err, entry, &esim);
                                        am |= ((uint32_t)grade | ((esim & 1u) << 3)) << (8u * t);
                                    }
                                    while (am) {
                                        uint32_t t = ((uint32_t)__ffs((int)am) - 1u) >> 3;
                                        uint32_t g = (am >> (8u * t)) & 0xFFu;
                                        am &= ~(0xFFu << (8u * t));
                                        if ((g & 7u) == (uint32_t)G_OOPS) continue;  // (never: the contig side is only similar in accepted pairs, d_check_position)
                                        const uint32_t j = jb + t;
                                        const uint32_t meta = (step & 0xFFFFFFu) | ((g & 7u) << 24) | ((g >> 3) << 27);
                                        if (nb < 4u) {  // (the lane's first four accepted candidates wait in registers for the wave's common append)
                                            bp0 = nb == 0u ? p0 + j : bp0, bm0 = nb == 0u ? meta : bm0;
                                            bp1 = nb == 1u ? p0 + j : bp1, bm1 = nb == 1u ? meta : bm1;
                                            bp2 = nb == 2u ? p0 + j : bp2, bm2 = nb == 2u ? meta : bm2;
                                            bp3 = nb == 3u ? p0 + j : bp3, bm3 = nb == 3u ? meta : bm3;
                                            ++nb;
                                        } else {  // a fifth: the four go first (a vertex's records stay in order), the rest of its list follows them directly
                                            if (nb == 4u) {
                                                emit_append(S, W, u, (uint64_t)G.newid[bp0] | ((uint64_t)bm0 << 32));
                                                emit_append(S, W, u, (uint64_t)G.newid[bp1] | ((uint64_t)bm1 << 32));
                                                emit_append(S, W, u, (uint64_t)G.newid[bp2] | ((uint64_t)bm2 << 32));
                                                emit_append(S, W, u, (uint64_t)G.newid[bp3] | ((uint64_t)bm3 << 32));
                                                nb = 5u;
                                            }
                                            emit_append(S, W, u, (uint64_t)G.newid[p0 + j] | ((uint64_t)meta << 32));
                                        }
                                    }
                                }
                            }
                        }
                        marker = inc;
                    }
                }
            }
            {   // the wave's common append: every lane's waiting records (none left where a fifth came), their targets' new ids asked for together
                const uint32_t nw = nb <= 4u ? nb : 0u;
                const uint32_t t0 = nw > 0u ? G.newid[bp0] : 0u, t1 = nw > 1u ? G.newid[bp1] : 0u, t2 = nw > 2u ? G.newid[bp2] : 0u, t3 = nw > 3u ? G.newid[bp3] : 0u;
                const uint64_t at = emit_alloc_wave(S, W, nw);
                if (nw > 0u && at + 0u < S.cap) S.key[at + 0u] = u_self, S.val[at + 0u] = (uint64_t)t0 | ((uint64_t)bm0 << 32);
                if (nw > 1u && at + 1u < S.cap) S.key[at + 1u] = u_self, S.val[at + 1u] = (uint64_t)t1 | ((uint64_t)bm1 << 32);
                if (nw > 2u && at + 2u < S.cap) S.key[at + 2u] = u_self, S.val[at + 2u] = (uint64_t)t2 | ((uint64_t)bm2 << 32);
                if (nw > 3u && at + 3u < S.cap) S.key[at + 3u] = u_self, S.val[at + 3u] = (uint64_t)t3 | ((uint64_t)bm3 << 32);
                if (marker) emit_append(S, W, u_self, (uint64_t)u_self | ((uint64_t)marker_meta << 32));
            }
            const uint64_t hb = __ballot(heavy);
            if (hb) {  // (one atomic per wave)
                const int first = __ffsll((long long)hb) - 1;
                unsigned long long at = 0;
                if ((int)lane_id() == first) at = atomicAdd(heavy_n, (unsigned long long)__popcll(hb));
                at = __shfl(at, first);
                if (heavy) heavy_list[at + (uint32_t)__popcll(hb & lanemask_lt())] = (uint32_t)v;
            }
        }
    } else {
        const uint32_t lane = lane_id();
        const uint64_t total = *heavy_n, n_waves = (uint64_t)gridDim.x * (blockDim.x / 64u);
        for (uint64_t i = (uint64_t)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; i < total; i += n_waves) {
            const uint64_t v = heavy_list[i];
            const uint32_t u = G.newid[v];
            const bool marker = vertex_incomplete(G, inc_bands, inc_lds, v, u);  // (a poisoned vertex never comes here)
            const uint64_t rootp = G.vpos[v];
            const uint32_t rc = (uint32_t)(rootp >> 32), rr = (uint32_t)rootp;
            const uint32_t node = G.vnode[v];
            const uint32_t e_lo = G.nedge_off[node], e_hi = G.nedge_off[node + 1];
            for (uint32_t e = e_lo; e < e_hi; ++e) {
                const uint32_t to = G.eto[e];
                if (to == PAG_NONE) continue;
                uint32_t step, p0, q;
                edge_target(G, to, G.estep[e], &step, &p0, &q);
                const uint32_t entry = step < RATIO_TAB_N ? ratio_tab[step] : RATIO_TAB_NONE;
                for (uint32_t jb = 0; jb < q; jb += 64u) {
                    const uint32_t j = jb + lane;
                    if (j < q) {
                        const uint64_t pp = G.vpos[p0 + j];
                        uint32_t esim;
                        const int grade = d_check_position_any(rc, rr, (uint32_t)(pp >> 32), (uint32_t)pp, step, dev, err, entry, &esim);
                        if (grade != G_OOPS) emit_append(S, W, u, emit_value(G.newid[p0 + j], step, (uint32_t)grade, esim));
                    }
                }
            }
            if (marker && lane == 0u) emit_append(S, W, u, (uint64_t)u | ((uint64_t)marker_meta << 32));
        }
    }
    if (lane_id() == 0u && W->n) atomicAdd(S.n_real, (unsigned long long)W->n);
}

// the offset table from the sorted stream's keys: succ_off[u] = first record whose source is >= u, u = 0 .. n_pos.  Thread i
// writes the entries of the ids in (key[i - 1], key[i]] — nothing when its record continues a run; thread n_rec those above
// the last source.
__global__ void k_succ_offsets(const uint32_t *__restrict__ key, uint64_t n_rec, uint64_t n_pos, uint32_t *__restrict__ succ_off) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_rec; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hi = i < n_rec ? (uint64_t)key[i] : n_pos;
        const uint64_t lo = i ? (uint64_t)key[i - 1] + 1ull : 0ull;
        for (uint64_t u = lo; u <= hi; ++u) succ_off[u] = (uint32_t)i;
    }
}

// The sorted stream -> the records a walk reads.  One thread per record IN COORDINATE ORDER: the record learns its target's
// contig coordinate and its target's record range (offset + count clamped to 15 = "15 or more: look the range up"), so that a
// walk step never waits for succ_off or upos; the targets of neighbouring records are neighbours on the strand, so these reads
// hit the caches.  A poison / marker record names its own vertex and has no coordinate: neither a leap nor subject to the
// coordinate windows, the grade alone rejects it.
__global__ void k_succ_finish(TravGraph G, const uint64_t *__restrict__ val, uint64_t n_rec) {
    // four records per thread and trip, their loads issued together (one dependent gather each: latency-bound otherwise)
    const uint64_t T = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += 4 * T) {
        uint64_t x[4];
        uint32_t t0[4], t1[4], pc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t at = i + (uint64_t)q * T;
            x[q] = val[at < n_rec ? at : i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t tgt = (uint32_t)x[q];
            t0[q] = G.succ_off[tgt];
            t1[q] = G.succ_off[tgt + 1];
            pc[q] = (((uint32_t)(x[q] >> 56)) & 7u) >= GRADE_POISON_IF_LEAP ? 0u : (uint32_t)(G.upos[tgt] >> 32);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t at = i + (uint64_t)q * T;
            if (at >= n_rec) continue;
            const uint32_t tc = t1[q] - t0[q] < 15u ? t1[q] - t0[q] : 15u;
            SuccRec r;
            r.tgt = (uint32_t)x[q];
            r.pc = pc[q];
            r.meta = ((uint32_t)(x[q] >> 32) & 0x0FFFFFFFu) | (tc << 28);
            r.toff = t0[q];
            G.succ[at] = r;  // one 16-byte store
        }
    }
}

// pag_successors: one wave looks one vertex up in the coordinate order — the vertices without a contig coordinate come first,
// by reference coordinate; the others by contig coordinate; equal keys in k-mer-major order (trav_order) — and hands its records
// back in the caller's terms
struct SuccOut {
    uint32_t code, step;
    uint64_t pos;
    uint32_t grade, ctg_similar;
};
__global__ void k_successors_of(TravGraph G, uint32_t code, uint64_t pos, SuccOut *__restrict__ recs, uint64_t cap, unsigned long long *__restrict__ out) {
    const uint32_t lane = threadIdx.x;
    const bool zero = (pos >> 32) == 0;
    const uint64_t lo0 = zero ? 0 : G.n_zero, hi0 = zero ? G.n_zero : G.n_pos;
    const uint32_t want = zero ? (uint32_t)pos : (uint32_t)(pos >> 32);
    uint64_t lo = lo0, hi = hi0;  // first u of the stretch whose key is >= want
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t p = G.upos[mid];
        const uint32_t key = zero ? (uint32_t)p : (uint32_t)(p >> 32);
        if (key < want) lo = mid + 1;
        else hi = mid;
    }
    unsigned long long found = ~0ull;
    for (uint64_t base = lo; base < hi0; base += 64) {
        const uint64_t u = base + lane;
        bool same_key = false, hit = false;
        if (u < hi0) {
            const uint64_t p = G.upos[u];
            same_key = (zero ? (uint32_t)p : (uint32_t)(p >> 32)) == want;
            hit = p == pos && G.ncode[G.vnode[G.uold[u]]] == code;
        }
        const unsigned long long hits = __ballot(hit);
        if (hits) {
            found = base + (unsigned long long)__builtin_ctzll(hits);
            break;
        }
        if (__ballot(same_key) != ~0ull) break;  // (the run of this key ends inside these 64)
    }
    if (found == ~0ull) {
        if (lane == 0) out[0] = ~0ull;
        return;
    }
    const uint32_t a = G.succ_off[found], b = G.succ_off[found + 1];
    bool marker = false;
    for (uint32_t i = a + lane; i < b; i += 64) {
        const SuccRec r = G.succ[i];
        const uint32_t grade = (r.meta >> 24) & 7u;
        if (grade >= GRADE_POISON_IF_LEAP) {
            marker = true;
        } else if ((uint64_t)(i - a) < cap) {
            SuccOut o;
            o.code = G.ncode[G.vnode[G.uold[r.tgt]]];
            o.step = r.meta & 0xFFFFFFu;
            o.pos = G.upos[r.tgt];
            o.grade = grade;
            o.ctg_similar = (r.meta >> 27) & 1u;
            recs[i - a] = o;
        }
    }
    const bool any_marker = __ballot(marker) != 0ull;
    if (lane == 0) out[0] = any_marker ? ~1ull : (unsigned long long)(b - a);
}
int trav_successors_of(TravGraph G, uint32_t code, uint64_t pos, void *recs, uint64_t cap, unsigned long long *out, hipStream_t s) {
    static_assert(sizeof(SuccOut) == sizeof(pag_succ), "pag_succ layout");
    k_successors_of<<<dim3(1), dim3(64), 0, s>>>(G, code, pos, (SuccOut *)recs, cap, out);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
// The successor records of every vertex of G (see k_succ_emit): trav_succ_emit fills and sorts the emission stream — two
// (key, value) arrays of `cap` slots that it ping-pongs between, sort_tmp of sort_tmp_bytes(cap) bytes, counters: 4 device
// words, heavy_list: [n_pos] (null or heavy_limit 0: every vertex by its own thread) —, trav_succ_finish turns the sorted
// stream into G.succ_off / G.succ (allocated by the caller from *n_rec).  *n_slots receives the slots the waves took; when that
// is more than `cap` the stream was too small: *sorted_key is null, nothing else is valid, the caller comes back with room for
// *n_slots.
int trav_succ_emit(TravGraph G, uint32_t dev, double err, uint32_t *key0, uint64_t *val0, uint32_t *key1, uint64_t *val1, uint64_t cap, void *sort_tmp,
                   unsigned long long *counters, uint32_t *heavy_list, uint32_t heavy_limit, uint64_t *n_slots, uint64_t *n_rec, uint64_t *n_heavy,
                   const uint32_t **sorted_key, const uint64_t **sorted_val, hipStream_t s) {
    *n_slots = *n_rec = *n_heavy = 0;
    *sorted_key = key0;
    *sorted_val = val0;
    const uint64_t n = G.n_pos;
    if (!n) return PAG_OK;
    if (heavy_limit == 0) heavy_list = nullptr;
    PAG_HIP_TRY(hipMemsetAsync(key0, 0xFF, cap * 4, s));
    PAG_HIP_TRY(hipMemsetAsync(counters, 0, 4 * sizeof(unsigned long long), s));
    EmitStream S;
    S.key = key0;
    S.val = val0;
    S.cap = cap;
    S.cursor = counters;
    S.n_real = counters + 1;
    k_succ_emit<0, 4><<<dim3(std::min(grid_for(n), EMIT_GRID_THREADS)), dim3(256), 0, s>>>(G, dev, err, S, heavy_list, counters + 2, heavy_limit);
    if (heavy_list) k_succ_emit<1, 4><<<dim3(EMIT_GRID_WAVES), dim3(256), 0, s>>>(G, dev, err, S, heavy_list, counters + 2, heavy_limit);
    unsigned long long h[3] = {0, 0, 0};
    PAG_HIP_TRY(hipMemcpyAsync(h, counters, sizeof(h), hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    *n_slots = h[0];
    *n_rec = h[1];
    *n_heavy = h[2];
    if (h[0] > cap) {
        *sorted_key = nullptr;
        *sorted_val = nullptr;
        return PAG_OK;
    }
    // fillers are all ones: behind every id in the sorted bits as long as those can represent n itself
    int key_bits = 1;
    while (key_bits < 32 && (n >> key_bits) != 0) ++key_bits;
    int in0 = 1, rc;
    if ((rc = sort_pairs(key0, val0, key1, val1, h[0], key_bits, sort_tmp, &in0, s, nullptr, nullptr))) return rc;
    *sorted_key = in0 ? key0 : key1;
    *sorted_val = in0 ? val0 : val1;
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
int trav_succ_finish(TravGraph G, const uint32_t *key, const uint64_t *val, uint64_t n_rec, hipStream_t s) {
    if (!G.n_pos) return PAG_OK;
    k_succ_offsets<<<dim3(grid_for(n_rec + 1)), dim3(256), 0, s>>>(key, n_rec, G.n_pos, G.succ_off);
    if (n_rec) k_succ_finish<<<dim3(grid_for(n_rec)), dim3(256), 0, s>>>(G, val, n_rec);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

// test hook: the device match predicates on caller-supplied rows (tests/test_gpu_predicates.py feeds the reference's
// truth table tests/golden/func_predicate.txt.gz and a dense sweep around the 0.15 ratio boundary)
__global__ void k_debug_predicates(const uint32_t *__restrict__ rows, uint64_t n, double err, uint8_t *__restrict__ grade,
                                   uint8_t *__restrict__ edge_sim) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + 6 * i;  // a_ctg a_ref b_ctg b_ref dist dev
    uint32_t es = 0;
    grade[i] = (uint8_t)d_check_position(r[0], r[1], r[2], r[3], r[4], r[5], err, &es);
    edge_sim[i] = (uint8_t)es;
}
// ... the way the successor kernels evaluate them: ratio tests through the LDS table (d_ratio_entry) where the step has an entry
__global__ void k_debug_predicates_tab(const uint32_t *__restrict__ rows, uint64_t n, double err, uint8_t *__restrict__ grade,
                                       uint8_t *__restrict__ edge_sim, unsigned long long *__restrict__ n_tab) {
    __shared__ uint32_t ratio_tab[RATIO_TAB_N];
    d_ratio_table_fill(ratio_tab, err);
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *r = rows + 6 * i;
    uint32_t es = 0;
    const uint32_t entry = r[4] < RATIO_TAB_N ? ratio_tab[r[4]] : RATIO_TAB_NONE;
    if (entry != RATIO_TAB_NONE) atomicAdd(n_tab, 1ull);
    grade[i] = (uint8_t)d_check_position_any(r[0], r[1], r[2], r[3], r[4], r[5], err, entry, &es);
    edge_sim[i] = (uint8_t)es;
}


}  // namespace pagdev

static int debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device, uint64_t *n_through_table);
extern "C" int pag_debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device) {
    return debug_predicates(rows, n, err, grade, edge_sim, device, nullptr);
}
// the same rows through the ratio table of the successor kernels; *n_through_table: how many rows had a table entry
extern "C" int pag_debug_predicates_tab(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device,
                                        uint64_t *n_through_table) {
    if (!n_through_table) return PAG_EINVAL;
    *n_through_table = 0;
    return debug_predicates(rows, n, err, grade, edge_sim, device, n_through_table);
}
static int debug_predicates(const uint32_t *rows, uint64_t n, double err, uint8_t *grade, uint8_t *edge_sim, int device, uint64_t *n_through_table) {
    using namespace pagdev;
    if (!rows || !grade || !edge_sim) return PAG_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return PAG_ENODEV;
    uint32_t *d_rows = nullptr;
    uint8_t *d_out = nullptr;
    if (n == 0) return PAG_OK;
    PAG_HIP_TRY(hipMalloc((void **)&d_rows, n * 24));
    if (hipMalloc((void **)&d_out, 2 * n + 16) != hipSuccess) {
        hipFree(d_rows);
        return PAG_ENOMEM;
    }
    unsigned long long *d_cnt = (unsigned long long *)(d_out + ((2 * n + 7) & ~(uint64_t)7));
    hipError_t e = hipMemcpy(d_rows, rows, n * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_cnt, 0, 8);
    if (e == hipSuccess) {
        if (n_through_table) k_debug_predicates_tab<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d_rows, n, err, d_out, d_out + n, d_cnt);
        else k_debug_predicates<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(d_rows, n, err, d_out, d_out + n);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && n_through_table) e = hipMemcpy(n_through_table, d_cnt, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(grade, d_out, n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(edge_sim, d_out + n, n, hipMemcpyDeviceToHost);
    hipFree(d_rows);
    hipFree(d_out);
    if (e != hipSuccess) {
        set_error("pag_debug_predicates: %s", hipGetErrorString(e));
        return PAG_EFAULT;
    }
    return PAG_OK;
}
