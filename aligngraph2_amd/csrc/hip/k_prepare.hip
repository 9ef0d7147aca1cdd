// k_prepare.hip — pag_prepare: the per-block bookkeeping AROUND the two extraction passes, on the device.
//
// What the reference does on the host before / at the top of its hot loops, from alignment records as the parser leaves them
// (paths under PAGraph/src/tools/):
//   align/Aligner.cpp:32-56        mergeAlignInfHelper: per-query lists of alignments (both names known), database order,
//                                  then an unstable std::sort by score, descending
//   align/Aligner.tcc:40-71        parseToCtg: static filters (selected contig, (qEnd - qBegin) / readLen >= 0.35, contig
//                                  interval inside the contig with `>=` on the end: quirk Q14), flipPosition, the second
//                                  orientation of the `ii` loop (:73-96) for contigs selected in reverse
//   align/Aligner.tcc:121-152      parseToRef: accepted reference, ratio >= 0.10, flipPosition (the coverage filter is
//                                  dynamic: cov_filter kernels, util.hip)
//   align/Aligner.cpp:70-82        covInfHelper counts every record whose reference name is known, listed or not
//   align/Aligner.cpp:97-202       simpleAlign: contig -> reference alignments of the accepted reference and configured
//   align/AlignReference.cpp:41-79 orientation, walked forward; insert(): base cb + k receives (refIdx + 1, refPos) for
//                                  k < ce - cb; addExtraPosition(): empty lists get (0, 0)
//   position/PositionMapper.cpp:16-42   single coordinates
//   util/MultiThreadTools.tcc:8-14 thread-major strided emission order
//
// The result is a device-resident pag_build_input, array for array what the host restatement (tests/harness/graph_input.cpp,
// the former product code, now the checker of this stage) produces.  Two things stay on the host because they ARE libstdc++
// behaviour: the std::sort of a query's list when it has more than 16 entries (up to 16 std::sort is a stable insertion sort,
// which the device does; longer lists go through introsort, whose tie order only the same library reproduces), and the sort
// of the handful of contig -> reference lists.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cstring>
#include <vector>

#include "pag_graph_impl.hpp"

namespace pagdev {

namespace {

constexpr int PREP_SLOT0 = 128;  // pool slots of this stage
constexpr uint32_t LIST_INSERTION_MAX = 16;  // libstdc++ _S_threshold: std::sort of <= 16 elements is __insertion_sort

__device__ __forceinline__ void d_flip(uint64_t &left, uint64_t &right, uint64_t length) {  // Aligner::flipPosition (Aligner.cpp:235-239)
    const uint64_t tmp = left;
    left = length - right;
    right = length - tmp;
}

// ---- per-query lists -------------------------------------------------------------------------------------------------
__global__ void prep_keys(const pag_raw_aln *__restrict__ rec, uint64_t n, uint32_t n_queries, uint32_t *__restrict__ key,
                          uint64_t *__restrict__ val) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pag_raw_aln r = rec[i];
    key[i] = (r.query != PAG_NONE && r.target != PAG_NONE && r.query < n_queries) ? r.query : n_queries;
    val[i] = i;
}
// off[q] = first sorted position whose key is >= q (q = 0 .. n_queries: off[n_queries] = number of listed records)
__global__ void prep_offsets(const uint32_t *__restrict__ key, uint64_t n, uint32_t n_queries, uint64_t *__restrict__ off) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n_queries) return;
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (key[mid] < (uint32_t)q) lo = mid + 1;
        else hi = mid;
    }
    off[q] = lo;
}
// std::sort by score, descending, of every query's list: the stable insertion sort libstdc++ runs for <= 16 elements;
// longer lists are reported (host: the same std::sort)
__global__ void prep_sort_lists(const pag_raw_aln *__restrict__ rec, const uint64_t *__restrict__ off, uint32_t n_queries,
                                uint64_t *__restrict__ val, uint32_t *__restrict__ long_list, uint32_t *__restrict__ n_long, uint32_t long_cap) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const uint64_t a = off[q], b = off[q + 1];
    const uint64_t n = b - a;
    if (n < 2) return;
    if (n > LIST_INSERTION_MAX) {
        const uint32_t slot = atomicAdd(n_long, 1u);
        if (slot < long_cap) long_list[slot] = (uint32_t)q;
        return;
    }
    uint64_t sc[LIST_INSERTION_MAX], id[LIST_INSERTION_MAX];
    for (uint32_t i = 0; i < n; ++i) {
        id[i] = val[a + i];
        sc[i] = rec[id[i]].score;
    }
    for (uint32_t i = 1; i < n; ++i) {
        const uint64_t s = sc[i], v = id[i];
        uint32_t j = i;
        while (j > 0 && s > sc[j - 1]) {  // comp(val, prev) = val.score > prev.score
            sc[j] = sc[j - 1];
            id[j] = id[j - 1];
            --j;
        }
        sc[j] = s;
        id[j] = v;
    }
    for (uint32_t i = 0; i < n; ++i) val[a + i] = id[i];
}

struct PrepTables {
    const uint32_t *read_len;
    uint32_t n_reads;
    const uint32_t *ctg_len;
    const uint8_t *ctg_selected, *ctg_forward;
    uint32_t n_ctgs;
    const uint32_t *ref_len;
    const uint8_t *ref_accepted;
    uint32_t n_refs;
    double ratio;
};

__device__ __forceinline__ uint32_t d_col_class(const uint32_t *__restrict__ diff, uint64_t diff_off, uint64_t c) {
    return (diff[diff_off + (c >> 4)] >> ((c & 15u) * 2u)) & 3u;
}

// the head of parseToCtg for the sorted record at position p: the pag_aln record and whether it is listed
__global__ void prep_pass1(const pag_raw_aln *__restrict__ rec, const uint32_t *__restrict__ key, const uint64_t *__restrict__ val,
                           uint64_t n_listed, PrepTables T, const uint32_t *__restrict__ diff, pag_aln *__restrict__ out,
                           uint32_t *__restrict__ keep, uint32_t *__restrict__ err) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_listed) return;
    const pag_raw_aln a = rec[val[p]];
    const uint32_t r = key[p], c = a.target;
    keep[p] = 0;
    if (c >= T.n_ctgs || !T.ctg_selected[c]) return;
    const uint64_t readLen = T.read_len[r];
    uint64_t readBegin = a.q_begin, readEnd = a.q_end;
    if ((double)(readEnd - readBegin) * 1.0 / (double)readLen < T.ratio) return;
    uint64_t ctgBegin = a.t_begin, ctgEnd = a.t_end;
    const uint64_t ctgLen = T.ctg_len[c];
    if (ctgEnd >= ctgLen || ctgBegin >= ctgLen) return;  // (>= on the end: quirk Q14)
    bool isForward = a.forward != 0;
    if (!isForward) d_flip(readBegin, readEnd, readLen);
    bool walkBack = false;
    if (!T.ctg_forward[c]) {  // the ii == 1 iteration (Aligner.tcc:73-96)
        isForward = !isForward;
        d_flip(readBegin, readEnd, readLen);
        d_flip(ctgBegin, ctgEnd, ctgLen);
        walkBack = true;
    }
    pag_aln o;
    o.query = r;
    o.target = c;
    o.t_begin = (uint32_t)a.t_begin;  // (< ctgLen, checked above)
    o.t_end = (uint32_t)a.t_end;
    o.q_start = PAG_NONE;
    o.t_start = 0;
    o.n_cols = a.n_cols;
    o.n_valid = 0;
    o.diff_off = a.diff_off;
    o.flags = PAG_ALN_ELIGIBLE | (isForward ? 0u : PAG_ALN_REV_STRAND) | (walkBack ? PAG_ALN_WALK_BACK : 0u);
    o.reserved = 0;
    if (readBegin < readLen) {
        uint64_t nValid = a.n_emit < readLen - readBegin ? a.n_emit : readLen - readBegin;
        // bases whose contig coordinate falls off the contig have an empty list (AlignReference::query,
        // AlignReference.cpp:60-67): clipped exactly, by walking (rare: an alignment that overhangs the contig's end)
        if (ctgBegin + a.n_radv >= ctgLen) {
            uint64_t k = 0, t = ctgBegin, firstBad = nValid;
            bool found = false;
            for (uint64_t jj = 0; jj < a.n_cols && !found;) {
                // a whole word of sixteen columns at a time while the contig's end stays out of reach (a column at a time these few
                // records — alignments that overhang a contig's end — were the kernel: one thread walking ten thousand columns,
                // a dependent global load each; the classes' totals of a word do not depend on the order inside it)
                const uint64_t sc = walkBack ? a.n_cols - jj - 1 : jj;  // storage column of walk column jj
                const bool word_start = walkBack ? (sc & 15u) == 15u : (sc & 15u) == 0u;
                if (word_start && jj + 16 <= a.n_cols) {
                    const uint32_t w = diff[a.diff_off + (sc >> 4)];
                    const uint32_t qd = w & 0x55555555u, rd = (w >> 1) & 0x55555555u;
                    const uint32_t n1 = (uint32_t)__popc(qd & ~rd), n2 = (uint32_t)__popc(rd & ~qd);  // classes 01 (target only) and 10 (query only)
                    if (t + (16u - n2) < ctgLen) {  // every column of the word sees a target position before the end
                        k += 16u - n1;
                        t += 16u - n2;
                        jj += 16;
                        continue;
                    }
                }
                const uint32_t cls = d_col_class(diff, a.diff_off, sc);
                if (cls == 1u) {
                    ++t;
                } else {
                    if (t >= ctgLen) {
                        firstBad = k;
                        found = true;
                    }
                    ++k;
                    if (cls != 2u) ++t;
                }
                ++jj;
            }
            nValid = nValid < firstBad ? nValid : firstBad;
        }
        if (readBegin > 0xFFFFFFFFull || ctgBegin > 0xFFFFFFFFull) atomicOr(err, 1u);
        o.q_start = (uint32_t)readBegin;
        o.t_start = (uint32_t)ctgBegin;
        o.n_valid = (uint32_t)nValid;
    }
    out[p] = o;
    keep[p] = 1;
}

__device__ __forceinline__ void d_clamp_cov(const pag_raw_aln &a, uint64_t size, pag_aln &o) {
    // coverage loops run for (j = begin; j < end; ++j) { if (j >= size) break; ... } (Aligner.cpp:76-81, Aligner.tcc:142-145)
    uint64_t b = a.t_begin < size ? a.t_begin : size, e = a.t_end < size ? a.t_end : size;
    if (e < b) e = b;
    o.t_begin = (uint32_t)b;
    o.t_end = (uint32_t)e;
}

__global__ void prep_pass2(const pag_raw_aln *__restrict__ rec, const uint32_t *__restrict__ key, const uint64_t *__restrict__ val,
                           uint64_t n_listed, PrepTables T, pag_aln *__restrict__ out, uint32_t *__restrict__ keep,
                           uint8_t *__restrict__ in_list, uint32_t *__restrict__ err) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_listed) return;
    const uint64_t ri = val[p];
    const pag_raw_aln a = rec[ri];
    const uint32_t r = key[p], t = a.target;
    keep[p] = 0;
    if (t >= T.n_refs || !T.ref_accepted[t]) return;
    const uint64_t readLen = T.read_len[r];
    uint64_t readBegin = a.q_begin, readEnd = a.q_end;
    if ((double)(readEnd - readBegin) * 1.0 / (double)readLen < T.ratio) return;
    const bool isForward = a.forward != 0;
    if (!isForward) d_flip(readBegin, readEnd, readLen);
    pag_aln o;
    o.query = r;
    o.target = t;
    d_clamp_cov(a, T.ref_len[t], o);
    o.q_start = PAG_NONE;
    o.t_start = 0;
    o.n_cols = a.n_cols;
    o.n_valid = 0;
    o.diff_off = a.diff_off;
    o.flags = PAG_ALN_ELIGIBLE | (isForward ? 0u : PAG_ALN_REV_STRAND);
    o.reserved = 0;
    if (readBegin < readLen) {
        o.n_valid = (uint32_t)(a.n_emit < readLen - readBegin ? a.n_emit : readLen - readBegin);
        if (readBegin > 0xFFFFFFFFull || a.t_begin > 0xFFFFFFFFull) atomicOr(err, 1u);
        o.q_start = (uint32_t)readBegin;
        o.t_start = (uint32_t)a.t_begin;
    }
    out[p] = o;
    keep[p] = 1;
    in_list[ri] = 1;
}
// every other record whose reference name is known still counts for coverage (Aligner::covInfHelper ignores the query)
__global__ void prep_cov_only_flags(const pag_raw_aln *__restrict__ rec, uint64_t n, const uint8_t *__restrict__ in_list, uint32_t n_refs,
                                    uint32_t *__restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (!in_list[i] && rec[i].target != PAG_NONE && rec[i].target < n_refs) ? 1u : 0u;
}
__global__ void prep_cov_only(const pag_raw_aln *__restrict__ rec, uint64_t n, const uint32_t *__restrict__ flag, const uint64_t *__restrict__ pos,
                              const uint32_t *__restrict__ ref_len, pag_aln *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const pag_raw_aln a = rec[i];
    pag_aln o{};
    o.query = PAG_NONE;
    o.target = a.target;
    d_clamp_cov(a, ref_len[a.target], o);
    out[pos[i]] = o;
}
__global__ void prep_compact(const pag_aln *__restrict__ tmp, const uint32_t *__restrict__ keep, const uint64_t *__restrict__ pos, uint64_t n,
                             pag_aln *__restrict__ out) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || !keep[p]) return;
    out[pos[p]] = tmp[p];
}
// query_off of the compacted lists: where the first listed record of query q went (pos[n_listed] = number kept)
__global__ void prep_query_off(const uint64_t *__restrict__ off, const uint64_t *__restrict__ pos, uint32_t n_queries, uint64_t *__restrict__ qoff) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > n_queries) return;
    qoff[q] = pos[off[q]];
}
// thread-major strided order (MultiThreadTools.tcc:8-14 under the serialising shim): t = 0: 0, T, 2T ..; t = 1: ..
__global__ void prep_emit_order(uint32_t n, uint32_t T, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // reads with i % T == t come after all reads of smaller residues: #{j < n : j % T < t} + i / T
    const uint32_t t = i % T, full = n / T, rem = n % T;
    const uint32_t before = t * full + (t < rem ? t : rem);
    out[before + i / T] = i;
}

// ---- contig -> reference map -----------------------------------------------------------------------------------------
struct CtgMapJob {  // one listed contig -> reference alignment of a selected contig
    uint64_t cb, span;      // first contig base (after the flip of a reverse alignment), ce - cb
    uint64_t ref_begin;
    uint64_t diff_off;
    uint64_t map_off;       // first offset slot of the contig
    uint64_t chunk0;        // first entry of this job in the chunk prefix array
    uint32_t n_cols, len;   // alignment columns, contig length
    uint32_t ref_single_base;
    uint32_t ctg, rank, alone;  // contig; position in the contig's list; 1: the contig's ONLY listed alignment (no base of it is touched twice)
};
constexpr uint32_t MAP_CHUNK = 1024;  // columns per chunk: 64 lanes x 16

__device__ __forceinline__ void d_cols16(const uint32_t *__restrict__ diff, uint64_t diff_off, uint32_t n_cols, uint32_t j0, uint32_t *emitb,
                                         uint32_t *radvb) {
    *emitb = *radvb = 0;
    if (j0 >= n_cols) return;
    const uint32_t n = n_cols - j0 > 16u ? 16u : n_cols - j0;
    const uint32_t bits = diff[diff_off + (j0 >> 4)];
    const uint32_t valid = n >= 16 ? 0x55555555u : (((1u << (2 * n)) - 1u) & 0x55555555u);
    const uint32_t qd = bits & 0x55555555u, rd = (bits >> 1) & 0x55555555u;
    *emitb = ~(qd & ~rd) & valid;  // class != 01: the query base is emitted
    *radvb = ~(rd & ~qd) & valid;  // class != 10: the target advances
}
// per job and 1024-column chunk: emitted columns / target advances before the chunk (one wave per job)
__global__ __launch_bounds__(64) void ctgmap_chunks(const CtgMapJob *__restrict__ jobs, uint32_t n_jobs, const uint32_t *__restrict__ diff,
                                                    uint2 *__restrict__ pre) {
    if (blockIdx.x >= n_jobs) return;
    const CtgMapJob J = jobs[blockIdx.x];
    const uint32_t n_chunks = (J.n_cols + MAP_CHUNK - 1) / MAP_CHUNK;
    uint32_t e = 0, r = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        if (lane_id() == 0) pre[J.chunk0 + c] = make_uint2(e, r);
        uint32_t eb, rb;
        d_cols16(diff, J.diff_off, J.n_cols, c * MAP_CHUNK + lane_id() * 16, &eb, &rb);
        e += wave_sum(__popc(eb));
        r += wave_sum(__popc(rb));
    }
}
// AlignReference::insert over one chunk of one alignment.  FILL = false: entries per base are counted (atomics: several
// alignments of a contig may cover a base); FILL = true: the jobs of ONE list rank are written, base by base at the
// base's cursor — inside a rank every base is touched by at most one alignment, and the ranks are launched in order, so
// the entries of a base come out in list order without atomics.
template <bool FILL>
__global__ __launch_bounds__(64) void ctgmap_walk(const CtgMapJob *__restrict__ jobs, uint32_t n_jobs, const uint64_t *__restrict__ chunk_first,
                                                  uint64_t n_chunks_total, uint32_t rank, const uint32_t *__restrict__ diff,
                                                  const uint2 *__restrict__ pre, uint32_t *__restrict__ cnt, uint32_t *__restrict__ multi,
                                                  const uint32_t *__restrict__ ent_off, uint32_t *__restrict__ ent) {
    const uint64_t gc = blockIdx.x;
    if (gc >= n_chunks_total) return;
    // the job this chunk belongs to (chunk_first ascending, one entry per job + the total)
    uint32_t lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunk_first[mid] <= gc) lo = mid;
        else hi = mid;
    }
    const CtgMapJob J = jobs[lo];
    if (FILL && J.rank != rank) return;
    const uint32_t c = (uint32_t)(gc - J.chunk0);
    const uint2 p0 = pre[gc];
    uint32_t eb, rb;
    d_cols16(diff, J.diff_off, J.n_cols, c * MAP_CHUNK + lane_id() * 16, &eb, &rb);
    uint32_t tot;
    const uint64_t eex = (uint64_t)wave_excl_sum(__popc(eb), &tot) + p0.x;
    const uint64_t rex = (uint64_t)wave_excl_sum(__popc(rb), &tot) + p0.y;
    uint32_t m = eb;
    while (m) {
        const uint32_t bit = __ffs(m) - 1;
        m &= m - 1;
        const uint32_t below = (1u << bit) - 1u;
        const uint64_t k = eex + __popc(eb & below);
        const uint64_t ref_cur = J.ref_begin + rex + __popc(rb & below);
        const uint64_t b = J.cb + k;
        if (k < J.span && b < J.len) {
            if (J.alone) {
                // (the contig's only alignment: every base gets this one entry — a plain store where the general case pays a global
                // atomic per column, 55 M of them at BASELINE configs[1], and a cursor per base in the fill)
                if (!FILL) cnt[J.map_off + b] = 1u;
                else ent[ent_off[J.map_off + b]] = (uint32_t)((uint64_t)J.ref_single_base + ref_cur);
            } else if (!FILL) {
                if (atomicAdd(&cnt[J.map_off + b], 1u) >= 1u) multi[J.ctg] = 1u;
            } else {
                const uint32_t at = cnt[J.map_off + b];
                cnt[J.map_off + b] = at + 1;
                ent[ent_off[J.map_off + b] + at] = (uint32_t)((uint64_t)J.ref_single_base + ref_cur);
            }
        }
    }
}
// entries per offset slot: max(count, 1) for a base (addExtraPosition), 0 for the slot behind a contig's last base
__global__ void ctgmap_runs(const uint32_t *__restrict__ cnt, const uint8_t *__restrict__ is_end, uint64_t n, uint32_t *__restrict__ run) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    run[i] = is_end[i] ? 0u : (cnt[i] ? cnt[i] : 1u);
}
__global__ void prep_narrow(const uint64_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (uint32_t)in[i];
}
__global__ void ctgmap_mark_ends(const uint64_t *__restrict__ end_slot, uint32_t n, uint8_t *__restrict__ is_end) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) is_end[end_slot[i]] = 1;
}
__global__ void ctgmap_set_multi(pag_ctg *__restrict__ tab, const uint32_t *__restrict__ multi, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i].multi = multi[i] ? 1u : 0u;
}

unsigned blocks_for(uint64_t n, unsigned per = 256) { return (unsigned)((n + per - 1) / per) + (n == 0); }

// PositionMapper (position/PositionMapper.cpp:16-31): start[0] = len[0], start[i] = start[i-1] + 3 len[i-1] + max(len[i-1], len[i]),
// extra end = + 4 len[last]
std::vector<uint64_t> mapper_starts(const uint32_t *len, uint64_t n) {
    std::vector<uint64_t> st;
    if (!n) return st;
    st.push_back(len[0]);
    for (uint64_t i = 1; i < n; ++i) st.push_back(st.back() + 3ull * len[i - 1] + std::max<uint64_t>(len[i - 1], len[i]));
    st.push_back(st.back() + 4ull * len[n - 1]);
    return st;
}

// the lists of one read database -> compacted pag_aln records + query_off (device)
int prepare_read_db(pag_graph *g, int pass, const pag_raw_db &db, const uint32_t *d_diff, PrepTables T, int slot0, pag_aln_db *out,
                    uint32_t *d_err) {
    hipStream_t s = g->stream;
    const uint64_t n = db.n;
    const uint32_t nq = T.n_reads;
    int slot = slot0;
    DevBuf b_rec(g, slot++), b_k0(g, slot++), b_v0(g, slot++), b_k1(g, slot++), b_v1(g, slot++), b_tmp(g, slot++), b_off(g, slot++),
        b_long(g, slot++), b_alnt(g, slot++), b_keep(g, slot++), b_pos(g, slot++), b_aln(g, slot++), b_qoff(g, slot++), b_inl(g, slot++),
        b_pos2(g, slot++);
    int rc;
    if ((rc = b_rec.alloc((n + 1) * sizeof(pag_raw_aln))) || (rc = b_k0.alloc((n + 1) * 4)) || (rc = b_v0.alloc((n + 1) * 8)) ||
        (rc = b_k1.alloc((n + 1) * 4)) || (rc = b_v1.alloc((n + 1) * 8)) ||
        (rc = b_tmp.alloc(std::max(sort_tmp_bytes(n + 1), scan_tmp_bytes(n + 2) + 64))) || (rc = b_off.alloc(((uint64_t)nq + 2) * 8)) ||
        (rc = b_long.alloc(((uint64_t)nq + 2) * 4)) || (rc = b_alnt.alloc((n + 1) * sizeof(pag_aln))) || (rc = b_keep.alloc((n + 2) * 4)) ||
        (rc = b_pos.alloc((n + 3) * 8)) || (rc = b_aln.alloc((n + 1) * sizeof(pag_aln))) || (rc = b_qoff.alloc(((uint64_t)nq + 2) * 8)) ||
        (rc = b_inl.alloc(n + 16)) || (rc = b_pos2.alloc((n + 3) * 8)))
        return rc;
    if (n) PAG_HIP_TRY(hipMemcpyAsync(b_rec.p, db.rec, n * sizeof(pag_raw_aln), hipMemcpyHostToDevice, s));
    const pag_raw_aln *d_rec = b_rec.as<pag_raw_aln>();
    uint64_t n_listed = 0;
    uint32_t *key = b_k0.as<uint32_t>();
    uint64_t *val = b_v0.as<uint64_t>();
    if (n) {
        prep_keys<<<dim3(blocks_for(n)), dim3(256), 0, s>>>(d_rec, n, nq, key, val);
        int bits = 1;
        while (bits < 32 && ((uint64_t)nq >> bits) != 0) ++bits;
        int in0 = 1;
        if ((rc = sort_pairs(b_k0.as<uint32_t>(), b_v0.as<uint64_t>(), b_k1.as<uint32_t>(), b_v1.as<uint64_t>(), n, bits, b_tmp.p, &in0, s, nullptr, nullptr)))
            return rc;
        if (!in0) {
            key = b_k1.as<uint32_t>();
            val = b_v1.as<uint64_t>();
        }
    }
    prep_offsets<<<dim3(blocks_for((uint64_t)nq + 1)), dim3(256), 0, s>>>(key, n, nq, b_off.as<uint64_t>());
    uint32_t *d_nlong = b_long.as<uint32_t>();
    PAG_HIP_TRY(hipMemsetAsync(d_nlong, 0, 4, s));
    if (nq) prep_sort_lists<<<dim3(blocks_for(nq)), dim3(256), 0, s>>>(d_rec, b_off.as<uint64_t>(), nq, val, d_nlong + 1, d_nlong, nq);
    uint32_t n_long = 0;
    PAG_HIP_TRY(hipMemcpyAsync(&n_long, d_nlong, 4, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipMemcpyAsync(&n_listed, b_off.as<uint64_t>() + nq, 8, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    if (n_long) {
        // lists of more than 16 alignments: std::sort itself (introsort: its order among equal scores is the library's — the
        // host's libstdc++ has to be the one the reference was built with).  A list arrives in database order
        // (the device sort by query is stable), exactly mergeAlignInfHelper's input.  All of them in ONE round trip: the
        // offset table and the id array come down once, the lists are sorted by a pool of host threads, the id array goes
        // back once (a repeat-rich read set has tens of thousands of such lists: three blocking copies per list, as until
        // round 4, would have cost seconds).
        std::vector<uint32_t> qs(n_long);
        PAG_HIP_TRY(hipMemcpy(qs.data(), d_nlong + 1, (size_t)n_long * 4, hipMemcpyDeviceToHost));
        std::vector<uint64_t> offs((size_t)nq + 1), ids((size_t)n_listed);
        PAG_HIP_TRY(hipMemcpy(offs.data(), b_off.as<uint64_t>(), ((size_t)nq + 1) * 8, hipMemcpyDeviceToHost));
        if (n_listed) PAG_HIP_TRY(hipMemcpy(ids.data(), val, (size_t)n_listed * 8, hipMemcpyDeviceToHost));
        struct ListEntry {
            uint64_t score, rec;
        };
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            std::vector<ListEntry> list;
            for (size_t x; (x = next.fetch_add(1)) < qs.size();) {
                const uint64_t a0 = offs[qs[x]], a1 = offs[(size_t)qs[x] + 1];
                const size_t m = (size_t)(a1 - a0);
                list.resize(m);
                for (size_t i = 0; i < m; ++i) list[i] = ListEntry{db.rec[ids[a0 + i]].score, ids[a0 + i]};
                std::sort(list.begin(), list.end(), [](const ListEntry &l, const ListEntry &r) { return l.score > r.score; });
                for (size_t i = 0; i < m; ++i) ids[a0 + i] = list[i].rec;
            }
        };
        const unsigned nthr = (unsigned)std::min<size_t>(qs.size(), std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nthr; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (n_listed) PAG_HIP_TRY(hipMemcpy(val, ids.data(), (size_t)n_listed * 8, hipMemcpyHostToDevice));
    }
    // filters, flips, n_valid; compaction of the listed records
    uint64_t n_kept = 0, n_cov = 0;
    PAG_HIP_TRY(hipMemsetAsync(b_keep.p, 0, (n + 2) * 4, s));
    if (pass == 1) PAG_HIP_TRY(hipMemsetAsync(b_inl.p, 0, n + 16, s));
    if (n_listed) {
        if (pass == 0)
            prep_pass1<<<dim3(blocks_for(n_listed)), dim3(256), 0, s>>>(d_rec, key, val, n_listed, T, d_diff, b_alnt.as<pag_aln>(), b_keep.as<uint32_t>(), d_err);
        else
            prep_pass2<<<dim3(blocks_for(n_listed)), dim3(256), 0, s>>>(d_rec, key, val, n_listed, T, b_alnt.as<pag_aln>(), b_keep.as<uint32_t>(),
                                                                         b_inl.as<uint8_t>(), d_err);
    }
    if ((rc = scan_u32_to_u64(b_keep.as<uint32_t>(), b_pos.as<uint64_t>(), n_listed + 1, nullptr, b_tmp.p, s))) return rc;
    if (n_listed) prep_compact<<<dim3(blocks_for(n_listed)), dim3(256), 0, s>>>(b_alnt.as<pag_aln>(), b_keep.as<uint32_t>(), b_pos.as<uint64_t>(), n_listed, b_aln.as<pag_aln>());
    prep_query_off<<<dim3(blocks_for((uint64_t)nq + 1)), dim3(256), 0, s>>>(b_off.as<uint64_t>(), b_pos.as<uint64_t>(), nq, b_qoff.as<uint64_t>());
    PAG_HIP_TRY(hipMemcpyAsync(&n_kept, b_pos.as<uint64_t>() + n_listed, 8, hipMemcpyDeviceToHost, s));
    if (pass == 1 && n) {
        prep_cov_only_flags<<<dim3(blocks_for(n)), dim3(256), 0, s>>>(d_rec, n, b_inl.as<uint8_t>(), T.n_refs, b_k0.as<uint32_t>() == key ? b_k1.as<uint32_t>() : b_k0.as<uint32_t>());
        uint32_t *flag = b_k0.as<uint32_t>() == key ? b_k1.as<uint32_t>() : b_k0.as<uint32_t>();
        PAG_HIP_TRY(hipMemsetAsync(flag + n, 0, 4, s));
        if ((rc = scan_u32_to_u64(flag, b_pos2.as<uint64_t>(), n + 1, nullptr, b_tmp.p, s))) return rc;
        PAG_HIP_TRY(hipMemcpyAsync(&n_cov, b_pos2.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        prep_cov_only<<<dim3(blocks_for(n)), dim3(256), 0, s>>>(d_rec, n, flag, b_pos2.as<uint64_t>(), T.ref_len, b_aln.as<pag_aln>() + n_kept);
    } else {
        PAG_HIP_TRY(hipStreamSynchronize(s));
    }
    PAG_HIP_TRY(hipGetLastError());
    out->n_aln = n_kept + n_cov;
    out->aln = b_aln.as<pag_aln>();
    out->query_off = b_qoff.as<uint64_t>();
    out->diff = d_diff;
    out->n_diff_words = db.n_diff_words;
    return PAG_OK;
}

}  // namespace
}  // namespace pagdev

using namespace pagdev;

extern "C" int pag_prepare(pag_graph *g, const pag_raw_input *raw, pag_build_input *out) {
    if (!g || !raw || !out) return PAG_EINVAL;
    PAG_HIP_TRY(hipSetDevice(g->device));
    hipStream_t s = g->stream;
    const bool bulk_dev = raw->bulk_on_device != 0;
    const uint64_t n_reads = raw->reads.n_seqs, n_ctgs = raw->n_ctgs, n_refs = raw->n_refs;
    if (n_reads >= 0xFFFFFFFFull || n_ctgs >= 0x7FFFFFFFull || n_refs >= 0x7FFFFFFFull) {
        set_error("pag_prepare: too many sequences");
        return PAG_EINVAL;
    }
    // single-coordinate spaces (PositionMapper over ALL contigs / references of the files)
    const std::vector<uint64_t> cst = mapper_starts(raw->ctg_len, n_ctgs), rst = mapper_starts(raw->ref_len, n_refs);
    if ((!cst.empty() && cst.back() >= 0xFFFFFFFFull) || (!rst.empty() && rst.back() >= 0xFFFFFFFFull)) {
        set_error("coordinate space exceeds 32 bits (the reference truncates silently, PositionProcessor.cpp:48-51); split the input per reference sequence");
        return PAG_EINVAL;
    }
    int rc;
    int slot = PREP_SLOT0;
    DevBuf b_roff(g, slot++), b_rlen(g, slot++), b_packed(g, slot++), b_d1(g, slot++), b_d2(g, slot++), b_d3(g, slot++), b_clen(g, slot++),
        b_csel(g, slot++), b_cfwd(g, slot++), b_rflen(g, slot++), b_racc(g, slot++), b_ctab(g, slot++), b_rtab(g, slot++), b_order(g, slot++),
        b_err(g, slot++), b_jobs(g, slot++), b_cfirst(g, slot++), b_pre(g, slot++), b_cnt(g, slot++), b_multi(g, slot++), b_isend(g, slot++),
        b_run(g, slot++), b_scan(g, slot++), b_stmp(g, slot++), b_eoff(g, slot++), b_ent(g, slot++), b_ends(g, slot++);
    const int SLOT_DB1 = slot, SLOT_DB2 = slot + 16;
    auto put = [&](DevBuf &b, const void *src, size_t bytes, bool on_dev, const void **dst) -> int {
        if (on_dev) {
            *dst = src;
            return PAG_OK;
        }
        int r = b.alloc(bytes + 64);
        if (r) return r;
        if (bytes) PAG_HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s));
        *dst = b.p;
        return PAG_OK;
    };
    const void *d_roff, *d_rlen, *d_packed, *d_d1, *d_d2, *d_d3, *d_clen, *d_csel, *d_cfwd, *d_rflen, *d_racc;
    if ((rc = put(b_roff, raw->reads.byte_off, n_reads * 8, bulk_dev, &d_roff)) || (rc = put(b_rlen, raw->reads.len, n_reads * 4, bulk_dev, &d_rlen)) ||
        (rc = put(b_packed, raw->reads.packed, raw->reads.packed_bytes, bulk_dev, &d_packed)) ||
        (rc = put(b_d1, raw->read_to_ctg.diff, raw->read_to_ctg.n_diff_words * 4, bulk_dev, &d_d1)) ||
        (rc = put(b_d2, raw->read_to_ref.diff, raw->read_to_ref.n_diff_words * 4, bulk_dev, &d_d2)) ||
        (rc = put(b_d3, raw->ctg_to_ref.diff, raw->ctg_to_ref.n_diff_words * 4, bulk_dev, &d_d3)) ||
        (rc = put(b_clen, raw->ctg_len, n_ctgs * 4, false, &d_clen)) || (rc = put(b_csel, raw->ctg_selected, n_ctgs, false, &d_csel)) ||
        (rc = put(b_cfwd, raw->ctg_forward, n_ctgs, false, &d_cfwd)) || (rc = put(b_rflen, raw->ref_len, n_refs * 4, false, &d_rflen)) ||
        (rc = put(b_racc, raw->ref_accepted, n_refs, false, &d_racc)))
        return rc;
    if ((rc = b_err.alloc(64))) return rc;
    PAG_HIP_TRY(hipMemsetAsync(b_err.p, 0, 64, s));

    // ---- contig table + contig -> reference map
    std::vector<pag_ctg> ctab((size_t)n_ctgs);
    std::vector<pag_ref> rtab((size_t)n_refs);
    for (uint64_t i = 0; i < n_refs; ++i) {
        rtab[i].len = raw->ref_len[i];
        rtab[i].accepted = raw->ref_accepted[i] ? 1u : 0u;
        rtab[i].single_base = (uint32_t)rst[i];
        rtab[i].reserved = 0;
    }
    uint64_t off_total = 0;
    std::vector<uint64_t> end_slots;
    for (uint64_t c = 0; c < n_ctgs; ++c) {
        pag_ctg &t = ctab[c];
        const uint64_t len = raw->ctg_len[c];
        t.len = (uint32_t)len;
        t.selected = raw->ctg_selected[c] ? 1u : 0u;
        t.single_base = (uint32_t)(raw->ctg_forward[c] ? cst[c] : cst[c] + 2 * len);  // dualToSingle(+-(c + 1), 0)
        t.multi = 0;
        t.map_off = off_total;
        if (!t.selected) continue;
        off_total += len + 1;
        end_slots.push_back(off_total - 1);
    }
    // per-contig lists (mergeAlignInfHelper on the contig -> reference database: a handful of records, host std::sort)
    std::vector<CtgMapJob> jobs;
    std::vector<uint64_t> chunk_first;
    uint64_t n_chunks = 0;
    uint32_t max_rank = 0;
    {
        struct ListEntry {
            uint64_t score, rec;
        };
        std::vector<std::vector<ListEntry>> lists((size_t)n_ctgs);
        for (uint64_t i = 0; i < raw->ctg_to_ref.n; ++i) {
            const pag_raw_aln &r = raw->ctg_to_ref.rec[i];
            if (r.query != PAG_NONE && r.query < n_ctgs && r.target != PAG_NONE && r.target < n_refs) lists[r.query].push_back(ListEntry{r.score, i});
        }
        for (uint64_t c = 0; c < n_ctgs; ++c) {
            auto &l = lists[c];
            std::sort(l.begin(), l.end(), [](const ListEntry &a, const ListEntry &b) { return a.score > b.score; });
            if (!raw->ctg_selected[c]) continue;
            uint32_t rank = 0;
            const size_t first_job = jobs.size();
            for (const ListEntry &e : l) {
                const pag_raw_aln &r = raw->ctg_to_ref.rec[e.rec];
                if (!raw->ref_accepted[r.target]) continue;
                if ((r.forward != 0) != (raw->ctg_forward[c] != 0)) continue;
                uint64_t cb = r.q_begin, ce = r.q_end;
                const uint64_t len = raw->ctg_len[c];
                if (!r.forward) {  // flip (Aligner.cpp:235-239)
                    const uint64_t tmp = cb;
                    cb = len - ce;
                    ce = len - tmp;
                }
                CtgMapJob J{};
                J.cb = cb;
                J.span = ce > cb ? ce - cb : 0;
                J.ref_begin = r.t_begin;
                J.diff_off = r.diff_off;
                J.map_off = ctab[c].map_off;
                J.chunk0 = n_chunks;
                J.n_cols = r.n_cols;
                J.len = (uint32_t)len;
                J.ref_single_base = rtab[r.target].single_base;
                J.ctg = (uint32_t)c;
                J.rank = rank++;
                if (!J.n_cols) continue;
                chunk_first.push_back(n_chunks);
                n_chunks += (J.n_cols + MAP_CHUNK - 1) / MAP_CHUNK;
                jobs.push_back(J);
                max_rank = std::max(max_rank, J.rank + 1);
            }
            // (rank counts every listed alignment of the contig, also those without columns: alone = it was the only one)
            if (rank == 1 && jobs.size() == first_job + 1) jobs.back().alone = 1u;
        }
        chunk_first.push_back(n_chunks);
    }
    if ((rc = b_cnt.alloc((off_total + 2) * 4)) || (rc = b_multi.alloc((n_ctgs + 1) * 4)) || (rc = b_isend.alloc(off_total + 16)) ||
        (rc = b_run.alloc((off_total + 2) * 4)) || (rc = b_scan.alloc((off_total + 3) * 8)) || (rc = b_stmp.alloc(scan_tmp_bytes(off_total + 2) + 64)) ||
        (rc = b_eoff.alloc((off_total + 2) * 4)) || (rc = b_ends.alloc((end_slots.size() + 1) * 8)) || (rc = b_ctab.alloc((n_ctgs + 1) * sizeof(pag_ctg))) ||
        (rc = b_rtab.alloc((n_refs + 1) * sizeof(pag_ref))))
        return rc;
    PAG_HIP_TRY(hipMemsetAsync(b_cnt.p, 0, (off_total + 2) * 4, s));
    PAG_HIP_TRY(hipMemsetAsync(b_multi.p, 0, (n_ctgs + 1) * 4, s));
    PAG_HIP_TRY(hipMemsetAsync(b_isend.p, 0, off_total + 16, s));
    if (n_ctgs) PAG_HIP_TRY(hipMemcpyAsync(b_ctab.p, ctab.data(), n_ctgs * sizeof(pag_ctg), hipMemcpyHostToDevice, s));
    if (n_refs) PAG_HIP_TRY(hipMemcpyAsync(b_rtab.p, rtab.data(), n_refs * sizeof(pag_ref), hipMemcpyHostToDevice, s));
    if (!end_slots.empty()) {
        PAG_HIP_TRY(hipMemcpyAsync(b_ends.p, end_slots.data(), end_slots.size() * 8, hipMemcpyHostToDevice, s));
        ctgmap_mark_ends<<<dim3(blocks_for(end_slots.size())), dim3(256), 0, s>>>(b_ends.as<uint64_t>(), (uint32_t)end_slots.size(), b_isend.as<uint8_t>());
    }
    const uint32_t n_jobs = (uint32_t)jobs.size();
    if (n_jobs) {
        if ((rc = b_jobs.alloc(jobs.size() * sizeof(CtgMapJob))) || (rc = b_cfirst.alloc(chunk_first.size() * 8)) || (rc = b_pre.alloc((n_chunks + 1) * sizeof(uint2))))
            return rc;
        PAG_HIP_TRY(hipMemcpyAsync(b_jobs.p, jobs.data(), jobs.size() * sizeof(CtgMapJob), hipMemcpyHostToDevice, s));
        PAG_HIP_TRY(hipMemcpyAsync(b_cfirst.p, chunk_first.data(), chunk_first.size() * 8, hipMemcpyHostToDevice, s));
        ctgmap_chunks<<<dim3(n_jobs), dim3(64), 0, s>>>(b_jobs.as<CtgMapJob>(), n_jobs, (const uint32_t *)d_d3, b_pre.as<uint2>());
        ctgmap_walk<false><<<dim3((unsigned)n_chunks), dim3(64), 0, s>>>(b_jobs.as<CtgMapJob>(), n_jobs, b_cfirst.as<uint64_t>(), n_chunks, 0u, (const uint32_t *)d_d3,
                                                                        b_pre.as<uint2>(), b_cnt.as<uint32_t>(), b_multi.as<uint32_t>(), nullptr, nullptr);
    }
    uint64_t ent_total = 0;
    if (off_total) {
        ctgmap_runs<<<dim3(blocks_for(off_total)), dim3(256), 0, s>>>(b_cnt.as<uint32_t>(), b_isend.as<uint8_t>(), off_total, b_run.as<uint32_t>());
        PAG_HIP_TRY(hipMemsetAsync(b_run.as<uint32_t>() + off_total, 0, 4, s));
        if ((rc = scan_u32_to_u64(b_run.as<uint32_t>(), b_scan.as<uint64_t>(), off_total + 1, nullptr, b_stmp.p, s))) return rc;
        prep_narrow<<<dim3(blocks_for(off_total + 1)), dim3(256), 0, s>>>(b_scan.as<uint64_t>(), off_total + 1, b_eoff.as<uint32_t>());
        PAG_HIP_TRY(hipMemcpyAsync(&ent_total, b_scan.as<uint64_t>() + off_total, 8, hipMemcpyDeviceToHost, s));
        PAG_HIP_TRY(hipStreamSynchronize(s));
        if (ent_total > 0xFFFFFFFFull) {
            set_error("value does not fit 32 bits: contig map offset");
            return PAG_EINVAL;
        }
    }
    if ((rc = b_ent.alloc((ent_total + 1) * 4))) return rc;
    PAG_HIP_TRY(hipMemsetAsync(b_ent.p, 0, (ent_total + 1) * 4, s));
    if (n_jobs) {
        PAG_HIP_TRY(hipMemsetAsync(b_cnt.p, 0, (off_total + 2) * 4, s));  // (now the per-base cursors of the fill)
        for (uint32_t rank = 0; rank < max_rank; ++rank)
            ctgmap_walk<true><<<dim3((unsigned)n_chunks), dim3(64), 0, s>>>(b_jobs.as<CtgMapJob>(), n_jobs, b_cfirst.as<uint64_t>(), n_chunks, rank, (const uint32_t *)d_d3,
                                                                           b_pre.as<uint2>(), b_cnt.as<uint32_t>(), nullptr, b_eoff.as<uint32_t>(), b_ent.as<uint32_t>());
        ctgmap_set_multi<<<dim3(blocks_for(n_ctgs)), dim3(256), 0, s>>>(b_ctab.as<pag_ctg>(), b_multi.as<uint32_t>(), (uint32_t)n_ctgs);
    }

    // ---- the two read databases
    PrepTables T{};
    T.read_len = (const uint32_t *)d_rlen;
    T.n_reads = (uint32_t)n_reads;
    T.ctg_len = (const uint32_t *)d_clen;
    T.ctg_selected = (const uint8_t *)d_csel;
    T.ctg_forward = (const uint8_t *)d_cfwd;
    T.n_ctgs = (uint32_t)n_ctgs;
    T.ref_len = (const uint32_t *)d_rflen;
    T.ref_accepted = (const uint8_t *)d_racc;
    T.n_refs = (uint32_t)n_refs;
    pag_build_input o{};
    T.ratio = raw->read_to_ctg_ratio;
    if ((rc = prepare_read_db(g, 0, raw->read_to_ctg, (const uint32_t *)d_d1, T, SLOT_DB1, &o.read_to_ctg, b_err.as<uint32_t>()))) return rc;
    T.ratio = raw->read_to_ref_ratio;
    if ((rc = prepare_read_db(g, 1, raw->read_to_ref, (const uint32_t *)d_d2, T, SLOT_DB2, &o.read_to_ref, b_err.as<uint32_t>()))) return rc;

    // ---- emission order
    if ((rc = b_order.alloc((n_reads + 1) * 4))) return rc;
    const uint32_t Tn = raw->n_threads ? raw->n_threads : 1u;
    if (n_reads) prep_emit_order<<<dim3(blocks_for(n_reads)), dim3(256), 0, s>>>((uint32_t)n_reads, Tn, b_order.as<uint32_t>());
    uint32_t err = 0;
    PAG_HIP_TRY(hipMemcpyAsync(&err, b_err.p, 4, hipMemcpyDeviceToHost, s));
    PAG_HIP_TRY(hipStreamSynchronize(s));
    PAG_HIP_TRY(hipGetLastError());
    if (err) {
        set_error("value does not fit 32 bits: alignment coordinate");
        return PAG_EINVAL;
    }
    o.on_device = 1;
    o.n_threads = raw->n_threads;
    o.reads.n_seqs = n_reads;
    o.reads.byte_off = (const uint64_t *)d_roff;
    o.reads.len = (const uint32_t *)d_rlen;
    o.reads.packed = (const uint8_t *)d_packed;
    o.reads.packed_bytes = raw->reads.packed_bytes;
    o.emit_order = b_order.as<uint32_t>();
    o.n_ctgs = n_ctgs;
    o.ctgs = b_ctab.as<pag_ctg>();
    o.ctg_ent_off = b_eoff.as<uint32_t>();
    o.n_ctg_ent_off = off_total ? off_total : 1;
    o.ctg_ent = b_ent.as<uint32_t>();
    o.n_ctg_ent = ent_total + 1;
    o.n_refs = n_refs;
    o.refs = b_rtab.as<pag_ref>();
    o.eps = raw->eps;
    o.cov_filter = raw->cov_filter;
    o.outer_sample = raw->outer_sample;
    o.topk_ctg = raw->topk_ctg;
    o.topk_ref = raw->topk_ref;
    o.reserved = 0;
    *out = o;
    return PAG_OK;
}
