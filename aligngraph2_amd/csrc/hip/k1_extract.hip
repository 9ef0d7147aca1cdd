// k1_extract.hip — K0/K1: alignment-column walk, k-mer extraction from 2-bit packed read strands,
// solid-set test, greedy >= outer_sample sampling, and emission of position / edge tuples in the
// canonical order.  One wavefront (= one 64-thread workgroup) owns one (read, strand) job; all
// cross-lane work is wave shuffles/ballots, per-job scratch lives in 12.5 KB of LDS.
//
// Reference semantics reproduced here (paths under PAGraph/src/tools/):
//   align/ParseAlignTools.tcc:44-70   exactAlign: column classes -> (read pos, target pos) pairs
//   align/Aligner.tcc:73-96, 152-166  which read strand receives positions; per-base position lists
//   align/Aligner.cpp:222-233         queryContig: one list entry per contig->ref entry of the base
//   position/PositionProcessor.cpp:37-55  DualPos = (ctgSingle, refSingle), u32
//   kmer/KmerHelper.cpp:7-25          rolling 2-bit code, first base most significant
//   seq/CompressedSeq.cpp:56-74       reverse strand = complement read back to front
//   graph/PABruijnGraph.tcc:5-26      sampleSequence: candidate = has position AND solid; keep the
//                                     first, then every candidate >= outer_sample after the last kept
//   graph/PABruijnGraph.cpp:238-257   addPositionAndEdge: all DualPos of a kept base, then one edge
//                                     (prev kept k-mer -> this k-mer, step = index difference)
//
// Canonical order of the emitted streams: pass 1 then pass 2; inside a pass the reads
// in emit_order; forward strand then reverse strand; samples ascending; per sample the list order
// (alignment order x contig->ref entry order).  Offsets come from an exclusive scan over per-job counts
// (count launch -> scan -> emit launch), so the streams are dense and need no atomics.
#include "pag_device.hpp"

namespace pagdev {

constexpr int TILE = 1024;  // read positions per tile: 64 lanes x 16
constexpr int CHUNK = 1024; // alignment columns per colidx chunk: 64 lanes x 16

// Kept samples are at least `outer` positions apart (sampleSequence), so a tile of 1024 positions holds at most
// ceil(1024 / outer) of them: with the pipeline's outer = 3 (pagraph.cpp:113) the per-sample arrays need 342 entries, not
// 1024 — 4.6 KB of LDS per wave instead of 12.5 KB, which is what decides how many waves a compute unit carries (the
// kernel is bound by its chain of LDS round trips and gathers per tile, not by bandwidth: 12 waves per unit before,
// as many as the registers allow now).  MAXS = 1024 is the build for outer < 3.
template <int MAXS>
struct WaveLds {
    uint32_t kept[64];   // per lane-slot: 16-bit mask of kept positions
    uint32_t rank[64];   // per lane-slot: tile-local rank of its first kept sample
    uint32_t scode[MAXS]; // per tile-local sample: k-mer code
    uint32_t scnt[MAXS];  // per sample: tuple count, then exclusive offset inside the tile
    uint32_t srun[MAXS];  // per sample: tuples written so far
    uint32_t st[MAXS];    // per sample: target position under the alignment being written (F2), all ones = not covered by it
};
constexpr int MAXS_SPACED = 352;  // >= ceil(1024 / 3), a multiple of 16

// ---------------------------------------------------------------------------------------------
// column index: for every alignment, per 1024-column chunk in WALK order, the number of emitting
// columns and of target-advancing columns before the chunk.  One wave per alignment.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_cols16(const uint32_t *__restrict__ diff, uint64_t diff_off, uint32_t n_cols,
                                                bool back, uint32_t j0, uint32_t *n_lane) {
    // returns the column classes of walk-order columns j0 .. j0+15 (2 bits each, jj at bits 2jj)
    if (j0 >= n_cols) {
        *n_lane = 0;
        return 0;
    }
    uint32_t n = n_cols - j0;
    if (n > 16) n = 16;
    *n_lane = n;
    if (!back) return diff[diff_off + (j0 >> 4)];  // j0 is a multiple of 16
    int64_t s_lo = (int64_t)n_cols - 16 - (int64_t)j0;  // storage column of walk column j0 + 15
    uint32_t w32;
    if (s_lo >= 0) {
        uint64_t w = diff_off + ((uint64_t)s_lo >> 4);
        uint32_t sh = ((uint32_t)s_lo & 15u) * 2u;
        uint32_t lo = diff[w];
        w32 = lo >> sh;
        if (sh) w32 |= diff[w + 1] << (32 - sh);
    } else {
        uint32_t m = (uint32_t)(-s_lo);  // 1..15 missing low columns
        w32 = diff[diff_off] << (2 * m);
    }
    return rev2(w32);
}

__device__ __forceinline__ void col_masks(uint32_t bits, uint32_t n_lane, uint32_t *emitb, uint32_t *radvb) {
    uint32_t valid = n_lane >= 16 ? 0x55555555u : (((1u << (2 * n_lane)) - 1u) & 0x55555555u);
    uint32_t qd = bits & 0x55555555u, rd = (bits >> 1) & 0x55555555u;
    *emitb = ~(qd & ~rd) & valid;  // class != 01: the query base is emitted
    *radvb = ~(rd & ~qd) & valid;  // class != 10: the target advances
}

__global__ __launch_bounds__(64) void colidx_kernel(const pag_aln *__restrict__ aln, uint64_t n_aln,
                                                    const uint32_t *__restrict__ diff,
                                                    const uint64_t *__restrict__ colidx_off, uint2 *__restrict__ colidx) {
    uint64_t ai = blockIdx.x;
    if (ai >= n_aln) return;
    pag_aln al = aln[ai];
    if (!(al.flags & PAG_ALN_ELIGIBLE) || al.query == PAG_NONE) return;
    bool back = (al.flags & PAG_ALN_WALK_BACK) != 0;
    uint32_t n_chunks = (al.n_cols + CHUNK - 1) / CHUNK;
    uint32_t e = 0, r = 0;
    uint64_t base = colidx_off[ai];
    for (uint32_t c = 0; c < n_chunks; ++c) {
        if (lane_id() == 0) colidx[base + c] = make_uint2(e, r);
        uint32_t n_lane, eb, rb;
        uint32_t bits = load_cols16(diff, al.diff_off, al.n_cols, back, c * CHUNK + lane_id() * 16, &n_lane);
        col_masks(bits, n_lane, &eb, &rb);
        e += wave_sum(__popc(eb));
        r += wave_sum(__popc(rb));
    }
}

int launch_colidx(const pag_aln *aln, uint64_t n_aln, const uint32_t *diff, const uint64_t *colidx_off, uint2 *colidx,
                  hipStream_t s) {
    if (n_aln == 0) return PAG_OK;
    colidx_kernel<<<dim3((unsigned)n_aln), dim3(64), 0, s>>>(aln, n_aln, diff, colidx_off, colidx);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

// ---------------------------------------------------------------------------------------------
// wave-cooperative walk of the columns of one alignment that emit read positions in [lo, hi).
// f(q, t) is called by the lane that owns the column: q = read-strand position, t = target position.
// ---------------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void walk_alignment(const pag_aln &al, const uint32_t *__restrict__ diff,
                                               const uint2 *__restrict__ cidx, uint32_t lo, uint32_t hi, F f) {
    uint32_t q_end = al.q_start + al.n_valid;
    uint32_t a = lo > al.q_start ? lo : al.q_start;
    uint32_t b = hi < q_end ? hi : q_end;
    if (a >= b) return;
    uint32_t e0 = a - al.q_start, e1 = b - al.q_start;  // emit ordinals wanted: [e0, e1)
    bool back = (al.flags & PAG_ALN_WALK_BACK) != 0;
    uint32_t n_chunks = (al.n_cols + CHUNK - 1) / CHUNK;
    // last chunk whose emit prefix is <= e0 (uniform binary search)
    uint32_t c_lo = 0, c_hi = n_chunks;
    while (c_hi - c_lo > 1) {
        uint32_t mid = (c_lo + c_hi) >> 1;
        if (cidx[mid].x <= e0) c_lo = mid;
        else c_hi = mid;
    }
    for (uint32_t c = c_lo; c < n_chunks; ++c) {
        uint2 pre = cidx[c];
        if (pre.x >= e1) break;
        uint32_t n_lane, eb, rb;
        uint32_t bits = load_cols16(diff, al.diff_off, al.n_cols, back, c * CHUNK + lane_id() * 16, &n_lane);
        col_masks(bits, n_lane, &eb, &rb);
        uint32_t tot;
        uint32_t eex = wave_excl_sum(__popc(eb), &tot) + pre.x;
        uint32_t rex = wave_excl_sum(__popc(rb), &tot) + pre.y;
        uint32_t m = eb;
        while (m) {
            uint32_t bit = __ffs(m) - 1;  // even bit index 2*jj
            m &= m - 1;
            uint32_t below = (1u << bit) - 1u;
            uint32_t ord = eex + __popc(eb & below);
            if (ord >= e0 && ord < e1) f(al.q_start + ord, al.t_start + rex + __popc(rb & below));
        }
    }
}

// ... for the KEPT samples of the tile [lo, lo + TILE) only: g(sample, t) with sample = the kept sample's number inside the tile
// (from the per-lane masks / ranks `kept`, `rank` of the tile's positions) and t its target position.  walk_alignment visits
// every emitting column — sixteen turns of ~25 instructions per lane and chunk — although one position in six is kept: here a
// lane looks its (at most sixteen) positions up in the kept masks first and only finds the columns of those (the n-th set bit
// of its emit mask).  The emission launch is bound by the instructions it issues.
__device__ __forceinline__ uint32_t even_bits16(uint32_t x) {  // bits 0, 2, 4, .. 30 -> bits 0 .. 15
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0F0F0F0Fu;
    x = (x | (x >> 4)) & 0x00FF00FFu;
    return (x | (x >> 8)) & 0xFFFFu;
}
__device__ __forceinline__ uint32_t nth_set_bit16(uint32_t x, uint32_t n) {  // position of the (n + 1)-th set bit (it exists)
    uint32_t pos = 0, c = __popc(x & 0xFFu);
    if (n >= c) n -= c, pos = 8u, x >>= 8;
    c = __popc(x & 0xFu);
    if (n >= c) n -= c, pos += 4u, x >>= 4;
    c = __popc(x & 0x3u);
    if (n >= c) n -= c, pos += 2u, x >>= 2;
    return pos + (n >= (x & 1u) ? 1u : 0u);
}
template <typename G>
__device__ __forceinline__ void walk_kept(const pag_aln &al, const uint32_t *__restrict__ diff, const uint2 *__restrict__ cidx, uint32_t lo,
                                          uint32_t hi, const uint32_t (&kept)[64], const uint32_t (&rank)[64], G g) {
    const uint32_t q_end = al.q_start + al.n_valid;
    const uint32_t a = lo > al.q_start ? lo : al.q_start;
    const uint32_t b = hi < q_end ? hi : q_end;
    if (a >= b) return;
    const uint32_t e0 = a - al.q_start, e1 = b - al.q_start;  // emit ordinals wanted: [e0, e1)
    const bool back = (al.flags & PAG_ALN_WALK_BACK) != 0;
    const uint32_t n_chunks = (al.n_cols + CHUNK - 1) / CHUNK;
    uint32_t c_lo = 0, c_hi = n_chunks;
    while (c_hi - c_lo > 1) {
        const uint32_t mid = (c_lo + c_hi) >> 1;
        if (cidx[mid].x <= e0) c_lo = mid;
        else c_hi = mid;
    }
    for (uint32_t c = c_lo; c < n_chunks; ++c) {
        const uint2 pre = cidx[c];
        if (pre.x >= e1) break;
        uint32_t n_lane, eb, rb;
        const uint32_t bits = load_cols16(diff, al.diff_off, al.n_cols, back, c * CHUNK + lane_id() * 16, &n_lane);
        col_masks(bits, n_lane, &eb, &rb);
        const uint32_t ne = __popc(eb);
        uint32_t tot;
        const uint32_t eex = wave_excl_sum(ne, &tot) + pre.x;
        const uint32_t rex = wave_excl_sum(__popc(rb), &tot) + pre.y;
        const uint32_t o_lo = eex > e0 ? eex : e0, o_hi = eex + ne < e1 ? eex + ne : e1;  // my emit ordinals inside [e0, e1)
        if (o_lo >= o_hi) continue;
        const uint32_t d_lo = al.q_start + o_lo - lo, n = o_hi - o_lo;  // their positions in the tile: d_lo .. d_lo + n - 1 (n <= 16)
        const uint32_t w = d_lo >> 4, sh = d_lo & 15u;
        const bool two = sh + n > 16u;
        const uint32_t k0 = kept[w] & 0xFFFFu, k1 = two ? kept[w + 1] & 0xFFFFu : 0u;
        uint32_t km = ((k0 | (k1 << 16)) >> sh) & ((1u << n) - 1u);  // bit i: position d_lo + i is a kept sample
        if (!km) continue;
        const uint32_t e16 = even_bits16(eb), r16 = even_bits16(rb);
        const uint32_t r0 = rank[w], r1 = two ? rank[w + 1] : 0u;
        while (km) {
            const uint32_t i = (uint32_t)__ffs((int)km) - 1u;
            km &= km - 1u;
            const uint32_t col = nth_set_bit16(e16, o_lo + i - eex);  // my column that emits ordinal o_lo + i
            const uint32_t t = al.t_start + rex + __popc(r16 & ((1u << col) - 1u));
            const uint32_t d = d_lo + i;
            const bool first = (d >> 4) == w;
            g((first ? r0 : r1) + __popc((first ? k0 : k1) & ((1u << (d & 15u)) - 1u)), t);
        }
    }
}

// k-mer codes of the 16 positions p0 .. p0+15 of one read strand (kmer2Code / reverse strand of
// CompressedSeq): a 64-bit window of the 2-bit packed read, 2-bit-group reversal for the forward strand
// (first base most significant), complement for the reverse strand.
__device__ __forceinline__ void lane_codes(const uint32_t *__restrict__ words, uint32_t strand, uint32_t len, uint32_t k,
                                           uint32_t kmask, uint32_t p0, uint32_t n_mine, uint32_t (&code)[16]) {
    uint64_t W = 0;
    uint32_t sh0 = 0;  // reverse strand: window bit offset of position j is 2*(15 - j) + sh0
    if (n_mine) {      // lanes past the end of the strand must not touch memory
        if (strand == 0) {
            uint32_t w0 = p0 >> 4;
            W = (uint64_t)words[w0] | ((uint64_t)words[w0 + 1] << 32);
        } else {
            int64_t a0 = (int64_t)len - (int64_t)k - (int64_t)p0 - 15;
            uint32_t a1 = a0 > 0 ? (uint32_t)a0 : 0u;
            uint32_t w = a1 >> 4, sh = (a1 & 15u) * 2u;
            uint64_t lo64 = (uint64_t)words[w] | ((uint64_t)words[w + 1] << 32);
            W = lo64 >> sh;
            if (sh) W |= (uint64_t)words[w + 2] << (64 - sh);
            sh0 = (uint32_t)((a0 - (int64_t)a1) * 2);  // <= 0 as a signed value; fine for valid j
        }
    }
    if (strand == 0) {
        fwd_codes16(W, k, kmask, code);
        return;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t shj = (uint32_t)(2 * (15 - j)) + sh0;  // wraps harmlessly for invalid j
        const uint32_t x = (uint32_t)(W >> (shj & 63u)) & kmask;
        code[j] = (~x) & kmask;
    }
}

// The same window, kept, for a lane that wants the codes of a FEW of its sixteen positions (the emission launch: the kept
// samples, one position in six): forward strand — all groups reversed once (fwd_codes16's R), a code is a shift and a mask;
// reverse strand — the window and its offset, a code is a shift, a complement and a mask.
struct LaneWindow {
    uint64_t x = 0;   // forward: the window with its 2-bit groups reversed; reverse: the window
    uint32_t s0 = 0;  // forward: shift of position 0 (2 (32 - k)); reverse: sh0 of lane_codes
    __device__ __forceinline__ uint32_t code(uint32_t strand, uint32_t kmask, uint32_t j) const {
        if (strand == 0) return (uint32_t)(x >> (s0 - 2u * j)) & kmask;
        return ~(uint32_t)(x >> ((2u * (15u - j) + s0) & 63u)) & kmask;
    }
};
__device__ __forceinline__ LaneWindow lane_window(const uint32_t *__restrict__ words, uint32_t strand, uint32_t len, uint32_t k, uint32_t p0,
                                                  uint32_t n_mine) {
    LaneWindow L;
    if (!n_mine) return L;  // lanes past the end of the strand must not touch memory
    if (strand == 0) {
        const uint32_t w0 = p0 >> 4;
        const uint32_t lo = words[w0], hi = words[w0 + 1];
        L.x = ((uint64_t)rev2(lo) << 32) | (uint64_t)rev2(hi);
        L.s0 = 2u * (32u - k);
    } else {
        const int64_t a0 = (int64_t)len - (int64_t)k - (int64_t)p0 - 15;
        const uint32_t a1 = a0 > 0 ? (uint32_t)a0 : 0u;
        const uint32_t w = a1 >> 4, sh = (a1 & 15u) * 2u;
        const uint64_t lo64 = (uint64_t)words[w] | ((uint64_t)words[w + 1] << 32);
        L.x = lo64 >> sh;
        if (sh) L.x |= (uint64_t)words[w + 2] << (64 - sh);
        L.s0 = (uint32_t)((a0 - (int64_t)a1) * 2);
    }
    return L;
}

// Solid-set membership of every k-mer start of every read strand that some alignment (of either pass)
// touches: ONE random bitmap gather per read position for the whole build.  The four extraction launches
// (count / emit x pass 1 / pass 2) then read 16-bit masks per 16 positions instead of gathering again —
// the gathers (a 64-byte line each from a 4^k-bit table that no L2 holds) were 98 % of their HBM traffic.
//
// XCD SLICES (round 3).  The table (32 MB at k = 14) is eight times an XCD's 4 MB L2, so every gather of the one-wave-per-
// strand version missed it: 57 GB of line fetches for < 1 GB of input, 15.7 ms at configs[1].  Now EIGHT workgroups visit
// every strand, workgroup number b on XCD b mod 8 (consecutive workgroups go round the XCDs), and workgroup b only looks
// up the codes whose top three bits are b mod 8: each XCD's L2 then only ever sees its own eighth of the table.  Every
// workgroup derives all codes of its tiles (a few ALU instructions each) and writes its own mask array; the extraction
// ORs the eight.
constexpr uint32_t SOLID_SLICES = 8;
__global__ __launch_bounds__(64) void solid_mask_kernel(ExtractArgs A, const pag_aln *__restrict__ aln2,
                                                        const uint64_t *__restrict__ qoff2, uint16_t *__restrict__ mask) {
    const uint32_t job = blockIdx.x / SOLID_SLICES, slice = blockIdx.x % SOLID_SLICES;
    const uint32_t lane = lane_id();
    const uint32_t r = A.emit_order[job >> 1];
    const uint32_t strand = job & 1u;
    const uint32_t len = A.read_len[r];
    const uint32_t k = A.k;
    if (len < k) return;
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const uint32_t slice_shift = 2 * k - 3;  // (k >= 2)
    const uint32_t *__restrict__ words = (const uint32_t *)(A.packed + A.read_off[r]);
    const uint32_t n_pos = len - k + 1;
    // union of the intervals any eligible alignment of this strand covers (both databases)
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (int db = 0; db < 2; ++db) {
        const pag_aln *al = db == 0 ? A.aln : aln2;
        const uint64_t *qo = db == 0 ? A.query_off : qoff2;
        for (uint64_t ai = qo[r]; ai < qo[r + 1]; ++ai) {
            pag_aln a = al[ai];
            if (!(a.flags & PAG_ALN_ELIGIBLE) || a.q_start == PAG_NONE || a.n_valid == 0) continue;
            if (((a.flags & PAG_ALN_REV_STRAND) ? 1u : 0u) != strand) continue;
            lo = a.q_start < lo ? a.q_start : lo;
            hi = a.q_start + a.n_valid > hi ? a.q_start + a.n_valid : hi;
        }
    }
    if (hi > n_pos) hi = n_pos;
    if (lo >= hi) return;
    uint16_t *m = mask + (uint64_t)slice * A.solid_mask_stride + 2ull * (A.read_off[r] >> 2) + strand;
    for (uint32_t t0 = lo & ~(uint32_t)(TILE - 1); t0 < hi; t0 += TILE) {
        const uint32_t p0 = t0 + lane * 16;
        const uint32_t n_mine = p0 >= n_pos ? 0u : (n_pos - p0 > 16 ? 16u : n_pos - p0);
        if (p0 + 16 <= lo || p0 >= hi || n_mine == 0) continue;
        uint32_t code[16];
        lane_codes(words, strand, len, k, kmask, p0, n_mine, code);
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if ((uint32_t)j < n_mine && (code[j] >> slice_shift) == slice) bits |= ((A.solid_bits[code[j] >> 5] >> (code[j] & 31u)) & 1u) << j;
        m[2ull * (p0 >> 4)] = (uint16_t)bits;
    }
}

int launch_solid_mask(const ExtractArgs &a, const pag_aln *aln2, const uint64_t *qoff2, uint16_t *mask, hipStream_t s) {
    if (a.n_reads == 0) return PAG_OK;
    solid_mask_kernel<<<dim3(SOLID_SLICES * 2u * a.n_reads), dim3(64), 0, s>>>(a, aln2, qoff2, mask);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
uint32_t solid_mask_slices() { return SOLID_SLICES; }

// ---------------------------------------------------------------------------------------------
// sampler automaton: state = min(S, positions since the last kept sample), S = "free".
// A function state -> state is a packed table of 4-bit entries.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tab_get(uint32_t tab, uint32_t s) { return (tab >> (4 * s)) & 15u; }
// ... of up to four states as four bytes: "first, then" is out.byte[s] = then.byte[first.byte[s]] — one v_perm_b32 (selector
// bytes 0..3 pick from the second operand)
__device__ __forceinline__ uint32_t byte_compose(uint32_t first, uint32_t then) { return __builtin_amdgcn_perm(then, then, first); }
__device__ __forceinline__ uint32_t byte_get(uint32_t tab, uint32_t s) { return (tab >> (8 * s)) & 255u; }
__device__ __forceinline__ uint32_t tab_compose(uint32_t first, uint32_t then, uint32_t S) {
    uint32_t out = 0;
    for (uint32_t s = 0; s <= S; ++s) out |= tab_get(then, tab_get(first, s)) << (4 * s);
    return out;
}

template <bool EMIT, int MAXS>
__global__ __launch_bounds__(64) void extract_kernel(ExtractArgs A) {
    __shared__ WaveLds<MAXS> L;
    const uint32_t job = A.exec_perm ? A.exec_perm[blockIdx.x] : blockIdx.x;
    const uint32_t lane = lane_id();
    const uint32_t r = A.emit_order[job >> 1];
    const uint32_t strand = job & 1u;
    const uint32_t len = A.read_len[r];
    const uint32_t k = A.k;
    const uint32_t S = A.outer;
    const uint64_t a_lo = A.query_off[r], a_hi = A.query_off[r + 1];
    const uint32_t kmask = k >= 16 ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
    const uint32_t *__restrict__ words = (const uint32_t *)(A.packed + A.read_off[r]);

    // every lane runs the same loop over the read's alignment list (uniform); `body(al, ai)` is called
    // for the alignments that put positions on THIS strand
    auto for_active = [&](auto body) {
        int done = 0;
        for (uint64_t ai = a_lo; ai < a_hi; ++ai) {
            if (A.topk >= 0 && done >= A.topk) break;
            pag_aln al = A.aln[ai];
            if (!(al.flags & PAG_ALN_ELIGIBLE)) continue;
            if (A.cov_ok && !A.cov_ok[ai]) continue;
            ++done;
            if (((al.flags & PAG_ALN_REV_STRAND) ? 1u : 0u) != strand) continue;
            if (al.q_start == PAG_NONE || al.n_valid == 0) continue;
            body(al, ai);
        }
    };

    uint32_t job_samples = 0;
    uint64_t job_tuples = 0;
    bool any = false;
    if (len >= k) for_active([&](const pag_aln &, uint64_t) { any = true; });
    if (!any) {
        if (!EMIT && lane == 0) {
            A.job_samples[A.job_base + job] = 0;
            A.job_tuples[A.job_base + job] = 0;
        }
        return;
    }
    const uint32_t n_pos = len - k + 1;
    // EMIT: how many tuples a kept sample gets is, for alignments whose bases all have ONE list entry (pass 2; pass 1 on a contig
    // without multi-entry bases), the number of active alignments that cover its position — interval arithmetic on the 16-bit
    // masks, as the counting launch does it, instead of a first walk over the alignments' columns per tile (round 5).  Up to 15
    // alignments per strand: four bits per position.
    bool fast_counts = false;
    if (EMIT) {
        uint32_t n_act = 0;
        bool any_multi = false;
        for_active([&](const pag_aln &al, uint64_t) {
            ++n_act;
            any_multi = any_multi || (A.pass == 0 && A.ctgs[al.target].multi != 0);
        });
        fast_counts = !any_multi && n_act <= 15u;
    }
    const uint64_t tuple_base = EMIT ? A.tuple_off[A.job_base + job] : 0;
    const uint64_t edge_base = EMIT ? A.edge_off[A.job_base + job] : 0;

    uint32_t state = S;  // free
    // (S <= 3) the two transition tables of a position — without / with a candidate — as bytes: state s -> byte s
    uint32_t byte_t0 = 0, byte_t1 = 0;
    for (uint32_t st = 0; st < 4u; ++st) {
        byte_t0 |= (st + 1u > S ? S : st + 1u) << (8u * st);
        byte_t1 |= (st + 1u >= S ? 0u : st + 1u) << (8u * st);
    }
    bool carry_valid = false;
    uint32_t carry_pos = 0, carry_code = 0;

    for (uint32_t t0 = 0; t0 < n_pos; t0 += TILE) {
        const uint32_t p0 = t0 + lane * 16;
        const uint32_t t_hi = t0 + TILE < n_pos ? t0 + TILE : n_pos;

        // ---- A. which of my 16 positions have a non-empty position list
        uint32_t ne = 0;
        for_active([&](const pag_aln &al, uint64_t) {
            uint32_t lo = al.q_start > p0 ? al.q_start : p0;
            uint32_t hi = al.q_start + al.n_valid;
            if (hi > p0 + 16) hi = p0 + 16;
            if (hi > n_pos) hi = n_pos;
            if (lo < hi) ne |= ((1u << (hi - p0)) - 1u) & ~((1u << (lo - p0)) - 1u);
        });
        if (__ballot(ne != 0) == 0ull) {
            state = S;  // >= S positions without a candidate (or the end of the strand)
            continue;
        }

        // ---- B. the window my positions' k-mer codes come from (the codes of the kept ones are taken in E) + solid test
        uint32_t cand = 0;
        const uint32_t n_mine = p0 >= n_pos ? 0u : (n_pos - p0 > 16 ? 16u : n_pos - p0);
        LaneWindow win;
        if (EMIT) win = lane_window(words, strand, len, k, p0, n_mine);
        if (A.all_solid) {
            cand = ne;
        } else if (ne) {
            // the solid bits of this strand were gathered once by solid_mask_kernel, one array per table slice
            const uint64_t mi = 2ull * ((A.read_off[r] >> 2) + (p0 >> 4)) + strand;
            uint32_t sm = 0;
#pragma unroll
            for (uint32_t sl = 0; sl < SOLID_SLICES; ++sl) sm |= A.solid_mask[(uint64_t)sl * A.solid_mask_stride + mi];
            cand = ne & sm;
        }

        // ---- C. greedy sampling as an associative scan of state-transition tables
        uint32_t v;
        if (S <= 3u) {
            // (the pipeline's outer = 3, pagraph.cpp:113) four states: a table is four BYTES and "first, then" is ONE byte permute
            // (byte_compose) — the nibble tables below cost ~30 vector instructions per composition, 22 compositions per lane and
            // tile: 700 of the kernel's instructions per tile, and the kernel is bound by their number (SQ counters, round 5)
            uint32_t tab = 0x03020100u;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j)
                if (j < n_mine) tab = byte_compose(tab, ((cand >> j) & 1u) ? byte_t1 : byte_t0);
            uint32_t incl = tab;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t prev = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl = byte_compose(prev, incl);
            }
            const uint32_t excl = __shfl_up(incl, 1, 64);
            v = lane == 0 ? state : byte_get(excl, state);
            state = byte_get(__shfl(incl, 63, 64), state);
        } else {
            uint32_t tab = 0;
            for (uint32_t s = 0; s <= S; ++s) tab |= s << (4 * s);
            for (uint32_t j = 0; j < n_mine; ++j) {
                uint32_t c = (cand >> j) & 1u, nt = 0;
                for (uint32_t s = 0; s <= S; ++s) {
                    uint32_t x = tab_get(tab, s);
                    x = (c && x + 1 >= S) ? 0u : (x + 1 > S ? S : x + 1);
                    nt |= x << (4 * s);
                }
                tab = nt;
            }
            uint32_t incl = tab;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t prev = __shfl_up(incl, d, 64);
                if ((int)lane >= d) incl = tab_compose(prev, incl, S);
            }
            uint32_t excl = __shfl_up(incl, 1, 64);
            v = lane == 0 ? state : tab_get(excl, state);
            state = tab_get(__shfl(incl, 63, 64), state);
        }
        uint32_t kept = 0;
        for (uint32_t j = 0; j < n_mine; ++j) {
            uint32_t c = (cand >> j) & 1u;
            if (c && v + 1 >= S) {
                kept |= 1u << j;
                v = 0;
            } else {
                v = v + 1 > S ? S : v + 1;
            }
        }

        // ---- D. sample ranks
        uint32_t tile_samples;
        const uint32_t rank0 = wave_excl_sum(__popc(kept), &tile_samples);
        if (tile_samples == 0) continue;

        if (EMIT) {
            uint64_t cnt4 = 0;  // (fast_counts) tuples of my position j at bits 4j
            if (fast_counts && kept)
                for_active([&](const pag_aln &al, uint64_t) {
                    uint32_t lo = al.q_start > p0 ? al.q_start : p0;
                    uint32_t hi = al.q_start + al.n_valid;
                    if (hi > p0 + 16) hi = p0 + 16;
                    if (lo >= hi) return;
                    uint32_t m = kept & ((1u << (hi - p0)) - 1u) & ~((1u << (lo - p0)) - 1u);
                    while (m) {
                        cnt4 += 1ull << (4u * (uint32_t)(__ffs(m) - 1));
                        m &= m - 1u;
                    }
                });
            __syncthreads();  // previous tile's readers are done with L
            L.kept[lane] = kept;
            L.rank[lane] = rank0;
            // ---- E. edges between consecutive kept samples (prev -> this)
            uint32_t last_pos = 0, last_code = 0;
            if (kept) {
                const uint32_t jl = 31u - (uint32_t)__clz((int)kept);
                last_pos = p0 + jl;
                last_code = win.code(strand, kmask, jl);
            }
            uint64_t has = __ballot(kept != 0);
            uint64_t before = has & lanemask_lt();
            int src = before ? 63 - __clzll((long long)before) : 0;
            uint32_t pp = __shfl(last_pos, src, 64), pc = __shfl(last_code, src, 64);
            bool pvalid = before != 0;
            if (!pvalid) {
                pp = carry_pos;
                pc = carry_code;
                pvalid = carry_valid;
            }
            // (a turn per KEPT sample — at most six of a lane's sixteen positions, they are three apart — with its code taken from
            // the window: sixteen predicated bodies over a code array were a third of the launch's instructions)
            uint32_t i = 0;
            for (uint32_t m = kept; m; m &= m - 1u) {
                const uint32_t j = (uint32_t)__ffs((int)m) - 1u;
                const uint32_t cj = win.code(strand, kmask, j);
                const uint32_t s_local = rank0 + i;
                L.scode[s_local] = cj;
                L.scnt[s_local] = (uint32_t)(cnt4 >> (4 * j)) & 15u;  // (0 without fast_counts: counted by the walk below)
                L.srun[s_local] = 0;
                if (pvalid) {
                    const uint64_t slot = edge_base + (uint64_t)job_samples + s_local - 1;
                    A.ekey[slot] = pc;
                    A.eval[slot] = ((uint64_t)cj << 32) | ((uint64_t)(p0 + j - pp) << 1) | (uint64_t)A.pass;
                }
                pp = p0 + j;
                pc = cj;
                pvalid = true;
                ++i;
            }
            int top = 63 - __clzll((long long)has);
            carry_pos = __shfl(last_pos, top, 64);
            carry_code = __shfl(last_code, top, 64);
            carry_valid = true;
            __syncthreads();
        }

        // ---- F. tuples of the kept samples
        uint32_t tile_tuples = 0;
        if (!EMIT) {
            for_active([&](const pag_aln &al, uint64_t ai) {
                bool multi = A.pass == 0 && A.ctgs[al.target].multi != 0;
                if (!multi) {
                    uint32_t lo = al.q_start > p0 ? al.q_start : p0;
                    uint32_t hi = al.q_start + al.n_valid;
                    if (hi > p0 + 16) hi = p0 + 16;
                    if (lo < hi) tile_tuples += __popc(kept & ((1u << (hi - p0)) - 1u) & ~((1u << (lo - p0)) - 1u));
                } else {
                    // entry counts differ per base: walk.  kept masks must be visible to other lanes
                    __syncthreads();
                    L.kept[lane] = kept;
                    __syncthreads();
                    const pag_ctg cg = A.ctgs[al.target];
                    walk_alignment(al, A.diff, A.colidx + A.colidx_off[ai], t0, t_hi, [&](uint32_t q, uint32_t t) {
                        uint32_t d = q - t0;
                        if ((L.kept[d >> 4] >> (d & 15u)) & 1u)
                            tile_tuples += A.ctg_ent_off[cg.map_off + t + 1] - A.ctg_ent_off[cg.map_off + t];
                    });
                }
            });
            job_tuples += wave_sum(tile_tuples);
        } else {
            // F1: tuples per sample
            if (!fast_counts) for_active([&](const pag_aln &al, uint64_t ai) {
                const bool ctg_pass = A.pass == 0;
                pag_ctg cg{};
                if (ctg_pass) cg = A.ctgs[al.target];
                walk_alignment(al, A.diff, A.colidx + A.colidx_off[ai], t0, t_hi, [&](uint32_t q, uint32_t t) {
                    uint32_t d = q - t0;
                    uint32_t km = L.kept[d >> 4];
                    if ((km >> (d & 15u)) & 1u) {
                        uint32_t s_local = L.rank[d >> 4] + __popc(km & ((1u << (d & 15u)) - 1u));
                        uint32_t c = ctg_pass ? A.ctg_ent_off[cg.map_off + t + 1] - A.ctg_ent_off[cg.map_off + t] : 1u;
                        L.scnt[s_local] += c;
                    }
                });
                __syncthreads();
            });
            // exclusive scan of scnt over the tile's samples (PER per lane)
            {
                constexpr int PER = MAXS / 64 + (MAXS % 64 ? 1 : 0);
                uint32_t loc[PER], sum = 0;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    uint32_t idx = lane * PER + i;
                    loc[i] = idx < tile_samples ? L.scnt[idx] : 0u;
                    sum += loc[i];
                }
                uint32_t ex = wave_excl_sum(sum, &tile_tuples);
                __syncthreads();
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    uint32_t idx = lane * PER + i;
                    if (idx < tile_samples) L.scnt[idx] = ex;
                    ex += loc[i];
                }
                __syncthreads();
            }
            // F2: write.  Two phases per alignment (round 5): the walk only notes the target position of every kept sample it
            // covers (LDS), then the lanes take the samples IN ORDER — lane l the samples l, l + 64, .. — with the map lookups
            // of all its samples in flight together and the tuples of neighbouring samples written by neighbouring lanes.
            // (Before, the lane that owned a column did the lookups and the stores from inside the walk: up to sixteen
            // columns per lane one after the other, each kept one a chain of two dependent gathers and two stores.)
            const uint64_t out0 = tuple_base + job_tuples;
            for_active([&](const pag_aln &al, uint64_t ai) {
                const bool ctg_pass = A.pass == 0;
                pag_ctg cg{};
                uint32_t ref_base = 0;
                if (ctg_pass) cg = A.ctgs[al.target];
                else ref_base = A.refs[al.target].single_base;
                constexpr int PER = MAXS / 64 + (MAXS % 64 ? 1 : 0);
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    if (lane + 64u * i < tile_samples) L.st[lane + 64u * i] = 0xFFFFFFFFu;
                __syncthreads();
                walk_kept(al, A.diff, A.colidx + A.colidx_off[ai], t0, t_hi, L.kept, L.rank, [&](uint32_t sample, uint32_t t) { L.st[sample] = t; });
                __syncthreads();
                constexpr int G = 6;  // samples per lane and turn (MAXS = 352: one turn)
                for (uint32_t sb = 0; sb < tile_samples; sb += 64u * G) {
                    uint32_t tt[G], e0[G], e1[G], ent0[G];
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        const uint32_t sx = sb + lane + 64u * i;
                        tt[i] = sx < tile_samples ? L.st[sx] : 0xFFFFFFFFu;
                    }
                    if (ctg_pass) {
#pragma unroll
                        for (int i = 0; i < G; ++i)
                            if (tt[i] != 0xFFFFFFFFu) {
                                e0[i] = A.ctg_ent_off[cg.map_off + tt[i]];
                                e1[i] = A.ctg_ent_off[cg.map_off + tt[i] + 1];
                            }
#pragma unroll
                        for (int i = 0; i < G; ++i)  // (a base's first entry — nearly always its only one — asked for together as well)
                            if (tt[i] != 0xFFFFFFFFu && e1[i] > e0[i]) ent0[i] = A.ctg_ent[e0[i]];
                    }
#pragma unroll
                    for (int i = 0; i < G; ++i) {
                        if (tt[i] == 0xFFFFFFFFu) continue;
                        const uint32_t sx = sb + lane + 64u * i;
                        const uint32_t run = L.srun[sx];
                        uint64_t dst = out0 + L.scnt[sx] + run;
                        const uint32_t kc = L.scode[sx];
                        if (ctg_pass) {
                            const uint64_t hi = (uint64_t)(cg.single_base + tt[i]) << 32;
                            for (uint32_t e = e0[i]; e < e1[i]; ++e) {
                                A.tkey[dst] = kc;
                                A.tval[dst] = hi | (e == e0[i] ? ent0[i] : A.ctg_ent[e]);
                                ++dst;
                            }
                            L.srun[sx] = run + (e1[i] - e0[i]);
                        } else {
                            A.tkey[dst] = kc;
                            A.tval[dst] = (uint64_t)(uint32_t)(ref_base + tt[i]);
                            L.srun[sx] = run + 1;
                        }
                    }
                }
                __syncthreads();
            });
            job_tuples += tile_tuples;
        }
        job_samples += tile_samples;
    }

    if (!EMIT && lane == 0) {
        A.job_samples[A.job_base + job] = job_samples;
        A.job_tuples[A.job_base + job] = (uint32_t)job_tuples;
    }
}

// Pass 0 looks every kept sample's contig base up in the contig -> reference entry map (queryContig): jobs that run at the same
// time should work on the same stretch of the contigs, so that those 4-byte gathers share their sectors in the L2s.  Key of a
// job = single coordinate of the first contig base its read's first alignment on that strand covers (jobs without one: last).
__global__ void exec_keys(ExtractArgs A, uint32_t n_jobs, uint32_t *__restrict__ key, uint64_t *__restrict__ val) {
    const uint32_t job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= n_jobs) return;
    const uint32_t r = A.emit_order[job >> 1], strand = job & 1u;
    uint32_t k = 0xFFFFFFFFu;
    int done = 0;
    for (uint64_t ai = A.query_off[r]; ai < A.query_off[r + 1]; ++ai) {
        if (A.topk >= 0 && done >= A.topk) break;
        const pag_aln al = A.aln[ai];
        if (!(al.flags & PAG_ALN_ELIGIBLE)) continue;
        ++done;
        if (((al.flags & PAG_ALN_REV_STRAND) ? 1u : 0u) != strand) continue;
        if (al.q_start == PAG_NONE || al.n_valid == 0) continue;
        k = A.ctgs[al.target].single_base + al.t_start;
        break;
    }
    key[job] = k;
    val[job] = job;
}
__global__ void exec_perm_narrow(const uint64_t *__restrict__ val, uint32_t n, uint32_t *__restrict__ perm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[i] = (uint32_t)val[i];
}
// perm [2 n_reads]: pass 0's jobs in the order they should run.  key / val / key2 / val2: [2 n_reads] scratch, tmp: sort_tmp_bytes
int launch_exec_perm(const ExtractArgs &a, uint32_t *key, uint64_t *val, uint32_t *key2, uint64_t *val2, void *tmp, uint32_t *perm, hipStream_t s) {
    const uint32_t n_jobs = 2u * a.n_reads;
    if (!n_jobs) return PAG_OK;
    exec_keys<<<dim3((n_jobs + 255) / 256), dim3(256), 0, s>>>(a, n_jobs, key, val);
    int in0 = 1, rc;
    if ((rc = sort_pairs(key, val, key2, val2, n_jobs, 32, tmp, &in0, s, nullptr, nullptr))) return rc;
    exec_perm_narrow<<<dim3((n_jobs + 255) / 256), dim3(256), 0, s>>>(in0 ? val : val2, n_jobs, perm);
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

int launch_extract(const ExtractArgs &a, bool emit, hipStream_t s) {
    if (a.n_reads == 0) return PAG_OK;
    dim3 grid(2u * a.n_reads), block(64);
    if (a.outer >= 3) {
        if (emit) extract_kernel<true, MAXS_SPACED><<<grid, block, 0, s>>>(a);
        else extract_kernel<false, MAXS_SPACED><<<grid, block, 0, s>>>(a);
    } else {
        if (emit) extract_kernel<true, TILE><<<grid, block, 0, s>>>(a);
        else extract_kernel<false, TILE><<<grid, block, 0, s>>>(a);
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}

}  // namespace pagdev
