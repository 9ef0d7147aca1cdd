// Internal header of libpagraph_hip.so: device helpers shared by the kernels + launcher prototypes.
// gfx950 only: wavefront = 64 lanes everywhere (no 32-wide paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pagraph_hip.h"

#define PAG_WAVE 64

namespace pagdev {

// ---------------------------------------------------------------- environment (host side; read where it is asked for: tests set it between calls)
// PAGRAPH_TIMING: stage timers on stderr.  PAG_DEVICE_SHARERS=<n>: processes that share the device (the one-GPU test boxes run
// the ranks of a node that way): grids and memory budgets are a share of it.
bool env_timing();
size_t env_device_sharers();
long long env_int(const char *name, long long otherwise);  // the variable as an integer; `otherwise` when it is not set

// ---------------------------------------------------------------- error plumbing (host side)
void set_error(const char *fmt, ...);
const char *last_error();
#define PAG_HIP_TRY(expr)                                                                 \
    do {                                                                                  \
        hipError_t e__ = (expr);                                                          \
        if (e__ != hipSuccess) {                                                          \
            pagdev::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return PAG_EFAULT;                                                            \
        }                                                                                 \
    } while (0)

// ---------------------------------------------------------------- wave-level primitives (device)
#ifdef __HIPCC__
__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// exclusive prefix sum across the 64 lanes of a wave; *total receives the wave sum
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t *total) {
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d, 64);
        if ((int)lane_id() >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

__device__ __forceinline__ uint64_t wave_excl_sum64(uint64_t v, uint64_t *total) {
    uint64_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t y = __shfl_up(x, d, 64);
        if ((int)lane_id() >= d) x += y;
    }
    *total = __shfl(x, 63, 64);
    return x - v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// "this pointer points to device memory": a pointer loaded from memory is generic to the compiler (flat instructions, which
// count against the LDS counter too); after a cast to the global address space and back its uses become global loads / stores —
// address-space inference starts from such casts, as it does for kernel arguments.  The empty asm keeps the two casts from
// being folded into nothing; "s": the pointer is the same for the whole wave (a job's buffer) and stays in scalar registers.
template <typename T>
__device__ __forceinline__ T *as_global(T *p) {
    const uint64_t x = (uint64_t)p;
    const uint64_t u = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
                       ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
    __attribute__((address_space(1))) T *g = (__attribute__((address_space(1))) T *)u;
    asm volatile("" : "+s"(g));
    return (T *)g;
}

// reverse the order of the sixteen 2-bit groups of a 32-bit word
__device__ __forceinline__ uint32_t rev2(uint32_t x) {
    x = __brev(x);
    return ((x & 0xAAAAAAAAu) >> 1) | ((x & 0x55555555u) << 1);
}
// the k-mer codes of the sixteen positions a 64-bit window of a 2-bit packed strand begins with (kmer2Code: base i of the stream in
// bits 2i..2i+1, the first base of a k-mer most significant in its code): ALL thirty-two groups reversed once, every code then a
// shift and a mask — instead of a shift, a mask and a reversal of its own per position (nine vector instructions each in
// kernels that are bound by their number)
__device__ __forceinline__ void fwd_codes16(uint64_t W, uint32_t k, uint32_t kmask, uint32_t (&code)[16]) {
    const uint64_t R = ((uint64_t)rev2((uint32_t)W) << 32) | (uint64_t)rev2((uint32_t)(W >> 32));  // group g -> group 31 - g
    const uint32_t s0 = 2u * (32u - k);  // base j .. j + k - 1 of the stream = groups 32 - j - k .. 31 - j of R
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j) code[j] = (uint32_t)(R >> (s0 - 2u * j)) & kmask;
}
#endif

// ---------------------------------------------------------------- launchers (host side, defined in *.hip)

// exclusive prefix sum of n u32 values into u64 (out[i] = sum of in[0..i)); *total_dev (device u64) gets the sum.
// tmp must hold scan_tmp_bytes(n) bytes.
size_t scan_tmp_bytes(uint64_t n);
int scan_u32_to_u64(const uint32_t *in, uint64_t *out, uint64_t n, uint64_t *total_dev, void *tmp, hipStream_t s);

// stable LSD radix sort of (u32 key, u64 value) pairs on key bits [first_bit, first_bit + key_bits).  Ping-pongs
// between (k0, v0) and (k1, v1); returns in *result_in_0 which pair holds the result.  tmp: sort_tmp_bytes(n).
// ASYNCHRONOUS on stream s unless ms_dominant_kernel is given (the timing reads events and waits for the stream): the result,
// both pairs and tmp belong to the stream until the caller has ordered behind it — on the same stream, or after a
// synchronisation before the host or another stream touches them.  (Every caller continues on s.)
size_t sort_tmp_bytes(uint64_t n);
int sort_pairs(uint32_t *k0, uint64_t *v0, uint32_t *k1, uint64_t *v1, uint64_t n, int key_bits, void *tmp,
               int *result_in_0, hipStream_t s, float *ms_dominant_kernel, int *n_passes, int first_bit = 0);

struct ExtractArgs {
    // reads
    const uint64_t *read_off;
    const uint32_t *read_len;
    const uint8_t *packed;
    const uint32_t *emit_order;
    const uint32_t *exec_perm;  // job of workgroup b (null: b): the order the jobs RUN in — their outputs go to the places the
                                // emission order gives them whatever it is (k1_extract.hip exec_keys)
    uint32_t n_reads;
    // alignment database of this pass
    const pag_aln *aln;
    const uint64_t *query_off;
    const uint32_t *diff;
    const uint64_t *colidx_off;  // [n_aln] first entry of each alignment in colidx
    const uint2 *colidx;         // per 1024-column chunk (walk order): (emits before, target advances before)
    const uint8_t *cov_ok;       // pass 2 only: 1 = alignment passes the coverage filter; nullptr = all pass
    int pass;                    // 0 = read->contig, 1 = read->reference
    int topk;
    const pag_ctg *ctgs;
    const uint32_t *ctg_ent_off;
    const uint32_t *ctg_ent;
    const pag_ref *refs;
    // solid set
    const uint32_t *solid_bits;
    const uint16_t *solid_mask;  // per table slice, (read word, strand): solid bits of 16 k-mer starts (solid_mask_kernel)
    uint64_t solid_mask_stride;  // entries of one slice's array
    int all_solid;
    uint32_t k;
    uint32_t outer;
    // outputs
    uint32_t *job_samples;  // count mode: per job (2 per read: forward, reverse strand)
    uint32_t *job_tuples;
    const uint64_t *tuple_off;  // emit mode: exclusive prefix over jobs (global, both passes)
    const uint64_t *edge_off;
    uint32_t *tkey;
    uint64_t *tval;
    uint32_t *ekey;
    uint64_t *eval;
    uint32_t job_base;  // index of this pass's first job in the per-job arrays
};
int launch_colidx(const pag_aln *aln, uint64_t n_aln, const uint32_t *diff, const uint64_t *colidx_off, uint2 *colidx,
                  hipStream_t s);
int launch_extract(const ExtractArgs &a, bool emit, hipStream_t s);
// pass 0's jobs in the order they should run: sorted by where on the contigs their read aligns (k1_extract.hip exec_keys)
int launch_exec_perm(const ExtractArgs &a, uint32_t *key, uint64_t *val, uint32_t *key2, uint64_t *val2, void *tmp, uint32_t *perm, hipStream_t s);
int launch_solid_mask(const ExtractArgs &a, const pag_aln *aln2, const uint64_t *qoff2, uint16_t *mask, hipStream_t s);
uint32_t solid_mask_slices();

// coverage filter (pass 2): cov_ok[i] for every alignment
int launch_cov_filter(const pag_aln *aln, uint64_t n_aln, const pag_ref *refs_dev, const pag_ref *refs_host,
                      uint64_t n_refs, uint32_t cov_filter, uint8_t *cov_ok, void *tmp, size_t tmp_bytes,
                      hipStream_t s);
size_t cov_tmp_bytes(const pag_ref *refs_host, uint64_t n_refs);

// cluster + sort positions inside each k-mer segment of the sorted tuple stream (in place in `val`)
constexpr uint32_t SEG_LEADER = 0x80000000u;
struct ClusterOut {
    uint32_t *seg_len;  // [n] at a segment head: its leaders (slots i .. i + seg_len - 1); at the other leader slots SEG_LEADER | offset behind the head; else 0
    uint16_t *cnt;      // [n] abundance of the leader stored at i
    uint64_t *counters; // device: [0] leaders with ctg != 0, [1] all leaders, [2] segments
};
// (wide: the short path takes segments of up to 64 records instead of 32 — for streams whose segments are long, k34_segments.hip)
int launch_cluster(const uint32_t *key, uint64_t *val, uint64_t *scratch, uint64_t n, uint32_t eps, ClusterOut out,
                   uint64_t *long_list, uint32_t *long_count, hipStream_t s, bool wide);

// sort + unique the (to, step, pass-tag) payloads inside each `from` segment of the sorted edge stream
struct EdgeOut {
    uint32_t *seg_len;  // [n] unique edges of the segment starting at i (0 when i is not a head)
    uint64_t *counters; // device: [0] unique (to,step) groups, [1] groups whose first member is pass 1
};
int launch_edges(const uint32_t *key, uint64_t *val, uint64_t *scratch, uint64_t n, EdgeOut out, uint64_t *long_list,
                 uint32_t *long_count, hipStream_t s, bool wide);

}  // namespace pagdev
