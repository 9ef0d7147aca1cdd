// k5_walk_aux.hip — what runs around the walker waves (k5_travel.hip): id ranges of the strands, seed scans (searchPANode /
// searchPANode2, PAGraph/src/tools/graph/PAlgorithm.tcc:300-365), checkpoints of the segment jobs, the kernels that hand finished
// paths to the host (packed with their block tables, gathered into pag_path_node records), the global visited marks of a
// committed walk, the clearing of a batch of job buffers, the last round of a leaping contig put together on the device.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "pag_device.hpp"
#include "pag_travel.hpp"
#include "trav_device.hpp"

namespace pagdev {


// [first, last) new-id range of the vertices whose contig coordinate lies in [lo, hi)
__global__ void k_ranges(TravGraph G, TravContig *ctgs, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    auto lower = [&](uint32_t x) {
        uint64_t lo = 0, hi = G.n_pos;
        while (lo < hi) {
            uint64_t mid = (lo + hi) >> 1;
            if ((uint32_t)(G.upos[mid] >> 32) < x) lo = mid + 1;
            else hi = mid;
        }
        return (uint32_t)lo;
    };
    ctgs[i].in_lo = lower(ctgs[i].ctg_left);
    ctgs[i].in_hi = lower(ctgs[i].ctg_right);
    ctgs[i].g_lo = ctgs[i].in_lo;
    ctgs[i].g_hi = ctgs[i].in_hi;
}

// first new id whose contig coordinate is >= coords[i] (the vertices with a coordinate are ordered by it)
__global__ void k_id_bounds(TravGraph G, const uint32_t *__restrict__ coords, uint32_t n, uint32_t *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = coords[i];
    uint64_t lo = 0, hi = G.n_pos;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if ((uint32_t)(G.upos[mid] >> 32) < x) lo = mid + 1;
        else hi = mid;
    }
    out[i] = (uint32_t)lo;
}

// =================================================================================================
// seeds
// =================================================================================================
// searchPANode(onlyFirst = true) (PAlgorithm.tcc:300-327): the first contig k-mer one of whose positions
// lies on this contig strand within `dev` of the k-mer's own offset; all such positions of that k-mer.
// One wave per contig.  out[0] = count, then (vertex, node) pairs.
__global__ __launch_bounds__(64) void k_seed_first(TravGraph G, const TravContig *__restrict__ ctgs, uint32_t n_ctgs_sel,
                                                   uint64_t dev, uint32_t *__restrict__ out, uint32_t out_stride) {
    const uint32_t c = blockIdx.x;
    if (c >= n_ctgs_sel) return;
    const TravContig C = ctgs[c];
    const uint32_t lane = lane_id();
    uint32_t *o = out + (uint64_t)c * out_stride;
    uint32_t found = 0;
    for (uint32_t base = 0; base < C.n_kmers && !found; base += 64) {
        uint32_t i = base + lane;
        bool hit = false;
        uint32_t node = PAG_NONE;
        if (i < C.n_kmers) {
            node = C.nodes[i];
            if (node != PAG_NONE) {
                for (uint32_t p = G.npos_off[node]; p < G.npos_off[node + 1] && !hit; ++p) {
                    uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
                    if (pc >= C.ctg_left && pc < C.ctg_right) {
                        uint64_t off = pc - C.ctg_left;
                        uint64_t d = off > i ? off - i : (uint64_t)i - off;
                        hit = d <= dev;
                    }
                }
            }
        }
        uint64_t m = __ballot(hit);
        if (m) {
            int src = __ffsll((long long)m) - 1;
            if ((int)lane == src) {
                uint32_t n = 0;
                for (uint32_t p = G.npos_off[node]; p < G.npos_off[node + 1]; ++p) {
                    uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
                    if (pc >= C.ctg_left && pc < C.ctg_right) {
                        uint64_t off = pc - C.ctg_left;
                        uint64_t d = off > i ? off - i : (uint64_t)i - off;
                        if (d <= dev && 1 + 2 * n + 1 < out_stride) {
                            o[1 + 2 * n] = p;
                            o[2 + 2 * n] = i;
                            ++n;
                        }
                    }
                }
                o[0] = n;
            }
            found = 1;
        }
    }
    if (!found && lane == 0) o[0] = 0;
}

// searchPANode2 (PAlgorithm.tcc:329-365): every (contig offset in [left, right], position) pair whose
// position lies on this strand within `dev` of `pos`, in order.  Duplicates of a vertex are removed on
// the host (first occurrence wins).  TRAV_SEED_PARTS waves per request, each scanning one part of the offset range
// (the window spans 1000 x deviation offsets on either side); part p of request r writes out[(r * PARTS + p) * stride]:
// [0] = count, then vertex ids; the host concatenates the parts in order.
__global__ __launch_bounds__(64) void k_seed_window(TravGraph G, const TravContig *__restrict__ ctgs,
                                                    const TravSeedReq *__restrict__ reqs, uint32_t n_req, uint64_t dev,
                                                    uint32_t *__restrict__ out, uint32_t out_stride) {
    const uint32_t r = blockIdx.x, part = blockIdx.y;
    if (r >= n_req) return;
    const TravSeedReq R = reqs[r];
    const TravContig C = ctgs[R.ctg];
    const uint32_t lane = lane_id();
    uint32_t *o = out + ((uint64_t)r * TRAV_SEED_PARTS + part) * out_stride;
    uint32_t n_out = 0;
    const uint64_t right_all = R.right < (uint64_t)C.n_kmers ? R.right + 1 : C.n_kmers;  // exclusive
    const uint64_t span = right_all > R.left ? right_all - R.left : 0;
    const uint64_t per = ((span + TRAV_SEED_PARTS - 1) / TRAV_SEED_PARTS + 63) & ~63ull;  // offsets per part (whole wave rows)
    const uint64_t left = R.left + (uint64_t)part * per;
    const uint64_t right = left + per < right_all ? left + per : right_all;
    for (uint64_t base = left; base < right; base += 64) {
        uint64_t i = base + lane;
        uint32_t node = i < right ? C.nodes[i] : PAG_NONE;
        uint32_t p0 = 0, p1 = 0;
        if (node != PAG_NONE) {
            p0 = G.npos_off[node];
            p1 = G.npos_off[node + 1];
        }
        // lanes emit in lane order, positions in order: serialise over the lanes that have matches
        // filterPANodes (PAlgorithm.cpp:97-105): vertices of the contig's globalUniqueTable are dropped here, on the device
        // (a per-vertex predicate: applying it before the host removes duplicates gives the same list)
        auto visited = [&](uint32_t p) -> bool {
            if (!C.gbits) return false;
            const uint32_t u = G.newid[p];
            if (u >= C.g_lo && u < C.g_hi) return ((C.gbits[(u - C.g_lo) >> 5] >> ((u - C.g_lo) & 31u)) & 1u) != 0u;
            return C.gset ? hs_has(C.gset, C.gmask, u) : false;
        };
        uint32_t cnt = 0;
        for (uint32_t p = p0; p < p1; ++p) {
            uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc >= C.ctg_left && pc < C.ctg_right) {
                uint64_t off = pc - C.ctg_left;
                uint64_t d = off > R.pos ? off - R.pos : R.pos - off;
                cnt += (d <= dev && !visited(p)) ? 1u : 0u;
            }
        }
        uint32_t tot;
        uint32_t ex = wave_excl_sum(cnt, &tot);
        uint32_t w = n_out + ex;
        for (uint32_t p = p0; p < p1 && cnt; ++p) {
            uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc >= C.ctg_left && pc < C.ctg_right) {
                uint64_t off = pc - C.ctg_left;
                uint64_t d = off > R.pos ? off - R.pos : R.pos - off;
                if (d <= dev && !visited(p)) {
                    if (1 + w < out_stride) o[1 + w] = p;
                    ++w;
                }
            }
        }
        n_out += tot;
    }
    if (lane == 0) o[0] = n_out;
}

// Checkpoints of the segment-parallel walk (k5_travel_host.hip): for every request (contig, contig offset) the most
// abundant vertex that lies ON the contig strand (its k-mer is the contig's k-mer at offset i and its contig coordinate
// is within `dev` of i, like a seed of searchPANode) for i in [left, right], and that is not in the contig's global
// visited set.  Ties: the lowest offset, then position order.  One wave per request; out = (old vertex id, contig
// coordinate, abundance) or (PAG_NONE, 0, 0).  Which vertex is picked has no influence on the results of the
// traversal, only on how soon the walk that arrives from behind meets the piece started here.
__global__ __launch_bounds__(64) void k_checkpoints(TravGraph G, const TravContig *__restrict__ ctgs, const TravSeedReq *__restrict__ reqs,
                                                    uint32_t n_req, uint64_t dev, uint32_t *__restrict__ out) {
    const uint32_t r = blockIdx.x;
    if (r >= n_req) return;
    const TravSeedReq R = reqs[r];
    const TravContig C = ctgs[R.ctg];
    const uint32_t lane = lane_id();
    const uint64_t right = R.right < (uint64_t)C.n_kmers ? R.right + 1 : C.n_kmers;  // exclusive
    uint64_t best = 0;  // abundance << 40 | (0xFFFFF - (offset - left)) << 20 | (0xFFFFF - position rank): larger is better
    uint32_t best_v = PAG_NONE, best_pc = 0;
    for (uint64_t base = R.left; base < right; base += 64) {
        const uint64_t i = base + lane;
        const uint32_t node = i < right ? C.nodes[i] : PAG_NONE;
        if (node == PAG_NONE) continue;
        const uint32_t p0 = G.npos_off[node], p1 = G.npos_off[node + 1];
        for (uint32_t p = p0; p < p1; ++p) {
            const uint32_t pc = (uint32_t)(G.vpos[p] >> 32);
            if (pc < C.ctg_left || pc >= C.ctg_right) continue;
            const uint64_t off = pc - C.ctg_left;
            const uint64_t d = off > i ? off - i : i - off;
            if (d > dev) continue;
            const uint32_t u = G.newid[p];
            if (C.gbits && u >= C.g_lo && u < C.g_hi && ((C.gbits[(u - C.g_lo) >> 5] >> ((u - C.g_lo) & 31u)) & 1u)) continue;
            const uint64_t key = ((uint64_t)G.vcnt[p] << 40) | ((uint64_t)(0xFFFFFu - (uint32_t)((i - R.left) & 0xFFFFFu)) << 20) |
                                 (uint64_t)(0xFFFFFu - ((p - p0) & 0xFFFFFu));
            if (key > best) {
                best = key;
                best_v = p;
                best_pc = pc;
            }
        }
    }
    for (int d2 = 32; d2 >= 1; d2 >>= 1) {
        const uint64_t ob = __shfl_xor(best, d2, 64);
        const uint32_t ov = (uint32_t)__shfl_xor((int)best_v, d2, 64), op = (uint32_t)__shfl_xor((int)best_pc, d2, 64);
        if (ob > best) {
            best = ob;
            best_v = ov;
            best_pc = op;
        }
    }
    if (lane == 0) {
        out[3 * r] = best_v;
        out[3 * r + 1] = best_pc;
        out[3 * r + 2] = (uint32_t)(best >> 40);
    }
}

// The new parts of the sequences of a batch of finished jobs, packed for ONE copy to the host: per job its vertices (new
// ids), its steps and the contig coordinates of its vertices (+ the two words of the iteration log of a TRAV_MODE_LEAP
// job), each `len` words, at out + off — and behind them the job's BLOCK TABLES (walk_stitch.hpp: AGG_WORDS words per 64
// entries, + AGG_XWORDS for a leap job): a wave copies 64 consecutive entries per turn and reduces them while it holds them.
__global__ void k_pack_paths(TravGraph G, const TravPackDesc *__restrict__ descs, uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t j = blockIdx.y;
    if (j >= n) return;
    const TravPackDesc D = descs[j];
    uint32_t *o = out + D.off;
    const uint32_t lane = lane_id();
    const uint64_t n_arrays = D.seq_x ? 5 : 3;
    uint32_t *agg = o + n_arrays * D.len;
    uint32_t *xagg = agg + ((D.len + 63) / 64) * 5;
    // (wave-uniform loop: every lane of a wave takes part in the reductions of its block)
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < D.len; i0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = i0 + lane;
        const bool valid = i < D.len;
        uint32_t v = 0, st = 0, c = 0, xl = 0, xh = 0;
        if (valid) {
            v = D.seq_v[i];
            st = D.seq_s[i];
            c = (uint32_t)(G.upos[v] >> 32);
            o[i] = v;
            o[D.len + i] = st;
            o[2 * D.len + i] = c;
            if (D.seq_x) {
                const uint64_t x = D.seq_x[i];
                xl = (uint32_t)x;
                xh = (uint32_t)(x >> 32);
                o[3 * D.len + i] = xl;
                o[4 * D.len + i] = xh;
            }
        }
        const uint32_t mx = wave_max_u32(valid ? c : 0u);
        const uint32_t m0 = wave_max_u32(valid && c == 0u ? v + 1u : 0u);
        const uint32_t lo = wave_min_u32(valid ? c : 0xFFFFFFFFu);
        const uint32_t lnz = wave_min_u32(valid && c != 0u ? c : 0xFFFFFFFFu);
        const uint32_t sum = wave_sum(valid ? st : 0u);
        const uint64_t blk = i0 >> 6;
        if (lane == 0) {
            uint32_t *a = agg + blk * 5;
            a[0] = mx;
            a[1] = m0;
            a[2] = lo;
            a[3] = lnz;
            a[4] = sum;
        }
        if (D.seq_x) {
            const bool bd = valid && (xh >> 31) != 0u;
            const uint32_t elow = wave_min_u32(bd ? (xh & 0x7FFFFFFFu) : 0xFFFFFFFFu);
            const uint32_t xm0 = wave_min_u32(bd ? xl : 0xFFFFFFFFu);
            if (lane == 0) {
                xagg[blk * 2] = elow;
                xagg[blk * 2 + 1] = xm0;
            }
        }
    }
}

// contig coordinates of a path (new ids) for the host-side stitch
__global__ void k_gather_pc(TravGraph G, const uint32_t *__restrict__ seq_v, uint64_t len, uint32_t *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = (uint32_t)(G.upos[seq_v[i]] >> 32);
}

// record a finished walk (new ids) in the contig's global visited structures
__global__ void k_commit(const uint32_t *__restrict__ seq_v, uint64_t len, uint32_t in_lo, uint32_t in_hi, uint32_t *gbits,
                         uint32_t *gset, uint32_t gmask) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t u = seq_v[i];
        if (u >= in_lo && u < in_hi) atomicOr(&gbits[(u - in_lo) >> 5], 1u << ((u - in_lo) & 31u));
        else hs_insert(gset, gmask, u);
    }
}

// vertex attributes of a path (new ids) for the host
__global__ void k_gather_path(TravGraph G, const uint32_t *__restrict__ seq_v, const uint32_t *__restrict__ seq_s, uint64_t len,
                              pag_path_node *__restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t v = G.uold[seq_v[i]];
        uint64_t p = G.vpos[v];
        pag_path_node o;
        o.code = G.ncode[G.vnode[v]];
        o.ctg = (uint32_t)(p >> 32);
        o.ref = (uint32_t)p;
        o.cnt = G.vcnt[v];
        o.reserved = 0;
        o.step = (int32_t)seq_s[i];
        o.vid = v;
        out[i] = o;
    }
}

__global__ void k_gather_vertices(TravGraph G, const uint32_t *__restrict__ vids, uint32_t n, pag_path_node *__restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = vids[i];
    uint64_t p = G.vpos[v];
    pag_path_node o;
    o.code = G.ncode[G.vnode[v]];
    o.ctg = (uint32_t)(p >> 32);
    o.ref = (uint32_t)p;
    o.cnt = G.vcnt[v];
    o.reserved = 0;
    o.step = 0;
    o.vid = v;
    out[i] = o;
}

void trav_launch_seed_first(TravGraph G, const TravContig *ctgs, uint32_t n, uint64_t dev, uint32_t *out, uint32_t stride,
                            hipStream_t s) {
    if (n) k_seed_first<<<dim3(n), dim3(64), 0, s>>>(G, ctgs, n, dev, out, stride);
}
void trav_launch_seed_window(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev,
                             uint32_t *out, uint32_t stride, hipStream_t s) {
    if (n) k_seed_window<<<dim3(n, TRAV_SEED_PARTS), dim3(64), 0, s>>>(G, ctgs, reqs, n, dev, out, stride);
}
void trav_launch_checkpoints(TravGraph G, const TravContig *ctgs, const TravSeedReq *reqs, uint32_t n, uint64_t dev, uint32_t *out,
                             hipStream_t s) {
    if (n) k_checkpoints<<<dim3(n), dim3(64), 0, s>>>(G, ctgs, reqs, n, dev, out);
}
void trav_launch_id_bounds(TravGraph G, const uint32_t *coords, uint32_t n, uint32_t *out, hipStream_t s) {
    if (n) k_id_bounds<<<dim3((n + 63) / 64), dim3(64), 0, s>>>(G, coords, n, out);
}
void trav_launch_pack_paths(TravGraph G, const TravPackDesc *descs, uint32_t n, uint64_t max_len, uint32_t *out, hipStream_t s) {
    if (!n) return;
    const unsigned gx = (unsigned)std::min<uint64_t>(std::max<uint64_t>((max_len + 255) / 256, 1), 64);
    k_pack_paths<<<dim3(gx, n), dim3(256), 0, s>>>(G, descs, n, out);
}
void trav_launch_gather_pc(TravGraph G, const uint32_t *seq_v, uint64_t len, uint32_t *out, hipStream_t s) {
    if (len) k_gather_pc<<<dim3(grid_for(len)), dim3(256), 0, s>>>(G, seq_v, len, out);
}
void trav_launch_commit(const uint32_t *seq_v, uint64_t len, uint32_t in_lo, uint32_t in_hi, uint32_t *gbits, uint32_t *gset,
                        uint32_t gmask, hipStream_t s) {
    if (len) k_commit<<<dim3(grid_for(len)), dim3(256), 0, s>>>(seq_v, len, in_lo, in_hi, gbits, gset, gmask);
}
void trav_launch_ranges(TravGraph G, TravContig *ctgs, uint32_t n, hipStream_t s) {
    if (n) k_ranges<<<dim3((n + 63) / 64), dim3(64), 0, s>>>(G, ctgs, n);
}

// blockIdx.y = range; 16 bytes per lane and turn where the range allows (the job buffers are 256-byte aligned), bytes at its edges
__global__ void k_clear_ranges(const TravClear *__restrict__ ranges) {
    const TravClear c = ranges[blockIdx.y];
    uint8_t *p = as_global((uint8_t *)c.p);  // (device memory: global stores, not flat ones)
    const uint64_t head = (16u - ((uintptr_t)p & 15u)) & 15u, h = head < c.bytes ? head : c.bytes;
    const uint64_t n16 = (c.bytes - h) >> 4, tail = (c.bytes - h) & 15u;
    uint4 *q = (uint4 *)(p + h);
    const uint4 w = make_uint4(c.word, c.word, c.word, c.word);
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) q[i] = w;
    if (blockIdx.x == 0) {
        if (threadIdx.x < h) p[threadIdx.x] = (uint8_t)c.word;
        if (threadIdx.x < tail) p[h + (n16 << 4) + threadIdx.x] = (uint8_t)c.word;
    }
}
int trav_clear_ranges(const TravClear *ranges_dev, size_t n, hipStream_t s) {
    for (size_t at = 0; at < n; at += 32768) {
        const uint32_t m = (uint32_t)std::min<size_t>(32768, n - at);
        k_clear_ranges<<<dim3(32, m), dim3(256), 0, s>>>(ranges_dev + at);
    }
    PAG_HIP_TRY(hipGetLastError());
    return PAG_OK;
}
void trav_launch_gather_path(TravGraph G, const uint32_t *seq_v, const uint32_t *seq_s, uint64_t len, pag_path_node *out,
                             hipStream_t s, unsigned max_blocks) {
    const unsigned grid = max_blocks ? std::min(grid_for(len), max_blocks) : grid_for(len);
    if (len) k_gather_path<<<dim3(grid), dim3(256), 0, s>>>(G, seq_v, seq_s, len, out);
}
// blockIdx.y = the part; the blocks of a row stride over its entries
__global__ void __launch_bounds__(256) k_concat_parts(const TravConcatPart *__restrict__ parts, uint32_t *__restrict__ out_v,
                                                      uint32_t *__restrict__ out_s, uint32_t first_step) {
    const TravConcatPart P = parts[blockIdx.y];
    for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < P.n; x += (uint64_t)gridDim.x * blockDim.x) {
        out_v[P.start + x] = P.v[x];
        out_s[P.start + x] = (P.start + x == 0) ? first_step : P.s[x];
    }
}
void trav_launch_concat_parts(const TravConcatPart *parts, uint32_t n_parts, uint32_t *out_v, uint32_t *out_s, uint32_t first_step,
                              hipStream_t s) {
    for (uint32_t at = 0; at < n_parts; at += 32768) {
        const uint32_t n = std::min<uint32_t>(32768, n_parts - at);
        k_concat_parts<<<dim3(16, n), dim3(256), 0, s>>>(parts + at, out_v, out_s, first_step);
    }
}
void trav_launch_gather_vertices(TravGraph G, const uint32_t *vids, uint32_t n, pag_path_node *out, hipStream_t s) {
    if (n) k_gather_vertices<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(G, vids, n, out);
}

}  // namespace pagdev
