/* oracle/pag_oracle.c — TEST INFRASTRUCTURE ONLY (see pag_oracle.h for who may call this).
 *
 * Plain-C restatement of the reference's PAGraph graph build, written in the reference's own shape
 * (per-read per-base position lists, one growable node per solid k-mer, greedy in-place clustering)
 * rather than in the flat sort-based shape of the HIP implementation, so that the two are independent
 * derivations of the same result.  Each function cites the reference lines it follows
 * (paths relative to /root/reference/PAGraph/src/tools/).
 *
 * Pinned against the compiled reference (oracle/_ref/graph_dump) on the fixtures in tests/golden/.
 */
#include "pag_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- small helpers */

typedef struct {
    uint32_t ctg, ref;
} dualpos;

typedef struct {
    uint64_t to; /* dense index of the child k-mer */
    int step;
} adjedge;

typedef struct {
    adjedge *child;
    size_t n_child, cap_child;
    dualpos *pos;
    size_t n_pos, cap_pos;
    uint16_t *cnt; /* std::vector<CountType>: NOT shrunk by cluster(), see node_cluster */
    size_t n_cnt, cap_cnt;
} node;

struct pago_graph {
    uint32_t k;
    uint64_t n_solid;
    uint64_t *codes; /* sorted unique = _kmerIndexArr (graph/PABruijnGraph.cpp:32-37) */
    node *nodes;     /* dense table */
    uint32_t *dense; /* k <= 14: code -> dense index (0xFFFFFFFF = not solid), the reference's unordered_map as a direct
                        table; only a faster searchDenseIndex, the answers are those of the binary search */
    /* optional record of the emitted streams, same encoding as the HIP library's debug hook */
    int dbg;
    uint32_t *dbg_tkey, *dbg_ekey;
    uint64_t *dbg_tval, *dbg_eval;
    uint32_t *dbg_tread, *dbg_eread; /* emission position (index into emit_order) of the read that emitted the record */
    uint32_t dbg_cur;
    size_t dbg_nt, dbg_ct, dbg_ne, dbg_ce;
};

#define GROW(ptr, n, cap, type)                                  \
    do {                                                         \
        if ((n) == (cap)) {                                      \
            (cap) = (cap) ? (cap) * 2 : 4;                       \
            (ptr) = (type *)realloc((ptr), (cap) * sizeof(type)); \
        }                                                        \
    } while (0)

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

/* ---------------------------------------------------------------- k-mer codec */

/* kmer/KmerHelper.hpp acgt(): A/a/other 0, C 1, G 2, T 3 */
static unsigned acgt(char c) {
    switch (c) {
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 0;
    }
}

/* kmer/KmerHelper.cpp:7-25 */
uint64_t pago_kmer_codes(const char *seq, uint64_t len, uint32_t k, uint64_t *out) {
    uint64_t code = 0, n = 0;
    uint64_t mask = k >= 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    for (uint64_t i = 0; i < k && i < len; ++i) code = (code << 2) | acgt(seq[i]);
    if (len >= k) out[n++] = code;
    for (uint64_t i = k; i < len; ++i) {
        code = ((code << 2) | acgt(seq[i])) & mask;
        out[n++] = code;
    }
    return n;
}

/* seq/CompressedSeq.cpp:56-74 toString(forward) on the packed read */
static void read_to_string(const pag_seqs *s, uint64_t id, int forward, char *out) {
    const char *table = forward ? "ACGT" : "TGCA";
    uint64_t n = s->len[id];
    const uint8_t *p = s->packed + s->byte_off[id];
    for (uint64_t i = 0; i < n; ++i) {
        unsigned code = (p[i >> 2] >> ((i & 3) * 2)) & 3u;
        out[forward ? i : n - 1 - i] = table[code];
    }
}

/* ---------------------------------------------------------------- graph object */

/* graph/PABruijnGraph.cpp:10-45: sort + unique of every word the iterator yields */
pago_graph *pago_create(const uint64_t *codes, uint64_t n_codes, uint32_t k) {
    pago_graph *g = (pago_graph *)calloc(1, sizeof(*g));
    g->k = k;
    g->codes = (uint64_t *)malloc((n_codes ? n_codes : 1) * sizeof(uint64_t));
    memcpy(g->codes, codes, n_codes * sizeof(uint64_t));
    qsort(g->codes, n_codes, sizeof(uint64_t), cmp_u64);
    uint64_t m = 0;
    for (uint64_t i = 0; i < n_codes; ++i)
        if (m == 0 || g->codes[m - 1] != g->codes[i]) g->codes[m++] = g->codes[i];
    g->n_solid = m;
    g->nodes = (node *)calloc(m ? m : 1, sizeof(node));
    if (k <= 14 && m < 0xFFFFFFFFull) {
        const uint64_t space = 1ull << (2 * k);
        g->dense = (uint32_t *)malloc(space * sizeof(uint32_t));
        if (g->dense) {
            memset(g->dense, 0xFF, space * sizeof(uint32_t));
            for (uint64_t i = 0; i < m; ++i)
                if (g->codes[i] < space) g->dense[g->codes[i]] = (uint32_t)i;
        }
    }
    return g;
}

static void node_free(node *nd) {
    free(nd->child);
    free(nd->pos);
    free(nd->cnt);
    memset(nd, 0, sizeof(*nd));
}

/* graph/PABruijnGraph.cpp:310-318 resetAllNodes */
int pago_reset(pago_graph *g) {
    for (uint64_t i = 0; i < g->n_solid; ++i) node_free(&g->nodes[i]);
    g->dbg_nt = g->dbg_ne = 0;
    return PAG_OK;
}

void pago_debug_enable(pago_graph *g, int on) { g->dbg = on; }
int pago_debug_stream_sizes(const pago_graph *g, uint64_t *n_tuples, uint64_t *n_edges) {
    *n_tuples = g->dbg_nt;
    *n_edges = g->dbg_ne;
    return PAG_OK;
}
int pago_debug_streams(const pago_graph *g, uint32_t *tkey, uint64_t *tval, uint32_t *ekey, uint64_t *eval) {
    memcpy(tkey, g->dbg_tkey, g->dbg_nt * 4);
    memcpy(tval, g->dbg_tval, g->dbg_nt * 8);
    memcpy(ekey, g->dbg_ekey, g->dbg_ne * 4);
    memcpy(eval, g->dbg_eval, g->dbg_ne * 8);
    return PAG_OK;
}

/* test hook: for every record of the debug streams, the emission position of the read that emitted it */
int pago_debug_stream_reads(const pago_graph *g, uint32_t *tread, uint32_t *eread) {
    memcpy(tread, g->dbg_tread, g->dbg_nt * 4);
    memcpy(eread, g->dbg_eread, g->dbg_ne * 4);
    return PAG_OK;
}

void pago_destroy(pago_graph *g) {
    if (!g) return;
    pago_reset(g);
    free(g->nodes);
    free(g->codes);
    free(g->dense);
    free(g->dbg_tread);
    free(g->dbg_eread);
    free(g->dbg_tkey);
    free(g->dbg_tval);
    free(g->dbg_ekey);
    free(g->dbg_eval);
    free(g);
}

uint64_t pago_solid_count(const pago_graph *g) { return g->n_solid; }

/* graph/PABruijnGraph.cpp:98-104 searchDenseIndex (hash map there, binary search here) */
static int64_t dense_index(const pago_graph *g, uint64_t code) {
    if (g->dense && code < (1ull << (2 * g->k))) return g->dense[code] == 0xFFFFFFFFu ? -1 : (int64_t)g->dense[code];
    uint64_t lo = 0, hi = g->n_solid;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (g->codes[mid] < code) lo = mid + 1;
        else hi = mid;
    }
    return (lo < g->n_solid && g->codes[lo] == code) ? (int64_t)lo : -1;
}

/* ---------------------------------------------------------------- predicates */

/* graph/PABruijnGraph.cpp:379-383 isPosSimilar, one coordinate */
static int coord_similar(uint32_t a, uint32_t b, uint64_t deviation) {
    return a != 0 && b != 0 && (uint64_t)((a > b ? a : b) - (a > b ? b : a)) <= deviation;
}

/* the lambda of mergeKmerPosition, graph/PABruijnGraph.cpp:265-270 */
int pago_cluster_similar(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, uint64_t eps) {
    int s1 = coord_similar(a_ctg, b_ctg, eps) || (a_ctg == 0 && b_ctg == 0);
    int s2 = coord_similar(a_ref, b_ref, eps) || (a_ref == 0 && b_ref == 0);
    return s1 && s2;
}

/* graph/PABruijnGraph.cpp:385-400 isEdgeSimilar */
int pago_edge_similar(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, int dist, uint64_t deviation,
                      double error_rate) {
    uint32_t t_ctg = a_ctg != 0 ? a_ctg + (uint32_t)dist : 0;
    uint32_t t_ref = a_ref != 0 ? a_ref + (uint32_t)dist : 0;
    int s1 = coord_similar(t_ctg, b_ctg, deviation);
    int s2 = coord_similar(t_ref, b_ref, deviation);
    s1 = s1 || (a_ctg != 0 && b_ctg != 0 && fabs(1.0 - ((uint32_t)(b_ctg - a_ctg) * 1.0 / dist)) <= error_rate);
    s2 = s2 || (a_ref != 0 && b_ref != 0 && fabs(1.0 - ((uint32_t)(b_ref - a_ref) * 1.0 / dist)) <= error_rate);
    return (s1 ? 1 : 0) | (s2 ? 2 : 0);
}

/* graph/PABruijnGraph.cpp:143-165 checkPosition (note the un-guarded second ratio test, quirk Q6) */
int pago_check_position(uint32_t a_ctg, uint32_t a_ref, uint32_t b_ctg, uint32_t b_ref, uint32_t dist,
                        uint32_t deviation, double error_rate) {
    int st = pago_edge_similar(a_ctg, a_ref, b_ctg, b_ref, (int)dist, deviation, error_rate);
    int s1 = st & 1, s2 = (st >> 1) & 1;
    s1 = s1 || fabs(1.0 - ((uint32_t)(b_ctg - a_ctg) * 1.0 / dist)) <= error_rate;
    s2 = s2 || fabs(1.0 - ((uint32_t)(b_ref - a_ref) * 1.0 / dist)) <= error_rate;
    enum { Oops, Skip, Good, Excellent, Amazing };
    if (a_ctg == 0 || b_ctg == 0) return s2 ? (b_ctg != 0 ? Excellent : (a_ctg != 0 ? Skip : Good)) : Oops;
    if (a_ref == 0 || b_ref == 0) return s1 ? (b_ref != 0 ? Excellent : Good) : Oops;
    return (s1 && s2) ? Amazing : (s1 ? Excellent : (s2 ? Skip : Oops));
}

/* ---------------------------------------------------------------- node operations */

/* node/KMerAdjNode.tcc:169-173 addPosition(vector): positions and as many count-1 entries */
static void node_add_positions(node *nd, const dualpos *p, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        GROW(nd->pos, nd->n_pos, nd->cap_pos, dualpos);
        nd->pos[nd->n_pos++] = p[i];
    }
    for (size_t i = 0; i < n; ++i) {
        GROW(nd->cnt, nd->n_cnt, nd->cap_cnt, uint16_t);
        nd->cnt[nd->n_cnt++] = 1;
    }
}

static void node_add_child(node *nd, uint64_t to, int step) {
    GROW(nd->child, nd->n_child, nd->cap_child, adjedge);
    nd->child[nd->n_child].to = to;
    nd->child[nd->n_child].step = step;
    nd->n_child++;
}

static int cmp_edge(const void *a, const void *b) {
    const adjedge *x = (const adjedge *)a, *y = (const adjedge *)b;
    if (x->to != y->to) return x->to < y->to ? -1 : 1;
    return x->step < y->step ? -1 : x->step > y->step;
}

/* node/KMerAdjNode.tcc:45-71 removeDuplicate: sort, drop exact duplicates, return how many went */
static size_t node_merge_children(node *nd) {
    if (nd->n_child == 0) return 0;
    qsort(nd->child, nd->n_child, sizeof(adjedge), cmp_edge);
    size_t p = 1;
    for (size_t k = 1; k < nd->n_child; ++k)
        if (cmp_edge(&nd->child[p - 1], &nd->child[k]) != 0) nd->child[p++] = nd->child[k];
    size_t reduce = nd->n_child - p;
    nd->n_child = p;
    return reduce;
}

/* node/KMerAdjNode.tcc:73-112 cluster: greedy leader clustering in insertion order.  Positions and
 * counts are walked in lock step; the position vector is resized to the leader count, the count vector
 * is NOT (so it keeps stale tail entries, all equal to 1).  Counts are u16 and wrap. */
static size_t node_cluster(node *nd, uint64_t eps) {
    size_t p = 0;
    size_t n = nd->n_pos < nd->n_cnt ? nd->n_pos : nd->n_cnt;
    for (size_t it = 0; it < n; ++it) {
        dualpos item = nd->pos[it];
        uint16_t c = nd->cnt[it];
        int similar = 0;
        for (size_t i = 0; i < p; ++i) {
            if (pago_cluster_similar(item.ctg, item.ref, nd->pos[i].ctg, nd->pos[i].ref, eps)) {
                similar = 1;
                nd->cnt[i] = (uint16_t)(nd->cnt[i] + c);
                break;
            }
        }
        if (!similar) {
            nd->pos[p] = item;
            nd->cnt[p] = c;
            ++p;
        }
    }
    size_t reduce = nd->n_pos - p;
    nd->n_pos = p;
    return reduce;
}

typedef struct {
    dualpos pos;
    uint16_t cnt;
} poscnt;

static int cmp_poscnt(const void *a, const void *b) {
    const poscnt *x = (const poscnt *)a, *y = (const poscnt *)b;
    if (x->pos.ctg != y->pos.ctg) return x->pos.ctg < y->pos.ctg ? -1 : 1;
    if (x->pos.ref != y->pos.ref) return x->pos.ref < y->pos.ref ? -1 : 1;
    return 0;
}

/* node/KMerAdjNode.tcc:114-137 sortWithCount: pairs up positions and counts (lock step), sorts by
 * position, and rebuilds both vectors with equal length */
static void node_sort_positions(node *nd) {
    size_t n = nd->n_pos < nd->n_cnt ? nd->n_pos : nd->n_cnt;
    if (n <= 1) { /* nothing to order; both vectors end with the common length all the same */
        nd->n_pos = n;
        nd->n_cnt = n;
        return;
    }
    poscnt *tmp = (poscnt *)malloc((n ? n : 1) * sizeof(poscnt));
    for (size_t i = 0; i < n; ++i) {
        tmp[i].pos = nd->pos[i];
        tmp[i].cnt = nd->cnt[i];
    }
    qsort(tmp, n, sizeof(poscnt), cmp_poscnt);
    for (size_t i = 0; i < n; ++i) {
        nd->pos[i] = tmp[i].pos;
        nd->cnt[i] = tmp[i].cnt;
    }
    nd->n_pos = n;
    nd->n_cnt = n;
    free(tmp);
}

/* ---------------------------------------------------------------- one read strand */

typedef struct {
    uint32_t q;
    dualpos pos;
} qpos;

typedef struct {
    qpos *v;
    size_t n, cap;
} qpos_vec;

static void qv_push(qpos_vec *v, uint32_t q, uint32_t ctg, uint32_t ref) {
    GROW(v->v, v->n, v->cap, qpos);
    v->v[v->n].q = q;
    v->v[v->n].pos.ctg = ctg;
    v->v[v->n].pos.ref = ref;
    v->n++;
}

/* graph/PABruijnGraph.cpp:238-257 addPositionAndEdge + .tcc:5-26 sampleSequence.
 * `items` holds (read position, DualPos) in append order; per-base lists are recovered with a stable
 * counting sort on the read position. */
static void add_position_and_edge(pago_graph *g, const char *seq, uint64_t len, const qpos_vec *items,
                                  uint32_t outer_sample, uint64_t *n_tuples, uint64_t *n_edges, int pass) {
    uint32_t k = g->k;
    if (len < k) return;
    uint64_t n_codes = len - k + 1;
    uint64_t *codes = (uint64_t *)malloc(n_codes * sizeof(uint64_t));
    pago_kmer_codes(seq, len, k, codes);

    /* per-base lists (CSR over read positions, stable) */
    uint64_t *off = (uint64_t *)calloc(len + 1, sizeof(uint64_t));
    for (size_t i = 0; i < items->n; ++i) off[items->v[i].q + 1]++;
    for (uint64_t i = 0; i < len; ++i) off[i + 1] += off[i];
    dualpos *lists = (dualpos *)malloc((items->n ? items->n : 1) * sizeof(dualpos));
    uint64_t *cur = (uint64_t *)malloc((len + 1) * sizeof(uint64_t));
    memcpy(cur, off, (len + 1) * sizeof(uint64_t));
    for (size_t i = 0; i < items->n; ++i) lists[cur[items->v[i].q]++] = items->v[i].pos;
    free(cur);

    /* sampling: candidate iff the base has a position AND the k-mer is solid; keep the first, then
     * every candidate at least outer_sample after the last kept one */
    int64_t last = -1;
    int64_t prev_idx = -1;
    uint64_t prev_pos = 0;
    for (uint64_t i = 0; i < n_codes; ++i) {
        if (off[i + 1] == off[i]) continue;
        int64_t idx = dense_index(g, codes[i]);
        if (idx < 0) continue;
        if (!(last < 0 || i - (uint64_t)last >= outer_sample)) continue;
        last = (int64_t)i;
        node_add_positions(&g->nodes[idx], lists + off[i], (size_t)(off[i + 1] - off[i]));
        *n_tuples += off[i + 1] - off[i];
        if (g->dbg) {
            for (uint64_t x = off[i]; x < off[i + 1]; ++x) {
                if (g->dbg_nt == g->dbg_ct) {
                    g->dbg_ct = g->dbg_ct ? g->dbg_ct * 2 : 1024;
                    g->dbg_tkey = (uint32_t *)realloc(g->dbg_tkey, g->dbg_ct * 4);
                    g->dbg_tval = (uint64_t *)realloc(g->dbg_tval, g->dbg_ct * 8);
                    g->dbg_tread = (uint32_t *)realloc(g->dbg_tread, g->dbg_ct * 4);
                }
                g->dbg_tread[g->dbg_nt] = g->dbg_cur;
                g->dbg_tkey[g->dbg_nt] = (uint32_t)codes[i];
                g->dbg_tval[g->dbg_nt] = ((uint64_t)lists[x].ctg << 32) | lists[x].ref;
                g->dbg_nt++;
            }
        }
        if (prev_idx >= 0) {
            node_add_child(&g->nodes[prev_idx], (uint64_t)idx, (int)(i - prev_pos));
            *n_edges += 1;
            if (g->dbg) {
                if (g->dbg_ne == g->dbg_ce) {
                    g->dbg_ce = g->dbg_ce ? g->dbg_ce * 2 : 1024;
                    g->dbg_ekey = (uint32_t *)realloc(g->dbg_ekey, g->dbg_ce * 4);
                    g->dbg_eval = (uint64_t *)realloc(g->dbg_eval, g->dbg_ce * 8);
                    g->dbg_eread = (uint32_t *)realloc(g->dbg_eread, g->dbg_ce * 4);
                }
                g->dbg_eread[g->dbg_ne] = g->dbg_cur;
                g->dbg_ekey[g->dbg_ne] = (uint32_t)g->codes[prev_idx];
                g->dbg_eval[g->dbg_ne] = ((uint64_t)codes[i] << 32) | ((uint64_t)(i - prev_pos) << 1) | (uint64_t)pass;
                g->dbg_ne++;
            }
        }
        prev_idx = idx;
        prev_pos = i;
    }
    free(lists);
    free(off);
    free(codes);
}

/* ---------------------------------------------------------------- the two passes */

static unsigned col_class(const pag_aln_db *db, const pag_aln *a, uint64_t c) {
    return (db->diff[a->diff_off + (c >> 4)] >> ((c & 15) * 2)) & 3u;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* position/PositionProcessor.cpp:79-151 process() */
int pago_process(pago_graph *g, const pag_build_input *in, pag_build_stats *st) {
    if (in->on_device) return PAG_EINVAL;
    memset(st, 0, sizeof(*st));
    const pag_seqs *reads = &in->reads;

    /* align/Aligner.cpp:58-88 covInfHelper: per-base read coverage of each reference, THEN SORTED
     * ascending (quirk Q3) */
    uint32_t **cov = (uint32_t **)calloc(in->n_refs ? in->n_refs : 1, sizeof(uint32_t *));
    for (uint64_t r = 0; r < in->n_refs; ++r) cov[r] = (uint32_t *)calloc(in->refs[r].len ? in->refs[r].len : 1, 4);
    for (uint64_t i = 0; i < in->read_to_ref.n_aln; ++i) {
        const pag_aln *a = &in->read_to_ref.aln[i];
        if (a->target == PAG_NONE) continue;
        for (uint64_t j = a->t_begin; j < a->t_end; ++j) {
            if (j >= in->refs[a->target].len) break;
            cov[a->target][j]++;
        }
    }
    for (uint64_t r = 0; r < in->n_refs; ++r) qsort(cov[r], in->refs[r].len, 4, cmp_u32);

    uint64_t max_len = 0;
    for (uint64_t r = 0; r < reads->n_seqs; ++r)
        if (reads->len[r] > max_len) max_len = reads->len[r];
    char *seq = (char *)malloc(max_len + 1);
    qpos_vec items[2] = {{0, 0, 0}, {0, 0, 0}};

    for (int pass = 0; pass < 2; ++pass) {
        const pag_aln_db *db = pass == 0 ? &in->read_to_ctg : &in->read_to_ref;
        int topk = pass == 0 ? in->topk_ctg : in->topk_ref;
        for (uint64_t e = 0; e < reads->n_seqs; ++e) {
            uint32_t r = in->emit_order[e];
            uint64_t len = reads->len[r];
            g->dbg_cur = (uint32_t)e;
            items[0].n = items[1].n = 0;
            int useful[2] = {0, 0};
            int done = 0;
            for (uint64_t ai = db->query_off[r]; ai < db->query_off[r + 1]; ++ai) {
                if (topk >= 0 && done >= topk) break;
                const pag_aln *a = &db->aln[ai];
                if (!(a->flags & PAG_ALN_ELIGIBLE)) continue;
                int strand = (a->flags & PAG_ALN_REV_STRAND) ? 1 : 0;
                int back = (a->flags & PAG_ALN_WALK_BACK) ? 1 : 0;
                if (pass == 1) {
                    /* align/Aligner.tcc:140-149: max of the SORTED coverage over the target interval */
                    uint32_t max_cov = 0;
                    for (uint64_t p = a->t_begin; p < a->t_end; ++p) {
                        if (p >= in->refs[a->target].len) break;
                        if (cov[a->target][p] > max_cov) max_cov = cov[a->target][p];
                    }
                    if (max_cov < in->cov_filter) continue;
                    useful[strand] = 1;
                }
                /* align/ParseAlignTools.tcc:44-70 exactAlign */
                /* q_start == PAG_NONE: the (flipped) read begin lies outside the read, nothing can pass
                 * the `curRead < positions.size()` guard (align/Aligner.tcc:81, :158) */
                uint64_t q = a->q_start, t = a->t_start;
                if (a->q_start != PAG_NONE) {
                    for (uint64_t jj = 0; jj < a->n_cols; ++jj) {
                        unsigned cls = col_class(db, a, back ? a->n_cols - jj - 1 : jj);
                        int emit = cls != 1, radv = cls != 2;
                        if (emit && q < len) {
                            if (pass == 0) {
                                /* align/Aligner.cpp:222-233 queryContig via AlignReference::query */
                                const pag_ctg *c = &in->ctgs[a->target];
                                if (t < c->len) {
                                    uint32_t lo = in->ctg_ent_off[c->map_off + t], hi = in->ctg_ent_off[c->map_off + t + 1];
                                    for (uint32_t x = lo; x < hi; ++x)
                                        qv_push(&items[strand], (uint32_t)q, (uint32_t)(c->single_base + (uint32_t)t),
                                                in->ctg_ent[x]);
                                    if (hi > lo) useful[strand] = 1;
                                }
                            } else {
                                qv_push(&items[strand], (uint32_t)q, 0,
                                        (uint32_t)(in->refs[a->target].single_base + (uint32_t)t));
                            }
                        }
                        if (emit) ++q;
                        if (radv) ++t;
                    }
                }
                ++done;
            }
            /* the functor of process() (position/PositionProcessor.cpp:90-115): forward, then reverse */
            for (int s = 0; s < 2; ++s) {
                if (!useful[s]) continue;
                read_to_string(reads, r, s == 0, seq);
                add_position_and_edge(g, seq, len, &items[s], in->outer_sample, &st->n_tuples[pass], &st->n_edges[pass], pass);
            }
        }
        /* mergeEdge / totalPosition / mergeKmerPosition (graph/PABruijnGraph.cpp:285-297, 320-331, 259-274) */
        uint64_t me = 0, tp = 0, mp = 0;
        for (uint64_t i = 0; i < g->n_solid; ++i) me += node_merge_children(&g->nodes[i]);
        for (uint64_t i = 0; i < g->n_solid; ++i) tp += g->nodes[i].n_pos;
        for (uint64_t i = 0; i < g->n_solid; ++i) mp += node_cluster(&g->nodes[i], in->eps);
        st->merge_edge[pass] = me;
        st->total_pos[pass] = tp;
        st->merge_pos[pass] = mp;
    }
    /* sortKmerPosition (graph/PABruijnGraph.cpp:276-283) */
    for (uint64_t i = 0; i < g->n_solid; ++i) node_sort_positions(&g->nodes[i]);

    for (uint64_t i = 0; i < g->n_solid; ++i) {
        if (g->nodes[i].n_pos || g->nodes[i].n_child) st->n_nodes++;
        st->n_pos += g->nodes[i].n_pos;
        st->n_uniq_edges += g->nodes[i].n_child;
    }
    free(items[0].v);
    free(items[1].v);
    free(seq);
    for (uint64_t r = 0; r < in->n_refs; ++r) free(cov[r]);
    free(cov);
    return PAG_OK;
}

int pago_csr_sizes(const pago_graph *g, uint64_t *n_nodes, uint64_t *n_pos, uint64_t *n_edges) {
    uint64_t nn = 0, np = 0, ne = 0;
    for (uint64_t i = 0; i < g->n_solid; ++i) {
        if (g->nodes[i].n_pos || g->nodes[i].n_child) nn++;
        np += g->nodes[i].n_pos;
        ne += g->nodes[i].n_child;
    }
    *n_nodes = nn;
    *n_pos = np;
    *n_edges = ne;
    return PAG_OK;
}

int pago_export_csr(const pago_graph *g, pag_csr *out) {
    uint64_t nn, np, ne;
    pago_csr_sizes(g, &nn, &np, &ne);
    if (out->n_nodes < nn || out->n_pos < np || out->n_edges < ne) {
        out->n_nodes = nn;
        out->n_pos = np;
        out->n_edges = ne;
        return PAG_ERANGE;
    }
    uint64_t in = 0, ip = 0, ie = 0;
    for (uint64_t i = 0; i < g->n_solid; ++i) {
        const node *nd = &g->nodes[i];
        if (!nd->n_pos && !nd->n_child) continue;
        out->node_code[in] = (uint32_t)g->codes[i];
        out->pos_off[in] = ip;
        out->edge_off[in] = ie;
        for (size_t j = 0; j < nd->n_pos; ++j) {
            out->pos_ctg[ip] = nd->pos[j].ctg;
            out->pos_ref[ip] = nd->pos[j].ref;
            out->pos_cnt[ip] = nd->cnt[j];
            ++ip;
        }
        for (size_t j = 0; j < nd->n_child; ++j) {
            out->edge_to[ie] = (uint32_t)g->codes[nd->child[j].to];
            out->edge_step[ie] = nd->child[j].step;
            ++ie;
        }
        ++in;
    }
    out->pos_off[in] = ip;
    out->edge_off[in] = ie;
    out->n_nodes = nn;
    out->n_pos = np;
    out->n_edges = ne;
    return PAG_OK;
}

/* ================================================================ kmer_counter
 * kmer_counter.cpp:19-96.  One dense size_t table like the reference's (the four partial tables are summed there
 * before use, :59-66); abundances visited through a sorted list of the occurring values like its std::map. */

int pago_kmer_count(const pag_seqs *reads, uint32_t k, double threshold, uint64_t *min_abundance, uint32_t *bitmap) {
    if (!reads || !bitmap || k < 1 || k > 15) return PAG_EINVAL;
    const uint64_t n_codes = 1ull << (2 * k);
    const uint64_t mask = n_codes - 1;
    uint64_t *table = (uint64_t *)calloc(n_codes, sizeof(uint64_t));
    if (!table) return PAG_ENOMEM;
    for (uint64_t r = 0; r < reads->n_seqs; ++r) {
        const uint8_t *p = reads->packed + reads->byte_off[r];
        const uint64_t len = reads->len[r];
        /* KmerHelper::kmer2Code (KmerHelper.cpp:7-25) on the unpacked bases */
        uint64_t code = 0;
        for (uint64_t i = 0; i < len; ++i) {
            const uint64_t b = (p[i >> 2] >> (2 * (i & 3))) & 3u;
            code = ((code << 2) | b) & mask;
            if (i + 1 >= k) ++table[code];
        }
    }
    /* mergeMap (:57-66): occurring abundances with their multiplicities, ascending */
    uint64_t *sorted = (uint64_t *)malloc(n_codes * sizeof(uint64_t));
    if (!sorted) {
        free(table);
        return PAG_ENOMEM;
    }
    memcpy(sorted, table, n_codes * sizeof(uint64_t));
    qsort(sorted, n_codes, sizeof(uint64_t), cmp_u64);
    uint64_t sum = 0, min_ab = 0;
    for (uint64_t i = 0; i < n_codes;) {
        uint64_t j = i;
        while (j < n_codes && sorted[j] == sorted[i]) ++j;
        sum += j - i;
        if (1 - sum * 1.0 / (double)n_codes <= threshold) { /* :71-76 */
            min_ab = sorted[i];
            break;
        }
        i = j;
    }
    free(sorted);
    const uint64_t n_words = (n_codes + 31) / 32;
    memset(bitmap, 0, n_words * 4);
    for (uint64_t c = 0; c < n_codes; ++c)
        if (table[c] >= min_ab) bitmap[c >> 5] |= 1u << (c & 31);
    free(table);
    if (min_abundance) *min_abundance = min_ab;
    return PAG_OK;
}

uint64_t pago_kmer_file_words(const uint32_t *bitmap, uint32_t k, uint32_t threads, uint64_t *out, uint64_t cap) {
    const uint64_t n_codes = 1ull << (2 * k);
    uint64_t n = 0;
    if (out && n < cap) out[n] = k;
    ++n;
    if (threads == 0) threads = 1;
    for (uint32_t t = 0; t < threads; ++t) /* MultiThreadTools::traversalHelper: i = t, t + T, ... */
        for (uint64_t c = t; c < n_codes; c += threads)
            if ((bitmap[c >> 5] >> (c & 31)) & 1u) {
                if (out && n < cap) out[n] = c;
                ++n;
            }
    return n;
}
