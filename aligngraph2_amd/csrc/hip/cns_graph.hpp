// cns_graph.hpp — the partial-order alignment graph of ONE backbone part of pa_cns on flat arrays: the code
// the device kernel runs (k_cns.hip, one thread per part), written so that a host compiler takes it too (tests/harness/
// cns_graph_test.cpp runs the very same functions on the CPU against the goldens).
//
// Reference semantics (PAGraph/src/tools/cns/AlnGraphBoost.cpp): AlnGraphBoost(backbone) :16-39, addAln :64-113, addEdge
// :115-135, mergeNodes :137-168, mergeInNodes :170-223, mergeOutNodes :225-275, markForReaper :277-281, consensus :293-333,
// bestPath :383-467.  What decides ties there, and how it is kept here:
//   * the graph is boost::adjacency_list<vecS, vecS, bidirectionalS>: out- and in-edge lists are vectors in insertion order,
//     clear_vertex() erases entries in place, edge(u, v) finds the first match in u's out list, add_edge() appends.  Here every
//     edge is a member of two doubly-linked lists (its source's out list, its target's in list): unlinking keeps the order of
//     the others, appending goes to the tail — the iteration orders are the vectors'.
//   * `_bbMap` (std::map read with operator[]): an array preset to 0 — a vertex that was never entered maps to vertex 0.
//   * mergeInNodes groups the candidates in a std::map<char, vector>: ascending base, insertion order inside a group, the groups
//     captured BEFORE anything is merged; its recursion is an explicit stack of such captures.
//   * bestPath: float scores, `>` keeps the first best out-edge.
// Memory: node / edge slots of a part come out of fixed regions sized by the caller (cns_caps); edge slots of cleared vertices
// are reused (nothing depends on an edge's number).  Running out of a region, a queue or the stack sets an error code — never a
// silent difference.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CNS_HD __host__ __device__ inline
#else
#define CNS_HD inline
#endif

namespace pagcns {

constexpr uint32_t NONE = 0xFFFFFFFFu;
enum { CNS_OK = 0, CNS_E_NODES = 1, CNS_E_EDGES = 2, CNS_E_QUEUE = 3, CNS_E_STACK = 4, CNS_E_OVERRUN = 5, CNS_E_OUT = 6, CNS_E_EMPTY_LIST = 7 };

struct Aln {  // one gap-normalised alignment of a part (dagcon::Alignment after normalizeGaps, cns/Alignment.cpp:134-215)
    uint64_t str_off;  // its two rows in the q / t string pools
    uint32_t len;      // columns
    uint32_t start;    // 1-based backbone position of its first column inside the part
    int32_t weight;    // AlignData::weightAln
    uint32_t pad;
};
struct Part {
    uint64_t bb_off;     // the part's backbone slice
    uint32_t bb_len;
    uint32_t n_aln;
    uint64_t aln_first;  // its alignments, in the order they are threaded through the graph
    uint64_t node_base, edge_base, aux_base, out_off;  // its regions in the node / edge / scratch arrays and in the output
    uint32_t node_cap, edge_cap, aux_cap, out_cap;
};

// the arrays of ALL parts (a part works inside its regions)
struct Arrays {
    uint8_t *n_base, *n_flags;                                 // flags: 1 backbone, 2 deleted
    int32_t *n_cov, *n_weight;
    uint32_t *n_bb, *n_oh, *n_ot, *n_ih, *n_it, *n_oc, *n_ic;  // _bbMap; heads / tails / sizes of the out and in lists
    float *n_score;
    int32_t *n_best;
    uint32_t *e_src, *e_dst, *e_on, *e_op, *e_in, *e_ip;       // links: next / previous in the source's out list, the target's in list
    int32_t *e_count;
    uint8_t *e_vis;
    uint32_t *aux;                                             // queue | stack
};

struct Graph {
    uint8_t *nb, *nf;
    int32_t *ncov, *nw;
    uint32_t *nbb, *noh, *not_, *nih, *nit, *noc, *nic;
    float *nscore;
    int32_t *nbest;
    uint32_t *es, *ed, *eon, *eop, *ein, *eip;
    int32_t *ec;
    uint8_t *ev;
    uint32_t *queue, *stack;
    uint32_t n_nodes, node_cap, n_edges_hi, edge_cap, free_head, q_cap, st_cap;
    uint32_t enter, exit_;
    int err;
};

CNS_HD void bind(Graph &g, const Arrays &A, const Part &P) {
    g.nb = A.n_base + P.node_base;
    g.nf = A.n_flags + P.node_base;
    g.ncov = A.n_cov + P.node_base;
    g.nw = A.n_weight + P.node_base;
    g.nbb = A.n_bb + P.node_base;
    g.noh = A.n_oh + P.node_base;
    g.not_ = A.n_ot + P.node_base;
    g.nih = A.n_ih + P.node_base;
    g.nit = A.n_it + P.node_base;
    g.noc = A.n_oc + P.node_base;
    g.nic = A.n_ic + P.node_base;
    g.nscore = A.n_score + P.node_base;
    g.nbest = A.n_best + P.node_base;
    g.es = A.e_src + P.edge_base;
    g.ed = A.e_dst + P.edge_base;
    g.eon = A.e_on + P.edge_base;
    g.eop = A.e_op + P.edge_base;
    g.ein = A.e_in + P.edge_base;
    g.eip = A.e_ip + P.edge_base;
    g.ec = A.e_count + P.edge_base;
    g.ev = A.e_vis + P.edge_base;
    g.q_cap = P.aux_cap / 2;
    g.st_cap = P.aux_cap - g.q_cap;
    g.queue = A.aux + P.aux_base;
    g.stack = g.queue + g.q_cap;
    g.n_nodes = 0;
    g.node_cap = P.node_cap;
    g.n_edges_hi = 0;
    g.edge_cap = P.edge_cap;
    g.free_head = NONE;
    g.err = CNS_OK;
}

CNS_HD uint32_t new_node(Graph &g) {
    if (g.n_nodes >= g.node_cap) {
        g.err = g.err ? g.err : CNS_E_NODES;
        return 0;
    }
    const uint32_t v = g.n_nodes++;
    g.nb[v] = 'N';
    g.nf[v] = 0;
    g.ncov[v] = 0;
    g.nw[v] = 0;
    g.nbb[v] = 0;
    g.noh[v] = g.not_[v] = g.nih[v] = g.nit[v] = NONE;
    g.noc[v] = g.nic[v] = 0;
    return v;
}

// boost::add_edge: appended to both lists
CNS_HD uint32_t add_edge_raw(Graph &g, uint32_t u, uint32_t v) {
    uint32_t e;
    if (g.free_head != NONE) {
        e = g.free_head;
        g.free_head = g.eon[e];
    } else {
        if (g.n_edges_hi >= g.edge_cap) {
            g.err = g.err ? g.err : CNS_E_EDGES;
            return 0;
        }
        e = g.n_edges_hi++;
    }
    g.es[e] = u;
    g.ed[e] = v;
    g.ec[e] = 0;
    g.ev[e] = 0;
    g.eon[e] = NONE;
    g.eop[e] = g.not_[u];
    if (g.not_[u] != NONE) g.eon[g.not_[u]] = e;
    else g.noh[u] = e;
    g.not_[u] = e;
    g.noc[u] += 1;
    g.ein[e] = NONE;
    g.eip[e] = g.nit[v];
    if (g.nit[v] != NONE) g.ein[g.nit[v]] = e;
    else g.nih[v] = e;
    g.nit[v] = e;
    g.nic[v] += 1;
    return e;
}
CNS_HD void unlink_out(Graph &g, uint32_t e) {  // out of its source's out list
    const uint32_t u = g.es[e], p = g.eop[e], n = g.eon[e];
    if (p != NONE) g.eon[p] = n;
    else g.noh[u] = n;
    if (n != NONE) g.eop[n] = p;
    else g.not_[u] = p;
    g.noc[u] -= 1;
}
CNS_HD void unlink_in(Graph &g, uint32_t e) {  // out of its target's in list
    const uint32_t v = g.ed[e], p = g.eip[e], n = g.ein[e];
    if (p != NONE) g.ein[p] = n;
    else g.nih[v] = n;
    if (n != NONE) g.eip[n] = p;
    else g.nit[v] = p;
    g.nic[v] -= 1;
}
CNS_HD bool check_vertex(Graph &g, uint32_t v) {
    // (the reference indexes its vertex vector without a check; an alignment that runs past its part is undefined behaviour
    // there — here, as in the host restatement, an error)
    if (v >= g.n_nodes) {
        g.err = g.err ? g.err : CNS_E_OVERRUN;
        return false;
    }
    return true;
}
// AlnGraphBoost::addEdge (:115-135): every in-edge of v that comes from u gains the weight; none: a new edge
CNS_HD void add_edge(Graph &g, uint32_t u, uint32_t v, int weight) {
    if (!check_vertex(g, v)) return;
    bool exists = false;
    for (uint32_t e = g.nih[v]; e != NONE; e = g.ein[e])
        if (g.es[e] == u) {
            g.ec[e] += weight;
            exists = true;
        }
    if (!exists) {
        const uint32_t e = add_edge_raw(g, u, v);
        if (!g.err) g.ec[e] += weight;
    }
}
CNS_HD int find_edge(const Graph &g, uint32_t u, uint32_t v) {  // boost::edge(u, v, g): first match in u's out list
    for (uint32_t e = g.noh[u]; e != NONE; e = g.eon[e])
        if (g.ed[e] == v) return (int)e;
    return -1;
}

// AlnGraphBoost(backbone) (:16-39)
CNS_HD void init_backbone(Graph &g, const char *bb, uint32_t blen) {
    for (uint32_t i = 0; i < blen + 2 && !g.err; ++i) new_node(g);
    if (g.err) return;
    for (uint32_t i = 0; i < blen + 1 && !g.err; ++i) add_edge_raw(g, i, i + 1);
    g.enter = 0;
    g.nb[0] = '^';
    g.nf[0] = 1;
    for (uint32_t i = 0; i < blen; ++i) {
        g.nf[i + 1] = 1;
        g.nw[i + 1] = 1;
        g.nb[i + 1] = (uint8_t)bb[i];
        g.nbb[i + 1] = i + 1;
    }
    g.exit_ = blen + 1;
    g.nb[g.exit_] = '$';
    g.nf[g.exit_] = 1;
}

// addAln (:64-113)
CNS_HD void add_aln(Graph &g, const char *q, const char *t, uint32_t len, uint32_t start, int weight) {
    if (weight <= 0) return;
    uint32_t bb_pos = start, prev = g.enter;
    for (uint32_t i = 0; i < len && !g.err; ++i) {
        const char qb = q[i], tb = t[i];
        const uint32_t cur = bb_pos;
        if (qb == tb) {  // match
            if (!check_vertex(g, cur)) return;
            const uint32_t bbv = g.nbb[cur];
            g.ncov[bbv] += weight;
            g.nb[bbv] = (uint8_t)tb;
            g.nw[cur] += weight;
            add_edge(g, prev, cur, weight);
            bb_pos++;
            prev = cur;
        } else if (qb == '-' && tb != '-') {  // query deletion
            if (!check_vertex(g, cur)) return;
            const uint32_t bbv = g.nbb[cur];
            g.ncov[bbv] += weight;
            g.nb[bbv] = (uint8_t)tb;
            bb_pos++;
        } else if (qb != '-' && tb == '-') {  // query insertion
            const uint32_t nv = new_node(g);
            if (g.err) return;
            g.nb[nv] = (uint8_t)qb;
            g.nw[nv] += weight;
            g.nbb[nv] = bb_pos;
            add_edge(g, prev, nv, weight);
            prev = nv;
        }
    }
    if (!g.err) add_edge(g, prev, g.exit_, weight);
}

// boost::clear_vertex for a bidirectional vecS graph + the slots of its edges handed back
CNS_HD void clear_vertex(Graph &g, uint32_t n) {
    for (uint32_t oe = g.noh[n]; oe != NONE; oe = g.eon[oe]) {  // the in lists of its targets lose every edge that comes from n
        const uint32_t d = g.ed[oe];
        for (uint32_t x = g.nih[d]; x != NONE;) {
            const uint32_t nx = g.ein[x];
            if (g.es[x] == n) unlink_in(g, x);
            x = nx;
        }
    }
    for (uint32_t ie = g.nih[n]; ie != NONE; ie = g.ein[ie]) {  // the out lists of its sources lose every edge that goes to n
        const uint32_t s = g.es[ie];
        for (uint32_t x = g.noh[s]; x != NONE;) {
            const uint32_t nx = g.eon[x];
            if (g.ed[x] == n) unlink_out(g, x);
            x = nx;
        }
    }
    // (n's own lists are intact as chains: their members go to the free list)
    for (uint32_t oe = g.noh[n]; oe != NONE;) {
        const uint32_t nx = g.eon[oe];
        g.eon[oe] = g.free_head;
        g.free_head = oe;
        oe = nx;
    }
    for (uint32_t ie = g.nih[n]; ie != NONE;) {
        const uint32_t nx = g.ein[ie];
        g.eon[ie] = g.free_head;
        g.free_head = ie;
        ie = nx;
    }
    g.noh[n] = g.not_[n] = g.nih[n] = g.nit[n] = NONE;
    g.noc[n] = g.nic[n] = 0;
}
CNS_HD void mark_for_reaper(Graph &g, uint32_t n) {
    g.nf[n] |= 2u;
    clear_vertex(g, n);
}

// the next group of a capture (std::map<char, vector>: ascending char — signed on the reference's platform): the smallest base
// above `last` among cap[0 .. n); NO_BASE when none is left
constexpr int NO_BASE = 1000, BELOW_ALL_BASES = -1000;
CNS_HD int next_base(const Graph &g, const uint32_t *cap, uint32_t n, int last) {
    int best = NO_BASE;
    for (uint32_t i = 0; i < n; ++i) {
        const int b = (int)(int8_t)g.nb[cap[i]];
        if (b > last && b < best) best = b;
    }
    return best;
}
// the x-th member (x >= 0) of the group with base b, NONE when there are no more
CNS_HD uint32_t group_member(const Graph &g, const uint32_t *cap, uint32_t n, int b, uint32_t x) {
    for (uint32_t i = 0; i < n; ++i)
        if ((int)(int8_t)g.nb[cap[i]] == b) {
            if (x == 0) return cap[i];
            --x;
        }
    return NONE;
}

// mergeInNodes (:170-223).  The groups of a call are captured before anything is merged, and the call recurses into the
// surviving node of every group it merges: an explicit stack of captures, a frame = [members ..., n, count, last base] with
// its header on top.
CNS_HD void merge_in_nodes(Graph &g, uint32_t n0) {
    uint32_t sp = 0;  // words in use
    auto push = [&](uint32_t n) {
        uint32_t cnt = 0;
        for (uint32_t ie = g.nih[n]; ie != NONE; ie = g.ein[ie]) {
            const uint32_t s = g.es[ie];
            if (g.noc[s] == 1) {
                if (sp + cnt + 4 > g.st_cap) {
                    g.err = g.err ? g.err : CNS_E_STACK;
                    return;
                }
                g.stack[sp + cnt++] = s;
            }
        }
        if (sp + cnt + 3 > g.st_cap) {
            g.err = g.err ? g.err : CNS_E_STACK;
            return;
        }
        g.stack[sp + cnt] = n;
        g.stack[sp + cnt + 1] = cnt;
        g.stack[sp + cnt + 2] = (uint32_t)BELOW_ALL_BASES;
        sp += cnt + 3;
    };
    push(n0);
    while (sp != 0 && !g.err) {
        const uint32_t cnt = g.stack[sp - 2];
        const uint32_t *cap = g.stack + (sp - 3 - cnt);
        const int b = next_base(g, cap, cnt, (int)g.stack[sp - 1]);
        if (b == NO_BASE) {  // the call returns
            sp -= cnt + 3;
            continue;
        }
        g.stack[sp - 1] = (uint32_t)b;
        if (group_member(g, cap, cnt, b, 1) == NONE) continue;  // a group of one
        const uint32_t an = group_member(g, cap, cnt, b, 0);
        for (uint32_t x = 1;; ++x) {  // accumulate out edge information
            const uint32_t m = group_member(g, cap, cnt, b, x);
            if (m == NONE) break;
            if (g.noh[an] == NONE || g.noh[m] == NONE) {  // (.front() of an empty vector in the reference: undefined there)
                g.err = g.err ? g.err : CNS_E_EMPTY_LIST;
                return;
            }
            g.ec[g.noh[an]] += g.ec[g.noh[m]];
            g.nw[an] += g.nw[m];
        }
        for (uint32_t x = 1; !g.err; ++x) {  // accumulate in edge information, merge nodes
            const uint32_t m = group_member(g, cap, cnt, b, x);
            if (m == NONE) break;
            for (uint32_t ie = g.nih[m]; ie != NONE && !g.err; ie = g.ein[ie]) {
                const uint32_t n1 = g.es[ie];
                const int e = find_edge(g, n1, an);
                if (e >= 0) {
                    g.ec[e] += g.ec[ie];
                } else {
                    const int c = g.ec[ie];
                    const uint8_t vis = g.ev[ie];
                    const uint32_t ne = add_edge_raw(g, n1, an);
                    if (g.err) return;
                    g.ec[ne] = c;
                    g.ev[ne] = vis;
                }
            }
            mark_for_reaper(g, m);
        }
        push(an);  // mergeInNodes(an)
    }
}

// mergeOutNodes (:225-275); the capture sits on the (otherwise empty) stack
CNS_HD void merge_out_nodes(Graph &g, uint32_t n) {
    uint32_t cnt = 0;
    for (uint32_t oe = g.noh[n]; oe != NONE; oe = g.eon[oe]) {
        const uint32_t d = g.ed[oe];
        if (g.nic[d] == 1) {
            if (cnt + 1 > g.st_cap) {
                g.err = g.err ? g.err : CNS_E_STACK;
                return;
            }
            g.stack[cnt++] = d;
        }
    }
    const uint32_t *cap = g.stack;
    for (int last = BELOW_ALL_BASES; !g.err;) {
        const int b = next_base(g, cap, cnt, last);
        if (b == NO_BASE) break;
        last = b;
        if (group_member(g, cap, cnt, b, 1) == NONE) continue;
        const uint32_t an = group_member(g, cap, cnt, b, 0);
        for (uint32_t x = 1;; ++x) {  // accumulate inner edge information
            const uint32_t m = group_member(g, cap, cnt, b, x);
            if (m == NONE) break;
            if (g.nih[an] == NONE || g.nih[m] == NONE) {
                g.err = g.err ? g.err : CNS_E_EMPTY_LIST;
                return;
            }
            g.ec[g.nih[an]] += g.ec[g.nih[m]];
            g.nw[an] += g.nw[m];
        }
        for (uint32_t x = 1; !g.err; ++x) {  // accumulate and merge outer edge information
            const uint32_t m = group_member(g, cap, cnt, b, x);
            if (m == NONE) break;
            for (uint32_t oe = g.noh[m]; oe != NONE && !g.err; oe = g.eon[oe]) {
                const uint32_t n2 = g.ed[oe];
                const int e = find_edge(g, an, n2);
                if (e >= 0) {
                    g.ec[e] += g.ec[oe];
                } else {
                    const int c = g.ec[oe];
                    const uint8_t vis = g.ev[oe];
                    const uint32_t ne = add_edge_raw(g, an, n2);
                    if (g.err) return;
                    g.ec[ne] = c;
                    g.ev[ne] = vis;
                }
            }
            mark_for_reaper(g, m);
        }
    }
}

struct Queue {  // std::queue on a ring
    uint32_t head = 0, count = 0;
};
CNS_HD void q_push(Graph &g, Queue &q, uint32_t v) {
    if (q.count >= g.q_cap) {
        g.err = g.err ? g.err : CNS_E_QUEUE;
        return;
    }
    g.queue[(q.head + q.count) % g.q_cap] = v;
    q.count += 1;
}
CNS_HD uint32_t q_pop(Graph &g, Queue &q) {
    const uint32_t v = g.queue[q.head];
    q.head = (q.head + 1) % g.q_cap;
    q.count -= 1;
    return v;
}

// mergeNodes (:137-168)
CNS_HD void merge_nodes(Graph &g) {
    Queue q;
    q_push(g, q, g.enter);
    while (q.count && !g.err) {
        const uint32_t u = q_pop(g, q);
        merge_in_nodes(g, u);
        if (g.err) return;
        merge_out_nodes(g, u);
        for (uint32_t oe = g.noh[u]; oe != NONE && !g.err; oe = g.eon[oe]) {
            g.ev[oe] = 1;
            const uint32_t v = g.ed[oe];
            int not_visited = 0;
            for (uint32_t ie = g.nih[v]; ie != NONE; ie = g.ein[ie])
                if (!g.ev[ie]) not_visited++;
            if (not_visited == 0) q_push(g, q, v);
        }
    }
}

// bestPath (:383-467) + consensus (:293-333): the consensus string goes to out[0 .. *out_len)
CNS_HD void consensus(Graph &g, int min_weight, char *out, uint32_t out_cap, uint32_t *out_len) {
    // (edges(_g) only holds the edges that still exist; the flags of erased ones do not matter)
    for (uint32_t e = 0; e < g.n_edges_hi; ++e) g.ev[e] = 0;
    for (uint32_t v = 0; v < g.n_nodes; ++v) {
        g.nbest[v] = -1;
        g.nscore[v] = 0.0f;
    }
    Queue q;
    q_push(g, q, g.exit_);
    while (q.count && !g.err) {
        const uint32_t n = q_pop(g, q);
        bool found = false;
        float best_score = -3.402823466e+38f;  // -FLT_MAX
        int best_edge = -1;
        for (uint32_t oe = g.noh[n]; oe != NONE; oe = g.eon[oe]) {
            const uint32_t od = g.ed[oe];
            float new_score;
            const float score = g.nscore[od];
            if ((g.nf[od] & 1u) && g.nw[od] == 1) {
                new_score = score - 10.0f;
            } else {
                const uint32_t bbv = g.nbb[od];
                if (bbv >= g.n_nodes) {
                    g.err = g.err ? g.err : CNS_E_OVERRUN;
                    return;
                }
                new_score = (float)g.ec[oe] - (float)g.ncov[bbv] * 0.5f + score;
            }
            if (new_score > best_score) {
                best_score = new_score;
                best_edge = (int)oe;
                found = true;
            }
        }
        if (found) {
            g.nscore[n] = best_score;
            g.nbest[n] = best_edge;
        }
        for (uint32_t ie = g.nih[n]; ie != NONE && !g.err; ie = g.ein[ie]) {
            g.ev[ie] = 1;
            const uint32_t in_node = g.es[ie];
            int not_visited = 0;
            for (uint32_t oe = g.noh[in_node]; oe != NONE; oe = g.eon[oe])
                if (!g.ev[oe]) not_visited++;
            if (not_visited == 0) q_push(g, q, in_node);
        }
    }
    if (g.err) return;
    // the path from the enter vertex along the best edges, its bases except those that look like the enter / exit vertex's;
    // the longest stretch whose nodes all weigh at least min_weight (:293-333)
    const uint8_t enter_base = g.nb[g.enter], exit_base = g.nb[g.exit_];
    uint32_t n_out = 0;
    int offs = 0, best_offs = 0, length = 0, idx = 0;
    bool met = false;
    uint32_t prev = g.enter;
    for (uint32_t guard = 0; guard <= g.n_nodes; ++guard) {
        const uint8_t base = g.nb[prev];
        if (!(base == enter_base || base == exit_base)) {
            if (n_out >= out_cap) {
                g.err = CNS_E_OUT;
                return;
            }
            out[n_out++] = (char)base;
            const int wgt = g.nw[prev];
            if (!met && wgt >= min_weight) {
                offs = idx;
                met = true;
            } else if (met && wgt < min_weight) {
                if ((idx - offs) > length) {
                    best_offs = offs;
                    length = idx - offs;
                }
                met = false;
            }
            idx++;
        }
        if (g.nbest[prev] < 0) break;
        prev = g.ed[g.nbest[prev]];
    }
    if (met && (idx - offs) > length) {
        best_offs = offs;
        length = idx - offs;
    }
    // cns.substr(bestOffs, length), in place
    for (int i = 0; i < length; ++i) out[i] = out[best_offs + i];
    *out_len = (uint32_t)length;
}

// one part, start to finish (pa_cns.cpp:98-124): backbone, its alignments in order, merge, consensus
CNS_HD int run_part(const Arrays &A, const Part &P, const char *backbone, const Aln *alns, const char *qpool, const char *tpool, int min_weight, char *out,
                    uint32_t *out_len) {
    Graph g;
    bind(g, A, P);
    *out_len = 0;
    init_backbone(g, backbone + P.bb_off, P.bb_len);
    for (uint32_t a = 0; a < P.n_aln && !g.err; ++a) {
        const Aln &al = alns[P.aln_first + a];
        add_aln(g, qpool + al.str_off, tpool + al.str_off, al.len, al.start, al.weight);
    }
    if (!g.err) merge_nodes(g);
    if (!g.err) consensus(g, min_weight, out + P.out_off, P.out_cap, out_len);
    return g.err;
}

}  // namespace pagcns
