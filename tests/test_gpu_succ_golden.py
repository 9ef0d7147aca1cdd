"""GPU: EVERY successor record of the traversal graph against the reference's own epsilon-join.

tests/golden/<case>/succ.txt.gz holds what the compiled reference's PABruijnGraph::successors (PABruijnGraph.cpp:167-197, with the
traversal's arguments: deviation 2 * epsilon, error rate 0.15) returns for every vertex of the finished graph — target, step,
checkPosition grade (:143-165), isEdgeSimilar().first (:385-400), in the reference's order (oracle/ref_harness/graph_dump.cpp
--succ 1, written by tests/golden/make_golden.py).  The device's records (pag_debug_succ) are mapped back to (k-mer code,
position) through pag_debug_trav_vertices and compared record for record:

  * the whole graph's view: every vertex present, every list equal, through every way the records are built (staged by the
    candidate bound / count + fill / fused: one evaluation with dense staging; vertices with many candidate pairs by a whole wave
    with limits 64, 4, never);
  * the view cut to what the block's traversals can examine (pag_travel_prepare_for), with the default margins and with tight
    ones: a kept vertex's list is the reference's list restricted to the kept targets, marker records (GRADE_POISON_IF_LEAP
    behind the list, GRADE_POISON in place of it) aside.

The case succ_corners_t8 carries the corners: k-mers with > 255 clustered positions (edge_target's clamped count), vertices
with > 64 candidate pairs (k_succ_heavy), steps >= 1024 (the f64 ratio tests instead of the integer-interval table).
What every record says about its target (contig coordinate, first record, record count clamped to 15) is checked as well."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

import goldens
import pagctl
from aligngraph2_amd.workload import PagRawInput, PagSeqs


class TravelParams(C.Structure):
    _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                ("start_split", C.c_double), ("min_len", C.c_uint64)]


def reference_blocks(name):
    """per config block: keys (code, ctg, ref) of the vertices in the reference's order, record offsets, records as
    (target vertex index, step, grade, ctg-similar)"""
    graph = gzip.open(os.path.join(goldens.GOLDEN, name, "graph.txt.gz"), "rt").read().split("\n")
    pos_of = []  # per block: {code: [(ctg, ref), ...]}
    for line in graph:
        if line.startswith("S"):
            pos_of.append({})
        elif line.startswith("K"):
            cur = pos_of[-1].setdefault(int(line.split()[1]), [])
        elif line.startswith("P"):
            p = line.split()
            cur.append((int(p[1]), int(p[2])))
    blocks = []
    for line in gzip.open(os.path.join(goldens.GOLDEN, name, "succ.txt.gz"), "rt"):
        p = line.split()
        if p[0] == "B":
            blocks.append({"keys": [], "off": [0], "rec": [], "pos": pos_of[int(p[1])]})
            b = blocks[-1]
        elif p[0] == "V":
            code, j = int(p[1]), int(p[2])
            b["keys"].append((code,) + b["pos"][code][j])
            b["off"].append(b["off"][-1] + int(p[3]))
        else:
            code, j = int(p[1]), int(p[2])
            b["rec"].append(((code,) + b["pos"][code][j], int(p[3]), int(p[4]), int(p[5])))
    for b in blocks:
        index = {key: i for i, key in enumerate(b["keys"])}
        assert len(index) == len(b["keys"]), "(code, position) does not name a vertex uniquely"
        assert b["off"][-1] == len(b["rec"])
        b["index"] = index
        b["off"] = np.array(b["off"], dtype=np.int64)
        b["rec"] = np.array([(index[t], s, g, e) for t, s, g, e in b["rec"]], dtype=np.int64).reshape(-1, 4)
    return blocks


def block_orient(ind, block, ctg_names):
    """PAG_ORIENT_* per contig of the -c file for config block `block` (hip_backend.cpp orientations())"""
    lines = open(os.path.join(ind, "config.txt")).read().split("\n")
    blocks, i = [], 0
    while i < len(lines) and lines[i]:
        i += 4
        ctgs = []
        while i < len(lines) and lines[i]:
            ctgs.append((lines[i], lines[i + 1].strip() == "1"))
            i += 2
        blocks.append(ctgs)
        i += 1
    orient = np.full(len(ctg_names), -1, dtype=np.int32)
    for nm, fwd in blocks[block]:
        c = ctg_names.index(nm)
        mine = 1 if fwd else 0
        orient[c] = mine if orient[c] in (-1, mine) else 2
    return orient


def device_records(hip, g):
    n_pos, n_succ = C.c_uint64(), C.c_uint64()
    assert hip.pag_debug_succ_sizes(g, C.byref(n_pos), C.byref(n_succ)) == 0
    off = np.zeros(n_pos.value + 1, dtype=np.uint32)
    recs = np.zeros((n_succ.value, 4), dtype=np.uint32)
    assert hip.pag_debug_succ(g, off.ctypes.data, recs.ctypes.data) == 0, hip.pag_last_error()
    code = np.zeros(n_pos.value, dtype=np.uint32)
    pos = np.zeros(n_pos.value, dtype=np.uint64)
    assert hip.pag_debug_trav_vertices(g, code.ctypes.data, pos.ctypes.data) == 0, hip.pag_last_error()
    return off.astype(np.int64), recs, code, pos


def check_against_reference(ref, off, recs, code, pos, whole, label):
    """the device's records (new ids) against the reference's (its vertex order); returns counts for the test's own asserts"""
    n = len(code)
    r_of_u = np.array([ref["index"][(int(c), int(p >> np.uint64(32)), int(p & np.uint64(0xFFFFFFFF)))] for c, p in zip(code, pos)], dtype=np.int64)
    assert len(np.unique(r_of_u)) == n, f"{label}: two ids of the view are the same vertex"
    if whole:
        assert n == len(ref["keys"]), f"{label}: the whole graph's view holds {n} of {len(ref['keys'])} vertices"
    u_of_r = np.full(len(ref["keys"]), -1, dtype=np.int64)
    u_of_r[r_of_u] = np.arange(n)
    assert int(off[-1]) == len(recs)
    tgt, pc, meta, toff = (recs[:, i].astype(np.int64) for i in range(4))
    step, grade, esim, tcnt = meta & 0xFFFFFF, (meta >> 24) & 7, (meta >> 27) & 1, meta >> 28
    cnt = off[1:] - off[:-1]
    owner = np.repeat(np.arange(n), cnt)
    marker, poison = grade == 6, grade == 7
    real = ~(marker | poison)
    # what a record says about its target
    assert np.array_equal(pc[real], (pos[tgt[real]] >> np.uint64(32)).astype(np.int64)), f"{label}: contig coordinate of a target"
    assert np.array_equal(toff[real], off[tgt[real]]), f"{label}: first record of a target"
    assert np.array_equal(tcnt[real], np.minimum(cnt[tgt[real]], 15)), f"{label}: record count of a target"
    if whole:
        assert not marker.any() and not poison.any(), f"{label}: marker records in the whole graph's view"
    # markers: a poison record stands alone, a conditional one is the last of its vertex
    assert np.all(cnt[owner[poison]] == 1)
    assert np.all(np.flatnonzero(marker) + 1 == off[owner[marker] + 1])
    poisoned = np.zeros(n, dtype=bool)
    poisoned[owner[poison]] = True
    # the reference's lists of the kept, unpoisoned vertices, restricted to kept targets, in the device's vertex order
    r_cnt = ref["off"][1:] - ref["off"][:-1]
    keep_u = np.flatnonzero(~poisoned)
    r_sel = r_of_u[keep_u]
    starts = ref["off"][r_sel]
    lens = r_cnt[r_sel]
    idx = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens) + np.arange(int(lens.sum()))
    rr = ref["rec"][idx]
    r_owner = np.repeat(keep_u, lens)
    t_u = u_of_r[rr[:, 0]]
    kept = t_u >= 0
    if whole:
        assert kept.all()
    want = np.stack([r_owner[kept], t_u[kept], rr[kept, 1], rr[kept, 2], rr[kept, 3]], axis=1)
    got = np.stack([owner[real], tgt[real], step[real], grade[real], esim[real]], axis=1)
    if want.shape != got.shape or not np.array_equal(want, got):
        m = min(len(want), len(got))
        d = np.flatnonzero((want[:m] != got[:m]).any(axis=1))
        at = int(d[0]) if len(d) else m
        raise AssertionError(f"{label}: {len(got)} records, the reference has {len(want)}; first difference at record {at}: "
                             f"(vertex, target, step, grade, ctg-similar) device {got[at:at + 3].tolist()} reference {want[at:at + 3].tolist()}")
    return {"vertices": n, "records": int(real.sum()), "poisoned": int(poisoned.sum()), "markers": int(marker.sum()),
            "big_steps": int((step[real] >= 1024).sum()), "dropped_targets": int((~kept).sum())}


MODES = [  # (label, view, environment): the one way the records are built (k_succ_emit -> sort by source -> k_succ_finish), with
          # its thread-per-vertex / wave-per-vertex split at the default limit, at 4 candidate pairs (most vertices by a wave) and never
    ("whole", "whole", {}),
    ("whole heavy=4", "whole", {"PAG_SUCC_HEAVY": "4"}),
    ("whole heavy=0", "whole", {"PAG_SUCC_HEAVY": "0"}),
    ("whole by PAG_TRAVEL_VIEW", "for", {"PAG_TRAVEL_VIEW": "whole"}),
    ("cut default", "for", {}),
    ("cut tight", "for", {"PAG_VIEW_HALO": "300", "PAG_VIEW_MARGIN": "50"}),
    ("cut tight heavy=4", "for", {"PAG_SUCC_HEAVY": "4", "PAG_VIEW_HALO": "300", "PAG_VIEW_MARGIN": "50"}),
    ("cut tight heavy=0", "for", {"PAG_SUCC_HEAVY": "0", "PAG_VIEW_HALO": "300", "PAG_VIEW_MARGIN": "50"}),
]
SWITCHES = ("PAG_SUCC_HEAVY", "PAG_TRAVEL_VIEW", "PAG_VIEW_HALO", "PAG_VIEW_MARGIN")


@pytest.mark.gpu
@pytest.mark.parametrize("name", goldens.case_names())
def test_every_successor_record_equals_the_reference(name, workdir, monkeypatch):
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    ref_blocks = reference_blocks(name)
    hip = pagctl.hip_lib()
    hip.pag_travel_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_travel_prepare_for.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_debug_succ_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.pag_debug_succ.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.pag_debug_trav_vertices.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    ctg_names = [l[1:].split()[0] for l in open(os.path.join(ind, "ctg.fasta")) if l.startswith(">")]
    totals = {"big_steps": 0, "cut": 0, "markers": 0}
    for block, ref in enumerate(ref_blocks):
        inp = pagctl.LoadedInput(ind, threads=spec["threads"], eps=spec["epsilon"], cov=spec["cov"], block=block)
        err = C.c_int()
        g = C.c_void_p(hip.pag_create(inp.kmer_words, inp.n_kmer_words, inp.k, 0, C.byref(err)))
        assert g, hip.pag_last_error()
        try:
            raw = PagRawInput.from_address(inp.raw_view)
            ctg_len = np.ctypeslib.as_array(C.cast(raw.ctg_len, C.POINTER(C.c_uint32)), shape=(raw.n_ctgs,)).copy()
            ref_len = np.ctypeslib.as_array(C.cast(raw.ref_len, C.POINTER(C.c_uint32)), shape=(raw.n_refs,)).copy()
            assert len(ctg_names) == raw.n_ctgs
            ctgs = PagSeqs(raw.n_ctgs, None, ctg_len.ctypes.data, None, 0)  # (the view is laid out from the lengths alone)
            orient = block_orient(ind, block, ctg_names)
            prm = TravelParams(spec["threads"], 0, 2 * spec["epsilon"], 0.15, 0.90, 50)
            prepared = pagctl._prepared_view(hip, g, inp)
            for label, view, env in MODES:
                for s in SWITCHES:
                    monkeypatch.delenv(s, raising=False)
                for k_, v in env.items():
                    monkeypatch.setenv(k_, v)
                st = pagctl.BuildStats()
                assert hip.pag_process(g, C.byref(prepared), C.byref(st)) == 0, hip.pag_last_error()
                if view == "whole":
                    rc = hip.pag_travel_prepare(g, C.byref(ctgs), ref_len.ctypes.data, len(ref_len), C.byref(prm), None)
                else:
                    rc = hip.pag_travel_prepare_for(g, C.byref(ctgs), orient.ctypes.data, ref_len.ctypes.data, len(ref_len), C.byref(prm), None)
                assert rc == 0, hip.pag_last_error()
                off, recs, code, pos = device_records(hip, g)
                whole = view == "whole" or env.get("PAG_TRAVEL_VIEW") == "whole"
                got = check_against_reference(ref, off, recs, code, pos, whole, f"{name} block {block} [{label}]")
                if whole:
                    assert got["records"] == len(ref["rec"])
                    totals["big_steps"] = max(totals["big_steps"], got["big_steps"])
                else:
                    totals["cut"] = max(totals["cut"], len(ref["keys"]) - got["vertices"])
                    totals["markers"] = max(totals["markers"], got["markers"] + got["poisoned"])
        finally:
            hip.pag_destroy(g)
            inp.close()
    if name == "succ_corners_t8":
        assert totals["big_steps"] > 0, "no record with a step of 1024 or more: the f64 ratio tests were not exercised"
        npos = [len(v) for v in ref_blocks[0]["pos"].values()]
        assert max(npos) > 255, "no k-mer with more than 255 positions: edge_target's clamp was not exercised"
    if name in ("succ_corners_t8", "two_blocks_both_orient_t16"):
        assert totals["cut"] > 0, "the tight view kept every vertex: the cut was not exercised"


class PagSucc(C.Structure):
    _fields_ = [("code", C.c_uint32), ("step", C.c_uint32), ("pos", C.c_uint64), ("grade", C.c_uint32), ("ctg_similar", C.c_uint32)]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["join_fwd_t1", "succ_corners_t8", "two_blocks_both_orient_t16"])
def test_pag_successors_answers_like_the_reference_s_successors(name, workdir, monkeypatch):
    """the public call (pagraph_hip.h pag_successors = PABruijnGraph::successors, PABruijnGraph.cpp:167-197): vertices named by
    (k-mer, position) — the longest lists, the empty ones, a spread of the rest, a vertex that does not exist, a buffer that is
    too small, a view that was cut"""
    for s in SWITCHES:
        monkeypatch.delenv(s, raising=False)
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in_q"))
    ref_blocks = reference_blocks(name)
    hip = pagctl.hip_lib()
    hip.pag_travel_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_travel_prepare_for.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_successors.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64]
    hip.pag_successors.restype = C.c_int64
    ctg_names = [l[1:].split()[0] for l in open(os.path.join(ind, "ctg.fasta")) if l.startswith(">")]
    for block, ref in enumerate(ref_blocks):
        inp = pagctl.LoadedInput(ind, threads=spec["threads"], eps=spec["epsilon"], cov=spec["cov"], block=block)
        err = C.c_int()
        g = C.c_void_p(hip.pag_create(inp.kmer_words, inp.n_kmer_words, inp.k, 0, C.byref(err)))
        assert g, hip.pag_last_error()
        try:
            raw = PagRawInput.from_address(inp.raw_view)
            ctg_len = np.ctypeslib.as_array(C.cast(raw.ctg_len, C.POINTER(C.c_uint32)), shape=(raw.n_ctgs,)).copy()
            ref_len = np.ctypeslib.as_array(C.cast(raw.ref_len, C.POINTER(C.c_uint32)), shape=(raw.n_refs,)).copy()
            ctgs = PagSeqs(raw.n_ctgs, None, ctg_len.ctypes.data, None, 0)
            prm = TravelParams(spec["threads"], 0, 2 * spec["epsilon"], 0.15, 0.90, 50)
            prepared = pagctl._prepared_view(hip, g, inp)
            st = pagctl.BuildStats()
            assert hip.pag_process(g, C.byref(prepared), C.byref(st)) == 0, hip.pag_last_error()
            assert hip.pag_successors(g, 0, 0, None, 0) == -22  # (nothing prepared yet)
            assert hip.pag_travel_prepare(g, C.byref(ctgs), ref_len.ctypes.data, len(ref_len), C.byref(prm), None) == 0, hip.pag_last_error()
            keys, off, rec = ref["keys"], ref["off"], ref["rec"]
            cnt = off[1:] - off[:-1]
            order = np.argsort(-cnt, kind="stable")
            rng = np.random.default_rng(5)
            pick = list(order[:40]) + list(order[-20:]) + list(rng.choice(len(keys), size=min(400, len(keys)), replace=False))
            buf = (PagSucc * (int(cnt.max()) + 1))()
            for r in pick:
                code, ctg, rf = keys[r]
                n = hip.pag_successors(g, code, (ctg << 32) | rf, buf, len(buf))
                assert n == cnt[r], (name, block, keys[r], n, int(cnt[r]), hip.pag_last_error())
                got = [((b.code, b.pos >> 32, b.pos & 0xFFFFFFFF), b.step, b.grade, b.ctg_similar) for b in buf[:n]]
                want = [(keys[t], s_, g_, e_) for t, s_, g_, e_ in rec[off[r]:off[r + 1]].tolist()]
                assert got == want, (name, block, keys[r])
            # a buffer that is too small: the count is the whole list's, the first `cap` are written
            r = int(order[0])
            code, ctg, rf = keys[r]
            small = (PagSucc * 2)()
            assert hip.pag_successors(g, code, (ctg << 32) | rf, small, 2) == cnt[r]
            assert [(b.code, b.pos) for b in small[:min(2, int(cnt[r]))]] == [(keys[t][0], (keys[t][1] << 32) | keys[t][2]) for t in rec[off[r]:off[r + 1], 0].tolist()[:2]]
            # no such vertex: an existing k-mer at a position it does not have, and a k-mer the graph does not hold
            have = set(keys)
            miss = next((code, ctg, rf + d) for d in range(1, 1000) if (code, ctg, rf + d) not in have)
            assert hip.pag_successors(g, miss[0], (miss[1] << 32) | miss[2], buf, len(buf)) == -22
            codes = {k_[0] for k_ in keys}
            absent = next((c for c in range(4 ** inp.k) if c not in codes), None)
            if absent is not None:
                assert hip.pag_successors(g, absent, (ctg << 32) | rf, buf, len(buf)) == -22
            # a view cut for given traversals answers PAG_ERANGE instead of a restricted list
            orient = block_orient(ind, block, ctg_names)
            assert hip.pag_process(g, C.byref(prepared), C.byref(st)) == 0, hip.pag_last_error()
            assert hip.pag_travel_prepare_for(g, C.byref(ctgs), orient.ctypes.data, ref_len.ctypes.data, len(ref_len), C.byref(prm), None) == 0, hip.pag_last_error()
            assert hip.pag_successors(g, code, (ctg << 32) | rf, buf, len(buf)) == -34
        finally:
            hip.pag_destroy(g)
            inp.close()
