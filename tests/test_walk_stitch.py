"""CPU: the adoption conditions of a walk cut into pieces (aligngraph2_amd/csrc/hip/walk_stitch.hpp: try_merge outside the
leaping zone, try_merge_leap inside it) on constructed job outputs — each condition of the justification in that header is
driven to both sides of its threshold, and the result is compared with an independent restatement of the conditions
written from their description (below), not from the C++."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tests", "harness", "bin", "libpagh_stitch_test.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", ROOT, LIB[len(ROOT) + 1:]], check=True, capture_output=True)
    return C.CDLL(LIB)


def _u32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint32))


def stitch(lib, T, parts, P, *, leap=False, log=None, max_back=0, max_chosen=1, max_probe=0, wd_below=0, wd_forced=0xFFFFFFFF, usable=True,
           k=14, dev=20, split=10 ** 9, has_size=0):
    """T, P: lists of (vertex, step, coordinate); parts: lengths of T's parts; log: per P entry (boundary?, elow, m0)"""
    tv, ts, tpc = (_u32([x[i] for x in T]) for i in range(3))
    pv, ps, ppc = (_u32([x[i] for x in P]) for i in range(3))
    off = np.concatenate([[0], np.cumsum(parts)]).astype(np.uint64)
    assert int(off[-1]) == len(T)
    xl = xh = None
    if log is not None:
        xl = _u32([m0 for _, _, m0 in log])
        xh = _u32([((1 << 31) | (elow & 0x7FFFFFFF)) if b else 0 for b, elow, _ in log])
    out = (C.c_uint64 * 6)()
    tail = np.zeros(len(P) + 1, dtype=np.uint32)
    lib.pagt_stitch.argtypes = [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                                                                     C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                                                                     C.c_void_p, C.c_void_p]
    results = []
    for tables in (0, 1):  # every entry read / through the block tables the pack kernel attaches to a fetched path: the same answers
        lib.pagt_use_tables(tables)
        lib.pagt_stitch(tv.ctypes.data, ts.ctypes.data, tpc.ctypes.data, off.ctypes.data, len(parts), pv.ctypes.data, ps.ctypes.data, ppc.ctypes.data,
                        xl.ctypes.data if xl is not None else None, xh.ctypes.data if xh is not None else None, len(P), max_back, max_chosen, max_probe,
                        wd_below, wd_forced, int(usable), k, dev, split, has_size, int(leap), out, tail.ctypes.data)
        decision, adopted, why = int(out[0]), int(out[1]), C.c_int64(out[2]).value
        results.append((decision, adopted, why, int(out[3]), int(out[4]), list(tail[:adopted]), int(out[5])))
    lib.pagt_use_tables(0)
    assert results[0] == results[1], (results[0][:5], results[1][:5])
    return results[0][:6]


def stitch_kept(lib, T, parts, P, kept, *, leap=False, log=None, max_back=0, max_chosen=1, max_probe=0, wd_below=0, wd_forced=0xFFFFFFFF, k=14, dev=20,
                split=10 ** 9, has_size=0):
    """stitch() for a segment KEPT from an earlier round: kept = (segment's round, running round, g_lo, g_hi, g_free_hi)"""
    tv, ts, tpc = (_u32([x[i] for x in T]) for i in range(3))
    pv, ps, ppc = (_u32([x[i] for x in P]) for i in range(3))
    off = np.concatenate([[0], np.cumsum(parts)]).astype(np.uint64)
    xl = xh = None
    if log is not None:
        xl = _u32([m0 for _, _, m0 in log])
        xh = _u32([((1 << 31) | (elow & 0x7FFFFFFF)) if b else 0 for b, elow, _ in log])
    out = (C.c_uint64 * 6)()
    tail = np.zeros(len(P) + 1, dtype=np.uint32)
    kp = _u32(kept)
    lib.pagt_stitch_kept.argtypes = [C.c_void_p] * 4 + [C.c_uint32] + [C.c_void_p] * 5 + [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32,
                                                                                          C.c_int, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                                                                          C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pagt_use_tables(1)
    lib.pagt_stitch_kept(tv.ctypes.data, ts.ctypes.data, tpc.ctypes.data, off.ctypes.data, len(parts), pv.ctypes.data, ps.ctypes.data, ppc.ctypes.data,
                         xl.ctypes.data if xl is not None else None, xh.ctypes.data if xh is not None else None, len(P), max_back, max_chosen, max_probe,
                         wd_below, wd_forced, 1, k, dev, split, has_size, int(leap), out, tail.ctypes.data, kp.ctypes.data)
    lib.pagt_use_tables(0)
    return int(out[0]), int(out[1]), C.c_int64(out[2]).value


def line(v0, n, c0, step=3):
    """n consecutive path vertices v0, v0 + 1, .. at coordinates c0, c0 + step, .."""
    return [(v0 + i, step, c0 + i * step) for i in range(n)]


# ---------------------------------------------------------------------------------------------------------------------
# outside the leaping zone
# ---------------------------------------------------------------------------------------------------------------------
def expect_merge(T, P, max_back, max_chosen, max_probe, k, dev, split, has_size):
    """the conditions as walk_stitch.hpp states them in words"""
    last = T[-1][0]
    pv = [x[0] for x in P]
    if last not in pv:
        return 0, 0
    be = pv.index(last)
    t = 0  # common vertices going backwards beyond the pair itself (same vertices, same steps from the second one on)
    while t < len(T) - 1 and t < be and T[-2 - t][0] == P[be - t - 1][0] and T[-1 - t][1] == P[be - t][1]:
        t += 1
    if t < 4:
        return 0, 0
    a, b = len(T) - 1 - t, be - t
    dmax = max([x[2] for x in T[:a]] + [x[2] for x in P[:b]] + [0])
    size_T = sum(x[1] for x in T)
    base = has_size + k + size_T + max_probe + 1
    if base >= split:
        return 0, 0
    room = split - base
    q = max(0, be + 1 - max_chosen)
    low = min(x[2] for x in P[q:])
    lastx, acc = be, 0
    for x in range(be + 1, len(P)):
        if acc + P[x][1] < room:
            acc += P[x][1]
            lastx = x
        else:
            break
    if low <= dmax + max_back + dev:
        return 0, 0
    if lastx == be and be + 1 < len(P):
        return 0, 0
    return (1 if lastx + 1 == len(P) else 2), lastx - be


@pytest.mark.parametrize("case", ["adopt", "short_overlap", "step_mismatch", "not_on_segment", "d_too_close", "d_just_far_enough", "probe_went_back",
                                  "room_runs_out", "no_room", "unusable", "parts"])
def test_try_merge(lib, case):
    kw = dict(max_back=0, max_chosen=2, max_probe=50, k=14, dev=20, split=10 ** 7, has_size=0)
    P = line(1000, 40, 5000, step=10)
    T = line(100, 10, 4000) + P[5:12]          # the chain reached P[11] walking P[5..11] itself: 6 common vertices behind the pair
    parts = [10, 7]
    if case == "short_overlap":
        T = line(100, 10, 4000) + P[8:12]       # 3 common vertices only
        parts = [10, 4]
    elif case == "step_mismatch":
        T = line(100, 10, 4000) + [(v, s + (1 if i == 5 else 0), c) for i, (v, s, c) in enumerate(P[5:12])]
    elif case == "not_on_segment":
        T = line(100, 17, 4000)
        parts = [17]
    elif case == "d_too_close":
        T = [(100 + i, 3, 5076 + i) for i in range(10)] + P[5:12]   # a vertex only the chain has visited within reach of P's candidates
    elif case == "d_just_far_enough":
        # low = coord of P[be + 1 - max_chosen] = P[10] = 5100; D's highest coordinate must be < 5100 - 20 - 0
        T = line(100, 10, 4000)[:-1] + [(199, 3, 5079)] + P[5:12]
    elif case == "probe_went_back":
        kw["max_back"] = 1100                    # a probe of the segment's walk went 1100 below its branch vertex: D is within its reach
    elif case == "room_runs_out":
        kw["split"] = 14 + sum(x[1] for x in T) + 50 + 1 + 10 * 10   # room for 9 more steps of 10 (strictly below)
    elif case == "no_room":
        kw["split"] = 14 + sum(x[1] for x in T) + 50
    elif case == "parts":
        parts = [3, 4, 3, 2, 5]
    usable = case != "unusable"
    got = stitch(lib, T, parts, P, usable=usable, **kw)
    want = expect_merge(T, P, kw["max_back"], kw["max_chosen"], kw["max_probe"], kw["k"], kw["dev"], kw["split"], kw["has_size"]) if usable else (0, 0)
    assert (got[0], got[1]) == want, (case, got, want)
    expected_decision = {"adopt": 1, "short_overlap": 0, "step_mismatch": 0, "not_on_segment": 0, "d_too_close": 0, "d_just_far_enough": 1,
                         "probe_went_back": 0, "room_runs_out": 2, "no_room": 0, "unusable": 0, "parts": 1}[case]
    assert got[0] == expected_decision, (case, got)
    if got[0]:
        be = [x[0] for x in P].index(T[-1][0])
        assert got[5] == [x[0] for x in P[be + 1:be + 1 + got[1]]]           # the adopted stretch is P behind the junction
        assert got[3] == len(T) + got[1] and got[4] == sum(x[1] for x in T) + sum(x[1] for x in P[be + 1:be + 1 + got[1]])
    if case == "room_runs_out":
        assert got[1] == 9


# ---------------------------------------------------------------------------------------------------------------------
# inside the leaping zone
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,decision,why", [("adopt", 1, -1), ("no_log", 0, 0), ("not_a_boundary", 0, 2), ("cannot_leap_yet", 0, 3),
                                                ("window_top_differs", 0, 4), ("window_bottom_record", 0, 5), ("forced_record_below", 0, 5),
                                                ("segment_started_below_window", 0, 5), ("contig_following_record_in_d", 0, 6),
                                                ("coordinate_free_record_in_d", 0, 7), ("coordinate_free_record_beyond_d", 1, -1)])
def test_try_merge_leap(lib, case, decision, why):
    P = line(1000, 30, 90000)
    T = line(100, 10, 80000) + P[3:13]                   # T ends at P[12]; 9 common vertices behind the pair
    parts = [10, 10]
    log = [(i % 4 == 0, 90000 + 3 * i - 5, 0xFFFFFFFF) for i in range(30)]   # boundaries every 4th vertex, elow a little below them
    kw = dict(k=14, dev=20, split=1000, has_size=5000, wd_below=0, wd_forced=0xFFFFFFFF)
    if case == "no_log":
        log = None
    elif case == "not_a_boundary":
        log[12] = (False, 0, 0xFFFFFFFF)
    elif case == "cannot_leap_yet":
        kw["split"], kw["has_size"] = 10 ** 6, 0
    elif case == "window_top_differs":
        T = line(100, 9, 80000) + [(199, 3, 95000)] + P[3:13]     # the chain's window reaches higher than P[.. be]'s
    elif case == "window_bottom_record":
        kw["wd_below"] = 80000                                      # a window-dependent record examined at the chain's lowest coordinate
    elif case == "forced_record_below":
        kw["wd_forced"] = 79000
    elif case == "segment_started_below_window":
        P = [(999, 3, 70000)] + P
        T = line(100, 10, 80000) + P[4:14]
        log = [(False, 0, 0xFFFFFFFF)] + log
        log[13] = (True, 90030, 0xFFFFFFFF)
    elif case == "contig_following_record_in_d":
        log[16] = (True, 80010, 0xFFFFFFFF)                         # an iteration behind the junction examined a record at a coordinate of D
    elif case in ("coordinate_free_record_in_d", "coordinate_free_record_beyond_d"):
        T = line(100, 9, 80000) + [(7, 3, 0)] + P[3:13]             # the chain visited coordinate-free vertex 7 before the common stretch
        log[20] = (True, 90050, 5 if case.endswith("in_d") else 8)  # ... and a later iteration examined one at / beyond it (ids follow the reference)
    got = stitch(lib, T, parts, P, leap=True, log=log, **kw)
    assert (got[0], got[2]) == (decision, why), (case, got)
    if decision:
        be = [x[0] for x in P].index(T[-1][0])
        assert got[1] == len(P) - 1 - be and got[5] == [x[0] for x in P[be + 1:]]


def test_chain_before_over_parts(lib):
    for tables in (0, 1):
        lib.pagt_use_tables(tables)
        _chain_before_over_parts(lib)
    lib.pagt_use_tables(0)


def _chain_before_over_parts(lib):
    rng = np.random.default_rng(3)
    n = 200
    T = [(int(rng.integers(1, 10 ** 6)), 3, int(rng.integers(0, 2) * rng.integers(1, 10 ** 6))) for _ in range(n)]
    tv, ts, tpc = (_u32([x[i] for x in T]) for i in range(3))
    for parts in ([200], [1, 199], [50, 50, 50, 50], [7] * 28 + [4]):
        off = np.concatenate([[0], np.cumsum(parts)]).astype(np.uint64)
        lib.pagt_chain_before.argtypes = [C.c_void_p] * 4 + [C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
        for idx in (0, 1, 49, 50, 51, 199, 200):
            mx, m0 = C.c_uint32(), C.c_uint32()
            lib.pagt_chain_before(tv.ctypes.data, ts.ctypes.data, tpc.ctypes.data, off.ctypes.data, len(parts), idx, C.byref(mx), C.byref(m0))
            want_mx = max([x[2] for x in T[:idx]] + [0])
            want_m0 = max([x[0] + 1 for x in T[:idx] if x[2] == 0] + [0])
            assert (mx.value, m0.value) == (want_mx, want_m0), (parts, idx)


def test_block_tables_answer_like_the_entries(lib):
    """range_agg / range_xagg (walk_stitch.hpp) over random stretches of a long path: through the 64-entry block tables = over
    the entries = a plain restatement"""
    rng = np.random.default_rng(11)
    n = 3000
    v = _u32(rng.integers(1, 10 ** 6, n))
    s = _u32(rng.integers(1, 50, n))
    pc = _u32(np.where(rng.random(n) < 0.05, 0, rng.integers(1000, 10 ** 6, n)))
    xl = _u32(rng.integers(0, 10 ** 6, n))
    xh = _u32(np.where(rng.random(n) < 0.3, (1 << 31) | rng.integers(0, 10 ** 6, n), rng.integers(0, 10 ** 6, n)))
    lib.pagt_range_agg.argtypes = [C.c_void_p] * 3 + [C.c_uint64] * 3 + [C.c_void_p]
    lib.pagt_range_xagg.argtypes = [C.c_void_p] * 2 + [C.c_uint64] * 3 + [C.c_void_p]
    ranges = [(0, n), (0, 0), (5, 5), (0, 64), (1, 64), (0, 65), (63, 129), (64, 128), (100, 101), (2999, 3000), (2944, 3000), (2943, 2999)]
    ranges += [tuple(sorted(int(x) for x in rng.integers(0, n + 1, 2))) for _ in range(300)]
    for i, j in ranges:
        want = (int(pc[i:j].max(initial=0)), int(max([int(v[x]) + 1 for x in range(i, j) if pc[x] == 0] + [0])), int(pc[i:j].min(initial=0xFFFFFFFF)),
                int(min([int(pc[x]) for x in range(i, j) if pc[x] != 0] + [0xFFFFFFFF])), int(s[i:j].sum()))
        bd = [x for x in range(i, j) if xh[x] >> 31]
        wantx = (min([int(xh[x]) & 0x7FFFFFFF for x in bd] + [0xFFFFFFFF]), min([int(xl[x]) for x in bd] + [0xFFFFFFFF]))
        for tables in (0, 1):
            lib.pagt_use_tables(tables)
            out = (C.c_uint64 * 5)()
            lib.pagt_range_agg(v.ctypes.data, s.ctypes.data, pc.ctypes.data, n, i, j, out)
            assert tuple(int(x) for x in out) == want, (i, j, tables)
            outx = (C.c_uint32 * 2)()
            lib.pagt_range_xagg(xl.ctypes.data, xh.ctypes.data, n, i, j, outx)
            assert tuple(int(x) for x in outx) == wantx, (i, j, tables)
    lib.pagt_use_tables(0)


@pytest.mark.parametrize("seed", range(6))
def test_long_paths_through_the_block_tables(lib, seed):
    """adoptions at the scale of real jobs (thousands of entries, parts that begin and end inside blocks): the decision, the
    adopted stretch and the chain's aggregates are the same with and without the tables (stitch() asserts it), leap and not"""
    rng = np.random.default_rng(100 + seed)
    n_p = int(rng.integers(2000, 6000))
    P = line(5000, n_p, 200000, step=int(rng.integers(3, 9)))
    be = int(rng.integers(200, 700))
    t = int(rng.integers(6, 150))
    head = line(100, int(rng.integers(300, 3000)), 100000)
    T = head + P[be - t:be + 1]
    cut = sorted(set(int(x) for x in rng.integers(1, len(T) - 1, 5)))
    parts = [b - a for a, b in zip([0] + cut, cut + [len(T)])]
    got = stitch(lib, T, parts, P, max_back=int(rng.integers(0, 50)), max_chosen=int(rng.integers(1, 5)), max_probe=50, k=14, dev=20, split=10 ** 9, has_size=0)
    assert got[0] == 1 and got[1] == n_p - 1 - be
    # room runs out somewhere inside the rest
    size_T = sum(x[1] for x in T)
    room_steps = int(rng.integers(10, n_p - be - 10))
    got = stitch(lib, T, parts, P, max_back=0, max_chosen=2, max_probe=50, k=14, dev=20, split=14 + size_T + 50 + 1 + room_steps * P[0][1] + 1, has_size=0)
    assert got[0] == 2 and got[1] == room_steps
    # the leaping zone: boundaries everywhere, one iteration far behind the junction examined a record inside D / none did
    log = [(True, 200000 + P[0][1] * i - 5, 0xFFFFFFFF) for i in range(n_p)]
    ok = stitch(lib, T, parts, P, leap=True, log=log, k=14, dev=20, split=1000, has_size=5000)
    assert (ok[0], ok[2]) == (1, -1) and ok[1] == n_p - 1 - be
    far = int(rng.integers(be + 70, n_p))
    log[far] = (True, 100010, 0xFFFFFFFF)
    bad = stitch(lib, T, parts, P, leap=True, log=log, k=14, dev=20, split=1000, has_size=5000)
    assert (bad[0], bad[2]) == (0, 6)


# ---------------------------------------------------------------------------------------------------------------------
# what a chain does next (advance_chain): the two kinds of segments overlap around the point where leaping begins
# ---------------------------------------------------------------------------------------------------------------------
def advance(lib, T, segs, n_spec, zone_end, *, parts=None, split=10 ** 9, has_size=0, k=14, dev=20, seg_ov=150):
    """segs: list of dicts x, path (list of (v, s, pc)), done, stopped, usable, log (leap pieces)"""
    flat = [e for sg in segs for e in sg["path"]]
    off = np.concatenate([[0], np.cumsum([len(sg["path"]) for sg in segs])]).astype(np.uint64)
    pv, ps, ppc = (_u32([e[i] for e in flat] + [0]) for i in range(3))
    logs = [lg for sg in segs for lg in sg.get("log", [(False, 0, 0xFFFFFFFF)] * len(sg["path"]))]
    xl = _u32([m0 for _, _, m0 in logs] + [0])
    xh = _u32([((1 << 31) | (e & 0x7FFFFFFF)) if b else 0 for b, e, _ in logs] + [0])
    tv, ts, tpc = (_u32([e[i] for e in T]) for i in range(3))
    parts = parts or [len(T)]
    poff = np.concatenate([[0], np.cumsum(parts)]).astype(np.uint64)
    sx, sdone, sstop, suse = (_u32([int(sg.get(key, default)) for sg in segs] + [0]) for key, default in (("x", 0), ("done", True), ("stopped", True), ("usable", True)))
    out = (C.c_int64 * 12)()
    lib.pagt_advance.argtypes = [C.c_uint32] * 3 + [C.c_void_p] * 14 + [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    res = []
    for tables in (0, 1):
        lib.pagt_use_tables(tables)
        lib.pagt_advance(len(segs), n_spec, zone_end, sx.ctypes.data, sdone.ctypes.data, sstop.ctypes.data, suse.ctypes.data, off.ctypes.data, pv.ctypes.data, ps.ctypes.data, ppc.ctypes.data, xl.ctypes.data, xh.ctypes.data,
                         tv.ctypes.data, ts.ctypes.data, tpc.ctypes.data, poff.ctypes.data, len(parts), k, dev, split, has_size, seg_ov, out)
        res.append(dict(zip(("what", "stop", "until_leap", "len", "next_seg", "next_leap", "waiting", "final", "adopted", "fails", "leap_adopted", "why"),
                            [int(x) for x in out])))
    lib.pagt_use_tables(0)
    assert res[0] == res[1]
    return res[0]


def _round(n_vertices=400, step=10, c0=100000, cuts=(60, 160, 260), m=20):
    """a strand of n_vertices vertices; segments that cannot leap start at the vertices `cuts` and run m vertices into the next"""
    B = line(1000, n_vertices, c0, step=step)
    ends = list(cuts[1:]) + [n_vertices - m]
    segs = [{"x": B[a][2], "path": B[a:b + m]} for a, b in zip(cuts, ends)]
    T = B[:cuts[0] + m]
    return B, T, segs


def test_advance_adopts_segment_after_segment_and_waits_for_an_unfinished_one(lib):
    B, T, segs = _round()
    zone_end = B[-1][2] + 1000
    segs[2]["done"] = False
    r = advance(lib, T, segs, 3, zone_end)
    assert (r["what"], r["waiting"], r["next_seg"], r["len"]) == (0, 2, 2, 260 + 20) and r["adopted"] == 260 + 20 - len(T)
    segs[2]["done"] = True
    segs[2]["stopped"] = False                      # the last segment's walk ended by itself: so does the real one
    r = advance(lib, T, segs, 3, zone_end)
    assert (r["what"], r["final"], r["len"]) == (0, 1, len(B))


def test_advance_walks_on_exactly_where_nothing_can_be_adopted(lib):
    B, T, segs = _round()
    zone_end = B[-1][2] + 1000
    segs[1]["usable"] = False                       # (its job overflowed, say): adopted up to it, then exactly to the next checkpoint
    r = advance(lib, T, segs, 3, zone_end)
    assert (r["what"], r["until_leap"], r["fails"], r["next_seg"]) == (1, 0, 1, 2)
    assert r["stop"] == segs[2]["x"] + 150 and r["len"] == 160 + 20
    # a chain that has not reached the first checkpoint walks on to it
    r = advance(lib, T[:30], segs, 3, zone_end)
    assert (r["what"], r["stop"], r["len"]) == (1, segs[0]["x"] + 150, 30)


def test_advance_crosses_the_point_where_leaping_begins(lib):
    """the kinds overlap: segments that cannot leap are adopted up to where the chain's TRUE size allows, a short exact walk
    (until_leap) crosses the point, the pieces of the leaping zone started before it take over"""
    B, T, segs = _round()
    step = 10
    zone_end = B[-1][2] + 1000
    # leaping begins when k + steps reaches the size at vertex 200 (inside segment 1)
    split = 14 + 200 * step
    leap = [{"x": B[a][2], "path": B[a:], "stopped": False, "log": [(True, B[a + i][2] - 5, 0xFFFFFFFF) for i in range(len(B) - a)]} for a in (150, 230)]
    allsegs = segs + leap
    r = advance(lib, T, allsegs, 3, zone_end, split=split)
    # cut max_probe + 1 = 51 steps before the point (5 vertices and a bit), nothing of the leaping zone tried yet: cross the point exactly
    assert (r["what"], r["until_leap"], r["stop"]) == (1, 1, 0) and r["next_seg"] == 3 and r["leap_adopted"] == 0
    assert 190 <= r["len"] <= 196 and r["fails"] == 0
    # ... the exact walk has ended a few vertices behind the point (its path: the strand's own vertices): the piece that started
    # BEFORE the point is adopted to its end, which is the end of the strand
    T2 = B[:203]
    r = advance(lib, T2, allsegs, 3, zone_end, split=split, parts=[80, 123])
    assert (r["what"], r["final"], r["len"], r["leap_adopted"], r["why"]) == (0, 1, len(B), 1, -1)
    # without pieces of the leaping zone the exact walk goes to the end
    r = advance(lib, T, segs, 3, zone_end, split=split)
    assert (r["what"], r["until_leap"], r["stop"]) == (1, 0, 0)
    # a chain that can leap never adopts a segment that cannot (and the other way round): with only such segments left it walks on
    r = advance(lib, T2, segs, 3, zone_end, split=split)
    assert (r["what"], r["until_leap"], r["stop"], r["len"]) == (1, 0, 0, len(T2))


def test_advance_past_the_zone_of_the_segments_that_cannot_leap(lib):
    B, T, segs = _round()
    zone_end = B[170][2]                            # the zone ends inside segment 1
    leap = [{"x": B[300][2], "path": B[300:], "stopped": False, "log": [(True, B[300 + i][2] - 5, 0xFFFFFFFF) for i in range(len(B) - 300)]}]
    for sg in segs:
        sg["path"] = [e for e in sg["path"] if e[2] <= zone_end + 200]
    r = advance(lib, T, segs[:2] + leap, 2, zone_end)
    # adopted to the end of the zone; leaping is still impossible (true size): across the point, however far that is
    assert (r["what"], r["until_leap"], r["next_seg"]) == (1, 1, 2)


# ---------------------------------------------------------------------------------------------------------------------
# segments kept from an earlier round of the contig (the walk dead-ended, was committed, the next round re-seeds behind it)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("g_hi,decision", [(0, 1), (5079, 1), (5080, 0), (6000, 0)])
def test_kept_segment_outside_the_leaping_zone_counts_the_marks_committed_since(lib, g_hi, decision):
    """the vertices the rounds since have marked globally belong to D: the segment's candidates (lowest coordinate 5100, deviation
    20) must lie above all of them"""
    kw = dict(max_back=0, max_chosen=2, max_probe=50, k=14, dev=20, split=10 ** 7, has_size=0)
    P = line(1000, 40, 5000, step=10)
    T = line(100, 10, 4000) + P[5:12]
    same_round = stitch_kept(lib, T, [10, 7], P, (2, 2, 3000, 6000, 0), **kw)
    assert same_round[0] == 1  # (a segment of the running round: the global marks were there when it was walked)
    got = stitch_kept(lib, T, [10, 7], P, (1, 2, 3000 if g_hi else 0xFFFFFFFF, g_hi, 0), **kw)
    assert got[0] == decision, got


@pytest.mark.parametrize("case,decision,why", [("adopt", 1, -1), ("global_coordinate_in_reach", 0, 6), ("global_free_vertex_in_reach", 0, 7),
                                                ("forced_record_inside_the_global_window", 1, -1), ("forced_record_below_the_global_window", 0, 5),
                                                ("forced_record_in_a_gap_between_the_windows", 0, 5), ("record_below_the_forced_window_inside_the_global_one", 0, 5)])
def test_kept_segment_of_the_leaping_zone(lib, case, decision, why):
    P = line(1000, 30, 90000)
    T = line(100, 10, 80000) + P[3:13]
    log = [(i % 4 == 0, 90000 + 3 * i - 5, 0xFFFFFFFF) for i in range(30)]
    kw = dict(k=14, dev=20, split=1000, has_size=5000, wd_below=0, wd_forced=0xFFFFFFFF)
    g_lo, g_hi, g_free = 60000, 80000, 0          # the rounds since committed a path over [60000, 80000]: it touches T's window (from 80000)
    if case == "global_coordinate_in_reach":
        g_hi = 90040                              # ... up to a coordinate an iteration behind the junction examined (elow 90031 at P[12])
    elif case == "global_free_vertex_in_reach":
        g_free = 9
        log[20] = (True, 90050, 5)                # a later iteration examined coordinate-free vertex 5 <= the highest one committed (8)
    elif case == "forced_record_inside_the_global_window":
        kw["wd_forced"] = 70000                   # rejected by the segment (inside its forced window), by the real walk through the global window
    elif case == "forced_record_below_the_global_window":
        kw["wd_forced"] = 50000                   # rejected by the segment, ACCEPTED by the real walk (below both of its windows)
    elif case == "forced_record_in_a_gap_between_the_windows":
        g_hi = 75000                              # the global window ends below T's: a record at 78000 would fall between them
        kw["wd_forced"] = 78000
    elif case == "record_below_the_forced_window_inside_the_global_one":
        kw["wd_below"] = 65000                    # accepted by the segment (below its window), rejected by the real walk (global window)
    got = stitch_kept(lib, T, [10, 10], P, (1, 2, g_lo, g_hi, g_free), leap=True, log=log, **kw)
    assert (got[0], got[2]) == (decision, why), (case, got)
