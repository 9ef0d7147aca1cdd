"""GPU: the device match predicates (d_check_position / isEdgeSimilar inside it, csrc/hip/k5_travel.hip) in isolation.

(1) every row of the reference's own truth table tests/golden/func_predicate.txt.gz (checkPosition / isEdgeSimilar of
    PAGraph/src/tools/graph/PABruijnGraph.cpp:143-165, 385-400, written by oracle/ref_harness/func_golden.cpp);
(2) a dense sweep around the 0.15 ratio boundary — where the kernel's division-free early reject (d_ratio_ok) decides —
    against the C oracle, which is pinned on the same table: every step 1..4000 with the coordinate difference within
    +-3 of both band edges, for all four zero patterns, plus u32 wrap-around bases.
Both through the plain predicate and the way the successor kernels run it: the ratio tests of steps below 1024 looked up in a
table of integer intervals the kernel derives from d_ratio_ok itself (d_ratio_entry / d_check_position_tab)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

import goldens
import pagctl


def _device(rows, err=0.15, table=False):
    hip = pagctl.hip_lib()
    rows = np.ascontiguousarray(rows, dtype=np.uint32)
    n = len(rows)
    grade = np.zeros(n, np.uint8)
    es = np.zeros(n, np.uint8)
    if table:
        through = C.c_uint64()
        hip.pag_debug_predicates_tab.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        rc = hip.pag_debug_predicates_tab(rows.ctypes.data, n, err, grade.ctypes.data, es.ctypes.data, 0, C.byref(through))
        assert rc == 0, hip.pag_last_error()
        # every step 1 .. 1023 has a table entry (an interval that could not be pinned down would fall back to the plain path)
        assert through.value == int(((rows[:, 4] > 0) & (rows[:, 4] < 1024)).sum())
        return grade, es
    hip.pag_debug_predicates.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_int]
    rc = hip.pag_debug_predicates(rows.ctypes.data, n, err, grade.ctypes.data, es.ctypes.data, 0)
    assert rc == 0, hip.pag_last_error()
    return grade, es


@pytest.mark.gpu
@pytest.mark.parametrize("table", [False, True])
def test_device_predicates_equal_the_reference_truth_table(table):
    rows = np.loadtxt(gzip.open(os.path.join(goldens.GOLDEN, "func_predicate.txt.gz")), dtype=str)
    assert len(rows) > 30000
    q = rows[:, :6].astype(np.uint64).astype(np.uint32)
    grade, es = _device(q, table=table)
    want_grade = rows[:, 6].astype(np.uint8)
    want_es = np.array([int(x[0]) | (int(x[1]) << 1) for x in rows[:, 7]], np.uint8)
    bad = np.flatnonzero((grade != want_grade) | (es != want_es))
    assert len(bad) == 0, f"{len(bad)} rows differ, first: {rows[bad[0]]} device grade {grade[bad[0]]} edge-sim {es[bad[0]]}"


@pytest.mark.gpu
@pytest.mark.parametrize("table", [False, True])
def test_device_predicates_equal_the_oracle_around_the_ratio_boundary(table):
    lib = pagctl.oracle_lib()
    rows = []
    for dist in range(1, 4001):
        for edge in (0.85 * dist, 1.15 * dist, 0.84 * dist, 1.16 * dist):
            for d in range(int(edge) - 3, int(edge) + 4):
                if d < 0:
                    continue
                for base_c, base_r in ((1000, 7000), (4294967290, 5), (0, 7000), (1000, 0)):
                    bc = (base_c + d) & 0xFFFFFFFF if base_c else 0
                    br = (base_r + d) & 0xFFFFFFFF if base_r else 0
                    rows.append((base_c, base_r, bc, br, dist, 20))
                    rows.append((base_c, base_r, bc, (base_r + dist) & 0xFFFFFFFF if base_r else 0, dist, 10))
                    rows.append((base_c, base_r, 0, br, dist, 10))
    rows = np.array(rows, dtype=np.uint64).astype(np.uint32)
    grade, es = _device(rows, table=table)
    want_g = np.empty(len(rows), np.uint8)
    want_e = np.empty(len(rows), np.uint8)
    for i, (a1, a2, b1, b2, dist, dev) in enumerate(rows.tolist()):
        want_g[i] = lib.pago_check_position(a1, a2, b1, b2, dist, dev, 0.15)
        want_e[i] = lib.pago_edge_similar(a1, a2, b1, b2, dist, dev, 0.15)
    bad = np.flatnonzero((grade != want_g) | (es != want_e))
    assert len(bad) == 0, f"{len(bad)} of {len(rows)} rows differ, first: {rows[bad[0]].tolist()} device ({grade[bad[0]]}, {es[bad[0]]}) oracle ({want_g[bad[0]]}, {want_e[bad[0]]})"
