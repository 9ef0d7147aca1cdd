"""GPU: the successor records of the traversal graph (searchSuccessors + checkPosition for every vertex,
PABruijnGraph.cpp:143-197; k5_travel.hip k_succ_emit -> sort by source -> k_succ_finish) must be the same arrays, record for
record, however the work is dealt out: vertices with many candidate pairs by a whole wave at the default limit, at a limit of 4
(most vertices go that way) and never; the coordinate order applied in one and in eight slices of the vertex id range
(k_order_apply); an emission stream that starts too small and is made again."""
import ctypes as C
import os

import numpy as np
import pytest

import pagctl


class TravelParams(C.Structure):
    _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                ("start_split", C.c_double), ("min_len", C.c_uint64)]


@pytest.mark.gpu
def test_successor_records_do_not_depend_on_how_they_are_built(monkeypatch):
    import torch
    import bench
    import biggen
    hip, host = bench.load_libs()
    sp = biggen.BigSpec(seed=21, ref_len=1_500_000, n_reads=6000, read_span=4000, k=14, eps=10, ctg_len=300_000, gap_lo=300, gap_hi=3000,
                        rev_ctg_frac=0.3, threads=16, cov=2, solid_min_abundance=2, chunk_reads=512)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    hip.pag_create_from_bitmap.restype = C.c_void_p
    g = C.c_void_p(hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err)))
    ctg_seqs, keep = bench.host_seqs(w.contig_codes())
    ref_len = np.array([len(w.ref)], dtype=np.uint32)
    prm = TravelParams(sp.threads, 0, 2 * sp.eps, 0.15, 0.90, 50)
    hip.pag_travel_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_debug_succ_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.pag_debug_succ.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    got = {}
    for mode in ("default", "heavy=4", "heavy=0", "order=8", "order=1", "stream=4096"):
        monkeypatch.delenv("PAG_SUCC_HEAVY", raising=False)
        monkeypatch.delenv("PAG_DEBUG_EMIT_CAP", raising=False)
        monkeypatch.delenv("PAG_ORDER_SLICES", raising=False)
        if "heavy=" in mode:
            monkeypatch.setenv("PAG_SUCC_HEAVY", mode.split("=")[1])
        if "order=" in mode:  # (k_order_apply's scatter / gather in that many slices of the vertex id range)
            monkeypatch.setenv("PAG_ORDER_SLICES", mode.split("=")[1])
        if "stream=" in mode:  # (the first emission stream far too small: made again with what the waves asked for)
            monkeypatch.setenv("PAG_DEBUG_EMIT_CAP", mode.split("=")[1])
        st = pagctl.BuildStats()
        assert hip.pag_process(g, C.byref(inp), C.byref(st)) == 0, hip.pag_last_error()
        assert hip.pag_travel_prepare(g, C.byref(ctg_seqs), ref_len.ctypes.data, 1, C.byref(prm), None) == 0, hip.pag_last_error()
        n_pos, n_succ = C.c_uint64(), C.c_uint64()
        assert hip.pag_debug_succ_sizes(g, C.byref(n_pos), C.byref(n_succ)) == 0
        off = np.zeros(n_pos.value + 1, dtype=np.uint32)
        recs = np.zeros((n_succ.value, 4), dtype=np.uint32)
        assert hip.pag_debug_succ(g, off.ctypes.data, recs.ctypes.data) == 0, hip.pag_last_error()
        got[mode] = (off, recs)
    off0, recs0 = got["default"]
    assert len(recs0) > 2 * len(off0) * 0.5 and int(off0[-1]) == len(recs0)
    for mode, (off, recs) in got.items():
        assert np.array_equal(off, off0), f"{mode}: offsets differ from the default run"
        assert np.array_equal(recs, recs0), f"{mode}: records differ from the default run"
    hip.pag_destroy(g)
