"""GPU box: the successor stage alone (pag_process + pag_travel_prepare_for, no walks) on a bench-style block, a few times.
Prints the stage's wall time per repetition and the view's sizes; run it under rocprofv3 for the kernels (tests/succ_stage_probe.sh).

    python tests/succ_stage_bench.py [--reads 100000 --ref-len 50000000 --reps 4]      # BASELINE configs[1] by default
    python tests/succ_stage_bench.py --reads 93750 --ref-len 31250000                  # the same tuples at 30x coverage"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--read-span", type=int, default=10_000)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import pagctl
    from aligngraph2_amd import workload as biggen
    hip, host = bench.load_libs()
    spec = biggen.BigSpec(seed=args.seed, ref_len=args.ref_len, n_reads=args.reads, read_span=args.read_span, k=14, eps=10, cov=2, threads=16)
    w = biggen.BigWorkload(spec, device="cuda:0")
    torch.cuda.synchronize()
    raw = w.raw_input()
    inp = biggen.PagBuildInput()
    hip.pag_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.pag_prepare.restype = C.c_int
    err = C.c_int()
    g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, spec.k, 1, 0, C.byref(err))
    assert g, hip.pag_last_error()
    g = C.c_void_p(g)
    assert hip.pag_prepare(g, C.byref(raw), C.byref(inp)) == 0, hip.pag_last_error()
    ctg_seqs, keep1 = bench.host_seqs(w.contig_codes())
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    ref_len = np.array([len(w.ref)], dtype=np.uint32)

    class TravelParams(C.Structure):
        _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                    ("start_split", C.c_double), ("min_len", C.c_uint64)]
    prm = TravelParams(spec.threads, 0, 2 * spec.eps, 0.15, 0.90, 50)
    hip.pag_travel_prepare_for.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_travel_prepare_for.restype = C.c_int
    hip.pag_travel_view_sizes.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    for rep in range(args.reps):
        st = pagctl.BuildStats()
        assert hip.pag_process(g, C.byref(inp), C.byref(st)) == 0, hip.pag_last_error()
        ms = C.c_double()
        t0 = time.perf_counter()
        rc = hip.pag_travel_prepare_for(g, C.byref(ctg_seqs), orient.ctypes.data, ref_len.ctypes.data, 1, C.byref(prm), C.byref(ms))
        assert rc == 0, hip.pag_last_error()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        nn, npos, ne, ns = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        cut, fb = C.c_int(), C.c_uint64()
        hip.pag_travel_view_sizes(g, C.byref(nn), C.byref(npos), C.byref(ne), C.byref(ns), C.byref(cut), C.byref(fb))
        print(f"rep {rep}: successor stage {dt:.1f} ms (reported {ms.value:.1f}); view: {nn.value} nodes, {npos.value} vertices, {ne.value} edges, {ns.value} records; "
              f"graph: {st.n_pos} vertices; read bases {w.n_bases}", flush=True)
    hip.pag_destroy(g)


if __name__ == "__main__":
    main()
