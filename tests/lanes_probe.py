"""One-off measurement (GPU box): blocks of BASELINE configs[1] through L independent lanes on ONE device — every lane a graph handle
of its own (own stream, own walk arena, own host half) driven by a host thread of its own — against one lane.  Blocks of a run are
independent (PAssembly's block loop), so lanes are pipeline parallelism over blocks: the walks of one block (latency-bound, three
waves per CU) beside the build / successor stage of another (bandwidth-bound).  Prints blocks/s per setting."""
import argparse
import ctypes as C
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


class TravelParams(C.Structure):
    _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                ("start_split", C.c_double), ("min_len", C.c_uint64)]


def main():
    import torch
    from aligngraph2_amd import parallel
    from aligngraph2_amd import workload as biggen
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--lanes", default="1,2")
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    args = ap.parse_args()
    hip, host = bench.load_libs()
    spec = biggen.BigSpec(seed=2, ref_len=args.ref_len, n_reads=args.reads, read_span=10_000, k=14, eps=10, cov=2, threads=16)
    w = biggen.BigWorkload(spec, device="cuda:0")
    torch.cuda.synchronize()
    raw = w.raw_input()
    hip.pag_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.pag_prepare.restype = C.c_int
    hip.pag_travel_prepare_for.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_travel_prepare_for.restype = C.c_int
    ref_np = w.ref.cpu().numpy()
    ctg_seqs, keep1 = bench.host_seqs(w.contig_codes())
    ref_seqs, keep2 = bench.host_seqs([ref_np])
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    ref_len_u32 = np.array([len(ref_np)], dtype=np.uint32)
    prm = TravelParams(spec.threads, 0, 2 * spec.eps, 0.15, 0.90, 50)
    results = {}

    class Lane:
        def __init__(self, idx):
            err = C.c_int()
            self.g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, spec.k, 1, 0, C.byref(err))
            assert self.g, hip.pag_last_error()
            self.inp = biggen.PagBuildInput()
            self.st = parallel.BuildStats()
            self.ts = bench.TraverseStats()
            self.out = tempfile.mkdtemp(prefix=f"lane{idx}_", dir="/dev/shm")
            self.pending = False
            self.sums = set()
            self.t = {"prepare": 0.0, "process": 0.0, "succ": 0.0, "collect": 0.0, "walks": 0.0}

        def block(self):
            g = self.g
            t0 = time.perf_counter()
            assert hip.pag_prepare(g, C.byref(raw), C.byref(self.inp)) == 0, hip.pag_last_error()
            t1 = time.perf_counter()
            assert hip.pag_process(g, C.byref(self.inp), C.byref(self.st)) == 0, hip.pag_last_error()
            t2 = time.perf_counter()
            assert hip.pag_travel_prepare_for(g, C.byref(ctg_seqs), orient.ctypes.data, ref_len_u32.ctypes.data, 1, C.byref(prm), None) == 0, hip.pag_last_error()
            t3 = time.perf_counter()
            self.collect()
            t4 = time.perf_counter()
            assert host.pagh_traverse_begin(g, spec.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, spec.threads,
                                            spec.eps, 50, self.out.encode(), b"0_", 0) == 0, host.pagh_last_error()
            t5 = time.perf_counter()
            for kk, d in zip(("prepare", "process", "succ", "collect", "walks"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                self.t[kk] += d
            self.pending = True

        def collect(self):
            if self.pending:
                assert host.pagh_traverse_end(self.g, C.byref(self.ts)) == 0, host.pagh_last_error()
                self.sums.add(f"{int(self.ts.path_checksum):016x}")
                self.pending = False

        def run(self, n):
            for _ in range(n):
                self.block()
            self.collect()

    lanes = []
    for L in [int(x) for x in args.lanes.split(",")]:
        while len(lanes) < L:
            lanes.append(Lane(len(lanes)))
            lanes[-1].run(1)  # warm-up: arenas, host caches
        torch.cuda.synchronize()
        for ln in lanes:
            ln.t = {kk: 0.0 for kk in ln.t}
        t0 = time.perf_counter()
        thr = [threading.Thread(target=ln.run, args=(args.steps,)) for ln in lanes[:L]]
        for t in thr:
            t.start()
        for t in thr:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sums = set().union(*[ln.sums for ln in lanes[:L]])
        free_b, total_b = torch.cuda.mem_get_info()
        results[L] = dt / (L * args.steps) * 1e3
        print(f"lanes {L}: {L * args.steps} blocks in {dt:.2f} s = {results[L]:.1f} ms per block; checksums {sorted(sums)}; device memory in use {(total_b - free_b) / 1e9:.1f} GB", flush=True)
        for ln in lanes[:L]:
            print("    per block of a lane, ms: " + ", ".join(f"{kk} {v / args.steps * 1e3:.1f}" for kk, v in ln.t.items()), flush=True)


if __name__ == "__main__":
    main()
