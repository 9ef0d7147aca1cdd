"""Development aid (GPU): the traversal of one bench-scale workload three ways — device walkers with the speculative
choice, device walkers in exact mode (PAG_WALK_EXACT=1), and the host restatement of the reference's traversal over the
exported graph — and their fingerprints side by side.  python tests/walk_check.py [--reads N --ref-len L --k K] [--no-host]"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--read-span", type=int, default=10000)
    ap.add_argument("--ref-len", type=int, default=10_000_000)
    ap.add_argument("--k", type=int, default=14)
    ap.add_argument("--epsilon", type=int, default=10)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--no-host", action="store_true")
    ap.add_argument("--modes", default="exact,speculative,uncut", help="device modes to run besides the host walk")
    ap.add_argument("--min-abundance", type=int, default=-1, help=">= 0: abundance cut of the solid set (needed for k > 14)")
    args = ap.parse_args()
    import torch
    import bench
    import biggen
    import pagctl
    hip, host = bench.load_libs()
    sp = biggen.BigSpec(seed=args.seed, ref_len=args.ref_len, n_reads=args.reads, read_span=args.read_span, k=args.k,
                        eps=args.epsilon, cov=2, threads=16, solid_min_abundance=args.min_abundance)
    w = biggen.BigWorkload(sp, device="cuda:0")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    assert g, hip.pag_last_error()
    st = pagctl.BuildStats()
    assert hip.pag_process(g, C.byref(inp), C.byref(st)) == 0, hip.pag_last_error()
    ref_np = w.ref.cpu().numpy()
    ctg_codes = w.contig_codes()
    ctg_seqs, k1 = bench.host_seqs(ctg_codes)
    ref_seqs, k2 = bench.host_seqs([ref_np])
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    hostwalk = pagctl.walk_test_lib().pagt_traverse_hostwalk
    hostwalk.argtypes = host.pagh_traverse.argtypes
    res = {}
    for mode in (["host walk"] if not args.no_host else []) + [m for m in args.modes.split(",") if m]:
        out = tempfile.mkdtemp(prefix="walkcheck_", dir="/dev/shm")
        ts = bench.TraverseStats()
        os.environ.pop("PAG_WALK_EXACT", None)
        os.environ.pop("PAG_WALK_PIECES", None)
        if mode == "exact":
            os.environ["PAG_WALK_EXACT"] = "1"
        if mode == "uncut":  # one job per (contig, seed): the walk without the segment-parallel cut
            os.environ["PAG_WALK_PIECES"] = "0"
        fn = hostwalk if mode == "host walk" else host.pagh_traverse
        t0 = time.time()
        rc = fn(g, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, sp.threads, sp.eps, 50,
                out.encode(), b"0_", 0, C.byref(ts))
        assert rc == 0, host.pagh_last_error()
        res[mode] = (int(ts.n_path_nodes), int(ts.n_path_bases), f"{int(ts.path_checksum):016x}")
        print(f"{mode:18s} {res[mode]}  {time.time() - t0:.1f} s (walk {ts.ms_walk:.0f} ms, {ts.walk_jobs} jobs, successor records {ts.ms_successors:.0f} ms)", flush=True)
        files = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
        if "files" in res:
            bad = [f for f in files if files[f] != res["files"].get(f)]
            print(f"    files differing from the first mode: {len(bad)} of {len(files)} {bad[:5]}", flush=True)
            for f in [b for b in bad if b.endswith(".txt")][:6]:
                a, b = res["files"].get(f, b"").split(b"\n"), files[f].split(b"\n")
                i = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
                print(f"      {f}: {len(a)} vs {len(b)} lines, first difference at line {i}", flush=True)
                for j in range(max(0, i - 2), min(i + 3, max(len(a), len(b)))):
                    print(f"        [{j}] {a[j].decode() if j < len(a) else '-':60s} | {b[j].decode() if j < len(b) else '-'}", flush=True)
        else:
            res["files"] = files
    host.pagh_release(g)
    hip.pag_destroy(g)
    vals = {v for k, v in res.items() if k != "files"}
    print("ALL EQUAL" if len(vals) == 1 else "MISMATCH")
    return 0 if len(vals) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
