"""BASELINE configs[2] (1 M x 10 kb reads vs a 250 Mb reference, ref-block shard across 4 GPUs) executed on ONE MI355X: the N
ranks of the sharded build run one after the other (aligngraph2_amd/rank_serial.py), what a rank would take in over xGMI is
recomputed on the device when its turn comes (no host spills).  Writes a JSON record: geometry, per-rank per-stage device
bytes as measured (next to DESIGN.md §7's table), wire bytes, count lines, held fractions, times, one digest over the block's
output files per N.

    python tests/c3_rank_serial.py OUT.json                       # configs[2] geometry, N = 4 and N = 8
    python tests/c3_rank_serial.py OUT.json --reads 100000 --ref-len 50000000 --one-gpu   # a size one GPU holds: also == one GPU
    ... --reads 100000 --ref-len 50000000 --reference-digests profiles/r04_c2_text_parity.json   # BASELINE configs[1]: every output
                                            # file against the SHA-256 of what the compiled reference wrote for this workload

Asserted: count lines of every rank = sums over the owners; every rank holds < 1/N + 0.15 of the vertices; no walk leaves its
region (pag_travel would fail with PAG_ERANGE); the outputs are identical for every N (and equal to the one-GPU run's when
--one-gpu)."""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--ref-len", type=int, default=250_000_000)
    ap.add_argument("--read-span", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=14)
    ap.add_argument("--epsilon", type=int, default=10)
    ap.add_argument("--ranks", default="4,8")
    ap.add_argument("--halo", type=int, default=200_000)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--one-gpu", action="store_true", help="also run the block on one handle (it must fit) and compare the outputs")
    ap.add_argument("--solid-min-abundance", type=int, default=-1)
    ap.add_argument("--reference-digests", default=None, help="a JSON with compare.reference_sha256 (tests/c2_text_runs.py --compare) for THIS workload: "
                    "every output file of every run is held against the reference's digest of it")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bench
    import biggen
    import pagctl
    from aligngraph2_amd import rank_serial

    hip, host = bench.load_libs()
    dev = "cuda:0"
    rec = {"what": "one config block as N ranks run one after the other on ONE MI355X (aligngraph2_amd/rank_serial.py): every kernel, every "
                   "byte exchanged is the N-GPU run's; no multi-GPU TIMING exists (the build's boxes have one GPU)",
           "geometry": {"reads": args.reads, "read_span": args.read_span, "ref_len": args.ref_len, "k": args.k, "epsilon": args.epsilon, "seed": args.seed},
           "runs": []}

    def save():
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)

    t0 = time.perf_counter()
    spec = biggen.BigSpec(seed=args.seed, ref_len=args.ref_len, n_reads=args.reads, read_span=args.read_span, k=args.k, eps=args.epsilon, cov=2,
                          threads=16, solid_min_abundance=args.solid_min_abundance)
    w = biggen.BigWorkload(spec, device=dev)
    torch.cuda.synchronize()
    rec["s_generate"] = time.perf_counter() - t0
    rec["geometry"].update(read_bases=int(w.n_bases), contigs=len(w.ctgs), solid_kmers=int(w.n_solid), min_abundance=int(w.min_abundance))
    free, total = torch.cuda.mem_get_info(dev)
    rec["device_total_bytes"] = int(total)
    print(f"generated in {rec['s_generate']:.1f} s: {w.n_bases} read bases, {len(w.ctgs)} contigs, {w.n_solid} solid k-mers; "
          f"{(total - free) / 1e9:.1f} GB on the device", flush=True)
    save()

    inp = w.build_input()
    ref_np = w.ref.cpu().numpy()
    ctg_seqs, keep1 = bench.host_seqs(w.contig_codes())
    ref_seqs, keep2 = bench.host_seqs([ref_np])
    orient = [0 if r else 1 for _, _, r in w.ctgs]
    ctg_len = [e - s for s, e, _ in w.ctgs]
    g2r = w.g2r.cpu().numpy()
    alns = [(c, 0, int(g2r[s]), int(g2r[e - 1]) + 1) for c, (s, e, _) in enumerate(w.ctgs)]
    del g2r

    def make_handle():
        err = C.c_int()
        g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, spec.k, 1, 0, C.byref(err))
        if not g:
            raise RuntimeError(f"pag_create_from_bitmap failed ({err.value}): {hip.pag_last_error().decode()}")
        return g

    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    digests = {}
    if args.one_gpu:
        out = tempfile.mkdtemp(prefix="pagc3_one_", dir=shm)
        g = make_handle()
        st = pagctl.BuildStats()
        t1 = time.perf_counter()
        if hip.pag_process(C.c_void_p(g), C.byref(inp), C.byref(st)) != 0:
            raise SystemExit("pag_process: " + hip.pag_last_error().decode())
        ts = bench.TraverseStats()
        o_arr = np.array(orient, dtype=np.int32)
        rc = host.pagh_traverse(g, spec.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, o_arr.ctypes.data, spec.threads, spec.eps, 50, out.encode(),
                                b"0_", 0, C.byref(ts))
        if rc != 0:
            raise SystemExit("pagh_traverse: " + host.pagh_last_error().decode())
        dg, nbytes = rank_serial.digest_dir(out)
        digests["one_gpu"] = dg
        rec["one_gpu"] = {"count_lines": list(st.counts()), "vertices": int(st.n_pos), "outputs_sha256": dg, "outputs_bytes": nbytes,
                          "path_nodes": int(ts.n_path_nodes), "path_checksum": f"{ts.path_checksum:016x}", "s_total": time.perf_counter() - t1}
        host.pagh_release(C.c_void_p(g))
        hip.pag_destroy(C.c_void_p(g))
        shutil.rmtree(out, ignore_errors=True)
        torch.cuda.empty_cache()
        print("one GPU:", rec["one_gpu"], flush=True)
        save()

    for n in [int(x) for x in args.ranks.split(",") if x]:
        out = tempfile.mkdtemp(prefix=f"pagc3_n{n}_", dir=shm)
        res = rank_serial.run(hip, host, make_handle, inp, n_ranks=n, eps=spec.eps, k=spec.k, threads=spec.threads, ctgs=ctg_len, ctg_alns=alns,
                              ref_lens=[len(ref_np)], ctg_seqs=ctg_seqs, ref_seqs=ref_seqs, orient=orient, out_dir=out, device=dev, halo=args.halo,
                              log=lambda *a: print(f"[N={n}]", *a, flush=True))
        if args.reference_digests:
            import c2_text_runs
            want = json.load(open(args.reference_digests))["compare"]["reference_sha256"]
            got = c2_text_runs._digests(out)
            want = {f: h for f, h in want.items() if f != "contig.txt"}  # (written by the executable's driver, not by the block's host half)
            diff = sorted(f for f in set(want) | set(got) if want.get(f) != got.get(f))
            res["reference_digests"] = {"file": os.path.basename(args.reference_digests), "files": len(want), "differing_files": diff, "identical": not diff}
            assert not diff, f"N={n}: output files differ from the compiled reference's: {diff[:8]}"
        shutil.rmtree(out, ignore_errors=True)
        held = [r["held_fraction"] for r in res["ranks"]]
        res["properties"] = {"count_lines_equal_sums_over_owners": True, "max_held_fraction": max(held), "held_bound": 1.0 / n + 0.15,
                             "no_walk_left_its_region": True}
        assert max(held) < 1.0 / n + 0.15, held
        rec["runs"].append(res)
        digests[f"n{n}"] = res["outputs_sha256"]
        print(f"[N={n}] outputs {res['outputs_sha256'][:16]} ({res['outputs_bytes'] / 1e9:.2f} GB), count lines {res['count_lines_sum_over_owners']}, "
              f"{res['s_total']:.1f} s", flush=True)
        save()
    rec["outputs_identical_for_all_runs"] = len(set(digests.values())) == 1
    rec["digests"] = digests
    if "one_gpu" in rec:
        for r in rec["runs"]:
            assert r["count_lines_sum_over_owners"] == rec["one_gpu"]["count_lines"], "count lines differ from the one-GPU run"
    save()
    assert rec["outputs_identical_for_all_runs"], digests
    print("ok:", json.dumps({k: v[:16] for k, v in digests.items()}))


if __name__ == "__main__":
    main()
