"""BASELINE configs[3] (a 3 Gb human-like reference as 24 sequences <= 250 Mb, 30x synthetic PacBio CLR, ~5 k contigs, 8 GPUs)
executed on ONE MI355X, block after block: the pipeline runs one pagraph block per reference sequence (AlignGraph2.py:399-431,
pagraph.cpp:181-263), so the genome is 24 independent config blocks (SURVEY.md §8e level 1); a block one GPU holds goes through
pag_process + pagh_traverse on one handle, a larger one through the rank-serial driver of the sharded build (level 2,
aligngraph2_amd/rank_serial.py: the N ranks one after the other, what a rank takes in recomputed on the device).

Writes a JSON record: per block its geometry, how it ran, seconds, vertices, count lines, a digest over its output files; the
longest-first deal of the 24 blocks over 8 ranks (parallel.assign_blocks: what `run_config_blocks` does with a 24-block config.txt)
with the per-rank sums of the measured SINGLE-GPU block times.  No multi-GPU timing exists and none is implied: the sums say
how evenly the deal spreads the work, not how long 8 GPUs take.

Per block the solid k-mer set is kmer_counter's rule on THAT block's reads (the pipeline counts once over all reads; a
genome-wide table would be the same 4^14 counters filled by 24 x the reads — the per-block sets keep every block's generator
self-contained).  Checked on the way: count lines of a sharded block = sums over the owners for every rank, held fractions,
no walk leaves its region (pag_travel would fail); with --cross-check N the first N blocks that fit one GPU ALSO run as 4
ranks and must write the same files.

    python tests/c4_blocks.py OUT.json                 # the whole genome (about 20 GPU-minutes)
    python tests/c4_blocks.py OUT.json --scale 0.1     # every length / read count x 0.1 (a smoke run)"""
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

# GRCh38 primary assembly, Mb (1-22, X, Y): every sequence <= 250 Mb as the u32 coordinate space demands
HUMAN_MB = [248, 242, 198, 190, 182, 171, 159, 145, 138, 134, 135, 133, 114, 107, 102, 90, 83, 80, 59, 64, 47, 51, 156, 57]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scale", type=float, default=1.0, help="lengths and read counts times this")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--read-span", type=int, default=10_000)
    ap.add_argument("--ctg-len", type=int, default=600_000, help="mean contig length (3.1 Gb / 600 kb = about 5 k contigs)")
    ap.add_argument("--one-gpu-bases", type=float, default=3.0e9, help="blocks with at most this many read bases run on one handle")
    ap.add_argument("--ranks", type=int, default=4, help="ranks of the rank-serial run of a larger block")
    ap.add_argument("--cross-check", type=int, default=1, help="this many one-handle blocks also run as --ranks ranks (same files)")
    ap.add_argument("--blocks", default="", help="only these block numbers (comma separated)")
    ap.add_argument("--seed", type=int, default=40)
    ap.add_argument("--resume", action="store_true", help="keep the blocks OUT.json already holds, run the others")
    args = ap.parse_args()

    import numpy as np
    import torch
    import bench
    import biggen
    import pagctl
    from aligngraph2_amd import parallel, rank_serial

    hip, host = bench.load_libs()
    dev = "cuda:0"
    lens = [max(200_000, int(mb * 1e6 * args.scale)) for mb in HUMAN_MB]
    only = [int(x) for x in args.blocks.split(",") if x] or list(range(len(lens)))
    rec = {"what": "BASELINE configs[3]: 24 reference sequences of a 3 Gb human-like genome at 30x, block after block on ONE MI355X (one handle when the "
                   "block fits, the rank-serial driver of the sharded build otherwise); no multi-GPU timing exists",
           "geometry": {"sequences": len(lens), "reference_bases": int(sum(lens)), "coverage": args.coverage, "read_span": args.read_span, "k": 14, "epsilon": 10,
                        "mean_contig_len": args.ctg_len, "scale": args.scale},
           "blocks": []}
    free, total = torch.cuda.mem_get_info(dev)
    rec["device_total_bytes"] = int(total)
    if args.resume and os.path.exists(args.out):
        old = json.load(open(args.out))
        rec["blocks"] = [blk for blk in old.get("blocks", []) if "outputs_sha256" in blk]
        only = [b for b in only if b not in {blk["block"] for blk in rec["blocks"]}]
        print("resuming: kept blocks", sorted(blk["block"] for blk in rec["blocks"]), "to run", only, flush=True)

    def save():
        with open(args.out, "w") as f:
            json.dump(rec, f, indent=1)

    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    crossed = 0
    t_all = time.perf_counter()
    for b in only:
        L = lens[b]
        n_reads = max(1000, int(L * args.coverage / args.read_span))
        t0 = time.perf_counter()
        spec = biggen.BigSpec(seed=args.seed + b, ref_len=L, n_reads=n_reads, read_span=args.read_span, k=14, eps=10, cov=2, threads=16, ctg_len=args.ctg_len)
        w = biggen.BigWorkload(spec, device=dev)
        torch.cuda.synchronize()
        s_gen = time.perf_counter() - t0
        inp = w.build_input()
        ref_np = w.ref.cpu().numpy()
        ctg_seqs, keep1 = bench.host_seqs(w.contig_codes())
        ref_seqs, keep2 = bench.host_seqs([ref_np])
        orient = [0 if r else 1 for _, _, r in w.ctgs]
        ctg_len = [e - s for s, e, _ in w.ctgs]
        g2r = w.g2r.cpu().numpy()
        alns = [(c, 0, int(g2r[s]), int(g2r[e - 1]) + 1) for c, (s, e, _) in enumerate(w.ctgs)]
        del g2r
        info = {"block": b, "ref_len": L, "reads": n_reads, "read_bases": int(w.n_bases), "contigs": len(w.ctgs), "solid_kmers": int(w.n_solid),
                "min_abundance": int(w.min_abundance), "s_generate": s_gen}

        def make_handle():
            err = C.c_int()
            g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, spec.k, 1, 0, C.byref(err))
            if not g:
                raise RuntimeError(f"pag_create_from_bitmap failed ({err.value}): {hip.pag_last_error().decode()}")
            return g

        def one_handle():
            out = tempfile.mkdtemp(prefix=f"pagc4_b{b}_one_", dir=shm)
            g = make_handle()
            st = pagctl.BuildStats()
            t1 = time.perf_counter()
            if hip.pag_process(C.c_void_p(g), C.byref(inp), C.byref(st)) != 0:
                msg = hip.pag_last_error().decode()
                hip.pag_destroy(C.c_void_p(g))
                raise MemoryError("pag_process: " + msg)
            ts = bench.TraverseStats()
            o_arr = np.array(orient, dtype=np.int32)
            rc = host.pagh_traverse(g, spec.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, o_arr.ctypes.data, spec.threads, spec.eps, 50, out.encode(), b"0_", 0,
                                    C.byref(ts))
            if rc != 0:
                msg = host.pagh_last_error().decode()
                host.pagh_release(C.c_void_p(g))
                hip.pag_destroy(C.c_void_p(g))
                shutil.rmtree(out, ignore_errors=True)
                torch.cuda.empty_cache()
                raise MemoryError("pagh_traverse: " + msg)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            peak = int(total - torch.cuda.mem_get_info(dev)[0])
            dg, nbytes = rank_serial.digest_dir(out)
            host.pagh_release(C.c_void_p(g))
            hip.pag_destroy(C.c_void_p(g))
            shutil.rmtree(out, ignore_errors=True)
            torch.cuda.empty_cache()
            return {"mode": "one handle (pag_process + pagh_traverse)", "s_block": dt, "count_lines": list(st.counts()), "vertices": int(st.n_pos),
                    "outputs_sha256": dg, "outputs_bytes": nbytes, "path_nodes": int(ts.n_path_nodes), "device_bytes_in_use_after": peak}

        def as_ranks(n):
            out = tempfile.mkdtemp(prefix=f"pagc4_b{b}_n{n}_", dir=shm)
            res = rank_serial.run(hip, host, make_handle, inp, n_ranks=n, eps=spec.eps, k=spec.k, threads=spec.threads, ctgs=ctg_len, ctg_alns=alns,
                                  ref_lens=[len(ref_np)], ctg_seqs=ctg_seqs, ref_seqs=ref_seqs, orient=orient, out_dir=out, device=dev,
                                  log=lambda *a: print(f"[block {b} N={n}]", *a, flush=True))
            shutil.rmtree(out, ignore_errors=True)
            held = [r["held_fraction"] for r in res["ranks"]]
            assert len(ctg_len) < 16 * n or max(held) < 1.0 / n + 0.15, held  # (a dozen contigs do not deal out evenly)
            return {"mode": f"{n} ranks one after the other (rank_serial)", "s_block": res["s_total"], "count_lines": res["count_lines_sum_over_owners"],
                    "vertices": res["vertices_total"], "outputs_sha256": res["outputs_sha256"], "outputs_bytes": res["outputs_bytes"], "path_nodes": res["path_nodes"],
                    "max_held_fraction": max(held), "device_bytes_peak_of_the_serial_run": res["device_bytes_peak_of_the_serial_run"],
                    "per_rank_traversal_peak_bytes": [r["bytes_traversal_peak"] for r in res["ranks"]],
                    "per_rank_owner_build_bytes": [r.get("bytes_owner_build", 0) for r in res["ranks"]], "host_bytes_growth": res["host_bytes_growth"]}

        fits = w.n_bases <= args.one_gpu_bases
        if fits:
            try:
                info.update(one_handle())
            except MemoryError as e:  # (the block does not fit one handle after all: it runs as ranks, the attempt is on record)
                info["one_handle_attempt"] = str(e)[:300]
                print(f"block {b}: one handle failed: {str(e)[:300]}; {(total - torch.cuda.mem_get_info(dev)[0]) / 1e9:.1f} GB in use after releasing it", flush=True)
                fits = False
        if fits:
            if crossed < args.cross_check:
                x = as_ranks(args.ranks)
                info["cross_check"] = {"mode": x["mode"], "s_block": x["s_block"], "outputs_sha256": x["outputs_sha256"], "max_held_fraction": x["max_held_fraction"],
                                       "identical": x["outputs_sha256"] == info["outputs_sha256"] and x["count_lines"] == info["count_lines"]}
                assert info["cross_check"]["identical"], f"block {b}: the {args.ranks}-rank run differs from the one-handle run"
                crossed += 1
        else:
            info.update(as_ranks(args.ranks))
        rec["blocks"].append(info)
        print(f"block {b}: {L / 1e6:.0f} Mb, {w.n_bases / 1e9:.2f} Gbases, {len(w.ctgs)} contigs: {info['mode']}, {info['s_block']:.1f} s, {info['vertices']} vertices, "
              f"outputs {info['outputs_sha256'][:12]} ({info['outputs_bytes'] / 1e9:.2f} GB)", flush=True)
        save()
        del w, inp, ref_np, ctg_seqs, ref_seqs, keep1, keep2
        torch.cuda.empty_cache()

    rec["blocks"].sort(key=lambda blk: blk["block"])
    # level 1 of SURVEY §8e: the blocks dealt over 8 ranks, longest first (parallel.assign_blocks — the deal run_config_blocks makes)
    done = rec["blocks"]
    deal = parallel.assign_blocks([blk["read_bases"] for blk in done], 8)
    rec["deal_over_8_ranks"] = [{"rank": r, "blocks": [done[i]["block"] for i in d], "read_bases": int(sum(done[i]["read_bases"] for i in d)),
                                 "sum_of_single_gpu_block_seconds": sum(done[i]["s_block"] for i in d)} for r, d in enumerate(deal)]
    rec["read_bases_total"] = int(sum(blk["read_bases"] for blk in done))
    rec["contigs_total"] = int(sum(blk["contigs"] for blk in done))
    rec["s_total"] = time.perf_counter() - t_all
    save()
    print("ok:", len(done), "blocks,", rec["read_bases_total"] / 1e9, "Gbases,", rec["contigs_total"], "contigs,", f"{rec['s_total']:.0f} s")


if __name__ == "__main__":
    main()
