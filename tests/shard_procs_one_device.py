#!/usr/bin/env python3
"""One-off measurement (GPU box): ONE config block built by N cooperating `bin/pagraph` processes (PAGRAPH_SHARD=r/N) that
share the box's one device and exchange through the rendezvous directory (transport "host": RCCL refuses ranks on one
device) — BASELINE configs[1] as text by default.  Kept per rank: wall time, the driver's [timing] lines, the library's
[shard timing] line (seconds per stage of pag_shard_run, payload out in each bulk exchange), and how much of the ALN text
the rank classified.  The outputs rank 0 writes are held against the per-file SHA-256 of what the compiled reference wrote
for the same workload (profiles/r04_c2_text_parity.json) and against a one-process run.  The processes take turns on one
device, so the stage times are NOT what N GPUs would show: the record is the stage MIX (what an overlap of the exchange with
the extraction could hide at most) and the bytes.

usage: python tests/shard_procs_one_device.py OUT.json [--world 4] [--reads N --ref-len L]"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from c2_text_runs import _digests, _quota  # noqa: E402


def run_ranks(argv, world, extra):
    rdv = tempfile.mkdtemp(prefix="pagshard_", dir="/dev/shm")
    procs, t0 = [], time.time()
    for r in range(world):
        env = dict(os.environ, PAGRAPH_SHARD=f"{r}/{world}", PAGRAPH_SHARD_DIR=rdv, PAGRAPH_SHARD_TRANSPORT="host", PAG_COMM_TIMEOUT_S="600",
                   PAG_DEVICE_SHARERS=str(world), PAGRAPH_TIMING="1", **extra)
        procs.append(subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    ranks = []
    for r, pr in enumerate(procs):
        so, se = pr.communicate(timeout=1500)
        ranks.append({"rank": r, "returncode": pr.returncode, "wall_s": time.time() - t0,
                      "shard_timing": [ln for ln in se.splitlines() if ln.startswith("[shard timing]")],
                      "timing": [ln for ln in se.splitlines() if ln.startswith("[timing] ") and not ln.startswith("[timing]   ")][-30:],
                      "count_lines": [ln.strip() for ln in so.splitlines() if ln.strip().startswith(("merge edge", "total pos", "merge pos"))],
                      "stderr_tail": se[-1500:] if pr.returncode else ""})
    shutil.rmtree(rdv, ignore_errors=True)
    return ranks, time.time() - t0


def stage_mix(ranks):
    """seconds per stage, max over ranks, from the [shard timing] lines (round 5: the tuples travel in chunks beside the next
    chunk's extraction, a selection beside the next selection)"""
    pat = re.compile(r"tuples in (\d+) chunks: extraction \+ owner partition ([\d.]+) s, exchange ([\d.]+) s of which ([\d.]+) s beside the next chunk's "
                     r"extraction \(loop ([\d.]+) s, (\d+) B out\), owner layout ([\d.]+) s, K2-K4 ([\d.]+) s; regions (\w+): selections for \d+ ranks ([\d.]+) s, "
                     r"exchange ([\d.]+) s of which ([\d.]+) s beside the next selection \(loop ([\d.]+) s, (\d+) B out\), owner order ([\d.]+) s, "
                     r"import\+region\+release ([\d.]+) s")
    names = ["extraction + owner partition", "tuple exchange", "tuple exchange beside an extraction", "tuple loop (wall)", "owner layout", "K2-K4", "selections",
             "region exchange", "region exchange beside a selection", "region loop (wall)", "owner order", "import+region+release"]
    rows = []
    for rk in ranks:
        for ln in rk["shard_timing"]:
            m = pat.search(ln)
            if m:
                v = [float(m.group(i)) for i in (2, 3, 4, 5, 7, 8, 10, 11, 12, 13, 15, 16)]
                rows.append({"rank": rk["rank"], "chunks": int(m.group(1)), "regions": m.group(9), "seconds": dict(zip(names, v)),
                             "tuple_bytes_out": int(m.group(6)), "region_bytes_out": int(m.group(14))})
    if not rows:
        return None
    worst = {n: max(r["seconds"][n] for r in rows) for n in names}
    xfer = sum(r["seconds"]["tuple exchange"] + r["seconds"]["region exchange"] for r in rows)
    hidden = sum(r["seconds"]["tuple exchange beside an extraction"] + r["seconds"]["region exchange beside a selection"] for r in rows)
    could = sum(min(r["seconds"]["tuple exchange"], r["seconds"]["extraction + owner partition"]) + min(r["seconds"]["region exchange"], r["seconds"]["selections"]) for r in rows)
    return {"per_rank": rows, "max_over_ranks_s": worst,
            "overlap_fraction_of_the_exchange_time": hidden / xfer if xfer else None,
            "overlap_fraction_of_what_could_overlap": hidden / could if could else None,
            "overlap_note": "seconds of a background exchange that ran while the rank's own stream extracted the next chunk / made the next selection, over all ranks; "
                            "'could overlap' = min(exchange, extraction) + min(exchange, selections) per rank.  Four processes take turns on ONE device and the "
                            "exchange goes through files, so the seconds are not what N GPUs over xGMI would show"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    ap.add_argument("--reference-digests", default=os.path.join(ROOT, "profiles", "r04_c2_text_parity.json"))
    args = ap.parse_args()
    import biggen
    import synth
    sp = biggen.BigSpec(seed=2, ref_len=args.ref_len, n_reads=args.reads, k=14, eps=10, cov=2, threads=16)
    w = biggen.BigWorkload(sp, device="cuda")
    d = "/dev/shm/c2_text"
    shutil.rmtree(d, ignore_errors=True)
    w.write_text(d)
    n_bases = w.n_bases
    del w
    import torch
    torch.cuda.empty_cache()
    ours = os.path.join(ROOT, "aligngraph2_amd", "bin", "pagraph")
    rec = {"workload": f"{args.reads} x 10 kb reads vs {args.ref_len / 1e6:g} Mb reference, k=14, epsilon=10 (seed 2), text inputs in /dev/shm",
           "read_bases": n_bases, "world": args.world, "device": "ONE MI355X shared by all ranks (transport: files of the rendezvous directory)",
           "cgroup_cpu_quota": _quota(), "measured": time.strftime("round 5, %Y-%m-%d"),
           "input_bytes": {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}}
    full = args.reads == 100_000 and args.ref_len == 50_000_000
    want = None
    if full and os.path.exists(args.reference_digests):
        want = json.load(open(args.reference_digests))["compare"]["reference_sha256"]
    digests = {}
    for name, extra in (("one process", None), (f"{args.world} processes", {}), (f"{args.world} processes, whole exchanges (round 4's order of work)", {"PAG_SHARD_CHUNKS": "1", "PAG_SHARD_PIPELINE": "0"})):
        time.sleep(8)
        out = "/dev/shm/shard_procs_out"
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        argv = synth.pagraph_argv(ours, d, out, threads=16, epsilon=10, cov=2)
        if extra is None:
            t0 = time.time()
            r = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, PAGRAPH_TIMING="1"))
            run = {"returncode": r.returncode, "wall_s": time.time() - t0, "stderr_tail": r.stderr[-1500:] if r.returncode else "",
                   "timing": [ln for ln in r.stderr.splitlines() if ln.startswith("[timing] ") and not ln.startswith("[timing]   ")][-30:]}
            ok = r.returncode == 0
        else:
            ranks, wall = run_ranks(argv, args.world, extra)
            ok = all(rk["returncode"] == 0 for rk in ranks)
            run = {"wall_s": wall, "ranks": ranks, "stages": stage_mix(ranks)}
        got = _digests(out) if ok else {}
        digests[name] = got
        run["output_files"] = len(got)
        if want is not None and ok:
            diff = sorted(f for f in set(want) | set(got) if want.get(f) != got.get(f))
            run["against_the_reference_digests"] = {"source": os.path.basename(args.reference_digests), "differing_files": diff, "identical": not diff}
        rec[name] = run
        print(name, "ok" if ok else "FAILED", round(run["wall_s"], 2), "s", run.get("against_the_reference_digests"), flush=True)
        if not ok:
            print(json.dumps(run, indent=1)[-4000:], flush=True)
        shutil.rmtree(out, ignore_errors=True)
    base = digests["one process"]
    rec["all_runs_wrote_the_same_bytes"] = bool(base) and all(v == base for v in digests.values())
    print("all runs wrote the same bytes:", rec["all_runs_wrote_the_same_bytes"], flush=True)
    shutil.rmtree(d, ignore_errors=True)
    json.dump(rec, open(args.out, "w"), indent=1)
    if not rec["all_runs_wrote_the_same_bytes"]:
        sys.exit(1)


if __name__ == "__main__":
    main()
