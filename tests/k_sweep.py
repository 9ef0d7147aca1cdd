#!/usr/bin/env python3
"""One-off measurement (GPU box), BASELINE configs[4]'s second half: the k-mer sort's digit width ("LDS-bucket sizing") against
HBM GB/s from rocprofv3 PMC counters, for k in {12, 14, 16} at the configs[1] read set (the key has 2k bits: three 8-bit,
four 7-bit, four 8-bit passes; a pass ranks its tile into 2^bits LDS buckets).  Per k: the bench's own HIP-event timings
(per scatter launch, whole sort stage) from a build-only run, and FETCH_SIZE / WRITE_SIZE per launch of sort_scatter /
sort_hist from two more build-only runs under `rocprofv3 --pmc` (separate passes, --kernel-trace only; FETCH_SIZE doubled,
the gfx950 correction of MI355X_MICROARCH.md).

usage (from /tmp, TMPDIR=/tmp): python <repo>/tests/k_sweep.py OUT.json [--reads N --ref-len L]"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench_argv(k, args):
    return [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--build-only", "--k", str(k),
            "--reads", str(args.reads), "--ref-len", str(args.ref_len)] + (["--solid-min-abundance", "2"] if k > 14 else [])


def pmc_pass(counter, k, args, tmp):
    shutil.rmtree(tmp, ignore_errors=True)
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "c", "--"] + bench_argv(k, args),
                   capture_output=True, text=True, timeout=1200)
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(tmp, "**", "c_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "pagdev::sort_" in r["Kernel_Name"]:
                kn = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[kn].append(float(r["Counter_Value"]))
    # (a kernel name covers the k-mer sort's launches and the few-thousand-record sorts of pag_prepare: the average is over
    # the launches of the k-mer sort, i.e. those within a factor of two of the largest)
    out = {}
    for kn, vals in acc.items():
        big = [v for v in vals if v >= 0.5 * max(vals)]
        out[kn] = (sum(big) / max(1, len(big)), len(big))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    ap.add_argument("--ks", type=lambda v: [int(x) for x in v.split(",")], default=[12, 14, 16])
    args = ap.parse_args()
    rec = {"note": __doc__.split("usage")[0].strip(), "points": []}
    for k in args.ks:
        r = subprocess.run(bench_argv(k, args), capture_output=True, text=True, timeout=1200)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            rec["points"].append({"k": k, "error": r.stderr[-500:]})
            continue
        d = json.loads(lines[-1])
        pp = d["roofline"]["per_pass"]
        passes = (2 * k + 7) // 8
        bits = (2 * k + passes - 1) // passes
        pt = {"k": k, "key_bits": 2 * k, "passes_per_stream": passes, "digit_bits": bits, "lds_buckets_per_tile": 1 << bits,
              "records_per_launch": pp["records_per_launch"], "ms_per_scatter_launch": pp["ms_per_launch"],
              "scatter_GBps_algorithmic": pp["achieved"], "ms_sort_stage": d["roofline"]["ms_sort"], "whole_sort_GBps_algorithmic": d["roofline"]["achieved"],
              "whole_sort_frac": d["roofline"]["frac"], "position_tuples": d["config"]["position_tuples"], "edge_tuples": d["config"]["edge_tuples"]}
        f = pmc_pass("FETCH_SIZE", k, args, "/tmp/ks_f")
        w = pmc_pass("WRITE_SIZE", k, args, "/tmp/ks_w")
        for kn in sorted(set(f) | set(w)):
            fk, wk = f.get(kn, (0.0, 0))[0], w.get(kn, (0.0, 0))[0]
            hbm = (2 * fk + wk) * 1024
            ent = {"launches_seen": f.get(kn, (0, 0))[1], "FETCH_SIZE_KiB_raw_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk, "hbm_bytes_per_launch": hbm}
            if "sort_scatter" in kn and pp["ms_per_launch"] > 0:
                # (launches of the traversal's own sorts are absent: build-only runs)
                ent["hbm_GBps_at_the_bench_launch_time"] = hbm / (pp["ms_per_launch"] * 1e-3) / 1e9
                ent["hbm_over_algorithmic"] = hbm / (24.0 * pp["records_per_launch"]) if pp["records_per_launch"] else None
            pt.setdefault("pmc", {})[kn] = ent
        rec["points"].append(pt)
        print(json.dumps(pt)[:600], flush=True)
    json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
