#!/usr/bin/env python3
"""One-off measurement (GPU box): instruction mix and LDS behaviour of the build / successor-stage kernels from rocprofv3 PMC
passes over `bench.py --steps 1 --warmup 0` (a few SQ counters per pass, --kernel-trace only), per launch.
usage (from /tmp, TMPDIR=/tmp): python <repo>/tests/pmc_kernel_mix.py OUT.json"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"], ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"],
          ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY"], ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT"],
          ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"], ["SQ_INSTS_LDS_ATOMIC", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS"]]


def one_pass(counters, tmp):
    shutil.rmtree(tmp, ignore_errors=True)
    r = subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "c", "--", sys.executable,
                        os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-file-to-file"], capture_output=True, text=True,
                       env=dict(os.environ, PAG_WALK_IDLE_S="5"), timeout=900)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(tmp, "**", "c_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "pagdev" not in row["Kernel_Name"]:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc, r.returncode


def main():
    out = {}
    for p in PASSES:
        acc, rc = one_pass(p, "/tmp/pmc_mix")
        for k, cs in acc.items():
            for c, (v, n) in cs.items():
                out.setdefault(k, {})[c] = v / max(n, 1)
                out[k]["launches"] = n
        print(p, "rc", rc, len(acc), "kernels", flush=True)
    json.dump({"note": "rocprofv3 --pmc <a few SQ counters per pass> --kernel-trace over bench.py --steps 1 --warmup 0 at BASELINE configs[1]; per launch (summed over "
                       "the device's shader engines as rocprofv3 reports them)", "kernels": out}, open(sys.argv[1], "w"), indent=1)
    for k in sorted(out):
        if any(x in k for x in ("extract_kernel<true", "sort_scatter<7", "k_succ<", "k_order_apply", "solid_mask", "cluster_short", "k_walk")):
            print(k, {c: round(v) for c, v in out[k].items()})


if __name__ == "__main__":
    main()
