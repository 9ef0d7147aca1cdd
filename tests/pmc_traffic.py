#!/usr/bin/env python3
"""One-off measurement (GPU box): HBM bytes per launch of the build / successor-stage kernels from rocprofv3 PMC passes over
`bench.py --steps 1 --warmup 0` (FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only; FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950).  Counter collection serialises dispatches, so a pass ends at the resident walker
grid (its watchdog fails the run after PAG_WALK_IDLE_S seconds): the kernels before it are what the record holds.

usage (from /tmp, TMPDIR=/tmp): python <repo>/tests/pmc_traffic.py OUT.json"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, tmp):
    shutil.rmtree(tmp, ignore_errors=True)
    env = dict(os.environ, PAG_WALK_IDLE_S="5")
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "c", "--",
                    sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-file-to-file"],
                   capture_output=True, text=True, env=env, timeout=900)
    acc = collections.defaultdict(lambda: [0.0, 0])
    dur = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(tmp, "**", "c_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or "pagdev" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    for f in glob.glob(os.path.join(tmp, "**", "c_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "pagdev" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            dur[k][1] += 1
    return acc, dur


def main():
    out = sys.argv[1]
    fetch, dur = one_pass("FETCH_SIZE", "/tmp/pmc_f")
    write, _ = one_pass("WRITE_SIZE", "/tmp/pmc_w")
    rec = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over bench.py --steps 1 --warmup 0 at "
                   "BASELINE configs[1] (tests/pmc_traffic.py). FETCH_SIZE doubled (gfx950: a wide coalesced read is reported at half its bytes, "
                   "MI355X_MICROARCH.md; not calibrated for 4-byte gathers: the gather-heavy kernels are upper-bounded by the doubled figure), "
                   "WRITE_SIZE as is. The passes stop at the walker (counter collection serialises dispatches; the resident grid starves the kernels "
                   "it waits for), so the traversal kernels behind the successor stage are not in them.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        nf, nw = fetch[k][1] or 1, write[k][1] or 1
        f_kib, w_kib = fetch[k][0] / nf, write[k][0] / nw
        rec["kernels"][k] = {"launches": fetch[k][1] or write[k][1], "FETCH_SIZE_KiB_raw_per_launch": f_kib, "WRITE_SIZE_KiB_per_launch": w_kib,
                             "hbm_GB_per_launch_fetch_x2_plus_write": (2 * f_kib + w_kib) * 1024 / 1e9,
                             "ms_per_launch_under_pmc": dur[k][0] / dur[k][1] if dur[k][1] else None}
    json.dump(rec, open(out, "w"), indent=1)
    for k, v in rec["kernels"].items():
        if v["hbm_GB_per_launch_fetch_x2_plus_write"] > 1:
            print(f"{k:45s} {v['launches']:3d} x  {v['hbm_GB_per_launch_fetch_x2_plus_write']:8.2f} GB  {v['ms_per_launch_under_pmc'] or 0:8.2f} ms")


if __name__ == "__main__":
    main()
