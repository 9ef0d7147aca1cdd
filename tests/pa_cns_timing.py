"""GPU box: the consensus step (bin/pa_cns: device / flat / host backends) against the compiled reference `oracle/_ref/pa_cns -t 16` on
ONE backbone at the pipeline's settings (part 5 000, top 3 000, alpha 250) and ~190x coverage — wall clock of the whole program
(parsing, slicing, graphs, FASTA), outputs compared byte for byte.

    python tests/pa_cns_timing.py OUT.json [--backbone 1000000] [--threads 16]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--backbone", type=int, default=1_000_000)
    ap.add_argument("--coverage", type=float, default=190.0)
    ap.add_argument("--threads", type=int, default=16)
    args = ap.parse_args()
    import cns_cases
    case = dict(seed=11, backbone=args.backbone, n_reads=int(args.backbone * args.coverage / 1500), read_len=1500, part=5000, top_k=3000, alpha=250)
    d = tempfile.mkdtemp(prefix="pacns_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    t0 = time.perf_counter()
    cns_cases.write_case(case, d)
    rec = {"what": "bin/pa_cns (three backends) and the compiled reference pa_cns on one backbone at the pipeline's settings; whole-program wall clock",
           "case": case, "parts": (args.backbone + 4999) // 5000, "threads": args.threads, "s_generate": time.perf_counter() - t0,
           "aln_bytes": os.path.getsize(os.path.join(d, "reads.ref")), "cpus": len(os.sched_getaffinity(0)), "runs": []}
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "pa_cns")
    ours = os.path.join(ROOT, "aligngraph2_amd", "bin", "pa_cns")
    outs = {}
    for label, exe, env in [("reference -t %d" % args.threads, ref_exe, {}), ("hip", ours, {"PA_CNS_BACKEND": "hip"}), ("flat", ours, {"PA_CNS_BACKEND": "flat"}),
                            ("host", ours, {"PA_CNS_BACKEND": "host"}), ("default", ours, {})]:
        if not os.path.exists(exe):
            rec["runs"].append({"label": label, "skipped": exe + " missing"})
            continue
        out = os.path.join(d, "out_" + label.split()[0] + ".fasta")
        e = dict(os.environ)
        e.pop("PA_CNS_BACKEND", None)
        e.update(env)
        best = None
        for rep in range(2):
            t1 = time.perf_counter()
            r = subprocess.run(cns_cases.argv(exe, d, out, case, threads=args.threads), capture_output=True, text=True, env=e, timeout=3000)
            dt = time.perf_counter() - t1
            assert r.returncode == 0, label + ": " + r.stderr[-1500:]
            best = dt if best is None else min(best, dt)
        outs[label] = open(out, "rb").read()
        rec["runs"].append({"label": label, "s_wall_best_of_2": best, "stdout_tail": r.stdout[-200:]})
        print(f"{label}: {best:.2f} s", flush=True)
    ref = next((v for k, v in outs.items() if k.startswith("reference")), None)
    rec["identical_to_reference"] = {k: (v == ref) for k, v in outs.items()} if ref is not None else None
    with open(args.out, "w") as f:
        json.dump(rec, f, indent=1)
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(rec["identical_to_reference"]))


if __name__ == "__main__":
    main()
