#!/usr/bin/env python3
"""One-off measurement (GPU box): BASELINE configs[1] as TEXT files, through (a) the compiled reference `pagraph -t 64`
(oracle/_ref, natural threads: a timing run, its outputs are not compared) and (b) the drop-in bin/pagraph — wall clock of
both whole programs, file parsing included.  Writes a JSON record that bench.py quotes as the like-for-like CPU baseline
(`cpu_baseline.full_workload`) and as `file_to_file_bases_per_s`; the record is committed under profiles/.

--compare: the reference runs with `-t 16` under the thread-serialising shim (oracle/_ref/libserial_threads.so: the
deterministic order the product reproduces) instead, its output directory is kept, and EVERY output file of the
drop-in executable is compared with it byte for byte (contig.txt as a set, quirk Q11): the whole product — parsers,
pag_prepare, build, device walkers, writers — pinned once at full BASELINE configs[1] size.

usage: python tests/c2_text_runs.py OUT.json [--reads N --ref-len L] [--compare]"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def _quota():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        return None


def _digests(out_dir):
    """SHA-256 per output file (contig.txt as a sorted set: the reference writes it in hash order)"""
    import hashlib
    dig = {}
    for f in sorted(os.listdir(out_dir)):
        data = open(os.path.join(out_dir, f), "rb").read()
        if f == "contig.txt":
            data = b"\n".join(sorted(data.split()))
        dig[f] = hashlib.sha256(data).hexdigest()
    return dig


def multi_block(args, rec, d, ours, n_bases):
    """one run of several config blocks through the drop-in, in the driver's modes; every mode must write the same bytes"""
    import hashlib
    import synth
    cfg = os.path.join(d, "config.txt")
    text = open(cfg).read().rstrip("\n") + "\n\n"
    open(cfg, "w").write(text * args.blocks)
    rec["blocks"] = args.blocks
    rec["modes"] = {}
    digests = {}
    modes = [("block after block", {"PAGRAPH_PREFETCH": "0", "PAGRAPH_OVERLAP": "0"}),
             ("next block parsed ahead", {"PAGRAPH_PREFETCH": "1", "PAGRAPH_OVERLAP": "0"}),
             ("parsed ahead + host half beside the next block", {"PAGRAPH_PREFETCH": "1", "PAGRAPH_OVERLAP": "1"}),
             ("writing the packed sidecars (.pagaln)", {"PAGRAPH_PREFETCH": "1", "PAGRAPH_OVERLAP": "1", "PAGRAPH_ALN_SIDECAR": "1"}),
             ("from the packed sidecars", {"PAGRAPH_PREFETCH": "1", "PAGRAPH_OVERLAP": "1"}),
             ("from the packed sidecars, block after block", {"PAGRAPH_PREFETCH": "0", "PAGRAPH_OVERLAP": "0"})]
    for name, env in modes:
        time.sleep(8)  # (a process started right after another one gave back ~100 GB of device memory waits 3-4 s for its first allocation)
        out = "/dev/shm/c2_out_multi"
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        t0 = time.time()
        r = subprocess.run(synth.pagraph_argv(ours, d, out, threads=16, epsilon=10, cov=2), capture_output=True, text=True,
                           env=dict(os.environ, PAGRAPH_TIMING="1", **env))
        dt = time.time() - t0
        h = hashlib.sha256()
        for f in sorted(os.listdir(out)):
            data = open(os.path.join(out, f), "rb").read()
            if f == "contig.txt":
                data = b"\n".join(sorted(data.split()))
            h.update(f.encode() + b"\0" + hashlib.sha256(data).digest())
        digests[name] = h.hexdigest()
        rec["modes"][name] = {"env": env, "returncode": r.returncode, "wall_s": dt, "s_per_block": dt / args.blocks, "bases_per_s": n_bases * args.blocks / dt,
                              "output_files": len(os.listdir(out)), "outputs_sha256": digests[name],
                              "timing_lines": [ln for ln in r.stderr.splitlines() if ln.startswith("[timing] ") and not ln.startswith("[timing]   ") and
                                               any(w in ln for w in ("load block", "successor records ", "wait for", "walks", "host half", "traversal", "traverse + write", "graph build", "prepare (", "load global"))][-40:],
                              "stderr_tail": r.stderr[-600:] if r.returncode else ""}
        print(name, round(dt, 2), "s", round(dt / args.blocks, 2), "s/block", digests[name][:12], "rc", r.returncode, flush=True)
        shutil.rmtree(out, ignore_errors=True)
    rec["all_modes_wrote_the_same_bytes"] = len(set(digests.values())) == 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=100_000)
    ap.add_argument("--ref-len", type=int, default=50_000_000)
    ap.add_argument("--ref-threads", type=int, default=64)
    ap.add_argument("--skip-reference", action="store_true", help="only time the drop-in executable")
    ap.add_argument("--compare", action="store_true", help="reference -t 16 under the serialising shim; byte-compare all output files")
    ap.add_argument("--compare-with", default=None, help="a JSON written by an earlier --compare run of the same workload: the drop-in's output files are "
                    "held against the reference's per-file SHA-256 kept there (no reference run)")
    ap.add_argument("--keep-text", action="store_true", help="leave the text files in /dev/shm/c2_text (for runs by hand)")
    ap.add_argument("--blocks", type=int, default=1, help="> 1: config.txt written that many times over (every block reads the same files); "
                    "the drop-in alone, block after block / next block parsed ahead / host half beside the next block / packed sidecars")
    args = ap.parse_args()
    import biggen
    import synth
    sp = biggen.BigSpec(seed=2, ref_len=args.ref_len, n_reads=args.reads, k=14, eps=10, cov=2, threads=16)
    t0 = time.time()
    w = biggen.BigWorkload(sp, device="cuda")
    d = "/dev/shm/c2_text"
    shutil.rmtree(d, ignore_errors=True)
    w.write_text(d)
    n_bases = w.n_bases
    del w
    rec = {"workload": f"{args.reads} x 10 kb reads vs {args.ref_len / 1e6:g} Mb reference, k=14, epsilon=10 (BASELINE configs[1], seed 2), text inputs in /dev/shm",
           "read_bases": n_bases, "host_cores": os.cpu_count(), "cgroup_cpu_quota": _quota(), "measured": time.strftime("round 6, %Y-%m-%d"),
           "generate_and_write_s": time.time() - t0,
           "input_bytes": {f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))}}
    print(rec, flush=True)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "pagraph")
    ours = os.path.join(ROOT, "aligngraph2_amd", "bin", "pagraph")
    if args.blocks > 1:
        multi_block(args, rec, d, ours, n_bases)
        shutil.rmtree(d, ignore_errors=True)
        json.dump(rec, open(args.out, "w"), indent=1)
        return
    ref_env, ref_threads = {}, args.ref_threads
    if args.compare:
        ref_env, ref_threads = {"LD_PRELOAD": os.path.join(ROOT, "oracle", "_ref", "libserial_threads.so")}, 16
    for name, exe, threads, env in (("ours", ours, 16, {"PAGRAPH_TIMING": "1"}), ("reference", ref_bin, ref_threads, ref_env)):
        if name == "reference" and (args.skip_reference or args.compare_with):
            continue
        out = f"/dev/shm/c2_out_{name}"
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        t0 = time.time()
        r = subprocess.run(synth.pagraph_argv(exe, d, out, threads=threads, epsilon=10, cov=2), capture_output=True, text=True,
                           env=dict(os.environ, **env))
        dt = time.time() - t0
        rec[name] = {"program": os.path.relpath(exe, ROOT), "threads_flag": threads, "returncode": r.returncode, "wall_s": dt,
                     "bases_per_s": n_bases / dt, "output_files": len(os.listdir(out)),
                     "count_lines": [ln.strip() for ln in r.stdout.splitlines() if ln.strip().startswith(("merge edge", "total pos", "merge pos"))],
                     "stderr_tail": r.stderr[-3000:] if name == "ours" else r.stderr[-300:]}
        print(name, rec[name]["wall_s"], rec[name]["bases_per_s"], flush=True)
        if name == "ours" and args.compare_with:
            want = json.load(open(args.compare_with))["compare"]["reference_sha256"]
            got = _digests(out)
            diff = sorted(f for f in set(want) | set(got) if want.get(f) != got.get(f))
            rec["compare"] = {"reference": "per-file SHA-256 of the reference's outputs kept in " + os.path.basename(args.compare_with), "files_reference": len(want),
                              "files_ours": len(got), "differing_files": diff, "identical": not diff}
            print("compare:", rec["compare"], flush=True)
        if not args.compare:
            shutil.rmtree(out, ignore_errors=True)
    if args.compare and "reference" in rec and rec["reference"]["returncode"] == 0 and rec["ours"]["returncode"] == 0:
        a, b = "/dev/shm/c2_out_reference", "/dev/shm/c2_out_ours"
        fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
        diff, total_bytes = [], 0
        for f in sorted(set(fa) | set(fb)):
            if f not in fa or f not in fb:
                diff.append(f + " (missing on one side)")
                continue
            x, y = open(os.path.join(a, f), "rb").read(), open(os.path.join(b, f), "rb").read()
            total_bytes += len(x)
            if f == "contig.txt":
                x, y = sorted(x.split()), sorted(y.split())
            if x != y:
                diff.append(f)
        rec["compare"] = {"reference_sha256": _digests(a), "reference": "oracle/_ref/pagraph -t 16 under libserial_threads.so", "files_reference": len(fa), "files_ours": len(fb),
                          "bytes_compared": total_bytes, "differing_files": diff, "identical": not diff and fa == fb,
                          "count_lines_equal": rec["reference"]["count_lines"] == rec["ours"]["count_lines"]}
        print("compare:", rec["compare"], flush=True)
        shutil.rmtree(a, ignore_errors=True)
        shutil.rmtree(b, ignore_errors=True)
    if not args.keep_text:
        shutil.rmtree(d, ignore_errors=True)
    json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
