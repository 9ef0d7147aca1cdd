#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ by running the COMPILED REFERENCE
(oracle/_ref/*, built from /root/reference by oracle/Makefile) — this container only.

Per case <name>/:
  spec.json            generator parameters (tests/synth.py) + pagraph flags
  inputs.sha256        hash of every generated input file (detects generator drift)
  inputs.tar.gz        the input files themselves (small cases only)
  out/                 the reference pagraph's complete -o directory (contig.txt sorted)
  graph.txt.gz         the reference's complete graph after PositionProcessor::process (graph_dump)
  succ.txt.gz          what the reference's PABruijnGraph::successors returns for EVERY vertex of that graph (graph_dump --succ 1:
                       target, step, checkPosition grade, isEdgeSimilar().first, in the reference's order; "B <block>" lines
                       separate the config blocks)
Function-level tables: func_{kmer,mapper,predicate,edit}.txt(.gz) from oracle/_ref/func_golden.

usage: python tests/golden/make_golden.py        (re-creates everything deterministically)
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")

# name -> (Spec kwargs, threads, epsilon, cov, keep_inputs)
CASES = {
    "join_fwd_t1": (dict(seed=101, ref_len=9000, n_reads=260, read_len=900, k=8,
                         contigs=[(150, 4200, False), (4450, 8850, False)]), 1, 10, 2, True),
    "join_rev_t16": (dict(seed=102, ref_len=9000, n_reads=260, read_len=900, k=8,
                          contigs=[(150, 4200, False), (4450, 8850, True)]), 16, 10, 2, True),
    "three_ctg_multi_t4": (dict(seed=103, ref_len=12000, n_reads=330, read_len=1000, k=9,
                                contigs=[(200, 3600, False), (3800, 7600, True), (7850, 11800, False)],
                                extra_ctg_aln=True, dup_read_aln=True), 4, 10, 1, True),
    "ignore_output_t1": (dict(seed=104, ref_len=10000, n_reads=90, read_len=1000, k=9,
                              contigs=[(300, 3500, False), (6500, 9700, False)]), 1, 10, 2, False),
    "extend_gap_t1": (dict(seed=104, ref_len=10000, n_reads=90, read_len=1000, k=9,
                           contigs=[(300, 4000, False), (5500, 9700, False)]), 1, 10, 2, False),
    "eps5_k7_repeats_t8": (dict(seed=105, ref_len=8000, n_reads=240, read_len=800, k=7, repeats=2, repeat_len=400,
                                contigs=[(100, 3700, True), (3950, 7900, False)]), 8, 5, 0, False),
    "eps20_k11_jitter_t16": (dict(seed=106, ref_len=14000, n_reads=200, read_len=1800, read_len_jitter=0.5, k=11,
                                  solid_min_abundance=2, clip_frac=0.8,
                                  contigs=[(200, 6700, False), (7000, 13800, False)]), 16, 20, 3, False),
    # corners of the successor records: a homopolymer tract (k-mers with > 255 clustered positions, vertices with > 64 candidate
    # pairs) and a stretch without solid k-mers (edges with steps >= 1024: the f64 ratio tests instead of the table)
    "succ_corners_t8": (dict(seed=109, ref_len=20000, n_reads=160, read_len=5000, k=11, solid_min_abundance=2,
                             contigs=[(200, 8600, False), (9750, 19800, False)], homopolymer=(3000, 2400),
                             desert=(13000, 1300)), 8, 5, 2, False),
}

# multi-block cases: name -> (list of Spec kwargs (one per config block), threads, epsilon, cov, keep_inputs).
# Blocks with different contig counts / path lengths / orientations after one another (storage of the previous block is
# reused by the product), and a contig listed with BOTH orientations in the second block.
MULTI_CASES = {
    "two_blocks_both_orient_t16": ([dict(seed=107, ref_len=9000, n_reads=260, read_len=900, k=8,
                                         contigs=[(150, 4200, False), (4450, 8850, True)]),
                                    dict(seed=108, ref_len=12000, n_reads=330, read_len=1000, k=8,
                                         contigs=[(200, 3600, False), (3800, 7600, False), (7850, 11800, True)],
                                         both_orient=(1,))], 16, 10, 2, True),
}


def sha_dir(d):
    out = {}
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            out[f] = hashlib.sha256(fh.read()).hexdigest()
    return out


def run(argv, threads):
    env = dict(os.environ)
    if threads > 1:
        env["LD_PRELOAD"] = os.path.join(REF, "libserial_threads.so")
    r = subprocess.run(argv, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{argv[0]} failed: {r.stderr[-2000:]}")
    return r


def main():
    only = set(sys.argv[1:])
    all_cases = [(n, c, False) for n, c in CASES.items()] + [(n, c, True) for n, c in MULTI_CASES.items()]
    for name, (kw, threads, eps, cov, keep), multi in all_cases:
        if only and name not in only:
            continue
        case = os.path.join(HERE, name)
        shutil.rmtree(case, ignore_errors=True)
        os.makedirs(case)
        with tempfile.TemporaryDirectory() as tmp:
            ind = os.path.join(tmp, "in")
            if multi:
                synth.generate_multi([synth.Spec(**b) for b in kw], ind)
            else:
                synth.generate(synth.Spec(**kw), ind)
            json.dump({("blocks" if multi else "spec"): kw, "threads": threads, "epsilon": eps, "cov": cov},
                      open(os.path.join(case, "spec.json"), "w"), indent=1)
            json.dump(sha_dir(ind), open(os.path.join(case, "inputs.sha256"), "w"), indent=1)
            if keep:
                with tarfile.open(os.path.join(case, "inputs.tar.gz"), "w:gz", compresslevel=9) as tf:
                    for f in sorted(os.listdir(ind)):
                        tf.add(os.path.join(ind, f), arcname=f)
            out = os.path.join(case, "out")
            os.makedirs(out)
            run(synth.pagraph_argv(os.path.join(REF, "pagraph"), ind, out, threads=threads, epsilon=eps, cov=cov), threads)
            ct = os.path.join(out, "contig.txt")
            lines = sorted(open(ct).read().split())
            open(ct, "w").write("".join(x + "\n" for x in lines))
            gd = os.path.join(tmp, "gd")
            os.makedirs(gd)
            run([os.path.join(REF, "graph_dump"), "-t", str(threads), "-k", ind + "/kmer.bin", "-c", ind + "/ctg.fasta",
                 "-R", ind + "/ref.fasta", "-p", ind, "-a", ind + "/aln", "-o", gd, "--epsilon", str(eps), "-v", str(cov),
                 "--succ", "1"], threads)
            dumps = sorted((f for f in os.listdir(gd) if f.endswith(".graph.txt")), key=lambda f: int(f.split(".")[0]))
            with gzip.GzipFile(os.path.join(case, "graph.txt.gz"), "wb", compresslevel=9, mtime=0) as dst:
                for f in dumps:  # (one dump per config block, in block order)
                    dst.write(open(os.path.join(gd, f), "rb").read())
            with gzip.GzipFile(os.path.join(case, "succ.txt.gz"), "wb", compresslevel=9, mtime=0) as dst:
                for f in dumps:
                    dst.write(b"B %d\n" % int(f.split(".")[0]))
                    dst.write(open(os.path.join(gd, f.replace(".graph.txt", ".succ.txt")), "rb").read())
        print(name, sorted(os.listdir(os.path.join(case, "out"))))
    for what in ("kmer", "mapper", "predicate", "edit") if not only else ():
        r = subprocess.run([os.path.join(REF, "func_golden"), what], capture_output=True, check=True)
        path = os.path.join(HERE, f"func_{what}.txt")
        if len(r.stdout) > 200_000:
            with gzip.GzipFile(path + ".gz", "wb", compresslevel=9, mtime=0) as f:
                f.write(r.stdout)
        else:
            open(path, "wb").write(r.stdout)
        print("func", what, len(r.stdout))


if __name__ == "__main__":
    main()
