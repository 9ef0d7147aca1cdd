"""Seeded inputs for the pre_process tests: a reads FASTQ, a contig FASTA, and the three 3-line
alignment files in the 10-field header form pre_process parses (AlignmentHelper.cpp:11-48)."""
import os

import numpy as np

CASES = {
    "three_refs_k1": dict(seed=1, n_ref=3, n_ctg=9, n_reads=120, top_k=1, min_ratio=0.15),
    "four_refs_k2_quirks": dict(seed=2, n_ref=4, n_ctg=12, n_reads=200, top_k=2, min_ratio=0.2, quirks=True),
    "two_refs_one_each": dict(seed=3, n_ref=2, n_ctg=2, n_reads=40, top_k=1, min_ratio=0.15),
}
# a malformed contig->reference header books an EMPTY contig name; a malformed read alignment then reaches
# std::stoll("") and the reference dies of the uncaught exception — so does the drop-in
CRASH_CASE = dict(seed=2, n_ref=4, n_ctg=12, n_reads=200, top_k=2, min_ratio=0.2, quirks=True, broken_ctg_header=True)


def _seq(rs, n):
    return np.frombuffer(b"ACGT", np.uint8)[rs.integers(0, 4, n)].tobytes().decode()


def write_case(case, d):
    os.makedirs(d, exist_ok=True)
    rs = np.random.default_rng(case["seed"])
    refs = [(f"ref{i + 1}", int(rs.integers(3000, 6000))) for i in range(case["n_ref"])]
    ctgs = [(f"ctg{i}", int(rs.integers(400, 1500))) for i in range(case["n_ctg"])]
    quirks = case.get("quirks", False)
    with open(os.path.join(d, "ctg.fasta"), "w") as f:
        for i, (name, L) in enumerate(ctgs):
            header = f">{name}" + (" some description" if quirks and i == 1 else "")
            s = _seq(rs, L)
            f.write(header + "\n" + "\n".join(s[j:j + 60] for j in range(0, L, 60)) + "\n")

    def aln_record(q, qsize, t, tsize, fwd, qb, qe, tb, te):
        n = max(1, qe - qb)
        row = _seq(rs, n)
        return f"{q} {t} {'F' if fwd else 'R'} {int(rs.integers(100, 9000))} {qb} {qe} {qsize} {tb} {te} {tsize}\n{row}\n{row}\n"

    with open(os.path.join(d, "ctg_to_ref.ref"), "w") as f:
        for i, (name, L) in enumerate(ctgs):
            for _ in range(int(rs.integers(1, 4))):
                t, tsize = refs[int(rs.integers(0, len(refs)))] if rs.random() < 0.5 else refs[i % len(refs)]
                frac = float(rs.choice([0.05, 0.12, 0.16, 0.3, 0.6, 0.9]))
                qb = int(rs.integers(0, max(1, int(L * (1 - frac)))))
                qe = min(L, qb + max(1, int(L * frac)))
                tb = int(rs.integers(0, tsize - (qe - qb)))
                f.write(aln_record(name, L, t, tsize, bool(rs.random() < 0.7), qb, qe, tb, tb + (qe - qb)))
        if quirks:
            f.write(aln_record("ghost_ctg", 500, refs[0][0], refs[0][1], True, 10, 400, 100, 490))  # unknown contig name
            if case.get("broken_ctg_header"):
                f.write("broken header line\nACGT\nACGT\n")                                         # malformed header
            f.write(aln_record(ctgs[2][0], ctgs[2][1], refs[1][0], refs[1][1], True, 0, ctgs[2][1], 5, 5 + ctgs[2][1]))
            f.write("dangling header without its two lines 1 2 3\n")
    n = case["n_reads"]
    read_len = [int(rs.integers(80, 400)) for _ in range(n)]
    with open(os.path.join(d, "reads.fastq"), "w") as f:
        for i in range(n):
            s = _seq(rs, read_len[i])
            f.write(f"@read_{i + 1} original name\n{s}\n+\n{'I' * len(s)}\n")
        if quirks:
            f.write("@partial\nACGT\n")  # trailing partial record
    for fname, targets in (("read_to_ctg.ref", ctgs), ("read_to_ref.ref", refs)):
        with open(os.path.join(d, fname), "w") as f:
            for i in range(n):
                for _ in range(int(rs.integers(0, 3))):
                    t, tsize = targets[int(rs.integers(0, len(targets)))]
                    L = read_len[i]
                    qb = int(rs.integers(0, L // 2))
                    qe = int(rs.integers(qb + 1, L + 1))
                    tb = int(rs.integers(0, max(1, tsize - (qe - qb))))
                    f.write(aln_record(str(i + 1), L, t, tsize, bool(rs.random() < 0.5), qb, qe, tb, min(tsize, tb + (qe - qb))))
            if quirks:
                f.write(aln_record(str(n + 7), 100, targets[0][0], targets[0][1], True, 0, 50, 0, 50))  # id beyond the read file
                f.write("x y z\nAC\nAC\n")
    return d


def argv(exe, d, out, case):
    return [exe, "-r", os.path.join(d, "reads.fastq"), "-c", os.path.join(d, "ctg.fasta"), "-x", os.path.join(d, "read_to_ctg.ref"),
            "-y", os.path.join(d, "read_to_ref.ref"), "-z", os.path.join(d, "ctg_to_ref.ref"), "-o", out,
            "-k", str(case["top_k"]), "-m", repr(case["min_ratio"])]
