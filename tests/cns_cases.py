"""Seeded inputs for the pa_cns tests: a backbone FASTA and a 3-line ALN file of reads aligned to it
(header `q t F|R score qBegin qEnd qSize tBegin tEnd tSize`, aligned query row, aligned target row)."""
import os

import numpy as np

import synth

CASES = {
    "one_part": dict(seed=1, backbone=1800, n_reads=60, read_len=500, part=5000, top_k=3000, alpha=250),
    "three_parts_ties": dict(seed=2, backbone=2600, n_reads=140, read_len=700, part=1000, top_k=3000, alpha=250, score_classes=4),
    "topk_cut": dict(seed=3, backbone=1500, n_reads=120, read_len=400, part=600, top_k=25, alpha=40),
    "low_cov_quirks": dict(seed=4, backbone=2100, n_reads=18, read_len=600, part=700, top_k=3000, alpha=250, quirks=True),
    "boundary_ends": dict(seed=5, backbone=2000, n_reads=90, read_len=500, part=500, top_k=3000, alpha=7, snap=True),
}
# compared with the reference binary only (no golden file kept): the pipeline's default part length at ~190x coverage
DEEP_CASE = dict(seed=9, backbone=12000, n_reads=1500, read_len=1500, part=5000, top_k=3000, alpha=250)


def write_case(case, d):
    os.makedirs(d, exist_ok=True)
    rs = np.random.default_rng(case["seed"])
    L = case["backbone"]
    truth = synth.random_seq(rs, L)
    # the backbone is a noisy copy of the truth (what a draft path is), the reads are noisy copies of the truth too
    bb, _, _ = synth.mutate(rs, truth, 0.02, 0.01, 0.01)
    with open(os.path.join(d, "backbone.fasta"), "w") as f:
        s = bb.tobytes().decode()
        f.write(">P_1 some text\n" + "\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + "\n")
    Lb = len(bb)
    with open(os.path.join(d, "reads.ref"), "w") as f:
        for i in range(case["n_reads"]):
            n = min(Lb, int(case["read_len"] * (0.6 + 0.8 * rs.random())))
            tb = int(rs.integers(0, Lb - n + 1))
            if case.get("snap") and i % 3 == 0:
                tb = (tb // case["part"]) * case["part"]           # starts exactly on a part boundary
                n = min(Lb - tb, max(1, (n // case["part"]) * case["part"]))  # ... and ends on one
            seg = bb[tb:tb + n]
            q, qrow, trow = synth.mutate(rs, seg, 0.03, 0.05, 0.04)
            classes = case.get("score_classes")
            score = int(rs.integers(0, classes)) * 100 + 500 if classes else int((qrow == trow).sum())
            f.write(f"r{i} P_1 {'F' if rs.random() < 0.5 else 'R'} {score} 0 {len(q)} {len(q)} {tb} {tb + n} {Lb}\n"
                    f"{qrow.tobytes().decode()}\n{trow.tobytes().decode()}\n")
        if case.get("quirks"):
            # lower case / N in the rows, '.' gaps, a negative score
            seg = bb[100:400]
            q, qrow, trow = synth.mutate(rs, seg, 0.03, 0.05, 0.04)
            qs, ts = qrow.tobytes().decode().replace("-", ".", 3), trow.tobytes().decode()
            f.write(f"weird P_1 F -7 0 {len(q)} {len(q)} 100 400 {Lb}\n{qs.lower()[:50] + qs[50:]}\n{ts}\n")
            f.write(f"nrow P_1 F 12 0 {len(q)} {len(q)} 100 400 {Lb}\n{qs.replace('A', 'N', 5)}\n{ts}\n")
    return d


def argv(exe, d, out, case, threads=4):
    return [exe, "-t", str(threads), "-l", str(case["part"]), "-k", str(case["top_k"]), "--alpha", str(case["alpha"]),
            "-i", os.path.join(d, "backbone.fasta"), "-o", out, "-a", os.path.join(d, "reads.ref")]
