"""Seeded read sets for the k-mer counter tests (kmer_counter)."""
import numpy as np

# name -> generator parameters.  `alphabet` includes N / lower case to exercise "anything but CGT is an A".
CASES = {
    "k5_fastq_t4": dict(seed=1, n=60, lo=3, hi=90, k=5, threshold=0.2, threads=4, fmt="fastq", alphabet="ACGT"),
    "k6_fasta_t1_lower_n": dict(seed=2, n=40, lo=1, hi=120, k=6, threshold=0.2, threads=1, fmt="fasta", alphabet="ACGTacgtNn"),
    "k7_thr0_t16": dict(seed=3, n=200, lo=20, hi=200, k=7, threshold=0.0, threads=16, fmt="fastq", alphabet="ACGT"),
    "k4_thr09_t3_repeats": dict(seed=4, n=30, lo=50, hi=60, k=4, threshold=0.9, threads=3, fmt="fastq", alphabet="AAAAACGT"),
    "k6_neg_threshold_t8": dict(seed=5, n=50, lo=10, hi=150, k=6, threshold=-0.5, threads=8, fmt="fasta", alphabet="ACGT"),
}


def sequences(case):
    rs = np.random.default_rng(case["seed"])
    alpha = np.frombuffer(case["alphabet"].encode(), dtype=np.uint8)
    out = []
    for _ in range(case["n"]):
        L = int(rs.integers(case["lo"], case["hi"] + 1))
        out.append(alpha[rs.integers(0, len(alpha), L)].tobytes().decode())
    return out


def write_reads(case, path):
    seqs = sequences(case)
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            if case["fmt"] == "fastq":
                f.write(f"@r{i} x\n{s}\n+\n{'I' * len(s)}\n")
            else:
                f.write(f">r{i} x\n{s}\n")
    return seqs


def pack(seqs):
    """2-bit packing of CompressedSeq (4 bases per byte, LSB first, non-CGT -> A), every read 4-byte aligned."""
    lut = np.zeros(256, np.uint8)
    for ch, v in (("C", 1), ("c", 1), ("G", 2), ("g", 2), ("T", 3), ("t", 3)):
        lut[ord(ch)] = v
    offs, lens, chunks, cur = [], [], [], 0
    for s in seqs:
        c = lut[np.frombuffer(s.encode(), np.uint8)]
        pad = (-len(c)) % 16
        cc = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
        b = (cc[:, 0] | (cc[:, 1] << 2) | (cc[:, 2] << 4) | (cc[:, 3] << 6)).astype(np.uint8)
        offs.append(cur)
        lens.append(len(s))
        chunks.append(b)
        cur += len(b)
    packed = np.concatenate(chunks + [np.zeros(64, np.uint8)]) if chunks else np.zeros(64, np.uint8)
    return np.array(offs, np.uint64), np.array(lens, np.uint32), packed
