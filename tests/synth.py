"""Seeded synthetic-input generator for the PAGraph hot path (test + bench infrastructure).

Writes the exact file set `pagraph` consumes: ref.fasta, ctg.fasta, aln (contig->ref,
3-line ALN), <P>.new.fastq, <P>.ctg.ref, <P>.ref.ref (read->contig / read->ref 3-line ALN),
config.txt and kmer.bin.  Alignments are written from the simulation truth, so no aligner is needed.

Model: reference = i.i.d. ACGT (+ optional planted repeats); target genome = reference
with SNPs/indels; contigs = target segments separated by gaps (some stored reverse-complemented);
reads = fragments of the target with PacBio-CLR-like errors, either strand.

3-line ALN record (reference PAGraph/src/tools/align/AlignmentHelper.cpp:11-48):
    qName rName F|R score qBegin qEnd qSize rBegin rEnd rSize
    <aligned query, reference orientation, '-' gaps>
    <aligned reference, '-' gaps>
query coordinates are on the query's forward strand, half open, 0 based.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt-N", b"TGCAtgca-N"):
    _COMP[_a] = _b
GAP = ord("-")


def revcomp(a: np.ndarray) -> np.ndarray:
    return _COMP[a[::-1]]


def random_seq(rng: np.random.Generator, n: int) -> np.ndarray:
    return ACGT[rng.integers(0, 4, size=n)]


def mutate(rng: np.random.Generator, src: np.ndarray, sub: float, ins: float, dele: float):
    """Derive a query from `src`.  Returns (query, aligned_query, aligned_src): the two aligned rows
    have equal length, '-' marks gaps.  Vectorised: per src base one of {match, sub, del}, and after
    each base an insertion run of geometric length with probability `ins`."""
    n = len(src)
    u = rng.random(n)
    is_del = u < dele
    is_sub = (~is_del) & (u < dele + sub)
    qbase = src.copy()
    if is_sub.any():
        shift = rng.integers(1, 4, size=int(is_sub.sum()))
        code = np.searchsorted(ACGT, src[is_sub])
        qbase[is_sub] = ACGT[(code + shift) % 4]
    n_ins = np.where(rng.random(n) < ins, rng.geometric(0.7, size=n), 0)
    # never start/end the alignment with a gap column: aligners do not emit those
    if n:
        is_del[0] = is_del[-1] = False
        n_ins[-1] = 0
    cols_per = 1 + n_ins
    start = np.concatenate(([0], np.cumsum(cols_per)))[:-1]
    total = int(cols_per.sum())
    arow = np.full(total, GAP, dtype=np.uint8)
    srow = np.full(total, GAP, dtype=np.uint8)
    srow[start] = src
    arow[start] = np.where(is_del, GAP, qbase)
    ins_total = int(n_ins.sum())
    if ins_total:
        ins_mask = np.ones(total, dtype=bool)
        ins_mask[start] = False
        arow[ins_mask] = random_seq(rng, ins_total)
    query = arow[arow != GAP]
    return query, arow, srow


def compose(a_q: np.ndarray, a_t: np.ndarray, b_t: np.ndarray, b_r: np.ndarray, t_lo: int):
    """Compose alignment A (query vs target slice starting at target coordinate t_lo) with alignment B
    (whole target vs reference) into query vs reference.  Rows are uint8 with '-' gaps.  Returns
    (q_row, r_row, r_begin, r_end): columns where both rows would be gaps are dropped, leading/trailing
    gap-only-on-one-side columns are trimmed (returning trimmed query offsets too)."""
    # index B by target coordinate
    b_tpos = np.cumsum(b_t != GAP) - 1  # target coordinate of each B column (for non-gap target)
    b_rpos = np.cumsum(b_r != GAP) - 1
    t_cols = np.flatnonzero(b_t != GAP)  # B column index of target base j
    q_out, r_out = [], []
    t = t_lo
    first_r = None
    last_r = None
    bcol_prev = t_cols[t] - 1 if t < len(t_cols) else len(b_t) - 1
    for qa, ta in zip(a_q.tolist(), a_t.tolist()):
        if ta == GAP:  # insertion in query relative to target
            q_out.append(qa)
            r_out.append(GAP)
            continue
        bcol = int(t_cols[t])
        # reference bases inserted (relative to target) between previous target base and this one
        for c in range(bcol_prev + 1, bcol):
            if b_r[c] != GAP and b_t[c] == GAP:
                q_out.append(GAP)
                r_out.append(int(b_r[c]))
        rb = int(b_r[bcol])
        if qa == GAP and rb == GAP:
            pass
        else:
            q_out.append(qa)
            r_out.append(rb)
        bcol_prev = bcol
        t += 1
    q_row = np.array(q_out, dtype=np.uint8)
    r_row = np.array(r_out, dtype=np.uint8)
    # trim: alignment must start and end on a column with both bases present
    both = np.flatnonzero((q_row != GAP) & (r_row != GAP))
    if len(both) == 0:
        return None
    lo, hi = int(both[0]), int(both[-1]) + 1
    q_trim_lo = int((q_row[:lo] != GAP).sum())
    q_trim_hi = int((q_row[hi:] != GAP).sum())
    q_row, r_row = q_row[lo:hi], r_row[lo:hi]
    # reference coordinate of the first kept column
    first_bcol = int(t_cols[t_lo])
    r_before = int((b_r[:first_bcol] != GAP).sum())
    # reference bases consumed by trimmed leading columns
    r_lead = int((np.array(r_out[:lo], dtype=np.uint8) != GAP).sum())
    r_begin = r_before + r_lead
    # careful: inserted-ref columns before the first target base were emitted after bcol_prev init
    r_end = r_begin + int((r_row != GAP).sum())
    return q_row, r_row, r_begin, r_end, q_trim_lo, q_trim_hi


@dataclass
class Contig:
    name: str
    t_lo: int  # target interval
    t_hi: int
    reverse: bool  # stored sequence is revcomp of the target segment
    seq: np.ndarray = field(default=None, repr=False)


@dataclass
class Spec:
    seed: int = 1
    ref_len: int = 40_000
    n_refs: int = 1  # extra decoy reference sequences are appended when > 1
    contigs: list = None  # list of (t_lo, t_hi, reverse) in target coordinates; default: two with a gap
    n_reads: int = 600
    read_len: int = 1500
    read_len_jitter: float = 0.0
    k: int = 10
    target_snp: float = 0.01
    target_indel: float = 0.002
    read_sub: float = 0.03
    read_ins: float = 0.05
    read_del: float = 0.04
    rev_read_frac: float = 0.5
    clip_frac: float = 0.3  # fraction of alignments with soft-clipped ends
    repeats: int = 0  # number of planted repeats (2 copies each)
    repeat_len: int = 800
    solid_min_abundance: int = 1  # solid set = read k-mers (forward strand) with abundance >= this
    extra_ctg_aln: bool = False  # add a second, overlapping contig->ref alignment (multi-entry bases)
    dup_read_aln: bool = False  # add lower-scoring duplicate read->ref alignments for some reads
    min_ctg_overlap: float = 0.35
    # multi-block inputs (generate_multi): names of this block's reference / contigs, its config block number (prefix
    # of the read / alignment files), and contigs listed in config.txt with BOTH orientations (indices into `contigs`)
    ref_name: str = "ref1"
    ctg_prefix: str = "ctg"
    block: int = 0
    both_orient: tuple = ()
    # corner-case plants (used by the successor-record golden; no random numbers are drawn for them, so every other case's
    # inputs are unchanged): a homopolymer tract (start, length) written into the reference — k-mers with hundreds of
    # clustered positions —, and a "desert" (start, length) in target coordinates: every k-mer a read carries over that
    # stretch is taken out of the solid set, so that reads crossing it have edges whose step is the whole stretch
    homopolymer: tuple = ()
    desert: tuple = ()


def _aln_record(qname, rname, strand, score, qb, qe, qsize, rb, re_, rsize, qrow, rrow):
    return (f"{qname} {rname} {strand} {score} {qb} {qe} {qsize} {rb} {re_} {rsize}\n"
            f"{qrow.tobytes().decode()}\n{rrow.tobytes().decode()}\n")


def kmer_codes(seq: np.ndarray, k: int) -> np.ndarray:
    """2-bit rolling codes (reference PAGraph/src/tools/kmer/KmerHelper.cpp:7-25): A/other=0 C=1 G=2 T=3."""
    lut = np.zeros(256, dtype=np.uint64)
    for ch, v in zip(b"CGTcgt", (1, 2, 3, 1, 2, 3)):
        lut[ch] = v
    d = lut[seq]
    n = len(d) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    code = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        code = (code << np.uint64(2)) | d[j:j + n]
    return code


def write_kmer_file(path: str, k: int, codes: np.ndarray):
    """solid-set file: u64 k then u64 codes, native endian (reference kmer_counter.cpp:87-95)."""
    with open(path, "wb") as f:
        f.write(np.array([k], dtype=np.uint64).tobytes())
        f.write(np.asarray(codes, dtype=np.uint64).tobytes())


def generate(spec: Spec, out_dir: str) -> dict:
    """Write one complete pagraph input set into out_dir; returns a dict of paths + counts."""
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(spec.seed)
    G = spec.ref_len
    ref = random_seq(rng, G)
    for _ in range(spec.repeats):
        a = int(rng.integers(0, G - spec.repeat_len))
        b = int(rng.integers(0, G - spec.repeat_len))
        ref[b:b + spec.repeat_len] = ref[a:a + spec.repeat_len]
    if spec.homopolymer:
        ref[spec.homopolymer[0]:spec.homopolymer[0] + spec.homopolymer[1]] = ord("A")
    # target genome and its alignment to the reference (target = query, ref = ref)
    target, b_t, b_r = mutate(rng, ref, spec.target_snp, spec.target_indel / 2, spec.target_indel / 2)
    TL = len(target)

    ctg_specs = spec.contigs
    if ctg_specs is None:
        gap = max(200, TL // 25)
        half = (TL - gap) // 2
        ctg_specs = [(TL // 50, half, False), (half + gap, TL - TL // 50, False)]
    contigs = []
    for i, (lo, hi, rev) in enumerate(ctg_specs):
        seg = target[lo:hi]
        contigs.append(Contig(f"{spec.ctg_prefix}{i}", lo, hi, rev, revcomp(seg) if rev else seg.copy()))

    refs = [(spec.ref_name, ref)]
    for j in range(1, spec.n_refs):
        refs.append((f"{spec.ref_name}_decoy{j + 1}" if spec.ref_name != "ref1" else f"ref{j + 1}", random_seq(rng, max(1000, G // 4))))

    def fasta(path, records):
        with open(path, "w") as f:
            for name, seq in records:
                f.write(f">{name}\n")
                s = seq.tobytes().decode()
                for i in range(0, len(s), 80):
                    f.write(s[i:i + 80] + "\n")

    fasta(os.path.join(out_dir, "ref.fasta"), refs)
    fasta(os.path.join(out_dir, "ctg.fasta"), [(c.name, c.seq) for c in contigs])

    # contig -> reference ALN (score column is "NULL" as paf2aln.py writes it)
    ident_t = np.full(0, GAP, dtype=np.uint8)
    with open(os.path.join(out_dir, "aln"), "w") as f:
        for c in contigs:
            seg = target[c.t_lo:c.t_hi]
            res = compose(seg, seg, b_t, b_r, c.t_lo)
            q_row, r_row, rb, re_, tl, th = res
            clen = len(seg)
            # region of the segment that survived trimming: [tl, clen - th) in target orientation
            if c.reverse:
                qb, qe = th, clen - tl
                strand = "R"
            else:
                qb, qe = tl, clen - th
                strand = "F"
            f.write(_aln_record(c.name, spec.ref_name, strand, "NULL", qb, qe, clen, rb, re_, G, q_row, r_row))
            if spec.extra_ctg_aln and clen > 2000:
                # a second alignment of the contig's first third (same truth) -> multi-entry bases
                sub_hi = clen // 3
                seg2 = seg[:sub_hi]
                r2 = compose(seg2, seg2, b_t, b_r, c.t_lo)
                q2, rr2, rb2, re2, tl2, th2 = r2
                if c.reverse:
                    qb2, qe2 = clen - (sub_hi - th2), clen - tl2
                else:
                    qb2, qe2 = tl2, sub_hi - th2
                f.write(_aln_record(c.name, spec.ref_name, strand, "NULL", qb2, qe2, clen, rb2, re2, G, q2, rr2))
    del ident_t

    # reads
    B = spec.block
    fq = open(os.path.join(out_dir, f"{B}.new.fastq"), "w")
    f_ctg = open(os.path.join(out_dir, f"{B}.ctg.ref"), "w")
    f_ref = open(os.path.join(out_dir, f"{B}.ref.ref"), "w")
    all_codes = []
    desert_codes = []
    n_bases = 0
    for rid in range(1, spec.n_reads + 1):
        L0 = spec.read_len
        if spec.read_len_jitter > 0:
            L0 = max(spec.k + 5, int(L0 * (1 + spec.read_len_jitter * (rng.random() * 2 - 1))))
        L0 = min(L0, TL - 1)
        s = int(rng.integers(0, TL - L0))
        src = target[s:s + L0]
        frag, a_q, a_t = mutate(rng, src, spec.read_sub, spec.read_ins, spec.read_del)
        n = len(frag)
        read_rev = rng.random() < spec.rev_read_frac
        read = revcomp(frag) if read_rev else frag
        fq.write(f"@{rid}\n{read.tobytes().decode()}\n+\n{'~' * n}\n")
        n_bases += n
        all_codes.append(kmer_codes(read, spec.k))
        if spec.desert and n >= spec.k:
            ft = (s + np.cumsum(a_t != GAP) - 1)[a_q != GAP]  # target coordinate under every fragment base
            c = np.concatenate(([0], np.cumsum((ft >= spec.desert[0]) & (ft < spec.desert[0] + spec.desert[1]))))
            win = (c[spec.k:] - c[:-spec.k]) > 0  # k-mer windows of the fragment that touch the desert
            desert_codes.append(all_codes[-1][win[::-1] if read_rev else win])

        # optional soft clipping: drop some leading/trailing columns (whole columns)
        clo, chi = 0, len(a_q)
        if rng.random() < spec.clip_frac:
            clo = int(rng.integers(0, max(1, len(a_q) // 10)))
            chi = len(a_q) - int(rng.integers(0, max(1, len(a_q) // 10)))
        # clip must start/end on a both-present column
        both = np.flatnonzero((a_q != GAP) & (a_t != GAP))
        both = both[(both >= clo) & (both < chi)]
        if len(both) < 2:
            continue
        clo, chi = int(both[0]), int(both[-1]) + 1
        q_lo = int((a_q[:clo] != GAP).sum())  # fragment coordinate of first aligned base
        t_lo = s + int((a_t[:clo] != GAP).sum())
        cq, ct = a_q[clo:chi], a_t[clo:chi]
        q_hi = q_lo + int((cq != GAP).sum())
        t_hi = t_lo + int((ct != GAP).sum())

        # read -> reference
        res = compose(cq, ct, b_t, b_r, t_lo)
        if res is not None:
            q_row, r_row, rb, re_, tl, th = res
            fa, fb = q_lo + tl, q_hi - th  # fragment coordinates
            qb, qe = (n - fb, n - fa) if read_rev else (fa, fb)
            score = int((q_row == r_row).sum())
            f_ref.write(_aln_record(rid, spec.ref_name, "R" if read_rev else "F", score, qb, qe, n, rb, re_, G,
                                    q_row, r_row))
            if spec.dup_read_aln and rid % 7 == 0 and len(q_row) > 400:
                # a lower-scoring partial duplicate (first half of the columns, trimmed)
                half = len(q_row) // 2
                bb = np.flatnonzero((q_row[:half] != GAP) & (r_row[:half] != GAP))
                hi2 = int(bb[-1]) + 1
                q2, r2 = q_row[:hi2], r_row[:hi2]
                fb2 = fa + int((q2 != GAP).sum())
                re2 = rb + int((r2 != GAP).sum())
                qb2, qe2 = (n - fb2, n - fa) if read_rev else (fa, fb2)
                f_ref.write(_aln_record(rid, spec.ref_name, "R" if read_rev else "F", int((q2 == r2).sum()),
                                        qb2, qe2, n, rb, re2, G, q2, r2))

        # read -> contigs
        for c in contigs:
            olo, ohi = max(t_lo, c.t_lo), min(t_hi, c.t_hi)
            if ohi - olo < spec.min_ctg_overlap * 0.5 * n:
                continue
            # columns whose target coordinate lies in [olo, ohi)
            tcoord = t_lo + np.cumsum(ct != GAP) - 1
            sel = np.flatnonzero((ct != GAP) & (tcoord >= olo) & (tcoord < ohi) & (cq != GAP))
            if len(sel) < 2:
                continue
            x, y = int(sel[0]), int(sel[-1]) + 1
            sq, st = cq[x:y], ct[x:y]
            fa = q_lo + int((cq[:x] != GAP).sum())
            fb = fa + int((sq != GAP).sum())
            ta = t_lo + int((ct[:x] != GAP).sum())
            tb = ta + int((st != GAP).sum())
            clen = c.t_hi - c.t_lo
            if c.reverse:
                qrow, rrow = revcomp(sq), revcomp(st)
                strand = "F" if read_rev else "R"
                cb, ce = clen - (tb - c.t_lo), clen - (ta - c.t_lo)
            else:
                qrow, rrow = sq, st
                strand = "R" if read_rev else "F"
                cb, ce = ta - c.t_lo, tb - c.t_lo
            qb, qe = (n - fb, n - fa) if read_rev else (fa, fb)
            score = int((qrow == rrow).sum())
            f_ctg.write(_aln_record(rid, c.name, strand, score, qb, qe, n, cb, ce, clen, qrow, rrow))
    fq.close()
    f_ctg.close()
    f_ref.close()

    with open(os.path.join(out_dir, "config.txt"), "w") as f:
        f.write(f"{spec.ref_name}\n{B}.new.fastq\n{B}.ctg.ref\n{B}.ref.ref\n")
        for i, c in enumerate(contigs):
            if i in tuple(spec.both_orient):  # listed twice: the reference traverses both (PAssembly.cpp:28-36), the graph
                f.write(f"{c.name}\n{1 if c.reverse else 0}\n")  # is built with the LAST listed one (Aligner.cpp:311-320)
            f.write(f"{c.name}\n{0 if c.reverse else 1}\n")
        f.write("\n")

    codes = np.concatenate(all_codes) if all_codes else np.zeros(0, dtype=np.uint64)
    uniq, cnt = np.unique(codes, return_counts=True)
    solid = uniq[cnt >= spec.solid_min_abundance]
    if desert_codes:
        solid = np.setdiff1d(solid, np.concatenate(desert_codes))
    write_kmer_file(os.path.join(out_dir, "kmer.bin"), spec.k, solid)
    return {"dir": out_dir, "n_bases": n_bases, "n_solid": int(len(solid)), "k": spec.k,
            "contigs": [c.name for c in contigs]}


def generate_multi(specs, out_dir: str) -> dict:
    """Several config blocks in ONE pagraph input directory: every block is generated on its own (own reference sequence,
    own contigs, own read / alignment files <block>.*), the global files (ref.fasta, ctg.fasta, aln, kmer.bin, config.txt)
    are the concatenation / union over the blocks.  specs: list of Spec (names and block numbers are assigned here)."""
    import shutil
    import tempfile
    os.makedirs(out_dir, exist_ok=True)
    glob = {"ref.fasta": [], "ctg.fasta": [], "aln": [], "config.txt": []}
    solid = []
    info = []
    k = specs[0].k
    for b, sp in enumerate(specs):
        assert sp.k == k, "one solid-set file per run: all blocks share k"
        kw = dict(sp.__dict__)
        kw.update(ref_name=f"ref{b + 1}", ctg_prefix=f"b{b}ctg", block=b)
        with tempfile.TemporaryDirectory() as tmp:
            info.append(generate(Spec(**kw), tmp))
            for f in glob:
                glob[f].append(open(os.path.join(tmp, f), "rb").read())
            words = np.fromfile(os.path.join(tmp, "kmer.bin"), dtype=np.uint64)
            solid.append(words[1:])
            for f in (f"{b}.new.fastq", f"{b}.ctg.ref", f"{b}.ref.ref"):
                shutil.copy(os.path.join(tmp, f), os.path.join(out_dir, f))
    for f, parts in glob.items():
        open(os.path.join(out_dir, f), "wb").write(b"".join(parts))
    write_kmer_file(os.path.join(out_dir, "kmer.bin"), k, np.unique(np.concatenate(solid)))
    return {"dir": out_dir, "blocks": info, "k": k}


def pagraph_argv(binary: str, in_dir: str, out_dir: str, threads: int = 1, epsilon: int = 10, cov: int = 2,
                 min_len: int = 50):
    """The argv AlignGraph2.py uses (reference AlignGraph2.py:414-427), incl. the doubled -r (aligngraph2_amd.pagraph_argv)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import aligngraph2_amd
    return aligngraph2_amd.pagraph_argv(binary, in_dir, out_dir, threads, epsilon, cov, min_len)
