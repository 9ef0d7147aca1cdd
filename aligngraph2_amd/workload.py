"""Large synthetic PAGraph workloads generated directly as the flat C-ABI input (bench + scale tests).

Everything is built with torch on the chosen device, so that at bench scale (BASELINE.json config 2:
100k x 10 kb reads vs a 50 Mb reference) the inputs are resident in HBM before the timed region starts.
Model: reference = i.i.d. ACGT with planted repeats; TARGET genome = the reference with 1 % SNPs
and 0.2 % single-base indels (`target_snp`, `target_indel`; 0 / 0 gives the identity); contigs = segments of the
target (mean `ctg_len`, gaps 1-20 kb, ~10 % stored reverse-complemented), so contig and reference coordinates
drift apart and the contig->reference alignments carry gaps; reads = fixed span of the target, uniform start,
strand 50/50, PacBio-CLR-like errors 3 % sub / 4 % del / 5 % ins; read->contig alignments from the simulation
truth, read->reference alignments = that truth composed with the target->reference alignment; solid k-mer set by
the reference kmer_counter's rule (kmer_counter.cpp:68-77).

The same generator writes the TEXT form of a workload (FASTQ / 3-line ALN / FASTA / config / kmer.bin),
which is what the compiled reference needs for the cpu_baseline leg and for parity checks.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np
import torch

PAG_NONE = 0xFFFFFFFF
ALN_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("t_begin", "<u4"), ("t_end", "<u4"), ("q_start", "<u4"),
                      ("t_start", "<u4"), ("n_cols", "<u4"), ("n_valid", "<u4"), ("diff_off", "<u8"), ("flags", "<u4"),
                      ("reserved", "<u4")])
CTG_DTYPE = np.dtype([("len", "<u4"), ("selected", "<u4"), ("single_base", "<u4"), ("multi", "<u4"), ("map_off", "<u8")])
REF_DTYPE = np.dtype([("len", "<u4"), ("accepted", "<u4"), ("single_base", "<u4"), ("reserved", "<u4")])
FLAG_REV, FLAG_BACK, FLAG_ELIG = 1, 2, 4
# pag_raw_aln (include/pagraph_hip.h): an ALN record as the parser leaves it, names resolved to indices
RAW_DTYPE = np.dtype([("query", "<u4"), ("target", "<u4"), ("score", "<u8"), ("q_begin", "<u8"), ("q_end", "<u8"), ("t_begin", "<u8"),
                      ("t_end", "<u8"), ("diff_off", "<u8"), ("n_cols", "<u4"), ("n_emit", "<u4"), ("n_radv", "<u4"), ("forward", "<u4")])


class PagSeqs(C.Structure):
    _fields_ = [("n_seqs", C.c_uint64), ("byte_off", C.c_void_p), ("len", C.c_void_p), ("packed", C.c_void_p),
                ("packed_bytes", C.c_uint64)]


class PagAlnDb(C.Structure):
    _fields_ = [("n_aln", C.c_uint64), ("aln", C.c_void_p), ("query_off", C.c_void_p), ("diff", C.c_void_p),
                ("n_diff_words", C.c_uint64)]


class PagBuildInput(C.Structure):
    _fields_ = [("on_device", C.c_uint32), ("n_threads", C.c_uint32), ("reads", PagSeqs), ("emit_order", C.c_void_p),
                ("read_to_ctg", PagAlnDb), ("read_to_ref", PagAlnDb), ("n_ctgs", C.c_uint64), ("ctgs", C.c_void_p),
                ("ctg_ent_off", C.c_void_p), ("n_ctg_ent_off", C.c_uint64), ("ctg_ent", C.c_void_p),
                ("n_ctg_ent", C.c_uint64), ("n_refs", C.c_uint64), ("refs", C.c_void_p), ("eps", C.c_uint32),
                ("cov_filter", C.c_uint32), ("outer_sample", C.c_uint32), ("topk_ctg", C.c_int32), ("topk_ref", C.c_int32),
                ("reserved", C.c_uint32)]


class PagRawDb(C.Structure):
    _fields_ = [("n", C.c_uint64), ("rec", C.c_void_p), ("diff", C.c_void_p), ("n_diff_words", C.c_uint64)]


class PagRawInput(C.Structure):
    _fields_ = [("bulk_on_device", C.c_uint32), ("n_threads", C.c_uint32), ("reads", PagSeqs), ("read_to_ctg", PagRawDb),
                ("read_to_ref", PagRawDb), ("ctg_to_ref", PagRawDb), ("n_ctgs", C.c_uint64), ("ctg_len", C.c_void_p),
                ("ctg_selected", C.c_void_p), ("ctg_forward", C.c_void_p), ("n_refs", C.c_uint64), ("ref_len", C.c_void_p),
                ("ref_accepted", C.c_void_p), ("read_to_ctg_ratio", C.c_double), ("read_to_ref_ratio", C.c_double), ("eps", C.c_uint32),
                ("cov_filter", C.c_uint32), ("outer_sample", C.c_uint32), ("topk_ctg", C.c_int32), ("topk_ref", C.c_int32),
                ("reserved", C.c_uint32)]


@dataclass
class BigSpec:
    seed: int = 2
    ref_len: int = 50_000_000
    n_reads: int = 100_000
    read_span: int = 10_000  # reference bases under each read (read length ~ span * 1.01)
    k: int = 14
    ctg_len: int = 1_000_000
    gap_lo: int = 1_000
    gap_hi: int = 20_000
    rev_ctg_frac: float = 0.1
    sub: float = 0.03
    dele: float = 0.04
    ins: float = 0.05
    repeat_frac: float = 0.05
    target_snp: float = 0.01      # SNPs of the target genome against the reference
    target_indel: float = 0.002   # single-base indels (half insertions, half deletions)
    threads: int = 16  # the reference's -t: emission order + seed top-K
    eps: int = 10
    cov: int = 2
    solid_threshold: float = 0.2  # kmer_counter -m
    solid_min_abundance: int = -1  # >= 0: use this abundance cut instead of the kmer_counter rule
    chunk_reads: int = 8192


def _mapper_starts(lens):
    """PositionMapper::generateStartPosHelper (position/PositionMapper.cpp:16-31)."""
    start = [lens[0]]
    for i in range(1, len(lens)):
        start.append(start[-1] + 3 * lens[i - 1] + max(lens[i - 1], lens[i]))
    return start


class BigWorkload:
    """Holds every tensor of one config block + the ctypes view of it."""

    def __init__(self, spec: BigSpec, device: str = "cuda"):
        self.spec = spec
        self.dev = torch.device(device)
        self._keep = []
        self._generate()

    # ------------------------------------------------------------------ generation
    def _generate(self):
        sp, dev = self.spec, self.dev
        g = torch.Generator(device=dev)
        g.manual_seed(sp.seed)
        rs = np.random.default_rng(sp.seed)
        G, S, n = sp.ref_len, sp.read_span, sp.n_reads

        ref = torch.randint(0, 4, (G,), dtype=torch.uint8, device=dev, generator=g)
        # planted 2-10 kb repeats over ~repeat_frac of the reference
        n_rep = int(G * sp.repeat_frac / 6000)
        for _ in range(n_rep):
            L = int(rs.integers(2000, 10000))
            if G <= 2 * L:
                break
            a, b = int(rs.integers(0, G - L)), int(rs.integers(0, G - L))
            ref[b:b + L] = ref[a:a + L].clone()
        self.ref = ref

        # target genome: per reference base keep / SNP / delete, and after it possibly one inserted base.  (Deletions are
        # never adjacent and an insertion never follows a deleted base, so between two consecutive target bases at most
        # one reference-only column exists and every indel column is flanked by columns with both bases.)
        u = torch.rand(G, device=dev, generator=g)
        t_del = u < sp.target_indel / 2
        t_snp = (~t_del) & (u < sp.target_indel / 2 + sp.target_snp)
        t_ins = torch.rand(G, device=dev, generator=g) < sp.target_indel / 2
        t_del[0] = t_del[-1] = False
        t_ins[-1] = False
        t_del[1:] &= ~t_del[:-1]
        t_del[1:] &= ~t_ins[:-1]  # (no reference-only column right behind a target-only one)
        t_ins &= ~t_del
        t_ins[:-1] &= ~t_del[1:]
        t_snp &= ~t_del
        shift0 = torch.randint(1, 4, (G,), dtype=torch.uint8, device=dev, generator=g)
        tb = torch.where(t_snp, (ref + shift0) & 3, ref)
        ins_b = torch.randint(0, 4, (G,), dtype=torch.uint8, device=dev, generator=g)
        n_out = (~t_del).to(torch.int64) + t_ins.to(torch.int64)
        o_end = torch.cumsum(n_out, 0)
        o_beg = o_end - n_out
        LT = int(o_end[-1].item())
        tgt = torch.zeros(LT, dtype=torch.uint8, device=dev)
        g2r = torch.zeros(LT, dtype=torch.int64, device=dev)     # reference cursor at the target base (exactAlign's r)
        g_ins = torch.zeros(LT, dtype=torch.bool, device=dev)    # target base without a reference base
        ar = torch.arange(G, device=dev)
        keep = ~t_del
        tgt[o_beg[keep]] = tb[keep]
        g2r[o_beg[keep]] = ar[keep]
        ipos = (o_end - 1)[t_ins]
        tgt[ipos] = ins_b[t_ins]
        g2r[ipos] = ar[t_ins] + 1
        g_ins[ipos] = True
        # reference-only columns in front of a target base (0 or 1)
        adv = torch.where(g_ins, torch.zeros_like(g2r), torch.ones_like(g2r))
        g_rdel = torch.zeros(LT, dtype=torch.int64, device=dev)
        g_rdel[1:] = g2r[1:] - (g2r[:-1] + adv[:-1])
        assert int(g_rdel.min().item()) >= 0 and int(g_rdel.max().item()) <= 1
        self.tgt, self.g2r, self.g_ins, self.g_rdel = tgt, g2r, g_ins, g_rdel
        del u, t_del, t_snp, t_ins, shift0, tb, ins_b, n_out, o_end, o_beg, ar, keep, ipos, adv
        g_ins_np = g_ins.cpu().numpy()

        # contigs: segments of the target; they begin and end on target bases that have a reference base
        ctgs = []
        pos = int(rs.integers(0, max(1, min(sp.gap_hi, LT // 50))))
        while pos < LT - 2000:
            L = int(min(LT - pos, max(2000, rs.normal(sp.ctg_len, sp.ctg_len * 0.2))))
            s0, e0 = pos, pos + L
            while g_ins_np[s0]:
                s0 += 1
            while g_ins_np[e0 - 1]:
                e0 -= 1
            ctgs.append((s0, e0, bool(rs.random() < sp.rev_ctg_frac)))
            pos += L + int(rs.integers(sp.gap_lo, sp.gap_hi + 1))
        self.ctgs = ctgs
        nc = len(ctgs)
        ctg_lens = [e - s for s, e, _ in ctgs]
        ctg_start = _mapper_starts(ctg_lens)
        ref_start = _mapper_starts([G])
        if ctg_start[-1] + 4 * ctg_lens[-1] >= 2**32 - 1 or ref_start[-1] + 4 * G >= 2**32 - 1:
            raise ValueError("coordinate space exceeds 32 bits; split the reference (SURVEY §5)")
        self.ctg_mapper_start, self.ref_mapper_start = ctg_start, ref_start

        cs_t = torch.tensor([c[0] for c in ctgs], dtype=torch.int64, device=dev)
        ce_t = torch.tensor([c[1] for c in ctgs], dtype=torch.int64, device=dev)
        crev_t = torch.tensor([c[2] for c in ctgs], dtype=torch.bool, device=dev)

        starts = torch.randint(0, LT - S - 8, (n,), device=dev, generator=g)
        for _ in range(4):  # a read's span begins and ends on target bases that have a reference base
            bad = g_ins[starts] | g_ins[starts + S - 1]
            starts = torch.where(bad, starts + 1, starts)
        assert not bool((g_ins[starts] | g_ins[starts + S - 1]).any())
        rev = torch.rand(n, device=dev, generator=g) < 0.5

        read_len = torch.zeros(n, dtype=torch.int64, device=dev)
        packed_chunks, diff1_chunks, diff2_chunks = [], [], []
        rec1, rec2 = [], []  # per-chunk dicts of alignment fields
        byte_cursor = 0
        w1_cursor = 0
        w2_cursor = 0
        read_byte_off = torch.zeros(n, dtype=torch.int64, device=dev)
        kmer_hist = torch.zeros(4 ** sp.k, dtype=torch.int32, device=dev) if sp.k <= 14 else None
        code_chunks = []  # k > 14: the dense 4^k histogram is replaced by the list of k-mer codes seen
        k = sp.k

        for lo in range(0, n, sp.chunk_reads):
            hi = min(n, lo + sp.chunk_reads)
            m = hi - lo
            st = starts[lo:hi]
            idx = st[:, None] + torch.arange(S, device=dev)[None, :]
            rb = tgt[idx]  # [m, S] target bases under the read
            u = torch.rand(m, S, device=dev, generator=g)
            is_del = u < sp.dele
            is_sub = (~is_del) & (u < sp.dele + sp.sub)
            is_ins = torch.rand(m, S, device=dev, generator=g) < sp.ins
            is_del[:, 0] = False
            is_del[:, -1] = False
            is_ins[:, -1] = False
            shift = torch.randint(1, 4, (m, S), dtype=torch.uint8, device=dev, generator=g)
            qb = torch.where(is_sub, (rb + shift) & 3, rb)
            ins_base = torch.randint(0, 4, (m, S), dtype=torch.uint8, device=dev, generator=g)

            ncol_per = 1 + is_ins.to(torch.int32)
            col_end = torch.cumsum(ncol_per, dim=1)  # exclusive end column of each ref base (incl. its insertion)
            col0 = col_end - ncol_per  # column of the ref base itself
            n_cols = col_end[:, -1]  # [m]
            Cmax = int(n_cols.max().item())
            Cpad = (Cmax + 15) // 16 * 16
            # column classes: 0 match, 3 mismatch, 1 deletion (queryDiff only), 2 insertion (refDiff only)
            # scatter targets that do not exist go to a spare last column, sliced off afterwards
            W = Cpad + 16
            cls = torch.zeros(m, W, dtype=torch.uint8, device=dev)
            base_cls = torch.where(is_del, torch.ones_like(rb), torch.where(is_sub, torch.full_like(rb, 3), torch.zeros_like(rb)))
            cls.scatter_(1, col0.long(), base_cls)
            ins_col = torch.where(is_ins, (col0 + 1).long(), torch.full_like(col0.long(), W - 1))
            cls.scatter_(1, ins_col, torch.full_like(rb, 2))
            cls = cls[:, :Cpad].contiguous()
            # query base per column (undefined for deletions)
            qcol = torch.zeros(m, W, dtype=torch.uint8, device=dev)
            qcol.scatter_(1, col0.long(), qb)
            qcol.scatter_(1, ins_col, ins_base)
            qcol = qcol[:, :Cpad].contiguous()
            colmask = torch.arange(Cpad, device=dev)[None, :] < n_cols[:, None]
            emit = colmask & (cls != 1)
            # fragment = emitted bases in column order
            epos = torch.cumsum(emit.to(torch.int32), dim=1) - 1
            flen = (epos[:, -1] + 1).long()  # read length
            Lmax = int(flen.max().item())
            Lpad = (Lmax + 15) // 16 * 16
            frag = torch.zeros(m, Lpad + 16, dtype=torch.uint8, device=dev)
            tgt_col = torch.where(emit, epos.long(), torch.full_like(epos.long(), Lpad + 15))
            frag.scatter_(1, tgt_col, qcol)
            frag = frag[:, :Lpad].contiguous()
            # stored read = fragment or its reverse complement
            ar = torch.arange(Lpad, device=dev)[None, :]
            ridx = (flen[:, None] - 1 - ar).clamp(min=0)
            rc = 3 - frag.gather(1, ridx)
            rv = rev[lo:hi]
            stored = torch.where(rv[:, None], rc, frag)
            stored = torch.where(ar < flen[:, None], stored, torch.zeros_like(stored))
            read_len[lo:hi] = flen

            # k-mer histogram of the stored reads (forward strand only, like kmer_counter)
            if kmer_hist is not None:
                code = torch.zeros(m, Lpad - k + 1, dtype=torch.int64, device=dev)
                for j in range(k):
                    code = (code << 2) | stored[:, j:j + Lpad - k + 1].long()
                valid = torch.arange(Lpad - k + 1, device=dev)[None, :] < (flen[:, None] - k + 1)
                kmer_hist += torch.bincount(code[valid], minlength=4 ** k).to(torch.int32)
                del code, valid
            else:
                code = torch.zeros(m, Lpad - k + 1, dtype=torch.int64, device=dev)
                for j in range(k):
                    code = (code << 2) | stored[:, j:j + Lpad - k + 1].long()
                valid = torch.arange(Lpad - k + 1, device=dev)[None, :] < (flen[:, None] - k + 1)
                code_chunks.append(code[valid])
                del code, valid

            # pack reads: 4 bases / byte LSB first, each read padded to 16 bases (4 bytes)
            pb = (stored.view(m, Lpad // 4, 4).to(torch.int32) * torch.tensor([1, 4, 16, 64], device=dev)).sum(-1).to(torch.uint8)
            nbytes = ((flen + 15) // 16) * 4
            bmask = torch.arange(Lpad // 4, device=dev)[None, :] < nbytes[:, None]
            packed_chunks.append(pb[bmask])
            off = torch.cumsum(nbytes, 0) - nbytes + byte_cursor
            read_byte_off[lo:hi] = off
            byte_cursor += int(nbytes.sum().item())

            # ---- read -> ref alignment (whole read), columns in reference order
            def pack_cols(c, ncol):  # c: [m, Cpad] classes, rows valid for ncol columns
                cm = torch.arange(c.shape[1], device=dev)[None, :] < ncol[:, None]
                c = torch.where(cm, c, torch.zeros_like(c))
                w = (c.view(c.shape[0], -1, 16).to(torch.int64) << (2 * torch.arange(16, device=dev))).sum(-1)
                nw = (ncol + 15) // 16
                wm = torch.arange(w.shape[1], device=dev)[None, :] < nw[:, None]
                return w[wm].to(torch.int32), nw

            # read -> reference = (read -> target) o (target -> reference).  Per target base x of the span, in order: a
            # reference-only column if the reference has a base the target lacks in front of x (class 1), the base's own
            # column (both bases: 0 / 3 by comparing the READ base with the REFERENCE base; the read has it, the reference
            # does not: 2; the reference has it, the read does not: 1; neither: no column), the read's insertion (2).
            gi = g_ins[idx]
            rd = g_rdel[idx].clone()
            rd[:, 0] = 0
            rbase = ref[g2r[idx].clamp(max=G - 1)]
            base_present = ~(is_del & gi)
            base_cls2 = torch.where(is_del, torch.ones_like(rb), torch.where(gi, torch.full_like(rb, 2),
                                    torch.where(qb == rbase, torch.zeros_like(rb), torch.full_like(rb, 3))))
            ncol2_per = rd.to(torch.int32) + base_present.to(torch.int32) + is_ins.to(torch.int32)
            col2_end = torch.cumsum(ncol2_per, dim=1)
            col2_0 = (col2_end - ncol2_per).long()  # first column of base x
            n_cols2 = col2_end[:, -1].long()
            C2pad = (int(n_cols2.max().item()) + 15) // 16 * 16
            W2 = C2pad + 16
            cls2 = torch.zeros(m, W2, dtype=torch.uint8, device=dev)
            spare = torch.full_like(col2_0, W2 - 1)
            cls2.scatter_(1, torch.where(rd > 0, col2_0, spare), torch.ones_like(rb))
            cls2.scatter_(1, torch.where(base_present, col2_0 + rd, spare), base_cls2)
            cls2.scatter_(1, torch.where(is_ins, col2_0 + rd + base_present.long(), spare), torch.full_like(rb, 2))
            cls2 = cls2[:, :C2pad].contiguous()
            w2, nw2 = pack_cols(cls2, n_cols2)
            diff2_chunks.append(w2)
            off2 = torch.cumsum(nw2, 0) - nw2 + w2_cursor
            w2_cursor += int(nw2.sum().item())
            r_lo, r_hi = g2r[st], g2r[st + S - 1] + 1  # the reference interval under the read (both ends are aligned bases)
            rec2.append(dict(query=torch.arange(lo, hi, device=dev), target=torch.zeros(m, dtype=torch.int64, device=dev),
                             t_begin=r_lo, t_end=r_hi, q_start=torch.zeros(m, dtype=torch.int64, device=dev), t_start=r_lo,
                             n_cols=n_cols2, n_valid=flen, diff_off=off2,
                             flags=FLAG_ELIG + rv.long() * FLAG_REV, score=(cls2 == 0).sum(1) - (C2pad - n_cols2)))
            del gi, rd, rbase, base_present, base_cls2, ncol2_per, col2_end, col2_0, cls2, spare

            # ---- read -> contig alignments: up to two contigs per read
            j0 = torch.searchsorted(cs_t, st, right=True) - 1  # last contig starting at or before the read
            for cand in (0, 1):
                cj = j0 + cand
                ok = (cj >= 0) & (cj < nc)
                cjc = cj.clamp(0, nc - 1)
                c_s, c_e, c_r = cs_t[cjc], ce_t[cjc], crev_t[cjc]
                # the alignment must not end on the contig's last base IN CONTIG-FORWARD coordinates
                # (Aligner.tcc:62 drops `ctgEnd >= ctgLen`, quirk Q14): forward contigs lose their last
                # base, reverse-complemented contigs lose the base that sits first in reference order
                a = torch.maximum(st, c_s + c_r.long()) - st  # first ref base index inside the contig
                b = torch.minimum(st + S, c_e - 1 + c_r.long()) - st  # exclusive
                ok &= b - a >= 16
                a = a.clamp(0, S - 1)
                b = b.clamp(1, S)
                c0 = col0.gather(1, a[:, None]).squeeze(1).long()
                c1 = col0.gather(1, (b - 1)[:, None]).squeeze(1).long() + 1
                ncol = (c1 - c0).clamp(min=0)
                eb = torch.cat([torch.zeros(m, 1, dtype=torch.int32, device=dev), torch.cumsum(emit.to(torch.int32), 1)], 1)
                fa = eb.gather(1, c0[:, None]).squeeze(1).long()
                fb = eb.gather(1, c1[:, None]).squeeze(1).long()
                nq = fb - fa
                ok &= nq.double() / flen.double() >= 0.35  # readToCtgRatio (Aligner.tcc:52)
                if not bool(ok.any()):
                    continue
                sel = torch.nonzero(ok).squeeze(1)
                ms = sel.numel()
                cc = cls[sel]
                ncs, c0s, revs = ncol[sel], c0[sel], c_r[sel]
                Cs = int(ncs.max().item())
                Csp = (Cs + 15) // 16 * 16
                jj = torch.arange(Csp, device=dev)[None, :]
                src = torch.where(revs[:, None], c0s[:, None] + ncs[:, None] - 1 - jj, c0s[:, None] + jj).clamp(0, Cpad - 1)
                sub = cc.gather(1, src)  # reversed contigs store the columns in contig-forward (file) order
                w1, nw1 = pack_cols(sub, ncs)
                diff1_chunks.append(w1)
                off1 = torch.cumsum(nw1, 0) - nw1 + w1_cursor
                w1_cursor += int(nw1.sum().item())
                match = ((sub == 0) & (jj < ncs[:, None])).sum(1)
                tb_ref, te_ref = (a + st - c_s)[sel], (b + st - c_s)[sel]
                clen_s = (c_e - c_s)[sel]
                rec1.append(dict(query=sel + lo, target=cjc[sel], t_begin=torch.where(revs, clen_s - te_ref, tb_ref),
                                 t_end=torch.where(revs, clen_s - tb_ref, te_ref),
                                 q_start=fa[sel], t_start=(a + st - c_s)[sel], n_cols=ncs, n_valid=nq[sel], diff_off=off1,
                                 flags=FLAG_ELIG + rv[sel].long() * FLAG_REV + revs.long() * FLAG_BACK, score=match))
            del idx, rb, u, is_del, is_sub, is_ins, shift, qb, ins_base, cls, qcol, frag, stored, rc, pb

        self.n_bases = int(read_len.sum().item())
        self.read_len = read_len.to(torch.int32)
        self.read_byte_off = read_byte_off
        pad = torch.zeros(64, dtype=torch.uint8, device=dev)
        self.packed = torch.cat(packed_chunks + [pad])

        def finish_db(recs, diff_chunks):
            if recs:
                f = {kk: torch.cat([r[kk] for r in recs]) for kk in recs[0]}
            else:
                f = {kk: torch.zeros(0, dtype=torch.int64, device=dev) for kk in
                     ("query", "target", "t_begin", "t_end", "q_start", "t_start", "n_cols", "n_valid", "diff_off", "flags", "score")}
            # group by read; inside a read by score descending (stable on ties)
            order = torch.argsort(-f["score"], stable=True)
            order = order[torch.argsort(f["query"][order], stable=True)]
            f = {kk: v[order] for kk, v in f.items()}
            na = f["query"].numel()
            arr = np.zeros(na, dtype=ALN_DTYPE)
            for kk in ("query", "target", "t_begin", "t_end", "q_start", "t_start", "n_cols", "n_valid", "diff_off", "flags"):
                arr[kk] = f[kk].cpu().numpy()
            self._scores.append(f["score"].cpu().numpy().astype(np.uint64))
            qoff = torch.searchsorted(f["query"].contiguous(), torch.arange(n + 1, device=dev)).to(torch.int64)
            diff = torch.cat(diff_chunks + [torch.zeros(8, dtype=torch.int32, device=dev)]) if diff_chunks else torch.zeros(8, dtype=torch.int32, device=dev)
            return arr, qoff, diff

        self._scores = []
        self.aln1, self.qoff1, self.diff1 = finish_db(rec1, diff1_chunks)
        self.aln2, self.qoff2, self.diff2 = finish_db(rec2, diff2_chunks)
        self.score1, self.score2 = self._scores

        # contig table + contig->ref map (AlignReference: one entry per contig base = the reference cursor at that base,
        # both orientations alike)
        ctab = np.zeros(nc, dtype=CTG_DTYPE)
        ent_off, ents = [], []
        cursor = 0
        for c, (s, e, r) in enumerate(ctgs):
            L = e - s
            ctab[c] = (L, 1, (ctg_start[c] + (2 * L if r else 0)) & 0xFFFFFFFF, 0, cursor + c)
            ent_off.append(torch.arange(cursor, cursor + L + 1, dtype=torch.int64, device=dev))
            ents.append(g2r[s:e] + ref_start[0])
            cursor += L
        self.ctab = ctab
        self.ctg_ent_off = torch.cat(ent_off).to(torch.int32)
        self.ctg_ent = torch.cat(ents + [torch.zeros(1, dtype=torch.int64, device=dev)]).to(torch.int32)
        self.rtab = np.array([(G, 1, ref_start[0], 0)], dtype=REF_DTYPE)

        T = max(1, sp.threads)
        self.emit_order = torch.cat([torch.arange(t, n, T, device=dev) for t in range(T)]).to(torch.int32)

        # solid set, kmer_counter's rule (kmer_counter.cpp:60-77): smallest abundance a such that the
        # fraction of codes with abundance > a is <= threshold; solid = abundance >= a
        if kmer_hist is not None:
            if sp.solid_min_abundance >= 0:
                min_ab = sp.solid_min_abundance
            else:
                mx = int(kmer_hist.max().item())
                cnt = torch.bincount(kmer_hist.long().clamp(max=mx), minlength=mx + 1)
                cum = torch.cumsum(cnt, 0).double()
                okk = (1.0 - cum / float(4 ** k)) <= sp.solid_threshold
                min_ab = int(torch.nonzero(okk)[0].item())
            solid = kmer_hist >= min_ab
            solid[k] = True  # the file's header word (value k) is ingested as a code (quirk Q1)
            self.min_abundance = min_ab
            self.n_solid = int(solid.sum().item())
            bits = (solid.view(-1, 32).to(torch.int64) << torch.arange(32, device=dev)).sum(-1)
            self.solid_bits = bits.to(torch.int32)
            self.solid_mask = solid
            self.solid_codes = None
        else:
            # sparse form of the same rule (the zero-abundance codes are counted, not stored)
            uniq, counts = torch.unique(torch.cat(code_chunks), return_counts=True)
            if sp.solid_min_abundance >= 0:
                min_ab = sp.solid_min_abundance
            else:
                mx = int(counts.max().item())
                cnt = torch.bincount(counts, minlength=mx + 1)
                cnt[0] = 4 ** k - uniq.numel()
                cum = torch.cumsum(cnt, 0).double()
                okk = ((1.0 - cum / float(4 ** k)) <= sp.solid_threshold) & (cnt > 0)
                min_ab = int(torch.nonzero(okk)[0].item())
            if min_ab <= 0:
                raise ValueError("k > 14 with an all-solid set is not generated (4^k codes)")
            codes = torch.unique(torch.cat([uniq[counts >= min_ab], torch.tensor([k], dtype=torch.int64, device=dev)]))  # + Q1
            self.min_abundance = min_ab
            self.n_solid = int(codes.numel())
            words = torch.zeros(4 ** k // 32, dtype=torch.int64, device=dev)
            words.index_add_(0, codes >> 5, torch.ones_like(codes) << (codes & 31))
            self.solid_bits = torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32)
            self.solid_mask = None
            self.solid_codes = codes
        self._to_device_tables()

    def _to_device_tables(self):
        dev = self.dev
        self.aln1_t = torch.from_numpy(self.aln1.view(np.uint8).copy()).to(dev)
        self.aln2_t = torch.from_numpy(self.aln2.view(np.uint8).copy()).to(dev)
        self.ctab_t = torch.from_numpy(self.ctab.view(np.uint8).copy()).to(dev)
        self.rtab_t = torch.from_numpy(self.rtab.view(np.uint8).copy()).to(dev)

    def clone_to(self, device: str) -> "BigWorkload":
        """the same workload with every tensor on another device (e.g. 'cpu' for the oracle)"""
        import copy
        o = copy.copy(self)
        o.dev = torch.device(device)
        for name, v in list(vars(self).items()):
            if isinstance(v, torch.Tensor):
                setattr(o, name, v.to(device))
        return o

    # ------------------------------------------------------------------ C-ABI view
    def build_input(self) -> PagBuildInput:
        sp = self.spec
        on_dev = 1 if self.dev.type == "cuda" else 0
        p = lambda t: t.data_ptr()  # noqa: E731
        inp = PagBuildInput()
        inp.on_device = on_dev
        inp.n_threads = sp.threads
        inp.reads = PagSeqs(sp.n_reads, p(self.read_byte_off), p(self.read_len), p(self.packed), self.packed.numel())
        inp.emit_order = p(self.emit_order)
        inp.read_to_ctg = PagAlnDb(len(self.aln1), p(self.aln1_t), p(self.qoff1), p(self.diff1), self.diff1.numel())
        inp.read_to_ref = PagAlnDb(len(self.aln2), p(self.aln2_t), p(self.qoff2), p(self.diff2), self.diff2.numel())
        inp.n_ctgs = len(self.ctab)
        inp.ctgs = p(self.ctab_t)
        inp.ctg_ent_off = p(self.ctg_ent_off)
        inp.n_ctg_ent_off = self.ctg_ent_off.numel()
        inp.ctg_ent = p(self.ctg_ent)
        inp.n_ctg_ent = self.ctg_ent.numel()
        inp.n_refs = 1
        inp.refs = p(self.rtab_t)
        inp.eps = sp.eps
        inp.cov_filter = sp.cov
        inp.outer_sample = 3
        inp.topk_ctg = -1
        inp.topk_ref = -1
        return inp

    # ------------------------------------------------------------------ C-ABI view: records as the PARSER leaves them
    def raw_input(self) -> PagRawInput:
        """The block the way bin/pagraph's parsers hand it to pag_prepare (include/pagraph_hip.h): alignment records with
        their HEADER fields (what write_text puts on the header lines: intervals on the forward strands, the strand column,
        the score), in database order = by score, descending (AlnDb::sortByScore); contig / reference tables; the bulk
        arrays (packed reads, column classes) stay where they are (HBM).  The eligibility tests, flips, n_valid, the
        per-read lists and the contig->reference map are the device stage's work."""
        sp = self.spec
        dev = self.dev
        keep = {}

        def class_counts(diff, arr):
            # per record: columns of class 1 (target only) and class 2 (query only); padding columns are class 0
            w = diff.to(torch.int64) & 0xFFFFFFFF
            lo, hi = w & 0x55555555, (w >> 1) & 0x55555555
            c1 = lo & ~hi & 0x55555555
            c2 = hi & ~lo & 0x55555555

            def popc(x):
                x = x - ((x >> 1) & 0x55555555)
                x = (x & 0x33333333) + ((x >> 2) & 0x33333333)
                x = (x + (x >> 4)) & 0x0F0F0F0F
                return (x * 0x01010101 >> 24) & 0xFF
            z = torch.zeros(1, dtype=torch.int64, device=diff.device)
            s1 = torch.cat([z, torch.cumsum(popc(c1), 0)])
            s2 = torch.cat([z, torch.cumsum(popc(c2), 0)])
            off = torch.from_numpy(arr["diff_off"].astype(np.int64)).to(diff.device)
            nw = torch.from_numpy(((arr["n_cols"].astype(np.int64) + 15) // 16)).to(diff.device)
            return (s1[off + nw] - s1[off]).cpu().numpy(), (s2[off + nw] - s2[off]).cpu().numpy()

        rlen = self.read_len.cpu().numpy().astype(np.int64)

        def raw_db(arr, diff, tlen_of, scores):
            n = len(arr)
            raw = np.zeros(n, dtype=RAW_DTYPE)
            k1, k2 = class_counts(diff, arr) if n else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            ncol = arr["n_cols"].astype(np.int64)
            n_emit, n_radv = ncol - k1, ncol - k2
            rev = (arr["flags"] & FLAG_REV) != 0
            back = (arr["flags"] & FLAG_BACK) != 0
            qs, ts = arr["q_start"].astype(np.int64), arr["t_start"].astype(np.int64)
            nrd = rlen[arr["query"]]
            tlen = tlen_of(arr)
            raw["query"], raw["target"] = arr["query"], arr["target"]
            raw["score"] = scores
            raw["q_begin"] = np.where(rev, nrd - (qs + n_emit), qs)
            raw["q_end"] = np.where(rev, nrd - qs, qs + n_emit)
            raw["t_begin"] = np.where(back, tlen - (ts + n_radv), ts)
            raw["t_end"] = np.where(back, tlen - ts, ts + n_radv)
            raw["forward"] = np.where(back, rev, ~rev)
            raw["diff_off"], raw["n_cols"], raw["n_emit"], raw["n_radv"] = arr["diff_off"], ncol, n_emit, n_radv
            order = np.argsort(-scores.astype(np.int64), kind="stable")  # the database is sorted by score
            raw = np.ascontiguousarray(raw[order])
            keep[id(raw)] = raw
            return PagRawDb(n, raw.ctypes.data, diff.data_ptr(), diff.numel()), raw

        clen = np.array([e - s for s, e, _ in self.ctgs], dtype=np.int64)
        db1, self.raw1 = raw_db(self.aln1, self.diff1, lambda a: clen[a["target"]], self.score1)
        db2, self.raw2 = raw_db(self.aln2, self.diff2, lambda a: np.full(len(a), len(self.ref), np.int64), self.score2)
        # contig -> reference: one record per contig (write_text's `aln` file), score = qEnd - qBegin
        ctg_diff, ctg_raw = self._ctg_to_ref_db()
        keep["ctg_diff"] = ctg_diff
        db3 = PagRawDb(len(ctg_raw), ctg_raw.ctypes.data, ctg_diff.data_ptr(), ctg_diff.numel())
        inp = PagRawInput()
        inp.bulk_on_device = 1 if dev.type == "cuda" else 0
        inp.n_threads = sp.threads
        inp.reads = PagSeqs(sp.n_reads, self.read_byte_off.data_ptr(), self.read_len.data_ptr(), self.packed.data_ptr(), self.packed.numel())
        inp.read_to_ctg, inp.read_to_ref, inp.ctg_to_ref = db1, db2, db3
        t_clen = clen.astype(np.uint32)
        t_sel = np.ones(len(clen), np.uint8)
        t_fwd = np.array([0 if r else 1 for _, _, r in self.ctgs], np.uint8)
        t_rlen = np.array([len(self.ref)], np.uint32)
        t_racc = np.ones(1, np.uint8)
        keep.update(tables=(t_clen, t_sel, t_fwd, t_rlen, t_racc, ctg_raw))
        inp.n_ctgs, inp.ctg_len, inp.ctg_selected, inp.ctg_forward = len(clen), t_clen.ctypes.data, t_sel.ctypes.data, t_fwd.ctypes.data
        inp.n_refs, inp.ref_len, inp.ref_accepted = 1, t_rlen.ctypes.data, t_racc.ctypes.data
        inp.read_to_ctg_ratio, inp.read_to_ref_ratio = 0.35, 0.10
        inp.eps, inp.cov_filter, inp.outer_sample, inp.topk_ctg, inp.topk_ref = sp.eps, sp.cov, 3, -1, -1
        self._raw_keep = keep
        return inp

    def _ctg_to_ref_db(self):
        """the contig->reference alignments of write_text as column classes + raw records (built on the device)"""
        dev = self.dev
        G = len(self.ref)
        words, recs, cursor = [], np.zeros(len(self.ctgs), dtype=RAW_DTYPE), 0
        for c, (s, e, r) in enumerate(self.ctgs):
            seg = self.tgt[s:e]
            rd = self.g_rdel[s:e].clone().to(torch.int64)
            rd[0] = 0
            ncol = rd + 1
            col0 = torch.cumsum(ncol, 0) - 1
            Ccols = int(col0[-1].item()) + 1
            has_r = ~self.g_ins[s:e]
            # class per column: reference-only columns (query gap) = 1; the base's own column: 2 if the reference has no
            # base there, else 0 / 3 by comparing the bases
            cls = torch.ones(Ccols, dtype=torch.int64, device=dev)
            rbase = self.ref[self.g2r[s:e].clamp(max=G - 1)]
            own = torch.where(has_r, torch.where(seg == rbase, torch.zeros_like(col0), torch.full_like(col0, 3)), torch.full_like(col0, 2))
            cls[col0] = own
            pad = (-Ccols) % 16
            cw = torch.cat([cls, torch.zeros(pad, dtype=torch.int64, device=dev)]).view(-1, 16)
            w = (cw << (2 * torch.arange(16, device=dev))).sum(-1)
            words.append(w)
            rb, re_ = int(self.g2r[s].item()), int(self.g2r[e - 1].item()) + 1
            n1, n2 = int((cls == 1).sum().item()), int((cls == 2).sum().item())
            recs[c] = (c, 0, e - s, 0, e - s, rb, re_, cursor, Ccols, Ccols - n1, Ccols - n2, 0 if r else 1)
            cursor += w.numel()
        diff = torch.cat(words + [torch.zeros(8, dtype=torch.int64, device=dev)])
        diff = torch.where(diff >= 2 ** 31, diff - 2 ** 32, diff).to(torch.int32)
        order = np.argsort(-recs["score"].astype(np.int64), kind="stable")
        return diff, np.ascontiguousarray(recs[order])

    def contig_codes(self):
        """the contig sequences as stored (2-bit codes): target segments, reverse-complemented where the contig is"""
        t = self.tgt.cpu().numpy()
        return [(3 - t[s:e][::-1]) if r else t[s:e] for s, e, r in self.ctgs]

    def solid_words(self) -> np.ndarray:
        """every u64 word of the equivalent solid-set file (header word first)"""
        if self.solid_codes is not None:
            codes = self.solid_codes.cpu().numpy().astype(np.uint64)
        else:
            codes = torch.nonzero(self.solid_mask).squeeze(1).cpu().numpy().astype(np.uint64)
        return np.concatenate([np.array([self.spec.k], dtype=np.uint64), codes])

    # ------------------------------------------------------------------ text form (for the reference binary)
    def write_text(self, out_dir: str):
        os.makedirs(out_dir, exist_ok=True)
        sp = self.spec
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        comp = np.frombuffer(b"TGCA", dtype=np.uint8)
        ref = self.ref.cpu().numpy()
        G = len(ref)

        def fasta(path, recs):
            with open(path, "w") as f:
                for name, codes in recs:
                    f.write(f">{name}\n")
                    s = acgt[codes].tobytes().decode()
                    f.write("\n".join(s[i:i + 100] for i in range(0, len(s), 100)) + "\n")

        fasta(os.path.join(out_dir, "ref.fasta"), [("ref1", ref)])
        tgt = self.tgt.cpu().numpy()
        g2r = self.g2r.cpu().numpy()
        g_ins = self.g_ins.cpu().numpy()
        g_rdel = self.g_rdel.cpu().numpy()
        gap_tab = np.frombuffer(b"ACGT-", dtype=np.uint8)
        ctg_recs = []
        with open(os.path.join(out_dir, "aln"), "w") as f:
            for c, (s, e, r) in enumerate(self.ctgs):
                seg = tgt[s:e]
                seq = (3 - seg[::-1]) if r else seg
                ctg_recs.append((f"ctg{c}", seq))
                # the contig -> reference alignment in reference orientation: per target base an optional reference-only
                # column in front of it, then its own column (target-only if the reference has no base there)
                rd = g_rdel[s:e].copy()
                rd[0] = 0
                ncol = rd + 1
                col0 = np.cumsum(ncol) - 1  # column of the target base
                C = int(col0[-1]) + 1
                q_row = np.full(C, 4, dtype=np.uint8)
                t_row = np.full(C, 4, dtype=np.uint8)
                q_row[col0] = seg
                has_r = ~g_ins[s:e]
                t_row[col0[has_r]] = ref[g2r[s:e][has_r]]
                dcols = col0[rd > 0] - 1
                t_row[dcols] = ref[g2r[s:e][rd > 0] - 1]
                rb, re_ = int(g2r[s]), int(g2r[e - 1]) + 1
                f.write(f"ctg{c} ref1 {'R' if r else 'F'} NULL 0 {e - s} {e - s} {rb} {re_} {G}\n"
                        f"{gap_tab[q_row].tobytes().decode()}\n{gap_tab[t_row].tobytes().decode()}\n")
        fasta(os.path.join(out_dir, "ctg.fasta"), ctg_recs)

        packed = self.packed.cpu().numpy()
        boff = self.read_byte_off.cpu().numpy()
        rlen = self.read_len.cpu().numpy()

        def read_codes(i):
            b = packed[boff[i]:boff[i] + (rlen[i] + 3) // 4]
            return np.stack([(b >> s) & 3 for s in (0, 2, 4, 6)], 1).reshape(-1)[:rlen[i]]

        with open(os.path.join(out_dir, "0.new.fastq"), "w") as f:
            for i in range(sp.n_reads):
                s = acgt[read_codes(i)].tobytes().decode()
                f.write(f"@{i + 1}\n{s}\n+\n{'~' * len(s)}\n")

        def write_aln(path, arr, diff, target_name, target_len, target_seq):
            diff = diff.cpu().numpy().view(np.uint32)
            with open(path, "w") as f:
                for a in arr:
                    i = int(a["query"])
                    nc_ = int(a["n_cols"])
                    w = diff[int(a["diff_off"]):int(a["diff_off"]) + (nc_ + 15) // 16]
                    cls = np.stack([(w >> (2 * j)) & 3 for j in range(16)], 1).reshape(-1)[:nc_]  # file order
                    rev_read = bool(a["flags"] & FLAG_REV)
                    back = bool(a["flags"] & FLAG_BACK)
                    tname, tlen, tseq = target_name(a), target_len(a), target_seq(a)
                    rc = read_codes(i)
                    n = len(rc)
                    walk = cls[::-1] if back else cls
                    n_emit = int((walk != 1).sum())
                    n_radv = int((walk != 2).sum())
                    qs, ts = int(a["q_start"]), int(a["t_start"])
                    # strand string the positions refer to
                    strand = (3 - rc[::-1]) if rev_read else rc
                    GAPC = 4
                    fwd_tab = np.frombuffer(b"ACGT-", dtype=np.uint8)
                    cmp_tab = np.frombuffer(b"TGCA-", dtype=np.uint8)
                    q_walk = np.full(nc_, GAPC, dtype=np.uint8)
                    t_walk = np.full(nc_, GAPC, dtype=np.uint8)
                    q_walk[walk != 1] = strand[qs:qs + n_emit]
                    t_walk[walk != 2] = tseq[ts:ts + n_radv]
                    if back:
                        # file rows are in contig-forward orientation = reverse complement of the walk rows;
                        # relative strand flips; contig coordinates go back to the contig's forward strand
                        q_row, t_row = cmp_tab[q_walk[::-1]], cmp_tab[t_walk[::-1]]
                        f_is_forward = rev_read
                        tb, te = tlen - (ts + n_radv), tlen - ts
                    else:
                        q_row, t_row = fwd_tab[q_walk], fwd_tab[t_walk]
                        f_is_forward = not rev_read
                        tb, te = ts, ts + n_radv
                    # header query coordinates are on the read's forward strand
                    qb, qe = (n - (qs + n_emit), n - qs) if rev_read else (qs, qs + n_emit)
                    score = int((cls == 0).sum())
                    f.write(f"{i + 1} {tname} {'F' if f_is_forward else 'R'} {score} {qb} {qe} {n} {tb} {te} {tlen}\n"
                            f"{q_row.tobytes().decode()}\n{t_row.tobytes().decode()}\n")

        ctg_seqs = {}

        def ctg_strand_seq(a):  # the contig strand the walk's coordinates refer to = reference orientation
            c = int(a["target"])
            if c not in ctg_seqs:
                s, e, r = self.ctgs[c]
                ctg_seqs[c] = tgt[s:e]
            return ctg_seqs[c]

        write_aln(os.path.join(out_dir, "0.ctg.ref"), self.aln1, self.diff1, lambda a: f"ctg{int(a['target'])}",
                  lambda a: self.ctgs[int(a["target"])][1] - self.ctgs[int(a["target"])][0], ctg_strand_seq)
        write_aln(os.path.join(out_dir, "0.ref.ref"), self.aln2, self.diff2, lambda a: "ref1", lambda a: G, lambda a: ref)
        with open(os.path.join(out_dir, "config.txt"), "w") as f:
            f.write("ref1\n0.new.fastq\n0.ctg.ref\n0.ref.ref\n")
            for c, (s, e, r) in enumerate(self.ctgs):
                f.write(f"ctg{c}\n{0 if r else 1}\n")
            f.write("\n")
        with open(os.path.join(out_dir, "kmer.bin"), "wb") as f:
            f.write(self.solid_words().tobytes())
        return out_dir
