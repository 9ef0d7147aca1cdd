"""GPU: the ingest kernels (k_ingest.hip; SURVEY §8f.2 first half, north_star's "2-bit packing from aligned read blocks")
against the semantics of the reference's CompressedSeq (seq/CompressedSeq.cpp:8-38) and parseDiff
(align/ParseAlignTools.cpp:8-26), restated here in numpy, and against the arrays the product's host parsers (pinned by the
golden outputs) produce from the same files.  Bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import goldens
import pagctl

CODE = np.zeros(256, dtype=np.uint8)
for ch, v in ((b"C", 1), (b"c", 1), (b"G", 2), (b"g", 2), (b"T", 3), (b"t", 3)):
    CODE[ch[0]] = v


def pack_expect(seq: bytes) -> np.ndarray:
    n = len(seq)
    stored = ((n + 3) // 4 + 3) & ~3
    codes = np.zeros(stored * 4, dtype=np.uint8)
    codes[:n] = CODE[np.frombuffer(seq, dtype=np.uint8)]
    c = codes.reshape(-1, 4)
    return (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8)


def classify_expect(q: bytes, r: bytes):
    n = len(q)
    qa = np.frombuffer(q, dtype=np.uint8)
    ra = np.zeros(n, dtype=np.uint8)
    m = min(n, len(r))
    ra[:m] = np.frombuffer(r, dtype=np.uint8)[:m]
    cls = np.where(qa == ord("-"), 1, np.where(ra == ord("-"), 2, np.where(qa != ra, 3, 0))).astype(np.uint32)
    words = np.zeros((n + 15) // 16 * 16, dtype=np.uint32)
    words[:n] = cls
    w = (words.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
    return w, int((cls != 1).sum()), int((cls != 2).sum())


def _bind(hip):
    vp, u64 = C.c_void_p, C.c_uint64
    hip.pag_pack_text_seqs.argtypes = [vp, C.c_int, u64, vp, vp, u64, vp, vp, u64, C.c_int]
    hip.pag_pack_text_seqs.restype = C.c_int
    hip.pag_classify_columns.argtypes = [vp, C.c_int, u64, vp, vp, vp, vp, vp, u64, vp, u64, vp, vp, C.c_int]
    hip.pag_classify_columns.restype = C.c_int


def _pack_on_device(hip, text: bytes, spans, on_device):
    import torch
    off = np.array([a for a, _ in spans], dtype=np.uint64)
    ln = np.array([b for _, b in spans], dtype=np.uint32)
    stored = [(((int(b) + 3) // 4 + 3) & ~3) for b in ln]
    boff = np.concatenate([[0], np.cumsum(stored)]).astype(np.uint64)
    out = torch.full((int(boff[-1]) + 64,), 0xAB, dtype=torch.uint8, device="cuda")
    tbuf = np.frombuffer(text, dtype=np.uint8)
    if on_device:
        tdev = torch.from_numpy(tbuf.copy()).cuda()
        tptr = tdev.data_ptr()
    else:
        tptr = tbuf.ctypes.data
    boff0 = np.ascontiguousarray(boff[:-1])  # (kept alive across the call)
    rc = hip.pag_pack_text_seqs(tptr, 1 if on_device else 0, len(text), off.ctypes.data, ln.ctypes.data, len(spans), boff0.ctypes.data,
                                out.data_ptr(), int(boff[-1]), 0)
    assert rc == 0, hip.pag_last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy(), boff


@pytest.mark.gpu
@pytest.mark.parametrize("on_device", [True, False])
def test_text_sequences_are_packed_like_compressed_seq(on_device):
    hip = pagctl.hip_lib()
    _bind(hip)
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b"ACGTacgtNn-*xRYKM.", dtype=np.uint8)
    seqs = []
    for n in (0, 1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 1000, 4097, 70001):
        seqs.append(bytes(rng.choice(alphabet, size=n, p=None)))
    # the text: records glued together with junk in between, so that sequences start at every alignment
    text = b""
    spans = []
    for i, sq in enumerate(seqs):
        text += b"@" + str(i).encode() * (i % 5) + b"\n"
        spans.append((len(text), len(sq)))
        text += sq + b"\n+\n" + b"I" * len(sq) + b"\n"
    got, boff = _pack_on_device(hip, text, spans, on_device)
    for i, sq in enumerate(seqs):
        want = pack_expect(sq)
        assert np.array_equal(got[int(boff[i]):int(boff[i]) + len(want)], want), f"sequence {i} ({len(sq)} bases)"
    assert (got[int(boff[-1]):] == 0xAB).all()  # (nothing written past the last sequence)


@pytest.mark.gpu
def test_alignment_rows_are_classified_like_parse_diff():
    import torch
    hip = pagctl.hip_lib()
    _bind(hip)
    rng = np.random.default_rng(9)
    alphabet = np.frombuffer(b"ACGT-acgtN", dtype=np.uint8)
    recs = []
    for n in (0, 1, 15, 16, 17, 100, 1023, 1024, 1025, 20000):
        q = bytes(rng.choice(alphabet, size=n))
        r = bytearray(q)
        flips = rng.random(n) < 0.2
        r = bytes(np.where(flips, rng.choice(alphabet, size=n), np.frombuffer(q, dtype=np.uint8)).astype(np.uint8))
        recs.append((q, r))
    recs.append((b"ACGTACGTACGTACGTACGTAC", b"ACGTAC"))       # a reference row shorter than the query row: NUL past its end
    recs.append((b"ACGT", b"ACGTTTTTTTTTTTTTTTTTT"))          # ... and a longer one: ignored past the query row
    text, qo, ql, ro, rl = b"", [], [], [], []
    for i, (q, r) in enumerate(recs):
        text += b"hdr %d x F 9 0 1 2 0 1 2\n" % i
        qo.append(len(text)); ql.append(len(q)); text += q + b"\n"
        ro.append(len(text)); rl.append(len(r)); text += r + b"\n"
    nw = [(n + 15) // 16 for n in ql]
    doff = np.concatenate([[0], np.cumsum(nw)]).astype(np.uint64)
    diff = torch.full((int(doff[-1]) + 8,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")
    ne = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
    nr = torch.zeros(len(recs), dtype=torch.int32, device="cuda")
    tdev = torch.from_numpy(np.frombuffer(text, dtype=np.uint8).copy()).cuda()
    a = lambda x, dt: np.array(x, dtype=dt)  # noqa: E731
    qo_a, ql_a, ro_a, rl_a = a(qo, np.uint64), a(ql, np.uint32), a(ro, np.uint64), a(rl, np.uint32)
    doff0 = np.ascontiguousarray(doff[:-1])
    rc = hip.pag_classify_columns(tdev.data_ptr(), 1, len(text), qo_a.ctypes.data, ql_a.ctypes.data, ro_a.ctypes.data, rl_a.ctypes.data,
                                  doff0.ctypes.data, len(recs), diff.data_ptr(), int(doff[-1]), ne.data_ptr(), nr.data_ptr(), 0)
    assert rc == 0, hip.pag_last_error()
    torch.cuda.synchronize()
    got = diff.cpu().numpy().view(np.uint32)
    for i, (q, r) in enumerate(recs):
        w, e, rr = classify_expect(q, r)
        assert np.array_equal(got[int(doff[i]):int(doff[i]) + len(w)], w), f"record {i} ({len(q)} columns)"
        assert (int(ne[i]), int(nr[i])) == (e, rr), f"record {i} counts"
    assert (got[int(doff[-1]):] == 0x5A5A5A5A).all()


@pytest.mark.gpu
def test_a_golden_block_packs_to_what_the_host_parsers_hand_over(workdir):
    """the reads of a golden block (its FASTQ as text) through pag_pack_text_seqs = the packed array the product's FASTQ parser —
    pinned by the golden outputs — leaves for the same file"""
    hip = pagctl.hip_lib()
    _bind(hip)
    name = "three_ctg_multi_t4"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "ingest_in"))
    text = open(os.path.join(ind, "0.new.fastq"), "rb").read()
    lines = text.split(b"\n")
    spans, at = [], 0
    for i, ln in enumerate(lines):
        if i % 4 == 1:
            spans.append((at, len(ln)))
        at += len(ln) + 1
    got, boff = _pack_on_device(hip, text, spans, True)
    inp = pagctl.LoadedInput(ind, threads=spec["threads"], eps=spec["epsilon"], cov=spec["cov"])
    try:
        import biggen
        raw = C.cast(inp.raw_view, C.POINTER(biggen.PagRawInput)).contents
        n = raw.reads.n_seqs
        assert n == len(spans)
        h_off = np.ctypeslib.as_array(C.cast(raw.reads.byte_off, C.POINTER(C.c_uint64)), shape=(n,))
        h_len = np.ctypeslib.as_array(C.cast(raw.reads.len, C.POINTER(C.c_uint32)), shape=(n,))
        h_pk = np.ctypeslib.as_array(C.cast(raw.reads.packed, C.POINTER(C.c_uint8)), shape=(raw.reads.packed_bytes,))
        for i in range(n):
            assert int(h_len[i]) == spans[i][1]
            nb = (int(h_len[i]) + 3) // 4
            assert np.array_equal(got[int(boff[i]):int(boff[i]) + nb], h_pk[int(h_off[i]):int(h_off[i]) + nb]), f"read {i}"
    finally:
        inp.close()
