"""Packed alignment sidecars (host ingest): `<aln>.pagaln` written on request, loaded instead of the text
when its stamp (size + mtime of the text file) matches, ignored when stale or damaged.  The graph built from an input
loaded through sidecars must equal the one built from the text, stream for stream (checked with the oracle: CPU only)."""
import glob
import os

import numpy as np
import pytest

import pagctl
import synth


def _streams(in_dir):
    inp = pagctl.LoadedInput(in_dir, threads=4, eps=10, cov=2)
    try:
        r = pagctl.run_oracle(inp, streams=True)
        return {k: v.copy() for k, v in r["streams"].items()}, r["stats"].counts()
    finally:
        inp.close()


def _same(a, b):
    assert a[1] == b[1]
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k


@pytest.fixture()
def in_dir(workdir):
    d = str(workdir / "sidecar_in")
    synth.generate(synth.Spec(seed=31, ref_len=9000, n_reads=150, read_len=900, read_len_jitter=0.3, k=9,
                              contigs=[(200, 4300, False), (4500, 8800, True)]), d)
    return d


def _aln_files(d):
    return sorted(p for p in glob.glob(os.path.join(d, "**", "*"), recursive=True)
                  if os.path.isfile(p) and p.endswith(".pagaln"))


def test_sidecar_roundtrip_and_invalidation(in_dir, monkeypatch):
    monkeypatch.delenv("PAGRAPH_ALN_SIDECAR", raising=False)
    base = _streams(in_dir)
    assert _aln_files(in_dir) == []  # nothing is written unless asked for

    monkeypatch.setenv("PAGRAPH_ALN_SIDECAR", "1")
    _same(_streams(in_dir), base)
    side = _aln_files(in_dir)
    assert len(side) == 3  # read->contig, read->reference, contig->reference
    monkeypatch.delenv("PAGRAPH_ALN_SIDECAR")

    # the sidecars are what is read now: the text files are replaced by same-size junk with the original time stamps
    saved = {}
    for s in side:
        txt = s[:-len(".pagaln")]
        st = os.stat(txt)
        saved[txt] = (open(txt, "rb").read(), st)
        with open(txt, "wb") as f:
            f.write(b"x" * st.st_size)
        os.utime(txt, ns=(st.st_atime_ns, st.st_mtime_ns))
    _same(_streams(in_dir), base)

    # restore the text: a newer time stamp makes the sidecar stale, the text is parsed again, same result
    for txt, (data, st) in saved.items():
        with open(txt, "wb") as f:
            f.write(data)
        os.utime(txt, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    _same(_streams(in_dir), base)

    # a damaged sidecar (right stamp, truncated body) is ignored
    for txt, (data, st) in saved.items():
        os.utime(txt, ns=(st.st_atime_ns, st.st_mtime_ns))
    for s in side:
        body = open(s, "rb").read()
        with open(s, "wb") as f:
            f.write(body[:len(body) // 2])
    _same(_streams(in_dir), base)
