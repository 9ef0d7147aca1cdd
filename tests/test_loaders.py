"""CPU: the thread-pool sequence loaders (mapped file, parallel line index, FASTA records packed by base ranges, FASTQ records
packed per read) against the sequential getline loops they replace (PAGH_SEQUENTIAL_LOADERS=1), on inputs with the
corner cases the reference's reader has: text before the first header, empty lines, CRLF, no final newline, lower case,
other letters, duplicate names, a lone header, an empty file."""
import ctypes as C
import os
import subprocess
import sys

import pytest

import pagctl

CASES = {
    "plain.fa": b">a desc\nACGTACGTAC\nGGTT\n>b\nTTTT\n",
    "prefix.fa": b";comment first\nGATTACA\n>first rec\nACGT\nAC\n>second\n\nGG\n\n",
    "crlf.fa": b">r1\r\nACGT\r\nTTGA\r\n>r2\r\nCC\r\n",
    "nonl.fa": b">only\nACGTNNNNacgtRYKM",
    "dup.fa": b">same\nAAAA\n>same\nCCCCCC\n>other\n>empty_next\n\n>last\nT",
    "long.fa": b">big one\n" + b"\n".join((b"ACGTTGCAAGCT" * 9)[:100] for _ in range(5000)) + b"\nACG\n>tail\nA\n",
    "reads.fq": b"".join(b"@r%d extra\n%s\n+\n%s\n" % (i, (b"ACGTTGCA" * (i % 7 + 1))[: 5 + i % 11], b"I" * (5 + i % 11)) for i in range(300)),
    "partial.fq": b"@a\nACGT\n+\nIIII\n@b\nGG\n+\n",
    "empty.fa": b"",
}


def digest(lib, path):
    out = (C.c_uint64 * 3)()
    lib.pagt_seqdb_digest(path.encode(), out)
    return tuple(out)


@pytest.mark.parametrize("name", sorted(CASES))
def test_parallel_loader_equals_sequential(name, tmp_path):
    path = str(tmp_path / name)
    with open(path, "wb") as f:
        f.write(CASES[name])
    lib = pagctl.test_lib()
    lib.pagt_seqdb_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]
    lib.pagt_seqdb_digest.restype = None
    fast = digest(lib, path)
    # the sequential loops in a child process (the switch is read from the environment)
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import pagctl; lib = pagctl.test_lib(); "
            "lib.pagt_seqdb_digest.argtypes = [C.c_char_p, C.POINTER(C.c_uint64)]; lib.pagt_seqdb_digest.restype = None; "
            "out = (C.c_uint64 * 3)(); lib.pagt_seqdb_digest(%r.encode(), out); print(*out)") % (os.path.dirname(os.path.abspath(__file__)), path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PAGH_SEQUENTIAL_LOADERS="1"))
    assert r.returncode == 0, r.stderr
    slow = tuple(int(x) for x in r.stdout.split())
    assert fast == slow


def test_aln_record_filter_keeps_headers_and_own_columns(tmp_path):
    """a rank of a sharded build parses the columns of ITS reads' alignments only (aln_db.hpp, AlnRecordFilter): every
    record keeps its header, the wanted ones the column classes of the unfiltered parse, the others none"""
    import goldens
    ind = goldens.materialize_inputs("join_rev_t16", str(tmp_path / "in"))
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import pagctl; lib = pagctl.test_lib(); "
            "lib.pagt_aln_filter_check.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]; out = (C.c_uint64 * 5)(); "
            "rc = lib.pagt_aln_filter_check(%r.encode(), 4, 1, out); print(rc, *out)") % (os.path.dirname(os.path.abspath(__file__)), os.path.join(ind, "0.ref.ref"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rc, n, kept, ok, words_part, words_full = (int(x) for x in r.stdout.split())
    assert rc == 0 and ok == 1
    assert 0 < kept < n and abs(kept - n / 4) < n / 8
    assert words_part < 0.45 * words_full


@pytest.mark.parametrize("world,threads", [(4, 16), (2, 1), (8, 5)])
def test_reads_outside_a_rank_s_stretch_keep_name_and_length_only(tmp_path, world, threads):
    """a rank of a sharded build packs the bases of ITS stretch of the emission order only (seq_db.hpp, SeqDb::PackWindow): the
    other reads keep name and length and share one stretch of zero bytes; the windows of all ranks partition the reads"""
    import goldens
    ind = goldens.materialize_inputs("join_rev_t16", str(tmp_path / "in"))
    fq = [f for f in sorted(os.listdir(ind)) if f.endswith((".fastq", ".fq"))]
    assert fq, os.listdir(ind)
    total, n_reads = 0, None
    for rank in range(world):
        code = ("import ctypes as C, sys; sys.path.insert(0, %r); import pagctl; lib = pagctl.test_lib(); "
                "lib.pagt_seq_window_check.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_uint64)]; out = (C.c_uint64 * 5)(); "
                "rc = lib.pagt_seq_window_check(%r.encode(), %d, %d, %d, out); print(rc, *out)") % (
                    os.path.dirname(os.path.abspath(__file__)), os.path.join(ind, fq[0]), rank, world, threads)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rc, n, kept, ok, bytes_part, bytes_full = (int(x) for x in r.stdout.split())
        assert rc == 0 and ok == 1, (rank, r.stdout, r.stderr)
        assert abs(kept - n / world) <= 1
        assert bytes_part < bytes_full * (1.0 / world + 0.2)
        total += kept
        n_reads = n
    assert total == n_reads
