"""k-mer counter, CPU side: the oracle's restatement of kmer_counter against the golden files written
by the compiled reference (tests/golden/kmer_counter/make_golden.py), and against the reference binary itself where
it is present (oracle/_ref/kmer_counter)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kmer_cases
import pagctl
from biggen import PagSeqs

GOLD = os.path.join(pagctl.ROOT, "tests", "golden", "kmer_counter")


def oracle_file_words(seqs, k, threshold, threads):
    lib = pagctl.oracle_lib()
    lib.pago_kmer_count.argtypes = [C.POINTER(PagSeqs), C.c_uint32, C.c_double, C.POINTER(C.c_uint64), C.c_void_p]
    lib.pago_kmer_count.restype = C.c_int
    lib.pago_kmer_file_words.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64]
    lib.pago_kmer_file_words.restype = C.c_uint64
    offs, lens, packed = kmer_cases.pack(seqs)
    rs = PagSeqs(len(seqs), offs.ctypes.data, lens.ctypes.data, packed.ctypes.data, len(packed) - 64)
    bitmap = np.zeros((4 ** k + 31) // 32 + 1, np.uint32)
    mn = C.c_uint64()
    assert lib.pago_kmer_count(C.byref(rs), k, threshold, C.byref(mn), bitmap.ctypes.data) == 0
    n = lib.pago_kmer_file_words(bitmap.ctypes.data, k, threads, None, 0)
    words = np.zeros(n, np.uint64)
    lib.pago_kmer_file_words(bitmap.ctypes.data, k, threads, words.ctypes.data, n)
    return words, int(mn.value), bitmap


@pytest.mark.parametrize("name", list(kmer_cases.CASES))
def test_oracle_matches_reference_golden(name):
    case = kmer_cases.CASES[name]
    words, _, _ = oracle_file_words(kmer_cases.sequences(case), case["k"], case["threshold"], case["threads"])
    want = np.fromfile(os.path.join(GOLD, name, "expected.bin"), dtype=np.uint64)
    assert len(words) == len(want) and (words == want).all()


@pytest.mark.parametrize("seed,k,threads,threshold", [(11, 5, 2, 0.2), (12, 7, 16, 0.35), (13, 3, 5, 0.2)])
def test_oracle_matches_reference_binary_on_fresh_reads(seed, k, threads, threshold, tmp_path):
    ref = os.path.join(pagctl.REF_DIR, "kmer_counter")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/kmer_counter was not built (needs /root/reference at build time)")
    case = dict(seed=seed, n=80, lo=2, hi=300, k=k, threshold=threshold, threads=threads, fmt="fastq", alphabet="ACGTN")
    reads = str(tmp_path / "r.fastq")
    seqs = kmer_cases.write_reads(case, reads)
    out = str(tmp_path / "ref.bin")
    subprocess.run([ref, "-t", str(threads), "-i", reads, "-o", out, "-k", str(k), "-m", repr(threshold)], check=True, capture_output=True)
    words, _, _ = oracle_file_words(seqs, k, threshold, threads)
    want = np.fromfile(out, dtype=np.uint64)
    assert len(words) == len(want) and (words == want).all()
