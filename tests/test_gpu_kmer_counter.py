"""GPU: the device k-mer counter (pag_kmer_count) — bit-exact against the oracle through the C ABI, the
`kmer_counter` executable byte-identical to the reference's golden files and to the reference binary on fresh reads,
and the bench-scale read set against the generator's own solid set."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import kmer_cases
import pagctl
from biggen import PagSeqs
from test_kmer_counter_oracle import GOLD, oracle_file_words

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "kmer_counter")


class KmerCountResult(C.Structure):
    _fields_ = [("min_abundance", C.c_uint64), ("n_solid", C.c_uint64), ("n_kmers_counted", C.c_uint64),
                ("ms_count", C.c_double), ("ms_select", C.c_double)]


def hip_count(rs, on_device, k, threshold, bitmap_ptr, bitmap_on_device):
    lib = pagctl.hip_lib()
    lib.pag_kmer_count.argtypes = [C.POINTER(PagSeqs), C.c_int, C.c_uint32, C.c_double, C.c_int, C.c_void_p, C.c_int,
                                   C.POINTER(KmerCountResult)]
    lib.pag_kmer_count.restype = C.c_int
    res = KmerCountResult()
    rc = lib.pag_kmer_count(C.byref(rs), on_device, k, threshold, 0, bitmap_ptr, bitmap_on_device, C.byref(res))
    assert rc == 0, lib.pag_last_error().decode()
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("seed,k,threshold,n,hi", [(21, 3, 0.2, 50, 80), (22, 8, 0.2, 400, 900), (23, 11, 0.05, 3000, 2500),
                                                  (24, 2, 0.2, 10, 40), (25, 9, 0.0, 500, 700), (26, 10, -1.0, 100, 300),
                                                  (27, 6, 0.2, 5, 4)])
@pytest.mark.parametrize("slices", [None, "4"])  # (None: the library's own choice; "4": the table filled a quarter of the code range per launch)
def test_hip_counter_matches_oracle(seed, k, threshold, n, hi, slices, monkeypatch):
    if slices:
        monkeypatch.setenv("PAG_KC_SLICES", slices)
    case = dict(seed=seed, n=n, lo=1, hi=hi, k=k, threshold=threshold, threads=1, fmt="fastq", alphabet="ACGTNacgt")
    seqs = kmer_cases.sequences(case)
    _, mn, want = oracle_file_words(seqs, k, threshold, 1)
    offs, lens, packed = kmer_cases.pack(seqs)
    rs = PagSeqs(len(seqs), offs.ctypes.data, lens.ctypes.data, packed.ctypes.data, len(packed) - 64)
    got = np.zeros(len(want), np.uint32)
    res = hip_count(rs, 0, k, threshold, got.ctypes.data, 0)
    assert res.min_abundance == mn
    nw = (4 ** k + 31) // 32
    assert (got[:nw] == want[:nw]).all()
    assert res.n_solid == int(sum(bin(int(x)).count("1") for x in want[:nw]))
    assert res.n_kmers_counted == sum(max(0, len(s) - k + 1) for s in seqs)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(kmer_cases.CASES))
def test_kmer_counter_executable_matches_golden(name, tmp_path):
    case = kmer_cases.CASES[name]
    reads = os.path.join(GOLD, name, "reads." + case["fmt"])
    out = str(tmp_path / "out.bin")
    r = subprocess.run([EXE, "-t", str(case["threads"]), "-i", reads, "-o", out, "-k", str(case["k"]), "-m", repr(case["threshold"])],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert open(out, "rb").read() == open(os.path.join(GOLD, name, "expected.bin"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("k,threads,fmt", [(10, 16, "fastq"), (12, 3, "fasta"), (9, 0, "fastq")])  # (-t 0: header-only file)
def test_kmer_counter_executable_matches_reference_binary(k, threads, fmt, tmp_path):
    ref = os.path.join(pagctl.REF_DIR, "kmer_counter")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/kmer_counter was not built (needs /root/reference at build time)")
    case = dict(seed=100 + k, n=2000, lo=50, hi=3000, k=k, threshold=0.2, threads=threads, fmt=fmt, alphabet="ACGT")
    reads = str(tmp_path / ("r." + fmt))
    kmer_cases.write_reads(case, reads)
    a, b = str(tmp_path / "ours.bin"), str(tmp_path / "ref.bin")
    for exe, out in ((EXE, a), (ref, b)):
        r = subprocess.run([exe, "--thread", str(threads), "--in=" + reads, "-o" + out, "-k", str(k)], capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.gpu
def test_kmer_counter_usage_and_errors(tmp_path):
    assert subprocess.run([EXE], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "-h"], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "--nope"], capture_output=True).returncode == 1
    assert subprocess.run([EXE, "-k", "x"], capture_output=True).returncode == 1


@pytest.mark.gpu
def test_hip_counter_at_bench_scale_on_device_reads():
    """Device-resident reads of a mid-size workload: the counter reproduces the generator's histogram rule and set."""
    import torch

    import biggen
    w = biggen.BigWorkload(biggen.BigSpec(seed=5, ref_len=4_000_000, n_reads=8000, read_span=10_000, k=12), device="cuda")
    inp = w.build_input()
    nw = (4 ** 12) // 32
    bitmap = torch.zeros(nw + 8, dtype=torch.int32, device="cuda")
    res = hip_count(inp.reads, 1, 12, w.spec.solid_threshold, bitmap.data_ptr(), 1)
    torch.cuda.synchronize()
    assert res.min_abundance == w.min_abundance
    got_bits = bitmap[:nw].to(torch.int64) & 0xFFFFFFFF
    got = ((got_bits.unsqueeze(1) >> torch.arange(32, device="cuda")) & 1).bool().view(-1)
    diff = torch.nonzero(got != w.solid_mask).squeeze(1).tolist()
    assert diff in ([], [12]), diff[:10]  # code k is forced into the generator's set (file header quirk Q1)
