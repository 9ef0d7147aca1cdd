"""GPU parity of the graph build: HIP library vs the C oracle, through the C ABI, on seeded inputs.

Bit-exact (integer work): emitted tuple/edge streams in canonical order, the reference's six count
lines, and the finished CSR (k-mer codes, clustered positions, u16 counts, unique edges)."""
import os
import subprocess

import numpy as np
import pytest

import pagctl
import synth

CASES = {
    # name: (Spec kwargs, threads, eps, cov)
    "small_fwd": (dict(seed=11, ref_len=6000, n_reads=60, read_len=700, k=8,
                       contigs=[(100, 2800, False), (3100, 5900, False)]), 1, 10, 2),
    "rev_ctg_t4": (dict(seed=12, ref_len=12000, n_reads=200, read_len=900, k=9,
                        contigs=[(200, 5600, False), (5900, 11800, True)]), 4, 10, 2),
    "multi_entry": (dict(seed=13, ref_len=15000, n_reads=250, read_len=1200, k=9,
                         contigs=[(300, 7000, False), (7300, 14700, True)], extra_ctg_aln=True,
                         dup_read_aln=True), 16, 5, 1),
    "long_reads_k12": (dict(seed=14, ref_len=40000, n_reads=120, read_len=5000, read_len_jitter=0.5, k=12,
                            contigs=[(500, 19000, False), (19600, 39500, False)], repeats=3), 1, 20, 2),
    "outer_tiles": (dict(seed=15, ref_len=30000, n_reads=80, read_len=3000, k=10, solid_min_abundance=2), 8, 10, 0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_build_matches_oracle(name, workdir):
    kw, threads, eps, cov = CASES[name]
    d = str(workdir / name)
    synth.generate(synth.Spec(**kw), d)
    inp = pagctl.LoadedInput(d, threads=threads, eps=eps, cov=cov)
    try:
        ora = pagctl.run_oracle(inp, streams=True)
        hip = pagctl.run_hip(inp, streams=True)
        pagctl.compare_results(hip, ora, label=name)
        assert hip["stats"].n_pos == len(hip["csr"]["pos_ctg"])
        assert hip["stats"].n_pos > 0
    finally:
        inp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("segments,seed", [(2000, 1), (20000, 2), (300000, 3)])
def test_segment_kernels_match_sequential_restatement(segments, seed, wide):
    """K3/K4 alone (both widths of the short path, segments on both sides of every limit of the on-chip long paths, tile halos) against
    a sequential greedy scan / sort-unique."""
    import subprocess
    exe = os.path.join(pagctl.ROOT, "tests", "harness", "bin", "seg_kernels_test")
    assert os.path.exists(exe), "run `make harness`"
    r = subprocess.run([exe, str(segments), str(seed), str(wide)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n,bits", [(1, 5), (64, 6), (5000, 13), (5120, 14), (5121, 14), (7000, 32), (100000, 21), (1000003, 25),
                                    (10000000, 28)])
def test_radix_sort_is_a_stable_sort(n, bits):
    """K2 alone (tests/harness/sort_bench.hip): random keys of `bits` bits, payload = input index; the result must equal
    std::stable_sort's — full and partial tiles, one block and many, every digit width the pass splitter produces."""
    exe = os.path.join(pagctl.ROOT, "tests", "harness", "bin", "sort_bench")
    if not os.path.exists(exe):
        pytest.skip("tests/harness/bin/sort_bench not built")
    r = subprocess.run([exe, str(n), str(bits)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "check: 0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("n,offset", [(1, 0), (3, 0), (4, 0), (1023, 0), (4096, 0), (4097, 0), (4099, 1), (65536, 2), (1000003, 0), (1000003, 3),
                                      (11214848, 0), (30000001, 4)])
def test_exclusive_scan_u32_to_u64(n, offset):
    """util.hip's scan (16-byte loads and stores where the views allow them; used by every compaction and by the sort's
    histogram prefix): full and partial tiles, views that are not 16-byte aligned, sums beyond 2^32, nothing written past
    the end — against a sequential sum (tests/harness/sort_bench.hip, scan mode)."""
    exe = os.path.join(pagctl.ROOT, "tests", "harness", "bin", "sort_bench")
    if not os.path.exists(exe):
        pytest.skip("tests/harness/bin/sort_bench not built")
    r = subprocess.run([exe, "scan", str(n), str(offset)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "check: 0 mismatches" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
