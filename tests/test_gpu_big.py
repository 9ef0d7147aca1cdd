"""GPU: the bench-scale generator (tests/biggen.py, inputs resident in HBM, on_device = 1) at a size the
oracle still finishes in seconds: HIP build vs oracle bit for bit, and the complete product path
(pag_process + pagh_traverse) against the compiled reference binary run on the text form of the same
workload."""
import ctypes as C
import os

import numpy as np
import pytest

import pagctl


def _oracle_on(w_host):
    lib = pagctl.oracle_lib()
    words = w_host.solid_words()
    g = lib.pago_create(words.ctypes.data, len(words), w_host.spec.k)
    inp = w_host.build_input()
    st = pagctl.BuildStats()
    assert lib.pago_process(g, C.byref(inp), C.byref(st)) == 0
    nn, npos, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
    lib.pago_csr_sizes(g, C.byref(nn), C.byref(npos), C.byref(ne))
    lib.pago_destroy(g)
    return st, (nn.value, npos.value, ne.value)


# (k, epsilon) corners of BASELINE configs[4]'s sweep (k in {12, 14, 16} x epsilon in {5, 10, 20}); k = 16 is outside the
# reference PIPELINE's range (its kmer_counter stops at 15, quirk Q13) but inside pagraph's, which is what is compared
@pytest.mark.gpu
# all nine (k, epsilon) points of the sweep
@pytest.mark.parametrize("k,eps,seed,n_reads", [(12, 10, 5, 4000), (14, 20, 5, 6000), (16, 5, 5, 6000), (12, 5, 5, 4000), (16, 20, 5, 6000),
                                                (12, 20, 5, 4000), (14, 5, 5, 6000), (14, 10, 5, 6000), (16, 10, 5, 6000)])
def test_device_resident_input_matches_oracle_and_reference(k, eps, seed, n_reads, workdir):
    import torch
    import bench
    import biggen
    hip, host = bench.load_libs()
    sp = biggen.BigSpec(seed=seed, ref_len=1_500_000, n_reads=n_reads, read_span=4000, k=k, eps=eps, ctg_len=300_000, gap_lo=300,
                        gap_hi=3000, rev_ctg_frac=0.3, threads=16, cov=2, solid_min_abundance=2, chunk_reads=512)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    assert g, hip.pag_last_error()
    st = pagctl.BuildStats()
    assert hip.pag_process(g, C.byref(inp), C.byref(st)) == 0, hip.pag_last_error()

    # (1) oracle on a host copy of the very same tensors
    ost, osizes = _oracle_on(w.clone_to("cpu"))
    assert st.counts() == ost.counts()
    assert (st.n_nodes, st.n_pos, st.n_uniq_edges) == osizes
    assert tuple(st.n_tuples) == tuple(ost.n_tuples) and tuple(st.n_edges) == tuple(ost.n_edges)

    # (2) whole product path vs the compiled reference on the text form
    ref_np = w.ref.cpu().numpy()
    ctg_codes = w.contig_codes()
    ctg_seqs, k1 = bench.host_seqs(ctg_codes)
    ref_seqs, k2 = bench.host_seqs([ref_np])
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    ours = str(workdir / "big_ours")
    os.makedirs(ours, exist_ok=True)
    ts = bench.TraverseStats()
    rc = host.pagh_traverse(g, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, sp.threads, sp.eps, 50,
                            ours.encode(), b"0_", 0, C.byref(ts))
    assert rc == 0, host.pagh_last_error()
    assert ts.n_path_nodes > 0
    print(f"k={k} eps={eps}: {ts.n_path_nodes} path nodes, {ts.n_path_bases} bases in emitted chains")
    # (2b) device walkers vs the host walk over the exported graph: identical files and checksum
    hostw = str(workdir / "big_hostwalk")
    os.makedirs(hostw, exist_ok=True)
    ts2 = bench.TraverseStats()
    hostwalk = pagctl.walk_test_lib().pagt_traverse_hostwalk
    hostwalk.argtypes = host.pagh_traverse.argtypes
    rc = hostwalk(g, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, sp.threads,
                                     sp.eps, 50, hostw.encode(), b"0_", 0, C.byref(ts2))
    assert rc == 0, host.pagh_last_error()
    # (2c) the walkers without speculation (every probe to its end before the choice): identical again
    exactd = str(workdir / "big_exact")
    os.makedirs(exactd, exist_ok=True)
    ts3 = bench.TraverseStats()
    os.environ["PAG_WALK_EXACT"] = "1"
    try:
        rc = host.pagh_traverse(g, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, sp.threads, sp.eps,
                                50, exactd.encode(), b"0_", 0, C.byref(ts3))
    finally:
        os.environ.pop("PAG_WALK_EXACT", None)
    assert rc == 0, host.pagh_last_error()
    host.pagh_release(g)
    hip.pag_destroy(g)
    assert (ts.n_path_nodes, ts.n_path_bases, ts.path_checksum) == (ts2.n_path_nodes, ts2.n_path_bases, ts2.path_checksum)
    assert (ts.n_path_nodes, ts.n_path_bases, ts.path_checksum) == (ts3.n_path_nodes, ts3.n_path_bases, ts3.path_checksum)
    for f in sorted(os.listdir(ours)):
        assert open(os.path.join(exactd, f), "rb").read() == open(os.path.join(ours, f), "rb").read(), f
    assert sorted(os.listdir(hostw)) == sorted(os.listdir(ours))
    for f in sorted(os.listdir(ours)):
        assert open(os.path.join(hostw, f), "rb").read() == open(os.path.join(ours, f), "rb").read(), f

    if not os.path.exists(os.path.join(pagctl.REF_DIR, "pagraph")):
        pytest.skip("oracle/_ref/pagraph not built")
    txt = str(workdir / "big_txt")
    w.write_text(txt)
    r = pagctl.run_reference(txt, str(workdir / "big_ref"), threads=sp.threads, eps=sp.eps, cov=sp.cov)
    assert r.returncode == 0
    ref_files = sorted(f for f in os.listdir(workdir / "big_ref") if f != "contig.txt")
    assert ref_files == sorted(os.listdir(ours))
    for f in ref_files:
        assert open(workdir / "big_ref" / f, "rb").read() == open(os.path.join(ours, f), "rb").read(), f


@pytest.mark.gpu
def test_device_walks_equal_host_walk_on_multi_round_contigs(workdir):
    """25 Mb / 50 k reads: two dozen 1 Mb contigs, several of which need re-seeded rounds (global-visit tables in play) —
    the size at which a window-refill overflow once changed 1 contig in 24 while every smaller test passed.  Device
    walkers (speculative and exact) against the host restatement of the reference's traversal: same fingerprint, same
    files."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "tests", "walk_check.py"), "--reads", "50000", "--ref-len", "25000000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ALL EQUAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_bench_line_contract(workdir):
    """bench.py on a small workload: one JSON line with the fields the driver reads, a roofline object for the sort kernel
    and a cpu_baseline object from the compiled reference (or a stated failure where oracle/_ref is absent)."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--reads", "3000",
                        "--ref-len", "3000000"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["scaling"] == "weak" and "workload" in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and "sample" in cb and "cores" in cb
