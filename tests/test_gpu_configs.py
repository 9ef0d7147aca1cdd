"""GPU: the BASELINE.json configurations themselves.

configs[0]  10 k x 1 kb reads vs a 5 Mb reference, k = 14, `-t 16`: the TEXT form through the drop-in executable
            (bin/pagraph -t 16) against the compiled reference binary run under the thread-serialising shim; every output
            file byte for byte.  (Solid set = read k-mers of abundance >= 1: with 2x coverage the kmer_counter rule would
            declare all 4^14 codes solid, and the reference's 152-byte-per-k-mer node table then takes eight minutes.)
configs[1]  100 k x 10 kb vs 50 Mb at FULL size: pag_process on the device against the C oracle on a host copy of the
            same arrays — the six count lines, stream lengths, CSR sizes and EVERY CSR array — and the device walkers
            against the host restatement of the reference's traversal (tests/walk_check.py).  The oracle needs ~10 minutes
            and ~30 GB of host memory at this size, so the test only runs with PAG_C2_FULL=1 (gpurun it on its own; the log
            of the last run is kept under profiles/).  Without the variable a 1/10 scale instance runs (5 Mb, 10 k reads of
            10 kb: the same coverage and read length)."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np
import pytest

import pagctl
import synth

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pagraph")


@pytest.mark.gpu
def test_config0_text_inputs_through_the_executable_equal_the_reference_binary(workdir):
    ref_bin = os.path.join(pagctl.REF_DIR, "pagraph")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/pagraph was not built (needs /root/reference at build time)")
    import biggen
    sp = biggen.BigSpec(seed=1, ref_len=5_000_000, n_reads=10_000, read_span=1_000, k=14, eps=10, cov=2, threads=16,
                        ctg_len=1_000_000, solid_min_abundance=1)
    w = biggen.BigWorkload(sp, device="cuda")
    d = str(workdir / "c1")
    w.write_text(d + "/in")
    del w
    r = pagctl.run_reference(d + "/in", d + "/ref", threads=16, eps=10, cov=2)
    assert r.returncode == 0
    os.makedirs(d + "/ours", exist_ok=True)
    o = subprocess.run(synth.pagraph_argv(EXE, d + "/in", d + "/ours", threads=16, epsilon=10, cov=2), capture_output=True, text=True)
    assert o.returncode == 0, o.stderr[-2000:]
    assert "HIP gfx950" in o.stdout
    # the six count lines of PositionProcessor::process are printed by both programs
    pick = lambda txt: [ln.strip() for ln in txt.splitlines() if ln.strip().startswith(("merge edge", "total pos", "merge pos"))]  # noqa: E731
    assert pick(o.stdout) == pick(r.stdout) and len(pick(o.stdout)) == 6
    fr, fo = sorted(os.listdir(d + "/ref")), sorted(os.listdir(d + "/ours"))
    assert fr == fo and len(fr) >= 3
    for f in fr:
        a, b = open(f"{d}/ref/{f}", "rb").read(), open(f"{d}/ours/{f}", "rb").read()
        if f == "contig.txt":
            a, b = sorted(a.split()), sorted(b.split())
        assert a == b, f"{f} differs from the reference binary's output"


def _csr_arrays(lib, prefix, g, sizes):
    nn, npos, ne = sizes
    arrs = {"node_code": np.zeros(nn + 1, np.uint32), "pos_off": np.zeros(nn + 1, np.uint64), "pos_ctg": np.zeros(npos + 1, np.uint32),
            "pos_ref": np.zeros(npos + 1, np.uint32), "pos_cnt": np.zeros(npos + 1, np.uint16), "edge_off": np.zeros(nn + 1, np.uint64),
            "edge_to": np.zeros(ne + 1, np.uint32), "edge_step": np.zeros(ne + 1, np.int32)}
    csr = pagctl.Csr(nn, npos, ne, *[arrs[k].ctypes.data for k in ("node_code", "pos_off", "pos_ctg", "pos_ref", "pos_cnt", "edge_off",
                                                                  "edge_to", "edge_step")])
    rc = getattr(lib, prefix + "_export_csr")(g, C.byref(csr))
    assert rc == 0
    return arrs


@pytest.mark.gpu
def test_config1_graph_build_equals_oracle_on_every_csr_array(workdir):
    import torch
    import bench
    import biggen
    full = os.environ.get("PAG_C2_FULL") == "1"
    hip, host = bench.load_libs()
    sp = biggen.BigSpec(seed=2, k=14, eps=10, cov=2, threads=16) if full else \
        biggen.BigSpec(seed=2, ref_len=5_000_000, n_reads=10_000, k=14, eps=10, cov=2, threads=16, solid_min_abundance=3)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    assert g, hip.pag_last_error()
    st = pagctl.BuildStats()
    assert hip.pag_process(g, C.byref(inp), C.byref(st)) == 0, hip.pag_last_error()
    sizes = (st.n_nodes, st.n_pos, st.n_uniq_edges)
    t0 = time.time()
    ours = _csr_arrays(hip, "pag", g, sizes)
    print(f"[c2] device build: {w.n_bases} read bases, tuples {tuple(st.n_tuples)}, edges {tuple(st.n_edges)}, csr {sizes}; export {time.time() - t0:.0f} s", flush=True)
    hip.pag_destroy(g)

    wh = w.clone_to("cpu")
    del w
    torch.cuda.empty_cache()
    lib = pagctl.oracle_lib()
    words = wh.solid_words()
    t0 = time.time()
    og = lib.pago_create(words.ctypes.data, len(words), sp.k)
    del words
    oinp = wh.build_input()
    ost = pagctl.BuildStats()
    assert lib.pago_process(og, C.byref(oinp), C.byref(ost)) == 0
    print(f"[c2] oracle build {time.time() - t0:.0f} s", flush=True)
    assert st.counts() == ost.counts()
    assert tuple(st.n_tuples) == tuple(ost.n_tuples) and tuple(st.n_edges) == tuple(ost.n_edges)
    assert sizes == (ost.n_nodes, ost.n_pos, ost.n_uniq_edges)
    theirs = _csr_arrays(lib, "pago", og, sizes)
    lib.pago_destroy(og)
    for kk in ours:
        assert np.array_equal(ours[kk], theirs[kk]), f"CSR array {kk} differs from the oracle's"
    print(f"[c2] all 8 CSR arrays equal ({'FULL config 1' if full else '1/10 scale'})", flush=True)


@pytest.mark.gpu
def test_config1_device_walks_equal_host_walk_at_full_size():
    if os.environ.get("PAG_C2_FULL") != "1":
        pytest.skip("full-size configs[1] traversal check (~3 min of host walking): set PAG_C2_FULL=1")
    r = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "tests", "walk_check.py"), "--reads", "100000", "--ref-len", "50000000"],
                       capture_output=True, text=True, timeout=3000)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "ALL EQUAL" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
