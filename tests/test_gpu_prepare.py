"""GPU: pag_prepare (the preparation stage on the device, csrc/hip/k_prepare.hip) against the host restatement of the same
stage (tests/harness/graph_input.cpp — the code the oracle-backed harness runs, pinned with it on the reference's golden
graph dumps), array for array: per-query lists in std::sort order, filters, flips, n_valid, the contig->reference map with
its multi-entry bases, both tables, the emission order."""
import ctypes as C
import os

import numpy as np
import pytest

import biggen
import goldens
import pagctl
import synth


def _host_arrays(view_ptr):
    v = C.cast(view_ptr, C.POINTER(biggen.PagBuildInput)).contents

    def host(ptr, n, dt):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).copy() if n else np.zeros(0, dt)
    return v, host


def _compare(inp: pagctl.LoadedInput, label):
    lib = pagctl.hip_lib()
    err = C.c_int()
    g = lib.pag_create(inp.kmer_words, inp.n_kmer_words, inp.k, 0, C.byref(err))
    assert g, lib.pag_last_error()
    try:
        d = pagctl._prepared_view(lib, g, inp)
        h, host = _host_arrays(inp.view)

        def dev(ptr, n, dt):
            return pagctl.device_bytes(ptr, n * np.dtype(dt).itemsize).view(dt)
        assert d.on_device == 1
        nr = h.reads.n_seqs
        assert d.reads.n_seqs == nr
        for name in ("n_threads", "n_ctgs", "n_refs", "eps", "cov_filter", "outer_sample", "topk_ctg", "topk_ref"):
            assert getattr(d, name) == getattr(h, name), f"{label}: {name}"
        np.testing.assert_array_equal(dev(d.emit_order, nr, "<u4"), host(h.emit_order, nr, "<u4"), err_msg=f"{label}: emit order")
        np.testing.assert_array_equal(dev(d.ctgs, d.n_ctgs, biggen.CTG_DTYPE), host(h.ctgs, h.n_ctgs, biggen.CTG_DTYPE), err_msg=f"{label}: contig table")
        np.testing.assert_array_equal(dev(d.refs, d.n_refs, biggen.REF_DTYPE), host(h.refs, h.n_refs, biggen.REF_DTYPE), err_msg=f"{label}: reference table")
        assert d.n_ctg_ent_off == h.n_ctg_ent_off and d.n_ctg_ent == h.n_ctg_ent, f"{label}: contig map sizes {d.n_ctg_ent_off} {d.n_ctg_ent} vs {h.n_ctg_ent_off} {h.n_ctg_ent}"
        np.testing.assert_array_equal(dev(d.ctg_ent_off, d.n_ctg_ent_off, "<u4"), host(h.ctg_ent_off, h.n_ctg_ent_off, "<u4"), err_msg=f"{label}: contig map offsets")
        np.testing.assert_array_equal(dev(d.ctg_ent, d.n_ctg_ent, "<u4"), host(h.ctg_ent, h.n_ctg_ent, "<u4"), err_msg=f"{label}: contig map entries")
        for which in ("read_to_ctg", "read_to_ref"):
            dd, hh = getattr(d, which), getattr(h, which)
            assert dd.n_aln == hh.n_aln, f"{label}: {which} records {dd.n_aln} vs {hh.n_aln}"
            np.testing.assert_array_equal(dev(dd.query_off, nr + 1, "<u8"), host(hh.query_off, nr + 1, "<u8"), err_msg=f"{label}: {which} query_off")
            a, b = dev(dd.aln, dd.n_aln, biggen.ALN_DTYPE), host(hh.aln, hh.n_aln, biggen.ALN_DTYPE)
            for f in biggen.ALN_DTYPE.names:
                np.testing.assert_array_equal(a[f], b[f], err_msg=f"{label}: {which}.{f}")
        return int(d.read_to_ctg.n_aln), int(d.read_to_ref.n_aln)
    finally:
        lib.pag_destroy(g)


@pytest.mark.gpu
@pytest.mark.parametrize("name", goldens.case_names())
def test_prepare_equals_host_restatement_on_goldens(name, workdir):
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    n_blocks = open(os.path.join(ind, "config.txt")).read().count("\n\n") or 1
    for block in range(n_blocks):
        inp = pagctl.LoadedInput(ind, threads=spec["threads"], eps=spec["epsilon"], cov=spec["cov"], block=block)
        try:
            _compare(inp, f"{name} block {block}")
        finally:
            inp.close()


CASES = {
    # reverse contigs, repeats (multi-entry contig bases), jittered read lengths
    "rev_multi": dict(seed=31, ref_len=24000, n_reads=400, read_len=1500, read_len_jitter=0.4, k=9, contigs=[(200, 9000, True), (9500, 23000, False)], repeats=3, extra_ctg_aln=True, dup_read_aln=True),
    # many alignments per read: lists longer than 16 go through the host's std::sort (introsort tie order)
    "long_lists": dict(seed=32, ref_len=12000, n_reads=60, read_len=900, k=8, contigs=[(100, 5800, False), (6000, 11800, True)], dup_alignments=24),
    "threads5": dict(seed=33, ref_len=16000, n_reads=333, read_len=700, k=8, contigs=[(50, 15000, False)]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_prepare_equals_host_restatement_on_fresh_inputs(case, workdir):
    kw = dict(CASES[case])
    dup = kw.pop("dup_alignments", 0)
    d = str(workdir / ("prep_" + case))
    synth.generate(synth.Spec(**kw), d)
    if dup:
        # every read->contig / read->reference record `dup` times over with equal and with perturbed scores: lists of up to
        # dup x alignments per read, ties included
        for fn in ("0.ctg.ref", "0.ref.ref"):
            path = os.path.join(d, fn)
            lines = open(path).read().splitlines()
            out = []
            for i in range(0, len(lines) - 2, 3):
                hdr = lines[i].split()
                for j in range(dup):
                    h2 = list(hdr)
                    h2[3] = str(int(hdr[3]) + (j % 3))  # three score classes: many ties
                    out += [" ".join(h2), lines[i + 1], lines[i + 2]]
            open(path, "w").write("\n".join(out) + "\n")
    threads = 5 if case == "threads5" else 16
    inp = pagctl.LoadedInput(d, threads=threads, eps=10, cov=2)
    try:
        n1, n2 = _compare(inp, case)
        assert n1 > 0 and n2 > 0
        if dup:
            assert n2 >= 17 * kw["n_reads"] // 2, "the duplicated records did not make long lists"
        # and the graph built from the device-prepared input equals the oracle's on the host-prepared one
        pagctl.compare_results(pagctl.run_hip(inp, streams=True), pagctl.run_oracle(inp, streams=True), label=case)
    finally:
        inp.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k,threads", [(12, 16), (14, 7)])
def test_prepare_of_the_bench_generator_equals_its_own_digest(k, threads):
    """bench.py's step starts from the generator's records in PARSER form (BigWorkload.raw_input: header intervals, strand
    column, score) and lets pag_prepare derive what the generator also knows from its simulation truth (build_input:
    q_start / t_start / n_valid / flags, the per-read lists, the contig->reference map): the two must agree array for
    array — and so does the graph built from either."""
    import torch
    sp = biggen.BigSpec(seed=5, ref_len=1_500_000, n_reads=2500, read_span=6000, k=k, ctg_len=200_000, eps=10, cov=2, threads=threads,
                        chunk_reads=1000)
    w = biggen.BigWorkload(sp, device="cuda:0")
    torch.cuda.synchronize()
    lib = pagctl.hip_lib()
    lib.pag_create_from_bitmap.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.pag_create_from_bitmap.restype = C.c_void_p
    lib.pag_prepare.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    err = C.c_int()
    g = lib.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    assert g
    try:
        raw = w.raw_input()
        d = biggen.PagBuildInput()
        assert lib.pag_prepare(g, C.byref(raw), C.byref(d)) == 0, lib.pag_last_error()
        h = w.build_input()

        def dev(ptr, n, dt):
            return pagctl.device_bytes(ptr, n * np.dtype(dt).itemsize).view(dt)
        nr = sp.n_reads
        np.testing.assert_array_equal(dev(d.emit_order, nr, "<u4"), dev(h.emit_order, nr, "<u4"))
        np.testing.assert_array_equal(dev(d.ctgs, d.n_ctgs, biggen.CTG_DTYPE), dev(h.ctgs, h.n_ctgs, biggen.CTG_DTYPE))
        np.testing.assert_array_equal(dev(d.refs, 1, biggen.REF_DTYPE), dev(h.refs, 1, biggen.REF_DTYPE))
        n_off = int(h.n_ctg_ent_off)
        assert d.n_ctg_ent_off == n_off
        np.testing.assert_array_equal(dev(d.ctg_ent_off, n_off, "<u4"), dev(h.ctg_ent_off, n_off, "<u4"))
        n_ent = int(dev(h.ctg_ent_off, n_off, "<u4")[-1])
        np.testing.assert_array_equal(dev(d.ctg_ent, n_ent, "<u4"), dev(h.ctg_ent, n_ent, "<u4"))
        for which in ("read_to_ctg", "read_to_ref"):
            dd, hh = getattr(d, which), getattr(h, which)
            assert dd.n_aln == hh.n_aln, which
            np.testing.assert_array_equal(dev(dd.query_off, nr + 1, "<u8"), dev(hh.query_off, nr + 1, "<u8"), err_msg=which)
            a, b = dev(dd.aln, dd.n_aln, biggen.ALN_DTYPE), dev(hh.aln, hh.n_aln, biggen.ALN_DTYPE)
            for f in biggen.ALN_DTYPE.names:
                np.testing.assert_array_equal(a[f], b[f], err_msg=f"{which}.{f}")
        st1, st2 = pagctl.BuildStats(), pagctl.BuildStats()
        assert lib.pag_process(g, C.byref(d), C.byref(st1)) == 0, lib.pag_last_error()
        assert lib.pag_process(g, C.byref(h), C.byref(st2)) == 0, lib.pag_last_error()
        assert st1.counts() == st2.counts() and (st1.n_nodes, st1.n_pos, st1.n_uniq_edges) == (st2.n_nodes, st2.n_pos, st2.n_uniq_edges)
        assert st1.n_pos > 100000
    finally:
        lib.pag_destroy(g)
