"""CPU: pins the C oracle (oracle/pag_oracle.c) and the product's host pipeline (parsers, GraphInput,
traversal, writers) against golden vectors produced by the COMPILED REFERENCE (tests/golden/)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import goldens
import pagctl
import synth

BIN = os.path.join(pagctl.ROOT, "tests", "harness", "bin")


@pytest.fixture(scope="module", autouse=True)
def _built():
    subprocess.run(["make", "-C", pagctl.ROOT, "harness"], check=True, capture_output=True)


def _flags(spec):
    return ["-t", str(spec["threads"]), "--epsilon", str(spec["epsilon"]), "-v", str(spec["cov"])]


@pytest.mark.parametrize("name", goldens.case_names())
def test_oracle_graph_equals_reference_graph(name, workdir):
    """complete graph (positions, u16 counts, unique edges, the six count lines): byte-identical"""
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    out = str(workdir / name / "graph")
    os.makedirs(out, exist_ok=True)
    subprocess.run([os.path.join(BIN, "oracle_graph_dump"), "-k", ind + "/kmer.bin", "-c", ind + "/ctg.fasta", "-R",
                    ind + "/ref.fasta", "-p", ind, "-a", ind + "/aln", "-o", out] + _flags(spec), check=True)
    dumps = sorted((f for f in os.listdir(out) if f.endswith(".graph.txt")), key=lambda f: int(f.split(".")[0]))
    got = b"".join(open(os.path.join(out, f), "rb").read() for f in dumps)  # (one dump per config block)
    assert got == goldens.golden_graph(name)


@pytest.mark.parametrize("name", goldens.case_names())
def test_host_pipeline_outputs_equal_reference_outputs(name, workdir):
    """path dumps, FASTA, .con, .help, contig.txt of the driver (oracle backend): byte-identical"""
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    out = str(workdir / name / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(os.path.join(BIN, "pagraph_oracle"), ind, out, threads=spec["threads"],
                              epsilon=spec["epsilon"], cov=spec["cov"])
    r = subprocess.run(argv, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    goldens.compare_out_dir(name, out)


@pytest.mark.parametrize("blocks", [None, "0,2,3", "1,3"])
def test_driver_parses_the_next_block_ahead(blocks, workdir):
    """host logic of pagraph_driver.cpp (the loop over config blocks), with the oracle as the backend"""
    goldens.check_blocks_parsed_ahead(os.path.join(BIN, "pagraph_oracle"), blocks, str(workdir / ("ahead_" + (blocks or "all").replace(",", "_"))))


def test_function_level_goldens():
    lib = pagctl.oracle_lib()
    # k-mer codec
    for line in open(os.path.join(goldens.GOLDEN, "func_kmer.txt")):
        parts = line.split()
        seq, k, n = parts[0], int(parts[1]), int(parts[2])
        out = np.zeros(max(1, len(seq)), np.uint64)
        got = lib.pago_kmer_codes(seq.encode(), len(seq), k, out.ctypes.data)
        assert got == n
        assert [int(x.split(":")[0]) for x in parts[3:]] == out[:n].tolist()
    # predicates (checkPosition / isEdgeSimilar / isPosSimilar truth tables)
    rows = np.loadtxt(gzip.open(os.path.join(goldens.GOLDEN, "func_predicate.txt.gz")), dtype=str)
    assert len(rows) > 30000
    for a1, a2, b1, b2, dist, dev, grade, es, ps in rows[::7]:
        a1, a2, b1, b2, dist, dev = (int(x) for x in (a1, a2, b1, b2, dist, dev))
        assert lib.pago_check_position(a1, a2, b1, b2, dist, dev, 0.15) == int(grade)
        e = lib.pago_edge_similar(a1, a2, b1, b2, dist, dev, 0.15)
        assert f"{e & 1}{(e >> 1) & 1}" == es
