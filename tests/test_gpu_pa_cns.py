"""GPU: pa_cns with its graph stage ON THE DEVICE (the default backend; pag_cns_consensus, csrc/hip/k_cns.hip: one thread per
backbone part running csrc/hip/cns_graph.hpp — AlnGraphBoost.cpp's addAln / mergeNodes / bestPath / consensus on flat arrays
with linked edge lists) against the reference: the five golden outputs written by the compiled reference, the reference
binary itself on the same seeded inputs (incl. the pipeline's settings at ~190x coverage), and a backbone of 60 parts
(several waves of threads, parts of very different depth) against the host restatement."""
import os
import subprocess

import pytest

import cns_cases
import pagctl

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pa_cns")
REF = os.path.join(pagctl.REF_DIR, "pa_cns")
GOLD = os.path.join(pagctl.ROOT, "tests", "golden", "pa_cns")


def run(exe, d, out, case, threads=4, backend=None):
    env = dict(os.environ)
    env.pop("PA_CNS_BACKEND", None)
    if backend:
        env["PA_CNS_BACKEND"] = backend
    return subprocess.run(cns_cases.argv(exe, d, out, case, threads), capture_output=True, text=True, timeout=900, env=env)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(cns_cases.CASES))
def test_device_pa_cns_matches_golden_and_reference_binary(name, tmp_path):
    case = cns_cases.CASES[name]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours.fasta"), case, backend="hip")
    assert r.returncode == 0, r.stderr[-800:]
    ours = open(tmp_path / "ours.fasta", "rb").read()
    assert ours == open(os.path.join(GOLD, name + ".fasta"), "rb").read()
    assert r.stdout == open(os.path.join(GOLD, name + ".stdout")).read()
    if os.path.exists(REF):
        r2 = run(REF, d, str(tmp_path / "ref.fasta"), case, threads=5)
        assert r2.returncode == 0 and r2.stdout == r.stdout
        assert ours == open(tmp_path / "ref.fasta", "rb").read()


@pytest.mark.gpu
def test_device_pa_cns_at_pipeline_settings(tmp_path):
    """part length 5000, top 3000, alpha 250 at ~190x: the depth a graph has in the pipeline (tens of thousands of insertion
    vertices per part, in-lists of dozens of edges)"""
    case = cns_cases.DEEP_CASE
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=16, backend="hip")
    assert r.returncode == 0, r.stderr[-800:]
    want_exe, want_be = (REF, None) if os.path.exists(REF) else (EXE, "host")
    r2 = run(want_exe, d, str(tmp_path / "want.fasta"), case, threads=16, backend=want_be)
    assert r2.returncode == 0 and r2.stdout == r.stdout
    assert open(tmp_path / "ours.fasta", "rb").read() == open(tmp_path / "want.fasta", "rb").read()


@pytest.mark.gpu
def test_device_pa_cns_many_parts(tmp_path):
    case = dict(seed=11, backbone=30000, n_reads=900, read_len=1200, part=500, top_k=3000, alpha=250, score_classes=3)
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=8, backend="hip")
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.startswith("PartNum=6")
    r2 = run(EXE, d, str(tmp_path / "host.fasta"), case, threads=8, backend="host")
    assert r2.returncode == 0 and r2.stdout == r.stdout
    assert open(tmp_path / "ours.fasta", "rb").read() == open(tmp_path / "host.fasta", "rb").read()
