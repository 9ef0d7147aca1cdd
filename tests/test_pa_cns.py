"""pa_cns drop-in: the consensus FASTA and the three stdout lines byte-identical to the compiled reference's
(oracle/_ref/pa_cns: the reference's own sources + its vendored Boost) on seeded inputs, and to the committed golden outputs
where the reference is absent.  CPU: the two host builds of the graph stage — PA_CNS_BACKEND=host (std::vector / std::map
restatement) and =flat (the DEVICE's code, csrc/hip/cns_graph.hpp, compiled for the host: flat arrays, linked edge lists).
The device itself (the default backend): tests/test_gpu_pa_cns.py."""
import os
import subprocess

import pytest

import cns_cases
import pagctl

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pa_cns")
REF = os.path.join(pagctl.REF_DIR, "pa_cns")
GOLD = os.path.join(pagctl.ROOT, "tests", "golden", "pa_cns")


BACKENDS = ("host", "flat")


def run(exe, d, out, case, threads=4, backend=None):
    env = dict(os.environ)
    if backend:
        env["PA_CNS_BACKEND"] = backend
    return subprocess.run(cns_cases.argv(exe, d, out, case, threads), capture_output=True, text=True, timeout=600, env=env)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", list(cns_cases.CASES))
def test_pa_cns_matches_reference_binary(name, backend, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/pa_cns was not built (needs /root/reference at build time)")
    case = cns_cases.CASES[name]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=3, backend=backend)
    r2 = run(REF, d, str(tmp_path / "ref.fasta"), case, threads=5)
    assert r1.returncode == r2.returncode == 0, (r1.stderr[-500:], r2.stderr[-500:])
    assert r1.stdout == r2.stdout
    a, b = open(tmp_path / "ours.fasta", "rb").read(), open(tmp_path / "ref.fasta", "rb").read()
    assert a == b
    assert len(a) > case["backbone"] // 2


@pytest.mark.parametrize("backend", BACKENDS)
def test_pa_cns_matches_reference_binary_at_pipeline_settings(backend, tmp_path):
    if not (os.path.exists(EXE) and os.path.exists(REF)):
        pytest.skip("needs both binaries")
    case = cns_cases.DEEP_CASE
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=16, backend=backend)
    r2 = run(REF, d, str(tmp_path / "ref.fasta"), case, threads=16)
    assert r1.returncode == r2.returncode == 0 and r1.stdout == r2.stdout
    assert open(tmp_path / "ours.fasta", "rb").read() == open(tmp_path / "ref.fasta", "rb").read()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", list(cns_cases.CASES))
def test_pa_cns_matches_golden(name, backend, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    case = cns_cases.CASES[name]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours.fasta"), case, backend=backend)
    assert r.returncode == 0, r.stderr[-500:]
    assert open(tmp_path / "ours.fasta", "rb").read() == open(os.path.join(GOLD, name + ".fasta"), "rb").read()
    assert r.stdout == open(os.path.join(GOLD, name + ".stdout")).read()


def test_pa_cns_cli_contract(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    assert subprocess.run([EXE], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "-h"], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "--bogus", "1"], capture_output=True).returncode == 1
    r = subprocess.run([EXE, "-i", str(tmp_path / "missing.fasta"), "-o", str(tmp_path / "o"), "-a", str(tmp_path / "missing.ref")], capture_output=True, text=True)
    assert r.returncode == 2 and "No backbone" in r.stderr


def test_pa_cns_device_backend_fails_loudly_without_a_device(tmp_path):
    """PA_CNS_BACKEND=hip builds the graphs on the device: on a box without one the program must say so and fail — no silent
    host fallback (skipped where a GPU is present).  The default chooses by the number of parts (a contig of the pipeline: host
    threads running the device's code) and starts without the ROCm runtime: libpagraph_hip.so is loaded by the device backend only."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    case = cns_cases.CASES["one_part"]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    env = dict(os.environ, PA_CNS_BACKEND="hip")
    r = subprocess.run(cns_cases.argv(EXE, d, str(tmp_path / "o.fasta"), case), capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and ("pag_cns_consensus" in r.stderr or "libpagraph_hip.so" in r.stderr), r.stderr[-500:]
    assert not os.path.exists(tmp_path / "o.fasta") or os.path.getsize(tmp_path / "o.fasta") == 0


def test_pa_cns_default_backend_needs_no_device_library(tmp_path):
    """the default (by part count: one part here) runs the flat graph code on host threads; the executable does not link the HIP
    library (ldd), so the host backends start on a machine without the ROCm runtime"""
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    case = cns_cases.CASES["one_part"]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    env = dict(os.environ)
    env.pop("PA_CNS_BACKEND", None)
    r = subprocess.run(cns_cases.argv(EXE, d, str(tmp_path / "o.fasta"), case), capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-500:]
    env["PA_CNS_BACKEND"] = "host"
    r2 = subprocess.run(cns_cases.argv(EXE, d, str(tmp_path / "o2.fasta"), case), capture_output=True, text=True, env=env, timeout=300)
    assert r2.returncode == 0 and open(tmp_path / "o.fasta", "rb").read() == open(tmp_path / "o2.fasta", "rb").read()
    ldd = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "pagraph_hip" not in ldd and "amdhip" not in ldd, ldd
