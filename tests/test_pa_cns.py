"""pa_cns drop-in, host only like the reference's: the consensus FASTA and the three stdout lines
byte-identical to the compiled reference's (oracle/_ref/pa_cns: the reference's own sources + its vendored Boost) on seeded
inputs, and to the committed golden outputs where the reference is absent."""
import os
import subprocess

import pytest

import cns_cases
import pagctl

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pa_cns")
REF = os.path.join(pagctl.REF_DIR, "pa_cns")
GOLD = os.path.join(pagctl.ROOT, "tests", "golden", "pa_cns")


def run(exe, d, out, case, threads=4):
    return subprocess.run(cns_cases.argv(exe, d, out, case, threads), capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("name", list(cns_cases.CASES))
def test_pa_cns_matches_reference_binary(name, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/pa_cns was not built (needs /root/reference at build time)")
    case = cns_cases.CASES[name]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=3)
    r2 = run(REF, d, str(tmp_path / "ref.fasta"), case, threads=5)
    assert r1.returncode == r2.returncode == 0, (r1.stderr[-500:], r2.stderr[-500:])
    assert r1.stdout == r2.stdout
    a, b = open(tmp_path / "ours.fasta", "rb").read(), open(tmp_path / "ref.fasta", "rb").read()
    assert a == b
    assert len(a) > case["backbone"] // 2


def test_pa_cns_matches_reference_binary_at_pipeline_settings(tmp_path):
    if not (os.path.exists(EXE) and os.path.exists(REF)):
        pytest.skip("needs both binaries")
    case = cns_cases.DEEP_CASE
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours.fasta"), case, threads=16)
    r2 = run(REF, d, str(tmp_path / "ref.fasta"), case, threads=16)
    assert r1.returncode == r2.returncode == 0 and r1.stdout == r2.stdout
    assert open(tmp_path / "ours.fasta", "rb").read() == open(tmp_path / "ref.fasta", "rb").read()


@pytest.mark.parametrize("name", list(cns_cases.CASES))
def test_pa_cns_matches_golden(name, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pa_cns not built (make product)")
    case = cns_cases.CASES[name]
    d = cns_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours.fasta"), case)
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
