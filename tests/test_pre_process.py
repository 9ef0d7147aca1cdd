"""pre_process drop-in, host only: every output file byte-identical to the compiled reference's
(oracle/_ref/pre_process) on seeded inputs, and to the committed golden outputs where the reference is absent."""
import filecmp
import os
import subprocess

import pytest

import pagctl
import preproc_cases

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pre_process")
REF = os.path.join(pagctl.REF_DIR, "pre_process")
GOLD = os.path.join(pagctl.ROOT, "tests", "golden", "pre_process")


def run(exe, d, out, case):
    os.makedirs(out, exist_ok=True)
    return subprocess.run(preproc_cases.argv(exe, d, out, case), capture_output=True, text=True, timeout=300)


def same_dirs(a, b):
    fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
    assert fa == fb, (fa, fb)
    for f in fa:
        assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), f


@pytest.mark.parametrize("name", list(preproc_cases.CASES))
def test_pre_process_matches_reference_binary(name, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pre_process not built (make product)")
    if not os.path.exists(REF):
        pytest.skip("oracle/_ref/pre_process was not built (needs /root/reference at build time)")
    case = preproc_cases.CASES[name]
    d = preproc_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours"), case)
    r2 = run(REF, d, str(tmp_path / "ref"), case)
    assert r1.returncode == r2.returncode == 0, (r1.stderr[-500:], r2.stderr[-500:])
    same_dirs(str(tmp_path / "ours"), str(tmp_path / "ref"))
    assert os.path.getsize(os.path.join(str(tmp_path / "ours"), "config.txt")) > 0 or name == "two_refs_one_each"


@pytest.mark.parametrize("name", list(preproc_cases.CASES))
def test_pre_process_matches_golden(name, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pre_process not built (make product)")
    case = preproc_cases.CASES[name]
    d = preproc_cases.write_case(case, str(tmp_path / "in"))
    r = run(EXE, d, str(tmp_path / "ours"), case)
    assert r.returncode == 0, r.stderr[-500:]
    same_dirs(str(tmp_path / "ours"), os.path.join(GOLD, name))


def test_pre_process_usage_and_test_flag(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("aligngraph2_amd/bin/pre_process not built (make product)")
    assert subprocess.run([EXE], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "-h"], capture_output=True).returncode == 0
    assert subprocess.run([EXE, "--bogus"], capture_output=True).returncode == 1
    case = preproc_cases.CASES["three_refs_k1"]
    d = preproc_cases.write_case(case, str(tmp_path / "in"))
    out = str(tmp_path / "o")
    os.makedirs(out)
    r = subprocess.run(preproc_cases.argv(EXE, d, out, case) + ["--test"], capture_output=True, text=True)
    assert r.returncode == 0 and os.listdir(out) == []


def test_pre_process_dies_like_the_reference_on_a_non_numeric_read_name(tmp_path):
    if not (os.path.exists(EXE) and os.path.exists(REF)):
        pytest.skip("needs both binaries")
    case = preproc_cases.CRASH_CASE
    d = preproc_cases.write_case(case, str(tmp_path / "in"))
    r1 = run(EXE, d, str(tmp_path / "ours"), case)
    r2 = run(REF, d, str(tmp_path / "ref"), case)
    assert r1.returncode == r2.returncode != 0
    assert "stoll" in r1.stderr and "stoll" in r2.stderr
