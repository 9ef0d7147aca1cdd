"""GPU: BASELINE configs[1] at FULL size as text files (100 000 x 10 kb reads vs 50 Mb, 6.9 GB of inputs in /dev/shm) through the
drop-in executable, every one of its 53 output files held against the SHA-256 of what the compiled reference wrote for the
same files (profiles/r04_c2_text_parity.json: the reference's `-t 16` run under the thread-serialising shim, 1 639 s — kept as
digests so that this comparison costs a minute instead of half an hour).  Parsers, pag_prepare, build, walks, writers: the
whole product at the size the benchmark is quoted on."""
import json
import os
import shutil
import subprocess
import sys

import pytest

import pagctl

DIGESTS = os.path.join(pagctl.ROOT, "profiles", "r04_c2_text_parity.json")


@pytest.mark.gpu
def test_full_size_text_run_equals_the_reference_digests(tmp_path):
    if not os.path.isdir("/dev/shm") or shutil.disk_usage("/dev/shm").free < 12 * 2 ** 30:
        pytest.skip("needs 12 GB of /dev/shm for the text inputs and outputs")
    assert "reference_sha256" in json.load(open(DIGESTS))["compare"]
    out = tmp_path / "parity.json"
    r = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "tests", "c2_text_runs.py"), str(out), "--compare-with", DIGESTS], capture_output=True, text=True,
                       timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["ours"]["returncode"] == 0, rec["ours"]["stderr_tail"][-1500:]
    kept = json.load(open(DIGESTS))
    assert rec["input_bytes"] == kept["input_bytes"], "the generator no longer writes the text files the reference's digests were taken on (tests/biggen.py drift)"
    cmp = rec["compare"]
    assert cmp["files_ours"] == cmp["files_reference"] == 53
    assert cmp["identical"], cmp["differing_files"]
