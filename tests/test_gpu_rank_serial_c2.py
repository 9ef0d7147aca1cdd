"""GPU: the rank-serial driver of the sharded build (aligngraph2_amd/rank_serial.py: the N ranks of ONE block one after the other on
one device, what a rank takes in recomputed when its turn comes) at BASELINE configs[1] size — one tenth of configs[2], the
block it exists for — held to a GOLDEN and to the one-GPU run: every output file of the N = 4 run against the SHA-256 of what
the compiled reference wrote for this very workload (profiles/r04_c2_text_parity.json: `pagraph -t 16` under the
thread-serialising shim, 1 639 s, kept as digests), and the same digest over all files as the one-handle run."""
import json
import os
import subprocess
import sys

import pytest

import pagctl

DIGESTS = os.path.join(pagctl.ROOT, "profiles", "r04_c2_text_parity.json")


@pytest.mark.gpu
def test_rank_serial_run_at_configs1_size_equals_the_reference_digests(tmp_path):
    assert len(json.load(open(DIGESTS))["compare"]["reference_sha256"]) == 53
    out = tmp_path / "rs.json"
    r = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "tests", "c3_rank_serial.py"), str(out), "--reads", "100000", "--ref-len", "50000000",
                        "--ranks", "4", "--one-gpu", "--reference-digests", DIGESTS], capture_output=True, text=True, timeout=1200, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2500:]
    rec = json.load(open(out))
    run = rec["runs"][0]
    assert run["reference_digests"]["identical"] and run["reference_digests"]["files"] == 52
    assert rec["outputs_identical_for_all_runs"] and set(rec["digests"]) == {"one_gpu", "n4"}
    assert run["count_lines_sum_over_owners"] == rec["one_gpu"]["count_lines"]
    assert run["host_bytes_growth"] < 8e9, "the rank-serial run parks data on the host again"
    assert max(x["held_fraction"] for x in run["ranks"]) < 0.25 + 0.15
