#!/usr/bin/env python3
"""Writes tests/golden/pa_cns/<case>.{fasta,stdout}: the outputs of the COMPILED REFERENCE (oracle/_ref/pa_cns) on the seeded
inputs of tests/cns_cases.py.  Run in the build container (needs /root/reference for `make -C oracle ref`)."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cns_cases  # noqa: E402

REF = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "pa_cns")
for name, case in cns_cases.CASES.items():
    with tempfile.TemporaryDirectory() as d:
        cns_cases.write_case(case, d)
        out = os.path.join(HERE, "pa_cns", name + ".fasta")
        r = subprocess.run(cns_cases.argv(REF, d, out, case), capture_output=True, text=True, check=True)
        open(os.path.join(HERE, "pa_cns", name + ".stdout"), "w").write(r.stdout)
        print(name, os.path.getsize(out))
