"""Golden outputs of pre_process, written by the COMPILED REFERENCE (oracle/_ref/pre_process).  Needs /root/reference.

    python tests/golden/pre_process/make_golden.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import preproc_cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "pre_process")

for name, case in preproc_cases.CASES.items():
    d = preproc_cases.write_case(case, tempfile.mkdtemp())
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    subprocess.run(preproc_cases.argv(REF, d, out, case), check=True, capture_output=True)
    print(name, sorted(os.listdir(out)))
