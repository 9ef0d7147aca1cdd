"""Golden vectors for the k-mer counter: written by the COMPILED REFERENCE (oracle/_ref/kmer_counter, built from
/root/reference by oracle/Makefile).  Run here (needs /root/reference); the outputs are committed.

    python tests/golden/kmer_counter/make_golden.py
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import kmer_cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "kmer_counter")


def main():
    for name, case in kmer_cases.CASES.items():
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        reads = os.path.join(d, "reads." + case["fmt"])
        kmer_cases.write_reads(case, reads)
        out = os.path.join(d, "expected.bin")
        subprocess.run([REF, "-t", str(case["threads"]), "-i", reads, "-o", out, "-k", str(case["k"]), "-m", repr(case["threshold"])],
                       check=True, capture_output=True)
        print(name, os.path.getsize(out))


if __name__ == "__main__":
    main()
