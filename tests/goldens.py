"""Access to the committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py from the
compiled reference)."""
import gzip
import hashlib
import json
import os
import tarfile

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "spec.json")))


def load_spec(name):
    return json.load(open(os.path.join(GOLDEN, name, "spec.json")))


def materialize_inputs(name, dest):
    """Inputs of a golden case: the committed archive when present, otherwise regenerated from the spec
    and checked against the committed hashes (so generator drift cannot silently change the question)."""
    os.makedirs(dest, exist_ok=True)
    case = os.path.join(GOLDEN, name)
    tar = os.path.join(case, "inputs.tar.gz")
    if os.path.exists(tar):
        with tarfile.open(tar) as tf:
            tf.extractall(dest)
    else:
        generate_case(load_spec(name), dest)
    want = json.load(open(os.path.join(case, "inputs.sha256")))
    for f, h in want.items():
        got = hashlib.sha256(open(os.path.join(dest, f), "rb").read()).hexdigest()
        assert got == h, f"golden input drift in {name}/{f}"
    return dest


def generate_case(spec, dest):
    """spec["spec"]: Spec kwargs of a single-block case; spec["blocks"]: list of Spec kwargs of a multi-block case"""
    if "blocks" in spec:
        def fix(kw):
            kw = dict(kw)
            if "contigs" in kw:
                kw["contigs"] = [tuple(c) for c in kw["contigs"]]
            if "both_orient" in kw:
                kw["both_orient"] = tuple(kw["both_orient"])
            return kw
        return synth.generate_multi([synth.Spec(**fix(kw)) for kw in spec["blocks"]], dest)
    return synth.generate(synth.Spec(**spec["spec"]), dest)


def golden_graph(name):
    return gzip.open(os.path.join(GOLDEN, name, "graph.txt.gz"), "rb").read()


def golden_out_files(name):
    d = os.path.join(GOLDEN, name, "out")
    return {f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))}


def compare_out_dir(name, out_dir):
    want = golden_out_files(name)
    got_files = sorted(os.listdir(out_dir))
    assert got_files == sorted(want), f"{name}: output files {got_files} != golden {sorted(want)}"
    for f, data in want.items():
        got = open(os.path.join(out_dir, f), "rb").read()
        if f == "contig.txt":  # a set: the reference writes it in hash order
            got = b"".join(sorted(got.splitlines(keepends=True)))
        assert got == data, f"{name}: {f} differs from the reference's golden output"
