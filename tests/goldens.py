"""Access to the committed golden fixtures (tests/golden/, made by tests/golden/make_golden.py from the
compiled reference)."""
import gzip
import hashlib
import json
import os
import re
import subprocess
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


def repeat_config(ind, times):
    """config.txt written `times` times over: block b reads the files of block b % (blocks of the golden)"""
    cfg_path = os.path.join(ind, "config.txt")
    text = open(cfg_path).read().rstrip("\n") + "\n"
    open(cfg_path, "w").write("\n".join([text] * times))


def compare_repeated_blocks(name, out_dir, n_blocks, per_golden=2):
    """the outputs of a run over repeat_config()'s blocks: block b's files must be the golden's bytes of block b % per_golden
    under b's own prefix; contig.txt = the golden's set of names"""
    want = golden_out_files(name)
    got = {f: open(os.path.join(out_dir, f), "rb").read() for f in sorted(os.listdir(out_dir))}
    seen = {}
    for f, data in got.items():
        if f == "contig.txt":
            assert sorted(set(data.splitlines())) == sorted(set(want["contig.txt"].splitlines())), "contig.txt"
            continue
        no, rest = f.split("_", 1)
        no = int(no)
        assert 0 <= no < n_blocks, f
        seen[no] = seen.get(no, 0) + 1
        golden = want[f"{no % per_golden}_{rest}"]
        if rest.endswith((".con", ".fasta")):  # (the sequence names inside start with the block number too)
            golden = re.sub(rb"(?m)^(>?)%d_" % (no % per_golden), rb"\g<1>%d_" % no, golden)
        assert data == golden, f"{f} differs from the golden bytes of block {no % per_golden}"
    assert "contig.txt" in got
    for b in range(n_blocks):
        assert seen.get(b, 0) == sum(1 for f in want if f.startswith(f"{b % per_golden}_")), f"block {b}: files missing"


def check_blocks_parsed_ahead(exe, blocks, work):
    """The two-block golden's config.txt written twice over (blocks 2 and 3 read the files of blocks 0 and 1), run by `exe`
    (a pagraph_driver.cpp program) with and without parsing the next block ahead (PAGRAPH_PREFETCH) and running a block's host
    half beside the next block's device work (PAGRAPH_OVERLAP; the HIP backend): every block's outputs
    must be the golden's bytes under its own prefix — with all blocks handled, and with blocks skipped (PAGRAPH_BLOCKS =
    `blocks`) between the one that runs and the one parsed ahead."""
    name = "two_blocks_both_orient_t16"
    spec = load_spec(name)
    ind = materialize_inputs(name, os.path.join(work, "in"))
    cfg_path = os.path.join(ind, "config.txt")
    text = open(cfg_path).read()
    open(cfg_path, "w").write(text.rstrip("\n") + "\n\n" + text)
    want = golden_out_files(name)
    outs = {}
    for ahead in ("1", "0"):
        out = os.path.join(work, "out" + ahead)
        os.makedirs(out, exist_ok=True)
        argv = synth.pagraph_argv(exe, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
        env = dict(os.environ, PAGRAPH_PREFETCH=ahead, PAGRAPH_OVERLAP=ahead)
        if blocks:
            env.update(PAGRAPH_BLOCKS=blocks, PAGRAPH_PART="0")
        r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
        outs[ahead] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out))}
    assert sorted(outs["1"]) == sorted(outs["0"])
    handled = [int(b) for b in blocks.split(",")] if blocks else [0, 1, 2, 3]
    seen = set()
    for f, data in outs["1"].items():
        if f.startswith("contig.txt"):
            assert sorted(data.splitlines()) == sorted(outs["0"][f].splitlines())
            continue
        assert data == outs["0"][f], f
        no, rest = f.split("_", 1)
        no = int(no)
        assert no in handled, f
        seen.add(no)
        golden = want[f"{no % 2}_{rest}"]
        if rest.endswith((".con", ".fasta")):  # (the sequence names inside start with the block number too)
            golden = re.sub(rb"(?m)^(>?)%d_" % (no % 2), rb"\g<1>%d_" % no, golden)
        assert data == golden, f"{f} differs from the golden bytes of block {no % 2}"
    assert seen == set(handled)
    for b in handled:
        assert sum(1 for f in outs["1"] if f.startswith(f"{b}_")) == sum(1 for f in want if f.startswith(f"{b % 2}_"))
