"""GPU: the drop-in `pagraph` executable (HIP backend) against the reference's golden outputs and,
on fresh seeded inputs, against the compiled reference binary itself (oracle/_ref/pagraph travels to
the GPU box prebuilt).  Byte-exact on every output file."""
import os
import re
import subprocess

import pytest

import goldens
import pagctl
import synth

EXE = os.path.join(pagctl.ROOT, "aligngraph2_amd", "bin", "pagraph")


# walker modes: speculative choice among probes (default; a failed speculation is redone exactly) and PAG_WALK_EXACT=1
# (every probe runs to its end before the choice) — both must reproduce the reference byte for byte
# and the segment-parallel walk: at the default segment length the few-kb golden contigs are walked in one piece ("uncut"
# is what the other two modes run); "pieces" cuts them every 400 bases so that checkpoints, adoption of segments and
# resumed walks all happen on inputs whose exact outputs the reference wrote — the leaping zone included (its segments
# can leap and are spliced under other conditions, k5_travel_host.hip try_merge_leap); "pieces-noleap" leaves the leaping
# zone to one exact walk, as round 1 of the cut did
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["speculative", "exact", "pieces", "pieces-exact", "pieces-noleap"])
@pytest.mark.parametrize("name", goldens.case_names())
def test_pagraph_matches_golden(name, mode, workdir):
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    out = str(workdir / name / ("out_" + mode))
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    env = dict(os.environ)
    for v in ("PAG_WALK_EXACT", "PAG_SEG_LEN", "PAG_SEG_OVERLAP", "PAG_SEG_SAFETY", "PAG_WALK_PIECES", "PAG_LEAP_PIECES"):
        env.pop(v, None)
    if mode.startswith("pieces"):
        # (PAG_DEBUG_CHECK_AGGS: the block tables the pack kernel attaches to every fetched path are recomputed on the host and compared)
        env.update(PAG_SEG_LEN="400", PAG_SEG_OVERLAP="150", PAG_SEG_SAFETY="200", PAGRAPH_TIMING="1", PAG_DEBUG_CHECK_AGGS="1")
    if mode.endswith("exact"):
        env["PAG_WALK_EXACT"] = "1"
    if mode.endswith("noleap"):
        env["PAG_LEAP_PIECES"] = "0"
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    assert "HIP gfx950" in r.stdout
    goldens.compare_out_dir(name, out)
    if mode == "pieces":
        print([ln for ln in r.stderr.splitlines() if "pieces:" in ln or "leaping zone:" in ln])
        for ln in r.stderr.splitlines():
            m = re.search(r"leaping zone: (\d+) segment jobs, (\d+) adopted", ln)
            if m:
                _LEAP_SEEN["jobs"] += int(m.group(1))
                _LEAP_SEEN["adopted"] += int(m.group(2))
            m = re.search(r"pieces: (\d+) segment jobs, (\d+) resume jobs, (\d+) vertices adopted", ln)
            if m:
                _LEAP_SEEN["vertices"] += int(m.group(3))


_LEAP_SEEN = {"jobs": 0, "adopted": 0, "vertices": 0}


@pytest.mark.gpu
def test_pieces_mode_really_spliced_on_the_goldens():
    """The "pieces" runs above (same process, earlier in this file) must have adopted segments — in the leaping zone too:
    outputs that equal the goldens because every splice was refused would prove nothing about the splice conditions."""
    if _LEAP_SEEN["jobs"] == 0:
        pytest.skip("the pieces-mode golden tests did not run in this session")
    assert _LEAP_SEEN["vertices"] > 1000
    assert _LEAP_SEEN["adopted"] >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("seed,threads", [(501, 1), (502, 16)])
def test_pagraph_matches_reference_binary_on_fresh_input(seed, threads, workdir):
    ref_bin = os.path.join(pagctl.REF_DIR, "pagraph")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/pagraph was not built (needs /root/reference at build time)")
    d = str(workdir / f"fresh{seed}")
    synth.generate(synth.Spec(seed=seed, ref_len=20000, n_reads=500, read_len=1200, read_len_jitter=0.3, k=10,
                              contigs=[(300, 9500, False), (9900, 19700, seed % 2 == 0)], repeats=2), d + "/in")
    r = pagctl.run_reference(d + "/in", d + "/ref", threads=threads, eps=10, cov=2)
    assert r.returncode == 0
    os.makedirs(d + "/ours", exist_ok=True)
    o = subprocess.run(synth.pagraph_argv(EXE, d + "/in", d + "/ours", threads=threads, epsilon=10, cov=2),
                       capture_output=True, text=True)
    assert o.returncode == 0, o.stderr[-2000:]
    fr, fo = sorted(os.listdir(d + "/ref")), sorted(os.listdir(d + "/ours"))
    assert fr == fo
    for f in fr:
        a, b = open(f"{d}/ref/{f}", "rb").read(), open(f"{d}/ours/{f}", "rb").read()
        if f == "contig.txt":
            a, b = sorted(a.split()), sorted(b.split())
        assert a == b, f"{f} differs from the reference binary's output"


@pytest.mark.gpu
def test_walker_regrows_its_buffers_and_stays_exact(workdir, monkeypatch):
    """Tiny initial walk buffers: every contig overflows and is re-posted with doubled capacity until the walk fits."""
    name = goldens.case_names()[0]
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "regrow" / "in"))
    out = str(workdir / "regrow" / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    env = dict(os.environ, PAG_DEBUG_SEQCAP="48")
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    goldens.compare_out_dir(name, out)
    # ... from 16 vertices on, on the golden with the longest paths (far beyond 64 doublings' worth of the old limit)
    name = "join_fwd_t1"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "regrow16" / "in"))
    out = str(workdir / "regrow16" / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    r = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, PAG_DEBUG_SEQCAP="16", PAG_WALK_PIECES="0"), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    goldens.compare_out_dir(name, out)


# The walker waves are elastic (walker_grid.hpp): the result must not depend on how many of them the device carries, on
# waves leaving for lack of work and being replaced, or on a grid larger than the device can hold at once.
@pytest.mark.gpu
@pytest.mark.parametrize("grid", ["8-waves", "1-wave", "leave-at-once", "oversubscribed", "tiny-ring"])
@pytest.mark.parametrize("name", goldens.case_names()[:3] + ["join_fwd_t1", "join_rev_t16"])
def test_walks_do_not_depend_on_the_walker_grid(name, grid, workdir):
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / name / "in"))
    out = str(workdir / name / ("out_grid_" + grid))
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    env = dict(os.environ, PAG_SEG_LEN="400", PAG_SEG_OVERLAP="150", PAG_SEG_SAFETY="200")
    env.update({"8-waves": {"PAG_WALK_WAVES": "8"}, "1-wave": {"PAG_WALK_WAVES": "1"},
                "leave-at-once": {"PAG_WALK_WAVES": "16", "PAG_WALK_IDLE_US": "1"},     # every idle wave leaves: constant relaunching
                "oversubscribed": {"PAG_WALK_WAVES_PER_CU": "8"},                       # twice what the LDS lets be resident
                "tiny-ring": {"PAG_DEBUG_RING": "4"}}[grid])                             # rings of 4 jobs: a round's jobs wait in the backlog
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
    goldens.compare_out_dir(name, out)


@pytest.mark.gpu
def test_two_pagraph_processes_share_the_device(workdir):
    """Two drop-in processes at the same time on ONE GPU, each wanting a full grid of walkers (the LDS of every compute unit):
    both must finish, with the goldens' outputs (the caller of AlignGraph2.py may run several pipelines on a node)."""
    names = goldens.case_names()[:2]
    procs = []
    inputs = {name: goldens.materialize_inputs(name, str(workdir / "share" / name / "in")) for name in names}
    for rep in range(3):
        for name in names:
            spec = goldens.load_spec(name)
            ind = inputs[name]
            out = str(workdir / "share" / f"{name}_{rep}")
            os.makedirs(out, exist_ok=True)
            argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
            env = dict(os.environ, PAG_SEG_LEN="400", PAG_SEG_OVERLAP="150", PAG_SEG_SAFETY="200", PAG_WALK_IDLE_S="20")
            procs.append((name, out, subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)))
    for name, out, pr in procs:
        so, se = pr.communicate(timeout=300)
        assert pr.returncode == 0, se[-2000:] + so[-2000:]
        goldens.compare_out_dir(name, out)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name", ["join_fwd_t1", "join_rev_t16", "three_ctg_multi_t4", "two_blocks_both_orient_t16"])
def test_one_block_built_by_several_pagraph_processes(name, world, workdir):
    """PAGRAPH_SHARD=r/N: N drop-in processes (one per GPU on a real node; here they share the one device and exchange
    through the rendezvous directory, PAGRAPH_SHARD_TRANSPORT=host) build every config block TOGETHER — reads split for the
    extraction, k-mer ranges for sort / cluster / edges, every rank holding only the region of the graph its contigs need —
    and rank 0 writes the outputs: byte for byte the reference's golden files."""
    import tempfile
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / f"shardexe{world}" / name / "in"))
    out = str(workdir / f"shardexe{world}" / name / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    rdv = tempfile.mkdtemp(prefix="pagshard_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    for r in range(world):
        env = dict(os.environ, PAGRAPH_SHARD=f"{r}/{world}", PAGRAPH_SHARD_DIR=rdv, PAGRAPH_SHARD_TRANSPORT="host", PAG_COMM_TIMEOUT_S="120",
                   PAG_DEVICE_SHARERS=str(world))
        procs.append(subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    for r, pr in enumerate(procs):
        so, se = pr.communicate(timeout=300)
        assert pr.returncode == 0, f"rank {r}: " + se[-2000:] + so[-1000:]
    goldens.compare_out_dir(name, out)


@pytest.mark.gpu
@pytest.mark.parametrize("label,world,extra", [
    ("whole exchanges", 2, {"PAG_SHARD_CHUNKS": "1", "PAG_SHARD_PIPELINE": "0"}),
    ("three chunks", 4, {"PAG_SHARD_CHUNKS": "3"}),
    ("more chunks than a rank has reads", 2, {"PAG_SHARD_CHUNKS": "64"}),
    ("one rank over RCCL with itself", 1, {"PAG_COMM_FORCE_RCCL": "1", "PAGRAPH_SHARD_TRANSPORT": "rccl"}),
])
def test_sharded_build_in_chunks_and_pipelined_equals_the_golden(label, world, extra, workdir):
    """pag_shard_run sends a rank's tuples in chunks while the next chunk is extracted, and the selection for one rank while the
    next one is made (round 5), and every rank writes the path dumps of its own contigs: any number of chunks, the whole exchanges of
    round 4, and — the only way RCCL's grouped sends and
    receives of that path can run on a one-GPU box — ONE rank exchanging with itself over RCCL: the golden files every time."""
    import tempfile
    name = "two_blocks_both_orient_t16"
    spec = goldens.load_spec(name)
    tag = label.replace(" ", "_")
    ind = goldens.materialize_inputs(name, str(workdir / f"shardvar_{tag}" / name / "in"))
    out = str(workdir / f"shardvar_{tag}" / name / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    rdv = tempfile.mkdtemp(prefix="pagshard_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    procs = []
    for r in range(world):
        env = dict(os.environ, PAGRAPH_SHARD=f"{r}/{world}", PAGRAPH_SHARD_DIR=rdv, PAGRAPH_SHARD_TRANSPORT="host", PAG_COMM_TIMEOUT_S="120",
                   PAG_DEVICE_SHARERS=str(world), PAGRAPH_TIMING="1")
        env.update(extra)
        procs.append(subprocess.Popen(argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    for r, pr in enumerate(procs):
        so, se = pr.communicate(timeout=300)
        assert pr.returncode == 0, f"{label}, rank {r}: " + se[-2000:] + so[-1000:]
        assert "[shard timing]" in se, se[-500:]
        # every rank writes the path dumps of the contigs it walked (rank 0 the rest of the block's files)
        if r > 0:
            assert "path dumps of this rank" in se, se[-800:]
    goldens.compare_out_dir(name, out)


@pytest.mark.gpu
def test_config_blocks_dealt_out_over_processes_equal_the_golden(workdir):
    """PAGRAPH_BLOCKS: the blocks of one config.txt processed by different bin/pagraph processes (what parallel.
    run_config_blocks does with one process per GPU) — here one after the other on the one GPU — and contig.txt merged."""
    name = "two_blocks_both_orient_t16"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "blocks" / "in"))
    out = str(workdir / "blocks" / "out")
    os.makedirs(out, exist_ok=True)
    argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
    for part, blocks in enumerate(("1", "0")):
        r = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, PAGRAPH_BLOCKS=blocks, PAGRAPH_PART=str(part)), timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
    names = []
    for part in range(2):
        names += open(os.path.join(out, f"contig.txt.part{part}")).read().split()
        os.remove(os.path.join(out, f"contig.txt.part{part}"))
    open(os.path.join(out, "contig.txt"), "w").write("".join(n + "\n" for n in dict.fromkeys(names)))
    goldens.compare_out_dir(name, out)


def _blocks_worker_gpu(rank, world, port, in_dir, out_dir, argv):
    import sys
    sys.path.insert(0, pagctl.ROOT)
    from aligngraph2_amd import parallel
    # (the ranks share the box's one device: LOCAL_RANK 0 for all, the walker grids take a share each)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      PAG_DEVICE_SHARERS=str(world), PAG_WALK_IDLE_S="60")
    dist = parallel.init("gloo")
    codes = parallel.run_config_blocks(in_dir, out_dir, argv, dist, exe=EXE)
    assert codes == [0] * world, codes
    dist.destroy_process_group()


@pytest.mark.gpu
def test_24_config_blocks_over_eight_processes_sharing_the_device(workdir):
    """BASELINE configs[3]'s schedule at its shape (24 reference-sequence blocks over 8 ranks, parallel.run_config_blocks: the
    blocks weighed by their files, dealt longest-first, ONE bin/pagraph per rank for its three blocks, contig.txt merged by rank
    0) with the eight processes sharing the box's one MI355X (PAG_DEVICE_SHARERS=8) — against the one-process run of the same
    24 blocks and the reference's golden bytes of every block."""
    import torch.multiprocessing as mp
    name = "two_blocks_both_orient_t16"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "b24" / "in"))
    goldens.repeat_config(ind, 12)
    outs = {}
    for mode in ("one", "eight"):
        out = str(workdir / "b24" / mode)
        os.makedirs(out, exist_ok=True)
        argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
        if mode == "one":
            r = subprocess.run(argv, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
        else:
            port = 29100 + os.getpid() % 300
            mp.spawn(_blocks_worker_gpu, args=(8, port, ind, out, argv[1:]), nprocs=8, join=True)
        goldens.compare_repeated_blocks(name, out, 24)
        outs[mode] = {f: open(os.path.join(out, f), "rb").read() for f in sorted(os.listdir(out)) if f != "contig.txt"}
    assert outs["one"] == outs["eight"]


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", [None, "0,2,3", "1,3"])
def test_next_block_is_parsed_ahead_without_changing_any_output(blocks, workdir):
    """The driver parses the next block's text files while the current block is on the device (pagraph_driver.cpp):
    goldens.check_blocks_parsed_ahead, here with the HIP backend."""
    goldens.check_blocks_parsed_ahead(EXE, blocks, str(workdir / ("ahead_" + (blocks or "all").replace(",", "_"))))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["join_fwd_t1", "join_rev_t16", "two_blocks_both_orient_t16"])
def test_walk_that_leaves_the_cut_view_is_walked_again_on_the_whole_graph(name, workdir):
    """The safety net of the traversal's cut view (DESIGN §3 "The view"; pag_travel, k5_travel_host.hip): with no halo around the
    reference bands and no margin before the leaping zones every walk that can leap examines a marker record (GRADE_POISON /
    GRADE_POISON_IF_LEAP) -> PAG_ERANGE -> the half-run WalkSession is torn down, the whole graph's view built (view_off) and
    every contig walked again.  Outputs must be the reference's golden bytes — as they are with the cut switched off
    (PAG_TRAVEL_VIEW=whole) — and the walk-again path must really have run."""
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(workdir / "viewnet" / name / "in"))
    seen = {}
    for label, extra in (("nohalo", dict(PAG_VIEW_HALO="0", PAG_VIEW_MARGIN="0")), ("whole", dict(PAG_TRAVEL_VIEW="whole")), ("default", {})):
        out = str(workdir / "viewnet" / name / label)
        os.makedirs(out, exist_ok=True)
        argv = synth.pagraph_argv(EXE, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])
        env = dict(os.environ, PAGRAPH_TIMING="1", **extra)
        for v in ("PAG_VIEW_HALO", "PAG_VIEW_MARGIN", "PAG_TRAVEL_VIEW"):
            if v not in extra:
                env.pop(v, None)
        r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-2000:]
        goldens.compare_out_dir(name, out)
        seen[label] = r.stderr.count("a walk left the view")
    assert seen["nohalo"] > 0, "no walk left the view without halo and margin: the walk-again path did not run"
    assert seen["whole"] == 0 and seen["default"] == 0
