"""CPU, world_size 2, gloo: the N > 1 path of the framework — block-to-rank assignment, sharded execution
with gathered exit codes, and the max-time / sum-of-units aggregation bench.py uses."""
import os
import sys

import pytest
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligngraph2_amd import parallel  # noqa: E402


def test_assign_blocks_is_balanced_and_complete():
    sizes = [250, 30, 240, 90, 90, 10, 200, 200]
    for world in (1, 2, 4, 8):
        a = parallel.assign_blocks(sizes, world)
        assert sorted(i for r in a for i in r) == list(range(len(sizes)))
        loads = [sum(sizes[i] for i in r) for r in a]
        assert max(loads) - min(loads) <= max(sizes)
    assert parallel.assign_blocks(sizes, 2) == parallel.assign_blocks(sizes, 2)  # deterministic


def _worker(rank, world, port, tmp):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist = parallel.init("gloo")
    dirs = [os.path.join(tmp, f"ref{i}") for i in range(5)]
    sizes = [50, 10, 40, 30, 20]

    def run_one(d, local):
        os.makedirs(d, exist_ok=True)
        open(os.path.join(d, "DONE"), "w").write(f"rank{rank} gpu{local}\n")
        return 0 if not d.endswith("ref3") else 7

    codes = parallel.run_sharded(dirs, sizes, run_one, dist)
    assert codes == [0, 0, 0, 7, 0]
    secs, units = parallel.aggregate(dist, 1.0 + rank, 100.0 * (rank + 1))
    assert secs == float(world) and units == 100.0 * world * (world + 1) / 2
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_run_world2_gloo(tmp_path):
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    done = sorted(os.listdir(tmp_path))
    assert done == [f"ref{i}" for i in range(5)]
    owners = {d: open(tmp_path / d / "DONE").read().split()[0] for d in done}
    assert set(owners.values()) == {"rank0", "rank1"}  # both ranks did work


# ---- one graph over several ranks: the exchange of real tuples (SURVEY §8e level 2) -------------------------------------
def _oracle_streams_with_reads(in_dir):
    """The C oracle's emitted tuple / edge streams of a small synthetic block + the emission position of the read that
    emitted every record (test hook pago_debug_stream_reads)."""
    import ctypes as C
    import numpy as np
    import pagctl
    inp = pagctl.LoadedInput(in_dir, threads=4, eps=10, cov=2)
    lib = pagctl.oracle_lib()
    g = lib.pago_create(inp.kmer_words, inp.n_kmer_words, inp.k)
    lib.pago_debug_enable(g, 1)
    st = pagctl.BuildStats()
    assert lib.pago_process(g, inp.view, C.byref(st)) == 0
    nt, ne = C.c_uint64(), C.c_uint64()
    lib.pago_debug_stream_sizes(g, C.byref(nt), C.byref(ne))
    s = {"tkey": np.zeros(nt.value, np.uint32), "tval": np.zeros(nt.value, np.uint64), "ekey": np.zeros(ne.value, np.uint32),
         "eval": np.zeros(ne.value, np.uint64), "tread": np.zeros(nt.value, np.uint32), "eread": np.zeros(ne.value, np.uint32)}
    lib.pago_debug_streams(g, *[s[k].ctypes.data for k in ("tkey", "tval", "ekey", "eval")])
    lib.pago_debug_stream_reads.argtypes = [C.c_void_p] * 3
    lib.pago_debug_stream_reads(g, s["tread"].ctypes.data, s["eread"].ctypes.data)
    n_reads = C.cast(inp.view, C.POINTER(C.c_uint64))[1]  # pag_build_input: {u32 on_device, u32 n_threads, pag_seqs reads{u64 n_seqs ...}}
    t1, e1 = int(st.n_tuples[0]), int(st.n_edges[0])
    lib.pago_destroy(g)
    inp.close()
    return s, int(n_reads), int(inp.k), t1, e1


def _shard_worker(rank, world, port, in_dir, out_dir):
    import numpy as np
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = parallel.init("gloo")
    s, n_reads, k, t1_all, e1_all = _oracle_streams_with_reads(in_dir)
    lg = {1: 0, 2: 1, 4: 2, 8: 3}[world]
    lo, hi = n_reads * rank // world, n_reads * (rank + 1) // world
    res = {}
    for name, n1_all in (("t", t1_all), ("e", e1_all)):
        key, val, rd = s[name + "key"], s[name + "val"], s[name + "read"]
        is_p1 = np.arange(len(key)) < n1_all
        # what THIS rank would extract: the records of its reads, in canonical order (pass 1 then pass 2, emission order)
        mine = (rd >= lo) & (rd < hi)
        mk, mv, mp1 = key[mine], val[mine], is_p1[mine]
        owner = (mk >> np.uint32(2 * k - lg)).astype(np.int64) if lg else np.zeros(len(mk), np.int64)
        order = np.argsort(owner, kind="stable")  # = the stable partition of pag_shard_extract
        mk, mv, mp1, owner = mk[order], mv[order], mp1[order], owner[order]
        counts = np.zeros((world, 2), np.int64)
        for o in range(world):
            counts[o, 0] = int(((owner == o) & mp1).sum())
            counts[o, 1] = int(((owner == o) & ~mp1).sum())
        allc = [torch.zeros(world, 2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allc, torch.from_numpy(counts))
        allc = torch.stack(allc).numpy()  # [src][dst][pass]
        (rk, rv), n1 = parallel.exchange_stream((torch.from_numpy(mk.view(np.int32)), torch.from_numpy(mv.view(np.int64))), allc, rank, world, dist=dist)
        rk, rv = rk.numpy().view(np.uint32), rv.numpy().view(np.uint64)
        # the owner's merge: stable sort by k-mer of [pass 1 from rank 0 ..][pass 2 from rank 0 ..]
        so = np.argsort(rk, kind="stable")
        got_k, got_v = rk[so], rv[so]
        # expectation from the un-sharded stream: the records of this owner's k-mer range, stably sorted by k-mer
        gowner = (key >> np.uint32(2 * k - lg)).astype(np.int64) if lg else np.zeros(len(key), np.int64)
        sel = gowner == rank
        wk, wv = key[sel], val[sel]
        wo = np.argsort(wk, kind="stable")
        assert n1 == int((sel & is_p1).sum())
        assert np.array_equal(got_k, wk[wo]) and np.array_equal(got_v, wv[wo]), f"rank {rank}: stream {name} differs after the exchange"
        res[name] = int(len(got_k))
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(f"{res['t']} {res['e']} {int(sel.sum())}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_tuple_exchange_world2_gloo_restores_the_canonical_order(tmp_path):
    """Real tuples (the C oracle's emitted streams of a synthetic block, 4-thread emission order) through the exchange the
    multi-GPU build uses — all_to_all_single with split sizes, here over gloo — and the owner-side layout + stable merge:
    every owner ends up with exactly the records of its k-mer range in exactly the order a single process has them."""
    import synth
    d = str(tmp_path / "in")
    synth.generate(synth.Spec(seed=11, ref_len=9000, n_reads=300, read_len=800, k=8, contigs=[(150, 4200, False), (4450, 8850, True)]), d)
    port = 29500 + (os.getpid() + 17) % 1000
    mp.spawn(_shard_worker, args=(2, port, d, str(tmp_path)), nprocs=2, join=True)
    got = [open(tmp_path / f"rank{r}.txt").read().split() for r in range(2)]
    assert all(int(g[0]) > 100 and int(g[1]) > 100 for g in got)  # both owners received records


# ---- config blocks of one pre_process directory over the ranks -------------------------------------------
def _blocks_worker(rank, world, port, in_dir, out_dir, exe, argv):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist = parallel.init("gloo")
    codes = parallel.run_config_blocks(in_dir, out_dir, argv, dist, exe=exe)
    assert codes == [0, 0]
    dist.destroy_process_group()


def test_config_blocks_over_two_ranks_equal_the_reference_outputs(tmp_path):
    """The two-block golden (written by the compiled reference in ONE run) with its blocks dealt out over two ranks: each rank
    runs the driver once for its block (here the oracle-backed harness build of the same driver, no GPU needed), rank 0 merges
    contig.txt — the output directory must be the reference's, file for file."""
    import subprocess
    import goldens
    import pagctl
    import synth
    subprocess.run(["make", "-C", pagctl.ROOT, "harness"], check=True, capture_output=True)
    name = "two_blocks_both_orient_t16"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(tmp_path / "in"))
    out = str(tmp_path / "out")
    os.makedirs(out)
    exe = os.path.join(pagctl.ROOT, "tests", "harness", "bin", "pagraph_oracle")
    argv = synth.pagraph_argv(exe, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])[1:]
    assert len(parallel.read_config_blocks(ind)) == 2
    port = 29500 + (os.getpid() + 41) % 1000
    mp.spawn(_blocks_worker, args=(2, port, ind, out, exe, argv), nprocs=2, join=True)
    goldens.compare_out_dir(name, out)


def _blocks_worker_n(rank, world, port, in_dir, out_dir, exe, argv, extra_env):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.update(extra_env)
    dist = parallel.init("gloo")
    codes = parallel.run_config_blocks(in_dir, out_dir, argv, dist, exe=exe)
    assert codes == [0] * world, codes
    dist.destroy_process_group()


def test_24_config_blocks_dealt_over_eight_ranks(tmp_path):
    """The schedule of BASELINE configs[3] (a genome of 24 reference sequences over the 8 GPUs of a node: level 1, one pagraph
    process per GPU for the blocks it was dealt, no data-path collective) at its shape: a config.txt of 24 blocks of two
    sizes, dealt longest-first over 8 ranks (3 blocks each), every rank's driver run once — here the oracle-backed harness
    build of the driver, no GPU needed — rank 0 merging contig.txt.  Every block's files must be the reference's golden bytes
    under the block's own number."""
    import subprocess
    import goldens
    import pagctl
    import synth
    subprocess.run(["make", "-C", pagctl.ROOT, "harness"], check=True, capture_output=True)
    name = "two_blocks_both_orient_t16"
    spec = goldens.load_spec(name)
    ind = goldens.materialize_inputs(name, str(tmp_path / "in"))
    goldens.repeat_config(ind, 12)
    out = str(tmp_path / "out")
    os.makedirs(out)
    exe = os.path.join(pagctl.ROOT, "tests", "harness", "bin", "pagraph_oracle")
    argv = synth.pagraph_argv(exe, ind, out, threads=spec["threads"], epsilon=spec["epsilon"], cov=spec["cov"])[1:]
    blocks = parallel.read_config_blocks(ind)
    assert len(blocks) == 24
    sizes = [sum(os.path.getsize(os.path.join(ind, f)) for f in b[1:4]) for b in blocks]
    deal = parallel.assign_blocks(sizes, 8)
    assert sorted(x for d in deal for x in d) == list(range(24)) and all(len(d) == 3 for d in deal)
    loads = [sum(sizes[i] for i in d) for d in deal]
    assert max(loads) - min(loads) <= max(sizes)  # (longest-first: no rank is more than one block behind)
    port = 29500 + (os.getpid() + 77) % 1000
    mp.spawn(_blocks_worker_n, args=(8, port, ind, out, exe, argv, {}), nprocs=8, join=True)
    goldens.compare_repeated_blocks(name, out, 24)
