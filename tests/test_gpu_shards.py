"""GPU: ONE graph built by N shards (SURVEY §8e level 2) — emulated in one process on one device: N handles play the N
ranks, the collectives are replaced by slicing the other "ranks'" buffers (parallel.exchange_stream(peers=...)), every
kernel and every piece of host logic is the one the multi-GPU run uses (pag_shard_extract / _build / _import through
aligngraph2_amd.parallel.ShardedBuild).  N shards must reproduce the single pag_process BYTE FOR BYTE: count lines, every
CSR array, and the traversal outputs — also when the contigs are walked by different "ranks" and the travel sequences are
gathered for one pagh_assemble_paths."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import pagctl

sys.path.insert(0, pagctl.ROOT)
from aligngraph2_amd import parallel  # noqa: E402


def _csr(hip, g):
    nn, npos, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
    hip.pag_csr_sizes(C.c_void_p(g), C.byref(nn), C.byref(npos), C.byref(ne))
    arrs = {"node_code": np.zeros(nn.value + 1, np.uint32), "pos_off": np.zeros(nn.value + 1, np.uint64),
            "pos_ctg": np.zeros(npos.value + 1, np.uint32), "pos_ref": np.zeros(npos.value + 1, np.uint32),
            "pos_cnt": np.zeros(npos.value + 1, np.uint16), "edge_off": np.zeros(nn.value + 1, np.uint64),
            "edge_to": np.zeros(ne.value + 1, np.uint32), "edge_step": np.zeros(ne.value + 1, np.int32)}
    csr = pagctl.Csr(nn.value, npos.value, ne.value, *[arrs[k].ctypes.data for k in ("node_code", "pos_off", "pos_ctg", "pos_ref", "pos_cnt",
                                                                                    "edge_off", "edge_to", "edge_step")])
    assert hip.pag_export_csr(C.c_void_p(g), C.byref(csr)) == 0, hip.pag_last_error()
    return (nn.value, npos.value, ne.value), arrs


def _bind(hip):
    hip.pag_create_from_bitmap.restype = C.c_void_p
    hip.pag_export_csr.argtypes = [C.c_void_p, C.POINTER(pagctl.Csr)]
    hip.pag_csr_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
    parallel.bind_shard_api(hip)


def _sharded(hip, w, n_shards, regions=None):
    """the emulated N-rank build; returns the N handles and the total stats.  regions = None: every handle holds the WHOLE
    graph afterwards; regions = parallel.regions_for(...): handle r holds what rank r's traversals need (pag_shard_select)"""
    import torch
    sp = w.spec
    inp = w.build_input()
    gs, sbs = [], []
    for r in range(n_shards):
        err = C.c_int()
        g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
        assert g, hip.pag_last_error()
        gs.append(g)
        sbs.append(parallel.ShardedBuild(hip, g, inp, r, n_shards, "cuda"))
    ext = [sb.extract() for sb in sbs]  # (counts[world][4], (tk, tv), (ek, ev)) per rank
    allc = np.stack([e[0] for e in ext])  # [src][dst][4]
    slices, stats = [], []
    for r, sb in enumerate(sbs):
        rt, t1 = parallel.exchange_stream(ext[r][1], allc[:, :, 0:2], r, n_shards, peers=[e[1] for e in ext])
        re_, e1 = parallel.exchange_stream(ext[r][2], allc[:, :, 2:4], r, n_shards, peers=[e[2] for e in ext])
        sb.build(rt, t1, re_, e1, sp.eps)
        if regions is None:
            sl, st = sb.export()
        else:  # (what this owner sends to every rank; all selections before any import: an import replaces the handle's graph)
            sl, st = zip(*[sb.select(regions[d]) for d in range(n_shards)])
        slices.append(sl)
        stats.append(st)
    if regions is None:
        totals = [sb.import_all(slices, stats) for sb in sbs]
    else:
        totals = []
        for d, sb in enumerate(sbs):
            totals.append(sb.import_all([slices[o][d] for o in range(n_shards)], [stats[o][d] for o in range(n_shards)]))
            sb.set_region(regions[d])
    torch.cuda.synchronize()
    assert all(t.counts() == totals[0].counts() for t in totals)
    return gs, totals[0]


def _traverse(host, g, w, out, orient):
    import bench
    ref_np = w.ref.cpu().numpy()
    ctg_codes = w.contig_codes()
    ctg_seqs, k1 = bench.host_seqs(ctg_codes)
    ref_seqs, k2 = bench.host_seqs([ref_np])
    os.makedirs(out, exist_ok=True)
    ts = bench.TraverseStats()
    rc = host.pagh_traverse(g, w.spec.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, w.spec.threads, w.spec.eps, 50,
                            out.encode(), b"0_", 0, C.byref(ts))
    assert rc == 0, host.pagh_last_error()
    return ts, (ctg_seqs, ref_seqs, k1, k2)


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [2, 4, 8])
def test_n_shards_build_the_graph_of_one(n_shards, workdir):
    import torch
    import bench
    import biggen
    hip, host = bench.load_libs()
    _bind(hip)
    sp = biggen.BigSpec(seed=7, ref_len=1_500_000, n_reads=6000, read_span=4000, k=14, eps=10, ctg_len=300_000, gap_lo=300, gap_hi=3000,
                        rev_ctg_frac=0.3, threads=16, cov=2, solid_min_abundance=2, chunk_reads=512)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    g1 = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    st1 = pagctl.BuildStats()
    assert hip.pag_process(C.c_void_p(g1), C.byref(inp), C.byref(st1)) == 0, hip.pag_last_error()
    sizes1, csr1 = _csr(hip, g1)

    gs, tot = _sharded(hip, w, n_shards)
    assert tot.counts() == st1.counts()
    assert tuple(tot.n_tuples) == tuple(st1.n_tuples) and tuple(tot.n_edges) == tuple(st1.n_edges)
    assert (tot.n_nodes, tot.n_pos, tot.n_uniq_edges) == (st1.n_nodes, st1.n_pos, st1.n_uniq_edges)
    for g in (gs[0], gs[-1]):
        sizes, csr = _csr(hip, g)
        assert sizes == sizes1
        for kk in csr1:
            assert np.array_equal(csr[kk], csr1[kk]), f"{n_shards} shards: CSR array {kk} differs"

    # traversal: one handle walks everything ...
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    ts1, _ = _traverse(host, g1, w, str(workdir / f"sh{n_shards}_one"), orient)
    tsN, seqs = _traverse(host, gs[0], w, str(workdir / f"sh{n_shards}_all"), orient)
    assert (ts1.n_path_nodes, ts1.n_path_bases, ts1.path_checksum) == (tsN.n_path_nodes, tsN.n_path_bases, tsN.path_checksum)
    # ... or the contigs are dealt out over the "ranks", each walks its own on its copy of the graph, and the travel
    # sequences are gathered for one chain selection
    n_ctg = len(w.ctgs)
    hip.pag_travel_path_oriented.restype = C.c_void_p
    hip.pag_travel_path_oriented.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    paths = (C.c_void_p * (2 * n_ctg))()
    lens = (C.c_uint64 * (2 * n_ctg))()
    keep = []

    class TravelParams(C.Structure):
        _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                    ("start_split", C.c_double), ("min_len", C.c_uint64)]
    prm = TravelParams(sp.threads, 0, 2 * sp.eps, 0.15, 0.90, 50)
    ctg_seqs, ref_seqs = seqs[0], seqs[1]
    ref_len = np.array([len(w.ref)], dtype=np.uint32)
    hip.pag_travel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    for r, g in enumerate(gs):
        mine = np.array([orient[c] if c % n_shards == r else -1 for c in range(n_ctg)], dtype=np.int32)
        assert hip.pag_travel(g, C.byref(ctg_seqs), mine.ctypes.data, ref_len.ctypes.data, 1, C.byref(prm), None) == 0, hip.pag_last_error()
        for c in range(n_ctg):
            if mine[c] < 0:
                continue
            n = C.c_uint64()
            p = hip.pag_travel_path_oriented(g, c, int(mine[c] != 0), C.byref(n))
            buf = C.create_string_buffer(C.string_at(p, n.value * 24), n.value * 24) if n.value else None  # (the "gather")
            keep.append(buf)
            slot = 2 * c + (0 if mine[c] else 1)
            paths[slot] = C.cast(buf, C.c_void_p).value if buf is not None else None
            lens[slot] = n.value
    out = str(workdir / f"sh{n_shards}_gathered")
    os.makedirs(out, exist_ok=True)
    tsG = bench.TraverseStats()
    host.pagh_assemble_paths.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p]
    rc = host.pagh_assemble_paths(None, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, paths, lens, sp.threads, sp.eps, 50,
                                  out.encode(), b"0_", 0, C.byref(tsG))
    assert rc == 0, host.pagh_last_error()
    one = str(workdir / f"sh{n_shards}_one")
    for d in (str(workdir / f"sh{n_shards}_all"), out):
        assert sorted(os.listdir(d)) == sorted(os.listdir(one))
        for f in os.listdir(one):
            assert open(os.path.join(d, f), "rb").read() == open(os.path.join(one, f), "rb").read(), f"{d}: {f}"
    assert (tsG.n_path_nodes, tsG.n_path_bases, tsG.path_checksum) == (ts1.n_path_nodes, ts1.n_path_bases, ts1.path_checksum)
    for g in gs + [g1]:
        hip.pag_destroy(C.c_void_p(g))


def _walk_dealt_and_assemble(hip, host, gs, w, deal, orient, seqs, out):
    """every "rank" walks the contigs it was dealt on ITS handle; the travel sequences are gathered for one chain selection"""
    import bench
    sp = w.spec
    n_ctg = len(w.ctgs)
    hip.pag_travel_path_oriented.restype = C.c_void_p
    hip.pag_travel_path_oriented.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    hip.pag_travel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    paths = (C.c_void_p * (2 * n_ctg))()
    lens = (C.c_uint64 * (2 * n_ctg))()
    keep = []

    class TravelParams(C.Structure):
        _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                    ("start_split", C.c_double), ("min_len", C.c_uint64)]
    prm = TravelParams(sp.threads, 0, 2 * sp.eps, 0.15, 0.90, 50)
    ctg_seqs, ref_seqs = seqs[0], seqs[1]
    ref_len = np.array([len(w.ref)], dtype=np.uint32)
    for r, g in enumerate(gs):
        mine = np.array([orient[c] if c in deal[r] else -1 for c in range(n_ctg)], dtype=np.int32)
        rc = hip.pag_travel(g, C.byref(ctg_seqs), mine.ctypes.data, ref_len.ctypes.data, 1, C.byref(prm), None)
        if rc != 0:
            return rc, hip.pag_last_error().decode()
        for c in range(n_ctg):
            if mine[c] < 0:
                continue
            n = C.c_uint64()
            p = hip.pag_travel_path_oriented(g, c, int(mine[c] != 0), C.byref(n))
            buf = C.create_string_buffer(C.string_at(p, n.value * 24), n.value * 24) if n.value else None  # (the "gather")
            keep.append(buf)
            slot = 2 * c + (0 if mine[c] else 1)
            paths[slot] = C.cast(buf, C.c_void_p).value if buf is not None else None
            lens[slot] = n.value
    os.makedirs(out, exist_ok=True)
    tsG = bench.TraverseStats()
    host.pagh_assemble_paths.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p]
    rc = host.pagh_assemble_paths(None, sp.k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient.ctypes.data, paths, lens, sp.threads, sp.eps, 50,
                                  out.encode(), b"0_", 0, C.byref(tsG))
    assert rc == 0, host.pagh_last_error()
    return 0, tsG


@pytest.mark.gpu
@pytest.mark.parametrize("n_shards", [2, 4])
def test_ranks_hold_their_region_only_and_walk_the_same_paths(n_shards, workdir):
    """The traversal side partitioned (BASELINE configs[2] does not fit one GPU): every "rank" imports, from every k-mer owner,
    only what pag_shard_select gives it for its region — its contigs' strands, the landing zones, the coordinate-free
    vertices of its reference bands — holds a fraction of the vertices, and the gathered outputs equal the one-GPU run byte
    for byte.  With NO halo the walks of the leaping zones leave the bands: that must be REPORTED (PAG_ERANGE), never a
    silently different path."""
    import torch
    import bench
    import biggen
    hip, host = bench.load_libs()
    _bind(hip)
    sp = biggen.BigSpec(seed=11, ref_len=3_000_000, n_reads=12000, read_span=4000, k=14, eps=10, ctg_len=250_000, gap_lo=300, gap_hi=3000,
                        rev_ctg_frac=0.3, threads=16, cov=2, solid_min_abundance=2, chunk_reads=512)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()
    err = C.c_int()
    g1 = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
    st1 = pagctl.BuildStats()
    assert hip.pag_process(C.c_void_p(g1), C.byref(inp), C.byref(st1)) == 0, hip.pag_last_error()
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    one = str(workdir / f"reg{n_shards}_one")
    ts1, seqs = _traverse(host, g1, w, one, orient)
    host.pagh_release(C.c_void_p(g1))
    hip.pag_destroy(C.c_void_p(g1))

    ctg_len = [e - s for s, e, _ in w.ctgs]
    g2r = w.g2r.cpu().numpy()
    deal = parallel.deal_contigs(ctg_len, n_shards, ref_begin=[int(g2r[s]) for s, _, _ in w.ctgs])
    alns = [(c, 0, int(g2r[s]), int(g2r[e - 1]) + 1) for c, (s, e, _) in enumerate(w.ctgs)]
    for halo, must_work in ((60_000, True), (0, False), ("hole", False)):
        if halo == "hole":
            # a band that does NOT cover where the rank's own contigs map (a contig whose reads also map elsewhere, a wrong
            # alignment list): vertices kept through their contig coordinate have a reference coordinate in the hole, their
            # coordinate-free successors are on no band of this rank — reported, or no path differs
            regions = parallel.regions_for(deal, ctg_len, orient, alns, [len(w.ref)], halo=60_000)
            for d in regions:
                iv = d["ref_iv"].reshape(-1, 2)
                lo, hi = int(iv[0][0]), int(iv[0][1])
                a, b = lo + (hi - lo) // 3, lo + (hi - lo) // 3 + 150_000
                d["ref_iv"] = np.array([[lo, a], [b, hi]] + [list(x) for x in iv[1:]], dtype=np.uint32).reshape(-1)
                d["ref_open"] = np.array([int(d["ref_open"][0]), 1, 1] + [int(x) for x in d["ref_open"][1:]], dtype=np.uint8)
                d["region"] = parallel.Region(len(d["ctg_iv"]) // 2, d["ctg_iv"].ctypes.data, len(d["ref_iv"]) // 2, d["ref_iv"].ctypes.data,
                                              d["ref_open"].ctypes.data)
        else:
            regions = parallel.regions_for(deal, ctg_len, orient, alns, [len(w.ref)], halo=halo)
        gs, tot = _sharded(hip, w, n_shards, regions=regions)
        assert tot.counts() == st1.counts()  # (the count lines stay the block's)
        held = []
        for g in gs:
            nn, npos, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
            hip.pag_csr_sizes(C.c_void_p(g), C.byref(nn), C.byref(npos), C.byref(ne))
            held.append(npos.value / st1.n_pos)
        out = str(workdir / f"reg{n_shards}_halo{halo}")
        if halo == "hole":
            assert max(held) < 1.0  # (something was left out)
        rc, res = _walk_dealt_and_assemble(hip, host, gs, w, deal, orient, seqs, out)
        for g in gs:
            hip.pag_destroy(C.c_void_p(g))
        if must_work:
            assert rc == 0, res
            # a fraction of the graph per rank: its share of the contigs + the landing zones + the halo of its bands
            assert max(held) < 1.0 / n_shards + 0.15, held
            assert sum(held) < 1.0 + 0.15 * n_shards, held
            print(f"{n_shards} ranks hold {[round(h, 3) for h in held]} of the vertices")
        else:
            assert rc in (0, -34), res  # PAG_ERANGE: reported, never silently different
            if rc != 0:
                assert "left the region" in res
                continue
            assert halo != "hole", "a walk over vertices whose reference neighbourhood this rank does not hold went unreported"
        for f in os.listdir(one):
            assert open(os.path.join(out, f), "rb").read() == open(os.path.join(one, f), "rb").read(), f"halo {halo}: {f}"
        assert sorted(os.listdir(out)) == sorted(os.listdir(one))
        assert (res.n_path_nodes, res.n_path_bases, res.path_checksum) == (ts1.n_path_nodes, ts1.n_path_bases, ts1.path_checksum)


@pytest.mark.gpu
def test_bench_shard_mode_two_processes_equal_one(workdir):
    """bench.py --mode shard with TWO processes (one device, gloo staged through the host: the PAG_BENCH_SINGLE_DEVICE hook) —
    the real multi-process control path: all_to_all_single of the tuples, all-gather of the slices, contigs dealt out,
    paths gathered, chains selected on rank 0 — against the one-process run of the same block: same graph, same paths."""
    import json
    import subprocess
    size = ["--reads", "3000", "--ref-len", "3000000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "bench.py")] + size, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    env = dict(os.environ, PAG_BENCH_SINGLE_DEVICE="1")
    port = 29600 + os.getpid() % 300
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(pagctl.ROOT, "bench.py"), "--gpus", "2", "--mode", "shard"] + size,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong"
    for kk in ("position_tuples", "edge_tuples", "path_nodes", "path_bases", "path_checksum", "chains"):
        assert d1["config"][kk] == d2["config"][kk], kk
    # (rank 0 holds its region of the graph, not the whole of it)
    assert d2["config"]["vertices"] < d1["config"]["vertices"]


@pytest.mark.gpu
def test_library_communicator_over_rccl_with_itself():
    """The RCCL transport of the library's communicator (grouped ncclSend / ncclRecv of device buffers) cannot be run between
    two ranks on a one-GPU box; a world of ONE exercises every RCCL call it makes (library loaded at run time, unique id
    through the rendezvous directory, communicator, a grouped send / receive with itself)."""
    import tempfile
    import torch
    hip = pagctl.hip_lib()
    hip.pag_comm_create.restype = C.c_void_p
    hip.pag_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    hip.pag_comm_all_to_all_v.argtypes = [C.c_void_p] * 5
    hip.pag_comm_destroy.argtypes = [C.c_void_p]
    os.environ["PAG_COMM_FORCE_RCCL"] = "1"
    try:
        d = tempfile.mkdtemp(prefix="pagcomm_")
        err = C.c_int()
        c = hip.pag_comm_create(0, 1, d.encode(), 0, b"rccl", C.byref(err))
        assert c, hip.pag_last_error()
        src = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
        dst = torch.zeros_like(src)
        n = (C.c_uint64 * 1)(src.numel() * 4)
        assert hip.pag_comm_all_to_all_v(c, src.data_ptr(), n, dst.data_ptr(), n) == 0, hip.pag_last_error()
        torch.cuda.synchronize()
        assert torch.equal(src, dst)
        hip.pag_comm_destroy(c)
    finally:
        os.environ.pop("PAG_COMM_FORCE_RCCL", None)


@pytest.mark.gpu
def test_bench_default_mode_two_processes(workdir):
    """bench.py as the driver launches it for the scaling runs (torch.distributed.run, --gpus N, the default mode: one block
    per rank, no data-path collective) with TWO processes on the one device of the box: the line must be the whole job's —
    twice the bases of one rank over the slowest rank's time — and rank 0's block must be the block a single process builds."""
    import json
    import subprocess
    size = ["--reads", "3000", "--ref-len", "3000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, os.path.join(pagctl.ROOT, "bench.py")] + size, capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    env = dict(os.environ, PAG_BENCH_SINGLE_DEVICE="1")
    port = 29300 + os.getpid() % 300
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(pagctl.ROOT, "bench.py"), "--gpus", "2"] + size,
                         capture_output=True, text=True, timeout=900, env=env)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["steps"] == 2 and d2["metric"] == d1["metric"]
    for kk in ("position_tuples", "edge_tuples", "path_nodes", "path_bases", "path_checksum", "chains", "vertices"):
        assert d1["config"][kk] == d2["config"][kk], kk   # (rank 0 runs the seed a single process runs)
    per_step_bases = d2["value"] * d2["ms_per_step"] / 1e3
    one_step_bases = d1["value"] * d1["ms_per_step"] / 1e3
    assert 1.8 * one_step_bases < per_step_bases < 2.2 * one_step_bases   # (rank 1 has its own seed: about as many bases)
    assert d2["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_rank_serial_run_equals_the_one_gpu_run(n_ranks, workdir):
    """aligngraph2_amd/rank_serial.py (the N ranks of the sharded build one after the other on ONE device, what a rank takes in
    recomputed when its turn comes — how BASELINE configs[2] is executed on the one GPU this build has): outputs, count lines and
    path statistics of the one-GPU run of the same block; every rank holds its share of the vertices."""
    import torch
    import bench
    import biggen
    from aligngraph2_amd import rank_serial
    hip, host = bench.load_libs()
    _bind(hip)
    sp = biggen.BigSpec(seed=13, ref_len=3_000_000, n_reads=12000, read_span=4000, k=14, eps=10, ctg_len=250_000, gap_lo=300, gap_hi=3000,
                        rev_ctg_frac=0.3, threads=16, cov=2, solid_min_abundance=2, chunk_reads=512)
    w = biggen.BigWorkload(sp, device="cuda")
    torch.cuda.synchronize()
    inp = w.build_input()

    def make_handle():
        err = C.c_int()
        g = hip.pag_create_from_bitmap(w.solid_bits.data_ptr(), w.n_solid, sp.k, 1, 0, C.byref(err))
        assert g, hip.pag_last_error()
        return g

    g1 = make_handle()
    st1 = pagctl.BuildStats()
    assert hip.pag_process(C.c_void_p(g1), C.byref(inp), C.byref(st1)) == 0, hip.pag_last_error()
    orient = np.array([0 if r else 1 for _, _, r in w.ctgs], dtype=np.int32)
    one = str(workdir / f"rs{n_ranks}_one")
    ts1, seqs = _traverse(host, g1, w, one, orient)
    host.pagh_release(C.c_void_p(g1))
    hip.pag_destroy(C.c_void_p(g1))
    want, _ = rank_serial.digest_dir(one)

    ctg_len = [e - s for s, e, _ in w.ctgs]
    g2r = w.g2r.cpu().numpy()
    alns = [(c, 0, int(g2r[s]), int(g2r[e - 1]) + 1) for c, (s, e, _) in enumerate(w.ctgs)]
    out = str(workdir / f"rs{n_ranks}_serial")
    res = rank_serial.run(hip, host, make_handle, inp, n_ranks=n_ranks, eps=sp.eps, k=sp.k, threads=sp.threads, ctgs=ctg_len, ctg_alns=alns,
                          ref_lens=[len(w.ref)], ctg_seqs=seqs[0], ref_seqs=seqs[1], orient=[int(x) for x in orient], out_dir=out, device="cuda:0",
                          halo=60_000)
    assert res["outputs_sha256"] == want
    assert res["count_lines_sum_over_owners"] == list(st1.counts())
    assert (res["path_nodes"], res["path_bases"]) == (ts1.n_path_nodes, ts1.n_path_bases)
    assert max(r["held_fraction"] for r in res["ranks"]) < 1.0 / n_ranks + 0.15
    assert all(r["bytes_extract"] > 0 and r["bytes_traversal_peak"] > 0 for r in res["ranks"])


def _comm_worker(rank, world, rdv, q, fail_rank):
    """a rank of a two-process job on the one device ("host" transport): rank `fail_rank` aborts instead of joining the exchange"""
    import torch
    hip = pagctl.hip_lib()
    hip.pag_comm_create.restype = C.c_void_p
    hip.pag_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.POINTER(C.c_int)]
    hip.pag_comm_all_to_all_v.argtypes = [C.c_void_p] * 5
    hip.pag_comm_abort.argtypes = [C.c_void_p, C.c_char_p]
    hip.pag_comm_destroy.argtypes = [C.c_void_p]
    err = C.c_int()
    c = hip.pag_comm_create(rank, world, rdv.encode(), 0, b"host", C.byref(err))
    if not c:
        q.put((rank, "create", err.value, hip.pag_last_error().decode()))
        return
    import time
    t0 = time.time()
    if rank == fail_rank:
        hip.pag_comm_abort(c, b"rank %d ran out of luck" % rank)
        q.put((rank, "aborted", 0, ""))
    else:
        n = 1 << 16
        src = torch.full((world * n,), rank + 1, dtype=torch.int32, device="cuda")
        dst = torch.zeros_like(src)
        sizes = (C.c_uint64 * world)(*([n * 4] * world))
        rc = hip.pag_comm_all_to_all_v(c, src.data_ptr(), sizes, dst.data_ptr(), sizes)
        torch.cuda.synchronize()
        ok = rc == 0 and all(int(dst[r * n]) == r + 1 and int(dst[(r + 1) * n - 1]) == r + 1 for r in range(world))
        q.put((rank, "exchanged" if ok else "failed", time.time() - t0, hip.pag_last_error().decode() if rc else ""))
    hip.pag_comm_destroy(c)


@pytest.mark.gpu
def test_library_communicator_between_two_processes_and_a_failing_rank():
    """pag_comm between TWO processes that share the box's one device (transport "host"): the job nonce agreed by live ranks
    in a directory that already holds another job's files; an all-to-all(v) of device buffers; and a rank that fails: its peer
    returns at once with the failing rank's message instead of waiting for PAG_COMM_TIMEOUT_S."""
    import tempfile
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    rdv = tempfile.mkdtemp(prefix="pagcomm2_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    # what a crashed job left behind: a hello of a dead process, an answer for it, files of its collectives
    open(os.path.join(rdv, "hello_1"), "wb").write((2 ** 22 + 12345).to_bytes(8, "little") + (77).to_bytes(8, "little") + (99).to_bytes(8, "little"))
    open(os.path.join(rdv, "job_1"), "wb").write((4242).to_bytes(8, "little") + (99).to_bytes(8, "little"))
    open(os.path.join(rdv, "j0000000000001092_g0_0"), "wb").write(b"\0")
    os.environ["PAG_COMM_TIMEOUT_S"] = "60"
    try:
        for fail_rank in (-1, 1):
            q = ctx.Queue()
            ps = [ctx.Process(target=_comm_worker, args=(r, 2, rdv, q, fail_rank)) for r in range(2)]
            for p in ps:
                p.start()
            got = dict()
            for _ in ps:
                r, what, x, msg = q.get(timeout=120)
                got[r] = (what, x, msg)
            for p in ps:
                p.join(timeout=60)
            if fail_rank < 0:
                assert got[0][0] == "exchanged" and got[1][0] == "exchanged", got
            else:
                assert got[1][0] == "aborted"
                what, secs, msg = got[0]
                assert what == "failed" and "ran out of luck" in msg and secs < 20, got  # (not the 60 s timeout)
    finally:
        os.environ.pop("PAG_COMM_TIMEOUT_S", None)
