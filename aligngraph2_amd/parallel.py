"""Multi-GPU layer of the PAGraph hot path: one process per GPU, torch.distributed (RCCL on GPUs, gloo
in CPU tests).  Two levels:

Level 1 — blocks over GPUs (run_sharded, run_config_blocks): the path shards by reference sequence; AlignGraph2 already
runs one pagraph per per-reference sub-directory, sequentially (reference AlignGraph2.py:399-431), and config blocks are
independent after resetAllNodes (pagraph.cpp:181-182).  The units are distributed over the ranks with NO data-path
collective; the only communication is the barrier / max / sum around the timed region and the gathering of exit codes.

Level 2 — ONE block over GPUs (ShardedBuild, build_sharded, regions_for, deal_contigs, gather_paths): reads are split for
the extraction, k-mer ranges for sort / cluster / edges with one all-to-all(v) of tuples in between; the traversal side is
partitioned by the contigs a rank is dealt: every rank receives, from every k-mer owner, only the vertices its traversals
can examine (pag_shard_select: its contigs' strands, the landing zones of all contigs, the coordinate-free vertices of the
reference bands its contigs map to) — one more all-to-all(v) instead of an all-gather of the whole graph.
"""
from __future__ import annotations

import os
import subprocess
from typing import Callable, List, Sequence


def assign_blocks(sizes: Sequence[int], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of blocks (by input size) to ranks.
    Deterministic: ties broken by block index.  Returns, per rank, the block indices in run order."""
    order = sorted(range(len(sizes)), key=lambda i: (-sizes[i], i))
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda x: (load[x], x))
        out[r].append(i)
        load[r] += sizes[i]
    return out


def init(backend: str | None = None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for world 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    return dist


def aggregate(dist, seconds: float, units: float, device: str = "cpu"):
    """(max over ranks of seconds, sum over ranks of units): whole-job time and work."""
    if dist is None:
        return seconds, units
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def run_sharded(sub_dirs: Sequence[str], sizes: Sequence[int], run_one: Callable[[str, int], int], dist=None) -> List[int]:
    """Run `run_one(sub_dir, local_rank)` for every sub-directory, blocks spread over the ranks.
    Returns the exit codes of ALL blocks on every rank (gathered), in sub_dirs order."""
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    mine = assign_blocks(sizes, world)[rank]
    codes = {i: run_one(sub_dirs[i], local) for i in mine}
    if dist is None:
        return [codes[i] for i in range(len(sub_dirs))]
    gathered: list = [None] * world
    dist.all_gather_object(gathered, codes)
    merged = {}
    for g in gathered:
        merged.update(g)
    return [merged[i] for i in range(len(sub_dirs))]


def pagraph_runner(argv_for_dir: Callable[[str], List[str]]):
    """run_one() for the drop-in executable: one pagraph process per sub-directory on this rank's GPU."""
    from . import PAGRAPH, require_built

    def run_one(sub_dir: str, local_rank: int) -> int:
        require_built()
        env = dict(os.environ, PAGRAPH_DEVICE=str(local_rank))
        return subprocess.run([PAGRAPH, *argv_for_dir(sub_dir)], env=env).returncode

    return run_one


def read_config_blocks(pre_dir: str):
    """config.txt as pre_process writes it (reference pre_process.cpp:271-287, read by pagraph.cpp:29-49): per block the
    reference name, the read file, the two alignment files (relative to pre_dir), then (contig, 1|0) line pairs up to an
    empty line.  Returns [(ref, read_file, ctg_aln, ref_aln, [(contig, forward)])]."""
    lines = open(os.path.join(pre_dir, "config.txt")).read().split("\n")
    blocks, i = [], 0
    while i + 3 < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        ref, reads, caln, raln = (lines[i + j].strip() for j in range(4))
        i += 4
        ctgs = []
        while i + 1 < len(lines) and lines[i].strip():
            ctgs.append((lines[i].strip(), lines[i + 1].strip() != "0"))
            i += 2
        blocks.append((ref, reads, caln, raln, ctgs))
        i += 1
    return blocks


def run_config_blocks(pre_dir: str, out_dir: str, argv: Sequence[str], dist=None, exe: str | None = None) -> List[int]:
    """The per-reference blocks of ONE pre_process output directory (the scheduler consumes pre_process'
    config.txt directly, no per-directory loop in Python) spread over the GPUs of the node: blocks are weighed by the size of
    their read + alignment files, dealt out longest-first (assign_blocks), and every rank runs the drop-in executable ONCE on
    its GPU for its blocks (PAGRAPH_BLOCKS; output files keep the block's number as prefix, exactly as one process over all
    blocks would write them).  Rank 0 merges the per-rank shares of contig.txt — a SET of names: the reference writes it in
    std::unordered_set iteration order (quirk Q11) and its only consumer reads it as a set (script/extract.py:11-13), so the
    merged file lists the names in rank order of first appearance; compare it sorted.  argv: the pagraph arguments (flags as
    AlignGraph2.py passes them, -p pre_dir -o out_dir included).  Returns the exit codes of all ranks."""
    from . import PAGRAPH, require_built
    require_built()
    blocks = read_config_blocks(pre_dir)
    sizes = []
    for _, reads, caln, raln, _ in blocks:
        sizes.append(sum(os.path.getsize(os.path.join(pre_dir, f)) for f in (reads, caln, raln) if os.path.exists(os.path.join(pre_dir, f))))
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    mine = sorted(assign_blocks(sizes, world)[rank])
    rc = 0
    if mine:
        env = dict(os.environ, PAGRAPH_DEVICE=str(local), PAGRAPH_BLOCKS=",".join(str(b) for b in mine), PAGRAPH_PART=str(rank))
        rc = subprocess.run([exe or PAGRAPH, *argv], env=env).returncode
    codes = [rc]
    if dist is not None:
        gathered: list = [None] * world
        dist.all_gather_object(gathered, rc)
        codes = list(gathered)
    if rank == 0:
        names = []
        for r in range(world):
            part = os.path.join(out_dir, f"contig.txt.part{r}")
            if os.path.exists(part):
                names += open(part).read().split()
                os.remove(part)
        with open(os.path.join(out_dir, "contig.txt"), "w") as f:
            for n in dict.fromkeys(names):
                f.write(n + "\n")
    if dist is not None:
        dist.barrier()
    return codes


# ======================================================================================================
# ONE graph built by several GPUs (SURVEY.md §8e level 2; C side: include/pagraph_hip.h, pag_shard_*)
# ======================================================================================================
# Reads are split over the ranks for the extraction, k-mer ranges for everything after it; in between ONE all-to-all(v) of
# 12-byte tuples over xGMI (RCCL all_to_all_single with split sizes), afterwards ONE all-gather of the owners' finished
# slices.  Order argument (why this is bit-identical to one GPU): the canonical order of the tuples of a k-mer is [pass 1
# in emission order] ++ [pass 2 in emission order]; rank r extracts the r-th contiguous range of the emission order, so
# the owner of a k-mer restores that order by laying what it receives out as [pass 1 from rank 0] .. [pass 1 from rank
# N-1] [pass 2 from rank 0] .. [pass 2 from rank N-1] and sorting STABLY by k-mer (pag_shard_build).  No sequence numbers
# travel.  Wire volume per GPU: (N-1)/N of its own tuples out (12 B each, two streams) + (N-1)/N of the finished graph in
# (18 B per tuple slot + 16 B per edge slot): for BASELINE configs[2] on 4 GPUs (1.08e10 tuples + as many edges) about
# 49 GB out and 69 GB in per GPU, i.e. ~0.11 s + ~0.15 s at 3 peer links x 153 GB/s if all links are driven at once.

import ctypes as _C


class BuildStats(_C.Structure):
    """pag_build_stats (include/pagraph_hip.h)"""
    _fields_ = [("merge_edge", _C.c_uint64 * 2), ("total_pos", _C.c_uint64 * 2), ("merge_pos", _C.c_uint64 * 2),
                ("n_tuples", _C.c_uint64 * 2), ("n_edges", _C.c_uint64 * 2), ("n_nodes", _C.c_uint64),
                ("n_pos", _C.c_uint64), ("n_uniq_edges", _C.c_uint64), ("ms_extract", _C.c_double),
                ("ms_sort", _C.c_double), ("ms_cluster", _C.c_double), ("ms_edges", _C.c_double),
                ("ms_total", _C.c_double), ("ms_sort_kernel", _C.c_double), ("sort_records", _C.c_uint64)]

    def counts(self):
        return (self.merge_edge[0], self.total_pos[0], self.merge_pos[0], self.merge_edge[1], self.total_pos[1],
                self.merge_pos[1])


class ShardSlice(_C.Structure):
    """pag_shard_slice (include/pagraph_hip.h)"""
    _fields_ = [("n_t", _C.c_uint64), ("n_e", _C.c_uint64), ("tkey", _C.c_void_p), ("tval", _C.c_void_p), ("tseg", _C.c_void_p),
                ("tcnt", _C.c_void_p), ("ekey", _C.c_void_p), ("eval", _C.c_void_p), ("eseg", _C.c_void_p), ("stats", BuildStats)]


class Region(_C.Structure):
    """pag_region (include/pagraph_hip.h)"""
    _fields_ = [("n_ctg_iv", _C.c_uint64), ("ctg_iv", _C.c_void_p), ("n_ref_iv", _C.c_uint64), ("ref_iv", _C.c_void_p), ("ref_open", _C.c_void_p)]


def bind_shard_api(hip):
    """ctypes signatures of the pag_shard_* entry points (include/pagraph_hip.h)"""
    vp, u64 = _C.c_void_p, _C.c_uint64
    hip.pag_shard_extract.argtypes = [vp, vp, _C.c_uint32, _C.c_uint32, _C.POINTER(u64)]
    hip.pag_shard_take.argtypes = [vp, vp, vp, vp, vp]
    hip.pag_shard_take_part.argtypes = [vp, _C.c_uint64, _C.c_uint64, vp, vp, _C.c_uint64, _C.c_uint64, vp, vp]
    hip.pag_shard_build.argtypes = [vp, vp, vp, u64, u64, vp, vp, u64, u64, _C.c_uint32, _C.POINTER(BuildStats)]
    hip.pag_shard_export.argtypes = [vp, _C.POINTER(ShardSlice)]
    hip.pag_shard_take_slice.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    hip.pag_shard_import.argtypes = [vp, _C.POINTER(ShardSlice), _C.c_uint32, _C.POINTER(BuildStats)]
    hip.pag_shard_select.argtypes = [vp, _C.POINTER(Region), _C.POINTER(ShardSlice)]
    hip.pag_shard_set_region.argtypes = [vp, _C.POINTER(Region)]
    hip.pag_shard_release_build.argtypes = [vp]
    for f in ("pag_shard_extract", "pag_shard_take", "pag_shard_take_part", "pag_shard_build", "pag_shard_export", "pag_shard_take_slice", "pag_shard_import",
              "pag_shard_select", "pag_shard_set_region"):
        getattr(hip, f).restype = _C.c_int
    hip.pag_last_error.restype = _C.c_char_p


def mapper_starts(lengths):
    """PositionMapper layout (position/PositionMapper.cpp:16-31): start of every sequence's forward strand + the extra end"""
    st = []
    for i, n in enumerate(lengths):
        st.append(int(n) if i == 0 else st[-1] + 3 * int(lengths[i - 1]) + max(int(lengths[i - 1]), int(n)))
    if st:
        st.append(st[-1] + 4 * int(lengths[-1]))
    return st


def _merge(ivs):
    out = []
    for lo, hi in sorted(ivs):
        if hi <= lo:
            continue
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return out


def regions_for(deal, ctg_len, orient, ctg_alns, ref_len, halo=200_000, start_split=0.90):
    """What every rank of a sharded build needs of the finished graph (see pag_shard_select in include/pagraph_hip.h).

    deal[r]: the contigs rank r traverses; orient[c]: PAG_ORIENT_* (-1 none, 0 reverse, 1 forward, 2 both) as config.txt
    lists them; ctg_alns: the contig->reference alignments as (contig, reference index, t_begin, t_end) of every listed
    alignment (whatever its strand: a superset costs memory, never correctness); ref_len: reference lengths.
    Returns, per rank, dict(ctg_iv, ref_iv, ref_open) of numpy arrays + a keep-alive Region()."""
    import numpy as np
    cst, rst = mapper_starts(ctg_len), mapper_starts(ref_len)
    leap_min = 1.0 - start_split
    # landing zones: the first leap_min of EVERY contig strand (offset <= len * leap_min survives the leap rule)
    landing = []
    for c, n in enumerate(ctg_len):
        z = int(float(n) * leap_min) + 2
        z = min(z, int(n))
        landing += [(cst[c], cst[c] + z), (cst[c] + 2 * int(n), cst[c] + 2 * int(n) + z)]
    out = []
    for mine in deal:
        civ, riv = list(landing), []
        for c in mine:
            n = int(ctg_len[c])
            if orient[c] in (1, 2):
                civ.append((cst[c], cst[c] + n))
            if orient[c] in (0, 2):
                civ.append((cst[c] + 2 * n, cst[c] + 3 * n))
            for (cc, ri, tb, te) in ctg_alns:
                if cc != c:
                    continue
                lo, hi = rst[ri], rst[ri] + int(ref_len[ri])
                riv.append((max(lo, rst[ri] + int(tb) - halo), min(hi, rst[ri] + int(te) + halo)))
        civ, riv = _merge(civ), _merge(riv)
        ref_ends = {}
        for ri, n in enumerate(ref_len):
            ref_ends[rst[ri]] = 0
            ref_ends[rst[ri] + int(n)] = 0
        ropen = []
        for lo, hi in riv:
            ropen += [0 if lo in ref_ends else 1, 0 if hi in ref_ends else 1]
        d = dict(ctg_iv=np.array(civ, dtype=np.uint32).reshape(-1), ref_iv=np.array(riv, dtype=np.uint32).reshape(-1),
                 ref_open=np.array(ropen, dtype=np.uint8))
        d["region"] = Region(len(civ), d["ctg_iv"].ctypes.data, len(riv), d["ref_iv"].ctypes.data, d["ref_open"].ctypes.data)
        out.append(d)
    return out


def layout_received(chunks, counts_to_me):
    """The records an owner receives, in the order pag_shard_build expects.

    chunks[r]: what rank r sent to this owner = [its pass-1 records][its pass-2 records] (a 1-D tensor / array);
    counts_to_me[r] = (n pass 1, n pass 2).  Returns ([pass 1 from rank 0] .. [pass 1 from rank N-1] [pass 2 from rank 0] ..,
    number of pass-1 records)."""
    import torch
    p1 = [chunks[r][:int(counts_to_me[r][0])] for r in range(len(chunks))]
    p2 = [chunks[r][int(counts_to_me[r][0]):int(counts_to_me[r][0]) + int(counts_to_me[r][1])] for r in range(len(chunks))]
    return torch.cat(p1 + p2), int(sum(int(c[0]) for c in counts_to_me))


def _is_gloo(dist):
    try:
        return dist.get_backend() == "gloo"
    except Exception:
        return False


def exchange_stream(arrays, counts, rank, world, dist=None, peers=None):
    """One tuple stream (a tuple of parallel 1-D tensors, e.g. (keys u32-as-i32, values u64-as-i64)), partitioned by owner
    as pag_shard_extract leaves it, to its owners.

    counts: [world(src)][world(dst)][2] records of this stream per (source, owner, pass), known to every rank.
    dist: torch.distributed (all_to_all_single with split sizes: RCCL over xGMI on GPUs, gloo in the CPU test);
    peers: instead of dist, the partitioned streams of ALL ranks (a list indexed by rank) — the single-process emulation the
    shard tests use.  Returns (arrays for pag_shard_build, number of pass-1 records among them)."""
    import torch
    send_splits = [int(counts[rank][o][0] + counts[rank][o][1]) for o in range(world)]
    recv_splits = [int(counts[r][rank][0] + counts[r][rank][1]) for r in range(world)]
    out = []
    n1 = 0
    for ai, a in enumerate(arrays):
        if peers is not None:
            chunks = []
            for r in range(world):
                off = sum(int(counts[r][o][0] + counts[r][o][1]) for o in range(rank))
                chunks.append(peers[r][ai][off:off + recv_splits[r]])
        else:
            # (gloo moves host memory only: the single-device test hook of bench.py stages through the host; RCCL sends
            # device memory over xGMI directly)
            stage = a.device.type != "cpu" and _is_gloo(dist)
            src = a.contiguous().cpu() if stage else a.contiguous()
            recv = torch.empty(sum(recv_splits), dtype=a.dtype, device=src.device)
            dist.all_to_all_single(recv, src, recv_splits, send_splits)
            if stage:
                recv = recv.to(a.device)
            chunks = list(torch.split(recv, recv_splits))
        laid, n1 = layout_received(chunks, [counts[r][rank] for r in range(world)])
        out.append(laid.contiguous())
    return tuple(out), n1


class ShardedBuild:
    """The phases of a sharded graph build for ONE rank; the driver (build_sharded below, or the single-process emulation
    in tests/test_gpu_shards.py) runs them with the collectives in between."""

    def __init__(self, hip, g, inp, rank, world, device):
        import numpy as np
        import torch
        self.hip, self.g, self.inp, self.rank, self.world, self.device = hip, g, inp, rank, world, device
        self.np, self.torch = np, torch
        self.Slice = ShardSlice
        self.BuildStats = BuildStats

    def extract(self):
        """K1 for this rank's reads + stable partition by owner.  -> counts[world][4], (tkey, tval), (ekey, eval)"""
        np, torch = self.np, self.torch
        c = (_C.c_uint64 * (4 * self.world))()
        rc = self.hip.pag_shard_extract(_C.c_void_p(self.g), _C.byref(self.inp), self.rank, self.world, c)
        if rc != 0:
            raise RuntimeError(f"pag_shard_extract failed ({rc}): {self.hip.pag_last_error().decode()}")
        counts = np.array(list(c), dtype=np.int64).reshape(self.world, 4)
        T, E = int(counts[:, 0:2].sum()), int(counts[:, 2:4].sum())
        tk = torch.empty(T, dtype=torch.int32, device=self.device)
        tv = torch.empty(T, dtype=torch.int64, device=self.device)
        ek = torch.empty(E, dtype=torch.int32, device=self.device)
        ev = torch.empty(E, dtype=torch.int64, device=self.device)
        rc = self.hip.pag_shard_take(_C.c_void_p(self.g), _C.c_void_p(tk.data_ptr()), _C.c_void_p(tv.data_ptr()), _C.c_void_p(ek.data_ptr()),
                                     _C.c_void_p(ev.data_ptr()))
        if rc != 0:
            raise RuntimeError(f"pag_shard_take failed ({rc}): {self.hip.pag_last_error().decode()}")
        return counts, (tk, tv), (ek, ev)

    def build(self, tuples, t1, edges, e1, eps):
        """K2-K4 over the records this owner received -> its slice (kept in the handle) and its share of the counts"""
        st = self.BuildStats()
        (tk, tv), (ek, ev) = tuples, edges
        rc = self.hip.pag_shard_build(_C.c_void_p(self.g), _C.c_void_p(tk.data_ptr()), _C.c_void_p(tv.data_ptr()), tk.numel(), t1,
                                      _C.c_void_p(ek.data_ptr()), _C.c_void_p(ev.data_ptr()), ek.numel(), e1, eps, _C.byref(st))
        if rc != 0:
            raise RuntimeError(f"pag_shard_build failed ({rc}): {self.hip.pag_last_error().decode()}")
        return st

    def export(self):
        """the slice as torch tensors (copies: they outlive the handle's buffers through the all-gather)"""
        torch = self.torch
        sl = self.Slice()
        rc = self.hip.pag_shard_export(_C.c_void_p(self.g), _C.byref(sl))
        if rc != 0:
            raise RuntimeError("pag_shard_export failed")
        out = {}
        for name, n, dt in (("tkey", sl.n_t, torch.int32), ("tval", sl.n_t, torch.int64), ("tseg", sl.n_t, torch.int32),
                            ("tcnt", sl.n_t, torch.int16), ("ekey", sl.n_e, torch.int32), ("eval", sl.n_e, torch.int64),
                            ("eseg", sl.n_e, torch.int32)):
            out[name] = torch.empty(int(n), dtype=dt, device=self.device)
        rc = self.hip.pag_shard_take_slice(_C.c_void_p(self.g), *[_C.c_void_p(out[n].data_ptr()) for n in
                                                                  ("tkey", "tval", "tseg", "tcnt", "ekey", "eval", "eseg")])
        if rc != 0:
            raise RuntimeError(f"pag_shard_take_slice failed ({rc}): {self.hip.pag_last_error().decode()}")
        return out, BuildStats.from_buffer_copy(bytes(sl.stats))

    def _slice_tensors(self, sl):
        torch = self.torch
        out = {}
        for name, n, dt in (("tkey", sl.n_t, torch.int32), ("tval", sl.n_t, torch.int64), ("tseg", sl.n_t, torch.int32),
                            ("tcnt", sl.n_t, torch.int16), ("ekey", sl.n_e, torch.int32), ("eval", sl.n_e, torch.int64),
                            ("eseg", sl.n_e, torch.int32)):
            out[name] = torch.empty(int(n), dtype=dt, device=self.device)
        return out

    def select(self, region):
        """the part of this owner's slice inside `region` (regions_for) as torch tensors + the stats that go with it"""
        torch = self.torch
        sl = self.Slice()
        rc = self.hip.pag_shard_select(_C.c_void_p(self.g), _C.byref(region["region"]), _C.byref(sl))
        if rc != 0:
            raise RuntimeError(f"pag_shard_select failed ({rc}): {self.hip.pag_last_error().decode()}")
        out = self._slice_tensors(sl)
        rt = _hip_runtime()
        for name, t in out.items():
            if t.numel():
                rc = rt.hipMemcpy(_C.c_void_p(t.data_ptr()), _C.c_void_p(getattr(sl, name)), t.numel() * t.element_size(), 3)  # device to device
                if rc != 0:
                    raise RuntimeError(f"hipMemcpy failed ({rc})")
        return out, BuildStats.from_buffer_copy(bytes(sl.stats))

    def set_region(self, region):
        rc = self.hip.pag_shard_set_region(_C.c_void_p(self.g), _C.byref(region["region"]))
        if rc != 0:
            raise RuntimeError(f"pag_shard_set_region failed ({rc}): {self.hip.pag_last_error().decode()}")

    def import_all(self, slices, stats_list):
        """the whole graph from the slices of all owners (lists in owner order) -> total count lines"""
        parts = (self.Slice * len(slices))()
        for i, (sl, st) in enumerate(zip(slices, stats_list)):
            parts[i].n_t = sl["tkey"].numel()
            parts[i].n_e = sl["ekey"].numel()
            for name in ("tkey", "tval", "tseg", "tcnt", "ekey", "eval", "eseg"):
                setattr(parts[i], name, sl[name].data_ptr())
            parts[i].stats = st
        tot = self.BuildStats()
        rc = self.hip.pag_shard_import(_C.c_void_p(self.g), parts, len(slices), _C.byref(tot))
        if rc != 0:
            raise RuntimeError(f"pag_shard_import failed ({rc}): {self.hip.pag_last_error().decode()}")
        return tot


_rt = {}


def _hip_runtime():
    """the HIP runtime torch loaded (device-to-device copies out of the handle's buffers)"""
    if "rt" not in _rt:
        import torch
        rt = _C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        rt.hipMemcpy.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_size_t, _C.c_int]
        _rt["rt"] = rt
    return _rt["rt"]


def build_sharded(hip, g, inp, eps, dist, device, regions=None):
    """One config block built by all ranks of `dist` (each rank passes the SAME input; it extracts its share of the reads).
    regions = None: afterwards every rank's handle holds the whole graph (all-gather of the owners' slices).
    regions = regions_for(...): every rank receives, from every owner, the part of the graph its traversals need (one
    all-to-all(v)), and the handle knows its region.  Returns the total count lines (pag_build_stats)."""
    import numpy as np
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    cdev = "cpu" if _is_gloo(dist) else device  # where the collectives' buffers live

    def all_gather(t):
        # as bytes: neither RCCL nor gloo carries every dtype (the u16 abundances are int16 tensors here)
        src = t.to(cdev).contiguous().view(torch.uint8)
        outs = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(outs, src)
        return [o.view(t.dtype) for o in outs]

    sb = ShardedBuild(hip, g, inp, rank, world, device)
    counts, tuples, edges = sb.extract()
    allc = torch.stack(all_gather(torch.from_numpy(counts))).cpu().numpy()  # [src][dst][4]
    rt, t1 = exchange_stream(tuples, allc[:, :, 0:2], rank, world, dist=dist)
    re, e1 = exchange_stream(edges, allc[:, :, 2:4], rank, world, dist=dist)
    del tuples, edges
    sb.build(rt, t1, re, e1, eps)
    del rt, re
    if regions is not None:
        return _exchange_selected(sb, regions, dist, cdev, device, all_gather)
    sl, st = sb.export()
    # all-gather of the slices (padded to the largest; sizes first)
    all_sizes = [x.cpu().tolist() for x in all_gather(torch.tensor([sl["tkey"].numel(), sl["ekey"].numel()], dtype=torch.int64))]
    slices = [dict() for _ in range(world)]
    for name in ("tkey", "tval", "tseg", "tcnt", "ekey", "eval", "eseg"):
        which = 0 if name[0] == "t" else 1
        mx = max(s[which] for s in all_sizes)
        pad = torch.zeros(mx, dtype=sl[name].dtype, device=device)
        pad[:sl[name].numel()] = sl[name]
        outs = all_gather(pad)
        for r in range(world):
            slices[r][name] = outs[r][:all_sizes[r][which]].to(device).contiguous()
    all_st = all_gather(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))
    stats_list = [BuildStats.from_buffer_copy(bytes(x.cpu().numpy().tobytes())) for x in all_st]
    return sb.import_all(slices, stats_list)


def _exchange_selected(sb, regions, dist, cdev, device, all_gather):
    """every owner selects, for every rank, the part of its slice inside that rank's region; one all-to-all(v) per array"""
    import torch
    rank, world = sb.rank, sb.world
    sel, stats = [], []
    for d in range(world):
        t, st = sb.select(regions[d])
        sel.append(t)
        stats.append(st)
    names = ("tkey", "tval", "tseg", "tcnt", "ekey", "eval", "eseg")
    mine = torch.tensor([[sel[d]["tkey"].numel(), sel[d]["ekey"].numel()] for d in range(world)], dtype=torch.int64)
    sizes = torch.stack(all_gather(mine)).cpu().numpy()  # [owner][dest][2]
    slices = [dict() for _ in range(world)]
    for name in names:
        which = 0 if name[0] == "t" else 1
        send_splits = [int(sizes[rank][d][which]) for d in range(world)]
        recv_splits = [int(sizes[o][rank][which]) for o in range(world)]
        src = torch.cat([sel[d][name] for d in range(world)]).to(cdev).contiguous().view(torch.uint8)
        esz = sel[0][name].element_size()
        recv = torch.empty(sum(recv_splits) * esz, dtype=torch.uint8, device=cdev)
        dist.all_to_all_single(recv, src, [n * esz for n in recv_splits], [n * esz for n in send_splits])
        chunks = torch.split(recv, [n * esz for n in recv_splits])
        for o in range(world):
            slices[o][name] = chunks[o].to(device).contiguous().view(sel[0][name].dtype)
    # the owners' stats that go with their selections for this rank
    blob = torch.cat([torch.frombuffer(bytearray(bytes(stats[d])), dtype=torch.uint8) for d in range(world)]).to(cdev)
    n_st = len(bytes(stats[0]))
    rblob = torch.empty_like(blob)
    dist.all_to_all_single(rblob, blob, [n_st] * world, [n_st] * world)
    stats_list = [BuildStats.from_buffer_copy(bytes(rblob[o * n_st:(o + 1) * n_st].cpu().numpy().tobytes())) for o in range(world)]
    tot = sb.import_all(slices, stats_list)
    sb.set_region(regions[rank])
    # the build's device memory (and torch's copies of the exchanged arrays) goes back before the traversal needs it
    del sel, slices, src, recv, chunks
    sb.hip.pag_shard_release_build(_C.c_void_p(sb.g))
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return tot


def native_comm(hip, dist, device_ordinal, transport=None):
    """The library's own communicator (include/pagraph_hip.h: pag_comm) for the ranks of `dist`: a rendezvous directory is
    made by rank 0 and announced through torch.distributed; the bulk exchanges then run inside the library (RCCL over xGMI,
    or the "host" transport when the ranks share one device) — no torch tensors in the data path."""
    import tempfile
    rank, world = dist.get_rank(), dist.get_world_size()
    box = [tempfile.mkdtemp(prefix="pagshard_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    hip.pag_comm_create.restype = _C.c_void_p
    hip.pag_comm_create.argtypes = [_C.c_int, _C.c_int, _C.c_char_p, _C.c_int, _C.c_char_p, _C.POINTER(_C.c_int)]
    hip.pag_comm_destroy.argtypes = [_C.c_void_p]
    hip.pag_comm_bytes_sent.argtypes = [_C.c_void_p]
    hip.pag_comm_bytes_sent.restype = _C.c_uint64
    hip.pag_shard_run.argtypes = [_C.c_void_p, _C.c_void_p, _C.c_void_p, _C.c_void_p, _C.POINTER(BuildStats)]
    hip.pag_shard_run.restype = _C.c_int
    err = _C.c_int()
    c = hip.pag_comm_create(rank, world, box[0].encode(), device_ordinal, transport.encode() if transport else None, _C.byref(err))
    if not c:
        raise RuntimeError(f"pag_comm_create failed ({err.value}): {hip.pag_last_error().decode()}")
    return c, box[0]


def build_sharded_native(hip, g, comm, inp, regions):
    """build_sharded with the exchanges inside the library (pag_shard_run): regions = regions_for(...) of ALL ranks"""
    arr = (Region * len(regions))(*[r["region"] for r in regions])
    st = BuildStats()
    rc = hip.pag_shard_run(_C.c_void_p(g), _C.c_void_p(comm), _C.byref(inp), arr, _C.byref(st))
    if rc != 0:
        raise RuntimeError(f"pag_shard_run failed ({rc}): {hip.pag_last_error().decode()}")
    return st


def deal_contigs(lengths, world, ref_begin=None):
    """contigs -> ranks for the traversal.  Without positions: longest first (assign_blocks).  ref_begin[c] = where contig c
    maps to on the reference (single coordinate): CONTIGUOUS runs of contigs in reference order, balanced by length — the
    reference bands of a rank's contigs then merge into one stretch and the halo (regions_for) is paid twice per rank
    instead of twice per contig."""
    if ref_begin is None:
        return assign_blocks(list(lengths), world)
    order = sorted(range(len(lengths)), key=lambda c: (ref_begin[c], c))
    total = float(sum(lengths)) or 1.0
    out, acc, r = [[] for _ in range(world)], 0.0, 0
    for c in order:
        # (a contig goes to the rank in whose share of the total its middle falls)
        mid = acc + lengths[c] / 2.0
        r = min(world - 1, int(mid / total * world))
        out[r].append(c)
        acc += lengths[c]
    return out


def gather_paths(hip, g, mine, n_ctgs, dist, device):
    """Travel sequences of the contigs this rank walked (pag_travel with the others PAG_ORIENT_NONE) -> on every rank, the
    arrays pagh_assemble_paths takes: (paths[2 * n_ctgs] of c_void_p, lens[2 * n_ctgs], keep-alive buffers).
    mine: {slot = 2 * contig + (reverse ? 1 : 0)} this rank walked."""
    import numpy as np
    import torch
    world = dist.get_world_size() if dist else 1
    REC = 24  # sizeof(pag_path_node)
    hip.pag_travel_path_oriented.restype = _C.c_void_p
    hip.pag_travel_path_oriented.argtypes = [_C.c_void_p, _C.c_uint64, _C.c_int, _C.POINTER(_C.c_uint64)]
    meta = np.zeros((2 * n_ctgs,), dtype=np.int64)
    chunks = []
    for slot in sorted(mine):
        n = _C.c_uint64()
        p = hip.pag_travel_path_oriented(_C.c_void_p(g), slot // 2, 1 if slot % 2 == 0 else 0, _C.byref(n))
        meta[slot] = n.value
        if n.value:
            chunks.append(np.frombuffer(_C.string_at(p, n.value * REC), dtype=np.uint8))
    blob = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    if dist is None or world == 1:
        all_meta, all_blob = [meta], [blob]
    else:
        cdev = "cpu" if _is_gloo(dist) else device
        m = torch.from_numpy(meta).to(cdev)
        ms = [torch.empty_like(m) for _ in range(world)]
        dist.all_gather(ms, m)
        all_meta = [x.cpu().numpy() for x in ms]
        mx = max(int(x.sum()) * REC for x in all_meta)
        pad = torch.zeros(mx, dtype=torch.uint8, device=cdev)
        pad[:len(blob)] = torch.from_numpy(blob.copy()).to(cdev)
        bs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bs, pad)
        all_blob = [b.cpu().numpy() for b in bs]
    paths = (_C.c_void_p * (2 * n_ctgs))()
    lens = (_C.c_uint64 * (2 * n_ctgs))()
    keep = []
    for mt, bl in zip(all_meta, all_blob):
        at = 0
        bl = np.ascontiguousarray(bl)
        keep.append(bl)
        for slot in range(2 * n_ctgs):
            if mt[slot]:
                paths[slot] = bl.ctypes.data + at
                lens[slot] = int(mt[slot])
                at += int(mt[slot]) * REC
    return paths, lens, keep
