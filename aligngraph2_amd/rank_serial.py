"""ONE config block too large for one GPU, built and traversed on ONE GPU: the N ranks of a sharded build (SURVEY.md §8e
level 2, `parallel.ShardedBuild` / include/pagraph_hip.h `pag_shard_*`) run ONE AFTER THE OTHER on the same device, and what
the ranks of a real N-GPU run would hand each other over xGMI waits in pinned host memory in between.

Every kernel, every piece of host logic and every byte exchanged is the N-GPU run's; only the transport differs (host memory
instead of RCCL) and the ranks take turns.  That makes it two things:

* a way to run BASELINE configs[2] (1 M x 10 kb reads vs a 250 Mb reference, sharded over 4 GPUs) on the one GPU this build
  has, with the per-rank device footprint of every stage MEASURED (hipMemGetInfo around the stages) instead of computed by
  hand;
* a product mode: a block whose graph does not fit one MI355X is still processed on it, at the price of the host spills.

Reference semantics: the block is one `PositionProcessor::process` + one `PAssembly::testTravel5` (pagraph.cpp:181-263);
the partition argument for bit-identity is the sharded build's (include/pagraph_hip.h, pag_shard_*).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import time

import numpy as np

from . import parallel

_NAMES = ("tkey", "tval", "tseg", "tcnt", "ekey", "eval", "eseg")


class TravelParams(C.Structure):
    """pag_travel_params (include/pagraph_hip.h)"""
    _fields_ = [("ref_threads", C.c_uint32), ("reserved", C.c_uint32), ("deviation", C.c_uint64), ("error_rate", C.c_double),
                ("start_split", C.c_double), ("min_len", C.c_uint64)]


def _used(torch, device):
    """bytes in use on the device as the driver sees them (library pools + torch's allocator)"""
    free, total = torch.cuda.mem_get_info(device)
    return int(total - free)


_PINNED = os.environ.get("PAG_RANK_SERIAL_PINNED") == "1"


def _to_host(torch, t):
    """a device tensor into host memory.  Pageable by default: torch's pinned allocator rounds every block up to a power of two
    and keeps it cached, which at BASELINE configs[2]'s 265 GB of spills is the difference between fitting the GPU box's
    300 GiB container and being killed by it (PAG_RANK_SERIAL_PINNED=1: pinned, faster copies)."""
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=_PINNED)
    h.copy_(t, non_blocking=_PINNED)
    return h


def _host_guard(limit_frac=0.8):
    """raises before the container's memory limit is reached (a run that is killed by it takes the GPU box with it)"""
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read())
    except (OSError, ValueError):
        return 0
    if lim != "max" and cur > limit_frac * int(lim):
        raise MemoryError(f"rank_serial: {cur / 1e9:.0f} GB of host memory in use, the container's limit is {int(lim) / 1e9:.0f} GB: the block's "
                          "spills do not fit this host")
    return cur


def _drop_host_cache(torch):
    fn = getattr(torch._C, "_host_emptyCache", None)
    if fn is not None:
        fn()


def digest_dir(path):
    """one SHA-256 over the names, sizes and contents of all files of a directory (sorted by name)"""
    h = hashlib.sha256()
    total = 0
    for name in sorted(os.listdir(path)):
        p = os.path.join(path, name)
        h.update(name.encode() + b"\0")
        with open(p, "rb") as f:
            while True:
                b = f.read(1 << 24)
                if not b:
                    break
                h.update(b)
                total += len(b)
    return h.hexdigest(), total


def run(hip, host, make_handle, inp, *, n_ranks, eps, k, threads, ctgs, ctg_alns, ref_lens, ctg_seqs, ref_seqs, orient, out_dir, device="cuda",
        halo=200_000, min_len=50, log=None):
    """The block as N ranks, rank after rank, on one device.

    make_handle() -> a fresh pag_graph* (e.g. pag_create_from_bitmap on the block's solid set); inp: the prepared
    pag_build_input (device resident, owned by the caller); ctgs: [(length)] per contig; ctg_alns / ref_lens: as
    parallel.regions_for takes them, with ref_begin taken from the first alignment of a contig; ctg_seqs / ref_seqs: host
    pag_seqs of the contigs / references (for pag_travel and the chain selection); orient[c]: PAG_ORIENT_*.
    Returns a dict: count lines, per-rank per-stage device bytes, wire bytes, held fractions, times, output digest."""
    import torch
    say = log or (lambda *a: None)
    parallel.bind_shard_api(hip)
    hip.pag_destroy.argtypes = [C.c_void_p]
    hip.pag_travel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    hip.pag_travel.restype = C.c_int
    hip.pag_travel_path_oriented.restype = C.c_void_p
    hip.pag_travel_path_oriented.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64)]
    hip.pag_csr_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 3
    host.pagh_assemble_paths.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_uint32, C.c_uint64, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_void_p]
    host.pagh_assemble_paths.restype = C.c_int
    host.pagh_last_error.restype = C.c_char_p
    N = n_ranks
    n_ctg = len(ctgs)
    first_aln = {}
    for (c, ri, tb, te) in ctg_alns:
        first_aln.setdefault(c, (ri, tb))
    rst = parallel.mapper_starts(ref_lens)
    ref_begin = [rst[first_aln[c][0]] + first_aln[c][1] if c in first_aln else 0 for c in range(n_ctg)]
    deal = parallel.deal_contigs(list(ctgs), N, ref_begin=ref_begin)
    regions = parallel.regions_for(deal, list(ctgs), orient, ctg_alns, list(ref_lens), halo=halo)
    res = {"n_ranks": N, "halo": halo, "ranks": [dict() for _ in range(N)], "contigs_per_rank": [len(d) for d in deal]}
    t_all = time.perf_counter()

    def sync():
        torch.cuda.synchronize(device)

    torch.cuda.empty_cache()
    base = _used(torch, device)  # (the block's inputs and whatever else the caller holds)
    res["device_bytes_inputs_and_caller"] = base

    # ---- phase A: every rank extracts its read range and partitions its two streams by k-mer owner ---------------------
    counts = np.zeros((N, N, 4), dtype=np.int64)
    spilled = []  # per rank: (tk, tv, ek, ev) in pinned host memory
    for r in range(N):
        t0 = time.perf_counter()
        g = make_handle()
        c = (C.c_uint64 * (4 * N))()
        rc = hip.pag_shard_extract(C.c_void_p(g), C.byref(inp), r, N, c)
        if rc != 0:
            raise RuntimeError(f"rank {r}: pag_shard_extract failed ({rc}): {hip.pag_last_error().decode()}")
        sync()
        counts[r] = np.array(list(c), dtype=np.int64).reshape(N, 4)
        res["ranks"][r]["bytes_extract"] = _used(torch, device) - base
        T, E = int(counts[r][:, 0:2].sum()), int(counts[r][:, 2:4].sum())
        tk = torch.empty(T, dtype=torch.int32, device=device)
        tv = torch.empty(T, dtype=torch.int64, device=device)
        ek = torch.empty(E, dtype=torch.int32, device=device)
        ev = torch.empty(E, dtype=torch.int64, device=device)
        rc = hip.pag_shard_take(C.c_void_p(g), C.c_void_p(tk.data_ptr()), C.c_void_p(tv.data_ptr()), C.c_void_p(ek.data_ptr()), C.c_void_p(ev.data_ptr()))
        if rc != 0:
            raise RuntimeError(f"rank {r}: pag_shard_take failed ({rc}): {hip.pag_last_error().decode()}")
        hip.pag_destroy(C.c_void_p(g))
        t1 = time.perf_counter()
        # what goes to owner o waits as its own host arrays (freed when o has taken them): [o] -> (tkey, tval, ekey, eval)
        per_dst = []
        for o in range(N):
            t_off, t_n = int(counts[r, :o, 0:2].sum()), int(counts[r, o, 0:2].sum())
            e_off, e_n = int(counts[r, :o, 2:4].sum()), int(counts[r, o, 2:4].sum())
            per_dst.append((_to_host(torch, tk[t_off:t_off + t_n]), _to_host(torch, tv[t_off:t_off + t_n]),
                            _to_host(torch, ek[e_off:e_off + e_n]), _to_host(torch, ev[e_off:e_off + e_n])))
        spilled.append(per_dst)
        res["host_bytes_peak"] = max(res.get("host_bytes_peak", 0), _host_guard())
        sync()
        del tk, tv, ek, ev
        torch.cuda.empty_cache()
        res["ranks"][r].update(tuples_extracted=T, edges_extracted=E, s_extract=t1 - t0, s_spill_streams=time.perf_counter() - t1,
                               wire_out_tuples_bytes=int(12 * (T + E - counts[r][r].sum())))
        say(f"rank {r}: extracted {T} + {E} records, {res['ranks'][r]['bytes_extract'] / 1e9:.1f} GB on the device, {t1 - t0:.1f} s + spill "
            f"{time.perf_counter() - t1:.1f} s")

    # ---- phase B: every owner sorts / clusters what it received and selects every rank's region of its slice ------------
    def received(o, which):
        """owner o's records of one stream ([pass 1 from rank 0] .. [pass 1 from rank N-1] [pass 2 from rank 0] ..), uploaded"""
        q0 = 2 * which
        n1 = int(counts[:, o, q0].sum())
        n = n1 + int(counts[:, o, q0 + 1].sum())
        key = torch.empty(n, dtype=torch.int32, device=device)
        val = torch.empty(n, dtype=torch.int64, device=device)
        d1, d2 = 0, n1
        for r in range(N):
            a, b = int(counts[r, o, q0]), int(counts[r, o, q0 + 1])
            hk, hv = spilled[r][o][2 * which], spilled[r][o][2 * which + 1]
            key[d1:d1 + a].copy_(hk[:a], non_blocking=True)
            val[d1:d1 + a].copy_(hv[:a], non_blocking=True)
            key[d2:d2 + b].copy_(hk[a:a + b], non_blocking=True)
            val[d2:d2 + b].copy_(hv[a:a + b], non_blocking=True)
            d1 += a
            d2 += b
        return key, val, n1

    selected = [[None] * N for _ in range(N)]  # [owner][dest] -> (dict of pinned arrays, stats)
    owner_stats = []
    for o in range(N):
        t0 = time.perf_counter()
        g = make_handle()
        sb = parallel.ShardedBuild(hip, g, inp, o, N, device)
        tk, tv, t1n = received(o, 0)
        ek, ev, e1n = received(o, 1)
        sync()
        for r in range(N):
            spilled[r][o] = None  # (taken)
        t1 = time.perf_counter()
        st = sb.build((tk, tv), t1n, (ek, ev), e1n, eps)
        sync()
        info = res["ranks"][o]
        info["bytes_owner_build"] = _used(torch, device) - base  # (received records + sort ping-pong + segment results)
        info["owner_tuples"], info["owner_edges"] = int(tk.numel()), int(ek.numel())
        del tk, tv, ek, ev
        torch.cuda.empty_cache()
        owner_stats.append(st)
        t2 = time.perf_counter()
        peak_sel = 0
        sel_bytes = 0
        for d in range(N):
            arrs, sst = sb.select(regions[d])
            sync()
            peak_sel = max(peak_sel, _used(torch, device) - base)
            selected[o][d] = ({nm: _to_host(torch, arrs[nm]) for nm in _NAMES}, sst)
            res["host_bytes_peak"] = max(res.get("host_bytes_peak", 0), _host_guard())
            sync()
            sel_bytes += sum(arrs[nm].numel() * arrs[nm].element_size() for nm in _NAMES) if d != o else 0
            del arrs
            torch.cuda.empty_cache()
        info["bytes_owner_select_peak"] = peak_sel
        info["wire_out_selection_bytes"] = int(sel_bytes)
        info.update(s_upload_received=t1 - t0, s_owner_build=t2 - t1, s_select_and_spill=time.perf_counter() - t2,
                    owner_n_pos=int(st.n_pos), owner_n_nodes=int(st.n_nodes), owner_n_uniq_edges=int(st.n_uniq_edges))
        hip.pag_destroy(C.c_void_p(g))
        torch.cuda.empty_cache()
        say(f"owner {o}: {info['owner_tuples']} + {info['owner_edges']} records built in {t2 - t1:.1f} s, {info['bytes_owner_build'] / 1e9:.1f} GB; "
            f"{st.n_pos} vertices; selections {time.perf_counter() - t2:.1f} s")
    del spilled
    _drop_host_cache(torch)
    # the block's count lines = sums over the owners
    tot_counts = [0] * 6
    n_pos_total = 0
    for st in owner_stats:
        for i, x in enumerate(st.counts()):
            tot_counts[i] += int(x)
        n_pos_total += int(st.n_pos)
    res["count_lines_sum_over_owners"] = tot_counts
    res["vertices_total"] = n_pos_total

    # ---- phase C: every rank imports its region from all owners and walks the contigs it was dealt ----------------------
    prm = TravelParams(threads, 0, 2 * eps, 0.15, 0.90, min_len)
    ref_len_arr = np.array(list(ref_lens), dtype=np.uint32)
    paths = (C.c_void_p * (2 * n_ctg))()
    lens = (C.c_uint64 * (2 * n_ctg))()
    keep = []
    for d in range(N):
        t0 = time.perf_counter()
        g = make_handle()
        sb = parallel.ShardedBuild(hip, g, inp, d, N, device)
        slices, stats = [], []
        for o in range(N):
            harr, sst = selected[o][d]
            slices.append({nm: harr[nm].to(device, non_blocking=True) for nm in _NAMES})
            stats.append(sst)
        sync()
        tot = sb.import_all(slices, stats)
        sb.set_region(regions[d])
        del slices
        torch.cuda.empty_cache()
        hip.pag_shard_release_build(C.c_void_p(g))
        sync()
        info = res["ranks"][d]
        info["bytes_region_imported"] = _used(torch, device) - base
        info["wire_in_region_bytes"] = int(sum(sum(selected[o][d][0][nm].numel() * selected[o][d][0][nm].element_size() for nm in _NAMES)
                                               for o in range(N) if o != d))
        for o in range(N):
            selected[o][d] = None  # (imported)
        if list(tot.counts()) != tot_counts:
            raise RuntimeError(f"rank {d}: count lines {list(tot.counts())} differ from the owners' sums {tot_counts}")
        nn, npos, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hip.pag_csr_sizes(C.c_void_p(g), C.byref(nn), C.byref(npos), C.byref(ne))
        info["held_vertices"], info["held_edges"] = int(npos.value), int(ne.value)
        info["held_fraction"] = npos.value / max(1, n_pos_total)
        t1 = time.perf_counter()
        mine = np.full(n_ctg, -1, dtype=np.int32)
        for cidx in deal[d]:
            mine[cidx] = orient[cidx]
        rc = hip.pag_travel(C.c_void_p(g), C.byref(ctg_seqs), mine.ctypes.data, ref_len_arr.ctypes.data, len(ref_len_arr), C.byref(prm), None)
        if rc != 0:
            raise RuntimeError(f"rank {d}: pag_travel failed ({rc}): {hip.pag_last_error().decode()}")
        sync()
        info["bytes_traversal_peak"] = _used(torch, device) - base  # (region + traversal graph + successor records + walk arena)
        n_nodes_path = 0
        for cidx in deal[d]:
            for fwd in ((1, 0) if mine[cidx] == 2 else ((1,) if mine[cidx] == 1 else (0,))):
                n = C.c_uint64()
                p = hip.pag_travel_path_oriented(C.c_void_p(g), cidx, fwd, C.byref(n))
                buf = C.create_string_buffer(C.string_at(p, n.value * 24), n.value * 24) if n.value else None  # (the "gather")
                keep.append(buf)
                slot = 2 * cidx + (0 if fwd else 1)
                paths[slot] = C.cast(buf, C.c_void_p).value if buf is not None else None
                lens[slot] = n.value
                n_nodes_path += n.value
        host.pagh_release(C.c_void_p(g))
        hip.pag_destroy(C.c_void_p(g))
        torch.cuda.empty_cache()
        info.update(s_import=t1 - t0, s_travel=time.perf_counter() - t1, path_nodes=int(n_nodes_path), contigs=len(deal[d]))
        say(f"rank {d}: holds {info['held_fraction']:.3f} of the vertices ({info['bytes_region_imported'] / 1e9:.1f} GB imported), traversal peak "
            f"{info['bytes_traversal_peak'] / 1e9:.1f} GB, {len(deal[d])} contigs walked in {info['s_travel']:.1f} s")
    del selected
    _drop_host_cache(torch)

    # ---- rank 0's part: the chains of the whole block from the gathered travel sequences ---------------------------------
    import shutil
    if os.path.isdir(out_dir):
        shutil.rmtree(out_dir)
    os.makedirs(out_dir)
    t0 = time.perf_counter()

    class TraverseStats(C.Structure):
        _fields_ = [("n_contigs", C.c_uint64), ("n_path_nodes", C.c_uint64), ("n_path_bases", C.c_uint64), ("n_chains_emitted", C.c_uint64),
                    ("n_fasta_bases", C.c_uint64), ("path_checksum", C.c_uint64), ("ms_export", C.c_double), ("ms_traverse", C.c_double),
                    ("ms_total", C.c_double), ("ms_successors", C.c_double), ("ms_walk", C.c_double), ("walk_rounds", C.c_uint64),
                    ("walk_jobs", C.c_uint64), ("walk_steps", C.c_uint64), ("walk_classifications", C.c_uint64)]
    ts = TraverseStats()
    orient_arr = np.array(list(orient), dtype=np.int32)
    rc = host.pagh_assemble_paths(None, k, C.byref(ctg_seqs), None, C.byref(ref_seqs), None, orient_arr.ctypes.data, paths, lens, threads, eps, min_len,
                                  out_dir.encode(), b"0_", 0, C.byref(ts))
    if rc != 0:
        raise RuntimeError(f"pagh_assemble_paths failed ({rc}): {host.pagh_last_error().decode()}")
    res["s_assemble"] = time.perf_counter() - t0
    res["outputs_sha256"], res["outputs_bytes"] = digest_dir(out_dir)
    res["host_bytes_peak"] = max(res.get("host_bytes_peak", 0), _host_guard())
    res.update(path_nodes=int(ts.n_path_nodes), path_bases=int(ts.n_path_bases), path_checksum=f"{ts.path_checksum:016x}",
               chains=int(ts.n_chains_emitted), s_total=time.perf_counter() - t_all)
    return res
