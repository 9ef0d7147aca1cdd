"""ONE config block too large for one GPU, built and traversed on ONE GPU: the N ranks of a sharded build (SURVEY.md §8e
level 2, `parallel.ShardedBuild` / include/pagraph_hip.h `pag_shard_*`) run ONE AFTER THE OTHER on the same device.

Nothing is parked in host memory (round 4 spilled every exchange — 265 GB at BASELINE configs[2] — and took the GPU box down
with it).  What a rank would RECEIVE over xGMI is RECOMPUTED when its turn comes; the schedule is destination-major:

    counts[r][o]                              one extraction per read range r (sizes of what r sends owner o)
    for rank d:                               (its contigs, its region of the graph)
        for owner o:
            for read range r: extract r, keep only o's records         -> o's received streams, canonical order
            K2-K4 on them (pag_shard_build)                             -> o's slice of the block's graph
            pag_shard_select for d's region                             -> appended to d's region, on the device
        import the region, release the build, pag_travel d's contigs    -> their travel sequences (host)
    chains + output files of the whole block from the gathered sequences (pagh_assemble_paths)

A build is 0.1-0.3 s per rank at configs[2]'s size, so the N^2 small builds cost seconds; the host holds the travel sequences
and nothing else.  Every kernel, every piece of host logic and every byte a rank takes in is the N-GPU run's; only the
transport differs (a device-to-device copy instead of RCCL) and the ranks take turns.  That makes it a MEASURING AID: BASELINE
configs[2] (1 M x 10 kb reads vs a 250 Mb reference, sharded over 4 GPUs) executes on the one GPU this build has, with the
per-rank device footprint of every stage measured (hipMemGetInfo around the stages) instead of computed by hand.
It is not how a multi-GPU node runs the block (that is `pag_shard_run` / `bin/pagraph` under PAGRAPH_SHARD), and no
multi-GPU timing follows from it.

Reference semantics: the block is one `PositionProcessor::process` + one `PAssembly::testTravel5` (pagraph.cpp:181-263,
PAssembly.cpp:30-79); the partition argument for bit-identity is the sharded build's (include/pagraph_hip.h, pag_shard_*).
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


def _host_bytes():
    try:
        return int(open("/sys/fs/cgroup/memory.current").read())
    except (OSError, ValueError):
        return 0


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
    """The block as N ranks, rank after rank, on one device (schedule: module docstring).

    make_handle() -> a fresh pag_graph* (e.g. pag_create_from_bitmap on the block's solid set); inp: the prepared
    pag_build_input (device resident, owned by the caller); ctgs: [(length)] per contig; ctg_alns / ref_lens: as
    parallel.regions_for takes them, with ref_begin taken from the first alignment of a contig; ctg_seqs / ref_seqs: host
    pag_seqs of the contigs / references (for pag_travel and the chain selection); orient[c]: PAG_ORIENT_*; out_dir: must not
    exist or be empty (the block's output files are written there).
    Returns a dict: count lines, per-rank per-stage device bytes, bytes a rank sends / takes in, held fractions, times, output digest."""
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
    if os.path.isdir(out_dir) and os.listdir(out_dir):
        raise ValueError(f"rank_serial: {out_dir} exists and is not empty (nothing of the caller's is ever removed)")
    os.makedirs(out_dir, exist_ok=True)
    N = n_ranks
    n_ctg = len(ctgs)
    first_aln = {}
    for (c, ri, tb, te) in ctg_alns:
        first_aln.setdefault(c, (ri, tb))
    rst = parallel.mapper_starts(ref_lens)
    ref_begin = [rst[first_aln[c][0]] + first_aln[c][1] if c in first_aln else 0 for c in range(n_ctg)]
    deal = parallel.deal_contigs(list(ctgs), N, ref_begin=ref_begin)
    regions = parallel.regions_for(deal, list(ctgs), orient, ctg_alns, list(ref_lens), halo=halo)
    res = {"n_ranks": N, "halo": halo, "schedule": "destination-major, everything a rank takes in recomputed on the device (no host spills)",
           "ranks": [dict() for _ in range(N)], "contigs_per_rank": [len(d) for d in deal]}
    t_all = time.perf_counter()
    host0 = _host_bytes()

    def sync():
        torch.cuda.synchronize(device)

    def used():
        return _used(torch, device)

    def destroy(g):
        hip.pag_destroy(C.c_void_p(g))
        torch.cuda.empty_cache()

    torch.cuda.empty_cache()
    base = used()  # (the block's inputs and whatever else the caller holds)
    res["device_bytes_inputs_and_caller"] = base
    peak = {"device": base}

    def note_peak():
        peak["device"] = max(peak["device"], used())

    def extract(g, r):
        c = (C.c_uint64 * (4 * N))()
        rc = hip.pag_shard_extract(C.c_void_p(g), C.byref(inp), r, N, c)
        if rc != 0:
            raise RuntimeError(f"read range {r}: pag_shard_extract failed ({rc}): {hip.pag_last_error().decode()}")
        return np.array(list(c), dtype=np.int64).reshape(N, 4)

    # ---- what every read range sends every owner: one extraction per range, each on a handle of its own, so that the device
    # bytes measured around it are the extraction stage of rank r and nothing else
    counts = np.zeros((N, N, 4), dtype=np.int64)
    for r in range(N):
        t0 = time.perf_counter()
        g = make_handle()
        counts[r] = extract(g, r)
        sync()
        info = res["ranks"][r]
        info["bytes_extract"] = used() - base
        note_peak()
        destroy(g)
        T, E = int(counts[r][:, 0:2].sum()), int(counts[r][:, 2:4].sum())
        info.update(tuples_extracted=T, edges_extracted=E, s_extract=time.perf_counter() - t0,
                    wire_out_tuples_bytes=int(12 * (T + E - counts[r][r].sum())))
        say(f"rank {r}: extracts {T} + {E} records, {info['bytes_extract'] / 1e9:.1f} GB on the device, {info['s_extract']:.1f} s")

    def gather_owner(gx, o):
        """owner o's received records ([pass 1 from range 0] .. [pass 1 from range N-1] [pass 2 from range 0] ..) as device tensors:
        every read range is extracted again on the handle gx and only o's part is kept"""
        out = []
        for which in (0, 1):
            q0 = 2 * which
            n1 = int(counts[:, o, q0].sum())
            n = n1 + int(counts[:, o, q0 + 1].sum())
            out.append((torch.empty(n, dtype=torch.int32, device=device), torch.empty(n, dtype=torch.int64, device=device), n1))
        (tk, tv, t1n), (ek, ev, e1n) = out
        at = [[0, t1n], [0, e1n]]  # next free place of [tuples, edges] x [pass 1, pass 2]
        for r in range(N):
            c = extract(gx, r)
            if not np.array_equal(c, counts[r]):
                raise RuntimeError(f"read range {r}: the extraction is not reproducible ({c.tolist()} then, {counts[r].tolist()} before)")
            for ps in (0, 1):
                t_off = int(counts[r, :o, 0:2].sum()) + (int(counts[r, o, 0]) if ps else 0)
                e_off = int(counts[r, :o, 2:4].sum()) + (int(counts[r, o, 2]) if ps else 0)
                t_n, e_n = int(counts[r, o, ps]), int(counts[r, o, 2 + ps])
                rc = hip.pag_shard_take_part(C.c_void_p(gx), t_off, t_n, C.c_void_p(tk.data_ptr() + 4 * at[0][ps]), C.c_void_p(tv.data_ptr() + 8 * at[0][ps]),
                                             e_off, e_n, C.c_void_p(ek.data_ptr() + 4 * at[1][ps]), C.c_void_p(ev.data_ptr() + 8 * at[1][ps]))
                if rc != 0:
                    raise RuntimeError(f"pag_shard_take_part failed ({rc}): {hip.pag_last_error().decode()}")
                at[0][ps] += t_n
                at[1][ps] += e_n
            note_peak()
        assert at[0] == [t1n, tk.numel()] and at[1] == [e1n, ek.numel()]
        return (tk, tv, t1n), (ek, ev, e1n)

    # ---- rank after rank: its region of the graph from all owners (recomputed), then its walks ---------------------------------
    prm = TravelParams(threads, 0, 2 * eps, 0.15, 0.90, min_len)
    ref_len_arr = np.array(list(ref_lens), dtype=np.uint32)
    paths = (C.c_void_p * (2 * n_ctg))()
    lens = (C.c_uint64 * (2 * n_ctg))()
    keep = []
    owner_counts = None
    n_pos_total = 0
    for d in range(N):
        info = res["ranks"][d]
        t0 = time.perf_counter()
        gx = make_handle()  # (all extractions of this rank's turn: its pools are sized by the first and reused)
        extract(gx, 0)      # (... and exist before the owner stages are measured)
        slices, stats, owner_stats = [], [], []
        t_extract = t_build = t_select = 0.0
        sel_bytes_in = 0
        for o in range(N):
            ta = time.perf_counter()
            m0 = used()
            (tk, tv, t1n), (ek, ev, e1n) = gather_owner(gx, o)
            sync()
            tb = time.perf_counter()
            go = make_handle()
            sb = parallel.ShardedBuild(hip, go, inp, o, N, device)
            st = sb.build((tk, tv), t1n, (ek, ev), e1n, eps)
            sync()
            if d == o:  # (this owner's own turn: what the stage holds on a rank of the N-GPU run)
                info["bytes_owner_build"] = used() - m0  # (received records + sort ping-pong + segment results)
                info["owner_tuples"], info["owner_edges"] = int(tk.numel()), int(ek.numel())
                info.update(owner_n_pos=int(st.n_pos), owner_n_nodes=int(st.n_nodes), owner_n_uniq_edges=int(st.n_uniq_edges))
            note_peak()
            del tk, tv, ek, ev
            torch.cuda.empty_cache()
            tc = time.perf_counter()
            arrs, sst = sb.select(regions[d])
            sync()
            note_peak()
            if d == o:
                info["bytes_owner_select_peak"] = used() - m0
            nbytes = sum(arrs[nm].numel() * arrs[nm].element_size() for nm in _NAMES)
            if o != d:
                sel_bytes_in += nbytes
                res["ranks"][o]["wire_out_selection_bytes"] = res["ranks"][o].get("wire_out_selection_bytes", 0) + int(nbytes)
            slices.append(arrs)
            stats.append(sst)
            owner_stats.append(st)
            destroy(go)
            t_extract += tb - ta
            t_build += tc - tb
            t_select += time.perf_counter() - tc
        destroy(gx)
        # the block's count lines = sums over the owners (the same sums in every rank's turn)
        tot_counts = [0] * 6
        for st in owner_stats:
            for i, x in enumerate(st.counts()):
                tot_counts[i] += int(x)
        if owner_counts is None:
            owner_counts = tot_counts
            n_pos_total = sum(int(st.n_pos) for st in owner_stats)
        elif owner_counts != tot_counts:
            raise RuntimeError(f"rank {d}: the owners' count lines {tot_counts} differ from those of rank 0's turn {owner_counts}")
        t1 = time.perf_counter()
        g = make_handle()
        sb = parallel.ShardedBuild(hip, g, inp, d, N, device)
        tot = sb.import_all(slices, stats)
        sb.set_region(regions[d])
        del slices
        torch.cuda.empty_cache()
        hip.pag_shard_release_build(C.c_void_p(g))
        sync()
        info["bytes_region_imported"] = used() - base
        info["wire_in_region_bytes"] = int(sel_bytes_in)
        if list(tot.counts()) != owner_counts:
            raise RuntimeError(f"rank {d}: count lines {list(tot.counts())} differ from the owners' sums {owner_counts}")
        nn, npos, ne = C.c_uint64(), C.c_uint64(), C.c_uint64()
        hip.pag_csr_sizes(C.c_void_p(g), C.byref(nn), C.byref(npos), C.byref(ne))
        info["held_vertices"], info["held_edges"] = int(npos.value), int(ne.value)
        info["held_fraction"] = npos.value / max(1, n_pos_total)
        t2 = time.perf_counter()
        mine = np.full(n_ctg, -1, dtype=np.int32)
        for cidx in deal[d]:
            mine[cidx] = orient[cidx]
        rc = hip.pag_travel(C.c_void_p(g), C.byref(ctg_seqs), mine.ctypes.data, ref_len_arr.ctypes.data, len(ref_len_arr), C.byref(prm), None)
        if rc != 0:
            raise RuntimeError(f"rank {d}: pag_travel failed ({rc}): {hip.pag_last_error().decode()}")
        sync()
        info["bytes_traversal_peak"] = used() - base  # (region + traversal graph + successor records + walk arena)
        note_peak()
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
        destroy(g)
        info.update(s_recompute_extract=t_extract, s_recompute_build=t_build, s_recompute_select=t_select, s_import=t2 - t1,
                    s_travel=time.perf_counter() - t2, s_turn=time.perf_counter() - t0, path_nodes=int(n_nodes_path), contigs=len(deal[d]))
        say(f"rank {d}: its region from {N} owners recomputed in {t1 - t0:.1f} s (extract {t_extract:.1f}, build {t_build:.1f}, select {t_select:.1f}); holds "
            f"{info['held_fraction']:.3f} of the vertices ({info['bytes_region_imported'] / 1e9:.1f} GB imported), traversal peak "
            f"{info['bytes_traversal_peak'] / 1e9:.1f} GB, {len(deal[d])} contigs walked in {info['s_travel']:.1f} s")
    res["count_lines_sum_over_owners"] = owner_counts
    res["vertices_total"] = n_pos_total
    res["device_bytes_peak_of_the_serial_run"] = peak["device"]  # (this schedule's own peak: several stages' buffers coexist)
    res["host_bytes_growth"] = max(0, _host_bytes() - host0)     # (the travel sequences)

    # ---- rank 0's part: the chains of the whole block from the gathered travel sequences ---------------------------------
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
    res.update(path_nodes=int(ts.n_path_nodes), path_bases=int(ts.n_path_bases), path_checksum=f"{ts.path_checksum:016x}",
               chains=int(ts.n_chains_emitted), s_total=time.perf_counter() - t_all)
    return res
