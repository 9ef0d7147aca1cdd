"""Multi-GPU layer of the PAGraph hot path: one process per GPU, torch.distributed (RCCL on GPUs, gloo
in CPU tests).

The path shards by reference sequence (SURVEY.md §8e level 1): AlignGraph2 already runs one pagraph per
per-reference sub-directory, sequentially (reference AlignGraph2.py:399-431); config blocks are
independent after resetAllNodes (pagraph.cpp:181-182).  So the units are distributed over the ranks with
NO data-path collective; the only communication is the barrier / max / sum around the timed region and
the gathering of exit codes.
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
