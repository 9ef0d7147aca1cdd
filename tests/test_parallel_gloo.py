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
