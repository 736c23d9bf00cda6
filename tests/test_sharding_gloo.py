"""world_size-2 CPU test (gloo) of the multi-GPU path: the sweep is embarrassingly parallel, so the only
cross-rank logic is (a) the shard assignment and (b) bench.py's barrier + max-over-ranks timing and the
sum of per-rank work.  No data-path collective exists."""
import os
import socket
import sys
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path[:0] = [str(REPO), str(REPO / "cuda-l2_amd")]
    import bench
    from tools import sweep

    dist.init_process_group("gloo", rank=rank, world_size=world)
    shapes = [f"{64 * (i + 1)}_128_64" for i in range(7)]
    mine = sweep.shard(shapes, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    # bench.py's reduction: max elapsed over ranks, sum of flops over ranks
    elapsed, flops = bench.reduce_over_ranks(0.010 * (rank + 1), float(sum(map(sweep.flops, mine))), "cpu")
    if rank == 0:
        out.put((gathered, elapsed, flops, float(sum(map(sweep.flops, shapes)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_reduction():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, elapsed, flops, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(gathered[0]) == 4 and len(gathered[1]) == 3
    assert sorted(gathered[0] + gathered[1]) == sorted(f"{64 * (i + 1)}_128_64" for i in range(7))
    assert abs(elapsed - 0.020) < 1e-9      # max over ranks
    assert flops == total                    # whole-job work = sum over ranks
