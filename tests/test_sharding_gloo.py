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
    assert {len(gathered[0]), len(gathered[1])} == {3, 4} and not set(gathered[0]) & set(gathered[1])
    assert sorted(gathered[0] + gathered[1]) == sorted(f"{64 * (i + 1)}_128_64" for i in range(7))
    assert abs(elapsed - 0.020) < 1e-9      # max over ranks
    assert flops == total                    # whole-job work = sum over ranks


def test_two_rank_cpu_sweep_run_then_merge(tmp_path):
    """tools/sweep.py end to end on two ranks without a GPU (BASELINE config 1, the harness's --device cpu plumbing path):
    launched the way the driver launches a multi-GPU job (torch.distributed.run, 127.0.0.1 rendezvous), every rank
    evaluates its shard through benchmarking_offline.py, the results meet on the filesystem and `merge` reads BOTH
    rank status files: ranks == 2, max-over-ranks wall, sum 2MNK / that wall and the CPU column are all there."""
    import json
    import subprocess

    shapes = ["64_4096_64", "64_128_64", "128_128_64", "64_64_128", "128_64_64", "64_64_64"]
    out = tmp_path / "sweep"
    env = dict(os.environ, OMP_NUM_THREADS="2")
    common = ["--out", str(out), "--acc_precise", "fp32", "--mode", "offline", "--shapes", ",".join(shapes)]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(_free_port()), str(REPO / "cuda-l2_amd" / "tools" / "sweep.py"), "run", "--device", "cpu",
                          "--warmup_seconds", "0.02", "--benchmark_seconds", "0.05", *common],
                         capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    status = sorted((out / "fp32_offline").glob("rank*_status.json"))
    assert [p.name for p in status] == ["rank0_status.json", "rank1_status.json"]
    st = [json.loads(p.read_text()) for p in status]
    assert [s["done"] for s in st] == [3, 3] and all(s["failed"] == 0 and s["world"] == 2 for s in st)
    merged = subprocess.run([sys.executable, str(REPO / "cuda-l2_amd" / "tools" / "sweep.py"), "merge", *common],
                            capture_output=True, text=True, env=env, timeout=300)
    assert merged.returncode == 0, merged.stdout + merged.stderr
    rep = json.loads(merged.stdout)
    agg = rep["aggregate"]
    total = sum(2.0 * m * n * k for m, n, k in (map(int, s.split("_")) for s in shapes))
    assert rep["cpu_plumbing_shapes"] == 6 and agg["ranks"] == 2
    assert agg["max_rank_wall_s"] == max(s["seconds"] for s in st) > 0
    assert agg["total_flops_one_pass"] == total
    assert abs(agg["sweep_tflops_sum2mnk_over_max_rank_wall"] - total / agg["max_rank_wall_s"] * 1e-12) < 1e-12
    cpu = agg["cpu_torch_matmul"]
    assert cpu["shapes"] == 6 and cpu["flop_weighted_tflops"] > 0 and cpu["os_cpu_count"] >= 1
    # a second sweep (other accumulate tree) keeps its own status files: nothing is overwritten
    assert not (out / "rank0_status.json").exists()


def test_cost_sorted_sharding_balances_the_grid():
    """SURVEY.md section 8e / VERDICT r4 item 9: the shard assignment is cost-sorted (longest-processing-time-first on the per-shape
    wall times the round-4 sweep recorded), a partition, deterministic, and leaves the ranks within 5 % of each other on the
    1000-shape grid for 2, 4 and 8 GPUs -- the slowest rank's wall is the denominator of the multi-GPU aggregate."""
    sys.path[:0] = [str(REPO / "cuda-l2_amd")]
    from tools import sweep
    from tools.gen_shape_kernels import grid_shapes

    shapes = grid_shapes()
    assert len(shapes) == 1000 and all(s in sweep.recorded_costs() for s in shapes)
    for world in (1, 2, 4, 8):
        parts = [sweep.shard(shapes, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == sorted(shapes) and sum(map(len, parts)) == len(shapes)      # a partition
        assert parts == [sweep.shard(shapes, r, world) for r in range(world)]                        # deterministic
        for part in parts:                                                                           # grid order kept
            idx = [shapes.index(s) for s in part[:50]]
            assert idx == sorted(idx)
        loads = sweep.shard_loads(shapes, world)
        assert max(loads) <= 1.05 * (sum(loads) / world), (world, loads)
        assert max(loads) - min(loads) <= max(map(sweep.estimated_cost, shapes))                     # LPT's bound
    # shapes off the grid fall back to the analytic estimate and still balance (costs differ by 100x here)
    odd = [f"{64 * i}_{4096 if i % 3 else 128}_{8192 if i % 2 else 64}" for i in range(1, 41)]
    loads = sweep.shard_loads(odd, 4)
    assert max(loads) <= 1.10 * (sum(loads) / 4)
    try:
        sweep.shard(shapes, 2, 2)
        raise AssertionError("rank outside world must raise")
    except ValueError:
        pass
