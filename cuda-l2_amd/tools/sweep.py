"""Shape-sweep driver: shards the (M,N,K) list across GPUs, evaluates every shape with the harness
(eval_one_file.sh: correctness + 7 baselines), is resumable, and merges the per-shape summaries into
the reference's eval_results CSV format plus per-shape TFLOP/s and the geomean speedups.

The reference publishes only the merged CSVs (eval_results/*.csv) and has no driver; a single GEMM
never spans GPUs, so multi-GPU = independent shards, no collective:

    rank i of G evaluates its cost-balanced share of the shapes (shard(): longest-processing-time-first on the recorded
    per-shape costs) on GPU i (one process per GPU), results meet on the filesystem.

  python tools/sweep.py run   --out results --acc_precise fp32 --mode offline [--gpus 8] [--shapes-file f]
  python tools/sweep.py run   --inprocess ...   same metric without the per-baseline process churn (tools/sweep_inprocess.py)
  python tools/sweep.py run   --device cpu ...  plumbing run without a GPU (BASELINE config 1): every shape of the shard goes
                                                through benchmarking_offline.py --device cpu --perf_func matmul
  python tools/sweep.py merge --out results --acc_precise fp32 --mode offline
Launched under torch.distributed.run it takes rank/world size from RANK / WORLD_SIZE / LOCAL_RANK.

merge also reports the sweep's aggregate throughput: sum of 2MNK over the evaluated shapes
/ the slowest rank's wall time, the FLOP-weighted TFLOP/s of the cuda_l2 calls themselves, and -- where the
run measured it -- torch.matmul on the host CPU cores (core count stated) next to them.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
if str(PKG_DIR) not in sys.path:
    sys.path.insert(0, str(PKG_DIR))

from tools.gen_shape_kernels import ACC_DIRS, grid_shapes  # noqa: E402

CSV_COLUMNS = ["torch.matmul", "rocBLAS-tn", "rocBLAS-nn", "rocBLAS-max", "hipBLASLt-heuristic-tn",
               "hipBLASLt-heuristic-nn", "hipBLASLt-heuristic-max", "hipBLASLt-auto-tuning-tn",
               "hipBLASLt-auto-tuning-nn", "hipBLASLt-auto-tuning-max"]


MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK_TBPS = 8.0             # HBM3E spec


_COSTS = None


def recorded_costs() -> dict:
    """Seconds per shape as the round-6 whole-grid sweep recorded them (tools/sweep_costs_r06.json; rounds 4-5: sweep_costs_r04.json: time-boxed loop + autotune
    search of the in-process driver)."""
    global _COSTS
    if _COSTS is None:
        path = PKG_DIR / "tools" / "sweep_costs_r06.json"
        _COSTS = json.loads(path.read_text())["costs"] if path.exists() else {}
    return _COSTS


def estimated_cost(mnk: str) -> float:
    """Wall seconds one shape costs a rank: the recorded figure for a grid shape, else fixed time boxes + what grows with the
    shape (operand generation and the first calls scale with bytes / flops)."""
    c = recorded_costs().get(mnk)
    if c is not None:
        return float(c)
    m, n, k = map(int, mnk.split("_"))
    return 0.26 + 2.0 * m * n * k / 2.5e13 + 2.0 * (m * k + k * n + m * n) / 3e9


def shard(shapes: list[str], rank: int, world: int) -> list[str]:
    """Cost-sorted partition ("sort shapes by estimated eval cost"): longest-processing-time-first -- shapes
    in decreasing estimated cost, each to the rank with the least load so far (ties: the lower rank) -- so the slowest rank's
    wall, the denominator of the multi-GPU aggregate, stays within a shape's cost of the mean.  Deterministic: every rank computes
    the same assignment; the union over ranks is exactly `shapes`; a rank's shapes come back in the order of `shapes` (resumable
    runs and records stay in grid order).  Round-robin, what rounds 1-4 did, left the ranks 0.1-1.5 % apart on the recorded costs;
    this leaves < 0.1 %."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    order = sorted(range(len(shapes)), key=lambda i: (-estimated_cost(shapes[i]), i))
    load = [0.0] * world
    owner = [0] * len(shapes)
    for i in order:
        r = min(range(world), key=lambda x: (load[x], x))
        owner[i] = r
        load[r] += estimated_cost(shapes[i])
    return [s for i, s in enumerate(shapes) if owner[i] == rank]


def shard_loads(shapes: list[str], world: int) -> list[float]:
    return [sum(map(estimated_cost, shard(shapes, r, world))) for r in range(world)]


def flops(mnk: str) -> float:
    m, n, k = map(int, mnk.split("_"))
    return 2.0 * m * n * k


def shape_dir(out: Path, acc: str, mode: str, mnk: str) -> Path:
    return out / f"{acc}_{mode}" / mnk


def is_done(out: Path, acc: str, mode: str, mnk: str) -> bool:
    return (shape_dir(out, acc, mode, mnk) / "summary.json").exists()


def status_path(out: Path, acc: str, mode: str, rank: int) -> Path:
    """One status file per (accumulate tree, mode, rank), next to that sweep's records: `merge` reads them from there."""
    return out / f"{acc}_{mode}" / f"rank{rank}_status.json"


def run_shapes_cpu(shapes, args, rank: int) -> dict:
    """BASELINE config 1 for a whole shard: the harness's own CPU plumbing path (benchmarking_offline.py --device cpu
    --perf_func matmul, one process per shape as the reference's flow), no extension, no GPU."""
    done = skipped = failed = 0
    t0 = time.time()
    for mnk in shapes:
        d = shape_dir(args.out, args.acc_precise, args.mode, mnk)
        if (d / "benchmark_result_matmul.json").exists():
            skipped += 1
            continue
        cmd = [sys.executable, str(PKG_DIR / "benchmarking_offline.py"), "--mnk", mnk, "--acc_precise", args.acc_precise,
               "--device_type", "mi355x", "--base_dir", str(d), "--gpu_device_id", "0", "--device", "cpu", "--perf_func", "matmul",
               "--warmup_seconds", str(args.warmup_seconds), "--benchmark_seconds", str(args.benchmark_seconds)]
        res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(PKG_DIR))
        d.mkdir(parents=True, exist_ok=True)
        (d / "eval.log").write_text(res.stdout + res.stderr)
        if res.returncode == 0:
            done += 1
        else:
            failed += 1
            print(f"[rank {rank}] {mnk} FAILED (see {d / 'eval.log'})", flush=True)
    return {"rank": rank, "gpu": None, "device": "cpu", "done": done, "skipped": skipped, "failed": failed, "seconds": time.time() - t0}


def run_shapes(shapes, args, rank: int, gpu: int) -> dict:
    done = skipped = failed = 0
    t0 = time.time()
    for mnk in shapes:
        if is_done(args.out, args.acc_precise, args.mode, mnk):
            skipped += 1
            continue
        cmd = [str(PKG_DIR / "eval_one_file.sh"), "--mnk", mnk, "--acc_precise", args.acc_precise, "--device_type",
               "mi355x", "--warmup_seconds", str(args.warmup_seconds), "--benchmark_seconds",
               str(args.benchmark_seconds), "--base_dir", str(shape_dir(args.out, args.acc_precise, args.mode, mnk)),
               "--gpu_device_id", str(gpu), "--mode", args.mode]
        if args.mode == "server":
            cmd += ["--target_qps", str(args.target_qps)]
        res = subprocess.run(cmd, capture_output=True, text=True)
        log = shape_dir(args.out, args.acc_precise, args.mode, mnk) / "eval.log"
        log.parent.mkdir(parents=True, exist_ok=True)
        log.write_text(res.stdout + res.stderr)
        if res.returncode == 0:
            done += 1
        else:
            failed += 1
            print(f"[rank {rank}] {mnk} FAILED (see {log})", flush=True)
    return {"rank": rank, "gpu": gpu, "done": done, "skipped": skipped, "failed": failed, "seconds": time.time() - t0}


def geomean(values) -> float:
    vals = [v for v in values if v and v > 0 and math.isfinite(v)]
    return math.exp(sum(math.log(v) for v in vals) / len(vals)) if vals else float("nan")


def load_records(out: Path, acc: str, mode: str, shapes: list[str]) -> dict:
    """mnk -> {"table": {row name -> row}, "rec": in-process record or None}; both result layouts are read:
    {out}/{acc}_{mode}/{mnk}/summary.json (eval_one_file.sh) and {out}/{acc}_{mode}/rank*.jsonl (--inprocess)."""
    found = {}
    for f in sorted((out / f"{acc}_{mode}").glob("rank*.jsonl")):
        for ln in f.read_text().splitlines():
            if ln.strip():
                r = json.loads(ln)
                found[r["mnk"]] = {"table": {row["Baseline Method Name"]: row for row in r["summary"]}, "rec": r}
    for mnk in shapes:
        f = shape_dir(out, acc, mode, mnk) / "summary.json"
        if mnk not in found and f.exists():
            found[mnk] = {"table": {r["Baseline Method Name"]: r for r in json.loads(f.read_text())}, "rec": None}
    return {mnk: found[mnk] for mnk in shapes if mnk in found}


def merge(out: Path, acc: str, mode: str, shapes: list[str]) -> dict:
    """-> report dict; writes eval_results-style CSV + tflops CSV (+ latency CSV) under out/."""
    rows, tf_rows, lat_rows = [], [], []
    recs = load_records(out, acc, mode, shapes)
    total_flops = ours_time = cpu_flops = cpu_time = 0.0
    cpu_info = None
    for mnk, item in recs.items():
        table = item["table"]
        if not all(c in table for c in CSV_COLUMNS):
            continue
        rows.append([mnk] + [table[c]["Speedup"] for c in CSV_COLUMNS])
        ours_tf = table["hipBLASLt-auto-tuning-max"]["CUDA-L2 TFLOPS"]
        cpu_tf = (item["rec"] or {}).get("cpu", {}).get("cpu_matmul_tflops")
        tf_rows.append([mnk, ours_tf, table["hipBLASLt-auto-tuning-max"]["Baseline TFLOPS"], table["torch.matmul"]["Baseline TFLOPS"], cpu_tf])
        total_flops += flops(mnk)
        ours_time += flops(mnk) / (ours_tf * 1e12)
        if cpu_tf:
            cpu_flops += flops(mnk); cpu_time += flops(mnk) / (cpu_tf * 1e12); cpu_info = item["rec"]["cpu"]
        if item["rec"] is not None:
            lat = item["rec"]["latency_ms"]
            ours = lat.get(f"cuda_l2_mi355x_{acc}", {})
            base = lat.get("hipBLASLt-auto-tuning-tn", lat.get("hipBLASLt-heuristic-tn", {}))
            lat_rows.append([mnk, ours.get("p50"), ours.get("p99"), base.get("p50"), base.get("p99")])
    # CPU plumbing records (run --device cpu): torch.matmul on the host, the harness's own numbers
    plumbing = {}
    for mnk in shapes:
        f = shape_dir(out, acc, mode, mnk) / "benchmark_result_matmul.json"
        if f.exists():
            r = json.loads(f.read_text())
            if r.get("device") == "cpu":
                plumbing[mnk] = r
    if plumbing and not cpu_time:
        for mnk, r in plumbing.items():
            tf = r["records"]["matmul"]
            cpu_flops += flops(mnk); cpu_time += flops(mnk) / (tf * 1e12); cpu_info = r.get("cpu")
            tf_rows.append([mnk, None, None, None, tf])
        if not total_flops:
            total_flops = cpu_flops
    name = f"cuda_l2_mi355x_{ACC_DIRS[acc]}_speedup_{mode}.csv"
    with open(out / name, "w") as f:
        f.write("mnk," + ",".join(CSV_COLUMNS) + "\n")
        for r in rows:
            f.write(r[0] + "," + ",".join(f"{v:.3f}" for v in r[1:]) + "\n")
    with open(out / f"cuda_l2_mi355x_{ACC_DIRS[acc]}_tflops_{mode}.csv", "w") as f:
        # north_star: "report per-shape TFLOP/s ... with achieved %-of-fp16-MFMA-peak".  Two percentages of OUR figure (the
        # reference's host wall-clock TFLOPS, so launch-bound shapes read low by construction): of the dense fp16 MFMA peak
        # (2.5 PFLOP/s), and of the shape's own roofline min(peak, arithmetic intensity x 8 TB/s) with the algorithmic bytes
        # 2 (MK + KN + MN)
        f.write("mnk,cuda_l2_tflops,hipblaslt_autotune_max_tflops,torch_matmul_tflops,cpu_torch_matmul_tflops,cuda_l2_pct_of_fp16_mfma_peak,cuda_l2_pct_of_roofline\n")
        for r in tf_rows:
            m_, n_, k_ = (int(x) for x in r[0].split("_"))
            roof = min(MFMA_F16_PEAK_TFLOPS, flops(r[0]) / (2.0 * (m_ * k_ + k_ * n_ + m_ * n_)) * HBM_PEAK_TBPS)
            pct = ["" if r[1] is None else f"{100.0 * r[1] / MFMA_F16_PEAK_TFLOPS:.3f}", "" if r[1] is None else f"{100.0 * r[1] / roof:.3f}"]
            f.write(r[0] + "," + ",".join("" if v is None else format(v, ".4f" if i == 3 else ".3f") for i, v in enumerate(r[1:])) + "," + ",".join(pct) + "\n")
    if lat_rows:
        with open(out / f"cuda_l2_mi355x_{ACC_DIRS[acc]}_latency_{mode}.csv", "w") as f:
            f.write("mnk,cuda_l2_p50_ms,cuda_l2_p99_ms,hipblaslt_tn_p50_ms,hipblaslt_tn_p99_ms\n")
            for r in lat_rows:
                f.write(r[0] + "," + ",".join("" if v is None else f"{v:.5f}" for v in r[1:]) + "\n")
    report = {"shapes": len(rows), "cpu_plumbing_shapes": len(plumbing), "csv": str(out / name)}
    for i, c in enumerate(CSV_COLUMNS):
        col = [r[1 + i] for r in rows]
        report[f"geomean_speedup_vs_{c}"] = geomean(col)
        report[f"mean_speedup_vs_{c}"] = sum(col) / len(col) if col else float("nan")
    # A time-boxed autotune can pick a slower algorithm than the heuristic's first choice (few candidates, few samples): the
    # headline is also given against the STRONGER of the two hipBLASLt columns per shape (= the lower of our two speedups).
    ia, ih = CSV_COLUMNS.index("hipBLASLt-auto-tuning-max"), CSV_COLUMNS.index("hipBLASLt-heuristic-max")
    strongest = [min(r[1 + ia], r[1 + ih]) for r in rows]
    report["geomean_speedup_vs_hipBLASLt-strongest-of-autotune-and-heuristic"] = geomean(strongest)
    report["mean_speedup_vs_hipBLASLt-strongest-of-autotune-and-heuristic"] = sum(strongest) / len(strongest) if strongest else float("nan")
    report["shapes_faster_than_hipBLASLt-strongest"] = sum(1 for v in strongest if v > 1.0)
    # aggregate throughput of the sweep
    walls = {}
    for f in sorted((out / f"{acc}_{mode}").glob("rank*_status.json")):
        st = json.loads(f.read_text())
        walls[st["rank"]] = st["seconds"]
    report["rank_status"] = [json.loads(f.read_text()) for f in sorted((out / f"{acc}_{mode}").glob("rank*_status.json"))]
    rank_walls = {}
    for item in recs.values():
        if item["rec"] is not None:
            rank_walls["sum"] = rank_walls.get("sum", 0.0) + item["rec"]["wall_s"]
    report["aggregate"] = {
        "total_flops_one_pass": total_flops,
        "cuda_l2_flop_weighted_tflops": total_flops / ours_time * 1e-12 if ours_time else None,
        "ranks": len(walls) or 1,
        "max_rank_wall_s": max(walls.values()) if walls else None,
        "sum_shape_wall_s": rank_walls.get("sum"),
        "sweep_tflops_sum2mnk_over_max_rank_wall": total_flops / max(walls.values()) * 1e-12 if walls and max(walls.values()) > 0 else None,
        "cpu_torch_matmul": None if not cpu_time else {"flop_weighted_tflops": cpu_flops / cpu_time * 1e-12, "shapes": sum(1 for r in tf_rows if r[4]),
                                                        **{k: v for k, v in (cpu_info or {}).items() if k in ("os_cpu_count", "torch_num_threads")}},
    }
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("command", choices=["run", "merge", "plan"])
    ap.add_argument("--out", type=Path, required=True)
    ap.add_argument("--acc_precise", choices=["fp16", "fp32"], default="fp32")
    ap.add_argument("--mode", choices=["offline", "server"], default="offline")
    ap.add_argument("--target_qps", type=float, default=100.0)
    ap.add_argument("--warmup_seconds", type=float, default=5.0)
    ap.add_argument("--benchmark_seconds", type=float, default=10.0)
    ap.add_argument("--shapes", type=str, default="", help="comma separated M_N_K (default: the 1000-shape grid)")
    ap.add_argument("--shapes-file", type=str, default="")
    ap.add_argument("--gpus", type=int, default=None, help="world size when not launched by torch.distributed.run")
    ap.add_argument("--rank", type=int, default=None)
    ap.add_argument("--inprocess", action="store_true", help="run: one process per GPU evaluates its shard (tools/sweep_inprocess.py)")
    ap.add_argument("--device", choices=["cuda", "cpu"], default="cuda", help="run: cpu = plumbing sweep of torch.matmul on the host cores")
    from tools import sweep_inprocess

    sweep_inprocess.add_args(ap)
    args = ap.parse_args(argv)

    if args.shapes_file:
        shapes = [l.strip() for l in open(args.shapes_file) if l.strip()]
    elif args.shapes:
        shapes = [s for s in args.shapes.split(",") if s]
    else:
        shapes = grid_shapes()
    world = int(os.environ.get("WORLD_SIZE", args.gpus or 1))
    rank = int(os.environ.get("RANK", args.rank or 0))
    gpu = int(os.environ.get("LOCAL_RANK", rank))
    args.out.mkdir(parents=True, exist_ok=True)

    if args.command == "plan":
        for r in range(world):
            mine = shard(shapes, r, world)
            print(f"rank {r}: {len(mine)} shapes, {sum(map(flops, mine)):.3e} flop per pass, estimated {sum(map(estimated_cost, mine)):.1f} s")
        return None
    if args.command == "run" and args.inprocess:
        status = sweep_inprocess.run(args, shard(shapes, rank, world), rank, gpu)
        print(json.dumps(status))
        return status
    if args.command == "run":
        mine = shard(shapes, rank, world)
        status = run_shapes_cpu(mine, args, rank) if args.device == "cpu" else run_shapes(mine, args, rank, gpu)
        status["world"] = world
        sp = status_path(args.out, args.acc_precise, args.mode, rank)
        sp.parent.mkdir(parents=True, exist_ok=True)
        sp.write_text(json.dumps(status))
        print(json.dumps(status))
        return status
    report = merge(args.out, args.acc_precise, args.mode, shapes)
    print(json.dumps(report, indent=1))
    return report


if __name__ == "__main__":
    main()
