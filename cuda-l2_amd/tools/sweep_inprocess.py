"""In-process shape sweep in the reference's metric -- what `tools/sweep.py run --inprocess` executes.

`tools/sweep.py run` evaluates a shape the way the reference does: eval_one_file.sh starts one Python process
per baseline (7 per shape, each importing torch and the extension), which costs ~1 minute per shape before a
single GEMM is timed: 1000 shapes x 2 accumulate trees x 2 modes = ~70 GPU-hours.  This driver keeps the
MEASUREMENT of the reference and drops the process churn:

  * one process per (GPU, accumulate tree, mode) loops over its shard of the shape list (tools/sweep.py shard(): cost-sorted,
    as tools/sweep.py), with ONE prebuilt `hgemm_lib` extension (the per-shape kernel files only pin the plan
    the library's tuned table already holds, so the generic entry point runs the same plan);
  * per shape and per baseline X the inner loop is the reference's (benchmarking_offline.py:115-137 /
    benchmarking_server.py:127-145): fresh randn fp16 operands, [X, cuda_l2] in shuffled order, each call timed
    by benchmarking_utils.run_benchmark (fill_(0); sync; t0; call; sync; t1 -- host wall-clock), TFLOPS =
    2MNK/t, mean over iterations; server mode sleeps Exp(1/target_qps) after every pair.  The seven baselines
    take turns round by round inside one time box instead of one process each, so all of them see the same
    thermal / clock history;
  * hipBLASLt autotune (find_best_algo_{nn,tn}_v2) runs once per shape, time-boxed by HGEMM_AUTOTUNE_MAX_SECONDS;
  * speedup rows follow summarize_result.py:43-53 (the "-max" row is the tn/nn variant against which cuda_l2's
    speedup is LOWER); p50 / p99 of the per-call milliseconds are kept per function (BASELINE config 4);
  * optional CPU column: torch.matmul on the host cores for shapes up to --cpu_max_flops.

Output: one JSON line per shape in {out}/{acc}_{mode}/rank{r}.jsonl (resumable); `tools/sweep.py merge` turns
the lines of all ranks into the eval_results CSVs.  The time boxes are flags; the committed sweeps state theirs.
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

PKG_DIR = Path(__file__).resolve().parent.parent
if str(PKG_DIR) not in sys.path:
    sys.path.insert(0, str(PKG_DIR))

from benchmarking_utils import run_all_perf_funcs_once  # noqa: E402
from harness_common import (BASELINE_PERF_FUNCS, cpu_cores, cuda_l2_name, destroy_baselines, init_baselines,  # noqa: E402
                            parse_mnk, percentile)
from summarize_result import NAME_ORDER, show_name  # noqa: E402

torch.set_grad_enabled(False)


def load_generic_extension(acc_precise: str, device_type: str, base_dir: Path):
    """The `hgemm_lib` extension built for one anchor shape; every other shape takes its fallback branch
    (hgemm_shape_entry.hpp), i.e. the library's tuned plan for that shape."""
    from tools.utils import build_from_sources

    return build_from_sources(mnk="64_4096_64", acc_precise=acc_precise, device_type=device_type, base_dir=str(base_dir), verbose=False)


def summarize_pairs(series: dict, cuda_name: str) -> list[dict]:
    """series[baseline] = {"base": [tflops...], "ours": [tflops...]} -> the reference's 10-row table."""
    rows = {}
    for method, s in series.items():
        if not s["base"]:
            continue
        b, o = float(np.mean(s["base"])), float(np.mean(s["ours"]))
        rows[show_name(method)] = {"Baseline Method Name": show_name(method), "Baseline TFLOPS": b, "CUDA-L2 TFLOPS": o,
                                   "Speedup": o / b}
    for family in ("rocBLAS", "hipBLASLt-heuristic", "hipBLASLt-auto-tuning"):
        tn, nn = rows.get(f"{family}-tn"), rows.get(f"{family}-nn")
        if tn is None or nn is None:
            continue
        worst = tn if tn["Speedup"] < nn["Speedup"] else nn
        rows[f"{family}-max"] = dict(worst, **{"Baseline Method Name": f"{family}-max"})
    return [rows[n] for n in NAME_ORDER if n in rows]


def eval_shape(hgemm, mnk: str, args, rng: np.random.Generator) -> dict:
    m, n, k = parse_mnk(mnk)
    cuda_name = cuda_l2_name(args.device_type, args.acc_precise)
    cuda_l2 = getattr(hgemm, cuda_name)
    t_find = time.time()
    auto_ok = True
    try:
        hgemm.find_best_algo_tn_v2_torch(m, n, k)
        hgemm.find_best_algo_nn_v2_torch(m, n, k)
    except RuntimeError:
        auto_ok = False
    torch.cuda.synchronize()
    t_find = time.time() - t_find
    funcs = {}
    for name in BASELINE_PERF_FUNCS:
        if "auto_tuning" in name and not auto_ok:
            continue
        funcs[name] = torch.matmul if name == "matmul" else getattr(hgemm, name)
    series = {name: {"base": [], "ours": [], "base_ms": [], "ours_ms": []} for name in funcs}

    def one_round(record: bool) -> None:
        order = list(funcs)
        random.shuffle(order)
        for name in order:
            pair = [funcs[name], cuda_l2]
            random.shuffle(pair)
            rec = run_all_perf_funcs_once(perf_func_list=pair, m=m, n=n, k=k, acc_precise=args.acc_precise, device_type=args.device_type,
                                          padding_m=0, padding_k=0, padding_n=0)
            if record:
                fname = funcs[name].__name__
                series[name]["base"].append(rec[fname]); series[name]["base_ms"].append(rec[fname + "_ms"])
                series[name]["ours"].append(rec[cuda_name]); series[name]["ours_ms"].append(rec[cuda_name + "_ms"])
            if args.mode == "server":
                time.sleep(rng.exponential(1.0 / args.target_qps))

    t0 = time.time()
    warm_rounds = 0
    while warm_rounds < 1 or time.time() - t0 < args.warmup_seconds:
        one_round(False)
        warm_rounds += 1
    t1 = time.time()
    rounds = 0
    while rounds < args.min_rounds or time.time() - t1 < args.benchmark_seconds:
        one_round(True)
        rounds += 1
    wall = time.time() - t0
    ours_ms = [x for s in series.values() for x in s["ours_ms"]]
    out = {"mnk": mnk, "acc_precise": args.acc_precise, "mode": args.mode, "rounds": rounds, "warmup_rounds": warm_rounds,
           "wall_s": round(wall, 4), "autotune_find_s": round(t_find, 3), "autotune_ok": auto_ok,
           "autotune_budget_s": float(os.environ.get("HGEMM_AUTOTUNE_MAX_SECONDS", "30")),
           "summary": summarize_pairs(series, cuda_name),
           "latency_ms": {cuda_name: {"mean": float(np.mean(ours_ms)), "p50": percentile(ours_ms, 50), "p99": percentile(ours_ms, 99)}}}
    for name, s in series.items():
        out["latency_ms"][show_name(name)] = {"mean": float(np.mean(s["base_ms"])), "p50": percentile(s["base_ms"], 50),
                                              "p99": percentile(s["base_ms"], 99)}
    if args.mode == "server":
        out["target_qps"] = args.target_qps
    # hipBLASLt fp16-compute availability (SURVEY: "fall back to 32F and say so")
    try:
        import ctypes

        lib = ctypes.CDLL(str(PKG_DIR / "lib" / "libhgemm_mi355x.so"))
        out["hipblaslt_compute16_fallback"] = {"heuristic_tn": lib.hgemm_hipblaslt_compute16_fallback(0, 1),
                                               "autotune_tn": lib.hgemm_hipblaslt_compute16_fallback(1, 1)}
        # (round 6) the search result came from the on-disk cache of winners (HGEMM_AUTOTUNE_CACHE) / was searched here
        lib.hgemm_hipblaslt_autotune_best_ms.restype = ctypes.c_double
        out["autotune"] = {lay: {"from_cache": bool(lib.hgemm_hipblaslt_autotune_from_cache(t)), "candidates": lib.hgemm_hipblaslt_autotune_candidates(t),
                                 "search_median_ms": lib.hgemm_hipblaslt_autotune_best_ms(t)} for lay, t in (("tn", 1), ("nn", 0))}
    except OSError:
        pass
    return out


def cpu_matmul_tflops(mnk: str, seconds: float) -> dict:
    """The reference's CPU expression (a.cpu().float() @ b.cpu().float()).half() (zero_one_correctness_check.py:85-90)
    on the host cores.  (torch.matmul on fp16 CPU tensors, what `--device cpu --perf_func matmul` times, runs an
    unvectorised path on this host: ~0.001 TFLOP/s; the fp32 expression is the meaningful CPU baseline.)"""
    m, n, k = parse_mnk(mnk)
    a = torch.randn((m, k)).half()
    b = torch.randn((k, n)).half()
    t0 = time.time()
    it = 0
    while it < 1 or time.time() - t0 < seconds:
        torch.matmul(a.float(), b.float()).half()
        it += 1
    dt = (time.time() - t0) / it
    return {"cpu_matmul_tflops": 2.0 * m * n * k / dt * 1e-12, "cpu_iterations": it, "cpu_expression": "(a.float() @ b.float()).half()",
            **cpu_cores()}


def run(args, shapes: list[str], rank: int, gpu: int) -> dict:
    if not torch.cuda.is_available():
        raise SystemExit("the sweep needs a visible MI355X")
    torch.cuda.set_device(gpu)
    out_dir = args.out / f"{args.acc_precise}_{args.mode}"
    out_dir.mkdir(parents=True, exist_ok=True)
    path = out_dir / f"rank{rank}.jsonl"
    done = set()
    if path.exists():
        done = {json.loads(ln)["mnk"] for ln in path.read_text().splitlines() if ln.strip()}
    if args.seed is not None:
        random.seed(args.seed + rank); np.random.seed(args.seed + rank); torch.manual_seed(args.seed + rank)
    rng = np.random.default_rng(None if args.seed is None else args.seed + rank)
    hgemm = load_generic_extension(args.acc_precise, args.device_type, PKG_DIR / "build" / f"ext_{args.acc_precise}")
    init_baselines(hgemm)
    t0 = time.time()
    n_done = 0
    with open(path, "a") as f:
        for mnk in shapes:
            if mnk in done:
                continue
            rec = eval_shape(hgemm, mnk, args, rng)
            m, n, k = parse_mnk(mnk)
            if args.cpu_max_flops > 0 and 2.0 * m * n * k <= args.cpu_max_flops:
                rec["cpu"] = cpu_matmul_tflops(mnk, args.cpu_seconds)
            f.write(json.dumps(rec) + "\n")
            f.flush()
            n_done += 1
            if args.time_limit and time.time() - t0 > args.time_limit:
                break
    destroy_baselines(hgemm)
    status = {"rank": rank, "gpu": gpu, "done": n_done, "skipped": len(done), "seconds": round(time.time() - t0, 1),
              "remaining": len([s for s in shapes if s not in done]) - n_done}
    (out_dir / f"rank{rank}_status.json").write_text(json.dumps(status))
    return status


def add_args(ap: argparse.ArgumentParser) -> None:
    ap.add_argument("--min_rounds", type=int, default=3, help="recorded rounds per shape at least (one round = every baseline once)")
    ap.add_argument("--device_type", default="mi355x")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cpu_max_flops", type=float, default=0.0, help="also time torch.matmul on the host for shapes up to this many flop")
    ap.add_argument("--cpu_seconds", type=float, default=0.05)
    ap.add_argument("--time_limit", type=float, default=0.0, help="stop after this many seconds (resumable)")
