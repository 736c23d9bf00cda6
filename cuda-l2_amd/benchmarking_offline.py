"""Offline (back-to-back) benchmark of one baseline against cuda_l2 for one (M,N,K).

Behaviour of the reference's benchmarking_offline.py (:52-170): time-boxed warm-up loop, time-boxed
benchmark loop with the {baseline, cuda_l2} order shuffled every iteration, fresh randn inputs each
iteration, mean of per-iteration TFLOPS, result written to {base_dir}/benchmark_result_{perf_func}.json
as {"records": {perf_func: tflops, cuda_l2_<dev>_<acc>: tflops, "version": ...}}.
Additions: --device cpu (plumbing run of torch.matmul on the host: BASELINE config 1), --seed,
and a "latency_ms" block (mean / p50 / p99 of the per-iteration milliseconds).
"""
import argparse
import gc
import json
import os
import random
import time

import pandas
import torch

from benchmarking_utils import run_all_perf_funcs_once
from harness_common import (RESULT_VERSION, add_common_args, cpu_cores, destroy_baselines, init_baselines, insitu_report,
                            load_kernel, parse_mnk, percentile, seed_everything)

torch.set_grad_enabled(False)

MODE_NAME = "Offline"


def build_arg_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=__doc__)
    add_common_args(parser, benchmark=True)
    return parser


def inter_arrival_sleep(args) -> None:
    """Offline mode issues requests back to back (server mode overrides this)."""


def prepare_perf_funcs(args, m, n, k):
    """-> (perf_func_list, cuda_l2_name or None, paddings, hgemm module or None)"""
    if args.device == "cpu":
        if args.perf_func != "matmul":
            raise SystemExit("--device cpu can only time --perf_func matmul (no HIP extension on the host)")
        return [torch.matmul], None, (0, 0, 0), None
    torch.cuda.set_device(args.gpu_device_id)
    load_start = time.time()
    kernel = load_kernel(args.mnk, args.acc_precise, args.device_type, args.base_dir)
    print(f"Load hgemm module time: {time.time() - load_start:.2f} seconds")
    hgemm = kernel.module
    init_baselines(hgemm)
    # the reference tests `args.perf_func in "<name>"` (a substring test); equality is what is meant
    if args.perf_func == "hgemm_cublaslt_auto_tuning_tn":
        t0 = time.time()
        hgemm.find_best_algo_tn_v2_torch(m, n, k)
        torch.cuda.synchronize()
        print(f"Find best algo time: {time.time() - t0:.2f} seconds")
    elif args.perf_func == "hgemm_cublaslt_auto_tuning_nn":
        t0 = time.time()
        hgemm.find_best_algo_nn_v2_torch(m, n, k)
        torch.cuda.synchronize()
        print(f"Find best algo time: {time.time() - t0:.2f} seconds")
    baseline = torch.matmul if args.perf_func == "matmul" else getattr(hgemm, args.perf_func)
    return [baseline, kernel.cuda_l2_func], kernel.cuda_l2_func_name, kernel.padding, hgemm


def timed_loop(seconds, perf_funcs, shuffle, args, m, n, k, pads):
    records = []
    start = time.time()
    while time.time() - start < seconds:
        if shuffle:
            random.shuffle(perf_funcs)
        rec = run_all_perf_funcs_once(perf_func_list=perf_funcs, m=m, n=n, k=k, acc_precise=args.acc_precise,
                                      device_type=args.device_type, padding_m=pads[0], padding_k=pads[1],
                                      padding_n=pads[2], device=args.device)
        rec["idx"] = len(records)
        records.append(rec)
        inter_arrival_sleep(args)
    return records, time.time() - start


def run(args, extra: dict | None = None) -> dict:
    print(f"=====================Benchmarking Script -- {MODE_NAME} Mode======================")
    seed_everything(args.seed)
    m, n, k = parse_mnk(args.mnk)
    start_time = time.time()
    print(f"m={m}, n={n}, k={k}, Warmup={args.warmup_seconds}s, Benchmark={args.benchmark_seconds}s")
    perf_funcs, cuda_l2_func_name, pads, hgemm = prepare_perf_funcs(args, m, n, k)
    names = [f.__name__ for f in perf_funcs]  # report order: baseline first
    print(f"Using padding_m={pads[0]}, padding_k={pads[1]}, padding_n={pads[2]}")

    print("Warmup...")
    warm, warm_s = timed_loop(args.warmup_seconds, list(perf_funcs), False, args, m, n, k, pads)
    print(f"Warmup done: {len(warm)} iterations in {warm_s:.2f} seconds.")
    # first-use plan selection (eval_one_file.sh --insitu): the first cuda_l2 call above timed the plan and its alternates
    insitu = insitu_report(m, n, k) if hgemm is not None else None
    if insitu is not None:
        print(f"in-situ plan selection: {insitu}")
    print("Benchmarking...")
    records, _ = timed_loop(args.benchmark_seconds, list(perf_funcs), True, args, m, n, k, pads)

    if hgemm is not None:
        destroy_baselines(hgemm)
        gc.collect()
        torch.cuda.empty_cache()
    print(f"Total time: {(time.time() - start_time):.2f} seconds, {len(records)} records collected.")
    if not records:
        raise SystemExit("benchmark window too short: no iteration completed")

    ms_names = [x + "_ms" for x in names]
    df = pandas.DataFrame.from_records(records, columns=["idx"] + names + ms_names)
    print(df.head().to_markdown())
    print(df.tail().to_markdown())
    merged = df[names].mean().to_dict()
    merged["version"] = RESULT_VERSION
    print(merged)
    print(df[ms_names].mean())
    latency = {x: {"mean": float(df[x + "_ms"].mean()), "p50": percentile(df[x + "_ms"], 50),
                   "p99": percentile(df[x + "_ms"], 99)} for x in names}
    result = {"records": merged, "latency_ms": latency, "iterations": len(records), "mode": MODE_NAME.lower(),
              "device": args.device}
    if args.device == "cpu":
        result["cpu"] = cpu_cores()
    if insitu is not None:
        result["insitu"] = insitu
    result.update(extra or {})
    if cuda_l2_func_name is not None:
        print(f"speedup over {args.perf_func}: {merged[cuda_l2_func_name] / merged[names[0]]:.2f}x")
    os.makedirs(args.base_dir, exist_ok=True)
    with open(os.path.join(args.base_dir, f"benchmark_result_{args.perf_func}.json"), "w") as f:
        json.dump(result, f)
    return result


def main(argv=None):
    run(build_arg_parser().parse_args(argv))
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
