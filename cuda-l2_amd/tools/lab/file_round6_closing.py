"""Files the outputs of round 6's closing GPU calls into the tree (run from the repository root; needs no GPU):

  call K (tools/lab/gpu_round6_k.sh -> gpurun_out/r6k/): check logs, parity / tolerance records of the shipped table, off-grid plan report,
      bench.py record + rocprofv3 kernel stats of the same command, per-geometry PMC table (+ the BASELINE shapes' files bench.py reads);
  calls L1 / L2 (gpu_round6_l.sh -> gpurun_out/r6l/): the four reference-metric sweeps, BASELINE config 4 at three request rates, the
      autotune cache as the sweeps left it (the torch wheel's hipBLASLt build completed), merged CSVs, generated README, figure;
  call M (gpu_round6_m.sh -> gpurun_out/r6m/): the whole-grid plan report with interleaved contenders (+ text summary), the reversed run on
      the >= 1e11-FLOP shapes, the energy table;
  the library's ISA fingerprint (tools/isa_fingerprint.py: what the closing run measured).

gpurun_out/ is scratch and not committed: this script is the record of how the committed files were derived from it.  Parts whose
inputs are missing are skipped (the calls arrive one by one)."""
import json
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[3]
PKG = REPO / "cuda-l2_amd"
K, L, M, T, P = (REPO / "gpurun_out" / "r6k", REPO / "gpurun_out" / "r6l", REPO / "gpurun_out" / "r6m", PKG / "tuning", REPO / "profiles")


def run(*cmd, cwd=PKG, stdout=None):
    subprocess.run([sys.executable, *map(str, cmd)], check=True, cwd=cwd, stdout=stdout)


def part_k():
    shutil.copy(K / "check_final.log", P / "r06_check_final.log")
    shutil.copy(K / "check_q_walk.log", P / "r06_check_q_item_walk.log")
    shutil.copy(K / "check_w_deep.log", P / "r06_check_w_deep_trips.log")
    for src, dst in (("grid_20260925.jsonl", "r06_parity_1000.jsonl"), ("grid_randn.jsonl", "r06_randn_1000.jsonl"), ("offgrid.jsonl", "r06_offgrid_parity.jsonl"),
                     ("offgrid_randn.jsonl", "r06_offgrid_randn.jsonl")):
        shutil.copy(K / "records" / src, T / dst)
    shutil.copy(K / "offgrid_plan_report.jsonl", T / "r06_offgrid_plan_report_mi355x.jsonl")
    with open(T / "r06_offgrid_plan_report.txt", "w") as f:
        run("tools/tune_report.py", T / "r06_offgrid_plan_report_mi355x.jsonl", stdout=f)
    shutil.copy(K / "bench.json", P / "r06_bench.json")
    shutil.copy(K / "bench_profiled.json", P / "r06_bench_py_profiled_run.json")
    stats = sorted((K / "prof").rglob("*kernel_stats.csv"))
    assert len(stats) == 1, stats
    shutil.copy(stats[0], P / "r06_bench_py_kernel_stats.csv")
    shutil.copy(K / "pmc_table.json", P / "r06_pmc_table.json")
    run("tools/pmc_table.py", "baseline", P / "r06_pmc_table.json", P)
    with open(P / "r06_isa_fingerprint_closing_run_library.json", "w") as f:
        run("tools/isa_fingerprint.py", "--note", "round 6, the library the closing run (tools/lab/gpu_round6_k.sh) measured", stdout=f)


def part_l():
    dest = PKG / "eval_results" / "r06_sweep"
    (dest / "records").mkdir(parents=True, exist_ok=True)
    (dest / "config4").mkdir(parents=True, exist_ok=True)
    for acc in ("fp32", "fp16"):
        for mode in ("offline", "server"):
            d = L / f"{acc}_{mode}"
            if not (d / "rank0.jsonl").exists():
                continue
            with open(L / f"merge_{acc}_{mode}.json", "w") as f:
                run("tools/sweep.py", "merge", "--out", L, "--acc_precise", acc, "--mode", mode, "--shapes-file", "tools/grid_shapes.txt", stdout=f)
            shutil.copy(d / "rank0.jsonl", dest / "records" / f"{acc}_{mode}_rank0.jsonl")
            text = (L / f"merge_{acc}_{mode}.json").read_text().replace(str(L) + "/", "eval_results/r06_sweep/").replace("../gpurun_out/r6l/", "eval_results/r06_sweep/")
            (dest / f"merge_{acc}_{mode}.json").write_text(text)
    for f in L.glob("*.csv"):
        shutil.copy(f, dest / f.name)
    for q in (10, 100, 1000):
        f = L / "config4" / f"qps_{q}" / "fp32_server" / "rank0.jsonl"
        if f.exists():
            shutil.copy(f, dest / "config4" / f"qps_{q}.jsonl")
    cache = L / "r06_hipblaslt_autotune_cache.txt"
    if cache.exists():
        shutil.copy(cache, T / "r06_hipblaslt_autotune_cache.txt")
    run("tools/sweep_readme_r06.py", "eval_results/r06_sweep", stdout=subprocess.DEVNULL)
    grid = T / "r06_grid_plan_report_autotune_interleaved_mi355x.jsonl"
    if grid.exists() and (dest / "cuda_l2_mi355x_F32F16F16F32_speedup_offline.csv").exists():
        run("tools/plot_speedups.py", "--csv", "eval_results/r06_sweep/cuda_l2_mi355x_F32F16F16F32_speedup_offline.csv", "--grid", grid,
            "--out", "assets/r06_speedup_summary.png", "--title", "MI355X, fp32-acc, round-6 table")


def part_m():
    shutil.copy(M / "grid_plan_report_autotune_interleaved.jsonl", T / "r06_grid_plan_report_autotune_interleaved_mi355x.jsonl")
    shutil.copy(M / "grid_1e11_up_plan_report_autotune_interleaved_reversed.jsonl", T / "r06_grid_1e11_up_plan_report_autotune_interleaved_reversed_mi355x.jsonl")
    with open(T / "r06_grid_plan_report_autotune_interleaved.txt", "w") as f:
        run("tools/tune_report.py", T / "r06_grid_plan_report_autotune_interleaved_mi355x.jsonl", 8, stdout=f)
    # the off-grid planner rule changed after call K (host code): its records and report are call M's
    for src, dst in (("offgrid.jsonl", "r06_offgrid_parity.jsonl"), ("offgrid_randn.jsonl", "r06_offgrid_randn.jsonl")):
        if (M / "records" / src).exists():
            shutil.copy(M / "records" / src, T / dst)
    if (M / "offgrid_plan_report.jsonl").exists():
        shutil.copy(M / "offgrid_plan_report.jsonl", T / "r06_offgrid_plan_report_mi355x.jsonl")
        with open(T / "r06_offgrid_plan_report.txt", "w") as f:
            run("tools/tune_report.py", T / "r06_offgrid_plan_report_mi355x.jsonl", stdout=f)
    power = REPO / "gpurun_out" / "r6p" / "power.jsonl"
    if power.exists():
        with open(P / "r06_power_table.json", "w") as f:
            run("tools/power_table.py", power, stdout=f)


def main():
    for name, part, probe in (("K", part_k, K / "bench.json"), ("M", part_m, M / "grid_plan_report_autotune_interleaved.jsonl"), ("L", part_l, L / "fp32_offline" / "rank0.jsonl")):
        if probe.exists():
            part()
            print(f"filed call {name}")
        else:
            print(f"call {name}: nothing to file yet")


if __name__ == "__main__":
    main()
