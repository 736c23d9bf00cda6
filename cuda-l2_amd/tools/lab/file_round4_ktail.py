"""Files the outputs of call K (tools/lab/gpu_round4_k.sh -> gpurun_out/r4k/), the closing run of the library with the ktail kernel
variants: check logs, the suite's parity / tolerance records of the shipped table and of the off-grid list, the off-grid plan
report of the final planner (the report of the closing run G, before the K tails of families q and r, is kept beside it for the
before / after table of DESIGN.md), rocprofv3 kernel stats of one K-tail problem, the PMC table of six K-tail shapes.
Run from the repository root; needs no GPU."""
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[3]
PKG = REPO / "cuda-l2_amd"
O, T, P = REPO / "gpurun_out" / "r4k", PKG / "tuning", REPO / "profiles"


def main():
    shutil.copy(O / "check_final.log", P / "r04_check_final.log")
    shutil.copy(O / "check_q_item_seams.log", P / "r04_check_q_item_seams.log")
    for src, dst in (("grid_20260925.jsonl", "r04_parity_1000.jsonl"), ("grid_randn.jsonl", "r04_randn_1000.jsonl"), ("offgrid.jsonl", "r04_offgrid_parity.jsonl"),
                     ("offgrid_randn.jsonl", "r04_offgrid_randn.jsonl")):
        shutil.copy(O / "records" / src, T / dst)
    before = T / "r04_offgrid_plan_report_before_ktail_mi355x.jsonl"
    if not before.exists():   # (the report of closing run G, as committed in e99337e)
        before.write_bytes(subprocess.run(["git", "show", "e99337e:cuda-l2_amd/tuning/r04_offgrid_plan_report_mi355x.jsonl"], check=True, cwd=REPO,
                                          capture_output=True).stdout)
    shutil.copy(O / "offgrid_plan_report.jsonl", T / "r04_offgrid_plan_report_mi355x.jsonl")
    with open(T / "r04_offgrid_plan_report.txt", "w") as f:
        subprocess.run([sys.executable, "tools/tune_report.py", T / "r04_offgrid_plan_report_mi355x.jsonl"], check=True, cwd=PKG, stdout=f)
    shutil.copy(O / "pmc_ktail_table.json", P / "r04_pmc_ktail_table.json")
    stats = sorted((O / "prof").glob("**/*_kernel_stats.csv"))
    shutil.copy(stats[0], P / "r04_ktail_4000_4000_4000_kernel_stats.csv")
    print("filed")


if __name__ == "__main__":
    main()
