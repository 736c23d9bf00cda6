"""Files the outputs of the closing GPU run (tools/lab/gpu_round4_g.sh -> gpurun_out/r4g/) and of call E's sweeps (gpurun_out/sweep_r04/) into the
tree: check log, parity / tolerance records, plan reports (+ their text summaries), per-geometry PMC table (+ the BASELINE shapes' files bench.py reads),
the sweep records with the re-measured rows spliced in by shape, the merged CSVs, the generated README and the figure.  Run from the repository root;
needs no GPU.  (gpurun_out/ is scratch and not committed: this script is the record of how the committed files were derived from it.)"""
import json
import shutil
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[3]
PKG = REPO / "cuda-l2_amd"
O, E, T, P = REPO / "gpurun_out" / "r4g", REPO / "gpurun_out" / "sweep_r04", PKG / "tuning", REPO / "profiles"


def run(*cmd, cwd=PKG, stdout=None):
    subprocess.run([sys.executable, *map(str, cmd)], check=True, cwd=cwd, stdout=stdout)


def main():
    shutil.copy(O / "check_final.log", P / "r04_check_final.log")
    for src, dst in (("grid_20260925.jsonl", "r04_parity_1000.jsonl"), ("grid_randn.jsonl", "r04_randn_1000.jsonl"), ("offgrid.jsonl", "r04_offgrid_parity.jsonl"),
                     ("offgrid_randn.jsonl", "r04_offgrid_randn.jsonl")):
        shutil.copy(O / "records" / src, T / dst)
    shutil.copy(O / "grid_plan_report.jsonl", T / "r04_grid_plan_report_mi355x.jsonl")
    shutil.copy(O / "offgrid_plan_report.jsonl", T / "r04_offgrid_plan_report_mi355x.jsonl")
    shutil.copy(O / "pmc_table.json", P / "r04_pmc_table.json")
    run("tools/pmc_table.py", "baseline", P / "r04_pmc_table.json", P)
    for name in ("r04_grid_plan_report", "r04_offgrid_plan_report"):
        with open(T / f"{name}.txt", "w") as f:
            run("tools/tune_report.py", T / f"{name}_mi355x.jsonl", stdout=f)
    # sweeps: call E's records with the rows that changed afterwards replaced by the closing run's measurements of the shipped plans
    changed = set((T / "r04_rows_changed_after_call_e.txt").read_text().split())
    work = REPO / "gpurun_out" / "sweep_r04_final"
    shutil.rmtree(work, ignore_errors=True)
    dest = PKG / "eval_results" / "r04_sweep"
    (dest / "records").mkdir(parents=True, exist_ok=True)
    for acc in ("fp32", "fp16"):
        for mode in ("offline", "server"):
            d = work / f"{acc}_{mode}"
            d.mkdir(parents=True)
            new = {json.loads(ln)["mnk"]: ln for ln in open(O / "sweep_changed_rows" / f"{acc}_{mode}" / "rank0.jsonl")}
            assert set(new) == changed, (acc, mode, len(new), len(changed))
            out = []
            for ln in open(E / f"{acc}_{mode}" / "rank0.jsonl"):
                mnk = json.loads(ln)["mnk"]
                if mnk in new:
                    r = json.loads(new[mnk])
                    r["remeasured_in"] = "tools/lab/gpu_round4_g.sh (the row changed after call E's sweep)"
                    out.append(json.dumps(r) + "\n")
                else:
                    out.append(ln)
            (d / "rank0.jsonl").write_text("".join(out))
            shutil.copy(E / f"{acc}_{mode}" / "rank0_status.json", d / "rank0_status.json")
            with open(work / f"merge_{acc}_{mode}.json", "w") as f:
                run("tools/sweep.py", "merge", "--out", work, "--acc_precise", acc, "--mode", mode, "--shapes-file", "tools/grid_shapes.txt", stdout=f)
            shutil.copy(d / "rank0.jsonl", dest / "records" / f"{acc}_{mode}_rank0.jsonl")
            text = (work / f"merge_{acc}_{mode}.json").read_text().replace(str(work) + "/", "eval_results/r04_sweep/").replace("../gpurun_out/sweep_r04_final/", "eval_results/r04_sweep/")
            (dest / f"merge_{acc}_{mode}.json").write_text(text)
    for f in work.glob("*.csv"):
        shutil.copy(f, dest / f.name)
    run("tools/sweep_readme.py", "eval_results/r04_sweep", "--grid-report", "tuning/r04_grid_plan_report_mi355x.jsonl", "--autotune-report",
        "tuning/r04_quarter_grid_plan_report_autotune_mi355x.jsonl", stdout=subprocess.DEVNULL)
    run("tools/plot_speedups.py", "--csv", "eval_results/r04_sweep/cuda_l2_mi355x_F32F16F16F32_speedup_offline.csv", "--grid", "tuning/r04_grid_plan_report_mi355x.jsonl",
        "--out", "assets/r04_speedup_summary.png", "--title", "MI355X, fp32-acc, round-4 table")
    print("filed")


if __name__ == "__main__":
    main()
