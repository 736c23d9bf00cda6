"""Figures for a sweep (row f1: the reference ships assets/speedup_summary_all.png next to its CSVs).

    python tools/plot_speedups.py --csv eval_results/r01_minisweep_fp32_offline/cuda_l2_mi355x_F32F16F16F32_speedup_offline.csv \
           --grid tuning/r01_grid_plan_report_mi355x.jsonl --out assets/r01_speedup_summary.png

Left: mean speedup (%) of cuda_l2_mi355x over the four baseline groups of the merged eval_results-format CSV
(reference convention: arithmetic mean over shapes of the `-max` column of each group, README "Speed
Comparison").  Right: per-shape device-time speedup over hipBLASLt-heuristic-max from a `hgemm_tune tune
--baselines` jsonl, against the GEMM's FLOP count.
"""
from __future__ import annotations

import argparse
import csv
import json
import math

GROUPS = [("torch.matmul", "torch.matmul"), ("rocBLAS", "rocBLAS-max"), ("hipBLASLt-heuristic", "hipBLASLt-heuristic-max"),
          ("hipBLASLt-AutoTuning", "hipBLASLt-auto-tuning-max")]
COLORS = ["#8ecfc9", "#f4a6a0", "#b5d99c", "#f0c75e"]


def mean_speedups(csv_path: str) -> tuple[list[float], int]:
    rows = list(csv.DictReader(open(csv_path)))
    return [sum(float(r[col]) for r in rows) / len(rows) for _, col in GROUPS], len(rows)


def grid_points(jsonl_path: str) -> tuple[list[float], list[float]]:
    xs, ys = [], []
    for line in open(jsonl_path):
        r = json.loads(line)
        m, n, k = (int(v) for v in r["mnk"].split("_"))
        xs.append(2.0 * m * n * k)
        ys.append(min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"])
    return xs, ys


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--csv", required=True)
    ap.add_argument("--grid", default="")
    ap.add_argument("--out", required=True)
    ap.add_argument("--title", default="MI355X, fp32-acc")
    a = ap.parse_args(argv)
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    means, nshapes = mean_speedups(a.csv)
    fig, axes = plt.subplots(1, 2 if a.grid else 1, figsize=(13 if a.grid else 6, 4.6), squeeze=False)
    ax = axes[0][0]
    bars = ax.bar([g for g, _ in GROUPS], [(m - 1) * 100 for m in means], color=COLORS, edgecolor="#555")
    for b, m in zip(bars, means):
        ax.text(b.get_x() + b.get_width() / 2, b.get_height() + 0.8, f"{(m - 1) * 100:+.1f}%", ha="center", fontsize=9, weight="bold")
    ax.set_ylabel("mean speedup of cuda_l2_mi355x (%)")
    ax.set_title(f"(a) reference metric, {nshapes} shapes — {a.title}")
    ax.grid(axis="y", ls="--", alpha=0.4)
    ax.tick_params(axis="x", labelsize=8)
    if a.grid:
        xs, ys = grid_points(a.grid)
        ax = axes[0][1]
        ax.scatter(xs, ys, s=6, alpha=0.6, color="#3b7dd8")
        ax.axhline(1.0, color="k", lw=0.8)
        gm = math.exp(sum(map(math.log, ys)) / len(ys))
        ax.axhline(gm, color="#d8433b", lw=1.0, ls="--", label=f"geomean {gm:.2f}x")
        ax.set_xscale("log")
        ax.set_xlabel("2·M·N·K (FLOP)")
        ax.set_ylabel("device-time speedup over hipBLASLt-heuristic-max")
        ax.set_title(f"(b) {len(xs)} grid shapes, shipped plans")
        ax.legend(loc="upper right", fontsize=9)
        ax.grid(ls="--", alpha=0.4)
    fig.tight_layout()
    fig.savefig(a.out, dpi=130)
    print(a.out)


if __name__ == "__main__":
    main()
