"""Prints DESIGN.md section 6.7 (round-5 figures; the paragraph on the report's order sensitivity -- calls H / J / K -- was generated the same way and is kept in DESIGN.md) from the committed records, so that no number in it is typed by hand:
python cuda-l2_amd/tools/lab/design_round5_figures.py > /tmp/section_6_7.md ; tests/test_evidence_consistency.py recomputes the same figures."""
import csv
import io
import json
import math
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[3]
PKG = REPO / "cuda-l2_amd"
sys.path.insert(0, str(PKG / "tools"))
import tune_report  # noqa: E402


def gm(xs):
    xs = list(xs)
    return math.exp(sum(map(math.log, xs)) / len(xs))


def recs(p):
    return [json.loads(l) for l in open(p) if l.strip()]


def quiet_report(path):
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        return tune_report.main(str(path), 0)
    finally:
        sys.stdout = old


def main():
    T, P = PKG / "tuning", REPO / "profiles"
    b = json.loads((P / "r05_bench.json").read_text())
    prof = json.loads((P / "r05_bench_py_profiled_run.json").read_text())
    rows = list(csv.DictReader(io.StringIO((P / "r05_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    avg_us = float(top["AverageNs"]) * 1e-3
    roof, v = b["roofline"], b["vs_hipblaslt_autotune_max"]
    out = []
    out.append("### 6.7 Round-5 figures (final tree = table after the four re-tune passes; `tools/lab/gpu_round5_g.sh` / `_h.sh`; every file named is committed)\n")
    out.append(f"**`bench.py`** (`profiles/r05_bench.json`, the closing run's box): `value` **{b['value']:.1f} TFLOP/s** (call E's box two hours earlier: 1468.1; round 4: 1412–1464 on three boxes — the "
               f"headline is flat: 4096³ is one 256² tile per CU at the board's power cap, and none of the round's schedule levers moved it by 1.5 %, §4.16).  `ms_per_step` {b['ms_per_step']:.2f} for 1024 GEMMs.  "
               f"`roofline.launch_us` {roof['launch_us']:.2f} µs (ONE clock since round 5: the mean of the dispatch-attached events; ADVICE r4) → `achieved` {roof['achieved']:.1f} TFLOP/s = **{roof['frac']:.3f}** of the "
               f"2.5 PFLOP/s dense fp16 peak; `wall_per_call_us` {roof['wall_per_call_us']:.2f} is reported beside it as the stream's throughput interval ({roof['throughput_tflops_wall']:.0f} TFLOP/s).  "
               f"rocprofv3 `--kernel-trace --stats` of the same command (`profiles/r05_bench_py_kernel_stats.csv` + the line that run printed, `r05_bench_py_profiled_run.json`): "
               f"{int(top['Calls'])} launches of `hgemm_tn_sq_kernel<CfgSQ<256,256,2,2,1,16>,1>`, **average {avg_us:.2f} µs** = {2 * 4096 ** 3 / avg_us * 1e-6:.0f} TFLOP/s = "
               f"**{2 * 4096 ** 3 / avg_us * 1e-6 / 2500:.3f}** of peak, {float(top['Percentage']):.2f} % of the device time.  `vs_hipblaslt_autotune_max` (same run, back to back, the strongest hipBLASLt variant): "
               f"{v['ours_tflops']:.1f} vs {v['hipblaslt_tflops']:.1f} TFLOP/s, ratio **{v['ratio']:.3f}**.  `traffic`: `profiles/r05_pmc_4096_4096_4096.json` "
               f"({json.loads((P / 'r05_pmc_4096_4096_4096.json').read_text())['dominant_kernel']['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch, 2.34× the algorithmic bytes = the floor of §4.14).  Per shape:\n")
    out.append("| shape | plan | ours µs (stream) | vs strongest hipBLASLt: device ratio | isolated ratio | wall ratio | hipGraph replay µs |\n|---|---|---|---|---|---|---|")
    for mnk, s in b["shapes"].items():
        pl = s["plan"]
        flags = "".join(f" + {k}" for k in ("nt_store", "xcd_stagger", "phase_offset", "phase_offset4") if pl.get(k))
        form = "" if pl["splits"] == 1 else f" ×{pl['splits']}" + (" fused" if pl["fused_split_k"] else "")
        out.append(f"| {mnk.replace('_', '×')} | `{pl['config']}`{form}{flags} | {s['ours_us']:.2f} | {s['speedup_vs_hipblaslt_auto_max']:.2f} | {s['speedup_isolated_vs_hipblaslt_max']:.2f} | "
                   f"{s['speedup_wall_vs_hipblaslt_best']:.2f} | {s.get('ours_graph_us', float('nan')):.1f} |")
    out.append("")
    # north star
    rep_path = T / "r05_grid_plan_report_autotune_mi355x.jsonl"
    rep = quiet_report(rep_path)
    iso, b2b = rep["vs_strongest_hipblaslt_isolated"], rep["vs_strongest_hipblaslt_back_to_back"]
    dec = lambda d: ", ".join(f"10{'⁰¹²³⁴⁵⁶⁷⁸⁹'[int(k) // 10] if int(k) >= 10 else ''}{'⁰¹²³⁴⁵⁶⁷⁸⁹'[int(k) % 10]} {v['geomean']:.3f} ({v['losers']} of {v['n']} lose)" for k, v in sorted(d.items(), key=lambda kv: int(kv[0])) if int(k) >= 9)
    out.append(f"**THE north-star figure — the whole grid against hipBLASLt-AUTOTUNE with a real budget, on the device clock** (`tuning/r05_grid_plan_report_autotune_mi355x.jsonl` + `.txt`, call H: "
               f"`hgemm_tune tune --plan-only --baselines --autotune --stream`, `HGEMM_AUTOTUNE_MAX_SECONDS=1` per layout = every candidate the heuristic returns, 50 warm-up + 100 timed shuffled rounds, "
               f"median — the reference's protocol, `cublas/fp32/hgemm_cublaslt_auto_tuning.cu:108-306`; per shape the STRONGEST of autotune / heuristic × tn / nn, the reference's `-max` rule; shipped plans, one box, "
               f"one run; {iso['shapes']} shapes):")
    out.append(f"* **isolated launches: geomean {iso['geomean_speedup']:.3f}** (arithmetic mean {iso['mean_speedup']:.3f}, the reference's README convention), {iso['shapes'] - iso['losers']} of {iso['shapes']} shapes faster, "
               f"{iso['losers_by_more_than_5pct']} lose by more than 5 %, {iso['losers_by_more_than_10pct']} by more than 10 %, minimum {iso['min_speedup']:.2f}; FLOP-weighted {iso['aggregate_tflops_ours']:.0f} vs "
               f"{iso['aggregate_tflops_hipblaslt_strongest']:.0f} TFLOP/s; by decade {dec(iso['by_log10_flops'])};")
    out.append(f"* **back to back: geomean {b2b['geomean_speedup']:.3f}** (mean {b2b['mean_speedup']:.3f}), {b2b['shapes'] - b2b['losers']} of {b2b['shapes']} faster, {b2b['losers_by_more_than_5pct']} lose by more than 5 %, "
               f"{b2b['losers_by_more_than_10pct']} by more than 10 %, minimum {b2b['min_speedup']:.2f}; FLOP-weighted {b2b['aggregate_tflops_ours']:.0f} vs {b2b['aggregate_tflops_hipblaslt_strongest']:.0f} TFLOP/s; "
               f"by decade {dec(b2b['by_log10_flops'])}.")
    hb2b = rep["back_to_back"]
    out.append(f"  Against hipBLASLt-heuristic alone (round 4's closing comparison: 1.103 isolated / 1.175 back to back): **{rep['geomean_speedup_vs_hipblaslt_heuristic_max']:.3f} / "
               f"{hb2b['geomean_speedup_vs_hipblaslt_heuristic_max']:.3f}**; round 4's quarter-grid figure against autotune was 1.122 isolated (10¹¹ 0.983, 10¹² 0.954).")
    # off-grid
    off = recs(T / "r05_offgrid_plan_report_mi355x.jsonl")
    i_ = lambda r: min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"]
    s_ = lambda r: min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"]
    worst = sorted(off, key=i_)[:3]
    out.append(f"\n**Off-grid** (`tuning/r05_offgrid_plan_report_mi355x.jsonl`, 80 never-tuned shapes at the planner's plans, all exact and within tolerance in the closing suite, `r05_offgrid_parity.jsonl` / `_randn.jsonl`; "
               f"against hipBLASLt-heuristic as in rounds 3–4): isolated geomean **{gm(map(i_, off)):.3f}** ({sum(i_(r) > 1 for r in off)} faster), back to back **{gm(map(s_, off)):.3f}** ({sum(s_(r) > 1 for r in off)} faster), "
               f"minimum **{min(map(i_, off)):.2f} / {min(map(s_, off)):.2f}** (round 4: 1.051 / 1.090, minimum 0.80 / 0.80).  Below 0.90: "
               + ", ".join(f"{r['mnk'].replace('_', '×')} {i_(r):.2f} (`{r['best']['config']}`)" for r in worst if i_(r) < 0.9) + ".")
    # pmc table
    tab = {r["mnk"]: r for r in json.loads((P / "r05_pmc_table.json").read_text())["rows"]}
    pick = ["16384_16384_16384", "12288_16384_8192", "4096_4096_4096", "512_4096_4096", "16384_128_16384", "64_4096_64"]
    out.append(f"\n**Per geometry** (`profiles/r05_pmc_table.json`, {len(tab)} shapes: the largest row of every geometry that serves ≥ 5 rows of the final table + the BASELINE shapes; three `--pmc` passes in the closing run): "
               + "; ".join(f"{m.replace('_', '×')} `{tab[m]['plan']['config']}` {tab[m]['tflops']:.0f} TFLOP/s, {tab[m]['roofline']['frac']:.3f} of its {tab[m]['roofline']['bound']} roof, traffic {tab[m]['traffic_ratio']:.2f}×" for m in pick if m in tab) + ".")
    print("\n".join(out))


if __name__ == "__main__":
    main()
