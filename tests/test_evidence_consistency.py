"""The figures DESIGN.md / README.md quote are recomputed here from the committed evidence files, so a number cannot drift away
from its file (CPU only: nothing here touches a GPU, the oracle or the reference)."""
import collections
import csv
import io
import json
import math
import re
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "cuda-l2_amd"
sys.path.insert(0, str(PKG / "tools"))


def _gm(xs):
    xs = list(xs)
    return math.exp(sum(map(math.log, xs)) / len(xs))


def _recs(path):
    return [json.loads(l) for l in open(path) if l.strip() and not l.startswith("#")]


def _design():
    return (REPO / "DESIGN.md").read_text()


def _shipped():
    out = {}
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            out[f"{m[1]}_{m[2]}_{m[3]}"] = (m[4], int(m[5]), int(m[6]))
    return out


R06_TABLE_UPDATES = ["r06_retune_fused_changes.jsonl", "r06_retune_wide_changes.jsonl", "r06_retune_flags_changes.jsonl"]   # in the order they were applied


def _shipped_r05():
    """The table as round 5 shipped it: today's table with round 6's recorded row changes undone, newest first -- round 5's reports
    and records describe THAT table.  (Every undo must find the row in the state the update left it in: the chain is complete.)"""
    out = _shipped()
    assert sorted(p.name for p in (PKG / "tuning").glob("r06_*_changes.jsonl")) == sorted(R06_TABLE_UPDATES)
    for name in reversed(R06_TABLE_UPDATES):
        for c in reversed(_recs(PKG / "tuning" / name)):
            assert out[c["mnk"]] == (c["to"]["config"], c["to"]["splits"], c["to"]["group_m"]), (name, c["mnk"])
            out[c["mnk"]] = (c["from"]["config"], c["from"]["splits"], c["from"]["group_m"])
    return out


def test_bench_record_and_its_rocprof_stats_match_the_design_text():
    b = json.loads((REPO / "profiles" / "r04_bench.json").read_text())
    assert b["metric"] == "HGEMM TFLOP/s" and b["n_gpus"] == 1 and b["dtype"] == "f16" and b["vs_baseline"] is None
    assert f"{b['value']:.1f}" in _design()
    roof = b["roofline"]
    assert roof["traffic_source"].startswith("profiles/r04_pmc_") and roof["launch_us"] == min(roof["avg_launch_us"], roof["wall_per_call_us"])
    assert f"{roof['frac']:.3f}" in _design()
    # the headline's own denominator is consistent with the per-shape report it comes from
    v = b["vs_hipblaslt_autotune_max"]
    w = b["shapes"]["4096_4096_4096"]
    assert abs(v["ratio"] - w["speedup_vs_hipblaslt_auto_max"]) < 1e-3 and abs(v["ours_tflops"] / v["hipblaslt_tflops"] - v["ratio"]) < 2e-3
    assert f"{v['ratio']:.3f}" in _design()
    # the profiled run of the same command: its own bench line and its kernel stats are committed together
    prof = json.loads((REPO / "profiles" / "r04_bench_py_profiled_run.json").read_text())
    assert prof["config"]["plan"] == b["config"]["plan"] and prof["steps"] == b["steps"] and prof["config"]["batch"] == b["config"]["batch"]
    rows = list(csv.DictReader(io.StringIO((REPO / "profiles" / "r04_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    assert "hgemm_tn_sq_kernel" in top["Name"] and "CfgSQ<256, 256" in top["Name"] and float(top["Percentage"]) > 99.9
    assert int(top["Calls"]) == (prof["steps"] + prof["warmup"]) * prof["config"]["batch"]
    avg_us = float(top["AverageNs"]) * 1e-3
    assert f"{avg_us:.2f}" in _design()
    # the three clocks agree: rocprofv3's kernel average is not above either live clock of the profiled run by more than 1 %
    assert 0.97 < avg_us / prof["roofline"]["launch_us"] < 1.01
    frac = 2.0 * 4096 ** 3 / avg_us * 1e-6 / 2500.0
    assert f"{frac:.3f}" in _design()


def test_grid_plan_reports_match_the_design_text():
    import tune_report

    rep = _recs(PKG / "tuning" / "r04_grid_plan_report_mi355x.jsonl")
    assert len(rep) == 1000 and all(r["stream_us"] > 0 for r in rep)
    g = _gm(min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"] for r in rep)
    gs = _gm(min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"] for r in rep)
    assert f"{g:.3f}" in _design() and f"{gs:.3f}" in _design()          # (round 4's closing figures: history in DESIGN section 6.6)
    out = tune_report.main(str(PKG / "tuning" / "r04_grid_plan_report_mi355x.jsonl"), 0)
    assert abs(out["back_to_back"]["geomean_speedup_vs_hipblaslt_heuristic_max"] - gs) < 1e-9
    for d in (10, 11, 12):
        assert f"{out['by_log10_flops'][d]['geomean']:.3f}" in _design(), d
    # (round 4's reports describe round 4's table: 303 rows moved on in round 5, whose own reports are checked below)
    auto = _recs(PKG / "tuning" / "r04_quarter_grid_plan_report_autotune_mi355x.jsonl")
    ga = _gm(min(v for v in (r["hipblaslt_auto_tn_us"], r["hipblaslt_auto_nn_us"], r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) if v > 0) / r["best"]["us"] for r in auto)
    assert len(auto) == 250 and f"{ga:.3f}" in _design()


def test_off_grid_report_and_parity_records():
    off = _recs(PKG / "tuning" / "r04_offgrid_plan_report_mi355x.jsonl")
    assert len(off) == 80
    g = _gm(min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"] for r in off)
    assert f"{g:.3f}" in _design()
    par = _recs(PKG / "tuning" / "r04_parity_1000.jsonl")
    assert len(par) == 2000 and all(r["pass"] for r in par)
    rn = _recs(PKG / "tuning" / "r04_randn_1000.jsonl")
    assert len(rn) == 2000 and all(r["pass"] for r in rn)
    assert f"{max(r['relative_error'] for r in rn):.1e}" in _design()
    for name in ("r04_offgrid_parity.jsonl", "r04_offgrid_randn.jsonl"):
        recs = _recs(PKG / "tuning" / name)
        assert len(recs) == 160 and all(r["pass"] for r in recs), name
    cand = sum(len(_recs(PKG / "tuning" / f)) for f in ("r04_candidate_parity_pass1.jsonl", "r04_candidate_parity_pass2.jsonl",
                                                         "r04_candidate_parity_family_r_flags.jsonl", "r04_candidate_parity_family_r_flags_pass2.jsonl",
                                                         "r04_candidate_parity_worst_rows.jsonl"))
    assert cand == 1990
    log = (REPO / "profiles" / "r04_check_final.log").read_text()
    runs = re.search(r"check: (\d+) runs, 0 failures", log)
    assert runs and f"{int(runs.group(1))} runs" in _design()


def test_pmc_table_feeds_bench_traffic_and_covers_every_geometry_with_five_rows():
    import collections

    tab = json.loads((REPO / "profiles" / "r04_pmc_table.json").read_text())
    rows = {r["mnk"]: r for r in tab["rows"]}
    for mnk in ("64_4096_64", "512_4096_4096", "4096_4096_4096"):
        d = json.loads((REPO / "profiles" / f"r04_pmc_{mnk}.json").read_text())["dominant_kernel"]
        assert d["mnk"] == mnk and abs(d["hbm_bytes_per_launch"] - rows[mnk]["hbm_bytes_per_launch"]) < 1
    r = rows["16384_128_16384"]
    assert f"{r['traffic_ratio']:.2f}" in _design()                      # the row section 6.6 builds its reading on


def test_sweep_readme_is_generated_from_the_records():
    import sweep_readme

    root = PKG / "eval_results" / "r04_sweep"
    text = sweep_readme.build(root, PKG / "tuning" / "r04_grid_plan_report_mi355x.jsonl", PKG / "tuning" / "r04_quarter_grid_plan_report_autotune_mi355x.jsonl")
    assert (root / "README.md").read_text() == text
    for acc, mode in (("fp32", "offline"), ("fp16", "offline"), ("fp32", "server"), ("fp16", "server")):
        d = json.loads((root / f"merge_{acc}_{mode}.json").read_text())
        assert d["shapes"] == 1000
        assert f"{d['geomean_speedup_vs_hipBLASLt-auto-tuning-max']:.3f}" in _design()
    hdr = (root / "cuda_l2_mi355x_F32F16F16F32_tflops_offline.csv").read_text().splitlines()[0]
    assert hdr.endswith("cuda_l2_pct_of_fp16_mfma_peak,cuda_l2_pct_of_roofline")


def test_late_candidate_generator_starts_every_line_with_the_shipped_plan(capsys):
    lib = PKG / "lib" / "libhgemm_mi355x.so"
    if not lib.exists():
        pytest.skip("library not built")
    import ctypes
    import make_round3_candidates

    assert make_round3_candidates.main() == 0
    out = capsys.readouterr().out.strip().splitlines()
    L = ctypes.CDLL(str(lib))
    L.hgemm_mi355x_config_by_name.argtypes = [ctypes.c_char_p]
    shipped = {}
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            shipped[f"{m[1]}_{m[2]}_{m[3]}"] = f"{m[4]}:{m[5]}:{m[6]}"
    assert len(out) > 300
    for ln in out:
        key, *cands = ln.split()
        assert cands[0] == shipped[key] and len(cands) <= 14 and len(set(cands)) == len(cands)
        for c in cands:
            name, s, g = c.rsplit(":", 2)
            assert L.hgemm_mi355x_config_by_name(name.encode()) >= 0 and int(s) & 0xFFFF >= 1 and int(g) >= 1


def test_k_tail_figures_match_the_records():
    """the before / after figures of the off-grid shapes with a K tail come from the two committed off-grid
    reports (closing run G = before the ktail variants of families q and r, call K = the final library), the kernel names and
    roofline fractions from the PMC table of call K, the run counts from the committed check logs."""
    d = _design()
    before = {r["mnk"]: r for r in _recs(PKG / "tuning" / "r04_offgrid_plan_report_before_ktail_mi355x.jsonl")}
    after = {r["mnk"]: r for r in _recs(PKG / "tuning" / "r04_offgrid_plan_report_call_k_mi355x.jsonl")}
    final = {r["mnk"]: r for r in _recs(PKG / "tuning" / "r04_offgrid_plan_report_mi355x.jsonl")}
    kt = [k for k in after if int(k.split("_")[2]) % 64 and int(k.split("_")[2]) % 8 == 0]
    assert len(kt) == 14 and set(before) == set(after)
    iso = lambda r: min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"]
    b2b = lambda r: min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"]
    for recs, f in ((before, iso), (after, iso), (before, b2b), (after, b2b)):
        assert f"{_gm(f(recs[k]) for k in kt):.3f}" in d
    assert f"{_gm(b2b(r) for r in after.values()):.3f}" in d and f"{_gm(b2b(r) for r in before.values()):.3f}" in d
    # the final report (call L: two more planner guards) has the same plans for the K-tail shapes, quotes its own geomeans and minima
    assert all((final[k]["best"]["config"], final[k]["best"]["splits"]) == (after[k]["best"]["config"], after[k]["best"]["splits"]) for k in kt)
    assert f"{_gm(iso(r) for r in final.values()):.3f}" in d and f"{_gm(b2b(r) for r in final.values()):.3f}" in d
    assert f"minimum **{min(iso(r) for r in final.values()):.2f} / {min(b2b(r) for r in final.values()):.2f}**" in d
    # every K-tail shape that left the classic family runs a ktail variant of family q or r now, and none of them got slower against hipBLASLt by more than noise
    moved = [k for k in kt if after[k]["best"]["config"] != before[k]["best"]["config"]]
    assert len(moved) == 11 and all(after[k]["best"]["config"][0] in "qr" and before[k]["best"]["config"][0] == "t" for k in moved)
    assert all(iso(after[k]) > iso(before[k]) - 0.02 for k in moved)
    for k, want in (("1332_3108_4440", "0.93"), ("12032_2048_7152", "0.91"), ("64_16384_9160", "0.95")):
        assert f"{iso(after[k]):.2f}" == want and want in d
    tab = json.loads((REPO / "profiles" / "r04_pmc_ktail_table.json").read_text())["rows"]
    assert len(tab) == 6
    for row in tab:   # the kernel that ran carries EPI_KTAIL (epilogue id 8..11) in its template arguments
        assert re.search(r"hgemm_tn_(sq|rs)_kernel<Cfg\w+<[\d, ]+>, (8|9|10|11)>", row["kernels"][0]), row["kernels"]
    r4000 = next(r for r in tab if r["mnk"] == "4000_4000_4000")
    assert f"{r4000['roofline']['frac']:.3f}" in d and f"{r4000['tflops']:.0f} TFLOP/s" in d
    seams = (REPO / "profiles" / "r04_check_q_item_seams.log").read_text()
    m = re.search(r"check: (\d+) runs, 0 failures", seams)
    assert m and f"{m.group(1)} runs" in d


def test_m32_counter_comparison_matches_its_record():
    """the 32x32x16 member of family q takes the same GPU cycles as the 16x16x32 one and a lower clock."""
    rec = json.loads((REPO / "profiles" / "r04_pmc_q256x256_m16_vs_m32_4096.json").read_text())
    m16, m32 = rec["rows"]
    assert (m16["config"], m32["config"]) == ("q256x256_w2x2", "q256x256_w2x2_m32") and "ELi32E" not in m16["kernel"]
    d = _design()
    assert f"{m32['gpu_cycles_per_xcd']:,}".replace(",", " ") in d and f"{m16['gpu_cycles_per_xcd']:,}".replace(",", " ") in d
    assert abs(rec["reading"]["cycles_ratio_m32_over_m16"] - 1.0) < 0.02 and rec["reading"]["clock_ratio"] < 0.92
    assert m16["counters"]["SQ_VALU_MFMA_BUSY_CYCLES"] == m32["counters"]["SQ_VALU_MFMA_BUSY_CYCLES"]
    for v in (m32["kernel_us"], m16["kernel_us"]):
        assert f"{v:.1f} µs" in d
    assert f"{m32['effective_clock_ghz']:.2f} GHz" in d and f"{m16['effective_clock_ghz']:.2f} GHz" in d
    assert f"{m32['waves_parked_pct']} %" in d and f"{m16['waves_parked_pct']} %" in d


def test_tuner_results_were_checked_before_they_were_timed():
    """Process rule of round 4 (tools/lab/README.md; VERDICT r3: a knob had been A/B-timed on 436 plans before it was ever run through
    `hgemm_tune check`, and computed wrong results): every round-4 tuner result file under cuda-l2_amd/tuning/ is listed in
    r04_checked_before_timed.json with the committed check logs that covered the geometries it times; the logs report 0 failures; every
    geometry that appears in the file is named by one of them (by the log's own `check-configs:` line when it has one); and in the lab
    script of the call the check comes before the tune."""
    man = json.loads((PKG / "tuning" / "r04_checked_before_timed.json").read_text())
    man.update(json.loads((PKG / "tuning" / "r05_checked_before_timed.json").read_text()))     # round 5: same rule, same test
    man.update(json.loads((PKG / "tuning" / "r06_checked_before_timed.json").read_text()))     # round 6
    tune_files = sorted(p for rnd in ("r04", "r05", "r06") for p in (PKG / "tuning").glob(f"{rnd}_*_mi355x.jsonl") if "plan_report" not in p.name)
    assert len([p for p in tune_files if p.name.startswith("r05_")]) >= 8 and len([p for p in tune_files if p.name.startswith("r06_")]) >= 3
    # round 6's single-launch forms at every batch depth of the last arriver's combine are in the check's own form list
    forms6 = re.search(r"^check-forms:(.*)$", (REPO / "profiles" / "r06_check_call_f_all_geometries_and_forms.log").read_text(), re.M).group(1)
    assert all(f"{n}|fused" in forms6 for n in (2, 3, 5, 7, 8, 13, 16, 21, 32, 37, 48))
    # round 5's family-q plan forms (kstagger variants, phase flags) are in the closing check's own form list
    forms = re.search(r"^check-forms:(.*)$", (REPO / "profiles" / "r05_check_final.log").read_text(), re.M).group(1)
    assert all(f in forms for f in ("1|xcd-stagger", "4|fused|xcd-stagger", "1|phase-offset", "1|phase-offset4", "1|phase-offset8", "1|wave-priority|nt-store"))
    for tf in tune_files:
        key = f"tuning/{tf.name}"
        assert key in man, f"{key}: no check evidence recorded -- a tuner result is not committed without it"
        covered = set()
        for entry in man[key]["logs"]:
            text = (REPO / entry["log"]).read_text()
            assert re.search(r"check: \d+ runs, 0 failures", text), entry["log"]
            own = re.search(r"^check-configs:(.*)$", text, re.M)
            covered |= set(own.group(1).split()) if own else set(entry["configs"])
        timed = set()
        for r in _recs(tf):
            timed |= {c["config"] for c in r["candidates"]}
        assert timed <= covered, f"{key}: timed but never checked: {sorted(timed - covered)[:5]}"
        script = (PKG / man[key].get("checked_in", man[key]["script"])).read_text()
        # (a script with several experiments names the two lines: the check that wrote the log, the tune that wrote the result)
        ci = script.index(man[key]["check_marker"]) if "check_marker" in man[key] else script.index(" check")
        ti = script.index(man[key]["tune_marker"]) if "tune_marker" in man[key] else script.index(" tune ")
        assert " check" in script and ci < ti, man[key]["script"]
    # plan-only reports time shipped plans (the table's own parity records cover those): nothing else may be in them
    for rep in [p for rnd in ("r04", "r05", "r06") for p in (PKG / "tuning").glob(f"{rnd}_*plan_report*_mi355x.jsonl")]:
        assert all(len(r["candidates"]) == 1 for r in _recs(rep)), rep.name


# ---- round 5 -------------------------------------------------------------------------------------------------------------------------
def test_round5_bench_record_and_its_rocprof_stats_match_the_design_text():
    b = json.loads((REPO / "profiles" / "r05_bench.json").read_text())
    assert b["metric"] == "HGEMM TFLOP/s" and b["n_gpus"] == 1 and b["dtype"] == "f16" and b["vs_baseline"] is None
    d = _design()
    roof = b["roofline"]
    # ONE clock for the roofline -- the dispatch-attached events; the wall clock per call is reported beside it, never substituted
    assert roof["launch_us"] == roof["avg_launch_us"] and "dispatch-attached" in roof["clock"] and roof["wall_per_call_us"] > 0
    assert abs(roof["achieved"] - 2.0 * 4096 ** 3 / roof["launch_us"] * 1e-6) < 1.0 and abs(roof["frac"] - roof["achieved"] / 2500.0) < 1e-3
    assert f"{b['value']:.1f}" in d and f"{roof['frac']:.3f}" in d and f"{b['vs_hipblaslt_autotune_max']['ratio']:.3f}" in d
    assert set(b["config"]["plan"]) >= {"config", "splits", "group_m", "nt_store", "xcd_stagger", "phase_offset", "phase_offset4", "wave_priority", "nt_loads"}
    prof = json.loads((REPO / "profiles" / "r05_bench_py_profiled_run.json").read_text())
    assert prof["config"]["plan"] == b["config"]["plan"] and prof["steps"] == b["steps"] and prof["config"]["batch"] == b["config"]["batch"]
    rows = list(csv.DictReader(io.StringIO((REPO / "profiles" / "r05_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    assert "hgemm_tn_sq_kernel" in top["Name"] and "CfgSQ<256, 256" in top["Name"] and float(top["Percentage"]) > 99.9
    assert int(top["Calls"]) == (prof["steps"] + prof["warmup"]) * prof["config"]["batch"]
    avg_us = float(top["AverageNs"]) * 1e-3
    assert f"{avg_us:.2f}" in d and f"{2.0 * 4096 ** 3 / avg_us * 1e-6 / 2500.0:.3f}" in d
    # the profiler's kernel average is below the event clock of its own run (events overstate: a start event fires while the predecessor drains)
    assert 0.93 < avg_us / prof["roofline"]["launch_us"] < 1.01
    # the traffic bench.py will cite from now on is this round's PMC record of the shipped kernel
    import bench

    t, src = bench.measured_traffic("4096_4096_4096")
    # (round 6's closing run took the counters again: bench.py now cites its record, the round-5 one stays on file)
    assert src == "profiles/r06_pmc_4096_4096_4096.json" and abs(t / (3 * 2 * 4096 ** 2) - 2.34) < 0.03
    r5 = json.loads((REPO / "profiles" / "r05_pmc_4096_4096_4096.json").read_text())["dominant_kernel"]
    assert abs(r5["hbm_bytes_per_launch"] / (3 * 2 * 4096 ** 2) - 2.34) < 0.03


def test_round5_north_star_report_matches_design_and_readme():
    """the metric BASELINE.json names -- geomean speedup over hipBLASLt-AUTOTUNE on the 1000-shape grid -- measured with a
    real autotune budget on the device clock, isolated and back to back, for the SHIPPED table; DESIGN / README lead with it."""
    import contextlib

    import tune_report

    path = PKG / "tuning" / "r05_grid_plan_report_autotune_mi355x.jsonl"
    rep = _recs(path)
    shipped = _shipped_r05()
    assert len(rep) >= 600 and len({r["mnk"] for r in rep}) == len(rep)
    for r in rep:
        assert (r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]], r["mnk"]      # the shipped plans, nothing else
        assert len(r["candidates"]) == 1 and r["stream_us"] > 0 and max(r["hipblaslt_auto_tn_us"], r["hipblaslt_auto_nn_us"]) > 0
    with contextlib.redirect_stdout(io.StringIO()):
        out = tune_report.main(str(path), 0)
    d, readme = _design(), (REPO / "README.md").read_text()
    iso, b2b = out["vs_strongest_hipblaslt_isolated"], out["vs_strongest_hipblaslt_back_to_back"]
    assert iso["shapes"] == len(rep) and b2b["shapes"] >= len(rep) - 5
    for blk in (iso, b2b):
        assert f"{blk['geomean_speedup']:.3f}" in d            # (README.md leads with round 6's figures since round 6; round 5's stay in DESIGN section 6.7)
        assert f"{blk['aggregate_tflops_ours']:.0f} vs {blk['aggregate_tflops_hipblaslt_strongest']:.0f} TFLOP/s" in d
        for dec in ("10", "11", "12"):
            if dec in {str(k) for k in blk["by_log10_flops"]}:
                v = blk["by_log10_flops"][int(dec)] if int(dec) in blk["by_log10_flops"] else blk["by_log10_flops"][dec]
                assert f"{v['geomean']:.3f}" in d, (dec, v)
    assert f"{out['geomean_speedup_vs_hipblaslt_heuristic_max']:.3f}" in d and f"{out['back_to_back']['geomean_speedup_vs_hipblaslt_heuristic_max']:.3f}" in d
    # independent recomputation of the headline (not through tune_report): strongest of autotune / heuristic x tn / nn per shape
    best = lambda r, suf: min(v for v in (r[f"hipblaslt_auto_tn{suf}"], r[f"hipblaslt_auto_nn{suf}"], r[f"hipblaslt_heur_tn{suf}"], r[f"hipblaslt_heur_nn{suf}"]) if v > 0)
    assert abs(_gm(best(r, "_us") / r["best"]["us"] for r in rep) - iso["geomean_speedup"]) < 1e-9
    # ... and the two runs that measured how much the report's own order moves the device-bound decades
    J = _recs(PKG / "tuning" / "r05_grid_1e10_up_plan_report_autotune_cooldown_mi355x.jsonl")
    K = _recs(PKG / "tuning" / "r05_compute_bound_sample_plan_report_autotune_long_boxes_mi355x.jsonl")
    H = {r["mnk"]: r for r in rep}
    fl = lambda r: 2.0 * math.prod(map(int, r["mnk"].split("_")))
    assert len(J) >= 300 and all(fl(r) >= 1e10 for r in J) and len(K) >= 30 and all(fl(r) >= 1e12 for r in K)
    for recs_, pairs in ((J, (("_us", lambda r: r["best"]["us"]), ("_stream_us", lambda r: r["stream_us"]))), (K, (("_stream_us", lambda r: r["stream_us"]),))):
        for suf, ours in pairs:
            assert f"{_gm(best(r, suf) / ours(r) for r in recs_):.3f}" in d                       # the run's own figure
            assert f"{_gm(best(H[r['mnk']], suf) / ours(H[r['mnk']]) for r in recs_):.3f}" in d   # call H on the same shapes
    big = [r for r in rep if fl(r) >= 1e11]
    assert f"{_gm(r['best']['us'] / r['stream_us'] for r in big):.3f}×" in d and f"{_gm(r['hipblaslt_heur_tn_us'] / r['hipblaslt_heur_tn_stream_us'] for r in big):.3f}×" in d
    for r in list(J) + list(K):
        assert (r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]], r["mnk"]


def test_round5_off_grid_traffic_and_parity_records():
    d = _design()
    off = _recs(PKG / "tuning" / "r05_offgrid_plan_report_mi355x.jsonl")
    assert len(off) == 80
    iso = lambda r: min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"]
    b2b = lambda r: min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"]
    assert f"{_gm(map(iso, off)):.3f}" in d and f"{_gm(map(b2b, off)):.3f}" in d
    assert f"minimum **{min(map(iso, off)):.2f} / {min(map(b2b, off)):.2f}**" in d
    par = _recs(PKG / "tuning" / "r05_parity_1000.jsonl")
    assert len(par) == 2000 and all(r["pass"] and r["bitwise_equal_unmasked"] for r in par)
    rn = _recs(PKG / "tuning" / "r05_randn_1000.jsonl")
    assert len(rn) == 2000 and all(r["pass"] for r in rn) and f"{max(r['relative_error'] for r in rn):.1e}" in d
    for name in ("r05_offgrid_parity.jsonl", "r05_offgrid_randn.jsonl"):
        recs = _recs(PKG / "tuning" / name)
        assert len(recs) == 160 and all(r["pass"] for r in recs), name
    cand = [len(_recs(PKG / "tuning" / f"r05_candidate_parity_pass{i}.jsonl")) for i in (1, 2, 3, 4)]
    assert " + ".join(map(str, cand)) + f" = {sum(cand)} candidate checks" in d
    changed = set()
    for i, n in zip((1, 2, 3, 4), (182, 163, 28, 21)):
        ch = _recs(PKG / "tuning" / f"r05_retune_pass{i}_changes.jsonl")
        assert len(ch) == n
        changed |= {c["mnk"] for c in ch}
    assert len(changed) == 303 and "303" in d
    log = (REPO / "profiles" / "r05_check_final.log").read_text()
    runs = re.search(r"check: (\d+) runs, 0 failures", log)
    assert runs and f"{int(runs.group(1)):,}".replace(",", " ") in d
    named = set(re.search(r"^check-configs:(.*)$", log, re.M).group(1).split())
    assert {c for c, _, _ in _shipped_r05().values()} <= named           # every geometry round 5 shipped is in its closing check
    # the phase-offset rows re-checked on a fresh box (section 4.14, call N): shipped plan against the same plan without the flag
    gains = []
    for r in _recs(PKG / "tuning" / "r05_phase_rows_recheck_mi355x.jsonl"):
        fig = lambda c: math.sqrt(c["us"] * c["stream_us"])
        ph = [c for c in r["candidates"] if c["splits"] & 0xA00000 and c.get("stream_us")]
        pl = [c for c in r["candidates"] if not c["splits"] & 0xA00000 and c.get("stream_us")]
        if ph and pl:
            assert (ph[0]["config"], ph[0]["splits"], ph[0]["group_m"]) == _shipped_r05()[r["mnk"]]
            gains.append(fig(pl[0]) / fig(ph[0]))
    assert len(gains) == 52 and f"+{(_gm(gains) - 1) * 100:.1f} % for the flag" in d
    assert f"{sum(g > 1.015 for g in gains)} gain more than 1.5 %, {sum(g < 0.985 for g in gains)} lose more than 1.5 %" in d
    # fabric traffic against K and against the raster group (section 4.14)
    tr = json.loads((REPO / "profiles" / "r05_pmc_traffic_vs_k.json").read_text())
    for mnk in ("4096_4096_4096", "8192_8192_8192", "16384_16384_16384"):
        o = tr["ours"][mnk]
        assert f"{o['traffic_over_algorithmic']:.2f}" in d and f"{o['floor_over_algorithmic']:.2f}" in d and f"{o['traffic_over_floor']:.3f}" in d
    assert f"{tr['hipblaslt']['16384_16384_16384']['traffic_over_algorithmic']:.2f}" in d
    rg = json.loads((REPO / "profiles" / "r05_pmc_16384_raster_groups.json").read_text())
    assert " / ".join(f"{r['traffic_over_algorithmic']:.2f}" for r in rg["rows"]) + "×" in d
    ab = json.loads((REPO / "profiles" / "r05_ab_round4_library_m0_clobber_lgkmcnt.json").read_text())["mean_us"]
    assert abs(ab["4096_4096_4096"]["ratio"] - 1.0) < 0.005 and abs(ab["16384_16384_256"]["ratio"] - 1.0) < 0.01
    # the per-geometry table covers every geometry that serves >= 5 rows of the FINAL table, and feeds bench.py's traffic
    import collections

    tab = json.loads((REPO / "profiles" / "r05_pmc_table.json").read_text())
    counts = collections.Counter(c for c, _, _ in _shipped_r05().values())
    assert {c for c, n in counts.items() if n >= 5} <= {r["plan"]["config"] for r in tab["rows"] if r["plan"]}
    rows = {r["mnk"]: r for r in tab["rows"]}
    for mnk in ("64_4096_64", "512_4096_4096", "4096_4096_4096"):
        dk = json.loads((REPO / "profiles" / f"r05_pmc_{mnk}.json").read_text())["dominant_kernel"]
        assert dk["mnk"] == mnk and abs(dk["hbm_bytes_per_launch"] - rows[mnk]["hbm_bytes_per_launch"]) < 1


# ---- round 6 -------------------------------------------------------------------------------------------------------------------------
def _r6_figures():
    sys.path.insert(0, str(PKG / "tools" / "lab"))
    import design_round6_figures

    return design_round6_figures.figures()


def test_round6_figures_in_design_and_readme_are_recomputed_from_the_records():
    """Every number of DESIGN.md's round-6 texts (lead paragraph, section 1.3, section 6.8) and of README.md's round-6 list is produced by
    tools/lab/design_round6_figures.py from the committed records; here the same module recomputes them and they must be in the texts."""
    f = _r6_figures()
    d, readme = _design(), (REPO / "README.md").read_text()
    in_design = ["BENCH_VALUE", "BENCH_FRAC", "PROF_AVG_US", "PROF_FRAC", "BENCH_RATIO", "GRID_ISO", "GRID_B2B", "GRID_ISO_MEAN", "GRID_B2B_MEAN", "REV_ISO_PCT", "REV_B2B_PCT",
                 "SW_FP32_OFF", "SW_FP16_OFF", "SW_FP32_SRV", "SW_FP16_SRV", "C4_ROW_10", "C4_ROW_100", "C4_ROW_1000", "OFF_ISO", "OFF_B2B", "CLS_COMPUTE", "CLS_SMALLK", "CLS_SKINNY",
                 "CLS_TINY", "CLS_MID", "RANDN_WORST", "C4_BENCH_US", "GRID_ISO_DEC", "GRID_B2B_DEC", "SW_DEC", "FAM_Q", "N_FUSED", "N_TWOPASS"]
    for k in in_design:
        assert f[k] in d, (k, f[k])
    for k in ("BENCH_VALUE", "PROF_FRAC", "GRID_ISO", "GRID_B2B", "SW_FP32_OFF", "SW_FP16_OFF", "SW_FP32_SRV", "SW_FP16_SRV", "C4_ROW_100", "CLS_COMPUTE", "CLS_TINY", "RANDN_WORST"):
        assert f[k] in readme, (k, f[k])


def test_round6_records_describe_the_shipped_table():
    shipped = _shipped()
    rep = _recs(PKG / "tuning" / "r06_grid_plan_report_autotune_interleaved_mi355x.jsonl")
    assert len(rep) == 1000 and len({r["mnk"] for r in rep}) == 1000
    for r in rep:
        assert (r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]], r["mnk"]      # the shipped plans, nothing else
        assert len(r["candidates"]) == 1 and r["stream_us"] > 0 and r["protocol"]["interleaved"] == 1 and r["protocol"]["reverse"] == 0
        assert r["protocol"]["autotune_from_cache"] == [1, 1]                                                       # nothing was searched inside the run
    rev = _recs(PKG / "tuning" / "r06_grid_1e11_up_plan_report_autotune_interleaved_reversed_mi355x.jsonl")
    assert len(rev) >= 150 and all(r["protocol"]["reverse"] == 1 and 2.0 * math.prod(map(int, r["mnk"].split("_"))) >= 1e11 for r in rev)
    for r in rev:
        assert (r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]], r["mnk"]
    # bench record + rocprofv3 stats of the same command
    b = json.loads((REPO / "profiles" / "r06_bench.json").read_text())
    prof = json.loads((REPO / "profiles" / "r06_bench_py_profiled_run.json").read_text())
    assert b["metric"] == "HGEMM TFLOP/s" and b["n_gpus"] == 1 and b["dtype"] == "f16" and b["vs_baseline"] is None and "cpu_baseline" in b
    assert prof["config"]["plan"] == b["config"]["plan"] and prof["steps"] == b["steps"] and prof["config"]["batch"] == b["config"]["batch"]
    plan = b["config"]["plan"]
    assert (plan["config"], plan["group_m"]) == (shipped["4096_4096_4096"][0], shipped["4096_4096_4096"][2])
    rows = list(csv.DictReader(io.StringIO((REPO / "profiles" / "r06_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    assert "hgemm_tn_sq_kernel" in top["Name"] and "CfgSQ<256, 256" in top["Name"] and float(top["Percentage"]) > 99.9
    assert int(top["Calls"]) == (prof["steps"] + prof["warmup"]) * prof["config"]["batch"]
    assert 0.93 < float(top["AverageNs"]) * 1e-3 / prof["roofline"]["launch_us"] < 1.01
    import bench

    t, src = bench.measured_traffic("4096_4096_4096")
    assert src == "profiles/r06_pmc_4096_4096_4096.json" and abs(t / (3 * 2 * 4096 ** 2) - 2.34) < 0.03
    # the closing check names every shipped geometry
    log = (REPO / "profiles" / "r06_check_final.log").read_text()
    assert re.search(r"check: \d+ runs, 0 failures", log)
    named = set(re.search(r"^check-configs:(.*)$", log, re.M).group(1).split())
    assert {c for c, _, _ in shipped.values()} <= named
    # three table updates, each stability-gated: the recorded row changes
    n = [len(_recs(PKG / "tuning" / name)) for name in R06_TABLE_UPDATES]
    assert n == [106, 88, 115] and len({c["mnk"] for name in R06_TABLE_UPDATES for c in _recs(PKG / "tuning" / name)}) == 262
    d = _design()
    assert "106 rows" in d and "88 rows" in d and "115 rows" in d and "262" in d


def test_round6_sweeps_use_the_cached_real_autotune_and_the_readme_is_generated():
    import sweep_readme_r06

    root = PKG / "eval_results" / "r06_sweep"
    assert (root / "README.md").read_text() == sweep_readme_r06.build(root)
    for acc in ("fp32", "fp16"):
        for mode in ("offline", "server"):
            recs = _recs(root / "records" / f"{acc}_{mode}_rank0.jsonl")
            assert len(recs) == 1000 and len({r["mnk"] for r in recs}) == 1000
            assert all(r["autotune_ok"] and r["autotune_budget_s"] == 1.0 for r in recs)                     # the real search: 1 s per layout
            assert all(r["autotune"]["tn"]["candidates"] >= 4 for r in recs)
            m = json.loads((root / f"merge_{acc}_{mode}.json").read_text())
            assert m["shapes"] == 1000
            col = [float(r["hipBLASLt-auto-tuning-max"]) for r in csv.DictReader(open(root / f"cuda_l2_mi355x_{'F32F16F16F32' if acc == 'fp32' else 'F16F16F16F16'}_speedup_{mode}.csv"))]
            assert abs(_gm(col) - m["geomean_speedup_vs_hipBLASLt-auto-tuning-max"]) < 2e-3
    # BASELINE config 4: three request rates, >= 1000 samples of cuda_l2 each, p50 <= p99
    for q in (10, 100, 1000):
        r = _recs(root / "config4" / f"qps_{q}.jsonl")[-1]
        lat = r["latency_ms"]["cuda_l2_mi355x_fp32"]
        assert r["mnk"] == "512_4096_4096" and r["target_qps"] == q and r["rounds"] * 7 >= 1000 and 0 < lat["p50"] <= lat["p99"]
    # the cache file: one record per (layout, shape) and hipBLASLt build, every grid shape present for both builds
    lines = [ln.split() for ln in (PKG / "tuning" / "r06_hipblaslt_autotune_cache.txt").read_text().splitlines() if ln and not ln.startswith("#")]
    keys = collections.Counter((ln[0], ln[1], ln[2], ln[3]) for ln in lines)
    assert len(keys) == 2000 and set(keys.values()) == {2} and all(float(ln[10]) == 1.0 for ln in lines)


def test_round6_energy_table_and_withdrawn_experiments_are_on_file():
    pw = json.loads((REPO / "profiles" / "r06_power_table.json").read_text())
    assert set(pw["summary"]) == {"4096_4096_4096", "8192_8192_8192", "16384_16384_16384"}
    for r in pw["rows"]:
        assert r["reads"] == 2 and r["socket_w"] > 300 and r["gfx_mhz"] > 800 and abs(r["joule_per_tflop"] - r["socket_w"] * r["us"] * 1e-6 / (2.0 * math.prod(map(int, r["mnk"].split("_"))) * 1e-12)) < 0.02
    assert (REPO / "profiles" / "withdrawn" / "r06_cuphase_first_look_mi355x.jsonl").exists()
    d = _design()
    assert "HGEMM_PLAN_CU_PHASE" in d and "fill-bound" in d


def test_round6_insitu_demo_matches_the_design_text():
    recs = _recs(REPO / "profiles" / "r06_insitu_selection_demo.jsonl")
    assert len(recs) == 64 and all(r["choice"] in r["candidates"] for r in recs)
    rep = [r for r in recs if not r["kept_the_table_plan"]]
    d = _design()
    assert f"keeps the table's plan on **{len(recs) - len(rep)}** and replaces **{len(rep)}**" in d
    assert f"**+{(_gm(r['table_plan_us'] / r['chosen_plan_us'] for r in rep) - 1) * 100:.1f} %** faster" in d
    assert not [r for r in rep if r["table_plan_us"] / r["chosen_plan_us"] < 0.97]


def test_round6_process_flow_validation_matches_the_design_text():
    root = PKG / "eval_results" / "r06_sweep"
    col = lambda sub: {r["mnk"]: float(r["hipBLASLt-auto-tuning-max"]) for r in csv.DictReader(open(root / sub / "cuda_l2_mi355x_F32F16F16F32_speedup_offline.csv"))}
    a, b = col("process_per_baseline"), col("inprocess_same_boxes")
    assert len(a) == 10 and set(a) == set(b) and {"64_4096_64", "512_4096_4096", "4096_4096_4096"} <= set(a)
    ga, gb = _gm(a.values()), _gm(b.values())
    assert f"{ga:.3f} (processes) / {gb:.3f} (in-process), ratio {ga / gb:.3f}" in _design()
    assert all(0.95 < a[m] / b[m] < 1.05 for m in a)


def test_round6_offgrid_insitu_demo_matches_the_design_text():
    recs = _recs(REPO / "profiles" / "r06_insitu_selection_demo_offgrid.jsonl")
    assert len(recs) == 80 and all(r["choice"] in r["candidates"] and 1 <= len(r["candidates"]) <= 3 for r in recs)
    rep = [r for r in recs if not r["kept_the_table_plan"]]
    d = _design()
    assert f"kept on **{len(recs) - len(rep)}**, replaced on **{len(rep)}**" in d
    assert f"**+{(_gm(r['table_plan_us'] / r['chosen_plan_us'] for r in rep) - 1) * 100:.1f} %** faster re-timed interleaved" in d


def test_round6_second_box_report_matches_the_design_text():
    sys.path.insert(0, str(PKG / "tools" / "lab"))
    import design_round6_figures as F

    rep = _recs(PKG / "tuning" / "r06_grid_plan_report_autotune_interleaved_second_box_mi355x.jsonl")
    shipped = _shipped()
    assert len(rep) == 1000 and all((r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]] and r["protocol"]["interleaved"] == 1 for r in rep)
    iso, b2b = F.ratios(rep)
    d = _design()
    assert f"back to back **{_gm(b2b):.3f}** ({sum(x > 1 for x in b2b)} faster" in d and f"isolated **{_gm(iso):.3f}** ({sum(x > 1 for x in iso)} faster)" in d
