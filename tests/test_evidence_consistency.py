"""The figures DESIGN.md / README.md quote are recomputed here from the committed evidence files, so a number cannot drift away
from its file (CPU only: nothing here touches a GPU, the oracle or the reference)."""
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


def test_bench_record_and_its_rocprof_stats_match_the_design_text():
    b = json.loads((REPO / "profiles" / "r03_bench.json").read_text())
    assert b["metric"] == "HGEMM TFLOP/s" and b["n_gpus"] == 1 and b["dtype"] == "f16" and b["vs_baseline"] is None
    assert f"{b['value']:.1f}" in _design()                                   # 1443.8
    assert b["roofline"]["traffic_source"] == "profiles/r03_pmc_4096_4096_4096.json" or b["roofline"]["traffic_source"].startswith("profiles/")
    rows = list(csv.DictReader(io.StringIO((REPO / "profiles" / "r03_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    assert "hgemm_tn_sq_kernel" in top["Name"] and "CfgSQ<256, 256" in top["Name"] and float(top["Percentage"]) > 99.9
    avg_us = float(top["AverageNs"]) * 1e-3
    assert f"{avg_us:.2f}" in _design()                                       # 94.84
    # rocprofv3's average and bench.py's own dispatch-attached events agree (the events include the predecessor's drain)
    assert 0.97 < avg_us / b["roofline"]["avg_launch_us"] <= 1.0
    frac = 2.0 * 4096 ** 3 / avg_us * 1e-6 / 2500.0
    assert f"{frac:.3f}" in _design()                                         # 0.580


def test_grid_plan_reports_match_the_design_text():
    import tune_report

    iso = _recs(PKG / "tuning" / "r03_grid_plan_report_mi355x.jsonl")
    assert len(iso) == 1000
    g = _gm(min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"] for r in iso)
    assert f"{g:.3f}" in _design()                                            # 1.089
    st = _recs(PKG / "tuning" / "r03_grid_plan_report_stream_mi355x.jsonl")
    assert len(st) == 1000 and all(r["stream_us"] > 0 for r in st)
    gs = _gm(min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"] for r in st)
    assert f"{gs:.3f}" in _design() and f"{gs:.3f}" in (REPO / "README.md").read_text()   # 1.145
    out = tune_report.main(str(PKG / "tuning" / "r03_grid_plan_report_stream_mi355x.jsonl"), 0)
    b2b = out["back_to_back"]
    assert abs(b2b["geomean_speedup_vs_hipblaslt_heuristic_max"] - gs) < 1e-9
    assert f"{b2b['by_log10_flops'][11]['geomean']:.3f}" in _design() and f"{b2b['by_log10_flops'][12]['geomean']:.3f}" in _design()
    # every reported plan is the shipped plan of that shape
    shipped = {}
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            shipped[f"{m[1]}_{m[2]}_{m[3]}"] = (m[4], int(m[5]), int(m[6]))
    for r in iso + st:
        assert (r["best"]["config"], r["best"]["splits"], r["best"]["group_m"]) == shipped[r["mnk"]], r["mnk"]


def test_off_grid_report_and_parity_records():
    off = _recs(PKG / "tuning" / "r03_offgrid_plan_report_mi355x.jsonl")
    assert len(off) == 80
    g = _gm(min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"] for r in off)
    assert 1.095 < g < 1.105 and "geomean **1.10**" in _design()
    par = _recs(PKG / "tuning" / "r03_parity_1000.jsonl")
    assert len(par) == 2000 and all(r["pass"] for r in par)
    cand = _recs(PKG / "tuning" / "r03_candidate_parity.jsonl")
    assert all(r["pass"] for r in cand) and f"{len(cand)} candidate checks" in (REPO / "README.md").read_text()


def test_pmc_table_feeds_bench_traffic_and_covers_every_geometry_with_five_rows():
    import collections

    tab = json.loads((REPO / "profiles" / "r03_pmc_table.json").read_text())
    rows = {r["mnk"]: r for r in tab["rows"]}
    for mnk in ("64_4096_64", "512_4096_4096", "4096_4096_4096"):
        d = json.loads((REPO / "profiles" / f"r03_pmc_{mnk}.json").read_text())["dominant_kernel"]
        assert d["mnk"] == mnk and abs(d["hbm_bytes_per_launch"] - rows[mnk]["hbm_bytes_per_launch"]) < 1
    counts = collections.Counter()
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{\d+, \d+, \d+, "(\w+)", \d+, \d+\}', ln)
        if m:
            counts[m[1]] += 1
    covered = {r["plan"]["config"] for r in tab["rows"] if r["plan"]}
    assert {c for c, n in counts.items() if n >= 5} <= covered


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
