"""Per-geometry rocprofv3 evidence: MFMA-busy %, achieved HBM GB/s, traffic ratio and roofline fraction for one
representative grid shape of every kernel geometry that serves at least `--min-rows` rows of the tuned table.

  python tools/pmc_table.py shapes [--min-rows 5]          -> the shape list (one M_N_K per line, largest-flop row of
                                                             each geometry plus the BASELINE shapes)
  python tools/pmc_table.py table PASS_ROOT SHAPES.txt       -> JSON table from the passes tools/pmc_table.sh collected
  python tools/pmc_table.py baseline TABLE.json OUT_DIR      -> OUT_DIR/r06_pmc_<M_N_K>.json of the BASELINE.json shapes
                                                             (what bench.py's roofline.traffic reads)

Collection (tools/pmc_table.sh): three rocprofv3 --pmc passes (SQ + GRBM counters, FETCH_SIZE, WRITE_SIZE -- their TCC
slots do not fit one pass, MI355X_MICROARCH.md) of ONE process that benches every shape with its shipped plan, 3 warm-up
+ 6 timed launches each, one launch at a time.  The per-shape groups are recovered from the dispatch order: the operand
fill kernels of `hgemm_tune bench` separate the shapes.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE
are KiB, wide coalesced reads are counted at half on gfx950: hbm = (2 FETCH_SIZE + WRITE_SIZE) * 1024.  A two-pass
split-K plan is two kernels per launch: their counters and durations are added.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import re
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent.parent
BASELINE_SHAPES = ["64_4096_64", "512_4096_4096", "4096_4096_4096"]
MFMA_PEAK_TF, HBM_PEAK_TBS = 2500.0, 8.0
ROUND = "r06"   # prefix of the files `baseline` writes (profiles/<ROUND>_pmc_<M_N_K>.json)


def table_rows():
    rows = []
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))))
    return rows


def pick_shapes(min_rows: int):
    by_cfg = collections.defaultdict(list)
    for r in table_rows():
        by_cfg[r[3]].append(r)
    shapes = list(BASELINE_SHAPES)
    for cfg, rows in sorted(by_cfg.items()):
        if len(rows) < min_rows:
            continue
        rows.sort(key=lambda r: r[0] * r[1] * r[2])
        for r in (rows[-1], rows[len(rows) // 2]):          # the largest and the median row of the geometry
            s = f"{r[0]}_{r[1]}_{r[2]}"
            if s not in shapes:
                shapes.append(s)
    return shapes


def groups_of(csv_path: str):
    """dispatches of one pass in order -> list of per-shape lists of rows (fill / transpose kernels separate the shapes)"""
    rows = sorted(csv.DictReader(open(csv_path)), key=lambda r: int(r["Dispatch_Id"]))
    groups, cur, in_fill = [], [], True
    for r in rows:
        setup = any(t in r["Kernel_Name"] for t in ("fill_normal", "transpose_kernel", "fillBuffer"))
        if setup:
            if cur and not in_fill:
                groups.append(cur)
                cur = []
            in_fill = True
            continue
        in_fill = False
        cur.append(r)
    if cur:
        groups.append(cur)
    return groups


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("cmd", choices=["shapes", "table", "baseline"])
    ap.add_argument("root", nargs="?")
    ap.add_argument("shapes", nargs="?")
    ap.add_argument("--min-rows", type=int, default=5)
    a = ap.parse_args()
    if a.cmd == "shapes":
        print("\n".join(pick_shapes(a.min_rows)))
        return 0
    if a.cmd == "baseline":
        # baseline <table.json> <out dir>: one r06_pmc_<M_N_K>.json per BASELINE.json shape, the form bench.py's
        # measured_traffic() reads (dominant_kernel.{mnk, hbm_bytes_per_launch})
        tab = json.load(open(a.root))
        for row in tab["rows"]:
            if row["mnk"] not in ("64_4096_64", "512_4096_4096", "4096_4096_4096"):
                continue
            rec = {"source": tab["source"], "hbm_bytes": tab["hbm_bytes"], "table": f"profiles/{ROUND}_pmc_table.json",
                   "dominant_kernel": {"mnk": row["mnk"], "kernel": row["kernels"][0] if row["kernels"] else None,
                                       "hbm_bytes_per_launch": row["hbm_bytes_per_launch"],
                                       "algorithmic_bytes_per_launch": row["algorithmic_bytes_per_launch"],
                                       "avg_kernel_us_profiled": row["avg_kernel_us_profiled"]},
                   "row": row}
            with open(f"{a.shapes}/{ROUND}_pmc_{row['mnk']}.json", "w") as f:
                json.dump(rec, f, indent=1)
        return 0
    shapes = [s.strip() for s in open(a.shapes) if s.strip()]
    plans = {f"{r[0]}_{r[1]}_{r[2]}": r for r in table_rows()}
    per_shape = [dict(counters=collections.defaultdict(float), us=None, kernels=set()) for _ in shapes]
    for p, d in enumerate(sorted(glob.glob(f"{a.root}/pass*"))):
        cc = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
        if not cc:
            continue
        # counter rows: one row per (dispatch, counter); regroup by dispatch first
        by_disp = collections.OrderedDict()
        for r in sorted(csv.DictReader(open(cc[0])), key=lambda r: int(r["Dispatch_Id"])):
            by_disp.setdefault(int(r["Dispatch_Id"]), {"Kernel_Name": r["Kernel_Name"], "Dispatch_Id": r["Dispatch_Id"], "c": {}, "t": (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))})
            by_disp[int(r["Dispatch_Id"])]["c"][r["Counter_Name"]] = by_disp[int(r["Dispatch_Id"])]["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        groups, cur, in_fill = [], [], True
        for disp in by_disp.values():
            setup = any(t in disp["Kernel_Name"] for t in ("fill_normal", "transpose_kernel", "fillBuffer"))
            if setup:
                if cur and not in_fill:
                    groups.append(cur); cur = []
                in_fill = True
                continue
            in_fill = False
            cur.append(disp)
        if cur:
            groups.append(cur)
        if len(groups) != len(shapes):
            print(f"pass {d}: {len(groups)} dispatch groups for {len(shapes)} shapes", file=sys.stderr)
            return 1
        for i, g in enumerate(groups):
            names = []
            for disp in g:
                if disp["Kernel_Name"] not in names:
                    names.append(disp["Kernel_Name"])
            kpl = len(names)                                  # kernels per launch (2 for a two-pass split-K plan)
            launches = len(g) // kpl
            timed = g[3 * kpl:] if launches > 3 else g        # drop the three warm-up launches
            n = max(1, len(timed) // kpl)
            for disp in timed:
                for k, v in disp["c"].items():
                    per_shape[i]["counters"][k] += v / n
            if p == 0:
                per_shape[i]["us"] = sum((disp["t"][1] - disp["t"][0]) for disp in timed) / n * 1e-3
                per_shape[i]["kernels"] = [re.sub(r"hgemm_mi355x::", "", x)[:90] for x in names]
    out = []
    for s, rec in zip(shapes, per_shape):
        m, n, k = map(int, s.split("_"))
        c, us = rec["counters"], rec["us"] or 0.0
        flops, alg = 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n)
        hbm = (2 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None
        ai = flops / alg
        bound = "mfma" if ai >= MFMA_PEAK_TF / HBM_PEAK_TBS else "hbm"
        tf = flops / us * 1e-6 if us else None
        row = {"mnk": s, "plan": None, "kernels": rec["kernels"], "avg_kernel_us_profiled": round(us, 2),
               "tflops": round(tf, 1) if tf else None,
               "mfma_busy_pct": round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 1) if c.get("GRBM_GUI_ACTIVE") else None,
               "mfma_busy_pct_of_wave_cycles": round(100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_WAVE_CYCLES"]), 1) if c.get("SQ_WAVE_CYCLES") else None,
               "effective_clock_ghz": round(c["GRBM_GUI_ACTIVE"] / 8 / us * 1e-3, 3) if c.get("GRBM_GUI_ACTIVE") and us else None,
               "waves_parked_pct": round(100.0 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 1) if c.get("SQ_WAVE_CYCLES") else None,
               "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
               "traffic_ratio": round(hbm / alg, 2) if hbm else None,
               "achieved_hbm_gbs_algorithmic": round(alg / us * 1e-3, 1) if us else None,
               "achieved_hbm_gbs_measured": round(hbm / us * 1e-3, 1) if us and hbm else None,
               "roofline": {"bound": bound, "frac": round((tf / MFMA_PEAK_TF) if bound == "mfma" else (alg / us * 1e-6 / HBM_PEAK_TBS), 4) if us else None,
                            "arithmetic_intensity": round(ai, 1)}}
        if s in plans:
            r = plans[s]
            row["plan"] = {"config": r[3], "splits": r[4] & 0xFFFF, "fused": bool(r[4] & 0x10000), "nt_store": bool(r[4] & 0x20000), "group_m": r[5]}
        out.append(row)
    print(json.dumps({"source": "rocprofv3 --pmc, three passes of `hgemm_tune bench --shapes ... --lib --reps 6` (tools/pmc_table.sh), MI355X, N(0,1) operands, "
                                "isolated launches (profiled clocks run ~5 % above back-to-back clocks; short kernels include dispatch time in GRBM_GUI_ACTIVE)",
                      "hbm_bytes": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024, MI355X_MICROARCH.md (gfx950 counts wide coalesced reads at half)",
                      "rows": out}, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
