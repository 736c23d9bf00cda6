"""Every number of DESIGN.md's round-6 texts (section 1.3, section 6.8) computed from the committed records -- nothing in them is typed by hand.

    python cuda-l2_amd/tools/lab/design_round6_figures.py                       -> prints the placeholder table (JSON)
    python cuda-l2_amd/tools/lab/design_round6_figures.py TEMPLATE.md > OUT.md    -> TEMPLATE with its @NAME@ placeholders filled in

tests/test_evidence_consistency.py imports figures() and checks that the values appear in DESIGN.md / README.md."""
import csv
import io
import json
import math
import re
import sys
from collections import Counter, defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parents[3]
PKG = REPO / "cuda-l2_amd"
T, P, E = PKG / "tuning", REPO / "profiles", PKG / "eval_results" / "r06_sweep"
ACC = {"fp32": "F32F16F16F32", "fp16": "F16F16F16F16"}


def gm(xs):
    xs = list(xs)
    return math.exp(sum(map(math.log, xs)) / len(xs)) if xs else float("nan")


def recs(p):
    return [json.loads(ln) for ln in open(p) if ln.strip()]


def flops(mnk):
    m, n, k = map(int, mnk.split("_"))
    return 2.0 * m * n * k


def best(r, suf):
    """the strongest hipBLASLt variant (autotune / heuristic x tn / nn) on one clock"""
    return min(v for v in (r.get("hipblaslt_auto_tn" + suf, -1), r.get("hipblaslt_auto_nn" + suf, -1), r["hipblaslt_heur_tn" + suf], r["hipblaslt_heur_nn" + suf]) if v > 0)


def ratios(rep):
    return [best(r, "_us") / r["best"]["us"] for r in rep], [best(r, "_stream_us") / r["stream_us"] for r in rep]


def shape_class(mnk):
    m, n, k = map(int, mnk.split("_"))
    if m * n <= 2 ** 18 and k >= 2048:
        return "tiny"
    if min(m, n) <= 256 and max(m, n) >= 4096 and k >= 2048:
        return "skinny"
    if k <= 512 and m * n >= 4096 * 4096:
        return "smallk"
    if flops(mnk) >= 1e12:
        return "compute"
    if flops(mnk) < 1e11 and flops(mnk) / (2.0 * (m * k + k * n + m * n)) >= 312:
        return "mid"
    return "other"


def decade_text(rep, vals):
    dec = defaultdict(list)
    for r, v in zip(rep, vals):
        dec[int(math.log10(flops(r["mnk"])))].append(v)
    return ", ".join(f"10{_sup(d)} {gm(v):.3f} ({sum(x < 1 for x in v)} of {len(v)} lose)" for d, v in sorted(dec.items()) if d >= 9)


def _sup(d):
    return str(d).translate(str.maketrans("0123456789", "⁰¹²³⁴⁵⁶⁷⁸⁹"))


def figures() -> dict:
    f = {}
    # ---- bench ----------------------------------------------------------------------------------------------------------------
    b = json.loads((P / "r06_bench.json").read_text())
    roof = b["roofline"]
    rows = list(csv.DictReader(io.StringIO((P / "r06_bench_py_kernel_stats.csv").read_text())))
    top = max(rows, key=lambda r: float(r["Percentage"]))
    avg = float(top["AverageNs"]) * 1e-3
    f.update(BENCH_VALUE=f"{b['value']:.1f}", BENCH_MS=f"{b['ms_per_step']:.2f}", BENCH_LAUNCH_US=f"{roof['launch_us']:.2f}", BENCH_ACHIEVED=f"{roof['achieved']:.1f}",
             BENCH_FRAC=f"{roof['frac']:.3f}", BENCH_WALL_US=f"{roof['wall_per_call_us']:.2f}", PROF_CALLS=top["Calls"], PROF_AVG_US=f"{avg:.2f}",
             PROF_TFLOPS=f"{2.0 * 4096 ** 3 / avg * 1e-6:.0f}", PROF_FRAC=f"{2.0 * 4096 ** 3 / avg * 1e-6 / 2500.0:.3f}", PROF_PCT=f"{float(top['Percentage']):.2f}",
             BENCH_OURS_TF=f"{b['vs_hipblaslt_autotune_max']['ours_tflops']:.1f}", BENCH_LT_TF=f"{b['vs_hipblaslt_autotune_max']['hipblaslt_tflops']:.1f}",
             BENCH_RATIO=f"{b['vs_hipblaslt_autotune_max']['ratio']:.3f}", BENCH_TRAFFIC_MB=f"{roof['traffic'] * 1e-6:.1f}" if roof.get("traffic") else "n/a",
             BENCH_TRAFFIC_RATIO=f"{roof['traffic'] / roof['algorithmic_bytes_per_launch']:.2f}" if roof.get("traffic") else "n/a", BENCH_CPU=f"{b['cpu_baseline']['value']:.3f}")
    srows = []
    for mnk in ("64_4096_64", "512_4096_4096", "4096_4096_4096"):
        v = b["shapes"][mnk]
        pl = v["plan"]
        name = f"`{pl['config']}`" + (f" ×{pl['splits']}" if pl["splits"] > 1 else "") + (" single-launch" if pl.get("fused_split_k") else "") + \
            (" + xcd_stagger" if pl.get("xcd_stagger") else "") + (" + nt_store" if pl.get("nt_store") else "")
        srows.append(f"| {mnk.replace('_', '×')} | {name} | {v['ours_us']:.2f} | {v['speedup_vs_hipblaslt_auto_max']:.2f} | {v['speedup_isolated_vs_hipblaslt_max']:.2f} | "
                     f"{v['speedup_wall_vs_hipblaslt_best']:.2f} | {v['ours_graph_us']:.1f} |")
    f["BENCH_SHAPE_ROWS"] = "\n".join(srows)
    c4 = b["shapes"]["512_4096_4096"]
    f.update(C4_BENCH_US=f"{c4['ours_us']:.1f}", C4_BENCH_RATIO=f"{c4['speedup_vs_hipblaslt_auto_max']:.3f}")
    # ---- whole-grid plan report ---------------------------------------------------------------------------------------------------
    rep = recs(T / "r06_grid_plan_report_autotune_interleaved_mi355x.jsonl")
    iso, b2b = ratios(rep)
    f.update(GRID_CACHED=str(sum(1 for r in rep if r["protocol"]["autotune_from_cache"] == [1, 1])), GRID_ISO=f"{gm(iso):.3f}", GRID_ISO_MEAN=f"{sum(iso) / len(iso):.3f}",
             GRID_ISO_FASTER=str(sum(x > 1 for x in iso)), GRID_ISO_LOSE5=str(sum(x < 0.95 for x in iso)), GRID_ISO_MIN=f"{min(iso):.2f}", GRID_ISO_DEC=decade_text(rep, iso),
             GRID_B2B=f"{gm(b2b):.3f}", GRID_B2B_MEAN=f"{sum(b2b) / len(b2b):.3f}", GRID_B2B_FASTER=str(sum(x > 1 for x in b2b)), GRID_B2B_LOSE5=str(sum(x < 0.95 for x in b2b)),
             GRID_B2B_MIN=f"{min(b2b):.2f}", GRID_B2B_DEC=decade_text(rep, b2b))
    cls = defaultdict(list)
    for r, i, s in zip(rep, iso, b2b):
        cls[shape_class(r["mnk"])].append((i, s))
    for key, name in (("compute", "CLS_COMPUTE"), ("smallk", "CLS_SMALLK"), ("skinny", "CLS_SKINNY"), ("tiny", "CLS_TINY"), ("mid", "CLS_MID")):
        v = cls[key]
        f[name] = f"{gm(x[1] for x in v):.3f} / {gm(x[0] for x in v):.3f} ({sum(x[1] < 0.97 for x in v)})"
    f.update(CLASS_TINY_ISO=f"{gm(x[0] for x in cls['tiny']):.3f}", CLASS_TINY_B2B=f"{gm(x[1] for x in cls['tiny']):.3f}", CLASS_TINY_LOSE=str(sum(x[1] < 0.97 for x in cls["tiny"])),
             CLASS_SKINNY_ISO=f"{gm(x[0] for x in cls['skinny']):.3f}", CLASS_SKINNY_B2B=f"{gm(x[1] for x in cls['skinny']):.3f}", CLASS_SMALLK_B2B=f"{gm(x[1] for x in cls['smallk']):.3f}")
    # ---- order reversal -------------------------------------------------------------------------------------------------------------
    rev = recs(T / "r06_grid_1e11_up_plan_report_autotune_interleaved_reversed_mi355x.jsonl")
    fwd = {r["mnk"]: r for r in rep}
    common = [r for r in rev if r["mnk"] in fwd]
    ri, rb = ratios(common)
    fi, fb = ratios([fwd[r["mnk"]] for r in common])
    med = lambda xs: sorted(xs)[len(xs) // 2]
    f.update(REV_N=str(len(common)), REV_ISO_FWD=f"{gm(fi):.3f}", REV_ISO_REV=f"{gm(ri):.3f}", REV_ISO_PCT=f"{(gm(ri) / gm(fi) - 1) * 100:+.1f}",
             REV_B2B_FWD=f"{gm(fb):.3f}", REV_B2B_REV=f"{gm(rb):.3f}", REV_B2B_PCT=f"{(gm(rb) / gm(fb) - 1) * 100:+.1f}",
             REV_MED_ISO=f"{med([abs(a / c - 1) * 100 for a, c in zip(ri, fi)]):.1f}", REV_MED_B2B=f"{med([abs(a / c - 1) * 100 for a, c in zip(rb, fb)]):.1f}")
    # ---- sweeps ----------------------------------------------------------------------------------------------------------------
    merges = {(a, m): json.loads((E / f"merge_{a}_{m}.json").read_text()) for a in ("fp32", "fp16") for m in ("offline", "server") if (E / f"merge_{a}_{m}.json").exists()}
    for (a, m), d in merges.items():
        f[f"SW_{a.upper()}_{'OFF' if m == 'offline' else 'SRV'}"] = f"{d['geomean_speedup_vs_hipBLASLt-auto-tuning-max']:.3f}"
    order = [("fp32", "offline"), ("fp16", "offline"), ("fp32", "server"), ("fp16", "server")]
    f["SW_STRONG"] = " / ".join(f"{merges[k]['geomean_speedup_vs_hipBLASLt-strongest-of-autotune-and-heuristic']:.3f}" for k in order if k in merges)
    f["SW_MEANS"] = " / ".join(f"{merges[k]['mean_speedup_vs_hipBLASLt-auto-tuning-max']:.3f}" for k in order if k in merges)
    sp = {r["mnk"]: r for r in csv.DictReader(open(E / f"cuda_l2_mi355x_{ACC['fp32']}_speedup_offline.csv"))}
    dec = defaultdict(list)
    for mnk, r in sp.items():
        dec[int(math.log10(flops(mnk)))].append(min(float(r["hipBLASLt-auto-tuning-max"]), float(r["hipBLASLt-heuristic-max"])))
    f["SW_DEC"] = ", ".join(f"10{_sup(d)} {gm(v):.3f}" for d, v in sorted(dec.items()))
    cache_lines = [ln for ln in (T / "r06_hipblaslt_autotune_cache.txt").read_text().splitlines() if ln and not ln.startswith("#")]
    f["CACHE_RECORDS"] = str(len(cache_lines))
    for q in (10, 100, 1000):
        r = recs(E / "config4" / f"qps_{q}.jsonl")[-1]
        o = r["latency_ms"]["cuda_l2_mi355x_fp32"]
        f[f"C4_P50_{q}"], f[f"C4_P99_{q}"] = f"{o['p50']:.4f}", f"{o['p99']:.4f}"
        f[f"C4_ROW_{q}"] = f"{o['p50']:.4f} / {o['p99']:.4f}"
        f[f"C4_N_{q}"] = str(r["rounds"] * 7)
    r100 = recs(E / "config4" / "qps_100.jsonl")[-1]
    lt = r100["latency_ms"]["hipBLASLt-auto-tuning-tn"]
    f["C4_LT_100"] = f"{lt['p50']:.4f} / {lt['p99']:.4f}"
    f["C4_SPEEDUPS"] = " / ".join(f"{ {x['Baseline Method Name']: x for x in recs(E / 'config4' / f'qps_{q}.jsonl')[-1]['summary']}['hipBLASLt-auto-tuning-max']['Speedup']:.3f}" for q in (10, 100, 1000))
    # ---- table composition -----------------------------------------------------------------------------------------------------------
    fam, forms = Counter(), Counter()
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.search(r'\{(\d+), (\d+), (\d+), "([^"]+)", (\d+), (\d+)\}', ln)
        if not m:
            continue
        cfg, spl = m[4], int(m[5])
        fam[cfg[0]] += 1
        s = spl & 0xFFFF
        if not spl & 0x40000 and s > 1:
            forms["fused" if spl & 0x10000 else "twopass"] += 1
        forms["nt"] += bool(spl & 0x20000)
        forms["stag"] += bool(spl & 0x80000 and cfg[0] == "q")
        forms["phase"] += bool(spl & 0xA00000)
        forms["two"] += cfg in ("q128x128_w2x2", "q192x128_w2x2", "q128x192_w2x2")
    f.update(FAM_Q=str(fam["q"]), FAM_T=str(fam["t"]), FAM_W=str(fam["w"]), FAM_R=str(fam["r"]), FAM_S=str(fam["s"]), TWO_RES=str(forms["two"]), N_FUSED=str(forms["fused"]),
             N_TWOPASS=str(forms["twopass"]), N_NT=str(forms["nt"]), N_STAG=str(forms["stag"]), N_PHASE=str(forms["phase"]))
    # ---- parity / off-grid ---------------------------------------------------------------------------------------------------------
    f["RANDN_WORST"] = f"{max(r['relative_error'] for r in recs(T / 'r06_randn_1000.jsonl')):.2e}"
    f["OFF_RANDN_WORST"] = f"{max(r['relative_error'] for r in recs(T / 'r06_offgrid_randn.jsonl')):.2e}"
    off = recs(T / "r06_offgrid_plan_report_mi355x.jsonl")
    oi = [min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"]) / r["best"]["us"] for r in off]
    ob = [min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"] for r in off]
    below = sorted((s, r["mnk"], r["best"]["config"]) for r, s in zip(off, ob) if s < 0.90)
    f.update(OFF_ISO=f"{gm(oi):.3f}", OFF_B2B=f"{gm(ob):.3f}", OFF_ISO_FASTER=str(sum(x > 1 for x in oi)), OFF_B2B_FASTER=str(sum(x > 1 for x in ob)), OFF_MIN_ISO=f"{min(oi):.2f}",
             OFF_MIN_B2B=f"{min(ob):.2f}", OFF_BELOW=", ".join(f"{m.replace('_', '×')} {s:.2f} (`{c}`)" for s, m, c in below) or "none")
    # ---- energy --------------------------------------------------------------------------------------------------------------------
    pw = json.loads((P / "r06_power_table.json").read_text())
    parts = []
    for mnk, summ in pw["summary"].items():
        rows_ = [r for r in pw["rows"] if r["mnk"] == mnk]
        ship = next(r for r in rows_ if "shipped plan" in r["variant"])
        lt_ = min((r for r in rows_ if r["variant"].startswith("hipblaslt")), key=lambda r: r["us"])
        m32 = next((r for r in rows_ if "_m32" in r["variant"]), None)
        lo, hi = min(r["joule_per_tflop"] for r in rows_), max(r["joule_per_tflop"] for r in rows_)
        parts.append(f"{mnk.replace('_', '×')}: shipped {ship['us']:.1f} µs at {ship['socket_w']:.0f} W / {ship['gfx_mhz']:.0f} MHz = {ship['joule_per_tflop']:.3f} J/TFLOP; hipBLASLt "
                     f"({lt_['variant']}) {lt_['us']:.1f} µs, {lt_['socket_w']:.0f} W, {lt_['gfx_mhz']:.0f} MHz, {lt_['joule_per_tflop']:.3f}"
                     + (f"; `_m32` {m32['us']:.1f} µs, {m32['gfx_mhz']:.0f} MHz, {m32['joule_per_tflop']:.3f}" if m32 else "")
                     + f"; all {len(rows_)} variants {lo:.3f}–{hi:.3f}; lowest: {summ['lowest_joule_per_tflop']}")
    f["POWER_TEXT"] = ".  ".join(parts) + "."
    f["POWER_SUMMARY"] = "; ".join(f"{mnk.replace('_', '×')}: within 1 % of the lowest J/TFLOP: {', '.join(s['within_1pct_of_it'])}" for mnk, s in pw["summary"].items())
    # ---- per geometry ----------------------------------------------------------------------------------------------------------------
    tab = {r["mnk"]: r for r in json.loads((P / "r06_pmc_table.json").read_text())["rows"]}
    txt = []
    for mnk in ("16384_16384_16384", "4096_4096_4096", "512_4096_4096", "16384_128_16384", "64_4096_64"):
        if mnk in tab:
            r = tab[mnk]
            txt.append(f"{mnk.replace('_', '×')} `{r['plan']['config']}` {r['tflops']:.0f} TFLOP/s, {r['roofline']['frac']:.3f} of its {r['roofline']['bound']} roof, MFMA busy "
                       f"{r['mfma_busy_pct']:.1f} %, traffic {r['traffic_ratio']:.2f}×")
    f["PMC_TEXT"] = "; ".join(txt) + "."
    return f


def main():
    f = figures()
    if len(sys.argv) > 1:
        text = Path(sys.argv[1]).read_text()
        missing = sorted(set(re.findall(r"@([A-Z0-9_]+)@", text)) - set(f))
        if missing:
            raise SystemExit(f"no figure for: {missing}")
        sys.stdout.write(re.sub(r"@([A-Z0-9_]+)@", lambda m: f[m.group(1)], text))
    else:
        json.dump(f, sys.stdout, indent=1, ensure_ascii=False)
        print()


if __name__ == "__main__":
    main()
