"""Summarise a bin/hgemm_tune --baselines run: geomean speedups, per-size buckets, worst shapes."""
import collections
import json
import math
import sys


def gm(x):
    x = list(x)
    return math.exp(sum(map(math.log, x)) / len(x)) if x else float("nan")


def main(path, show=12):
    recs = [json.loads(l) for l in open(path) if l.strip()]
    rows = []
    for r in recs:
        m, n, k = map(int, r["mnk"].split("_"))
        fl = 2.0 * m * n * k
        ours = r["best"]["us"]
        lt = min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"])
        rb = min(r["rocblas_tn_us"], r["rocblas_nn_us"])
        rows.append({"mnk": r["mnk"], "flops": fl, "ours": ours, "lt": lt, "rb": rb, "sp_lt": lt / ours, "sp_rb": rb / ours,
                     "sp_lt_tn": r["hipblaslt_heur_tn_us"] / ours, "sp_lt_nn": r["hipblaslt_heur_nn_us"] / ours, "best": r["best"]})
    auto = [(min(v for v in (r.get("hipblaslt_auto_tn_us", -1), r.get("hipblaslt_auto_nn_us", -1)) if v > 0), r["best"]["us"], r)
            for r in recs if max(r.get("hipblaslt_auto_tn_us", -1), r.get("hipblaslt_auto_nn_us", -1)) > 0]
    out = {"shapes": len(rows),
           "geomean_speedup_vs_hipblaslt_heuristic_max": gm(x["sp_lt"] for x in rows),
           "geomean_speedup_vs_hipblaslt_heuristic_tn": gm(x["sp_lt_tn"] for x in rows),
           "geomean_speedup_vs_hipblaslt_heuristic_nn": gm(x["sp_lt_nn"] for x in rows),
           "geomean_speedup_vs_rocblas_max": gm(x["sp_rb"] for x in rows),
           "mean_speedup_vs_hipblaslt_heuristic_max": sum(x["sp_lt"] for x in rows) / len(rows),
           "fraction_faster_than_hipblaslt_heuristic_max": sum(x["sp_lt"] > 1 for x in rows) / len(rows),
           "aggregate_tflops_ours": sum(x["flops"] for x in rows) / sum(x["ours"] for x in rows) * 1e-6,
           "aggregate_tflops_hipblaslt_max": sum(x["flops"] for x in rows) / sum(x["lt"] for x in rows) * 1e-6}
    if auto:
        # the reference's "-max" rule: the better of autotune and heuristic per layout counts as the baseline
        amax = [min(a, min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"])) / o for a, o, r in auto]
        out["autotune_shapes"] = len(auto)
        out["geomean_speedup_vs_hipblaslt_autotune"] = gm(a / o for a, o, _ in auto)
        out["geomean_speedup_vs_hipblaslt_autotune_max"] = gm(amax)
        out["fraction_faster_than_hipblaslt_autotune_max"] = sum(x > 1 for x in amax) / len(amax)
    buckets = collections.defaultdict(list)
    for x in rows:
        buckets[int(math.log10(x["flops"]))].append(x["sp_lt"])
    out["by_log10_flops"] = {b: {"n": len(v), "geomean": round(gm(v), 3), "min": round(min(v), 2), "max": round(max(v), 2)} for b, v in sorted(buckets.items())}
    # back-to-back columns (hgemm_tune tune --plan-only --baselines --stream): the same comparison on the device clock of a
    # busy queue
    st = [(r, min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) / r["stream_us"]) for r in recs if r.get("stream_us", -1) > 0]
    if st:
        sb = collections.defaultdict(list)
        for r, sp in st:
            m, n, k = map(int, r["mnk"].split("_"))
            sb[int(math.log10(2.0 * m * n * k))].append(sp)
        fl = lambda r: 2.0 * math.prod(map(int, r["mnk"].split("_")))
        out["back_to_back"] = {"shapes": len(st), "geomean_speedup_vs_hipblaslt_heuristic_max": gm(sp for _, sp in st),
                               "fraction_faster": sum(sp > 1 for _, sp in st) / len(st),
                               "aggregate_tflops_ours": sum(fl(r) for r, _ in st) / sum(r["stream_us"] for r, _ in st) * 1e-6,
                               "aggregate_tflops_hipblaslt_max": sum(fl(r) for r, _ in st) /
                               sum(min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]) for r, _ in st) * 1e-6,
                               "by_log10_flops": {b: {"n": len(v), "geomean": round(gm(v), 3), "min": round(min(v), 2), "max": round(max(v), 2)}
                                                  for b, v in sorted(sb.items())}}
    # THE north-star comparison (BASELINE.json: geomean speedup over hipBLASLt-autotune): per shape the STRONGEST hipBLASLt variant
    # -- autotune or heuristic, tn or nn (the reference's "-max" rule: the baseline against which our speedup is lower) -- isolated
    # and back to back, with the FLOP-weighted aggregate, the losers and the per-decade split
    def strongest(r, suffix):
        vals = [r.get(f"hipblaslt_{kind}_{lay}{suffix}", -1) for kind in ("auto", "heur") for lay in ("tn", "nn")]
        return min(v for v in vals if v and v > 0)
    full = [r for r in recs if max(r.get("hipblaslt_auto_tn_us", -1), r.get("hipblaslt_auto_nn_us", -1)) > 0]
    if full:
        def summary(pairs):   # pairs: (record, ours_us, baseline_us)
            sp = [b / o for _, o, b in pairs]
            dec = collections.defaultdict(list)
            for (r, o, b) in pairs:
                m, n, k = map(int, r["mnk"].split("_"))
                dec[int(math.log10(2.0 * m * n * k))].append(b / o)
            fl = lambda r: 2.0 * math.prod(map(int, r["mnk"].split("_")))
            return {"shapes": len(pairs), "geomean_speedup": gm(sp), "mean_speedup": sum(sp) / len(sp), "fraction_faster": sum(x > 1 for x in sp) / len(sp),
                    "losers": sum(x < 1 for x in sp), "losers_by_more_than_5pct": sum(x < 0.95 for x in sp), "losers_by_more_than_10pct": sum(x < 0.90 for x in sp),
                    "min_speedup": min(sp), "aggregate_tflops_ours": sum(fl(r) for r, _, _ in pairs) / sum(o for _, o, _ in pairs) * 1e-6,
                    "aggregate_tflops_hipblaslt_strongest": sum(fl(r) for r, _, _ in pairs) / sum(b for _, _, b in pairs) * 1e-6,
                    "by_log10_flops": {d: {"n": len(v), "geomean": round(gm(v), 3), "min": round(min(v), 2), "losers": sum(x < 1 for x in v)} for d, v in sorted(dec.items())}}
        out["vs_strongest_hipblaslt_isolated"] = summary([(r, r["best"]["us"], strongest(r, "_us")) for r in full])
        fs = [r for r in full if r.get("stream_us", -1) > 0 and max(r.get("hipblaslt_auto_tn_stream_us", -1), r.get("hipblaslt_auto_nn_stream_us", -1)) > 0]
        if fs:
            out["vs_strongest_hipblaslt_back_to_back"] = summary([(r, r["stream_us"], strongest(r, "_stream_us")) for r in fs])
    print(json.dumps(out, indent=1))
    if st:
        for r, sp in sorted(st, key=lambda t: t[1])[:show]:
            print("  worst back to back %-20s ours %9.1f us (%s)  hipblaslt %9.1f us  speedup %.2f" % (
                r["mnk"], r["stream_us"], r["best"]["config"], min(r["hipblaslt_heur_tn_stream_us"], r["hipblaslt_heur_nn_stream_us"]), sp))
    for x in sorted(rows, key=lambda x: x["sp_lt"])[:show]:
        print("  worst %-20s ours %9.1f us (%s s=%d g=%d)  hipblaslt %9.1f us  speedup %.2f" % (
            x["mnk"], x["ours"], x["best"]["config"], x["best"]["splits"], x["best"]["group_m"], x["lt"], x["sp_lt"]))
    return out


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
