"""Energy per flop of the compute-bound variants from `hgemm_tune bench --power` records.

    python tools/power_table.py gpurun_out/r6p/power.jsonl > ../profiles/r06_power_table.json

Per (shape, variant): time per call, sustained TFLOP/s, mean socket power and gfx clock over the timed run (rocm_smi gpu-metrics,
sampled by the tool), and from them  J per TFLOP = W x us / (2MNK x 1e-6)  and  cycles per call = us x MHz.  Records of the same
(shape, variant) -- the script reads every variant twice, with different predecessors -- are kept side by side and averaged.
"""
from __future__ import annotations

import json
import sys
from collections import OrderedDict


def main(argv=None) -> int:
    rows = OrderedDict()
    for ln in open((argv or sys.argv[1:])[0]):
        ln = ln.strip()
        if not ln.startswith("{"):
            continue
        r = json.loads(ln)
        t = r.get("telemetry") or {}
        if not t.get("samples"):
            continue
        m, n, k = (int(x) for x in r["mnk"].split("_"))
        flops = 2.0 * m * n * k
        key = (r["mnk"], r["what"] + ("" if "splits" not in r else ""))
        rows.setdefault(key, []).append({"us": r["us"], "tflops": r["tflops"], "socket_w": t["socket_w_mean"], "gfx_mhz": t["gfx_mhz_mean"],
                                         "gfx_mhz_min": t["gfx_mhz_min"], "joule_per_tflop": t["socket_w_mean"] * r["us"] * 1e-6 / (flops * 1e-12),
                                         "mcycles_per_call": r["us"] * t["gfx_mhz_mean"] * 1e-6})
    out = []
    for (mnk, what), recs in rows.items():
        mean = {k: sum(x[k] for x in recs) / len(recs) for k in recs[0]}
        out.append({"mnk": mnk, "variant": what, "reads": len(recs), **{k: round(v, 4) for k, v in mean.items()},
                    "spread_us_pct": round(100.0 * (max(x["us"] for x in recs) - min(x["us"] for x in recs)) / mean["us"], 2), "records": recs})
    by_shape = {}
    for r in out:
        by_shape.setdefault(r["mnk"], []).append(r)
    summary = {}
    for mnk, rs in by_shape.items():
        best = min(rs, key=lambda r: r["joule_per_tflop"])
        summary[mnk] = {"lowest_joule_per_tflop": best["variant"], "within_1pct_of_it": [r["variant"] for r in rs if r["joule_per_tflop"] <= best["joule_per_tflop"] * 1.01],
                        "fastest": min(rs, key=lambda r: r["us"])["variant"]}
    json.dump({"source": "hgemm_tune bench --power --seconds 2 (back-to-back launches, rocm_smi gpu-metrics sampled over the timed run); tools/lab/gpu_round6_power.sh",
               "units": {"joule_per_tflop": "socket W x s per 1e12 flop", "mcycles_per_call": "us x mean gfx MHz / 1e6"}, "summary": summary, "rows": out}, sys.stdout, indent=1)
    print()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
