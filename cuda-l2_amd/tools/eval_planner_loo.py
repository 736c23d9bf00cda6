"""Leave-one-out evaluation of the off-grid planner rules on the tuner's measured candidates.

    python tools/eval_planner_loo.py tuning/r02_grid_tune_runA_mi355x.jsonl tuning/r02_grid_tune_runB_mi355x.jsonl ...

Every grid shape is treated as if it were missing from the tuned table and planned (a) by the analytic model among
all of its measured candidate plans (what hgemm_api.hip model_plan does), (b) from the winning plans of its K nearest
grid shapes (log2 distance), ranked for this shape by the analytic model's estimate -- what neighbour_plan does with
the lattice corners around an off-grid shape.  Regret = measured time of the chosen plan / measured time of the
shape's best plan (geometric mean of the runs that measured it).  A neighbour's plan that was never measured on the
shape is skipped, as the library skips a plan whose geometry does not fit the shape.
"""
import json
import math
import sys


def gm(v):
    return math.exp(sum(map(math.log, v)) / len(v))


def load(paths):
    per = {}
    for p in paths:
        for line in open(p):
            r = json.loads(line)
            d = per.setdefault(r["mnk"], {})
            for c in r.get("candidates", []):
                k = (c["config"], c["splits"] & 0xFFFF, bool(c["splits"] & 0x10000))
                d.setdefault(k, []).append((c["us"], c.get("model_us")))
    return per


def main(paths, ks=(2, 4, 6)):
    per = load(paths)
    best = {}
    for s, d in per.items():
        t = {k: gm([u for u, _ in v]) for k, v in d.items()}
        b = min(t, key=t.get)
        best[s] = (b, t[b], t)
    shapes = list(per)
    key = {s: tuple(map(int, s.split("_"))) for s in shapes}

    def dist(a, b):
        return sum((math.log2(x) - math.log2(y)) ** 2 for x, y in zip(a, b))

    reg = []
    for s in shapes:
        _, tb, t = best[s]
        m = {k: v[0][1] for k, v in per[s].items() if v[0][1] is not None and not k[2]}
        if m:
            reg.append(t[min(m, key=m.get)] / tb)
    out = {"shapes": len(shapes), "model_all_geometries": {"regret_geomean": round(gm(reg), 4), "p90": round(sorted(reg)[int(0.9 * len(reg))], 3)}}
    for K in ks:
        reg = []
        for s in shapes:
            _, tb, t = best[s]
            nb = sorted((dist(key[s], key[o]), o) for o in shapes if o != s)[:K]
            cands = {best[o][0] for _, o in nb if best[o][0] in t}
            ranked = {k: per[s][k][0][1] for k in cands if per[s][k][0][1] is not None}
            if not ranked:
                continue
            reg.append(t[min(ranked, key=ranked.get)] / tb)
        out[f"neighbour_winners_k{K}"] = {"regret_geomean": round(gm(reg), 4), "p90": round(sorted(reg)[int(0.9 * len(reg))], 3), "planned": len(reg)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
