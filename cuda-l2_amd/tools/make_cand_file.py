"""Candidate plans for a targeted re-tune, from earlier tuning runs -> the file `hgemm_tune tune --cand-file` reads.

    python tools/make_cand_file.py tuning/r02_grid_tune_runA_mi355x.jsonl tuning/r02_grid_tune_runB_mi355x.jsonl ... > cand.txt

Per shape: the `--top` fastest distinct plans over the given runs (geometric mean of a plan's times over the runs that
measured it), plus -- when a persistent family ("s" / "q") is among the shape's ten fastest -- the family-q member of
every persistent geometry seen there (round 3 changed family q's K loop and the persistent epilogue, so q must be
re-measured against whatever won before), at the raster group of the fastest persistent plan.  One line per shape:
"M_N_K config:splits:group ...".
"""
import argparse
import json
import math
import sys


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("runs", nargs="+")
    ap.add_argument("--top", type=int, default=3)
    ap.add_argument("--max", type=int, default=6)
    a = ap.parse_args()
    per = {}
    for path in a.runs:
        for ln in open(path):
            if not ln.strip():
                continue
            r = json.loads(ln)
            plans = per.setdefault(r["mnk"], {})
            seen = set()
            for c in r.get("candidates", []):
                k = (c["config"], int(c["splits"]), int(c["group_m"]))
                if k in seen:
                    continue
                seen.add(k)
                plans.setdefault(k, []).append(c["us"])
    for mnk, plans in sorted(per.items(), key=lambda kv: tuple(map(int, kv[0].split("_")))):
        ranked = sorted(plans, key=lambda k: math.exp(sum(map(math.log, plans[k])) / len(plans[k])))
        out = ranked[:a.top]
        persistent = [k for k in ranked[:10] if k[0][0] in "sq"]
        if persistent:
            group = persistent[0][2]
            for k in persistent:
                q = ("q" + k[0][1:], k[1], k[2])
                for cand in (q, (q[0], q[1], group)):
                    if cand not in out and len(out) < a.max:
                        out.append(cand)
        print(mnk, " ".join(f"{c}:{s}:{g}" for c, s, g in out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
