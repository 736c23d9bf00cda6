"""Print a compact per-shape table of a `hgemm_tune tune` jsonl: best plan, baselines, top candidates."""
import json
import sys

for line in open(sys.argv[1]):
    r = json.loads(line)
    m, n, k = map(int, r["mnk"].split("_"))
    fl = 2.0 * m * n * k * 1e-6
    base = min(v for kk, v in r.items() if kk.endswith("_us") and isinstance(v, float) and v > 0) if any(kk.endswith("_us") for kk in r) else None
    s = f'{r["mnk"]:>20} best {r["best"]["config"]:>20} x{r["best"]["splits"]} g{r["best"]["group_m"]:<2} {fl / r["best"]["us"]:7.1f} TF'
    if base:
        s += f'  base {fl / base:7.1f} TF  speedup {base / r["best"]["us"]:.3f}'
    print(s)
    seen = set()
    for c in r["candidates"]:
        if c["config"] in seen:
            continue
        seen.add(c["config"])
        if len(seen) > int(sys.argv[2]) if len(sys.argv) > 2 else 6:
            break
        print(f'{"":>24} {c["config"]:>20} x{c["splits"]} g{c["group_m"]:<2} {fl / c["us"]:7.1f} TF')
