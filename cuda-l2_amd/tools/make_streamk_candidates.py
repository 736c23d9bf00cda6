"""Candidate file for a stream-K tune (bin/hgemm_tune tune --cand-file): per shape the shipped plan and the stream-K plans
(HGEMM_PLAN_STREAMK | workgroups) of the geometries that have the kernel, ranked by the analytic model.

    python tools/make_streamk_candidates.py --shapes M_N_K,... | --shape-file F | --from-report report.jsonl [--below 1.08]
                                            [--top 6] > candidates.txt

--from-report: the shapes of a plan report (hgemm_tune tune --plan-only --baselines) with >= 1e9 flop whose plan is a split-K
plan or whose speedup over hipBLASLt is below --below (where a better chip fill or a cheaper combine can matter).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(PKG_DIR))
STREAMK = 0x40000


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--shape-file", default="")
    ap.add_argument("--from-report", default="")
    ap.add_argument("--below", type=float, default=1.08)
    ap.add_argument("--top", type=int, default=6)
    a = ap.parse_args(argv)
    import build

    L = ctypes.CDLL(str(build.build_library()))
    L.hgemm_mi355x_config_name.restype = ctypes.c_char_p
    L.hgemm_mi355x_model_us.restype = ctypes.c_double
    L.hgemm_mi355x_model_us.argtypes = [ctypes.c_int] * 5
    shapes = [s for s in a.shapes.split(",") if s]
    if a.shape_file:
        shapes += [ln.strip() for ln in Path(a.shape_file).read_text().splitlines() if ln.strip() and not ln.startswith("#")]
    if a.from_report:
        for ln in Path(a.from_report).read_text().splitlines():
            r = json.loads(ln)
            m, n, k = map(int, r["mnk"].split("_"))
            lt = min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"])
            if 2.0 * m * n * k >= 1e9 and ((r["best"]["splits"] & 0xFFFF) > 1 or lt / r["best"]["us"] < a.below):
                shapes.append(r["mnk"])
    info = (ctypes.c_int * 8)()
    for mnk in dict.fromkeys(shapes):
        m, n, k = map(int, mnk.split("_"))
        cfg, sp, gm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(gm))
        toks = []
        if cfg.value >= 0:
            toks.append(f"{L.hgemm_mi355x_config_name(cfg.value).decode()}:{sp.value}:{gm.value}")
        sk = []
        for c in range(L.hgemm_mi355x_num_configs()):
            per_cu = L.hgemm_mi355x_config_streamk(c)
            if per_cu <= 0 or not L.hgemm_mi355x_streamk_runs(c, m, n, k):   # (direct K tail, > 65536 tiles: the launch would run data-parallel)
                continue
            L.hgemm_mi355x_config_info(c, info)
            if (info[0] > 2 * m and info[0] > 32) or (info[1] > 2 * n and info[1] > 32):
                continue
            for r in range(1, per_cu + 1):
                plan = STREAMK | (256 * r)
                sk.append((L.hgemm_mi355x_model_us(c, plan, m, n, k), f"{L.hgemm_mi355x_config_name(c).decode()}:{plan}:{L.hgemm_mi355x_default_group(c, m, n)}"))
        toks += [t for _, t in sorted(sk)[:a.top]]
        print(mnk, *toks)
    return 0


if __name__ == "__main__":
    sys.exit(main())
