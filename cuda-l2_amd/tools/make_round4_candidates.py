"""Candidate plans of the round-4 re-tune (bin/hgemm_tune tune --cand-file), one line per shape:

    M_N_K  config:splits:group  config:splits:group ...

Per grid shape: the SHIPPED plan (re-measured in the same run, same box) and, where they apply,
  * stream-K plans (HGEMM_PLAN_STREAMK | workgroups) of the geometries that have the kernel, the best few by the model --
    for shapes with >= 1e9 flop whose shipped plan is a split-K plan or sits below `--below` x hipBLASLt in the given report;
  * the double-buffered members of family r ("_d"): plain, split-K (both forms) and stream-K -- skinny shapes (min(M, N) <= 256,
    K >= 2048);
  * family w (wave-direct, no LDS): every member that fits, for K <= 128; with split-K so that ~128-512 workgroups exist, for
    tiny outputs (M * N <= 256 * 256) with K >= 512.

    python tools/make_round4_candidates.py --report tuning/r03_grid_plan_report_mi355x.jsonl > tuning/r04_candidates.txt
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(PKG_DIR))
STREAMK, FUSED = 0x40000, 0x10000


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--report", required=True, help="plan report (hgemm_tune tune --plan-only --baselines) of the shipped table")
    ap.add_argument("--below", type=float, default=1.08)
    ap.add_argument("--top-sk", type=int, default=6)
    ap.add_argument("--shapes-out", default="", help="also write the selected shapes, one per line")
    ap.add_argument("--pass2", action="store_true", help="second pass: family w beyond the first pass's domain (K <= 512 on outputs up to 2048^2 for "
                    "the one-wave-per-tile members, K >= 256 on outputs up to 1024^2 for the _k4 members), against the table as it stands")
    a = ap.parse_args(argv)
    import build

    L = ctypes.CDLL(str(build.build_library()))
    L.hgemm_mi355x_config_name.restype = ctypes.c_char_p
    L.hgemm_mi355x_model_us.restype = ctypes.c_double
    L.hgemm_mi355x_model_us.argtypes = [ctypes.c_int] * 5
    ncfg = L.hgemm_mi355x_num_configs()
    info = (ctypes.c_int * 8)()
    cfgs = []
    for c in range(ncfg):
        L.hgemm_mi355x_config_info(c, info)
        cfgs.append({"id": c, "name": L.hgemm_mi355x_config_name(c).decode(), "bm": info[0], "bn": info[1], "kgran": L.hgemm_mi355x_config_k_granularity(c),
                     "sk": L.hgemm_mi355x_config_streamk(c)})

    def fits(cf, m, n, k):
        return not ((cf["bm"] > 2 * m and cf["bm"] > 32) or (cf["bn"] > 2 * n and cf["bn"] > 32) or k % cf["kgran"])

    def grp(cf, m, n):
        return L.hgemm_mi355x_default_group(cf["id"], m, n)

    selected = []
    for ln in Path(a.report).read_text().splitlines():
        r = json.loads(ln)
        m, n, k = map(int, r["mnk"].split("_"))
        flops = 2.0 * m * n * k
        lt = min(r["hipblaslt_heur_tn_us"], r["hipblaslt_heur_nn_us"])
        toks = []
        if a.pass2:
            first = (k <= 128) or (m * n <= 256 * 256 and k >= 512)          # the first pass's family-w domain
            for cf in cfgs:
                if cf["name"][0] != "w" or not fits(cf, m, n, k) or first:
                    continue
                k4 = cf["name"].endswith("_k4")
                tiles = -(-m // cf["bm"]) * -(-n // cf["bn"])
                if not k4 and k <= 512 and m * n <= 2048 * 2048 and tiles <= 4096:
                    toks.append(f"{cf['name']}:1:{grp(cf, m, n)}")
                if k4 and k >= 256 and m * n <= 1024 * 1024:
                    for s in (1, 2, 4, 8, 16):
                        per = k // s
                        if per < 256 or per % 64 or tiles * s > 2048 or (tiles * s < 64 and s < 16):
                            continue
                        toks.append(f"{cf['name']}:{s | (FUSED if s > 1 else 0)}:{grp(cf, m, n)}")
            if toks:
                cfg, sp, gm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                L.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(gm))
                if cfg.value >= 0:
                    toks.insert(0, f"{L.hgemm_mi355x_config_name(cfg.value).decode()}:{sp.value}:{gm.value}")
                print(r["mnk"], *dict.fromkeys(toks))
                selected.append(r["mnk"])
            continue
        want_sk = flops >= 1e9 and ((r["best"]["splits"] & 0xFFFF) > 1 or lt / r["best"]["us"] < a.below)
        skinny = min(m, n) <= 256 and k >= 2048 and max(m, n) >= 2048
        tiny_k = k <= 128
        tiny_mn = m * n <= 256 * 256 and k >= 512
        if want_sk:
            sk = []
            for cf in cfgs:
                if cf["sk"] > 0 and fits(cf, m, n, k) and not cf["name"].endswith("_d"):
                    for rr in range(1, cf["sk"] + 1):
                        plan = STREAMK | (256 * rr)
                        sk.append((L.hgemm_mi355x_model_us(cf["id"], plan, m, n, k), f"{cf['name']}:{plan}:{grp(cf, m, n)}"))
            toks += [t for _, t in sorted(sk)[:a.top_sk]]
        if skinny:
            for cf in cfgs:
                if cf["name"].endswith("_d") and fits(cf, m, n, k):
                    tiles = -(-m // cf["bm"]) * -(-n // cf["bn"])
                    stages = k // cf["kgran"]
                    toks.append(f"{cf['name']}:{STREAMK | 256}:{grp(cf, m, n)}")
                    for s in (1, 2, 4):
                        if s > 1 and (stages // s < 4 or tiles * s > 768):
                            continue
                        if tiles * s < 96:
                            continue
                        toks.append(f"{cf['name']}:{s | (FUSED if s > 1 else 0)}:{grp(cf, m, n)}")
                        if s == 2:
                            toks.append(f"{cf['name']}:{s}:{grp(cf, m, n)}")
        if tiny_k or tiny_mn:
            for cf in cfgs:
                if cf["name"][0] != "w" or not fits(cf, m, n, k):
                    continue
                k4 = cf["name"].endswith("_k4")
                tiles = -(-m // cf["bm"]) * -(-n // cf["bn"])
                if tiny_k:
                    if not k4:
                        toks.append(f"{cf['name']}:1:{grp(cf, m, n)}")
                    continue
                for s in (1, 2, 4, 8, 16, 32):
                    per = k // s
                    if per < (512 if k4 else 128) or per % 64 or tiles * s > 1024:
                        continue
                    if tiles * s < 32 and s < 32:
                        continue
                    toks.append(f"{cf['name']}:{s | (FUSED if s > 1 else 0)}:{grp(cf, m, n)}")
                    if s > 1 and tiles * s <= 64:
                        toks.append(f"{cf['name']}:{s}:{grp(cf, m, n)}")
        if not toks:
            continue
        cfg, sp, gm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(gm))
        if cfg.value >= 0:
            toks.insert(0, f"{L.hgemm_mi355x_config_name(cfg.value).decode()}:{sp.value}:{gm.value}")
        print(r["mnk"], *dict.fromkeys(toks))
        selected.append(r["mnk"])
    if a.shapes_out:
        Path(a.shapes_out).write_text("\n".join(selected) + "\n")
    print(f"# {len(selected)} shapes", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
