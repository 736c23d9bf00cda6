"""Candidate plans for the off-grid shapes with a K tail (K % 64 != 0, K % 8 == 0), round 4: the plan the library shipped
before families q and r took K tails (a classic geometry, from the committed off-grid report), the plan the planner picks
now, and the obvious siblings -- so one tuner run prices old against new on the same box (tools/lab/gpu_round4_j.sh).

    python tools/make_ktail_candidates.py > tuning/r04_ktail_candidates.txt      (needs lib/libhgemm_mi355x.so, no GPU)
"""
import ctypes
import json
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parents[1]
FUSED, XCD, NT = 0x10000, 0x80000, 0x100000


def main():
    L = ctypes.CDLL(str(PKG / "lib" / "libhgemm_mi355x.so"))
    L.hgemm_mi355x_config_name.restype = ctypes.c_char_p
    L.hgemm_mi355x_config_by_name.argtypes = [ctypes.c_char_p]
    old = {}
    for ln in (PKG / "tuning" / "r04_offgrid_plan_report_mi355x.jsonl").read_text().splitlines():
        r = json.loads(ln)
        old[r["mnk"]] = r["best"]
    shapes = []
    for ln in (PKG / "tools" / "offgrid_shapes.txt").read_text().splitlines():
        ln = ln.strip()
        if not ln or ln.startswith("#"):
            continue
        m, n, k = map(int, ln.split("_"))
        if k % 64 and k % 8 == 0:
            shapes.append((ln, m, n, k))
    for mnk, m, n, k in shapes:
        c, s, g = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert L.hgemm_mi355x_plan(m, n, k, ctypes.byref(c), ctypes.byref(s), ctypes.byref(g)) == 0
        cands = []

        def add(name, splits, group):
            cid = L.hgemm_mi355x_config_by_name(name.encode())
            if cid < 0 or not L.hgemm_mi355x_config_accepts_k(cid, k):
                return
            info = (ctypes.c_int * 8)()
            L.hgemm_mi355x_config_info(cid, info)
            if (info[0] > 2 * m and info[0] > 32) or (info[1] > 2 * n and info[1] > 32):
                return
            tok = f"{name}:{splits}:{group}"
            if tok not in cands:
                cands.append(tok)

        o = old[mnk]
        add(o["config"], o["splits"], o["group_m"])                                   # shipped before
        add(L.hgemm_mi355x_config_name(c.value).decode(), s.value, g.value)           # the planner's plan now
        grp = g.value
        if min(m, n) <= 256:   # skinny: family r, unsplit / split, with and without its load flags
            for name in ("r64x64_k256", "r64x128_k128", "r128x64_k128", "r128x128_k128", "r64x128_k128_d", "r128x64_k128_d"):
                for sp in (1, 2 | FUSED, 4 | FUSED):
                    add(name, sp, grp)
                    add(name, sp | XCD | NT, grp)
        else:
            for name in ("q256x256_w2x2", "q256x128_w2x2", "q128x256_w2x2", "q128x128_w2x2_k128", "q128x128_w2x2", "q192x256_w2x2", "q256x192_w2x2"):
                add(name, 1, grp)
                if k >= 4096:
                    add(name, 2, grp)
        print(mnk, " ".join(cands[:24]))


if __name__ == "__main__":
    sys.exit(main())
