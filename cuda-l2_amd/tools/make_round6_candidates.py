"""Candidate plans of the round-6 re-tune (bin/hgemm_tune tune --cand-file), one line per shape:

    M_N_K  config:splits:group  config:splits:group ...

Per grid shape the SHIPPED plan first (re-measured in the same run, same box), then

  --what fused     round 6 batched the last arriver's combine of the single-launch split-K (fused_combine, csrc/hgemm_kernel.hpp:
                   up to 32 slabs in flight per round trip instead of one) -- the form that used to lose to the two-pass form's
                   second dispatch on all but the smallest splits.  For every row that ships a split-K plan, and for every small
                   output with a long K (M * N <= 1024^2, K >= 1024), the single-launch twin of the shipped plan, the same geometry
                   at half / twice the splits in both forms, and the small members of families t and w at the split counts that
                   give 64 .. 1024 workgroups, single-launch;
  --what cuphase   HGEMM_PLAN_CU_PHASE (the phase offset inside a CU, two-resident members of family q) on the small-K / large-MN
                   class: shapes with >= `--min-items` items of 128 x 128 and K <= `--max-k`: the three two-resident members with
                   and without the flag, with and without non-temporal C stores.

    python tools/make_round6_candidates.py --what fused --shapes-out tuning/r06_fused_shapes.txt > tuning/r06_fused_candidates.txt
"""
from __future__ import annotations

import argparse
import re
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
NT, STAGGER, MASK, STREAMK, FUSED, PHASE, PHASE4, CU_PHASE = 0x20000, 0x80000, 0xFFFF, 0x40000, 0x10000, 0x200000, 0x800000, 0x1000000
SMALL = {"w64x64": (64, 64, 64), "w32x128": (32, 128, 64), "w128x32": (128, 32, 64), "w32x64": (32, 64, 64), "w64x32": (64, 32, 64),
         "w16x16_k4": (16, 16, 128), "w32x32_k4": (32, 32, 128), "w16x32_k4": (16, 32, 128), "w32x16_k4": (32, 16, 128),
         "t32x64_w1x2_m16_s4": (32, 64, 64), "t64x32_w2x1_m16_s4": (64, 32, 64), "t64x64_w2x2_m16_s4": (64, 64, 64),
         "t32x32_w1x1_m16_s4": (32, 32, 64), "t128x64_w2x2_m16_s4": (128, 64, 64), "t64x128_w2x2_m16_s4": (64, 128, 64),
         "t128x128_w2x2_m16_s3": (128, 128, 64)}
ROW = re.compile(r'\{(\d+), (\d+), (\d+), "([^"]+)", (\d+), (\d+)\}')


def shipped_table():
    rows = []
    for ln in (PKG_DIR / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = ROW.search(ln)
        if m:
            rows.append((int(m[1]), int(m[2]), int(m[3]), m[4], int(m[5]), int(m[6])))
    return rows


def tile_of(cfg: str) -> tuple[int, int]:
    m = re.match(r"[a-z](\d+)x(\d+)", cfg)
    return int(m[1]), int(m[2])


def kgran_of(cfg: str) -> int:
    if cfg.endswith("_k128") or re.match(r"r\d+x\d+_k128", cfg):
        return 128
    if re.match(r"r\d+x\d+_k256", cfg):
        return 256
    return 64


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--what", required=True, choices=["fused", "cuphase"])
    ap.add_argument("--min-items", type=int, default=1024)
    ap.add_argument("--max-k", type=int, default=1024)
    ap.add_argument("--every", type=int, default=1, help="keep every n-th selected shape (first-look runs)")
    ap.add_argument("--shapes-out", default="")
    a = ap.parse_args(argv)
    lines, shapes = [], []
    for m, n, k, cfg, splits, group in shipped_table():
        cands = [f"{cfg}:{splits}:{group}"]

        def add(c):
            if c not in cands:
                cands.append(c)

        if a.what == "fused":
            sp = splits & MASK
            is_split = sp > 1 and not (splits & STREAMK)
            small_out = m * n <= 1024 * 1024 and k >= 1024
            bm, bn = tile_of(cfg)
            kg = kgran_of(cfg)
            tiles0 = -(-m // bm) * -(-n // bn)
            # a plan that leaves CUs idle because splitting used to cost a second dispatch or a serial combine: fewer than 1.5
            # work items per CU and a K long enough to cut (2048 x 1024 x 4096 ships 128 tiles of 128 x 128, unsplit)
            under_filled = not (splits & STREAMK) and tiles0 * sp < 384 and k >= 1024 and cfg[0] in "tqr"
            if not (is_split or small_out or under_filled):
                continue
            flags = splits & ~(MASK | FUSED | NT)
            if under_filled and not is_split:
                for s2 in (2, 4, 8):
                    if (k // kg) // s2 < 4 or tiles0 * s2 > 1024:
                        continue
                    add(f"{cfg}:{s2 | FUSED | flags}:{group}")
                    add(f"{cfg}:{s2 | flags}:{group}")
                    if cfg.startswith("q128x128_w2x2") and not cfg.endswith("_k128") and k % 128 == 0:
                        add(f"q128x128_w2x2_k128:{s2 | FUSED}:{group}")
                    if cfg.startswith("q"):
                        add(f"t128x128_w2x2_m16_s3:{s2 | FUSED}:{group}")
            if is_split:
                for s2 in (sp, sp * 2, max(2, sp // 2), sp * 4):
                    if s2 < 2 or s2 > 64 or (k // kg) // s2 < 2:
                        continue
                    add(f"{cfg}:{s2 | FUSED | flags}:{group}")
                    add(f"{cfg}:{s2 | flags}:{group}")
            if small_out:
                for name, (tm, tn, kg2) in SMALL.items():
                    if tm > 2 * m or tn > 2 * n:
                        continue
                    tiles = -(-m // tm) * -(-n // tn)
                    for s2 in (2, 4, 8, 16, 32):
                        if not (96 <= tiles * s2 <= 1024) or (k // kg2) // s2 < 2 or (name.startswith("w") and k % (32 * s2)):
                            continue
                        add(f"{name}:{s2 | FUSED}:{1 if tiles <= 64 else 4}")
        elif a.what == "cuphase":
            items = -(-m // 128) * -(-n // 128)
            if items < a.min_items or k > a.max_k or k < 64:
                continue
            for member in ("q128x128_w2x2", "q192x128_w2x2", "q128x192_w2x2"):
                for nt in (NT, 0):
                    add(f"{member}:{1 | nt | CU_PHASE}:4")
                    add(f"{member}:{1 | nt}:4")
                add(f"{member}:{1 | NT | CU_PHASE}:8")
        if len(cands) > 1:
            lines.append(f"{m}_{n}_{k} " + " ".join(cands))
            shapes.append(f"{m}_{n}_{k}")
    lines, shapes = lines[::a.every], shapes[::a.every]
    sys.stdout.write("\n".join(lines) + "\n")
    if a.shapes_out:
        Path(a.shapes_out).write_text("\n".join(shapes) + "\n")
    print(f"{len(lines)} shapes, {sum(len(l.split()) - 1 for l in lines)} candidates", file=sys.stderr)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
