"""Candidate plans of the round-5 re-tune (bin/hgemm_tune tune --cand-file), one line per shape:

    M_N_K  config:splits:group  config:splits:group ...

Per grid shape the SHIPPED plan first (re-measured in the same run, same box), then

  --what overlap   the two-resident members of family q (CfgSQ::WGS == 2: q128x128_w2x2, q192x128_w2x2, q128x192_w2x2 -- two
                   workgroups per CU, one's epilogue under the other's K loop), with and without non-temporal C stores, at raster
                   groups 4 and 8, for shapes with >= `--min-items` work items of 192 x 128 and K <= `--max-k`;
  --what stagger   the shipped plan + HGEMM_PLAN_XCD_STAGGER (family q's kstagger variant) for every row whose shipped plan is a
                   16x16x32 member of family q and whose K is a whole number of its stages.

    python tools/make_round5_candidates.py --what overlap --shapes-out tuning/r05_overlap_shapes.txt > tuning/r05_overlap_candidates.txt
"""
from __future__ import annotations

import argparse
import re
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
NT, STAGGER, MASK, STREAMK = 0x20000, 0x80000, 0xFFFF, 0x40000
ROW = re.compile(r'\{(\d+), (\d+), (\d+), "([^"]+)", (\d+), (\d+)\}')


def shipped_table() -> list[tuple[int, int, int, str, int, int]]:
    rows = []
    for ln in (PKG_DIR / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = ROW.search(ln)
        if m:
            rows.append((int(m[1]), int(m[2]), int(m[3]), m[4], int(m[5]), int(m[6])))
    return rows


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--what", required=True, choices=["overlap", "stagger"])
    ap.add_argument("--min-items", type=int, default=384)
    ap.add_argument("--max-k", type=int, default=2048)
    ap.add_argument("--every", type=int, default=1, help="keep every n-th selected shape (first-look runs)")
    ap.add_argument("--shapes-out", default="")
    a = ap.parse_args(argv)
    lines, shapes = [], []
    for m, n, k, cfg, splits, group in shipped_table():
        cands = [f"{cfg}:{splits}:{group}"]
        if a.what == "overlap":
            items = -(-m // 192) * -(-n // 128)
            if items < a.min_items or k > a.max_k or k < 64:
                continue
            for member in ("q128x128_w2x2", "q192x128_w2x2", "q128x192_w2x2"):
                for nt in (NT, 0):
                    for g in (4, 8):
                        cands.append(f"{member}:{1 | nt}:{g}")
        else:
            stage = 128 if cfg.endswith("_k128") else 64
            if not cfg.startswith("q") or cfg.endswith("_m32") or (splits & STREAMK) or k % stage or k // stage < 8:
                continue
            cands.append(f"{cfg}:{splits | STAGGER}:{group}")
        shapes.append(f"{m}_{n}_{k}")
        lines.append(f"{m}_{n}_{k} " + " ".join(dict.fromkeys(cands)))
    lines, shapes = lines[::a.every], shapes[::a.every]
    print("\n".join(lines))
    if a.shapes_out:
        Path(a.shapes_out).write_text("\n".join(shapes) + "\n")
    print(f"{len(shapes)} shapes", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
