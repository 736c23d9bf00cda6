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
NT, STAGGER, MASK, STREAMK, FUSED, PHASE, PHASE4 = 0x20000, 0x80000, 0xFFFF, 0x40000, 0x10000, 0x200000, 0x800000
W_MEMBERS = {"w64x64": (64, 64), "w32x128": (32, 128), "w128x32": (128, 32), "w32x64": (32, 64), "w64x32": (64, 32),
             "w16x16_k4": (16, 16), "w32x32_k4": (32, 32), "w16x32_k4": (16, 32), "w32x16_k4": (32, 16)}
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
    ap.add_argument("--what", required=True, choices=["overlap", "stagger", "wdeep", "phase", "raster", "mid", "pass2", "pass3", "pass4"])
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
        elif a.what == "wdeep":
            # family w after the deeper trips of round 5 (16 / 8 slices per round trip): every member x split count that yields
            # 32 .. 1024 workgroups with >= 128 of K per split, both split-K forms -- on the rows family w serves and on the small
            # outputs with a long K where it lost to hipBLASLt (K >= 256, M * N <= 1024^2)
            if not (cfg.startswith("w") or (m * n <= 1024 * 1024 and k >= 256 and 2.0 * m * n * k < 4e10)):
                continue
            for name, (bm, bn) in W_MEMBERS.items():
                tiles = -(-m // bm) * -(-n // bn)
                for sp in (1, 2, 4, 8, 16, 32):
                    if not (32 <= tiles * sp <= 1024) or (sp > 1 and k // sp < 128) or k % (32 * sp):
                        continue
                    g = 1 if tiles <= 64 else 4
                    cands.append(f"{name}:{sp}:{g}")
                    if sp > 1:
                        cands.append(f"{name}:{sp | FUSED}:{g}")
        elif a.what == "phase":
            # HGEMM_PLAN_PHASE_OFFSET on the persistent plans whose workgroups walk several items of a short K
            if not cfg.startswith("q") or (splits & STREAMK) or k > 1024:
                continue
            bm, bn = map(int, re.match(r"q(\d+)x(\d+)", cfg).groups())
            items = -(-m // bm) * -(-n // bn) * max(1, splits & MASK)
            if items < 512:
                continue
            cands.append(f"{cfg}:{splits | PHASE}:{group}")
            cands.append(f"{cfg}:{splits | PHASE4}:{group}")
            if k >= 512 and k % 64 == 0 and not cfg.endswith("_m32"):
                cands.append(f"{cfg}:{splits | PHASE | STAGGER}:{group}")
        elif a.what == "mid":
            # the two-resident members (q128x128 never ran two per CU before round 5) with split-K on the mid class: 128 .. 1024 work
            # items (two per CU resident), >= 256 of K per split; two-pass and single-launch form, with and without the stagger
            if not (5e9 <= 2.0 * m * n * k <= 3e11) or k < 512:
                continue
            for member, (bm, bn) in (("q128x128_w2x2", (128, 128)), ("q192x128_w2x2", (192, 128)), ("q128x192_w2x2", (128, 192))):
                tiles = -(-m // bm) * -(-n // bn)
                for sp in (1, 2, 3, 4, 6, 8):
                    if not (128 <= tiles * sp <= 1024) or k % (64 * sp) or k // sp < 256:
                        continue
                    g = 4 if tiles >= 32 else 1
                    for fl in ((0, FUSED) if sp > 1 else (NT,)):
                        cands.append(f"{member}:{sp | fl}:{g}")
                        if k // sp >= 512:
                            cands.append(f"{member}:{sp | fl | STAGGER}:{g}")
        elif a.what == "pass2":
            # second pass, against the table as pass 1 left it:
            #  * skinny / mid shapes with a long K (min(M, N) <= 1024, K >= 1024): pass 1 found the two-resident q128x128 with
            #    split-K ahead of family r by up to 20 % -- the wider sweep: three members x splits up to 32 x both forms x stagger;
            #  * the flags in combination on the persistent plans: raster group x stagger on the >= 1e11-FLOP rows, phase offset
            #    (two / four groups) x stagger on the walks of several short-K items.
            TWO_RES = (("q128x128_w2x2", (128, 128)), ("q192x128_w2x2", (192, 128)), ("q128x192_w2x2", (128, 192)))
            if min(m, n) <= 1024 and k >= 1024 and 2.0 * m * n * k >= 1e9:
                for member, (bm, bn) in TWO_RES:
                    tiles = -(-m // bm) * -(-n // bn)
                    for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                        if not (128 <= tiles * sp <= 1536) or k % (64 * sp) or k // sp < 256:
                            continue
                        g = 4 if tiles >= 32 else 1
                        for fl in ((0, FUSED) if sp > 1 else (NT, 0)):
                            cands.append(f"{member}:{sp | fl}:{g}")
                            if k // sp >= 512:
                                cands.append(f"{member}:{sp | fl | STAGGER}:{g}")
            if cfg.startswith("q") and not cfg.endswith("_m32") and not (splits & STREAMK):
                stage = 128 if cfg.endswith("_k128") else 64
                sp = max(1, splits & MASK)
                can_stag = k % stage == 0 and k // stage // sp >= 8
                bm, bn = map(int, re.match(r"q(\d+)x(\d+)", cfg).groups())
                items = -(-m // bm) * -(-n // bn) * sp
                wgs = 512 if cfg.split("_")[0] in ("q128x128", "q192x128", "q128x192") and not cfg.endswith("_k128") else 256
                base = splits & ~(STAGGER | PHASE | PHASE4)
                if 2.0 * m * n * k >= 1e11:
                    for g in (2, 4, 8):
                        cands.append(f"{cfg}:{base}:{g}")
                        if can_stag:
                            cands.append(f"{cfg}:{base | STAGGER}:{g}")
                if items >= 2 * wgs and k <= 1024:
                    for ph in (0, PHASE, PHASE4):
                        cands.append(f"{cfg}:{base | ph}:{group}")
                        if can_stag:
                            cands.append(f"{cfg}:{base | ph | STAGGER}:{group}")
            if len(cands) == 1:
                continue
        elif a.what == "pass3":
            # third pass: the phase offset in eight groups (both flag bits), against two / four groups and none, with and without the
            # stagger, on every persistent plan whose workgroups walk >= 2 items of K <= 2048
            if not cfg.startswith("q") or cfg.endswith("_m32") or (splits & STREAMK) or k > 2048:
                continue
            stage = 128 if cfg.endswith("_k128") else 64
            sp = max(1, splits & MASK)
            can_stag = k % stage == 0 and k // stage // sp >= 8
            bm, bn = map(int, re.match(r"q(\d+)x(\d+)", cfg).groups())
            items = -(-m // bm) * -(-n // bn) * sp
            wgs = 512 if cfg.split("_")[0] in ("q128x128", "q192x128", "q128x192") and not cfg.endswith("_k128") else 256
            if items < 2 * wgs:
                continue
            base = splits & ~(STAGGER | PHASE | PHASE4)
            for ph in (0, PHASE, PHASE4, PHASE | PHASE4):
                cands.append(f"{cfg}:{base | ph}:{group}")
                if can_stag:
                    cands.append(f"{cfg}:{base | ph | STAGGER}:{group}")
        elif a.what == "pass4":
            # fourth pass: the phase offset with its spacing capped at an epilogue's length (so that it costs a long-K walk ~10k cycles
            # once instead of half an item period): every persistent plan whose workgroups walk more than one item, any K
            if not cfg.startswith("q") or (splits & STREAMK):
                continue
            sp = max(1, splits & MASK)
            bm, bn = map(int, re.match(r"q(\d+)x(\d+)", cfg).groups())
            items = -(-m // bm) * -(-n // bn) * sp
            wgs = 512 if cfg.split("_")[0] in ("q128x128", "q192x128", "q128x192") and not cfg.endswith("_k128") else 256
            if items <= wgs:
                continue
            base = splits & ~(PHASE | PHASE4)
            for ph in (0, PHASE, PHASE4, PHASE | PHASE4):
                cands.append(f"{cfg}:{base | ph}:{group}")
        elif a.what == "raster":
            # raster groups of the largest persistent plans: the shipped group against 4 and 8 (the 8 x 4 / 4 x 8 patches whose
            # operand panels are the geometric floor of an XCD's fabric traffic, profiles/r05_pmc_traffic_vs_k.json)
            if not cfg.startswith("q") or (splits & STREAMK) or 2.0 * m * n * k < 2e11:
                continue
            for g in (2, 4, 8):
                cands.append(f"{cfg}:{splits}:{g}")
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
