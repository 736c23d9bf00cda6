"""Candidate plans for the round-3 targeted re-tune of the geometries added late in the round (format of
`hgemm_tune tune --cand-file`; every line starts with the shipped plan, re-measured in the same run):

  * 8-wave 128x64 / 64x128 tiles (t128x64_w4x2, t64x128_w2x4; 4- and 3-deep rings) for shapes with few tiles: no split-K
    where the 4-wave members needed one, or the chip-filling split;
  * 192-row / 192-column members of family q for shapes with a dimension that is a multiple of 192 (12288 on the grid);
  * 96-row / 96-column members of family r for the skinny shapes with a 12288 dimension;
  * streaming C stores (HGEMM_PLAN_NT_STORE) with the persistent tiles on the small-K / large-MN shapes, where the
    tuner's own NT trial only ever saw the isolated-time winner.

    python tools/make_round3_candidates.py > tuning/r03_late_candidates.txt
"""
import ctypes
import re
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent.parent
FUSED, NT = 0x10000, 0x20000


def main() -> int:
    lib = ctypes.CDLL(str(PKG / "lib" / "libhgemm_mi355x.so"))
    lib.hgemm_mi355x_config_by_name.argtypes = [ctypes.c_char_p]
    names_ok = lambda n: lib.hgemm_mi355x_config_by_name(n.encode()) >= 0
    group_of = lambda n, M, N: lib.hgemm_mi355x_default_group(lib.hgemm_mi355x_config_by_name(n.encode()), M, N)
    rows = []
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))))
    cdiv = lambda a, b: -(-a // b)
    n_shapes = 0
    for (M, N, K, cfg, splits, group) in rows:
        cands = []

        def add(name, s, g=None):
            if not names_ok(name):
                return
            t = (name, s, g if g else group_of(name, M, N))
            if t not in cands:
                cands.append(t)

        def with_fill(name, bm, bn, per_cu, kgran, nt_too=False):
            if bm > 2 * M or bn > 2 * N or K % kgran:
                return
            tiles = cdiv(M, bm) * cdiv(N, bn)
            add(name, 1)
            if nt_too and M * N >= 1 << 20:
                add(name, 1 | NT)
            s = min(16, (256 * per_cu) // tiles)
            if s >= 2 and K // kgran // s >= 4:
                add(name, s | FUSED)
                if name[0] == "q":
                    add(name, s)

        # (C) 96-wide streaming tiles
        if K >= 2048 and min(M, N) <= 256:
            for name, bm, bn, per_cu in (("r96x128_k128", 96, 128, 1), ("r96x64_k128", 96, 64, 2), ("r128x96_k128", 128, 96, 1),
                                         ("r64x96_k128", 64, 96, 2)):
                if (bm == 96 and M % 96) or (bn == 96 and N % 96) or bm > M or bn > N:
                    continue
                tiles = cdiv(M, bm) * cdiv(N, bn)
                for s in (1, 2, 4):
                    if tiles * s <= 256 * per_cu and (s == 1 or tiles * s // 2 < 256 * per_cu) and K // 128 // s >= 4:
                        add(name, s | FUSED if s > 1 else 1)
                        if s == 2:
                            add(name, s)
        # (A) 8-wave mid tiles
        if M >= 128 and N >= 128 and K >= 1024:
            for name, bm, bn, per_cu in (("t128x64_w4x2_m16_s4", 128, 64, 1), ("t64x128_w2x4_m16_s4", 64, 128, 1),
                                         ("t128x64_w4x2_m16_s3", 128, 64, 2), ("t64x128_w2x4_m16_s3", 64, 128, 2)):
                if cdiv(M, bm) * cdiv(N, bn) <= 512:
                    with_fill(name, bm, bn, per_cu, 64)
        # (B) 192-wide persistent tiles
        if K >= 256 and M * N >= 192 * 256 * 32:
            if M % 192 == 0 and M // 192 >= 2:
                with_fill("q192x256_w2x2", 192, 256, 1, 64, nt_too=True)
            if N % 192 == 0 and N // 192 >= 2:
                with_fill("q256x192_w2x2", 256, 192, 1, 64, nt_too=True)
            if M % 192 == 0 and N % 192 == 0:
                with_fill("q192x192_w2x2", 192, 192, 1, 64, nt_too=True)
        # (D) streaming C stores on the small-K / large-MN shapes
        if K <= 512 and M * N >= 1 << 22:
            for name, bm, bn in (("q256x128_w2x2", 256, 128), ("q128x256_w2x2", 128, 256), ("q256x256_w2x2", 256, 256)):
                if bm <= M and bn <= N:
                    add(name, 1)
                    add(name, 1 | NT)
            if (splits & 0xFFFF) == 1 and not splits & NT:
                add(cfg, 1 | NT, group)
        if not cands:
            continue
        cands = [(cfg, splits, group)] + [c for c in cands if c != (cfg, splits, group)]
        print(f"{M}_{N}_{K} " + " ".join(f"{n}:{s}:{g}" for n, s, g in cands[:14]))
        n_shapes += 1
    print(f"# {n_shapes} shapes", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
