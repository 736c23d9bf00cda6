"""Split-K factors that FILL the chip, as extra candidates for a targeted re-tune.

The tuner's own list tries powers of two.  For an under-filled problem the split that matters is floor(slots / tiles):
12288 x 256 x 16384 has 48 tiles of 256 x 256 -- split 4 leaves a quarter of the 256 CUs idle and every slice 64 K-steps
long, split 5 fills 240 of them with 52-step slices (the launch path accepts any count: slices are ceil(steps / splits)
K-steps, the last one shorter).  Reads the shipped table and writes, for every shape whose plan's geometry has fewer tiles
than resident workgroup slots, the current plan (re-measured in the same run) plus the filling factor for the current
geometry and for the 256 x 256 / 128 x 128 persistent members, in both split-K forms:

    python tools/make_fill_candidates.py > tuning/r03_fill_candidates.txt     (format of `hgemm_tune tune --cand-file`)
"""
import re
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent.parent
FUSED = 0x10000


def geometry(name: str):
    """-> (BM, BN, lds_bytes, persistent, k_granularity) from a config name of csrc/hgemm_configs.def"""
    m = re.match(r"([tsqr])(\d+)x(\d+)", name)
    fam, bm, bn = m.group(1), int(m.group(2)), int(m.group(3))
    if fam == "t":
        nbuf = int(re.search(r"_s(\d+)$", name).group(1))
        return bm, bn, (bm + bn) * 128 * nbuf, False, 64
    if fam == "r":
        bks = int(re.search(r"_k(\d+)$", name).group(1))
        return bm, bn, (bm + bn) * bks * 2, False, bks
    kt = 2 if name.endswith("_k128") else 1
    return bm, bn, 2 * kt * (bm + bn) * 128 + 64, True, 64 * kt


def slots(name: str) -> int:
    bm, bn, lds, persistent, _ = geometry(name)
    per_cu = max(1, (160 * 1024) // lds)
    if not persistent:
        waves = 4 if name[0] == "r" else int(re.search(r"_w(\d+)x(\d+)", name).group(1)) * int(re.search(r"_w(\d+)x(\d+)", name).group(2))
        per_cu = max(1, min(per_cu, 16 // waves))
    return 256 * per_cu


def main() -> int:
    rows = []
    for ln in (PKG / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        m = re.match(r'\s*\{(\d+), (\d+), (\d+), "(\w+)", (\d+), (\d+)\}', ln)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), int(m.group(5)), int(m.group(6))))
    n = 0
    for (M, N, K, cfg, splits, group) in rows:
        cands = [(cfg, splits, group)]
        for name in dict.fromkeys([cfg, "q256x256_w2x2", "q128x128_w2x2_k128", "q256x128_w2x2", "q128x256_w2x2"]):
            bm, bn, _, _, kgran = geometry(name)
            if K % kgran or bm > 2 * M or bn > 2 * N:
                continue
            tiles = -(-M // bm) * -(-N // bn)
            s = slots(name) // tiles
            if s < 2 or (s & (s - 1)) == 0 or K // kgran // s < 2:
                continue                                         # powers of two were measured already
            g = group if name == cfg else max(1, min(8, -(-M // bm)))
            for form in (s, s | FUSED):
                if (name, form, g) not in cands:
                    cands.append((name, form, g))
        if len(cands) > 1:
            print(f"{M}_{N}_{K}", " ".join(f"{c}:{s}:{g}" for c, s, g in cands[:7]))
            n += 1
    print(f"# {n} shapes", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
