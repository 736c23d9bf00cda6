"""Writer of the per-shape kernel files kernels/mi355x_<acc>/<M>_<N>_<K>.hip.

One file per (M,N,K) per accumulate mode, like the reference's kernels/<dev>_<acc>/ trees
(SURVEY.md section 2 rows 1-4).  A file records the plan (geometry, split-K, raster group) measured
by the native tuner (bin/hgemm_tune) or, for shapes never tuned, the analytic model's choice.

  python tools/gen_shape_kernels.py --grid                       # all 1000 grid shapes, both modes
  python tools/gen_shape_kernels.py --tuned tuned.jsonl          # overwrite with tuner results
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent
if str(PKG_DIR) not in sys.path:
    sys.path.insert(0, str(PKG_DIR))

GRID_DIMS = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384]  # reference shape grid (10^3 shapes)
ACC_DIRS = {"fp16": "F16F16F16F16", "fp32": "F32F16F16F32"}
ACC_TEXT = {"fp16": "F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)",
            "fp32": "F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)"}


def grid_shapes():
    return [f"{m}_{n}_{k}" for m in GRID_DIMS for n in GRID_DIMS for k in GRID_DIMS]


_lib = None


def _library():
    global _lib
    if _lib is None:
        import build

        _lib = ctypes.CDLL(str(build.build_library()))
        _lib.hgemm_mi355x_config_name.restype = ctypes.c_char_p
    return _lib


def model_plan(m: int, n: int, k: int):
    lib = _library()
    cfg, splits, group = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    st = lib.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(splits), ctypes.byref(group))
    if st != 0:
        raise RuntimeError(f"hgemm_mi355x_plan failed: {st}")
    name = lib.hgemm_mi355x_config_name(cfg.value)
    return (name.decode() if name else "generic"), splits.value, group.value


def write_shape_file(mnk: str, acc: str, device_type: str = "mi355x", plan=None, source: str = "analytic model") -> Path:
    m, n, k = map(int, mnk.split("_"))
    if plan is None:
        plan = model_plan(m, n, k)
    cfg, splits, group = plan
    out_dir = PKG_DIR / "kernels" / f"{device_type}_{ACC_DIRS[acc]}"
    out_dir.mkdir(parents=True, exist_ok=True)
    path = out_dir / f"{mnk}.hip"
    entry = "hgemm_mi355x_fp32" if acc == "fp32" else "hgemm_mi355x_fp16"
    flags = "".join(f", {t}" for bit, t in ((0x80000, "K stagger per XCD"), (0x100000, "NT loads of the streamed operand"), (0x200000, "phase offset"),
                                            (0x800000, "phase offset x4"), (0x400000, "wave priority")) if splits & bit)
    flags = flags.replace(", phase offset, phase offset x4", ", phase offset x8")   # (both bits: eight phase groups)
    text = (
        f"// M={m} N={n} K={k}  {ACC_TEXT[acc]}  MI355X / gfx950\n"
        f"// plan: geometry {cfg}, " + (f"stream-K on {splits & 0xFFFF} workgroups" if splits & 0x40000 else f"split-K {splits & 0xFFFF}{' (single launch)' if splits & 0x10000 else ''}") +
        f"{', non-temporal C stores' if splits & 0x20000 else ''}{flags}, raster group {group}  [{source}]\n"
        f"// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def\n"
        f"#define HGEMM_SHAPE_FALLBACK {entry}\n"
        f"#include \"hgemm_shape_entry.hpp\"\n"
        f"HGEMM_MI355X_SHAPE_ENTRY({m}, {n}, {k}, \"{cfg}\", {splits}, {group})\n"
    )
    path.write_text(text)
    return path


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--grid", action="store_true", help="write every grid shape from the library's current plan")
    ap.add_argument("--tuned", type=str, help="JSON-lines file written by bin/hgemm_tune tune")
    ap.add_argument("--shapes", type=str, default="", help="comma separated M_N_K list")
    args = ap.parse_args()
    count = 0
    if args.tuned:
        for line in open(args.tuned):
            rec = json.loads(line)
            best = rec["best"]
            for acc in ACC_DIRS:
                write_shape_file(rec["mnk"], acc, plan=(best["config"], best["splits"], best["group_m"]),
                                 source=f"tuned on MI355X: {best['us']:.1f} us, {best['tflops']:.0f} TFLOP/s")
                count += 1
    shapes = grid_shapes() if args.grid else [s for s in args.shapes.split(",") if s]
    for mnk in shapes:
        for acc in ACC_DIRS:
            write_shape_file(mnk, acc)
            count += 1
    print(f"wrote {count} shape files")


if __name__ == "__main__":
    main()
