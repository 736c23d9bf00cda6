"""Populate the on-disk cache of hipBLASLt autotune winners (include/hgemm_mi355x.h: hgemm_hipblaslt_autotune_set_cache).

The reference's strongest baseline is cuBLASLt auto-tuning: up to 100 heuristic candidates, 50 warm-up + 100 timed shuffled rounds,
median per candidate (cublas/fp32/hgemm_cublaslt_auto_tuning.cu:108-306), repeated by every benchmarking process.  Here the same
search (csrc/hgemm_baselines.hip: autotune_find, time-boxed by HGEMM_AUTOTUNE_MAX_SECONDS per layout) runs ONCE per
(layout, M, N, K, compute type) and its winner's solution index is written to the cache file; the sweeps, eval_one_file.sh and
hgemm_tune then reuse it through HGEMM_AUTOTUNE_CACHE.  No torch: plain ctypes over the C ABI.

  HGEMM_AUTOTUNE_MAX_SECONDS=1 python tools/build_autotune_cache.py --cache tuning/r06_hipblaslt_autotune_cache.txt \
      --shapes-file tools/grid_shapes_shuffled.txt [--acc fp32] [--time_limit 1500]

Resumable: problems already in the cache (searched with at least the current budget) are skipped by the library itself.
hipBLASLt has no HIPBLAS_COMPUTE_16F kernels for these problems on gfx950 (every fp16-accumulate request of rounds 2-5 fell back to
32F compute: `hipblaslt_compute16_fallback` = 1 in 4000 of 4000 sweep records), so the fp16 tree resolves to the same records; the
script reports the fallback flag it saw so that this stays checked.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent.parent


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--cache", type=Path, required=True)
    ap.add_argument("--shapes-file", type=Path, required=True)
    ap.add_argument("--acc", choices=["fp32", "fp16"], default="fp32")
    ap.add_argument("--time_limit", type=float, default=0.0, help="stop after this many seconds (resumable)")
    ap.add_argument("--report", type=Path, default=None, help="write a JSON summary here")
    ap.add_argument("--with-torch", action="store_true", help="import torch first, so that the hipBLASLt the torch wheel bundles is the one "
                    "mapped -- the build the harness processes (benchmarking_*.py, tools/sweep.py) search and run.  Without it this "
                    "process links /opt/rocm's hipBLASLt, the build bin/hgemm_tune uses.  A solution index is only valid for the build "
                    "that searched it (the library checks the recorded solution name), so the two flavours live side by side in one file")
    args = ap.parse_args(argv)
    if args.with_torch:
        import torch  # noqa: F401  (side effect: its bundled ROCm libraries are loaded before libhgemm_mi355x.so asks for them)
    lib = ctypes.CDLL(str(PKG_DIR / "lib" / "libhgemm_mi355x.so"))
    lib.hgemm_hipblaslt_autotune_best_ms.restype = ctypes.c_double
    lib.hgemm_hipblaslt_autotune_set_cache.argtypes = [ctypes.c_char_p]
    args.cache.parent.mkdir(parents=True, exist_ok=True)
    if lib.hgemm_hipblaslt_autotune_init() != 0:
        print("hipBLASLt init failed (no GPU?)", file=sys.stderr)
        return 1
    lib.hgemm_hipblaslt_autotune_set_cache(str(args.cache).encode())
    acc = 0 if args.acc == "fp32" else 1
    shapes = [ln.strip() for ln in args.shapes_file.read_text().splitlines() if ln.strip() and not ln.startswith("#")]
    t0 = time.time()
    searched = hit = failed = 0
    fallback = set()
    search_s = 0.0
    for i, mnk in enumerate(shapes):
        m, n, k = (int(x) for x in mnk.split("_"))
        for tn, find in ((1, lib.hgemm_hipblaslt_autotune_find_best_tn), (0, lib.hgemm_hipblaslt_autotune_find_best_nn)):
            t1 = time.time()
            st = find(m, n, k, acc)
            if st != 0:
                failed += 1
                continue
            if lib.hgemm_hipblaslt_autotune_from_cache(tn):
                hit += 1
            else:
                searched += 1
                search_s += time.time() - t1
            fallback.add(lib.hgemm_hipblaslt_compute16_fallback(1, tn))
        if (i + 1) % 50 == 0:
            print(f"{i + 1}/{len(shapes)} shapes, {searched} searched ({search_s:.0f} s), {hit} cached, {failed} failed, {time.time() - t0:.0f} s", flush=True)
        if args.time_limit and time.time() - t0 > args.time_limit:
            print(f"time limit after {i + 1} shapes")
            break
    h, ms = ctypes.c_int(), ctypes.c_int()
    records = lib.hgemm_hipblaslt_autotune_cache_stats(ctypes.byref(h), ctypes.byref(ms))
    lib.hgemm_hipblaslt_autotune_destroy()
    out = {"cache": str(args.cache), "hipblaslt": "torch wheel" if args.with_torch else "/opt/rocm", "records": records, "searched": searched, "cache_hits": hit, "failed": failed,
           "search_seconds": round(search_s, 1), "wall_seconds": round(time.time() - t0, 1), "acc": args.acc,
           "compute16_fallback_seen": sorted(fallback)}
    print(json.dumps(out))
    if args.report:
        args.report.write_text(json.dumps(out, indent=1) + "\n")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
