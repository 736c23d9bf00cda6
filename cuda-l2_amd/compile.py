"""Pre-build one shape's `hgemm_lib` extension (reference: compile.py:1-22 warms the JIT cache).

On a box without a GPU this only builds (hipcc cross-compiles gfx950); with --all-libs it builds
libhgemm_mi355x.so and the native tuner as well.
"""
import argparse
import time
from pathlib import Path

import build
from tools.utils import DEVICE_TYPES, PKG_DIR, ensure_kernel_source


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__)
    parser.add_argument("--base_dir", type=str, required=True)
    parser.add_argument("--mnk", type=str, required=True)
    parser.add_argument("--acc_precise", type=str, required=True, choices=["fp16", "fp32"])
    parser.add_argument("--device_type", type=str, required=True, choices=DEVICE_TYPES)
    parser.add_argument("--all-libs", action="store_true")
    args = parser.parse_args(argv)
    t0 = time.time()
    if args.all_libs:
        build.build_tools()
    so = build.build_extension(ensure_kernel_source(args.mnk, args.acc_precise, args.device_type),
                               PKG_DIR / "pybind" / f"hgemm_{args.device_type}_{args.acc_precise}.cc",
                               Path(args.base_dir))
    print(f"Compile hgemm module time: {time.time() - t0:.2f} seconds -> {so}")


if __name__ == "__main__":
    main()
