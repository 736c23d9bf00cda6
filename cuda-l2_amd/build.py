"""Build driver for the MI355X HGEMM library (hipcc, gfx950 only).

Replaces the reference's JIT path (tools/utils.py:39-107: torch cpp_extension.load with nvcc flags
and the CUTLASS include) with explicit, content-hash-cached hipcc invocations:

  csrc/*.hip                      -> lib/libhgemm_mi355x.so   (C ABI, include/hgemm_mi355x.h)
  csrc/tools/hgemm_tune.cpp       -> bin/hgemm_tune           (native autotuner / micro-bench)
  pybind/hgemm_mi355x_<acc>.cc    -> objects cached under build/ (linked per shape by tools/utils.py)

Objects are cached under build/obj keyed by the hash of (command line, source, every header in
csrc/ and include/), so a 1000-shape sweep compiles each distinct source once.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
REPO_DIR = PKG_DIR.parent
CSRC = PKG_DIR / "csrc"
INCLUDE = REPO_DIR / "include"
BUILD = PKG_DIR / "build"
OBJ = BUILD / "obj"
# experiment builds: HGEMM_LIB_SUFFIX=nt HGEMM_EXTRA_HIPFLAGS="-DHGEMM_DMA_AUX=2" python build.py -> lib_nt/
_SUFFIX = os.environ.get("HGEMM_LIB_SUFFIX", "")
LIB_DIR = PKG_DIR / ("lib_" + _SUFFIX if _SUFFIX else "lib")
BIN_DIR = PKG_DIR / "bin"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
HIPCC = str(ROCM / "bin" / "hipcc")
ARCH = "gfx950"

LIB_SOURCES = [
    "hgemm_inst_g0.hip",
    "hgemm_inst_g1.hip",
    "hgemm_inst_g2.hip",
    "hgemm_inst_g3.hip",
    "hgemm_registry.hip",
    "hgemm_api.hip",
    "hgemm_baselines.hip",
]

HIP_FLAGS = os.environ.get("HGEMM_EXTRA_HIPFLAGS", "").split() + [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-Wno-unused-result", "-Wno-unused-value", "-DROCBLAS_NO_DEPRECATED_WARNINGS", "-D__HIP_PLATFORM_AMD__"]


def _headers_digest() -> str:
    h = hashlib.sha256()
    for d in (CSRC, INCLUDE):
        for p in sorted(d.rglob("*")):
            if p.suffix in (".hpp", ".h", ".def", ".inc"):
                h.update(p.name.encode())
                h.update(p.read_bytes())
    return h.hexdigest()


def _run(cmd: list[str], verbose: bool) -> None:
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"build step failed ({res.returncode}): {' '.join(cmd)}")
    if verbose and (res.stdout or res.stderr):
        print(res.stdout + res.stderr, flush=True)


def compile_object(src: Path, extra_flags: list[str], hdr_digest: str, verbose: bool = False) -> Path:
    """hipcc -c one source into the object cache; returns the object path."""
    OBJ.mkdir(parents=True, exist_ok=True)
    cmd_core = [HIPCC, *HIP_FLAGS, *extra_flags, f"-I{CSRC}", f"-I{INCLUDE}", "-c"]
    key = hashlib.sha256()
    key.update(" ".join(cmd_core).encode())
    key.update(hdr_digest.encode())
    key.update(src.read_bytes())
    obj = OBJ / f"{src.stem}-{key.hexdigest()[:20]}.o"
    if not obj.exists():
        tmp = obj.with_suffix(f".tmp{os.getpid()}.o")
        _run([*cmd_core, str(src), "-o", str(tmp)], verbose)
        os.replace(tmp, obj)
    return obj


def build_library(verbose: bool = False, jobs: int | None = None) -> Path:
    """Build (or reuse) lib/libhgemm_mi355x.so and return its path."""
    hdr = _headers_digest()
    jobs = jobs or min(8, os.cpu_count() or 1)
    srcs = [CSRC / s for s in LIB_SOURCES]
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: compile_object(s, [], hdr, verbose), srcs))
    LIB_DIR.mkdir(exist_ok=True)
    lib = LIB_DIR / "libhgemm_mi355x.so"
    stamp = LIB_DIR / ".libhgemm_mi355x.stamp"
    want = hashlib.sha256(" ".join(o.name for o in objs).encode()).hexdigest()
    if lib.exists() and stamp.exists() and stamp.read_text() == want:
        return lib
    tmp = lib.with_suffix(f".tmp{os.getpid()}.so")
    _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", *map(str, objs), "-o", str(tmp),
          f"-L{ROCM / 'lib'}", "-lrocblas", "-lhipblaslt", "-lamdhip64",
          "-Wl,--no-undefined", "-Wl,-soname,libhgemm_mi355x.so"], verbose)
    os.replace(tmp, lib)
    stamp.write_text(want)
    return lib


def build_tools(verbose: bool = False) -> list[Path]:
    """Native tools linked against the library (no torch): tuner / micro-benchmark."""
    lib = build_library(verbose)
    hdr = _headers_digest()
    BIN_DIR.mkdir(exist_ok=True)
    out = []
    for name in ("hgemm_tune",):
        src = CSRC / "tools" / f"{name}.cpp"
        if not src.exists():
            continue
        obj = compile_object(src, ["-x", "hip"], hdr, verbose)
        exe = BIN_DIR / name
        stamp = BIN_DIR / f".{name}.stamp"
        want = obj.name + lib.name + str(lib.stat().st_mtime_ns)
        if not (exe.exists() and stamp.exists() and stamp.read_text() == want):
            _run([HIPCC, f"--offload-arch={ARCH}", str(obj), "-o", str(exe), f"-L{LIB_DIR}", "-lhgemm_mi355x",
                  f"-L{ROCM / 'lib'}", "-lamdhip64", "-lrocm_smi64", "-Wl,-rpath,$ORIGIN/../lib", f"-Wl,-rpath,{ROCM / 'lib'}"],
                 verbose)
            stamp.write_text(want)
        out.append(exe)
    return out


def _torch_build_env():
    import sysconfig

    from torch.utils.cpp_extension import include_paths, library_paths

    incs = include_paths() + [sysconfig.get_paths()["include"], str(ROCM / "include")]
    return incs, library_paths()


def build_extension(kernel_src: Path, pybind_src: Path, base_dir: Path, name: str = "hgemm_lib",
                    verbose: bool = False) -> Path:
    """Link {base_dir}/{name}.so = torch shim (pybind_src) + per-shape plan (kernel_src) against
    lib/libhgemm_mi355x.so.  Both objects come from the content-hash cache, so a sweep over many
    shapes pays the ~30 s torch-header compile of the shim exactly once."""
    lib = build_library(verbose)
    hdr = _headers_digest()
    incs, libdirs = _torch_build_env()
    shim_flags = ["-x", "c++", "-DUSE_ROCM=1", f"-DTORCH_EXTENSION_NAME={name}", "-DTORCH_API_INCLUDE_EXTENSION_H",
                  "-D_GLIBCXX_USE_CXX11_ABI=1", "-Wno-deprecated-declarations", f"-I{PKG_DIR / 'pybind'}",
                  *[f"-I{i}" for i in incs]]
    shim_obj = compile_object(pybind_src, shim_flags, hdr, verbose)
    kern_obj = compile_object(kernel_src, ["-x", "c++"], hdr, verbose)
    base_dir.mkdir(parents=True, exist_ok=True)
    out = base_dir / f"{name}.so"
    stamp = base_dir / f".{name}.stamp"
    want = f"{shim_obj.name} {kern_obj.name} {lib.stat().st_mtime_ns}"
    if out.exists() and stamp.exists() and stamp.read_text() == want:
        return out
    tmp = base_dir / f"{name}.tmp{os.getpid()}.so"
    _run([HIPCC, "-shared", "-fPIC", str(shim_obj), str(kern_obj), "-o", str(tmp), f"-L{LIB_DIR}", "-lhgemm_mi355x",
          *[f"-L{d}" for d in libdirs], "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
          f"-L{ROCM / 'lib'}", "-lamdhip64", f"-Wl,-rpath,{LIB_DIR}", *[f"-Wl,-rpath,{d}" for d in libdirs]], verbose)
    os.replace(tmp, out)
    stamp.write_text(want)
    return out


def clean() -> None:
    for d in (BUILD, LIB_DIR, BIN_DIR):
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--clean", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--tools", action="store_true", help="(default since round 5) also build bin/hgemm_tune")
    ap.add_argument("--lib-only", action="store_true", help="the library alone")
    a = ap.parse_args()
    if a.clean:
        clean()
    print(build_library(a.verbose))
    # The tool is rebuilt with the library by default: round 5 ran three GPU calls with a stale bin/hgemm_tune (the library was
    # rebuilt, the tool was not) whose `check` did not know the new plan forms.
    if not a.lib_only:
        for t in build_tools(a.verbose):
            print(t)
