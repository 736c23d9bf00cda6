"""Per-kernel fingerprints of the device code (gfx950 ISA as hipcc -S prints it), so a later change to the sources can show
which kernels' instruction streams it touched -- and therefore which committed measurements still describe the library.

    python tools/isa_fingerprint.py [--csrc DIR --include DIR] > profiles/<name>.json        (no GPU; ~20 s, four hipcc in parallel)

A kernel's fingerprint is the sha256 of its function body between the label and s_endpgm with comments, assembler directives
and basic-block label NUMBERS stripped (labels are numbered per translation unit: adding a kernel renumbers them all).
tests/test_build_audit.py compares the current sources with profiles/r04_isa_fingerprint_closing_run_library.json.
"""
import argparse
import hashlib
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

PKG = Path(__file__).resolve().parents[1]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def compile_groups(csrc: Path, include: Path, opt: str = "-O3") -> str:
    """hipcc -S --cuda-device-only of the four instantiation units (the flags of tests/test_build_audit.py); returns the text."""
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for grp in range(4):
            out = Path(tmp) / f"g{grp}.s"
            procs.append((out, subprocess.Popen([HIPCC, "--offload-arch=gfx950", opt, "-std=c++17", f"-I{csrc}", f"-I{include}", "-S", "--cuda-device-only",
                                                 str(csrc / f"hgemm_inst_g{grp}.hip"), "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
        text = ""
        for out, pr in procs:
            _, err = pr.communicate(timeout=900)
            if pr.returncode != 0:
                raise RuntimeError(err.decode()[-2000:])
            text += out.read_text()
    return text


def kernel_bodies(text: str) -> dict:
    out = {}
    for m in re.finditer(r"^(_ZN12hgemm_mi355x\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        body = re.sub(r";.*", "", m.group(2))
        body = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", body)
        out[m.group(1)] = "\n".join(ln.rstrip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith("."))
    return out


def fingerprints(text: str) -> dict:
    return {k: hashlib.sha256(v.encode()).hexdigest()[:20] for k, v in sorted(kernel_bodies(text).items())}


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--csrc", default=str(PKG / "csrc"))
    ap.add_argument("--include", default=str(PKG.parent / "include"))
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    fp = fingerprints(compile_groups(Path(a.csrc), Path(a.include)))
    json.dump({"note": a.note, "flags": "hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only hgemm_inst_g{0..3}.hip", "kernels": fp}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
