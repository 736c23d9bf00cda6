"""Pull one kernel's instruction stream out of a hipcc -save-temps gfx950 .s file.

  python tools/isa_extract.py FILE.s                      list kernels with instruction / MFMA / barrier / wait counts
  python tools/isa_extract.py FILE.s --kernel SUBSTR      print the instruction stream of the first kernel whose
                                                          mangled name contains SUBSTR (comments and directives dropped)
  ... --loop N                                            only the N-th innermost back-edge loop body (0 = first)
  ... --hist                                              mnemonic histogram instead of the stream

Used to audit the hand-scheduled families (what sits between two MFMAs, where the compiler put its waits) and to
compare a measurement build with the shipping build instruction by instruction.
"""
from __future__ import annotations

import argparse
import collections
import re
import sys


def kernels(text: str):
    """-> [(name, [instruction lines])] for every function that ends with .Lfunc_endN"""
    out = []
    lines = text.split("\n")
    name, body = None, []
    for ln in lines:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m:
            name, body = m.group(1), []
            continue
        if ln.startswith(".Lfunc_end"):
            if name:
                out.append((name, body))
            name = None
            continue
        if name is None:
            continue
        s = ln.split(";")[0].strip()
        if not s or s.startswith(".") and not s.endswith(":"):
            continue
        body.append(s)
    return out


def loops(body):
    """back edges: (start index, end index) of `s_cbranch* .LBBx_y` that jumps backwards"""
    label_at = {l[:-1]: i for i, l in enumerate(body) if l.endswith(":")}
    res = []
    for i, l in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < i:
            res.append((label_at[m.group(1)], i))
    return res


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("file")
    ap.add_argument("--kernel")
    ap.add_argument("--loop", type=int)
    ap.add_argument("--hist", action="store_true")
    a = ap.parse_args()
    ks = kernels(open(a.file).read())
    if not a.kernel:
        for name, body in ks:
            ins = [l for l in body if not l.endswith(":")]
            print(f"{name}\n   instrs {len(ins)}  mfma {sum(l.startswith('v_mfma') for l in ins)}  s_barrier "
                  f"{sum(l.startswith('s_barrier') for l in ins)}  s_waitcnt {sum(l.startswith('s_waitcnt') for l in ins)}  "
                  f"vmcnt(0) {sum('vmcnt(0)' in l for l in ins)}  lds-dma {sum(' lds' in l and l.startswith('buffer_load') for l in ins)}  "
                  f"ds_read {sum(l.startswith('ds_read') for l in ins)}  loops {[(e - s) for s, e in loops(body)]}")
        return 0
    for name, body in ks:
        if a.kernel in name:
            break
    else:
        print("no such kernel", file=sys.stderr)
        return 1
    if a.loop is not None:
        ls = loops(body)
        s, e = ls[a.loop]
        body = body[s:e + 1]
    if a.hist:
        h = collections.Counter(l.split()[0] for l in body if not l.endswith(":"))
        for k, v in h.most_common():
            print(f"{v:6d} {k}")
    else:
        print("\n".join(body))
    return 0


if __name__ == "__main__":
    sys.exit(main())
