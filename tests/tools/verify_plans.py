"""Parity of the SHIPPED plans on the whole shape grid, against the CPU oracle (test infrastructure).

The reference runs zero_one_correctness_check.py before it benchmarks every shape
(eval_one_file.sh:71-80).  This tool does the same for every row of tools/grid_shapes.txt, for both
entry points, through the C ABI:

  * inputs {0,1} (P = 1/2), or {0,0,1} when max(M,N,K) > 8192       zero_one_correctness_check.py:65-73
  * truth = (a.cpu().float() @ b.cpu().float()).half()              :85-90  (oracle.truth_prefix_k)
  * operands and C are views into flat buffers with 16384-element guard bars either side, C pre-filled
    with NaN                                                         :98-150
  * pass rule: max |out - truth| over the elements with |truth| <= 2047 must be exactly 0, and no byte
    outside the operand windows may change                           :92,167-172,263-268
  * additionally (stronger than the reference): bitwise equality of the WHOLE tile, unmasked -- fp32
    accumulation of 0/1 products is exact, so the rounded result must match everywhere.

One operand pair per input class serves all shapes: the (M,N,K) problem is the top-left sub-problem of the
16384^3 (or 8192^3) one, and the oracle's incremental K-prefix truths make the CPU side of the whole grid
~1e13 flop instead of 1.8e14.

  python tests/tools/verify_plans.py --out cuda-l2_amd/tuning/r02_parity_1000.jsonl             # shipped plans
  python tests/tools/verify_plans.py --plans tune.jsonl --top 3 --out verify_candidates.jsonl   # tuner candidates

The second form checks the `--top` fastest candidates of every shape of a tuner result file with explicit
plans; tools/make_tuned_table.py --verified only accepts plans that have an exact record there.

  python tests/tools/verify_plans.py --randn --out cuda-l2_amd/tuning/r04_randn_1000.jsonl       # N(0,1) tolerance, shipped plans

--randn: BASELINE.json's floating-point bar ("every shape must match torch.matmul within 1e-2 rel (fp16 acc) / 1e-3 rel
(fp32 acc)"; the reference itself has no such test, zero_one_correctness_check.py:65-92 is 0/1 only): per shape N(0,1)
operands (the top-left sub-problem of one 16384^3 pair), both entry points, relative error max|C - ref| / max|ref| <= 1e-3
over >= 128 sampled rows of C (all of them when M <= 128; the first and last row, the rows either side of every 64 / 96-row
tile seam that is sampled, random ones otherwise).  ref = the fp32 product of the sampled rows computed on the CPU
(torch fp32, the oracle expression's arithmetic) when it is cheap (N * K <= 2^24) and in fp64 on the GPU otherwise -- neither
involves the library under test.  What 0/1 inputs cannot see is seen here: the fp32 -> fp16 rounding of the epilogue,
fractional partial sums through the split-K / stream-K slabs, denormal products.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
for p in (str(REPO), str(REPO / "cuda-l2_amd"), str(REPO / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import gpu_common  # noqa: E402
from oracle import hgemm_oracle as oracle  # noqa: E402

BAR = oracle.BAR_SIZE


class Guarded:
    """A [rows, cols] fp16 window inside a flat buffer with BAR guard elements either side."""

    def __init__(self, rows: int, cols: int):
        self.n = rows * cols
        self.flat = torch.randn(self.n + 2 * BAR, dtype=torch.half, device="cuda")
        self.lo = self.flat[:BAR].clone()
        self.hi = self.flat[-BAR:].clone()
        self.view = self.flat[BAR:BAR + self.n].view(rows, cols)

    def ptr(self) -> int:
        return self.flat.data_ptr() + BAR * 2

    def bars_intact(self) -> bool:
        return bool(torch.equal(self.flat[:BAR], self.lo) and torch.equal(self.flat[-BAR:], self.hi))


def check_output(c: Guarded, truth: torch.Tensor) -> tuple[float, bool]:
    """(masked max diff as the reference computes it, whole-tile bitwise equality)."""
    out = c.view
    t32 = truth.float()
    diff = (out.float() - t32).abs()
    diff = torch.nan_to_num(diff, nan=float("inf"))          # an unwritten (NaN) output is a failure, not a skip
    diff = torch.where(t32.abs() > oracle.MAX_EXACT_FP16_INT, torch.zeros_like(diff), diff)
    return float(diff.max().item()), bool(torch.equal(out.view(torch.int16), truth.view(torch.int16)))


def sample_rows(m: int, rng: np.random.Generator, want: int = 128) -> np.ndarray:
    """Rows of C the N(0,1) check looks at: all of them up to `want`, else the edges, tile seams and random rows."""
    if m <= want:
        return np.arange(m)
    rows = {0, m - 1}
    for seam in rng.choice(np.arange(1, m // 32), size=min(16, m // 32 - 1), replace=False) * 32:   # (64 / 96 / 128 / 192 / 256-row tile seams)
        rows.update((int(seam) - 1, int(seam)))
    while len(rows) < want:
        rows.add(int(rng.integers(0, m)))
    return np.array(sorted(rows))


def randn_pass(a, L, stream, shapes) -> int:
    torch.manual_seed(a.seed)
    rng = np.random.default_rng(a.seed)
    dm, dn, dk = (max(s[i] for s in shapes) for i in range(3))
    a_full = torch.randn((dm, dk), dtype=torch.half, device="cuda")
    b_full = torch.randn((dk, dn), dtype=torch.half, device="cuda")
    out_f = open(a.out, "w")
    t_start = time.time()
    n_checks = n_fail = 0
    worst = 0.0
    for (m, n, k) in shapes:
        av = a_full[:m, :k].contiguous()
        bv = b_full[:k, :n].contiguous()
        btv = bv.t().contiguous()                                   # as_col_major storage (tools/utils.py:110-115)
        rows = sample_rows(m, rng)
        ridx = torch.from_numpy(rows).cuda()
        where = "cpu fp32" if n * k <= (1 << 24) else "gpu fp64"
        if where == "cpu fp32":
            ref = (av[ridx].cpu().float() @ bv.cpu().float()).double().cuda()
        else:
            ref = av[ridx].double() @ bv.double()
        scale = float(ref.abs().max().item())
        cfg, sp, gm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(gm))
        name = L.hgemm_mi355x_config_name(cfg.value)
        c = torch.empty((m, n), dtype=torch.half, device="cuda")
        for entry in ("fp32", "fp16"):
            c.fill_(float("nan"))
            status = getattr(L, f"hgemm_mi355x_{entry}")(av.data_ptr(), bv.data_ptr(), btv.data_ptr(), c.data_ptr(), m, n, k, stream)
            torch.cuda.synchronize()
            got = c[ridx].double()
            finite = bool(torch.isfinite(c).all().item())          # (whole tile: an unwritten NaN anywhere is a failure)
            rel = float(((got - ref).abs().max() / max(scale, 1e-30)).item()) if finite else float("inf")
            ok = status == 0 and finite and rel <= a.rel_tol
            rec = {"mnk": f"{m}_{n}_{k}", "run": entry, "status": status, "pass": ok, "relative_error": rel, "tolerance": a.rel_tol,
                   "rows_checked": int(len(rows)), "reference": where, "inputs": "N(0,1)",
                   "plan": {"config": name.decode() if name else ("ragged" if cfg.value == -2 else "generic"), "splits": sp.value & 0xFFFF,
                            "fused": bool(sp.value & 0x10000), "nt_store": bool(sp.value & 0x20000), "streamk": bool(sp.value & 0x40000),
                            "rs_flags": (sp.value >> 19) & 3, "plan_flags": sp.value >> 19, "group_m": gm.value}}
            out_f.write(json.dumps(rec) + "\n")
            n_checks += 1
            n_fail += 0 if ok else 1
            worst = max(worst, rel)
            if not ok:
                print("FAIL", json.dumps(rec), flush=True)
    out_f.close()
    print(json.dumps({"checks": n_checks, "failures": n_fail, "worst_relative_error": worst, "seconds": round(time.time() - t_start, 1), "out": a.out}))
    return 1 if n_fail else 0


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--shape-file", default=str(REPO / "cuda-l2_amd" / "tools" / "grid_shapes.txt"))
    ap.add_argument("--shapes", default="", help="comma separated M_N_K (overrides --shape-file)")
    ap.add_argument("--plans", default="", help="tuner result jsonl: verify its candidates with explicit plans")
    ap.add_argument("--top", type=int, default=3)
    ap.add_argument("--out", required=True)
    ap.add_argument("--repeats", type=int, default=2, help="runs per plan (split-K counters must return to zero)")
    ap.add_argument("--seed", type=int, default=20260925)
    ap.add_argument("--randn", action="store_true", help="N(0,1) operands, relative tolerance on sampled rows (see the module text)")
    ap.add_argument("--rel-tol", type=float, default=1e-3)
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("verify_plans needs an MI355X")
    L = gpu_common.lib()
    stream = gpu_common.stream()

    if a.shapes:
        shapes = [tuple(int(x) for x in s.split("_")) for s in a.shapes.split(",") if s]
    else:
        shapes = [tuple(int(x) for x in ln.split("_")) for ln in (raw.strip() for raw in Path(a.shape_file).read_text().splitlines())
                  if ln and not ln.startswith("#")]
    cand = {}
    if a.plans:
        for ln in Path(a.plans).read_text().splitlines():
            r = json.loads(ln)
            cand[tuple(int(x) for x in r["mnk"].split("_"))] = sorted(r["candidates"], key=lambda c: c["us"])[:a.top]
        shapes = [s for s in shapes if s in cand]

    L.hgemm_mi355x_config_by_name.argtypes = [ctypes.c_char_p]
    if a.randn:
        return randn_pass(a, L, stream, shapes)
    rng = np.random.default_rng(a.seed)
    out_f = open(a.out, "w")
    t_start = time.time()
    n_checks = n_fail = 0
    for sparse in (False, True):
        group = [s for s in shapes if (max(s) > 8192) == sparse]
        if not group:
            continue
        dm, dn, dk = (max(s[i] for s in group) for i in range(3))
        a_np, b_np = oracle.zero_one_inputs(dm, dn, dk, rng, force_sparse=sparse)
        a_full = torch.from_numpy(a_np).cuda()
        b_full = torch.from_numpy(b_np).cuda()
        ks = sorted({s[2] for s in group})
        t0 = time.time()
        for k, truth_np in oracle.truth_prefix_k(a_np, b_np, ks):
            truth_full = torch.from_numpy(truth_np).cuda()
            cpu_s = time.time() - t0
            for (m, n, kk) in [s for s in group if s[2] == k]:
                ga, gb, gbt, gc = Guarded(m, k), Guarded(k, n), Guarded(n, k), Guarded(m, n)
                ga.view.copy_(a_full[:m, :k])
                gb.view.copy_(b_full[:k, :n])
                gbt.view.copy_(b_full[:k, :n].t())                       # as_col_major storage (tools/utils.py:110-115)
                truth = truth_full[:m, :n]
                runs = []
                if a.plans:
                    for c in cand[(m, n, k)]:
                        cid = L.hgemm_mi355x_config_by_name(c["config"].encode())
                        runs.append((f"{c['config']}/s{c['splits']}/g{c['group_m']}",
                                     lambda cid=cid, c=c: L.hgemm_mi355x_launch(cid, c["splits"], c["group_m"], ga.ptr(), gb.ptr(), gbt.ptr(),
                                                                                gc.ptr(), m, n, k, k, k, n, stream), c))
                else:
                    for entry in ("fp32", "fp16"):
                        fn = getattr(L, f"hgemm_mi355x_{entry}")
                        runs.append((entry, lambda fn=fn: fn(ga.ptr(), gb.ptr(), gbt.ptr(), gc.ptr(), m, n, k, stream), None))
                cfg, sp, gm = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                L.hgemm_mi355x_plan(m, n, k, ctypes.byref(cfg), ctypes.byref(sp), ctypes.byref(gm))
                name = L.hgemm_mi355x_config_name(cfg.value)
                for label, call, c in runs:
                    diffs, exact, status = [], True, 0
                    for _ in range(a.repeats):
                        gc.view.fill_(float("nan"))
                        status = call()
                        torch.cuda.synchronize()
                        if status != 0:
                            break
                        d, e = check_output(gc, truth)
                        diffs.append(d)
                        exact = exact and e
                    bars = all(g.bars_intact() for g in (ga, gb, gbt, gc))
                    inputs_ok = bool(torch.equal(ga.view, a_full[:m, :k]) and torch.equal(gbt.view, b_full[:k, :n].t()))
                    ok = status == 0 and oracle.check_passes(diffs) and exact and bars and inputs_ok
                    rec = {"mnk": f"{m}_{n}_{k}", "run": label, "status": status, "pass": ok, "max_diff_masked": max(diffs) if diffs else None,
                           "bitwise_equal_unmasked": exact, "guard_bars_intact": bars, "inputs_unchanged": inputs_ok,
                           "inputs": "{0,0,1}" if sparse else "{0,1}", "repeats": len(diffs)}
                    if c is None:
                        rec["plan"] = {"config": name.decode() if name else ("ragged" if cfg.value == -2 else "generic"),
                                       "splits": sp.value & 0xFFFF, "fused": bool(sp.value & 0x10000), "nt_store": bool(sp.value & 0x20000),
                                       "streamk": bool(sp.value & 0x40000), "rs_flags": (sp.value >> 19) & 3, "plan_flags": sp.value >> 19, "group_m": gm.value}
                    else:
                        rec["plan"] = {"config": c["config"], "splits": c["splits"], "group_m": c["group_m"]}
                    out_f.write(json.dumps(rec) + "\n")
                    n_checks += 1
                    n_fail += 0 if ok else 1
                    if not ok:
                        print("FAIL", json.dumps(rec), flush=True)
            out_f.flush()
            print(f"[verify] {'sparse' if sparse else 'dense'} K={k}: oracle {cpu_s:.1f}s, total {time.time() - t_start:.0f}s, "
                  f"{n_checks} checks, {n_fail} failures", flush=True)
            t0 = time.time()
        del a_full, b_full
    out_f.close()
    print(json.dumps({"checks": n_checks, "failures": n_fail, "seconds": round(time.time() - t_start, 1), "out": a.out}))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
