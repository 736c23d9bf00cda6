"""Exact 0/1-matrix correctness check of cuda_l2 (and the baselines) against a CPU fp32 matmul.

The rule is the reference's (zero_one_correctness_check.py:48-271) and is restated for tests in
oracle/hgemm_oracle.py:
  * A, B are drawn from {0,1} (P(1)=1/2), or {0,0,1} (P(1)=1/3) when max(M,N,K) > 8192, so every
    partial sum is a small integer;
  * truth = (A.float() @ B.float()).half() on the CPU; entries with |truth| > 2047 are masked
    (integers above 2048 are not exactly representable in fp16);
  * per iteration the max |out - truth| over unmasked entries is recorded; cuda_l2 passes iff the
    average over iterations is EXACTLY 0 (bit-exact);
  * cuda_l2's operands live inside flat buffers with 16384-element random guard bars on both sides
    of A, B, B_col_major and C; any change to a bar = "memory overflow detected".
Up to 100 iterations or 60 s.  Writes {base_dir}/zero_one_correctness_check_result.json.
Unlike the reference (which prints FAILED but exits 0), a failed check also exits with status 1.
"""
import argparse
import json
import os
import sys
import time
import traceback
from pathlib import Path

import torch

from harness_common import (add_common_args, destroy_baselines, init_baselines, load_kernel, parse_mnk,
                            seed_everything)
from tools.utils import as_col_major

BAR_SIZE = 16384          # guard elements either side of every cuda_l2 operand
MAX_EXACT_FP16_INT = 2047.0
MAX_ITERATIONS = 100
MAX_SECONDS = 60


def zero_one_values(m: int, n: int, k: int, device) -> torch.Tensor:
    """{0,1} up to 8192, {0,0,1} beyond: keeps E[dot] = K/4 resp. K/9 below 2048."""
    vals = [0.0, 1.0] if max(m, n, k) <= 8192 else [0.0, 0.0, 1.0]
    return torch.tensor(vals, dtype=torch.half, device=device)


class GuardedOperand:
    """A [rows, cols] fp16 view in the middle of a flat random buffer, with a pristine copy of the bars."""

    def __init__(self, rows: int, cols: int, device):
        n = rows * cols
        self.flat = torch.randn(n + 2 * BAR_SIZE, dtype=torch.half, device=device)
        self.head = self.flat[:BAR_SIZE].clone()
        self.tail = self.flat[-BAR_SIZE:].clone()
        self.view = self.flat[BAR_SIZE:BAR_SIZE + n].view(rows, cols)
        self.view.fill_(0.0)

    def bars_intact(self) -> bool:
        return bool(torch.equal(self.flat[:BAR_SIZE], self.head) and torch.equal(self.flat[-BAR_SIZE:], self.tail))


@torch.no_grad()
def run_cuda_l2_guarded(func, a, b, m, n, k, pads):
    """Call cuda_l2 on padded, guard-barred operands; returns (out[:m,:n] on CPU, bars_ok)."""
    pm, pk, pn = pads
    ga = GuardedOperand(m + pm, k + pk, a.device)
    gb = GuardedOperand(k + pk, n + pn, a.device)
    gbt = GuardedOperand(k + pk, n + pn, a.device)
    gc = GuardedOperand(m + pm, n + pn, a.device)
    ga.view[:m, :k] = a
    gb.view[:k, :n] = b
    gbt.view.copy_(as_col_major(gb.view))
    for g in (ga, gb, gbt, gc):
        assert g.view.is_contiguous()
    torch.cuda.synchronize()
    func(ga.view, gb.view, gbt.view, gc.view)
    torch.cuda.synchronize()
    ok = all(g.bars_intact() for g in (ga, gb, gbt, gc))
    return gc.view[:m, :n].cpu(), ok


@torch.no_grad()
def compare_kernels_with_cpu_fp32(kernel_funcs, cuda_l2_func_name, m, n, k, num_iterations, pads, max_seconds=MAX_SECONDS):
    diffs = {f.__name__: [] for f in kernel_funcs}
    values = zero_one_values(m, n, k, "cuda")
    no_overflow = True
    start = time.time()
    done = 0
    for _ in range(num_iterations):
        if time.time() - start > max_seconds:
            break
        a = values[torch.randint(0, len(values), (m, k), device="cuda")].contiguous()
        b = values[torch.randint(0, len(values), (k, n), device="cuda")].contiguous()
        torch.cuda.synchronize()
        truth = torch.matmul(a.cpu().float(), b.cpu().float()).half()   # the CPU oracle
        mask = truth.abs() > MAX_EXACT_FP16_INT
        for func in kernel_funcs:
            tag = func.__name__
            if tag == cuda_l2_func_name:
                out, ok = run_cuda_l2_guarded(func, a, b, m, n, k, pads)
                no_overflow = no_overflow and ok
            else:
                a_use, b_use = a.clone(), b.clone()
                out_dev = torch.zeros((m, n), dtype=torch.half, device="cuda")
                torch.cuda.synchronize()
                if tag == "matmul":
                    torch.matmul(a_use, b_use, out=out_dev)
                else:
                    func(a_use, b_use, as_col_major(b_use), out_dev)
                torch.cuda.synchronize()
                out = out_dev.cpu()
            diff = (out - truth).abs()
            diff[mask] = 0.0
            diffs[tag].append(diff.max().item())
        done += 1
    result = {"if_success": True, "m": m, "n": n, "k": k, "num_iterations": done}
    avg = {tag: sum(v) / max(1, len(v)) for tag, v in diffs.items()}
    for tag, v in avg.items():
        result[f"avg_{tag}_diff"] = round(v, 6)
    result["best_kernel"] = min(avg, key=avg.get)
    return result, no_overflow


def judge(result: dict, cuda_l2_func_name: str, no_overflow: bool):
    """-> (success, message) from the averaged diffs; the pass bar is avg diff == 0.0 exactly."""
    if not no_overflow:
        return False, "memory overflow detected."
    key = f"avg_{cuda_l2_func_name}_diff"

    def finite(v):
        return isinstance(v, (int, float)) and v == v and v not in (float("inf"), float("-inf"))

    others = [v for kk, v in result.items() if kk.startswith("avg_") and kk.endswith("_diff") and kk != key and finite(v)]
    if key not in result or not others:
        raise RuntimeError("no comparison data available for correctness check.")
    mine = result[key]
    if not finite(mine):
        return False, f"{cuda_l2_func_name} has nan or Inf value: {mine}"
    here = Path(__file__).resolve().parent
    if mine > 0.0:
        return False, (f"{cuda_l2_func_name} diff ({mine:.6f}) exceeds 0 (max_other: {max(others):.6f}), "
                       f"see {here} for details.")
    return True, f"Precise Correctness check passed: v2_diff={mine:.6f}, max_other={max(others):.6f}, see {here} for details."


@torch.no_grad()
def run_correctness_check(kernel, m, n, k, num_iterations=MAX_ITERATIONS, max_seconds=MAX_SECONDS):
    hgemm = kernel.module
    init_baselines(hgemm)
    hgemm.find_best_algo_tn_v2_torch(m, n, k)
    hgemm.find_best_algo_nn_v2_torch(m, n, k)
    print("Initialize Done.")
    kernel_funcs = [hgemm.hgemm_cublas_tn, hgemm.hgemm_cublas_nn, hgemm.hgemm_cublaslt_heuristic_tn,
                    hgemm.hgemm_cublaslt_heuristic_nn, hgemm.hgemm_cublaslt_auto_tuning_tn,
                    hgemm.hgemm_cublaslt_auto_tuning_nn, torch.matmul, kernel.cuda_l2_func]
    try:
        result, no_overflow = compare_kernels_with_cpu_fp32(kernel_funcs, kernel.cuda_l2_func_name, m, n, k,
                                                            num_iterations, kernel.padding, max_seconds)
    except Exception as exc:  # harness convention: exceptions become success=False JSON
        traceback.print_exc()
        return False, str(exc), {}
    finally:
        destroy_baselines(hgemm)
    print(result)
    success, message = judge(result, kernel.cuda_l2_func_name, no_overflow)
    return success, message, result


def main(argv=None) -> int:
    print("======================Correctness Check======================")
    parser = argparse.ArgumentParser(description=__doc__)
    add_common_args(parser, benchmark=False)
    parser.add_argument("--max_seconds", type=float, default=MAX_SECONDS)
    args = parser.parse_args(argv)
    torch.set_grad_enabled(False)
    seed_everything(args.seed)
    m, n, k = parse_mnk(args.mnk)
    torch.cuda.set_device(args.gpu_device_id)
    t0 = time.time()
    kernel = load_kernel(args.mnk, args.acc_precise, args.device_type, args.base_dir)
    print(f"Load hgemm module time: {time.time() - t0:.2f} seconds")
    print(f"Running correctness check for m={m}, n={n}, k={k} ...")
    print("Padding: padding_m={}, padding_k={}, padding_n={}".format(*kernel.padding))
    success, message, result = run_correctness_check(kernel, m, n, k, max_seconds=args.max_seconds)
    os.makedirs(args.base_dir, exist_ok=True)
    with open(Path(args.base_dir) / "zero_one_correctness_check_result.json", "w") as f:
        json.dump({"success": success, "message": message, "result": result}, f, indent=4, ensure_ascii=False)
    print("Correctness Check PASSED:" if success else "Correctness Check FAILED:", message)
    return 0 if success else 1


if __name__ == "__main__":
    sys.exit(main())
