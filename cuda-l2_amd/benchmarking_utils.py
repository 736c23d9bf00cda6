"""The measurement itself (MI355X edition of the reference's benchmarking_utils.py).

Metric definition kept from the reference (benchmarking_utils.py:11-69):
  * one timed call = out.fill_(0); device sync; t0; f(a, b, b_col_major, out); device sync; t1
    (host wall-clock, so launch + sync overhead is part of the number);
  * TFLOPS = 2*m*n*k / t with the UNPADDED m, n, k even when the cuda_l2 kernel ran on padded tensors;
  * fresh N(0,1) fp16 A and B on every iteration, b_col_major built outside the timed region.
`device="cpu"` runs the same loop on host tensors (only torch.matmul can be timed there).
"""
import time

import torch

from tools.utils import as_col_major

torch.set_grad_enabled(False)


def _sync(device: str) -> None:
    if device != "cpu":
        torch.cuda.synchronize()


@torch.no_grad()
def run_benchmark(*, perf_func, a: torch.Tensor, b: torch.Tensor, b_col_major: torch.Tensor, out: torch.Tensor):
    """Time ONE call of perf_func; returns (out, elapsed_ms)."""
    device = "cpu" if out.device.type == "cpu" else "cuda"
    is_matmul = perf_func.__name__ == "matmul"
    out.fill_(0)
    _sync(device)
    t0 = time.time()
    if is_matmul:
        perf_func(a, b, out=out)
    else:
        perf_func(a, b, b_col_major, out)
    _sync(device)
    t1 = time.time()
    return out, (t1 - t0) * 1000.0


def _operands_for(func_name, cuda_l2_name, a, b, m, n, k, pads, device):
    pm, pk, pn = pads
    if func_name == cuda_l2_name and (pm or pk or pn):
        a_use = torch.zeros((m + pm, k + pk), dtype=torch.half, device=device)
        a_use[:m, :k] = a
        b_use = torch.zeros((k + pk, n + pn), dtype=torch.half, device=device)
        b_use[:k, :n] = b
        c_shape = (m + pm, n + pn)
    else:
        a_use, b_use, c_shape = a.clone(), b.clone(), (m, n)
    return a_use, b_use, as_col_major(b_use), torch.randn(c_shape, dtype=torch.half, device=device)


def run_all_perf_funcs_once(*, perf_func_list, m, n, k, acc_precise, device_type, padding_m, padding_k, padding_n,
                            device: str = "cuda"):
    """One benchmark iteration: fresh inputs, every function in perf_func_list timed once.

    Returns {name: TFLOPS, name_ms: milliseconds} for every function."""
    a = torch.randn((m, k), dtype=torch.half, device=device)
    b = torch.randn((k, n), dtype=torch.half, device=device)
    cuda_l2_name = f"cuda_l2_{device_type}_{acc_precise}"
    prepared = [
        _operands_for(f.__name__, cuda_l2_name, a, b, m, n, k, (padding_m, padding_k, padding_n), device)
        for f in perf_func_list
    ]
    _sync(device)
    record = {}
    for f, (a_use, b_use, b_cm, c_use) in zip(perf_func_list, prepared):
        _, ms = run_benchmark(perf_func=f, a=a_use, b=b_use, b_col_major=b_cm, out=c_use)
        record[f.__name__] = (2 * m * n * k) * 1e-12 * 1000 / ms
        record[f.__name__ + "_ms"] = ms
    return record
