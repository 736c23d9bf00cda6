"""Pieces shared by the harness scripts (correctness check, offline / server benchmark, summary).

The reference repeats this logic at module level in each script
(benchmarking_offline.py:20-49, benchmarking_server.py:21-50, zero_one_correctness_check.py:18-45);
here it is one module so the scripts stay importable (tests import them without a GPU).
"""
from __future__ import annotations

import argparse
import math
import os
import random
from dataclasses import dataclass

import numpy as np
import torch

from tools.utils import DEVICE_TYPES, build_from_sources, compute_padding, kernel_source_path, kernels_dir_name

RESULT_VERSION = "202511261845"  # schema tag of benchmark_result_*.json (reference benchmarking_offline.py:162)

BASELINE_PERF_FUNCS = [
    "hgemm_cublas_tn",
    "hgemm_cublas_nn",
    "hgemm_cublaslt_heuristic_tn",
    "hgemm_cublaslt_heuristic_nn",
    "hgemm_cublaslt_auto_tuning_tn",
    "hgemm_cublaslt_auto_tuning_nn",
    "matmul",
]


def add_common_args(parser: argparse.ArgumentParser, *, benchmark: bool) -> None:
    parser.add_argument("--mnk", type=str, required=True)
    parser.add_argument("--acc_precise", type=str, required=True, choices=["fp16", "fp32"])
    parser.add_argument("--device_type", type=str, required=True, choices=DEVICE_TYPES)
    parser.add_argument("--base_dir", type=str, required=True)
    parser.add_argument("--gpu_device_id", type=int, required=True)
    parser.add_argument("--seed", type=int, default=None, help="seed torch/numpy/random (reference: unseeded)")
    if benchmark:
        parser.add_argument("--warmup_seconds", type=float, required=True)
        parser.add_argument("--benchmark_seconds", type=float, required=True)
        parser.add_argument("--perf_func", type=str, required=True, choices=BASELINE_PERF_FUNCS)
        parser.add_argument("--device", type=str, default="cuda", choices=["cuda", "cpu"],
                            help="cpu = plumbing run: torch.matmul on host tensors, no extension, no GPU")


def seed_everything(seed) -> None:
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def parse_mnk(mnk: str) -> tuple[int, int, int]:
    m, n, k = (int(x) for x in mnk.split("_"))
    if min(m, n, k) <= 0:
        raise ValueError(f"bad --mnk {mnk!r}")
    return m, n, k


@dataclass
class LoadedKernel:
    module: object          # the hgemm_lib extension (None on the CPU plumbing path)
    cuda_l2_func: object    # callable (a, b, b_col_major, c)
    cuda_l2_func_name: str  # "cuda_l2_<device_type>_<acc>"
    padding: tuple[int, int, int]  # (padding_m, padding_k, padding_n)


def cuda_l2_name(device_type: str, acc_precise: str) -> str:
    kernels_dir_name(acc_precise)  # validates
    return f"cuda_l2_{device_type}_{acc_precise}"


def load_kernel(mnk: str, acc_precise: str, device_type: str, base_dir: str, verbose: bool = False) -> LoadedKernel:
    """Build/import the shape's extension and derive the harness-side padding from its kernel file."""
    module = build_from_sources(mnk=mnk, acc_precise=acc_precise, device_type=device_type, base_dir=base_dir,
                                verbose=verbose)
    name = cuda_l2_name(device_type, acc_precise)
    func = getattr(module, name)
    m, n, k = parse_mnk(mnk)
    code_text = kernel_source_path(mnk, acc_precise, device_type).read_text()
    return LoadedKernel(module, func, name, compute_padding(m, n, k, code_text))


def init_baselines(hgemm) -> None:
    hgemm.init_cublas_handle()
    hgemm.init_cublaslt_handle_v1()
    hgemm.init_cublaslt_handle_v2()
    torch.cuda.synchronize()


def destroy_baselines(hgemm) -> None:
    hgemm.destroy_cublas_handle()
    hgemm.destroy_cublaslt_handle_v1()
    hgemm.destroy_cublaslt_handle_v2()
    torch.cuda.synchronize()


def insitu_report(m: int, n: int, k: int) -> dict | None:
    """What the library's first-use plan selection did for (m, n, k) in THIS process, or None when it is off
    (HGEMM_MI355X_INSITU unset / 0).  The reference's counterpart is the first-call autotune inside a kernel file
    (kernels/h100_F32F16F16F32/64_4096_64.cu:623-690,702-721), which reports nothing; here the benchmark JSON names the
    candidates and the choice so that a result can be traced to a plan."""
    import ctypes
    from pathlib import Path

    flag = os.environ.get("HGEMM_MI355X_INSITU", "")
    if not flag or flag[0] == "0":
        return None
    lib = ctypes.CDLL(str(Path(__file__).resolve().parent / "lib" / "libhgemm_mi355x.so"))   # already mapped by hgemm_lib
    lib.hgemm_mi355x_config_name.restype = ctypes.c_char_p
    cfg, sp, gm = (ctypes.c_int * 3)(), (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
    ncand = lib.hgemm_mi355x_insitu_candidates(m, n, k, cfg, sp, gm)

    def plan(c, s, g):
        name = lib.hgemm_mi355x_config_name(c)
        return {"config": name.decode() if name else str(c), "splits": int(s), "group_m": int(g)}

    c0, s0, g0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    chosen = lib.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0))
    return {"enabled": bool(lib.hgemm_mi355x_insitu_enabled()), "candidates": [plan(cfg[i], sp[i], gm[i]) for i in range(ncand)],
            "chosen": plan(c0.value, s0.value, g0.value) if chosen == 1 else None,
            "kept_table_plan": bool(chosen == 1 and (c0.value, s0.value, g0.value) == (cfg[0], sp[0], gm[0]))}


def percentile(values, q: float) -> float:
    if not len(values):
        return float("nan")
    return float(np.percentile(np.asarray(values, dtype=np.float64), q))


def ceil_to(x: int, m: int) -> int:
    return int(math.ceil(x / m) * m)


def cpu_cores() -> dict:
    return {"os_cpu_count": os.cpu_count(), "torch_num_threads": torch.get_num_threads()}
