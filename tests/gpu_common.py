"""Helpers for the GPU parity tests: the C ABI through ctypes, torch only for device memory."""
import ctypes
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent.parent
PKG = REPO / "cuda-l2_amd"
for p in (str(REPO), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)

import bench  # noqa: E402  (bench.load_library: the product's only way in, fails loudly if the .so is missing)

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = bench.load_library()
        _lib.hgemm_mi355x_launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    return _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def gemm(a_np: np.ndarray, b_np: np.ndarray, entry: str = "fp32", plan=None) -> np.ndarray:
    """C = A.B on the GPU through the C ABI; plan = (config_id, splits, group_m) for an explicit launch."""
    L = lib()
    m, k = a_np.shape
    n = b_np.shape[1]
    a = torch.from_numpy(np.ascontiguousarray(a_np)).cuda()
    b = torch.from_numpy(np.ascontiguousarray(b_np)).cuda()
    bt = b.t().contiguous()
    c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")  # unwritten outputs stay NaN
    if plan is None:
        fn = L.hgemm_mi355x_fp16 if entry == "fp16" else L.hgemm_mi355x_fp32
        st = fn(a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, stream())
    else:
        st = L.hgemm_mi355x_launch(plan[0], plan[1], plan[2], a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n,
                                   k, k, k, n, stream())
    assert st == 0, L.hgemm_mi355x_strerror(st)
    torch.cuda.synchronize()
    return c.cpu().numpy()


def config_names():
    L = lib()
    return [L.hgemm_mi355x_config_name(i).decode() for i in range(L.hgemm_mi355x_num_configs())]
