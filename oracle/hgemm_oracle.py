"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the CUDA-L2 HGEMM hot path (numpy + ctypes over
hgemm_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (cuda-l2_amd/) never does.

Restated rules (reference file:line):
  truth_f32acc          zero_one_correctness_check.py:85-90   (a.float() @ b.float()).half()
  zero_one_values       zero_one_correctness_check.py:65-73   {0,1} / {0,0,1} beyond 8192
  mask / max diff       zero_one_correctness_check.py:92,167-172
  guard bars            zero_one_correctness_check.py:98-150  (16384 elements either side)
  pass rule             zero_one_correctness_check.py:263-268 (average max-diff == 0.0 exactly)
  as_col_major          tools/utils.py:110-115
  padding               tools/utils.py:8-36, benchmarking_offline.py:102-113
  TFLOPS                benchmarking_utils.py:66              2*m*n*k*1e-12*1000/ms (unpadded sizes)
  -max row              summarize_result.py:43-53             lower cuda_l2 speedup of tn/nn
Parity pinning: see hgemm_oracle.c header and tests/golden/make_golden.py.
"""
from __future__ import annotations

import ctypes
import math
import re
import subprocess
from pathlib import Path

import numpy as np

ORACLE_DIR = Path(__file__).resolve().parent
BAR_SIZE = 16384
MAX_EXACT_FP16_INT = 2047.0

_lib = None


def build(force: bool = False) -> Path:
    """Compile hgemm_oracle.c with gcc (seconds); returns the .so path."""
    so = ORACLE_DIR / "libhgemm_oracle.so"
    src = ORACLE_DIR / "hgemm_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["gcc", "-O2", "-std=c99", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
        u16p = ctypes.POINTER(ctypes.c_uint16)
        for name in ("hgemm_oracle_f32acc", "hgemm_oracle_f32acc_tn", "hgemm_oracle_f16acc"):
            getattr(_lib, name).argtypes = [u16p, u16p, u16p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            getattr(_lib, name).restype = None
        _lib.hgemm_oracle_as_col_major.argtypes = [u16p, u16p, ctypes.c_int, ctypes.c_int]
        _lib.hgemm_oracle_masked_max_diff.argtypes = [u16p, u16p, ctypes.c_size_t]
        _lib.hgemm_oracle_masked_max_diff.restype = ctypes.c_float
        _lib.hgemm_oracle_half_to_float.argtypes = [ctypes.c_uint16]
        _lib.hgemm_oracle_half_to_float.restype = ctypes.c_float
        _lib.hgemm_oracle_float_to_half.argtypes = [ctypes.c_float]
        _lib.hgemm_oracle_float_to_half.restype = ctypes.c_uint16
    return _lib


def _u16(x: np.ndarray):
    assert x.dtype == np.float16 and x.flags["C_CONTIGUOUS"]
    return x.view(np.uint16).ctypes.data_as(ctypes.POINTER(ctypes.c_uint16))


def _gemm(name: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float16)
    b = np.ascontiguousarray(b, dtype=np.float16)
    m, k = a.shape
    k2, n = b.shape
    assert k == k2
    c = np.empty((m, n), dtype=np.float16)
    getattr(lib(), name)(_u16(a), _u16(b), _u16(c), m, n, k)
    return c


def truth_f32acc(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """C restatement: fp32 accumulate in k order, one RNE rounding to fp16."""
    return _gemm("hgemm_oracle_f32acc", a, b)


def truth_f32acc_tn(a: np.ndarray, bt: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float16)
    bt = np.ascontiguousarray(bt, dtype=np.float16)
    m, k = a.shape
    n, k2 = bt.shape
    assert k == k2
    c = np.empty((m, n), dtype=np.float16)
    lib().hgemm_oracle_f32acc_tn(_u16(a), _u16(bt), _u16(c), m, n, k)
    return c


def truth_f16acc(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    return _gemm("hgemm_oracle_f16acc", a, b)


def truth_numpy(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """numpy restatement (BLAS summation order): the fast oracle for large sizes."""
    return (a.astype(np.float32) @ b.astype(np.float32)).astype(np.float16)


def truth_prefix_k(a: np.ndarray, b: np.ndarray, ks):
    """Truths of ALL K-prefixes of one {0,1} problem, for sweeping many (M,N,K) shapes with one pair of
    operands: yields (K, truth_K) for K in ascending `ks`, truth_K = (a[:, :K].float() @ b[:K, :].float()).half()
    (zero_one_correctness_check.py:85-90), and the truth of the sub-problem (M, N, K) is truth_K[:M, :N].
    Only valid for 0/1 operands: every partial sum is an integer <= K < 2**24, exact in fp32 whatever the
    summation order, so accumulating the K-slices incrementally is bit-identical to the one-shot product
    (pinned in tests/test_oracle.py); total work 2*M*N*max(ks) instead of 2*M*N*sum(ks)."""
    assert a.dtype == np.float16 and b.dtype == np.float16
    ks = sorted(set(int(k) for k in ks))
    assert ks and ks[-1] <= a.shape[1] == b.shape[0] and ks[-1] < 2 ** 24
    acc = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    k0 = 0
    for k in ks:
        if k > k0:
            sa, sb = a[:, k0:k] , b[k0:k, :]
            assert ((sa == 0) | (sa == 1)).all() and ((sb == 0) | (sb == 1)).all(), "truth_prefix_k is exact for {0,1} inputs only"
            acc += sa.astype(np.float32) @ sb.astype(np.float32)
            k0 = k
        yield k, acc.astype(np.float16)


def as_col_major(b: np.ndarray) -> np.ndarray:
    """[K,N] array -> array of the SAME shape whose memory is b^T (= reference as_col_major)."""
    return np.ascontiguousarray(b.T).reshape(b.shape)


def zero_one_values(m: int, n: int, k: int) -> np.ndarray:
    return np.array([0.0, 1.0] if max(m, n, k) <= 8192 else [0.0, 0.0, 1.0], dtype=np.float16)


def zero_one_inputs(m: int, n: int, k: int, rng: np.random.Generator, force_sparse: bool | None = None):
    vals = zero_one_values(m, n, k) if force_sparse is None else np.array(
        [0.0, 0.0, 1.0] if force_sparse else [0.0, 1.0], dtype=np.float16)
    a = vals[rng.integers(0, len(vals), size=(m, k))]
    b = vals[rng.integers(0, len(vals), size=(k, n))]
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def masked_max_diff(out: np.ndarray, truth: np.ndarray) -> float:
    diff = np.abs(out.astype(np.float32) - truth.astype(np.float32))
    diff[np.abs(truth.astype(np.float32)) > MAX_EXACT_FP16_INT] = 0.0
    return float(diff.max()) if diff.size else 0.0


def check_passes(max_diffs) -> bool:
    """Pass rule of the reference: the AVERAGE of the per-iteration max diffs is exactly 0.0."""
    return len(max_diffs) > 0 and (sum(max_diffs) / len(max_diffs)) == 0.0


def relative_error(out: np.ndarray, ref_f32: np.ndarray) -> float:
    """max|out - ref| / max|ref| : the tolerance metric for N(0,1) inputs (BASELINE.json: 1e-3 rel with
    fp32 accumulate, 1e-2 rel with fp16 accumulate)."""
    ref = ref_f32.astype(np.float64)
    return float(np.max(np.abs(out.astype(np.float64) - ref)) / max(np.max(np.abs(ref)), 1e-30))


def extract_bm_bk_bn(text: str):
    found = {"BM": -1, "BN": -1, "BK": -1}
    for raw in text.split("\n"):
        m = re.search(r"(BM|BN|BK)\s*=\s*Int<(\d+)>", raw.strip().replace(" ", ""))
        if m:
            found[m.group(1)] = int(m.group(2))
    if min(found.values()) > 0:
        return found["BM"], found["BK"], found["BN"]
    return -1, -1, -1


def paddings(m: int, n: int, k: int, text: str):
    bm, bk, bn = extract_bm_bk_bn(text)
    if bm > 0:
        return (math.ceil(m / bm) * bm - m, math.ceil(k / bk) * bk - k, math.ceil(n / bn) * bn - n)
    return 0, 0, 0


def tflops(m: int, n: int, k: int, ms: float) -> float:
    return (2 * m * n * k) * 1e-12 * 1000 / ms


def max_row(tn: dict, nn: dict) -> dict:
    """summarize_result.py:43-53: the variant with the LOWER cuda_l2 speedup."""
    return tn if tn["Speedup"] < nn["Speedup"] else nn
