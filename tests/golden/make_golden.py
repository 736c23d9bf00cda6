"""Generates tests/golden/hgemm_golden.npz + harness_golden.json from the REFERENCE itself.

Run once in the build container (needs /root/reference and the CPU torch of this image):
    python tests/golden/make_golden.py
The reference ships no golden vectors for this path; what pins parity is
  (1) its CPU oracle expression, executed here verbatim with torch on the CPU
      (zero_one_correctness_check.py:85-90:  torch.matmul(a.cpu().float(), b.cpu().float()).half()),
  (2) its own helper functions imported from /root/reference/tools/utils.py
      (as_col_major :110-115, extract_bm_bk_bn :8-36) applied to inputs written for this repo.
Only inputs we authored and the reference's OUTPUTS are stored; no reference source is copied.
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")


def load_reference_utils():
    spec = importlib.util.spec_from_file_location("reference_tools_utils", REF / "tools" / "utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_truth(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    # verbatim oracle expression of the reference (CPU, fp32 matmul, one rounding to fp16)
    return torch.matmul(a.cpu().float(), b.cpu().float()).half()


def main():
    ref = load_reference_utils()
    torch.manual_seed(20260925)
    out = {}
    cases = []
    # ---- 0/1 cases (the reference's correctness inputs), incl. ragged sizes and the mask case -----
    for name, (m, n, k), vals in [
        ("zo_64_96_128", (64, 96, 128), [0.0, 1.0]),
        ("zo_ragged_33_17_40", (33, 17, 40), [0.0, 1.0]),
        ("zo_sparse_24_40_192", (24, 40, 192), [0.0, 0.0, 1.0]),
        ("zo_tile_edge_130_260_64", (130, 260, 64), [0.0, 1.0]),
    ]:
        v = torch.tensor(vals, dtype=torch.half)
        a = v[torch.randint(0, len(vals), (m, k))].contiguous()
        b = v[torch.randint(0, len(vals), (k, n))].contiguous()
        out[name + "_a"], out[name + "_b"] = a.numpy(), b.numpy()
        out[name + "_truth"] = reference_truth(a, b).numpy()
        cases.append(name)
    # all-ones with K = 4096: every entry is 4096 > 2047 -> fully masked by the reference's rule
    a = torch.ones((4, 4096), dtype=torch.half)
    b = torch.ones((4096, 8), dtype=torch.half)
    out["zo_masked_4_8_4096_a"], out["zo_masked_4_8_4096_b"] = a.numpy(), b.numpy()
    out["zo_masked_4_8_4096_truth"] = reference_truth(a, b).numpy()
    cases.append("zo_masked_4_8_4096")
    # ---- N(0,1) cases (the benchmark's inputs, benchmarking_utils.py:36-37) -------------------------
    for name, (m, n, k) in [("randn_64_64_256", (64, 64, 256)), ("randn_48_80_512", (48, 80, 512))]:
        a = torch.randn((m, k)).half()
        b = torch.randn((k, n)).half()
        out[name + "_a"], out[name + "_b"] = a.numpy(), b.numpy()
        out[name + "_truth"] = reference_truth(a, b).numpy()
        out[name + "_f32"] = torch.matmul(a.float(), b.float()).numpy()
        cases.append(name)
    # ---- as_col_major (reference tools/utils.py:110-115) ------------------------------------------------
    for name, (k, n) in [("acm_3_5", (3, 5)), ("acm_64_16", (64, 16)), ("acm_1_7", (1, 7))]:
        x = torch.arange(k * n, dtype=torch.float32).reshape(k, n).half()
        y = ref.as_col_major(x)
        assert y.is_contiguous() and tuple(y.shape) == (k, n)
        out[name + "_x"], out[name + "_y"] = x.numpy(), y.numpy()
    np.savez_compressed(HERE / "hgemm_golden.npz", **out)

    # ---- tile-size regex (reference tools/utils.py:8-36) on kernel-text snippets written for this repo ---
    snippets = {
        "cute_style": "  using BM = Int<128>;\n  static constexpr auto BN = Int<160>{};\n  auto BK = Int<32>{};\n",
        "spaced": "BM   =   Int< 64 >;\nBN=Int<128>;\n  BK =Int<32>;",
        "last_wins": "BM = Int<64>;\nBM = Int<256>;\nBN = Int<64>;\nBK = Int<16>;",
        "missing_bk": "BM = Int<64>;\nBN = Int<64>;",
        "not_int_wrapper": "constexpr int BM = 128; constexpr int BN = 128; constexpr int BK = 64;",
        "mi355x_shape_file": "// plan: geometry t256x256_w2x4_m16_s2, split-K 1\nHGEMM_MI355X_SHAPE_ENTRY(4096, 4096, 4096, \"t256x256_w2x4_m16_s2\", 1, 8)\n",
        "two_on_one_line": "BM = Int<32>; BN = Int<48>;\nBK = Int<8>;",
    }
    regex = {name: list(ref.extract_bm_bk_bn(text)) for name, text in snippets.items()}
    (HERE / "harness_golden.json").write_text(json.dumps(
        {"cases": cases, "snippets": snippets, "extract_bm_bk_bn": regex,
         "generated_with": {"torch": torch.__version__, "reference": str(REF)}}, indent=1))
    print("wrote", HERE / "hgemm_golden.npz", (HERE / "hgemm_golden.npz").stat().st_size, "bytes;", regex)


if __name__ == "__main__":
    sys.exit(main())
