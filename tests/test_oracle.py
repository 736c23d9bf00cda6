"""Pins the CPU oracle (oracle/) against the fixtures generated from the reference
(tests/golden/make_golden.py) and checks the product's host-side helpers against the same fixtures."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import hgemm_oracle as oracle

GOLDEN = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN / "hgemm_golden.npz"), json.loads((GOLDEN / "harness_golden.json").read_text())


def _cases(prefix):
    meta = json.loads((GOLDEN / "harness_golden.json").read_text())
    return [c for c in meta["cases"] if c.startswith(prefix)]


@pytest.mark.parametrize("case", _cases("zo_"))
def test_zero_one_truth_is_bit_exact(gold, case):
    npz, _ = gold
    a, b, truth = npz[case + "_a"], npz[case + "_b"], npz[case + "_truth"]
    # integer partial sums: any summation order gives the same bits wherever |truth| <= 2047 ...
    for got in (oracle.truth_f32acc(a, b), oracle.truth_numpy(a, b), oracle.truth_f32acc_tn(a, np.ascontiguousarray(b.T))):
        assert oracle.masked_max_diff(got, truth) == 0.0
    # ... and here even beyond (fp32 holds integers up to 2^24 exactly, the fp16 rounding is shared)
    assert np.array_equal(oracle.truth_f32acc(a, b).view(np.uint16), truth.view(np.uint16))


def test_mask_rule_hides_values_above_2047(gold):
    npz, _ = gold
    truth = npz["zo_masked_4_8_4096_truth"]
    assert float(truth.min()) == 4096.0
    wrong = np.zeros_like(truth)                      # maximally wrong output
    assert oracle.masked_max_diff(wrong, truth) == 0.0  # ... is invisible: every entry is masked
    assert oracle.lib().hgemm_oracle_masked_max_diff(oracle._u16(wrong), oracle._u16(np.ascontiguousarray(truth)),
                                                     truth.size) == 0.0
    half = truth.copy()
    half[:] = 2047.0
    assert oracle.masked_max_diff(np.zeros_like(half), half) == 2047.0  # 2047 itself is NOT masked


@pytest.mark.parametrize("case", _cases("randn_"))
def test_randn_truth_within_one_ulp_and_tolerances(gold, case):
    npz, _ = gold
    a, b, truth, f32 = npz[case + "_a"], npz[case + "_b"], npz[case + "_truth"], npz[case + "_f32"]
    got = oracle.truth_f32acc(a, b)
    # fp32 summation order differs between the C loop and torch's BLAS, so a few results land on the
    # neighbouring fp16 value (more than one ulp only where the dot product cancels to ~0)
    assert (got.view(np.uint16) == truth.view(np.uint16)).mean() > 0.97
    assert oracle.relative_error(got, truth.astype(np.float32)) <= 2.0 ** -10
    # the tolerance contract of BASELINE.json: 1e-3 rel (fp32 accumulate), 1e-2 rel (fp16 accumulate)
    assert oracle.relative_error(got, f32) <= 1e-3
    assert oracle.relative_error(oracle.truth_f16acc(a, b), f32) <= 1e-2
    assert oracle.relative_error(oracle.truth_f16acc(a, b), f32) > oracle.relative_error(got, f32)


@pytest.mark.parametrize("name", ["acm_3_5", "acm_64_16", "acm_1_7"])
def test_as_col_major_matches_reference(gold, name):
    npz, _ = gold
    x, y = npz[name + "_x"], npz[name + "_y"]
    assert np.array_equal(oracle.as_col_major(x), y)
    from tools.utils import as_col_major  # the product's torch implementation

    got = as_col_major(torch.from_numpy(x))
    assert got.is_contiguous() and tuple(got.shape) == x.shape
    assert np.array_equal(got.numpy(), y)
    k, n = x.shape  # the memory of the result is x^T
    assert np.array_equal(got.numpy().reshape(-1), np.ascontiguousarray(x.T).reshape(-1))
    bt = np.empty((n, k), dtype=np.float16)
    oracle.lib().hgemm_oracle_as_col_major(oracle._u16(np.ascontiguousarray(x)), oracle._u16(bt), k, n)
    assert np.array_equal(bt.reshape(-1), y.reshape(-1))


def test_tile_regex_and_padding_match_reference(gold):
    _, meta = gold
    from tools.utils import compute_padding, extract_bm_bk_bn

    for name, text in meta["snippets"].items():
        want = tuple(meta["extract_bm_bk_bn"][name])
        assert oracle.extract_bm_bk_bn(text) == want, name
        assert extract_bm_bk_bn(text) == want, name
    text = meta["snippets"]["cute_style"]  # BM=128, BK=32, BN=160
    assert oracle.paddings(4096, 4096, 4096, text) == (0, 0, 64)
    assert compute_padding(4096, 4096, 4096, text) == (0, 0, 64)
    assert compute_padding(64, 4096, 64, text) == (64, 0, 64)
    assert compute_padding(64, 4096, 64, meta["snippets"]["mi355x_shape_file"]) == (0, 0, 0)


def test_pass_rule_and_generator():
    assert oracle.check_passes([0.0, 0.0]) and not oracle.check_passes([0.0, 1.0]) and not oracle.check_passes([])
    assert list(oracle.zero_one_values(64, 4096, 8192)) == [0.0, 1.0]
    assert list(oracle.zero_one_values(64, 12288, 64)) == [0.0, 0.0, 1.0]
    rng = np.random.default_rng(3)
    a, b = oracle.zero_one_inputs(16, 16, 512, rng, force_sparse=True)
    assert set(np.unique(a)) <= {0.0, 1.0} and 0.2 < a.mean() < 0.45
    assert oracle.tflops(4096, 4096, 4096, 0.1) == pytest.approx(1374.39, rel=1e-4)


def test_half_conversions_are_exact():
    L = oracle.lib()
    for h in range(0, 65536, 13):
        f = L.hgemm_oracle_half_to_float(h)
        e = float(np.array([h], dtype=np.uint16).view(np.float16)[0])
        assert f == e or (f != f and e != e)
    rng = np.random.default_rng(0)
    vals = (rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 5, 4000)).astype(np.float32)
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([L.hgemm_oracle_float_to_half(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, want)


def test_prefix_k_truths_are_bit_identical_to_the_one_shot_oracle():
    """tests/tools/verify_plans.py sweeps the 1000-shape grid with one operand pair per input class and the
    incremental K-prefix truths; for 0/1 inputs they must equal the C restatement bit for bit."""
    rng = np.random.default_rng(3)
    for sparse in (False, True):
        a, b = oracle.zero_one_inputs(96, 80, 640, rng, force_sparse=sparse)
        ks = [64, 128, 320, 640]
        seen = []
        for k, t in oracle.truth_prefix_k(a, b, ks):
            seen.append(k)
            for (m, n) in [(96, 80), (33, 17), (64, 64)]:
                ref = oracle.truth_f32acc(np.ascontiguousarray(a[:m, :k]), np.ascontiguousarray(b[:k, :n]))
                assert np.array_equal(t[:m, :n].view(np.uint16), ref.view(np.uint16))
        assert seen == ks
    with pytest.raises(AssertionError):
        list(oracle.truth_prefix_k(np.full((4, 64), 0.5, np.float16), np.ones((64, 4), np.float16), [64]))
