"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI of
libhgemm_mi355x.so, against the CPU oracle on the same seeded inputs.

Bar: bit-exact for the reference's {0,1}/{0,0,1} inputs wherever |truth| <= 2047 (the reference's own
pass rule, zero_one_correctness_check.py:263-268); max|C-ref|/max|ref| <= 1e-3 for N(0,1) inputs in
BOTH accumulate modes (BASELINE.json allows 1e-2 for fp16-acc; CDNA4 accumulates in fp32 either way)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = Path(__file__).resolve().parent / "golden"
PKG_DIR = Path(__file__).resolve().parent.parent / "cuda-l2_amd"
REL_TOL = 1e-3


@pytest.fixture(scope="module")
def g():
    import gpu_common

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (torch.cuda.is_available() is False)")
    return gpu_common


@pytest.fixture(scope="module")
def oracle():
    from oracle import hgemm_oracle

    return hgemm_oracle


def _golden_cases():
    return json.loads((GOLDEN / "harness_golden.json").read_text())["cases"]


@pytest.mark.parametrize("entry", ["fp32", "fp16"])
@pytest.mark.parametrize("case", _golden_cases())
def test_golden_fixtures(g, oracle, case, entry):
    npz = np.load(GOLDEN / "hgemm_golden.npz")
    a, b, truth = npz[case + "_a"], npz[case + "_b"], npz[case + "_truth"]
    got = g.gemm(a, b, entry)
    if case.startswith("zo_"):
        assert oracle.masked_max_diff(got, truth) == 0.0
    else:
        assert oracle.relative_error(got, npz[case + "_f32"]) <= REL_TOL


@pytest.mark.parametrize("shape", [(1, 64, 64), (7, 12, 64), (64, 64, 64), (200, 136, 128), (320, 448, 512),
                                   (1000, 520, 192), (33, 17, 40), (65, 30, 100), (5, 4, 8)])
def test_ragged_and_unaligned_shapes_match_oracle(g, oracle, shape):
    """Edge tiles are predicated in-kernel (no harness padding); K % 64 != 0 or N % 4 != 0 take the
    register-staged MFMA kernel (hgemm_kernel_rg.hpp), which pads on the way into LDS."""
    m, n, k = shape
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_f32acc(a, b)
    for entry in ("fp32", "fp16"):
        got = g.gemm(a, b, entry)
        assert not np.isnan(got).any(), "an output element was never written"
        assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))


def test_every_geometry_and_split_k_is_exact(g, oracle):
    """All kernel instantiations x split-K factors on one ragged shape, explicit plans."""
    m, n, k = 328, 456, 1024
    rng = np.random.default_rng(11)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    TWO_PASS = 0x10000   # HGEMM_SPLITK_FUSED: single-launch split-K; plain counts = slabs + combine kernel
    SK = 0x40000         # HGEMM_PLAN_STREAMK | persistent workgroups (0: one resident wave); ignored by families without the kernel
    for cid, name in enumerate(g.config_names()):
        for splits, group in [(1, 1), (1, 3), (2, 1), (5, 1), (16, 2), (2 | TWO_PASS, 1), (5 | TWO_PASS, 1), (16 | TWO_PASS, 2),
                              (SK, 1), (SK | 7, 3), (SK | 31, 1), (SK | 90, 2)]:
            for _ in range(2 if splits > 1 else 1):   # twice: the per-tile arrival counters must be back at zero
                got = g.gemm(a, b, plan=(cid, splits, group))
                assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), (name, splits, group)


STREAMK_CASES = [  # (config, workgroups, group, (M, N, K)): grids that cut tiles at odd stages, several segments per run, ragged edges
    ("r128x128_k128", 256, 4, (1536, 128, 8192)),       # 12 tiles x 64 stages on 256 workgroups: every tile has ~21 parts
    ("r128x128_k128", 37, 1, (1100, 300, 2048)),        # ragged M and N edges, 27 tiles x 16 stages over 37 workgroups
    ("r64x64_k256", 512, 8, (2048, 64, 8192)),          # two workgroups per CU, BKS = 256
    ("r96x128_k128", 200, 2, (1536, 256, 4096)),        # 96-row member
    ("r64x128_k128", 0, 4, (192, 4000, 2048)),          # default grid, ragged N
    ("t128x64_w4x2_m16_s4", 256, 4, (512, 4096, 4096)), # 256 tiles on 256 workgroups: nothing is cut (persistent data-parallel walk)
    ("t128x64_w4x2_m16_s4", 200, 4, (512, 4096, 4096)), # ... 1.28 tiles per workgroup
    ("t128x128_w2x2_m16_s3", 256, 8, (3072, 3072, 1024)),  # 576 tiles x 16 stages: 36 stages per workgroup = 2.25 tiles
    ("t64x128_w2x4_m16_s3", 512, 2, (1000, 1096, 2056)),   # ragged edges and a partial last stage (K % 64 = 8)
    ("t256x128_w4x2_m16_s2", 100, 2, (2048, 2048, 832)),   # 128 tiles x 13 stages over 100 workgroups
    ("t32x32_w1x1_m16_s4", 1024, 4, (96, 160, 16384)),     # 15 tiny tiles x 256 stages: every tile has ~68 parts
    ("t128x128_w2x2_m32_s2", 300, 1, (1152, 1152, 4096)),  # 32x32x16 MFMA accumulators through the slabs
]


@pytest.mark.parametrize("case", STREAMK_CASES, ids=lambda c: f"{c[0]}-G{c[1]}-{'x'.join(map(str, c[3]))}")
def test_stream_k_is_exact_and_deterministic(g, oracle, case):
    """Stream-K (HGEMM_PLAN_STREAMK; the reference's H100 StreamKScheduler shapes, kernels/h100_F32F16F16F32/128_4096_16384.cu:79):
    0/1 inputs bit-exact against the oracle, twice (the arrival counters must return to zero), and N(0,1) inputs within
    tolerance and bit-identical over repeats (the parts of a tile are added in K order, whoever completes it)."""
    cfg, wgs, group, (m, n, k) = case
    L = g.lib()
    cid = g.config_names().index(cfg)
    assert L.hgemm_mi355x_config_streamk(cid) > 0
    rng = np.random.default_rng(m + 3 * n + 7 * k)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    for _ in range(2):
        got = g.gemm(a, b, plan=(cid, 0x40000 | wgs, group))
        assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), case
    ar = torch.randn((m, k), dtype=torch.half, device="cuda")
    br = torch.randn((k, n), dtype=torch.half, device="cuda")
    btr = br.t().contiguous()
    ref = ar.float() @ br.float()
    first = None
    for rep in range(12):
        c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")
        assert L.hgemm_mi355x_launch(cid, 0x40000 | wgs, group, ar.data_ptr(), br.data_ptr(), btr.data_ptr(), c.data_ptr(), m, n, k, k, k, n,
                                     g.stream()) == 0
        torch.cuda.synchronize()
        if first is None:
            first = c
            assert ((c.float() - ref).abs().max() / ref.abs().max()).item() <= REL_TOL, case
        else:
            assert torch.equal(first.view(torch.int16), c.view(torch.int16)), (case, rep)


def test_stream_k_without_workspace_or_kernel_degrades_to_the_plain_launch(g, oracle):
    """A stream-K plan on a family without the kernel, or with a lent workspace that is too small, runs the geometry's plain
    launch: same result, no error."""
    import ctypes

    L = g.lib()
    L.hgemm_mi355x_set_workspace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    names = g.config_names()
    m, n, k = 640, 768, 1024
    rng = np.random.default_rng(5)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    got = g.gemm(a, b, plan=(names.index("q256x256_w2x2"), 0x40000 | 256, 2))
    assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))
    small = torch.zeros(300 << 10, dtype=torch.uint8, device="cuda")          # counters + 44 KiB: no room for the slabs
    assert L.hgemm_mi355x_set_workspace(ctypes_ptr(small), small.numel()) == 0
    try:
        got = g.gemm(a, b, plan=(names.index("t128x128_w2x2_m16_s3"), 0x40000 | 64, 2))
        assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))
    finally:
        assert L.hgemm_mi355x_set_workspace(None, 0) == 0
    got = g.gemm(a, b, plan=(names.index("t128x128_w2x2_m16_s3"), 0x40000 | 64, 2))
    assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))


def ctypes_ptr(t):
    import ctypes

    return ctypes.c_void_p(t.data_ptr())


@pytest.mark.parametrize("k,hot", [(512, 300), (552, 540)])
def test_special_values_round_like_the_reference(g, k, hot):
    """The epilogue's fp32 -> fp16 conversion and the non-finite / denormal paths, which 0/1 inputs (values <= 2047, exact in every
    rounding mode) cannot see: the result must equal the reference's oracle expression (a.float() @ b.float()).half()
    (zero_one_correctness_check.py:85-90) -- round-to-nearest-even incl. overflow to inf at 65520, ties to even at the bottom
    of the denormal range, inf / NaN propagation (inf x 0, inf - inf), denormal operands and results -- for every kernel
    family, both split-K forms (the slabs carry inf / NaN / denormals in fp32) and stream-K.  Every row's sum is exact in fp32
    whatever the summation order.  Second case: K = 552 = 8 x 64 + 40 with the second hot column inside the K tail, so the
    "ktail" variants of families q and r (and the classic family's padded last step) carry inf / NaN / denormals through their tails."""
    m, n = 128, 192
    a = torch.zeros((m, k), dtype=torch.half)
    b = torch.zeros((k, n), dtype=torch.half)
    b[0, :] = 1.0; b[1, :] = 1.0; b[hot, :] = 1.0
    b[0, 5::7] = 0.0                                  # columns where an inf in A meets a zero in B
    b[2, :] = 2.0 ** -10
    inf = float("inf")
    a[0, 0] = 65504.0                                  # the largest finite value survives
    a[1, 1] = 65504.0; a[1, hot] = 16.0                # 65520: the tie between 65504 and 2^16 rounds to even = inf
    a[2, 0] = 65504.0; a[2, hot] = 8.0                 # 65512 rounds down to 65504
    a[3, 0] = -65504.0; a[3, 1] = -16.0                # -inf
    a[4, 0] = inf                                      # inf, and NaN where B has a zero
    a[5, 1] = inf; a[5, hot] = -inf                    # inf - inf = NaN
    a[6, hot] = float("nan")                           # NaN x anything
    a[7, 1] = 2.0 ** -24                               # a denormal operand, a denormal result
    a[8, 2] = 2.0 ** -14                               # 2^-14 x 2^-10 = 2^-24: a denormal result from normal operands
    a[9, 2] = 2.0 ** -15                               # 2^-25: the tie between 0 and 2^-24 rounds to even = 0
    a[10, 2] = 1.5 * 2.0 ** -15                        # 1.5 x 2^-25 rounds up to 2^-24
    a[11, 1] = 1.0; a[11, hot] = 2.0 ** -11            # 1 + 2^-11: tie, rounds to even = 1
    a[12, 1] = 1.0; a[12, hot] = 3 * 2.0 ** -11        # 1 + 3 x 2^-11: tie, rounds to even = 1 + 2^-9
    a[13, 1] = 2048.0; a[13, hot] = 1.0                # 2049 -> 2048 (the first integer fp16 cannot hold)
    a[14, 0] = 60000.0; a[14, 1] = 60000.0             # finite operands, overflowing sum
    a[64:, :] = a[:64, :].clone()                      # the same rows in the second 64-row band of every tile
    truth = (a.float() @ b.float()).half()
    assert torch.isinf(truth[1]).all() and truth[2, 0] == 65504 and torch.isnan(truth[4, 5]) and truth[4, 0] == inf
    assert truth[8, 0] == 2.0 ** -24 and truth[9, 0] == 0 and truth[10, 0] == 2.0 ** -24 and truth[11, 0] == 1 and truth[13, 0] == 2048
    L = g.lib()
    names = g.config_names()
    ad, bd = a.cuda(), b.cuda()
    btd = bd.t().contiguous()
    plans = [("entry fp32", None), ("entry fp16", None), ("ragged", (-2, 1, 1)), ("generic", (-1, 1, 1))]
    for cfg, splits in [("t64x64_w2x2_m16_s4", 1), ("t128x128_w2x2_m32_s2", 2), ("t128x128_w2x2_m16_s3", 2 | 0x10000), ("t64x64_w2x2_m16_s4", 0x40000 | 5),
                        ("s256x128_w2x2", 2), ("q128x128_w2x2", 1), ("q256x256_w2x2", 1 | 0x20000), ("q128x256_w2x2", 2 | 0x10000), ("q192x256_w2x2", 1),
                        ("q256x256_w2x2_m32", 1), ("r64x64_k256", 1), ("r128x64_k128", 2 | 0x10000), ("r128x128_k128", 0x40000 | 3),
                        ("q128x128_w2x2_k128", 2), ("r64x128_k128_d", 1 | 0x180000)]:
        plans.append((f"{cfg}/{splits:#x}", (names.index(cfg), splits, 1)))
    for label, plan in plans:
        c = torch.full((m, n), 7.0, dtype=torch.half, device="cuda")
        if plan is None:
            fn = L.hgemm_mi355x_fp16 if label.endswith("fp16") else L.hgemm_mi355x_fp32
            st = fn(ad.data_ptr(), bd.data_ptr(), btd.data_ptr(), c.data_ptr(), m, n, k, g.stream())
        else:
            st = L.hgemm_mi355x_launch(plan[0], plan[1], plan[2], ad.data_ptr(), bd.data_ptr(), btd.data_ptr(), c.data_ptr(), m, n, k, k, k, n, g.stream())
        assert st == 0, label
        torch.cuda.synchronize()
        got = c.cpu()
        assert torch.equal(torch.isnan(got), torch.isnan(truth)), f"{label}: NaN pattern differs from the reference"
        same = torch.where(torch.isnan(truth), torch.zeros_like(truth), truth).view(torch.int16) == torch.where(torch.isnan(got), torch.zeros_like(got), got).view(torch.int16)
        bad = (~same).nonzero()
        assert bad.numel() == 0, f"{label}: {bad.shape[0]} elements differ, first at {bad[0].tolist()}: got {got[tuple(bad[0])].item()} want {truth[tuple(bad[0])].item()}"


def test_asymmetric_identity_catches_transposes(g):
    """A = I, B asymmetric: C must equal B bit for bit (an MFMA C-layout row/col swap would not)."""
    n = 512
    a = np.eye(n, dtype=np.float16)
    b = (np.arange(n * 384, dtype=np.float32).reshape(n, 384) % 251 - 125).astype(np.float16)
    assert not np.array_equal(b[:384, :384], b[:384, :384].T)
    for cid in range(len(g.config_names())):
        got = g.gemm(a, b, plan=(cid, 1, 2))
        assert np.array_equal(got.view(np.uint16), b.view(np.uint16)), g.config_names()[cid]


def test_guard_bars_stay_intact(g, oracle):
    """Operands are views into flat buffers with 16384-element bars either side (reference
    zero_one_correctness_check.py:98-150): no byte outside the operand windows may change."""
    L = g.lib()
    bar = 16384
    m, n, k = 200, 136, 256
    rng = np.random.default_rng(5)
    a_np, b_np = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_f32acc(a_np, b_np)

    def guarded(x):
        flat = torch.randn(x.size + 2 * bar, dtype=torch.half, device="cuda")
        flat[bar:bar + x.size] = torch.from_numpy(x.reshape(-1)).cuda()
        return flat, flat.clone()

    fa, ca = guarded(a_np)
    fb, cb = guarded(b_np)
    fbt, cbt = guarded(np.ascontiguousarray(b_np.T))
    fc, cc = guarded(np.zeros((m, n), dtype=np.float16))
    esz = 2
    for cid in [-1] + list(range(len(g.config_names()))):
        for splits in (1, 4):
            fc[bar:bar + m * n] = float("nan")
            st = L.hgemm_mi355x_launch(cid, splits, 1, fa.data_ptr() + bar * esz, fb.data_ptr() + bar * esz,
                                       fbt.data_ptr() + bar * esz, fc.data_ptr() + bar * esz, m, n, k, k, k, n, g.stream())
            assert st == 0
            torch.cuda.synchronize()
            out = fc[bar:bar + m * n].view(m, n).cpu().numpy()
            assert np.array_equal(out.view(np.uint16), truth.view(np.uint16))
            for flat, clone in ((fa, ca), (fb, cb), (fbt, cbt)):
                assert torch.equal(flat, clone)
            assert torch.equal(fc[:bar], cc[:bar]) and torch.equal(fc[-bar:], cc[-bar:])


def test_unaligned_pointers_take_the_register_staged_kernel(g, oracle):
    L = g.lib()
    m, n, k = 64, 64, 64
    rng = np.random.default_rng(9)
    a_np, b_np = oracle.zero_one_inputs(m, n, k, rng)
    flat = torch.zeros(m * k + 1, dtype=torch.half, device="cuda")
    flat[1:] = torch.from_numpy(a_np.reshape(-1)).cuda()          # A starts 2 bytes off a 16-byte boundary
    b = torch.from_numpy(b_np).cuda()
    bt = b.t().contiguous()
    c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")
    assert L.hgemm_mi355x_fp32(flat.data_ptr() + 2, b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, g.stream()) == 0
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy().view(np.uint16), oracle.truth_f32acc(a_np, b_np).view(np.uint16))


@pytest.mark.parametrize("mnk,entry", [((64, 4096, 64), "fp32"), ((512, 4096, 4096), "fp32"), ((4096, 4096, 4096), "fp16"),
                                       ((64, 64, 16384), "fp32"), ((12288, 64, 8192), "fp32"), ((64, 16384, 12288), "fp16")])
def test_baseline_sizes_zero_one_exact(g, mnk, entry):
    """BASELINE.json sizes (and large-K / 12288 / 16384 cases), the reference's rule end to end:
    0/1 operands, truth = fp32 matmul on the CPU rounded to fp16, masked above 2047, diff must be 0."""
    from oracle import hgemm_oracle as oracle

    m, n, k = mnk
    rng = np.random.default_rng(m + n + k)
    a, b = oracle.zero_one_inputs(m, n, k, rng)           # {0,0,1} automatically beyond 8192
    truth = torch.matmul(torch.from_numpy(a).float(), torch.from_numpy(b).float()).half().numpy()
    got = g.gemm(a, b, entry)
    assert oracle.masked_max_diff(got, truth) == 0.0
    assert (np.abs(truth.astype(np.float32)) <= 2047).mean() > 0.4  # the mask must leave real coverage


def test_full_size_linearity_property(g):
    """Size-independent property at 512x4096x4096: for 0/1 B split into disjoint supports B1 + B2,
    C(A,B) == C(A,B1) + C(A,B2) exactly (all sums are integers <= 2047 here)."""
    m, n, k = 512, 4096, 4096
    rng = np.random.default_rng(2)
    a = (rng.random((m, k)) < 0.25).astype(np.float16)
    b = (rng.random((k, n)) < 0.25).astype(np.float16)
    sel = rng.random((k, n)) < 0.5
    c = g.gemm(a, b).astype(np.float32)
    c1 = g.gemm(a, (b * sel).astype(np.float16)).astype(np.float32)
    c2 = g.gemm(a, (b * ~sel).astype(np.float16)).astype(np.float32)
    assert c.max() <= 2047 and np.array_equal(c, c1 + c2)


def test_randn_tolerance_at_4096_cubed(g, oracle):
    """N(0,1) operands (what the benchmark feeds): relative error vs fp32 on a 192-row sample of C."""
    n = 4096
    rng = np.random.default_rng(4)
    a = rng.standard_normal((n, n), dtype=np.float32).astype(np.float16)
    b = rng.standard_normal((n, n), dtype=np.float32).astype(np.float16)
    rows = rng.choice(n, 192, replace=False)
    ref = a[rows].astype(np.float32) @ b.astype(np.float32)
    for entry in ("fp32", "fp16"):
        got = g.gemm(a, b, entry)
        assert oracle.relative_error(got[rows], ref) <= REL_TOL
    again = g.gemm(a, b, "fp32")
    assert np.array_equal(again.view(np.uint16), g.gemm(a, b, "fp32").view(np.uint16))  # run-to-run identical


@pytest.mark.parametrize("mnk,entry", [((64, 4096, 64), "fp32"), ((512, 4096, 4096), "fp32"), ((4096, 4096, 4096), "fp16")])
def test_baseline_sizes_whole_tile_normal_inputs_against_the_cpu_expression(g, oracle, mnk, entry):
    """The three BASELINE.json shapes on N(0,1) operands, EVERY element of C (not a row sample) against the reference's own
    oracle expression evaluated on the host -- (a.float() @ b.float()).half(), zero_one_correctness_check.py:85-90 -- within
    the north-star tolerance (1e-3 relative for fp32 accumulate; the fp16 entry accumulates in fp32 too, so the same bound
    instead of 1e-2), plus a per-element bound: no element further than 2 fp16 ulps of the row's largest magnitude."""
    m, n, k = mnk
    rng = np.random.default_rng(7 * m + 3 * n + k)
    a = rng.standard_normal((m, k), dtype=np.float32).astype(np.float16)
    b = rng.standard_normal((k, n), dtype=np.float32).astype(np.float16)
    ref32 = torch.matmul(torch.from_numpy(a).float(), torch.from_numpy(b).float())
    ref = ref32.half().numpy()
    got = g.gemm(a, b, entry)
    assert got.shape == ref.shape and not np.isnan(got).any()
    assert oracle.relative_error(got, ref32.numpy()) <= REL_TOL
    diff = np.abs(got.astype(np.float32) - ref.astype(np.float32))
    scale = np.abs(ref.astype(np.float32)).max(axis=1, keepdims=True)
    assert (diff <= scale * 2.0 ** -9).all()          # 2 ulps of the row maximum (fp16: 10 mantissa bits)
    # different summation orders round differently in the last place only: almost every element is bit-identical
    assert (got.view(np.uint16) == ref.view(np.uint16)).mean() > 0.9


def test_split_k_is_deterministic_and_within_tolerance(g, oracle):
    m, n, k = 128, 256, 16384
    rng = np.random.default_rng(6)
    a = rng.standard_normal((m, k), dtype=np.float32).astype(np.float16)
    b = rng.standard_normal((k, n), dtype=np.float32).astype(np.float16)
    ref = a.astype(np.float32) @ b.astype(np.float32)
    cid = g.config_names().index("t64x64_w2x2_m16_s4")
    outs = [g.gemm(a, b, plan=(cid, 16, 1)) for _ in range(3)]
    assert all(np.array_equal(outs[0].view(np.uint16), o.view(np.uint16)) for o in outs[1:])
    assert oracle.relative_error(outs[0], ref) <= REL_TOL


def test_baselines_agree_with_oracle(g, oracle):
    """rocBLAS / hipBLASLt wrappers (the speed-up denominators) compute the same product."""
    L = g.lib()
    m, n, k = 256, 512, 1024
    rng = np.random.default_rng(8)
    a_np, b_np = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a_np, b_np)
    a = torch.from_numpy(a_np).cuda()
    b = torch.from_numpy(b_np).cuda()
    bt = b.t().contiguous()
    assert L.hgemm_rocblas_init() == 0 and L.hgemm_hipblaslt_heuristic_init() == 0 and L.hgemm_hipblaslt_autotune_init() == 0
    assert L.hgemm_hipblaslt_autotune_find_best_nn(m, n, k, 0) == 0 and L.hgemm_hipblaslt_autotune_find_best_tn(m, n, k, 0) == 0
    assert L.hgemm_hipblaslt_autotune_candidates(0) >= 1 and L.hgemm_hipblaslt_autotune_candidates(1) >= 1
    import ctypes

    for name, second in [("hgemm_rocblas_nn", b), ("hgemm_rocblas_tn", bt), ("hgemm_hipblaslt_heuristic_nn", b),
                         ("hgemm_hipblaslt_heuristic_tn", bt), ("hgemm_hipblaslt_autotune_nn", b),
                         ("hgemm_hipblaslt_autotune_tn", bt)]:
        fn = getattr(L, name)
        fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        for acc in (0, 1):
            if "autotune" in name and acc == 1:
                continue  # algorithm was selected for acc=0 only
            c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")
            assert fn(a.data_ptr(), second.data_ptr(), c.data_ptr(), m, n, k, acc, g.stream()) == 0, (name, acc)
            torch.cuda.synchronize()
            assert oracle.masked_max_diff(c.cpu().numpy(), truth) == 0.0, (name, acc)
    # an autotuned entry point refuses a problem it was not tuned for
    c = torch.zeros((m, n), dtype=torch.half, device="cuda")
    assert L.hgemm_hipblaslt_autotune_nn(a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k // 2, 0, g.stream()) == -5
    L.hgemm_rocblas_destroy(); L.hgemm_hipblaslt_heuristic_destroy(); L.hgemm_hipblaslt_autotune_destroy()


def test_fill_normal_statistics(g):
    L = g.lib()
    import ctypes

    L.hgemm_fill_normal_f16.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_ulonglong, ctypes.c_void_p]
    x = torch.empty(1 << 20, dtype=torch.half, device="cuda")
    assert L.hgemm_fill_normal_f16(x.data_ptr(), x.numel(), 42, g.stream()) == 0
    torch.cuda.synchronize()
    xf = x.float()
    assert abs(xf.mean().item()) < 0.01 and abs(xf.std().item() - 1.0) < 0.01
    y = torch.empty_like(x)
    L.hgemm_fill_normal_f16(y.data_ptr(), y.numel(), 42, g.stream())
    torch.cuda.synchronize()
    assert torch.equal(x, y)


def test_dispatch_attached_timing_hook_is_one_shot_and_plausible(g):
    """hgemm_mi355x_time_next_launch (bench.py's roofline timing): kernel-exact, one launch only."""
    import torch

    L = g.lib()
    m = n = k = 2048
    a = torch.randn((m, k), dtype=torch.half, device="cuda")
    b = torch.randn((k, n), dtype=torch.half, device="cuda")
    bt = b.t().contiguous()
    c = torch.empty((m, n), dtype=torch.half, device="cuda")
    call = lambda: L.hgemm_mi355x_fp32(a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, g.stream())  # noqa: E731
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert call() == 0
    e1.record()
    torch.cuda.synchronize()
    marker_us = e0.elapsed_time(e1) * 1e3

    h0, h1 = L.hgemm_mi355x_event_create(), L.hgemm_mi355x_event_create()
    assert h0 and h1
    assert L.hgemm_mi355x_time_next_launch(h0, None) != 0           # both or neither
    assert L.hgemm_mi355x_time_next_launch(h0, h1) == 0
    assert call() == 0
    us = L.hgemm_mi355x_event_elapsed_us(h0, h1)
    assert 0.3 * marker_us < us < 1.5 * marker_us, (us, marker_us)  # 2048^3 is ~25-35 us on an MI355X
    assert call() == 0                                              # hook disarmed: the events stay as they are
    torch.cuda.synchronize()
    assert abs(L.hgemm_mi355x_event_elapsed_us(h0, h1) - us) < 1e-3
    # the generic fallback (odd K) serves an armed hook with plain markers
    a2 = torch.randn((64, 72), dtype=torch.half, device="cuda")
    b2 = torch.randn((72, 64), dtype=torch.half, device="cuda")
    c2 = torch.empty((64, 64), dtype=torch.half, device="cuda")
    assert L.hgemm_mi355x_time_next_launch(h0, h1) == 0
    assert L.hgemm_mi355x_fp32(a2.data_ptr(), b2.data_ptr(), b2.t().contiguous().data_ptr(), c2.data_ptr(), 64, 64, 72, g.stream()) == 0
    assert L.hgemm_mi355x_event_elapsed_us(h0, h1) > 0
    torch.testing.assert_close(c2.float(), (a2.float() @ b2.float()).half().float(), rtol=2e-3, atol=2e-2)
    assert L.hgemm_mi355x_event_destroy(h0) == 0 and L.hgemm_mi355x_event_destroy(h1) == 0


@pytest.mark.parametrize("shape", [(4352, 4352, 4096), (4300, 4400, 4096)])
def test_hybrid_tail_schedule_is_exact_on_zero_one_inputs(g, shape):
    """Persistent family, tile count not a multiple of the resident workgroups: full rounds + K-split
    tail tiles + compact-slab combine (hgemm_api.hip).  Integer-valued inputs make every fp32 sum exact,
    so the result must be bit-identical to the CPU product whatever the summation order."""
    m, n, k = shape
    rng = np.random.default_rng(7)
    a = (rng.random((m, k)) < 0.25).astype(np.float16)
    b = (rng.random((k, n)) < 0.25).astype(np.float16)
    truth = (torch.from_numpy(a).float() @ torch.from_numpy(b).float()).half().numpy()
    assert float(np.abs(truth).max()) <= 2047
    names = g.config_names()
    for cfg in ("s256x256_w2x2", "s128x256_w2x2", "q256x256_w2x2", "q256x128_w2x2"):
        got = g.gemm(a, b, plan=(names.index(cfg), 1, 4))
        assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), cfg
    # the library's own plan for the shape (whatever it picks) agrees as well
    assert np.array_equal(g.gemm(a, b).view(np.uint16), truth.view(np.uint16))


@pytest.mark.parametrize("shape", [(1000, 520, 200), (65, 30, 100), (33, 17, 40), (5, 4, 8), (257, 130, 66), (129, 67, 257),
                                   (300, 260, 2048), (2100, 2050, 328)])
def test_ragged_kernel_and_reference_kernel_explicitly(g, oracle, shape):
    """HGEMM_CONFIG_RAGGED (-2) and HGEMM_CONFIG_GENERIC (-1) on shapes no LDS-DMA geometry accepts: every piece
    width of the ragged loader (K % 8 / % 4 / % 2 / odd), scalar and vector C stores, both tile sizes."""
    m, n, k = shape
    rng = np.random.default_rng(m + 3 * n + 5 * k)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    for cid in (-2, -1):
        got = g.gemm(a, b, plan=(cid, 1, 1))
        assert not np.isnan(got).any()
        assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), cid
    # a table geometry asked to run a ragged problem is served by the ragged kernel instead of failing
    got = g.gemm(a, b, plan=(0, 1, 1))
    assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))


@pytest.mark.parametrize("shape", [(1000, 520, 200), (320, 448, 520), (200, 136, 1000), (4000, 520, 4008), (257, 1028, 72), (64, 64, 8),
                                   (300, 260, 2104), (520, 392, 728)])
def test_k_tail_of_every_geometry_that_takes_one(g, oracle, shape):
    """K % 64 != 0, K % 8 == 0.  The classic geometries zero-fill the partial last K-step through out-of-range DMA lanes;
    families q (16x16x32 members) and r run their "ktail" kernel variants (round 4): whole stages through the pipeline, the
    remainder from fragments loaded straight from global memory (they need one whole stage: a geometry whose stage is deeper
    than K is served by the any-shape kernel, also exact).  Every geometry x split-K form (+ family r's load flags), plus the
    library's own plan, bit-exact; guard: NaN-prefilled C."""
    m, n, k = shape
    L = g.lib()
    rng = np.random.default_rng(m + 3 * n + 5 * k)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    for entry in ("fp32", "fp16"):
        assert np.array_equal(g.gemm(a, b, entry).view(np.uint16), truth.view(np.uint16)), entry
    for cid, name in enumerate(g.config_names()):
        if name[0] not in "tqr":
            continue
        for splits in (1, 3, 2 | 0x10000) + ((2 | 0x10000 | 0x180000,) if name[0] == "r" else ()):
            got = g.gemm(a, b, plan=(cid, splits, 2))
            assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), (name, hex(splits))
    q = g.config_names().index("q256x256_w2x2")
    assert L.hgemm_mi355x_config_accepts_k(q, k) == (1 if k >= 64 else 0)   # (so the loop above ran its ktail variant, not a fallback)
    # operands that end exactly at the end of their allocation: the partial step must not read past it harmfully
    a_t = torch.from_numpy(a).cuda().clone()
    assert np.array_equal(g.gemm(a_t.cpu().numpy(), b).view(np.uint16), truth.view(np.uint16))


def test_ragged_kernel_randn_tolerance(g, oracle):
    m, n, k = 1000, 520, 200
    rng = np.random.default_rng(12)
    a = rng.standard_normal((m, k), dtype=np.float32).astype(np.float16)
    b = rng.standard_normal((k, n), dtype=np.float32).astype(np.float16)
    ref = a.astype(np.float32) @ b.astype(np.float32)
    assert oracle.relative_error(g.gemm(a, b, "fp32"), ref) <= REL_TOL


def test_split_k_on_two_streams_does_not_share_partials(g, oracle):
    """The split-K workspace is private to the (device, stream) pair (one global workspace raced)."""
    L = g.lib()
    names = g.config_names()
    cid = names.index("t64x64_w2x2_m16_s4")
    rng = np.random.default_rng(21)
    probs = []
    for seed in range(2):
        m, n, k = 192, 320, 8192
        a_np, b_np = oracle.zero_one_inputs(m, n, k, rng)
        probs.append((torch.from_numpy(a_np).cuda(), torch.from_numpy(b_np).cuda(), torch.from_numpy(np.ascontiguousarray(b_np.T)).cuda(),
                      oracle.truth_numpy(a_np, b_np), (m, n, k)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for form in (16, 16 | 0x10000):
        outs = [torch.full((p[4][0], p[4][1]), float("nan"), dtype=torch.half, device="cuda") for p in probs]
        for rep in range(20):
            for (a, b, bt, _, (m, n, k)), st, c in zip(probs, streams, outs):
                assert L.hgemm_mi355x_launch(cid, form, 1, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n,
                                             st.cuda_stream) == 0
        torch.cuda.synchronize()
        for p, c in zip(probs, outs):
            assert np.array_equal(c.cpu().numpy().view(np.uint16), p[3].view(np.uint16)), form


def _graph_lib(g):
    import ctypes

    L = g.lib()
    L.hgemm_mi355x_reserve_workspace.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p]
    return L


def _device_problem(oracle, m, n, k, seed):
    a_np, b_np = oracle.zero_one_inputs(m, n, k, np.random.default_rng(seed))
    return a_np, b_np


def _truth_fast(a_np, b_np):
    """The reference's CPU expression (zero_one_correctness_check.py:85-90) through torch's fp32 matmul: exact for
    0/1 inputs whatever the summation order (every partial sum is an integer < 2^24), and quick at the large
    shapes where the C restatement's scalar loop takes minutes (tests/test_oracle.py pins the two together)."""
    return (torch.from_numpy(a_np).float() @ torch.from_numpy(b_np).float()).half().numpy()


def test_hipgraph_capture_and_replay_is_exact(g, oracle):
    """Every plan form records into a hipGraph (kernel nodes only) and the replay computes: library plans of a
    launch-bound shape and of a hybrid-tail shape, explicit two-pass and single-launch split-K plans.  The graph
    is replayed on NEW operand values written into the captured buffers (a cached result would be caught)."""
    L = _graph_lib(g)
    names = g.config_names()
    t64 = names.index("t64x64_w2x2_m16_s4")
    cases = [((64, 4096, 64), None), ((192, 320, 8192), (t64, 16, 1)), ((192, 320, 8192), (t64, 16 | 0x10000, 1)),
             ((4352, 4352, 4096), None), ((520, 264, 200), None)]
    s = torch.cuda.Stream()
    bufs = []
    for (m, n, k), plan in cases:
        assert L.hgemm_mi355x_reserve_workspace(m, n, k, s.cuda_stream) == 0
        bufs.append((torch.empty((m, k), dtype=torch.half, device="cuda"), torch.empty((k, n), dtype=torch.half, device="cuda"),
                     torch.empty((n, k), dtype=torch.half, device="cuda"), torch.empty((m, n), dtype=torch.half, device="cuda")))
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        st = torch.cuda.current_stream().cuda_stream
        for ((m, n, k), plan), (a, b, bt, c) in zip(cases, bufs):
            if plan is None:
                rc = L.hgemm_mi355x_fp32(a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, st)
            else:
                rc = L.hgemm_mi355x_launch(plan[0], plan[1], plan[2], a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(),
                                           m, n, k, k, k, n, st)
            assert rc == 0, L.hgemm_mi355x_strerror(rc)
    for seed in (1, 2):
        truths = []
        for i, (((m, n, k), _), (a, b, bt, c)) in enumerate(zip(cases, bufs)):
            a_np, b_np = _device_problem(oracle, m, n, k, 100 * seed + i)
            a.copy_(torch.from_numpy(a_np)); b.copy_(torch.from_numpy(b_np)); bt.copy_(torch.from_numpy(np.ascontiguousarray(b_np.T)))
            c.fill_(float("nan"))
            truths.append(_truth_fast(a_np, b_np))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for (shape, plan), (_, _, _, c), truth in zip(cases, bufs, truths):
            assert np.array_equal(c.cpu().numpy().view(np.uint16), truth.view(np.uint16)), (shape, plan, seed)


def test_hipgraph_capture_never_allocates_and_growth_keeps_graphs_valid(g, oracle):
    """(i) A split-K plan captured on a stream that has no workspace yet runs without split-K (nothing may be
    allocated while capturing) and is still exact.  (ii) A workspace a graph has captured survives a later growth
    of that stream's workspace: the old graph replays correctly afterwards."""
    L = _graph_lib(g)
    names = g.config_names()
    t64 = names.index("t64x64_w2x2_m16_s4")
    m, n, k = 192, 320, 8192
    a_np, b_np = _device_problem(oracle, m, n, k, 9)
    truth = _truth_fast(a_np, b_np)
    a, b = torch.from_numpy(a_np).cuda(), torch.from_numpy(b_np).cuda()
    bt = b.t().contiguous()

    def capture(stream, splits):
        c = torch.full((m, n), float("nan"), dtype=torch.half, device="cuda")
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            rc = L.hgemm_mi355x_launch(t64, splits, 1, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n,
                                       torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.hgemm_mi355x_strerror(rc)
        return graph, c

    def replay_exact(graph, c):
        c.fill_(float("nan"))
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        return np.array_equal(c.cpu().numpy().view(np.uint16), truth.view(np.uint16))

    fresh = torch.cuda.Stream()                       # (i) no workspace on this stream
    for splits in (16, 16 | 0x10000):
        graph, c = capture(fresh, splits)
        assert replay_exact(graph, c)
    s = torch.cuda.Stream()                           # (ii)
    assert L.hgemm_mi355x_reserve_workspace(m, n, k, s.cuda_stream) == 0
    graphs = [capture(s, 16), capture(s, 16 | 0x10000)]
    assert all(replay_exact(gr, c) for gr, c in graphs)
    big_m = big_n = 2560                              # 8 slabs of 2560^2 fp32 = 200 MiB > the 64 MiB first allocation
    ba_np, bb_np = _device_problem(oracle, big_m, big_n, 512, 10)
    with torch.cuda.stream(s):
        ba, bb = torch.from_numpy(ba_np).cuda(), torch.from_numpy(bb_np).cuda()
        bbt = bb.t().contiguous()
        bc = torch.full((big_m, big_n), float("nan"), dtype=torch.half, device="cuda")
        assert L.hgemm_mi355x_launch(t64, 8, 1, ba.data_ptr(), bb.data_ptr(), bbt.data_ptr(), bc.data_ptr(), big_m, big_n, 512, 512, 512,
                                     big_n, s.cuda_stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(bc.cpu().numpy().view(np.uint16), _truth_fast(ba_np, bb_np).view(np.uint16))
    for _ in range(3):
        assert all(replay_exact(gr, c) for gr, c in graphs)
    del graphs, graph
    torch.cuda.synchronize()
    assert L.hgemm_mi355x_release_workspaces() == 0


def test_lent_workspace_too_small_degrades_to_no_split(g, oracle):
    import ctypes

    L = g.lib()
    L.hgemm_mi355x_set_workspace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    m, n, k = 128, 128, 4096
    rng = np.random.default_rng(22)
    a, b = oracle.zero_one_inputs(m, n, k, rng)
    truth = oracle.truth_numpy(a, b)
    cid = g.config_names().index("t64x64_w2x2_m16_s4")
    small = torch.empty((256 << 10) + 1024, dtype=torch.uint8, device="cuda")       # counters fit, slabs do not
    big = torch.empty(int(L.hgemm_mi355x_workspace_bytes(m, n, 8)), dtype=torch.uint8, device="cuda")
    try:
        assert L.hgemm_mi355x_set_workspace(small.data_ptr(), 100) == -1             # not even the counters
        for buf in (small, big):
            assert L.hgemm_mi355x_set_workspace(buf.data_ptr(), buf.numel()) == 0
            for form in (8, 8 | 0x10000):
                got = g.gemm(a, b, plan=(cid, form, 1))
                assert np.array_equal(got.view(np.uint16), truth.view(np.uint16))
    finally:
        assert L.hgemm_mi355x_set_workspace(None, 0) == 0
    assert np.array_equal(g.gemm(a, b, plan=(cid, 8, 1)).view(np.uint16), truth.view(np.uint16))


def test_timing_hook_spans_every_kernel_of_a_two_pass_plan(g):
    """the hook used to time the first dispatch only.  For a two-pass split-K plan (GEMM + combine) it
    must now agree with plain event markers around the whole call (which include both kernels and add only a few
    microseconds of queue gaps), and be clearly longer than the split GEMM kernel alone could be."""
    L = g.lib()
    m, n, k = 2048, 2048, 2048
    a = torch.randn((m, k), dtype=torch.half, device="cuda")
    b = torch.randn((k, n), dtype=torch.half, device="cuda")
    bt = b.t().contiguous()
    c = torch.empty((m, n), dtype=torch.half, device="cuda")
    cid = g.config_names().index("t128x128_w2x2_m16_s3")
    h0, h1 = L.hgemm_mi355x_event_create(), L.hgemm_mi355x_event_create()
    call = lambda form: L.hgemm_mi355x_launch(cid, form, 4, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n, g.stream())  # noqa: E731

    def hooked(form):
        best = 1e30
        for _ in range(5):
            assert L.hgemm_mi355x_time_next_launch(h0, h1) == 0
            assert call(form) == 0
            best = min(best, L.hgemm_mi355x_event_elapsed_us(h0, h1))
        return best

    def markers(form):
        best = 1e30
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); assert call(form) == 0; e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        return best

    for form in (1, 4, 4 | 0x10000):
        h, mk = hooked(form), markers(form)
        assert h > 5 and 0.6 * mk - 8 < h < 1.1 * mk + 2, (form, h, mk)
    L.hgemm_mi355x_event_destroy(h0); L.hgemm_mi355x_event_destroy(h1)


def test_race_screen_repeated_runs_are_bit_identical(g):
    """SURVEY section 5 / VERDICT r1 item 10: the LDS-DMA pipelines are ordered by counted vmcnt + barriers only;
    a mis-placed wait shows up as rare wrong tiles that come and go.  50 repeats per path on N(0,1) data, with
    randomised raster groups, must reproduce the first result bit for bit (every path is deterministic by
    construction, split-K included) -- and the first result is within tolerance of the fp32 reference."""
    L = g.lib()
    names = g.config_names()
    rng = np.random.default_rng(31)
    cases = [  # (config, splits, (M, N, K))
        ("s256x256_w2x2", 1, (1024, 1536, 2048)),          # SP, persistent (24 tiles)
        ("s256x256_w2x2_m32", 1, (1024, 1536, 2048)),      # SP, 32x32x16 MFMA
        ("s256x256_w2x2_m32", 1, (4608, 4608, 512)),       # persistent with several items per workgroup
        ("s256x128_w2x2", 1, (4352, 4352, 1024)),          # hybrid tail (34 x 34 tiles)
        ("r64x64_k256", 1, (2048, 64, 8192)),              # register-staged streaming kernel, K stagger
        ("r128x64_k128", 2 | 0x10000, (4100, 64, 4096)),   # ... ragged M edge (out-of-range rows read as zeros), fused split-K
        ("r64x128_k128", 4, (192, 4000, 2048)),            # ... ragged N edge, two-pass split-K
        ("q128x128_w2x2_k128", 1, (1024, 4096, 4096)),     # BK=128 stages, one tile per CU
        ("q128x128_w2x2_k128", 1, (2304, 2304, 640)),      # ... several items per workgroup, odd stage count (5)
        ("q128x128_w2x2", 2 | 0x10000, (1152, 1152, 4096)),# BK=64, two workgroups per CU, single-launch split-K
        ("q256x256_w2x2", 1, (1024, 1536, 2048)),          # early-A split, persistent (24 tiles)
        ("q256x256_w2x2", 1, (4608, 4608, 576)),           # ... several items per workgroup, odd K-step count (9)
        ("q256x128_w2x2", 1, (4352, 4352, 1024)),          # ... hybrid tail
        ("q256x256_w2x2_m32", 1, (1024, 1536, 2048)),      # early-A split on the 32x32x16 MFMA
        ("q256x256_w2x2_m32", 1, (4608, 4608, 576)),       # ... several items per workgroup, odd K-step count
        ("q256x256_w2x2_m32", 4 | 0x10000, (1024, 1024, 4096)),  # ... single-launch split-K
        ("q256x256_w2x2_m32", 3, (1100, 1000, 4096)),      # ... two-pass split-K, ragged edges
        ("q128x256_w2x2", 3, (1024, 1024, 4096)),          # ... two-pass split-K
        ("s256x256_w2x2", 4 | 0x10000, (1024, 1024, 4096)),# SP + single-launch split-K
        ("s128x256_w2x2", 3, (1024, 1024, 4096)),          # SP + two-pass split-K
        ("t64x64_w2x2_m16_s4", 16 | 0x10000, (192, 320, 8192)),  # classic 4-deep ring + single-launch split-K
        ("t128x128_w2x2_m16_s3", 1, (2048, 2048, 1024)),   # classic 3-deep ring
        ("t256x256_w2x4_m16_s2", 1, (2048, 2048, 1024)),   # classic 8-wave double buffer
        ("t128x64_w4x2_m16_s4", 1, (1000, 1096, 2048)),    # 8-wave 128x64 tile, 4-deep ring, ragged edges
        ("t64x128_w2x4_m16_s3", 2 | 0x10000, (512, 4096, 4160)),  # 8-wave 64x128 tile, 3-deep ring, two workgroups per CU, single-launch split-K
        ("q192x256_w2x2", 1, (6144, 4608, 576)),           # 192-row persistent tile (staged epilogue), several items, odd K-step count (9)
        ("q192x256_w2x2", 1, (1000, 520, 192)),            # ... three K-steps, ragged edges (the shape the withdrawn 192x192 member failed)
        ("q192x256_w2x2", 1, (1000, 516, 192)),            # ... NARROW epilogue (N % 8 = 4) at an odd K-step count: the variant that carried the
                                                           #     VALU -> asm-MFMA source hazard (tests/test_build_audit.py)
        ("q256x192_w2x2", 1, (584, 1004, 448)),            # ... the sibling, narrow epilogue, seven K-steps
        ("q256x192_w2x2", 3, (1100, 1000, 4096)),          # 192-column persistent tile (row epilogue), two-pass split-K, ragged edges
        ("q256x192_w2x2", 2 | 0x10000, (1024, 1536, 4160)),# ... single-launch split-K, odd slice lengths
        ("r96x128_k128", 2 | 0x10000, (1536, 128, 4096)),  # 96-row streaming tile
        ("r64x96_k128", 1, (100, 1056, 2048)),             # 96-column streaming tile, ragged M
        # stream-K (HGEMM_PLAN_STREAMK | workgroups): the slab + arrival-counter protocol under repetition
        ("r128x128_k128", 0x40000 | 256, (1536, 128, 8192)),      # every tile cut into ~21 parts
        ("r128x64_k128", 0x40000 | 200, (4100, 64, 4096)),        # ragged M edge, 65 tiles x 32 stages over 200 workgroups
        ("r64x128_k128", 0x40000 | 512, (192, 4000, 2048)),       # ragged N edge, two workgroups per CU
        ("t128x64_w4x2_m16_s4", 0x40000 | 200, (1000, 1096, 2048)),   # 8-wave tile, ragged edges, tiles cut at odd stages
        ("t64x128_w2x4_m16_s3", 0x40000 | 512, (512, 4096, 4160)),    # 65 stages per tile (odd), two workgroups per CU
        ("t128x128_w2x2_m16_s3", 0x40000 | 256, (3072, 3072, 1024)),  # 2.25 tiles per workgroup
        ("t64x64_w2x2_m16_s4", 0x40000 | 768, (192, 320, 8192)),      # 15 tiles x 128 stages: ~51 parts per tile
        # family r with two LDS buffers ("_d": one barrier per stage) and family w (wave-direct, no LDS staging), round 4
        ("r128x128_k128_d", 2 | 0x10000, (1536, 128, 4096)),      # single-launch split-K
        ("r64x128_k128_d", 1, (192, 4000, 2048)),                 # ragged N edge
        ("r128x64_k128_d", 3, (4100, 64, 4224)),                  # ragged M edge, two-pass split-K, 11 stages per slice
        ("r64x64_k256_d", 0x40000 | 256, (2048, 64, 8192)),       # stream-K, BKS = 256
        ("w64x64", 1, (64, 4096, 64)),                            # BASELINE config 1/2: one trip of two K slices
        ("w32x128", 1, (1000, 520, 128)),                         # ragged edges, one trip of four slices
        ("w128x32", 4, (520, 100, 1792)),                         # two-pass split-K, 14 slices per split (4 + 4 + 4 + 2)
        ("w16x16_k4", 8 | 0x10000, (64, 64, 4096)),               # four waves per tile walk K, single-launch split-K on top
        ("w32x32_k4", 2 | 0x10000, (100, 260, 2112)),             # ... ragged edges, 33 slices per split over four waves
        ("w16x32_k4", 1, (48, 96, 1024)),                         # ... M = 48: one and a half tiles
        # "ktail" kernel variants of families q and r (round 4): whole stages through the pipeline + fragments loaded directly
        ("q256x256_w2x2", 1, (4608, 4608, 616)),                  # several items per workgroup: the tail runs at item seams (9 K-steps + 40)
        ("q128x128_w2x2_k128", 1, (2304, 2304, 696)),             # ... BK = 128 stages, 5 stages + 56
        ("q192x256_w2x2", 2 | 0x10000, (1000, 520, 4440)),        # single-launch split-K (the tail rides on the last split), ragged edges
        ("q256x128_w2x2", 1, (4352, 4352, 1048)),                 # hybrid tail pass + K tail
        ("q128x256_w2x2", 3, (1024, 1024, 4104)),                 # two-pass split-K, K % 64 = 8 (one quarter of a slice)
        ("r64x64_k256", 1, (2048, 64, 9160)),                     # 35 stages + 200 (seven slices: trips of four, then singles)
        ("r128x64_k128_d", 2 | 0x10000 | 0x100000, (4100, 64, 4168)),   # two LDS buffers, ragged M edge, fused split-K, NT loads
        ("r64x128_k128", 4 | 0x80000, (192, 4000, 2104)),         # ragged N edge, two-pass split-K, per-XCD stagger
    ]
    for cfg, splits, (m, n, k) in cases:
        cid = names.index(cfg)
        a = torch.randn((m, k), dtype=torch.half, device="cuda")
        b = torch.randn((k, n), dtype=torch.half, device="cuda")
        bt = b.t().contiguous()
        c = torch.empty((m, n), dtype=torch.half, device="cuda")
        first = None
        for rep in range(50):
            c.fill_(float("nan"))
            # (a stream-K run cuts the tiles where the raster order puts them: the summation grouping is a function of the
            # plan, raster group included, so those paths keep one group; the same holds for family r's per-XCD stagger, whose
            # K walk starts where the workgroup's XCD says: the raster group decides which workgroup gets the tile)
            group = 4 if splits & (0x40000 | 0x80000) else int(rng.choice([1, 2, 3, 4, 8, 16]))
            assert L.hgemm_mi355x_launch(cid, splits, group, a.data_ptr(), b.data_ptr(), bt.data_ptr(), c.data_ptr(), m, n, k, k, k, n,
                                         g.stream()) == 0
            torch.cuda.synchronize()
            if first is None:
                first = c.clone()
                ref = a.float() @ b.float()
                rel = ((first.float() - ref).abs().max() / ref.abs().max()).item()
                assert rel <= REL_TOL, (cfg, rel)
            else:
                assert torch.equal(first.view(torch.int16), c.view(torch.int16)), (cfg, splits, group, rep)


def _plan_classes():
    """One bucket per (kernel family, split-K form, hybrid tail) of the shipped table; >= 50 grid shapes in total,
    every bucket represented, sizes the CPU oracle finishes in seconds."""
    import re

    rows = []
    for ln in (PKG_DIR / "csrc" / "hgemm_tuned_table.inc").read_text().splitlines():
        mm = re.match(r'\s*\{(\d+), (\d+), (\d+), "([^"]+)", (\d+), (\d+)\}', ln)
        if mm:
            rows.append((int(mm[1]), int(mm[2]), int(mm[3]), mm[4], int(mm[5]), int(mm[6])))
    buckets = {}
    for r in rows:
        m, n, k, cfg, sp, _ = r
        if 2.0 * m * n * k > 3e11:
            continue
        key = (cfg[0] + ("32" if cfg.endswith("m32") else ""), "1" if (sp & 0xFFFF) == 1 else ("fused" if sp & 0x10000 else "2pass"))
        buckets.setdefault(key, []).append(r)
    picked, rest = [], []
    per = max(4, 60 // max(1, len(buckets)))
    for key, rs in sorted(buckets.items()):
        rs = sorted(rs, key=lambda r: (r[0] * 7919 + r[1] * 104729 + r[2]) % 1009)   # deterministic spread
        picked += rs[:per]
        rest += rs[per:]
    # small buckets (a family with three plans left) must not shrink the sample: top up from the others, spread the same way
    rest = sorted(rest, key=lambda r: (r[0] * 7919 + r[1] * 104729 + r[2]) % 1009)
    picked += rest[:max(0, 56 - len(picked))]
    return picked


def test_sample_of_grid_shapes_at_their_shipped_plans(g, oracle):
    """>= 50 rows of the tuned table (every family x split-K form), both entry points, the reference rule:
    0/1 inputs ({0,0,1} beyond 8192), CPU truth, mask > 2047, difference exactly 0 (zero_one_correctness_check.py:65-92,
    263-268).  The whole grid runs in tests/test_gpu_grid.py (and is recorded in cuda-l2_amd/tuning/r03_parity_1000.jsonl)."""
    import ctypes

    L = g.lib()
    picked = _plan_classes()
    assert len(picked) >= 50
    rng = np.random.default_rng(41)
    for (m, n, k, cfg, sp, gm) in picked:
        c_, s_, g_ = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert L.hgemm_mi355x_plan(m, n, k, ctypes.byref(c_), ctypes.byref(s_), ctypes.byref(g_)) == 0
        assert L.hgemm_mi355x_config_name(c_.value).decode() == cfg and (s_.value, g_.value) == (sp, gm)
        a, b = oracle.zero_one_inputs(m, n, k, rng)
        truth = oracle.truth_numpy(a, b)
        for entry in ("fp32", "fp16"):
            got = g.gemm(a, b, entry)
            assert oracle.masked_max_diff(got, truth) == 0.0, (m, n, k, cfg, sp, gm, entry)
            assert np.array_equal(got.view(np.uint16), truth.view(np.uint16)), (m, n, k, cfg, sp, gm, entry)


def test_first_use_selection_picks_a_listed_plan_and_stays_exact(g, oracle):
    """Opt-in in-situ plan selection (hgemm_mi355x_set_insitu; the reference's first-call autotune, kernels/h100_F32F16F16F32/
    64_4096_64.cu:623-690): the first call of a shape times the table's plan and its alternates on the call's operands, records one
    of them for the process, and C is that plan's (exact) result; later calls reuse the choice; switching it off forgets it."""
    import ctypes

    L = g.lib()
    L.hgemm_mi355x_set_insitu(0)                   # whatever HGEMM_MI355X_INSITU says: start from "off, nothing recorded"
    assert L.hgemm_mi355x_set_insitu(1) == 0 and L.hgemm_mi355x_insitu_enabled() == 1
    try:
        for m, n, k in ((2048, 2048, 2048), (8192, 2048, 256), (1000, 520, 200), (256, 16384, 4096)):
            rng = np.random.default_rng(m + 7 * n + k)
            a_np, b_np = oracle.zero_one_inputs(m, n, k, rng)
            truth = oracle.truth_f32acc(a_np, b_np)
            cfg, sp, gm = (ctypes.c_int * 3)(), (ctypes.c_int * 3)(), (ctypes.c_int * 3)()
            ncand = L.hgemm_mi355x_insitu_candidates(m, n, k, cfg, sp, gm)
            cands = {(cfg[i], sp[i], gm[i]) for i in range(ncand)}
            c0, s0, g0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert L.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0)) == 0
            for entry in ("fp32", "fp16", "fp32"):
                got = g.gemm(a_np, b_np, entry)
                assert oracle.masked_max_diff(got, truth) == 0.0
                assert L.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0)) == 1
                choice = (c0.value, s0.value, g0.value)
                assert choice in cands, (choice, cands)
            first = choice
            g.gemm(a_np, b_np, "fp32")
            L.hgemm_mi355x_insitu_choice(m, n, k, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0))
            assert (c0.value, s0.value, g0.value) == first          # one choice per process and shape
    finally:
        assert L.hgemm_mi355x_set_insitu(0) == 1
    c0, s0, g0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.hgemm_mi355x_insitu_choice(2048, 2048, 2048, ctypes.byref(c0), ctypes.byref(s0), ctypes.byref(g0)) == 0
