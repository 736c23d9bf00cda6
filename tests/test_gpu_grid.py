"""The reference checks EVERY shape before it times it (eval_one_file.sh:71-80 runs zero_one_correctness_check.py
first).  Here the whole 1000-shape grid -- and the off-grid shapes the neighbour planner serves -- go through that
rule inside the `-m gpu` suite, through the C ABI, against the CPU oracle: 0/1 inputs ({0,0,1} beyond 8192), guard
bars either side of every operand, NaN-prefilled C, masked difference exactly 0, unmasked bitwise equality, every
plan run twice (split-K arrival counters must return to zero).  Two operand seeds for the grid.  (~25 s per grid pass.)
"""
import json
import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO / "tests" / "tools")]

pytestmark = pytest.mark.gpu


def _run(tmp_path, name, extra):
    import verify_plans

    out = tmp_path / f"{name}.jsonl"
    rc = verify_plans.main(["--out", str(out), *extra])
    # HGEMM_RECORD_DIR: keep the records of this run (the committed cuda-l2_amd/tuning/r04_parity_1000.jsonl and
    # r04_randn_1000.jsonl are the grid passes of one `-m gpu` run: tools/lab/gpu_round4_d.sh)
    if os.environ.get("HGEMM_RECORD_DIR"):
        import shutil

        Path(os.environ["HGEMM_RECORD_DIR"]).mkdir(parents=True, exist_ok=True)
        shutil.copy(out, Path(os.environ["HGEMM_RECORD_DIR"]) / f"{name}.jsonl")
    recs = [json.loads(ln) for ln in out.read_text().splitlines()]
    bad = [r for r in recs if not r["pass"]]
    assert rc == 0 and not bad, bad[:5]
    return recs


@pytest.mark.parametrize("seed", [20260925, 7])
def test_every_grid_shape_is_exact_through_both_entry_points(tmp_path, seed):
    recs = _run(tmp_path, f"grid_{seed}", ["--seed", str(seed)])
    assert len(recs) == 2000 and {r["run"] for r in recs} == {"fp32", "fp16"}
    assert len({r["mnk"] for r in recs}) == 1000
    assert all(r["bitwise_equal_unmasked"] and r["guard_bars_intact"] and r["inputs_unchanged"] and r["repeats"] == 2 for r in recs)
    # every kernel family and both split-K forms are among the plans that were just checked
    plans = {(r["plan"]["config"][0], r["plan"]["splits"] > 1, r["plan"]["fused"]) for r in recs}
    assert {"t", "q", "r"} <= {p[0] for p in plans}
    assert any(p[1] and p[2] for p in plans) and any(p[1] and not p[2] for p in plans)


def test_off_grid_shapes_are_exact_at_the_neighbour_planner_s_plans(tmp_path):
    shapes = REPO / "cuda-l2_amd" / "tools" / "offgrid_shapes.txt"
    recs = _run(tmp_path, "offgrid", ["--shape-file", str(shapes)])
    n = len([ln for ln in shapes.read_text().splitlines() if ln.strip() and not ln.startswith("#")])
    assert len(recs) == 2 * n and n >= 50


def test_every_grid_shape_is_within_tolerance_on_normal_inputs(tmp_path):
    """BASELINE.json north_star: "every shape must match torch.matmul within 1e-2 rel (fp16 acc) / 1e-3 rel (fp32 acc)".  All
    1000 grid shapes x both entry points on N(0,1) operands, 1e-3 for both (CDNA4 accumulates in fp32 either way), reference
    rows computed without the library (CPU fp32 / GPU fp64): tests/tools/verify_plans.py --randn."""
    recs = _run(tmp_path, "grid_randn", ["--randn"])
    assert len(recs) == 2000 and {r["run"] for r in recs} == {"fp32", "fp16"} and len({r["mnk"] for r in recs}) == 1000
    assert all(r["relative_error"] <= 1e-3 and r["rows_checked"] >= min(128, int(r["mnk"].split("_")[0])) for r in recs)
    assert {"cpu fp32", "gpu fp64"} == {r["reference"] for r in recs}


def test_off_grid_shapes_are_within_tolerance_on_normal_inputs(tmp_path):
    shapes = REPO / "cuda-l2_amd" / "tools" / "offgrid_shapes.txt"
    recs = _run(tmp_path, "offgrid_randn", ["--randn", "--shape-file", str(shapes)])
    assert len(recs) >= 100 and all(r["pass"] for r in recs)
