#!/bin/bash
# Round-2 final validation of the shipped table: whole-grid parity through both entry points, full GPU test suite,
# no-selection re-measurement of the shipped plans (full grid vs hipBLASLt heuristic; quarter grid also vs autotune).
set -u
O=gpurun_out/r2z; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 900 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl > $O/verify.log 2>&1; echo "verify rc=$?"; tail -1 $O/verify.log
( timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
timeout 600 $T tune --shape-file cuda-l2_amd/tools/grid_shapes.txt --plan-only --baselines --out $O/grid_plan_report.jsonl > $O/plan_report.log 2>&1
echo "plan report rc=$? lines=$(wc -l < $O/grid_plan_report.jsonl)"
HGEMM_AUTOTUNE_MAX_SECONDS=0.05 timeout 400 $T tune --shape-file cuda-l2_amd/tools/grid_shapes_quarter.txt --plan-only --baselines --autotune --out $O/quarter_plan_report_autotune.jsonl > $O/plan_report_q.log 2>&1
echo "quarter report rc=$? lines=$(wc -l < $O/quarter_plan_report_autotune.jsonl)"
