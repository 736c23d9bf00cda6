#!/bin/bash
# Round-2 GPU step 6: the shipped (re-tuned, verified) table: whole-grid parity through both entry points,
# full GPU test suite, bench.py, no-selection re-measurement of the shipped plans with hipBLASLt heuristic + autotune.
set -u
O=gpurun_out/r2f; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 900 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl > $O/verify.log 2>&1; echo "verify rc=$?"; tail -1 $O/verify.log
( timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 1500 $O/bench.json
HGEMM_AUTOTUNE_MAX_SECONDS=0.2 timeout 1500 $T tune --shape-file cuda-l2_amd/tools/grid_shapes.txt --plan-only --baselines --autotune --out $O/grid_plan_report.jsonl > $O/plan_report.log 2>&1
echo "plan report rc=$? lines=$(wc -l < $O/grid_plan_report.jsonl)"
