#!/bin/bash
# Round-5 call H: THE north-star measurement (BASELINE.json: geomean speedup over hipBLASLt-autotune on the 1000-shape grid), on
# the device clock, for the shipped table of the closing run: per shape the shipped plan against rocBLAS, hipBLASLt-heuristic (tn, nn)
# and hipBLASLt-AUTOTUNE with a real budget (HGEMM_AUTOTUNE_MAX_SECONDS=1 per layout: every candidate the heuristic returns, 50 warm-up
# + 100 timed shuffled rounds, median -- the reference's protocol, cublas/fp32/hgemm_cublaslt_auto_tuning.cu:108-306), isolated launches
# AND back-to-back queues.  Shapes in a fixed shuffled order (tools/grid_shapes_shuffled.txt), so that a run cut short by the budget is a
# random sample of the grid, not its small end; --out is resumable.
set -u
O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
HGEMM_AUTOTUNE_MAX_SECONDS=1.0 timeout ${1:-2400} $T tune --plan-only --baselines --autotune --stream --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --out $O/grid_plan_report_autotune.jsonl > $O/grid_autotune.log 2>&1; echo "autotune report rc=$? lines=$(wc -l < $O/grid_plan_report_autotune.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/grid_plan_report_autotune.jsonl 8 > $O/grid_plan_report_autotune.txt 2>&1; head -c 1500 $O/grid_plan_report_autotune.txt
du -sh $O
