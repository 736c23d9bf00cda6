#!/bin/bash
# Round-6 call G: whole-grid plan report of the table after the "fused" re-tune, on the device clock, against rocBLAS / hipBLASLt-heuristic /
# hipBLASLt-AUTOTUNE (winners from the 1 s-per-layout cache: no search inside the run), contenders timed in INTERLEAVED rounds
# (hgemm_tune tune --plan-only --baselines --autotune --stream --interleave); shuffled shape order.  $1 = seconds limit.
set -u
O=gpurun_out/r6g; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
timeout ${1:-1500} $T tune --plan-only --baselines --autotune --stream --interleave --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --out $O/grid_plan_report_interleaved.jsonl > $O/grid.log 2>&1; echo "report rc=$? lines=$(wc -l < $O/grid_plan_report_interleaved.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/grid_plan_report_interleaved.jsonl 8 > $O/grid_plan_report_interleaved.txt 2>&1; head -c 1200 $O/grid_plan_report_interleaved.txt
grep -c '"autotune_from_cache": \[1, 1\]' $O/grid_plan_report_interleaved.jsonl
