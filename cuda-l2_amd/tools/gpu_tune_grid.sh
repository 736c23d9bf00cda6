#!/bin/bash
# Full-grid tuning run + oracle verification of the fastest candidates.
#   bash cuda-l2_amd/tools/gpu_tune_grid.sh RUN_ID     -> gpurun_out/tune_RUN_ID/{grid_tune.jsonl, verify_candidates.jsonl}
set -u
RUN=${1:-a}
O=gpurun_out/tune_$RUN; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 2400 $T tune --shape-file cuda-l2_amd/tools/grid_shapes.txt --fused --keep 3.0 --max-cand 24 --baselines --sweep-group \
    --out $O/grid_tune.jsonl > $O/tune.log 2>&1
echo "tune rc=$? lines=$(wc -l < $O/grid_tune.jsonl)"
timeout 1200 python tests/tools/verify_plans.py --plans $O/grid_tune.jsonl --top 4 --repeats 2 --out $O/verify_candidates.jsonl > $O/verify.log 2>&1
echo "verify rc=$?"; tail -2 $O/verify.log
