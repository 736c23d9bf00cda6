#!/bin/bash
# Targeted re-tune of a shape subset with every plan option (+ oracle verification of the fastest candidates).
#   bash cuda-l2_amd/tools/gpu_tune_subset.sh RUN_ID SHAPE_FILE
set -u
RUN=$1; SF=$2
O=gpurun_out/tune_$RUN; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 1200 $T tune --shape-file $SF --fused --keep ${KEEP:-3.0} --max-cand ${MAXC:-24} --baselines --sweep-group --out $O/grid_tune.jsonl > $O/tune.log 2>&1
echo "tune rc=$? lines=$(wc -l < $O/grid_tune.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/grid_tune.jsonl --top 4 --repeats 2 --out $O/verify_candidates.jsonl > $O/verify.log 2>&1
echo "verify rc=$?"; tail -1 $O/verify.log
