#!/bin/bash
# Round-4 call L (the last 2.8 GPU-minutes): two more off-grid planner guards (family w's "_k4" members beyond one workgroup per CU,
# family r workgroup counts between 1 and 1.75 rounds of the chip; host code only -- the kernels are call K's, checked there:
# profiles/r04_check_final.log; tests/test_build_audit.py fingerprints them).  The off-grid parity / tolerance tests at the new planner
# plans FIRST, then the off-grid plan report.
set -u
O=gpurun_out/r4l; mkdir -p $O
export TMPDIR=/tmp
HGEMM_RECORD_DIR=$O/records timeout 120 python -m pytest tests/test_gpu_grid.py -m gpu -q -k "off_grid" 2>&1 | tail -3
timeout 100 cuda-l2_amd/bin/hgemm_tune tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
