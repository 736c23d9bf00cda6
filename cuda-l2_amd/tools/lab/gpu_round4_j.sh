#!/bin/bash
# Round-4 call J: the "ktail" kernel variants of families q and r (K % stage != 0: whole stages through the pipeline, the rest from
# fragments loaded straight from global memory, hgemm_kernel.hpp: direct_k_tail).  ORDER = the rule of tools/lab/README.md:
#   1. `hgemm_tune check` of the whole library (every geometry x form, now with two K-tail shapes) -- and nothing else if it fails;
#   2. the tail at the item seams of family q's persistent walks (more items than resident workgroups);
#   3. the whole `-m gpu` suite (K-tail test of every geometry, special values through the tails, race screen, the 80 off-grid
#      shapes at their new planner plans, and the grid passes again: kept as THE records of this library), smoke;
#   4. only then timings: the off-grid plan report (isolated + back to back against hipBLASLt) and, for the 14 off-grid shapes with
#      a K tail, the plan shipped before (a classic geometry) beside the planner's new plan and their siblings in ONE run;
#   5. rocprofv3 kernel trace of one K-tail problem through the public entry point: the kernel name shows the variant that ran.
set -u
O=gpurun_out/r4j; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 500 $T check > $O/check_final.log 2>&1; rc=$?; echo "check rc=$rc"; tail -1 $O/check_final.log
if [ $rc -ne 0 ]; then grep -m 40 FAIL $O/check_final.log; echo "STOP: check failed, nothing else runs"; exit 1; fi
QS=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2
timeout 300 $T check --shapes 4352_4352_328,4608_4352_200,3000_4400_456 --configs $QS > $O/check_q_item_seams.log 2>&1; rc=$?; echo "seam check rc=$rc"; tail -1 $O/check_q_item_seams.log
if [ $rc -ne 0 ]; then grep -m 40 FAIL $O/check_q_item_seams.log; echo "STOP: seam check failed"; exit 1; fi
HGEMM_RECORD_DIR=$O/records timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tee $O/pytest_gpu.log | tail -8; grep -E "^(FAILED|E  +Assert)" $O/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
timeout 240 $T tune --shape-file cuda-l2_amd/tuning/r04_ktail_shapes.txt --cand-file cuda-l2_amd/tuning/r04_ktail_candidates.txt --rank both --baselines --stream --out $O/ktail_candidates.jsonl > $O/ktail_candidates.log 2>&1; echo "ktail tune rc=$? lines=$(wc -l < $O/ktail_candidates.jsonl)"
timeout 120 python tests/tools/verify_plans.py --plans $O/ktail_candidates.jsonl --top 3 --out $O/ktail_candidates_parity.jsonl 2>&1 | tail -1
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $T bench --shape 4000_4000_4000 --lib --reps 20 > $O/prof_4000.log 2>&1; echo "rocprof rc=$?"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; du -sh $O
