#!/bin/bash
# Round-4 call K: THE closing run of the library with the ktail kernel variants and the off-grid planner rules that the first
# K-tail measurements suggested (calls J / J3).  Same order as call J: full check, the item-seam check of family q, the whole
# `-m gpu` suite with its grid and off-grid passes kept as the records of this library, smoke -- and only then timings: the
# off-grid plan report of the final planner (isolated + back to back against hipBLASLt), rocprofv3 kernel stats of one K-tail problem
# through the public entry point, and PMC passes (MFMA-busy, FETCH_SIZE, WRITE_SIZE) of six K-tail shapes at their planner plans.
set -u
O=gpurun_out/r4k; mkdir -p $O/pmc_ktail
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 500 $T check > $O/check_final.log 2>&1; rc=$?; echo "check rc=$rc"; tail -1 $O/check_final.log
if [ $rc -ne 0 ]; then grep -m 40 FAIL $O/check_final.log; echo "STOP: check failed, nothing else runs"; exit 1; fi
QS=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2
timeout 300 $T check --shapes 4352_4352_328,4608_4352_200,3000_4400_456 --configs $QS > $O/check_q_item_seams.log 2>&1; rc=$?; echo "seam check rc=$rc"; tail -1 $O/check_q_item_seams.log
if [ $rc -ne 0 ]; then grep -m 40 FAIL $O/check_q_item_seams.log; echo "STOP: seam check failed"; exit 1; fi
HGEMM_RECORD_DIR=$O/records timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tee $O/pytest_gpu.log | tail -4; grep -E "^(FAILED|E  +Assert)" $O/pytest_gpu.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 200 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- $T bench --shape 4000_4000_4000 --lib --reps 20 > $O/prof_4000.log 2>&1; echo "rocprof rc=$?"
printf "4000_4000_4000\n12032_2048_7152\n1332_3108_4440\n9000_9000_520\n64_16384_9160\n128_8192_9616\n" > $O/pmc_ktail_shapes.txt
sed -i 's/timeout 240 rocprofv3/timeout 60 rocprofv3/' cuda-l2_amd/tools/pmc_table.sh
bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_ktail $O/pmc_ktail_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_ktail $O/pmc_ktail_shapes.txt > $O/pmc_ktail_table.json 2> $O/pmc_ktail_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_ktail_table.json)"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +4M -delete; du -sh $O
