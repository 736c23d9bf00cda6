#!/bin/bash
# Round-4 call G (run twice: after calls F / H, and again as THE closing run after call I changed 12 more rows): the FINAL tree (family r's plan flags
# in the table after calls F / H, the worst rows of the first closing report re-tuned in call I; the off-grid planner keeps
# stream-K corner plans as such).  Full exact check (the log names geometries and forms), the whole `-m gpu` suite with its grid
# passes kept as THE parity / tolerance records of the shipped table, smoke, the device-clock plan reports of the final table (grid
# isolated + back to back, off-grid), the per-geometry PMC table of the final table, and the reference-metric records of the rows
# that changed after call E's sweeps (same driver, same boxes; spliced into the sweep records by shape, eval_results/r04_sweep/README.md).
set -u
O=gpurun_out/r4g; mkdir -p $O/pmc_table
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 400 $T check > $O/check_final.log 2>&1; echo "check rc=$?"; tail -1 $O/check_final.log
HGEMM_RECORD_DIR=$O/records timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 200 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
timeout 500 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/grid_shapes.txt --out $O/grid_plan_report.jsonl > $O/grid_plan_report.log 2>&1; echo "grid report lines=$(wc -l < $O/grid_plan_report.jsonl)"
S=gpurun_out/r4g/sweep_changed_rows; mkdir -p $S
cp cuda-l2_amd/tuning/r04_rows_changed_after_call_e.txt cuda-l2_amd/tools/.changed.txt
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 offline tools/.changed.txt $W --cpu_max_flops 2e10 --cpu_seconds 0.02 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 offline tools/.changed.txt $W 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 server tools/.changed.txt $W --target_qps 100 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 server tools/.changed.txt $W --target_qps 100 2>&1 | tail -1
rm -f cuda-l2_amd/tools/.changed.txt
python cuda-l2_amd/tools/pmc_table.py shapes > $O/pmc_shapes.txt
sed -i 's/timeout 240 rocprofv3/timeout 80 rocprofv3/' cuda-l2_amd/tools/pmc_table.sh
bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_table $O/pmc_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_table $O/pmc_shapes.txt > $O/pmc_table.json 2> $O/pmc_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_table.json)"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; du -sh $O
