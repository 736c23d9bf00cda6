#!/bin/bash
# Round-4 call D: the final tree (table after both re-tune passes).  Exact check of every geometry x form (the log names what it
# covered), the whole `-m gpu` suite with its grid passes kept as records (0/1 parity and N(0,1) tolerance of the SHIPPED table),
# smoke, bench.py + rocprofv3 kernel stats of the same command, the per-geometry PMC table, and the device-clock plan reports:
# whole grid isolated + back to back against hipBLASLt-heuristic, quarter grid against hipBLASLt-AUTOTUNE with a 1 s budget per
# layout, the off-grid list.
set -u
O=gpurun_out/r4d; mkdir -p $O/pmc_table
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 400 $T check > $O/check_final.log 2>&1; echo "check rc=$?"; tail -1 $O/check_final.log
HGEMM_RECORD_DIR=$O/records timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
python cuda-l2_amd/tools/pmc_table.py shapes > $O/pmc_shapes.txt
sed -i 's/timeout 240 rocprofv3/timeout 80 rocprofv3/' cuda-l2_amd/tools/pmc_table.sh
bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_table $O/pmc_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_table $O/pmc_shapes.txt > $O/pmc_table.json 2> $O/pmc_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_table.json)"
timeout 500 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/grid_shapes.txt --out $O/grid_plan_report.jsonl > $O/grid_plan_report.log 2>&1; echo "grid report lines=$(wc -l < $O/grid_plan_report.jsonl)"
timeout 200 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
HGEMM_AUTOTUNE_MAX_SECONDS=1.0 timeout 480 $T tune --plan-only --baselines --autotune --shape-file cuda-l2_amd/tools/grid_shapes_quarter.txt --out $O/quarter_grid_plan_report_autotune.jsonl > $O/quarter_autotune.log 2>&1; echo "autotune report lines=$(wc -l < $O/quarter_grid_plan_report_autotune.jsonl)"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; du -sh $O
