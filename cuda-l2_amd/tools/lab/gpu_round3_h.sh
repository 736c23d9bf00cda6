#!/bin/bash
# Round-3 final GPU call: the whole `-m gpu` suite on the final tree, bench.py (the driver's command), rocprofv3 kernel
# stats of the same command, the per-geometry PMC table, timelines of the shipped compute-bound plan.
set -u
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
timeout 300 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl 2>&1 | tail -1
timeout 600 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/grid_shapes.txt --out $O/grid_plan_report.jsonl > $O/grid_plan_report.log 2>&1; echo "plan report lines=$(wc -l < $O/grid_plan_report.jsonl)"
timeout 300 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
python cuda-l2_amd/tools/pmc_table.py shapes > $O/pmc_shapes.txt
mkdir -p $O/pmc_table; bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_table $O/pmc_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_table $O/pmc_shapes.txt > $O/pmc_table.json 2> $O/pmc_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_table.json)"
for sh in 4096_4096_4096 8192_8192_8192; do
  echo "# tl $sh"; LD_LIBRARY_PATH=$P/lib_tl timeout 60 $T bench --shape $sh --lib --timeline
done > $O/timeline.jsonl 2>&1
for sh in 4096_4096_4096 8192_8192_8192 512_4096_4096 16384_16384_256 2048_8192_8192; do
  echo "# lib"; timeout 30 $T bench --shape $sh --lib --power --seconds 1.0
  echo "# hipblaslt"; timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 1.0
done > $O/stream_final.jsonl 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O/prof -name "*kernel_trace.csv" -size +20M -delete; find $O/pmc_table -name "*kernel_trace.csv" -delete; du -sh $O
