#!/bin/bash
# Round-6 call M: THE north-star measurement on the device clock for the table that ships: the whole grid at the shipped plans against
# rocBLAS, hipBLASLt-heuristic and hipBLASLt-AUTOTUNE (1 s per layout, winners from the cache: nothing is searched inside the run),
# isolated and back to back, contenders in INTERLEAVED rounds; then the same report with the rotation REVERSED on
# the shapes of >= 1e11 FLOP (how far does the order move the figures now?); then the energy table.
set -u
O=gpurun_out/r6m; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
# (the off-grid planner's sibling rule changed after call K -- host code only, one off-grid shape: its parity / tolerance records and the
#  off-grid plan report are taken again here; the grid's rows come from the table and are call K's)
timeout 300 $T check --shapes 1968_576_4096,1000_520_200 > $O/check_offgrid_rule.log 2>&1; echo "check rc=$? $(tail -1 $O/check_offgrid_rule.log)"
HGEMM_RECORD_DIR=$O/records timeout 900 python -m pytest tests/test_gpu_grid.py -m gpu -q -k "off" > $O/pytest_offgrid.log 2>&1; echo "pytest offgrid rc=$? $(tail -1 $O/pytest_offgrid.log)"
timeout 400 $T tune --plan-only --baselines --stream --interleave --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
timeout 1200 $T tune --plan-only --baselines --autotune --stream --interleave --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --out $O/grid_plan_report_autotune_interleaved.jsonl > $O/grid.log 2>&1; echo "report rc=$? lines=$(wc -l < $O/grid_plan_report_autotune_interleaved.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/grid_plan_report_autotune_interleaved.jsonl 8 > $O/grid_plan_report_autotune_interleaved.txt 2>&1; head -c 600 $O/grid_plan_report_autotune_interleaved.txt
python - <<PY > $O/shapes_1e11_up.txt
for ln in open("cuda-l2_amd/tools/grid_shapes_shuffled.txt"):
    m, n, k = map(int, ln.split("_"))
    if 2.0 * m * n * k >= 1e11: print(ln.strip())
PY
timeout 900 $T tune --plan-only --baselines --autotune --stream --interleave --reverse --shape-file $O/shapes_1e11_up.txt --out $O/grid_1e11_up_plan_report_autotune_interleaved_reversed.jsonl > $O/grid_rev.log 2>&1; echo "reversed rc=$? lines=$(wc -l < $O/grid_1e11_up_plan_report_autotune_interleaved_reversed.jsonl)"
bash cuda-l2_amd/tools/lab/gpu_round6_power.sh
