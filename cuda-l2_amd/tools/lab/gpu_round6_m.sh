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
timeout 1200 $T tune --plan-only --baselines --autotune --stream --interleave --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --out $O/grid_plan_report_autotune_interleaved.jsonl > $O/grid.log 2>&1; echo "report rc=$? lines=$(wc -l < $O/grid_plan_report_autotune_interleaved.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/grid_plan_report_autotune_interleaved.jsonl 8 > $O/grid_plan_report_autotune_interleaved.txt 2>&1; head -c 600 $O/grid_plan_report_autotune_interleaved.txt
python - <<PY > $O/shapes_1e11_up.txt
for ln in open("cuda-l2_amd/tools/grid_shapes_shuffled.txt"):
    m, n, k = map(int, ln.split("_"))
    if 2.0 * m * n * k >= 1e11: print(ln.strip())
PY
timeout 900 $T tune --plan-only --baselines --autotune --stream --interleave --reverse --shape-file $O/shapes_1e11_up.txt --out $O/grid_1e11_up_plan_report_autotune_interleaved_reversed.jsonl > $O/grid_rev.log 2>&1; echo "reversed rc=$? lines=$(wc -l < $O/grid_1e11_up_plan_report_autotune_interleaved_reversed.jsonl)"
bash cuda-l2_amd/tools/lab/gpu_round6_power.sh
