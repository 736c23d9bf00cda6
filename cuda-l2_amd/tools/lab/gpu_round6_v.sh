#!/bin/bash
# Round-6 call V: call M's whole-grid plan report once more, on ANOTHER box, same table, same protocol (interleaved contenders, autotune
# winners from the cache): how far do the two north-star figures repeat?
set -u
O=gpurun_out/r6v; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
timeout 1200 $T tune --plan-only --baselines --autotune --stream --interleave --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --out $O/grid_plan_report_autotune_interleaved_second_box.jsonl > $O/grid.log 2>&1; echo "report rc=$? lines=$(wc -l < $O/grid_plan_report_autotune_interleaved_second_box.jsonl)"
