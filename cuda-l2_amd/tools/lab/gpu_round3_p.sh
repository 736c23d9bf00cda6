#!/bin/bash
# Round-3 data for the next round: (1) the off-grid list back to back; (2) every grid shape whose shipped plan is a family-q
# kernel, back to back, with the shipped library and with the per-XCD K stagger build (lib_xs/, -DHGEMM_SQ_XSTAGGER=1): the
# map of where a stagger flag would pay.
set -u
O=gpurun_out/r3p; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
timeout 60 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report_stream.jsonl > $O/offgrid.log 2>&1; echo "offgrid rc=$? $(wc -l < $O/offgrid_plan_report_stream.jsonl)"
for lib in lib lib_xs; do
  LD_LIBRARY_PATH=$P/$lib timeout 70 $T tune --plan-only --baselines --stream --shape-file cuda-l2_amd/tuning/r03_q_plan_shapes.txt --out $O/q_plans_stream_$lib.jsonl > $O/q_$lib.log 2>&1; echo "$lib rc=$? $(wc -l < $O/q_plans_stream_$lib.jsonl)"
done
