#!/bin/bash
# Round-5 call K (the last GPU minutes): calls H and J disagree on the device-bound decades by the ORDER in which the report times its
# contenders (H: no cool-down, our isolated launch right behind the previous shape's autotune search; J: 40 ms cool-downs, our back-to-back
# box the first warm work after them) -- the 8-12 ms boxes are shorter than the board's power-management time constants.  Here: the 48
# shapes of >= 1e12 FLOP + every third of the 1e11 decade, same contenders and autotune budget, back-to-back boxes of 100 ms.
set -u
O=gpurun_out/r5k; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
HGEMM_AUTOTUNE_MAX_SECONDS=1.0 timeout ${1:-420} $T tune --plan-only --baselines --autotune --stream --stream-seconds 0.1 --shape-file cuda-l2_amd/tools/grid_shapes_compute_bound_sample.txt --out $O/compute_bound_long_boxes.jsonl > $O/run.log 2>&1; echo "rc=$? lines=$(wc -l < $O/compute_bound_long_boxes.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/compute_bound_long_boxes.jsonl 4 > $O/compute_bound_long_boxes.txt 2>&1
du -sh $O
