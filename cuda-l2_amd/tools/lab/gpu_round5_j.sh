#!/bin/bash
# Round-5 call J: call H's isolated column was biased by the report's own order -- our plan is timed first for every shape, right behind
# the previous shape's ~2 s of sustained autotune launches, while the board's power management is still throttling (ours: isolated 8.8 %
# above back to back on the >= 1e11-FLOP shapes in call H, 2.2 % in a run without the search; the heuristic's, timed later: 0.5 %).
# Same report, same budget (1 s per layout), on the 404 shapes of >= 1e10 FLOP (the decades a clock moves), with --cooldown-ms 40: every
# isolated timing -- ours, rocBLAS, heuristic, autotune -- starts 40 ms after the previous work has drained.  Shuffled order, resumable.
set -u
O=gpurun_out/r5j; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
HGEMM_AUTOTUNE_MAX_SECONDS=1.0 timeout ${1:-1020} $T tune --plan-only --baselines --autotune --stream --cooldown-ms 40 --shape-file cuda-l2_amd/tools/grid_shapes_shuffled_1e10_up.txt --out $O/grid_1e10_up_autotune_cooldown.jsonl > $O/run.log 2>&1; echo "rc=$? lines=$(wc -l < $O/grid_1e10_up_autotune_cooldown.jsonl)"
python cuda-l2_amd/tools/tune_report.py $O/grid_1e10_up_autotune_cooldown.jsonl 6 > $O/grid_1e10_up_autotune_cooldown.txt 2>&1; head -c 600 $O/grid_1e10_up_autotune_cooldown.txt
du -sh $O
