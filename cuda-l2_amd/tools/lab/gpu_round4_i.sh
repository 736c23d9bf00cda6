#!/bin/bash
# Round-4 call I: the 26 rows of the closing plan report (call G) below 0.9x of hipBLASLt with >= 2e9 FLOP, shipped plan first, against
# a broad candidate set (family r x split forms x load flags, the q members x split forms, the mid classic tiles, stream-K).  Every
# geometry and form in it is covered by the closing check that ran before (profiles/r04_check_final.log, call G).  Oracle parity of the
# three fastest plans per shape.
set -u
O=gpurun_out/r4i; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 600 $T tune --shape-file cuda-l2_amd/tuning/r04_retune_worst_rows_shapes.txt --cand-file cuda-l2_amd/tuning/r04_retune_worst_rows_candidates.txt --rank both --baselines --stream --out $O/worst_rows.jsonl > $O/worst_rows.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/worst_rows.jsonl)"
timeout 300 python tests/tools/verify_plans.py --plans $O/worst_rows.jsonl --top 3 --out $O/worst_rows_parity.jsonl 2>&1 | tail -1
du -sh $O
