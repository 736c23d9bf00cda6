#!/bin/bash
# Round-6 call H: WIDE re-tune pass over the whole grid (today's boxes rank the mid / skinny plans differently from round 5's closing
# box: the two-resident 128-wide members read 8 % slower): per shape the shipped plan (--with-shipped) beside the
# model's 16 best (geometry x split count x both split-K forms), ranking figure sqrt(isolated x back to back), NT stores tried for the
# winner.  Check first (same library as call F, re-checked on this box).  Oracle parity of the three fastest per shape.
set -u
O=gpurun_out/r6h; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "48|fused" && strings $T | grep -q "with-shipped" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
grep -q " 0 failures" $O/check_all.log || { echo "CHECK FAILED"; grep FAIL $O/check_all.log | head -30; exit 1; }
timeout ${1:-1800} $T tune --shape-file cuda-l2_amd/tools/grid_shapes_shuffled.txt --with-shipped --fused --rank both --nt --max-cand 16 --keep 3.0 --out $O/retune_wide_pass1.jsonl > $O/retune_wide_pass1.log 2>&1; echo "wide pass1 rc=$? lines=$(wc -l < $O/retune_wide_pass1.jsonl)"
timeout 1500 python tests/tools/verify_plans.py --plans $O/retune_wide_pass1.jsonl --top 3 --out $O/retune_wide_pass1_parity.jsonl 2>&1 | tail -2
du -sh $O
