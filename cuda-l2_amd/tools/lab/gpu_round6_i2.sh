#!/bin/bash
# Round-6 call I2: stability pass 2 of the wide re-tune -- the rows wide pass 1 would change, shipped plan beside its four fastest, on another box.
#  re-tune pass 2 = (a) every row pass 1 would change, shipped plan beside its four fastest of pass 1, on ANOTHER box (stability gate of
#  tools/update_tuned_table.py --stability); (b) the mid class (5e9 .. 3e11 FLOP, K >= 2048, 128-wide shipped tiles): the 256-wide members of
#  family q at single-launch splits 2 .. 16 with and without the K stagger.  Then oracle parity of the three fastest per shape.
set -u
O=gpurun_out/r6i2; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "48|fused" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
grep -q " 0 failures" $O/check_all.log || { echo "CHECK FAILED"; grep FAIL $O/check_all.log | head -30; exit 1; }
timeout 900 $T tune --shape-file cuda-l2_amd/tuning/r06_wide_pass2_shapes.txt --cand-file cuda-l2_amd/tuning/r06_wide_pass2_candidates.txt --rank both --out $O/retune_wide_pass2.jsonl > $O/retune_wide_pass2.log 2>&1; echo "retune pass3 rc=$? lines=$(wc -l < $O/retune_wide_pass2.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/retune_wide_pass2.jsonl --top 3 --out $O/retune_wide_pass2_parity.jsonl 2>&1 | tail -2
du -sh $O
