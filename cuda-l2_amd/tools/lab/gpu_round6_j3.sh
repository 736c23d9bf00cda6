#!/bin/bash
# Round-6 call J3: flag pass 3 -- per row the shipped plan beside the same plan with family q's K stagger / NT stores / phase offset / raster group toggled, family r's load flags, NT stores for the classic family.
#  re-tune pass 2 = (a) every row pass 1 would change, shipped plan beside its four fastest of pass 1, on ANOTHER box (stability gate of
#  tools/update_tuned_table.py --stability); (b) the mid class (5e9 .. 3e11 FLOP, K >= 2048, 128-wide shipped tiles): the 256-wide members of
#  family q at single-launch splits 2 .. 16 with and without the K stagger.  Then oracle parity of the three fastest per shape.
set -u
O=gpurun_out/r6j3; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "48|fused" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
grep -q " 0 failures" $O/check_all.log || { echo "CHECK FAILED"; grep FAIL $O/check_all.log | head -30; exit 1; }
timeout 900 $T tune --shape-file cuda-l2_amd/tuning/r06_flags_shapes.txt --cand-file cuda-l2_amd/tuning/r06_flags_candidates.txt --rank both --out $O/retune_flags_pass3.jsonl > $O/retune_flags_pass3.log 2>&1; echo "retune pass3 rc=$? lines=$(wc -l < $O/retune_flags_pass3.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/retune_flags_pass3.jsonl --top 3 --out $O/retune_flags_pass3_parity.jsonl 2>&1 | tail -2
du -sh $O
