#!/bin/bash
# Round-5 call C: the library with every kernel change of the round (m0 clobbers, plain lgkmcnt(0) in the staged epilogue, kstagger
# variants and two-resident members of family q with exact AGPR reservation, phase flags, family w's deeper trips):
#  (1) EXACT FIRST: hgemm_tune check of EVERY geometry x form on the default shapes (incl. two K-tail shapes), the persistent-walk shapes
#      for family q, and long-K shapes for family w's 16- and 8-slice trips.  Nothing is timed when a check fails.
#  (2) the round's re-tune: per shape the SHIPPED plan beside the candidates of tools/make_round5_candidates.py (merged:
#      tuning/r05_retune_candidates.txt), ranking figure sqrt(isolated x back to back), hipBLASLt in the same run;
#  (3) oracle parity of the three fastest plans per shape (verify_plans.py --plans).
set -u
O=gpurun_out/r5c; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 400 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
WCFG=w64x64,w32x128,w128x32,w32x64,w64x32,w16x16_k4,w32x32_k4,w16x32_k4,w32x16_k4
timeout 400 $T check --shapes 64_64_4096,128_64_8192,80_48_2080,256_256_2048,33_17_1056,512_64_16384 --configs $WCFG > $O/check_w_deep.log 2>&1; echo "check_w_deep rc=$? $(tail -1 $O/check_w_deep.log)"
for f in check_all check_q_walk check_w_deep; do grep -q " 0 failures" $O/$f.log || { echo "CHECK FAILED: $f"; grep FAIL $O/$f.log | head -30; exit 1; }; done
timeout 2400 $T tune --shape-file cuda-l2_amd/tuning/r05_retune_shapes.txt --cand-file cuda-l2_amd/tuning/r05_retune_candidates.txt --rank both --baselines --stream --out $O/retune.jsonl > $O/retune.log 2>&1; echo "retune rc=$? lines=$(wc -l < $O/retune.jsonl)"
timeout 1500 python tests/tools/verify_plans.py --plans $O/retune.jsonl --top 3 --out $O/retune_parity.jsonl 2>&1 | tail -2
du -sh $O
