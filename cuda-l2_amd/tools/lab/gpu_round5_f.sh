#!/bin/bash
# Round-5 call F: the phase offset's spacing capped at an epilogue's length (prologue arithmetic of every family-q kernel):
#  (1) EXACT FIRST: every geometry x form, walk shapes;  (2) re-tune pass 4 (every persistent plan whose workgroups walk more than one
#  item, any K: none / two / four / eight phase groups) + oracle parity of the three fastest per shape.
set -u
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "phase-offset8" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 400 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
for f in check_all check_q_walk; do grep -q " 0 failures" $O/$f.log || { echo "CHECK FAILED: $f"; grep FAIL $O/$f.log | head -30; exit 1; }; done
timeout 1200 $T tune --shape-file cuda-l2_amd/tuning/r05_retune_pass4_shapes.txt --cand-file cuda-l2_amd/tuning/r05_retune_pass4_candidates.txt --rank both --baselines --stream --out $O/retune4.jsonl > $O/retune4.log 2>&1; echo "retune4 rc=$? lines=$(wc -l < $O/retune4.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/retune4.jsonl --top 3 --out $O/retune4_parity.jsonl 2>&1 | tail -1
du -sh $O
