#!/bin/bash
# Round-3 call Q1: the per-XCD K-stagger variants of family q (plan flag HGEMM_PLAN_K_STAGGER): exact checks of every variant,
# pairwise re-measurement (shipped plan vs the same plan staggered, --rank both) on the shapes the whole-grid map pointed at,
# oracle parity of both plans per shape.
set -u
O=gpurun_out/r3q; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
Q=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2
timeout 40 $T check --configs $Q --plan-flags 0x40000 --shapes 300_260_2048,512_1024_2048,256_256_1024,4608_4608_1024,1000_520_4096,2304_2304_640,1000_516_1088,3072_2304_576 > $O/check_stagger.log 2>&1; tail -1 $O/check_stagger.log
timeout 40 $T tune --cand-file cuda-l2_amd/tuning/r03_stagger_candidates.txt --shape-file cuda-l2_amd/tuning/r03_stagger_shapes.txt --rank both --out $O/stagger_tune.jsonl > $O/stagger_tune.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/stagger_tune.jsonl)"
timeout 45 python tests/tools/verify_plans.py --plans $O/stagger_tune.jsonl --top 2 --out $O/stagger_candidate_parity.jsonl 2>&1 | tail -1
