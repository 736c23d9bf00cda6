#!/bin/bash
# Round-4 call A: first run of the stream-K kernels (classic + register-staged families).  Exact check of every geometry x
# split-K / stream-K form BEFORE any timing (the rule of tools/lab/README.md), the whole `-m gpu` suite (the refactored main
# loops of the classic and r families serve every existing plan: whole-grid parity is part of the suite), then stream-K
# candidates against the shipped plans and hipBLASLt on the shapes VERDICT r3 names.
set -u
O=gpurun_out/r4a; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 600 $T check > $O/check_all.log 2>&1; echo "check rc=$?"; tail -1 $O/check_all.log; grep -c FAIL $O/check_all.log
timeout 300 $T check --shapes 1536_1152_2048,12288_128_1024,100_4000_2112,2050_130_8192 --configs r128x128_k128,r64x64_k256,r96x128_k128,r64x96_k128,t128x64_w4x2_m16_s4,t64x128_w2x4_m16_s3,t128x128_w2x2_m16_s3,t256x128_w4x2_m16_s2,t64x64_w2x2_m16_s4 > $O/check_sk_big.log 2>&1; echo "check2 rc=$?"; tail -1 $O/check_sk_big.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 $T tune --shape-file cuda-l2_amd/tools/lab/r4_streamk_shapes.txt --cand-file cuda-l2_amd/tools/lab/r4_streamk_first_candidates.txt --rank both --baselines --stream --out $O/streamk_first.jsonl > $O/streamk_first.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/streamk_first.jsonl)"
du -sh $O
