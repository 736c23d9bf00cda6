#!/bin/bash
# Round-3 call J: the geometries added late in the round (8-wave 128x64 tiles, 192-wide family-q members, 96-wide family-r
# members): exact check, targeted re-tune against the shipped plans (ranked by isolated x back-to-back time), oracle parity of
# the two fastest plans per shape.
set -u
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
NEW=t128x64_w4x2_m16_s4,t64x128_w2x4_m16_s4,t128x64_w4x2_m16_s3,t64x128_w2x4_m16_s3,q192x256_w2x2,q256x192_w2x2,q192x192_w2x2,r96x128_k128,r96x64_k128,r128x96_k128,r64x96_k128
echo "== check"; timeout 200 $T check --configs $NEW > $O/check_default.log 2>&1; tail -2 $O/check_default.log
timeout 200 $T check --configs $NEW --shapes 384_768_1024,576_320_2048,1000_520_1024,12288_128_2048,192_192_256,96_96_2048,200_392_512,3072_3072_512 > $O/check_192.log 2>&1; tail -2 $O/check_192.log
echo "== tune"; timeout 420 $T tune --cand-file cuda-l2_amd/tuning/r03_late_candidates.txt --shape-file cuda-l2_amd/tuning/r03_late_shapes.txt --rank both --out $O/late_tune.jsonl > $O/late_tune.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/late_tune.jsonl)"; tail -3 $O/late_tune.log
echo "== parity"; timeout 300 python tests/tools/verify_plans.py --plans $O/late_tune.jsonl --top 2 --out $O/late_candidate_parity.jsonl 2>&1 | tail -1
du -sh $O
