#!/bin/bash
# Round-4 call B: exact checks of the new members (family r with two LDS buffers, family w) BEFORE any timing; the whole
# `-m gpu` suite (now with the N(0,1) tolerance over the whole grid and the special-values test); the round-4 re-tune (shipped
# plan re-measured beside stream-K / "_d" / family-w candidates, ranked by sqrt(isolated x back-to-back)); oracle parity of the
# two fastest plans of every re-tuned shape (tools/make_tuned_table.py --verified adopts nothing else).
set -u
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
NEW=r128x128_k128_d,r64x128_k128_d,r128x64_k128_d,r64x64_k256_d,w64x64,w32x128,w128x32,w32x64,w64x32,w16x16_k4,w32x32_k4,w16x32_k4,w32x16_k4
timeout 300 $T check --configs $NEW > $O/check_new.log 2>&1; echo "check rc=$?"; tail -1 $O/check_new.log; grep FAIL $O/check_new.log | head -20
timeout 200 $T check --configs $NEW --shapes 1536_1152_2048,12288_128_1024,100_4000_2112,2050_130_8192,64_4096_64,48_80_4096,1000_520_128 > $O/check_new2.log 2>&1; echo "check2 rc=$?"; tail -1 $O/check_new2.log; grep FAIL $O/check_new2.log | head -20
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15
timeout 1000 $T tune --shape-file cuda-l2_amd/tuning/r04_retune_shapes.txt --cand-file cuda-l2_amd/tuning/r04_retune_candidates.txt --rank both --nt --out $O/r04_retune.jsonl > $O/r04_retune.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/r04_retune.jsonl)"
timeout 600 python tests/tools/verify_plans.py --plans $O/r04_retune.jsonl --top 2 --out $O/r04_candidate_parity.jsonl 2>&1 | tail -3
du -sh $O
