#!/bin/bash
# Round-4 call H: family r's plan flags (exact: profiles/r04_check_family_r_flags.log, call F) on the rest of the skinny grid shapes
# (min(M, N) <= 256, the other side >= 2048, K >= 2048): shipped plan first, then the r members that fit x {nt loads, xcd stagger + nt
# loads}; oracle parity of the three fastest per shape.
set -u
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 100 $T check --configs r64x64_k256,r64x128_k128,r128x64_k128,r128x128_k128,r128x128_k128_d,r64x128_k128_d,r128x64_k128_d,r64x64_k256_d --shapes 512_1024_2048,1536_128_4096 > $O/check_r_flags_again.log 2>&1; echo "check rc=$? $(tail -1 $O/check_r_flags_again.log)"
timeout 500 $T tune --shape-file cuda-l2_amd/tuning/r04_retune_family_r_flags_pass2_shapes.txt --cand-file cuda-l2_amd/tuning/r04_retune_family_r_flags_pass2_candidates.txt --rank both --out $O/r_flags_pass2.jsonl > $O/r_flags_pass2.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/r_flags_pass2.jsonl)"
timeout 300 python tests/tools/verify_plans.py --plans $O/r_flags_pass2.jsonl --top 3 --out $O/r_flags_pass2_parity.jsonl 2>&1 | tail -1
du -sh $O
