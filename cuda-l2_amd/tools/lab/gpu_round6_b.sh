#!/bin/bash
# Round-6 call B: the library with the round's kernel changes -- the batched last-arriver combine of the single-launch split-K in
# families t / r / w / q (fused_combine, hgemm_kernel.hpp; 76 kernels changed, the 272 others are instruction-for-instruction round 5's)
# and the phase offset inside a CU for the two-resident members of family q (HGEMM_PLAN_CU_PHASE):
#  (1) EXACT FIRST: hgemm_tune check of EVERY geometry x form (the single-launch forms now also at 3 / 5 / 7 / 13 / 16 / 21 / 32 / 37 / 48
#      splits: every batch depth and remainder of the combine; the cu-phase forms), the persistent-walk shapes for family q (more
#      items than resident workgroups, incl. the fused forms at the item seams), long-K shapes for family w.  Nothing is timed when a
#      check fails;
#  (2) first look: HGEMM_PLAN_CU_PHASE on the small-K / large-MN class (shipped plan beside the two-resident members with and
#      without the flag);
#  (3) re-tune pass 1 "fused": per split-K row / small output with a long K the shipped plan beside its single-launch twin, neighbouring
#      split counts in both forms and the small members of t / w single-launch (tools/make_round6_candidates.py);
#  (4) oracle parity of the three fastest plans per shape of (2) and (3).
set -u
O=gpurun_out/r6b; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "48|fused" && strings $T | grep -q "cu-phase" || { echo "STALE hgemm_tune"; exit 1; }
timeout 1500 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 500 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
WCFG=w64x64,w32x128,w128x32,w32x64,w64x32,w16x16_k4,w32x32_k4,w16x32_k4,w32x16_k4
timeout 500 $T check --shapes 64_64_4096,128_64_8192,80_48_2080,256_256_2048,33_17_1056,512_64_16384 --configs $WCFG > $O/check_w_deep.log 2>&1; echo "check_w_deep rc=$? $(tail -1 $O/check_w_deep.log)"
for f in check_all check_q_walk check_w_deep; do grep -q " 0 failures" $O/$f.log || { echo "CHECK FAILED: $f"; grep FAIL $O/$f.log | head -30; exit 1; }; done
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
timeout 500 $T tune --shape-file cuda-l2_amd/tuning/r06_cuphase_shapes.txt --cand-file cuda-l2_amd/tuning/r06_cuphase_candidates.txt --rank both --out $O/cuphase_first_look.jsonl > $O/cuphase.log 2>&1; echo "cuphase rc=$? lines=$(wc -l < $O/cuphase_first_look.jsonl)"
timeout 1500 $T tune --shape-file cuda-l2_amd/tuning/r06_fused_shapes.txt --cand-file cuda-l2_amd/tuning/r06_fused_candidates.txt --rank both --out $O/retune_fused.jsonl > $O/retune_fused.log 2>&1; echo "retune fused rc=$? lines=$(wc -l < $O/retune_fused.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/retune_fused.jsonl --top 3 --out $O/retune_fused_parity.jsonl 2>&1 | tail -2
timeout 400 python tests/tools/verify_plans.py --plans $O/cuphase_first_look.jsonl --top 2 --out $O/cuphase_parity.jsonl 2>&1 | tail -2
du -sh $O
