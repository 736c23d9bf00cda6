#!/bin/bash
# Round-5 call A: first contact of the round's kernel changes with a GPU.
#  (1) EXACT FIRST: hgemm_tune check of every family-q member -- the kstagger variants (HGEMM_PLAN_XCD_STAGGER), the two-resident
#      members (q192x128, q128x192; q128x128 with exactly its accumulators reserved), the "m0" clobbers and the plain lgkmcnt(0) of the
#      staged epilogue changed every q kernel's instruction stream -- on the default shapes and on three shapes with more work items
#      than resident workgroups (persistent walk, item seams).
#  (2) only when (1) is clean: first timings -- two-resident members on the small-K / large-MN class, the kstagger variant on every
#      4th family-q row, raster groups on the largest shapes -- shipped plan re-measured beside the candidates, hipBLASLt in the same run.
#  (3) counters: L2 hit rate / fabric traffic of q256x256 against K, hipBLASLt's kernel beside it.
#  (4) A/B of the round-4 library (lib_r4/, built from 88beaa7) against this one on 4096^3 and 16384^2 x 256 (cost of the clobber / wait).
set -u
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2,q256x256_w2x2_m32
timeout 600 $T check --configs $QCFG > $O/check_q.log 2>&1; echo "check_q rc=$? $(tail -1 $O/check_q.log)"
timeout 400 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
OK=1; grep -q " 0 failures" $O/check_q.log && grep -q " 0 failures" $O/check_q_walk.log || OK=0
if [ $OK = 1 ]; then
  timeout 500 $T tune --shape-file cuda-l2_amd/tuning/r05_overlap_first_look_shapes.txt --cand-file cuda-l2_amd/tuning/r05_overlap_first_look_candidates.txt --rank both --baselines --stream --out $O/overlap_first_look.jsonl > $O/overlap.log 2>&1; echo "overlap rc=$? lines=$(wc -l < $O/overlap_first_look.jsonl)"
  timeout 400 $T tune --shape-file cuda-l2_amd/tuning/r05_stagger_first_look_shapes.txt --cand-file cuda-l2_amd/tuning/r05_stagger_first_look_candidates.txt --rank both --baselines --stream --out $O/stagger_first_look.jsonl > $O/stagger.log 2>&1; echo "stagger rc=$? lines=$(wc -l < $O/stagger_first_look.jsonl)"
  printf "8192_8192_8192\n16384_16384_16384\n12288_16384_8192\n" > $O/big.txt
  timeout 200 $T tune --shape-file $O/big.txt --cand-file cuda-l2_amd/tuning/r05_raster_group_candidates.txt --rank both --baselines --stream --out $O/raster_groups.jsonl > $O/raster.log 2>&1; echo "raster rc=$?"
else
  echo "CHECK FAILED: no timing of the new kernels"; grep FAIL $O/check_q.log $O/check_q_walk.log | head -40
fi
# (3) counters (kernels of the shipped table: exact since round 3 and re-checked above)
LIST=4096_4096_4096,4096_4096_8192,4096_4096_16384,8192_8192_8192,12288_16384_8192,16384_16384_16384
i=0
for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  timeout 120 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_k/pass$i -- $T bench --shapes $LIST --lib --reps 3 > $O/pmc_k_pass$i.log 2>&1 || echo "pmc pass $i failed"
  i=$((i+1))
done
i=0
for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  for sh in 8192_8192_8192 16384_16384_16384; do
    timeout 90 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_hipblaslt/${sh}_pass$i -- $T bench --shape $sh --baseline hipblaslt_tn --isolated --reps 3 > $O/pmc_hipblaslt_${sh}_pass$i.log 2>&1 || echo "pmc hipblaslt $sh pass $i failed"
  done
  i=$((i+1))
done
find $O -name "*agent_info.csv" -delete
# (4) A/B against the round-4 library
for rep in 1 2 3; do
  for sh in 4096_4096_4096 16384_16384_256; do
    LD_LIBRARY_PATH=cuda-l2_amd/lib_r4 timeout 60 $T bench --shape $sh --config q256x256_w2x2 --splits 131073 --group 4 --power --seconds 1.0 > $O/ab_r4_${sh}_$rep.log 2>&1
    timeout 60 $T bench --shape $sh --config q256x256_w2x2 --splits 131073 --group 4 --power --seconds 1.0 > $O/ab_r5_${sh}_$rep.log 2>&1
  done
done
for f in $O/ab_*.log; do echo "$f $(grep -o '"us": [0-9.]*\|gfx_mhz_mean": [0-9.]*\|socket_w_mean": [0-9.]*' $f | tr '\n' ' ')"; done
du -sh $O
