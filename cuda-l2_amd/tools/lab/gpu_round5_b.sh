#!/bin/bash
# Round-5 call B: does de-synchronising the epilogues of an XCD's workgroups turn "store phase + MFMA phase" into max() of the two?
# Two plan flags (host + kernel prologue only; the K loops and epilogues are call A's, exact there): HGEMM_PLAN_PHASE_OFFSET (half of an
# XCD's workgroups start half an item period late) on the shipped 256-wide plans, HGEMM_PLAN_WAVE_PRIORITY (static priorities for the
# two waves of a SIMD) on the two-resident members.  Exactness of the flagged launches first (results must be bit-identical), then
# timing; then FETCH / WRITE counters of 16384^3 at raster groups 2 (shipped), 4 and 8.
set -u
O=gpurun_out/r5b; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 400 $T check --shapes 512_1024_2048,2304_2304_1024,4224_4096_512,300_260_2104 --configs $QCFG > $O/check_q_flags.log 2>&1; echo "check rc=$? $(tail -1 $O/check_q_flags.log)"
grep -q " 0 failures" $O/check_q_flags.log || { echo "CHECK FAILED"; grep FAIL $O/check_q_flags.log | head; exit 1; }
timeout 500 $T tune --shape-file cuda-l2_amd/tuning/r05_phase_first_look_shapes.txt --cand-file cuda-l2_amd/tuning/r05_phase_first_look_candidates.txt --rank both --baselines --stream --keep 10 --out $O/phase.jsonl > $O/phase.log 2>&1; echo "phase rc=$? lines=$(wc -l < $O/phase.jsonl)"
for g in 2 4 8; do
  i=0
  for p in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
    timeout 90 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $O/pmc_g$g/pass$i -- $T bench --shape 16384_16384_16384 --config q256x256_w2x2 --splits 131073 --group $g --reps 3 > $O/pmc_g${g}_pass$i.log 2>&1 || echo "pmc g$g pass $i failed"
    i=$((i+1))
  done
done
find $O -name "*agent_info.csv" -delete
du -sh $O
