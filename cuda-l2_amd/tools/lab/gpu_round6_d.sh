#!/bin/bash
# Round-6 call D: does the hardware double up the workgroups of a two-resident member when the launch has no more workgroups than
# CUs?  The shipping library spreads such launches (dynamic LDS: one workgroup per CU, csrc/hgemm_launch.hpp: sq_dynamic_lds);
# lib_nospread/ is the same source without it (-DHGEMM_SQ_ONE_PER_CU=0).  Check first (the launches changed), then per shape the
# plan on both libraries, interleaved A B A B, isolated and back to back.
set -u
O=gpurun_out/r6d; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 600 $T check --configs q128x128_w2x2,q192x128_w2x2,q128x192_w2x2 > $O/check_two_resident.log 2>&1; echo "check rc=$? $(tail -1 $O/check_two_resident.log)"
grep -q " 0 failures" $O/check_two_resident.log || { echo "CHECK FAILED"; grep FAIL $O/check_two_resident.log | head; exit 1; }
timeout 300 $T check --shapes 2304_2304_1024,1024_2048_2048 --configs q128x128_w2x2,q192x128_w2x2,q128x192_w2x2 > $O/check_two_resident_walk.log 2>&1; echo "check walk rc=$? $(tail -1 $O/check_two_resident_walk.log)"
: > $O/ab.jsonl
for rep in 1 2 3; do
for spec in "1024_2048_12288 q128x128_w2x2 2 4" "2048_1024_4096 q128x128_w2x2 1 4" "512_4096_4096 q128x128_w2x2 524290 4" "512_4096_4096 q128x128_w2x2 589826 4" \
            "2048_2048_2048 q128x128_w2x2 655361 4" "1024_2048_4096 q128x128_w2x2 65538 4" "16384_128_16384 q128x128_w2x2 8 4" "128_8192_4096 q128x128_w2x2 4 4" \
            "512_4096_12288 q128x128_w2x2 65538 4" "1536_2048_4096 q192x128_w2x2 2 4"; do
  set -- $spec
  for lib in lib lib_nospread; do
    LD_LIBRARY_PATH=cuda-l2_amd/$lib timeout 60 $T bench --shape $1 --config $2 --splits $3 --group $4 --reps 30 | sed "s/^{/{\"lib\": \"$lib\", \"clock\": \"isolated\", /" >> $O/ab.jsonl
    LD_LIBRARY_PATH=cuda-l2_amd/$lib timeout 60 $T bench --shape $1 --config $2 --splits $3 --group $4 --power --seconds 0.3 | sed "s/^{/{\"lib\": \"$lib\", \"clock\": \"stream\", /" >> $O/ab.jsonl
  done
done
done
wc -l $O/ab.jsonl
