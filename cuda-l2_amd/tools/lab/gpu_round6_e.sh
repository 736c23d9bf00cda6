#!/bin/bash
# Round-6 call E: pass 1 / 2 read the two-resident members of family q 8 % slower than round 5's closing report while every other
# family read level: box, or the HGEMM_PLAN_CU_PHASE prologue (same register counts, another instruction stream)?  Three libraries on
# one box, interleaved, isolated and back to back: lib_r5/ = round 5's closing commit (4a082db8), lib_cuphase/ = the commit of calls
# B - D (with the prologue), lib/ = this tree (prologue removed: the two-resident plain / slab kernels are round 5's again by
# fingerprint).  The tool is this tree's; it only calls the C ABI.
set -u
O=gpurun_out/r6e; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
: > $O/ab.jsonl
for rep in 1 2 3; do
for spec in "1024_2048_12288 q128x128_w2x2 2 4" "128_12288_8192 q128x128_w2x2 LIB 0" "64_12288_8192 q128x128_w2x2 LIB 0" "128_4096_12288 q128x128_w2x2 LIB 0" \
            "16384_128_16384 q128x128_w2x2 LIB 0" "512_4096_4096 q128x128_w2x2 524290 4" "512_8192_12288 q128x128_w2x2_k128 1 4" "4096_16384_64 q192x128_w2x2 LIB 0" \
            "4096_4096_4096 q256x256_w2x2 LIB 0" "256_12288_4096 q LIB 0"; do
  set -- $spec
  for lib in lib_r5 lib_cuphase lib; do
    if [ "$3" == "LIB" ]; then A="--lib"; else A="--config $2 --splits $3 --group $4"; fi
    LD_LIBRARY_PATH=cuda-l2_amd/$lib timeout 60 $T bench --shape $1 $A --reps 30 | sed "s/^{/{\"lib\": \"$lib\", \"clock\": \"isolated\", /" >> $O/ab.jsonl
    LD_LIBRARY_PATH=cuda-l2_amd/$lib timeout 60 $T bench --shape $1 $A --power --seconds 0.3 | sed "s/^{/{\"lib\": \"$lib\", \"clock\": \"stream\", /" >> $O/ab.jsonl
  done
done
done
wc -l $O/ab.jsonl
