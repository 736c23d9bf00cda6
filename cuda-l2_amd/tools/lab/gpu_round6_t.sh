#!/bin/bash
# Round-6 call T: the reference-faithful flow (eval_one_file.sh: correctness check first, then ONE PROCESS PER BASELINE, 1 s warm-up + 2 s recorded
# each) on a FLOP-stratified 10-shape subset next to the in-process driver with the same boxes -- the in-process / process ratio per FLOP decade for
# the round-6 sweeps (round 4 did the same for its table).  hipBLASLt autotune from the cache in every process.
set -u
S=gpurun_out/r6t; mkdir -p $S
export TMPDIR=/tmp
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
printf "64_64_128\n64_4096_64\n512_256_512\n1024_1024_512\n256_4096_1024\n128_128_8192\n512_4096_4096\n4096_4096_4096\n1024_16384_4096\n8192_8192_8192\n" > cuda-l2_amd/tools/.subset10.txt
( cd cuda-l2_amd && timeout 1500 python tools/sweep.py run --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset10.txt --warmup_seconds 1 --benchmark_seconds 2 \
  ; python tools/sweep.py merge --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset10.txt > ../$S/process_per_baseline/merge_fp32_offline.json ) 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S/inprocess_same_boxes fp32 offline tools/.subset10.txt --warmup_seconds 1 --benchmark_seconds 2 2>&1 | tail -1
rm -f cuda-l2_amd/tools/.subset10.txt
find $S -name "*.so" -delete 2>/dev/null; find $S -name "*.o" -delete 2>/dev/null; find $S -name "build" -type d -prune -exec rm -rf {} + 2>/dev/null; du -sh $S
