#!/bin/bash
# Round-4 call E: the reference-metric sweeps of the SHIPPED (round-4) table on the whole grid, both accumulate trees, offline and
# server (in-process driver, the boxes of round 3, stated in every record); then the reference-faithful flow (eval_one_file.sh:
# correctness check first, ONE PROCESS PER BASELINE) on a FLOP-stratified subset next to the in-process driver with the same
# boxes, so the in-process / process ratio can be read per FLOP decade.
set -u
S=gpurun_out/sweep_r04; mkdir -p $S
export TMPDIR=/tmp
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 offline tools/grid_shapes.txt $W --cpu_max_flops 2e10 --cpu_seconds 0.02 --time_limit 560 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 offline tools/grid_shapes.txt $W --time_limit 520 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
# one or two shapes per FLOP decade 10^6 .. 10^12, the BASELINE shapes among them: 10 shapes (the flow costs ~66 s per shape)
printf "64_64_128\n64_4096_64\n512_256_512\n1024_1024_512\n256_4096_1024\n128_128_8192\n512_4096_4096\n4096_4096_4096\n1024_16384_4096\n8192_8192_8192\n" > cuda-l2_amd/tools/.subset14.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=0.05
( cd cuda-l2_amd && timeout 1100 python tools/sweep.py run --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset14.txt --warmup_seconds 1 --benchmark_seconds 2 \
  ; python tools/sweep.py merge --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset14.txt > ../$S/process_per_baseline/merge_fp32_offline.json ) 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S/inprocess_same_boxes fp32 offline tools/.subset14.txt --warmup_seconds 1 --benchmark_seconds 2 2>&1 | tail -1
rm -f cuda-l2_amd/tools/.subset14.txt
find $S -name "*.so" -delete 2>/dev/null; find $S -name "*.o" -delete 2>/dev/null; du -sh $S
