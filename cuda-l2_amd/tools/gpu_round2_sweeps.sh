#!/bin/bash
# Round-2 sweeps in the reference metric (in-process driver, one GPU): fp32 offline on the full grid; fp16 offline,
# fp32 / fp16 server (qps 100) on the quarter grid; qps sweep of BASELINE config 4.
set -u
S=gpurun_out/sweep_r02; mkdir -p $S
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 offline tools/grid_shapes.txt $W --cpu_max_flops 2e10 --cpu_seconds 0.02 --time_limit 620 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 offline tools/grid_shapes_quarter.txt $W --time_limit 200 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 server tools/grid_shapes_quarter.txt $W --target_qps 100 --time_limit 240 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 server tools/grid_shapes_quarter.txt $W --target_qps 100 --time_limit 240 2>&1 | tail -1
printf "512_4096_4096\n" > cuda-l2_amd/tools/.cfg4.txt
for q in 10 100 1000; do
  bash cuda-l2_amd/tools/gpu_sweep.sh gpurun_out/sweep_r02/qps_$q fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 2.5 --target_qps $q 2>&1 | tail -1
done
rm -f cuda-l2_amd/tools/.cfg4.txt
