#!/bin/bash
# Round-3 GPU call F: the reference-metric sweeps on the WHOLE grid for both accumulate trees and both modes (in-process
# driver, time boxes as in round 2 and stated in every record), then the device-time plan report against the vendor libraries.
set -u
S=gpurun_out/sweep_r03; mkdir -p $S
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 offline tools/grid_shapes.txt $W --cpu_max_flops 2e10 --cpu_seconds 0.02 --time_limit 560 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 offline tools/grid_shapes.txt $W --time_limit 520 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
O=gpurun_out/r3f; mkdir -p $O
timeout 300 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl 2>&1 | tail -1
timeout 600 cuda-l2_amd/bin/hgemm_tune tune --plan-only --baselines --shape-file cuda-l2_amd/tools/grid_shapes.txt --out $O/grid_plan_report.jsonl > $O/grid_plan_report.log 2>&1
echo "plan report lines=$(wc -l < $O/grid_plan_report.jsonl)"
du -sh $S $O
