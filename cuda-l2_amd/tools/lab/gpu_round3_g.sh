#!/bin/bash
# Round-3 GPU call G: (1) hipBLASLt autotune with a REAL budget (1 s per layout: every candidate, >= 20 timed rounds at
# 4096^3) on every 8th grid shape, both accumulate trees, offline; (2) the reference-faithful flow -- eval_one_file.sh: one
# process per baseline, correctness check first -- on a stratified 8-shape subset, next to the in-process driver on the same
# shapes with the same time boxes; (3) BASELINE config 4 (512x4096x4096 fp32, server mode) at qps 10 / 100 / 1000 with
# >= 1000 recorded samples each.
set -u
# (0) chip-filling split-K factors (tools/make_fill_candidates.py) for the under-filled shapes: targeted re-tune + oracle check
O=gpurun_out/r3g; mkdir -p $O
awk '{print $1}' cuda-l2_amd/tuning/r03_fill_candidates.txt > $O/fill_shapes.txt
timeout 600 cuda-l2_amd/bin/hgemm_tune tune --shape-file $O/fill_shapes.txt --cand-file cuda-l2_amd/tuning/r03_fill_candidates.txt --nt --baselines --out $O/fill_tune.jsonl > $O/fill_tune.log 2>&1
echo "fill tune rc=$? lines=$(wc -l < $O/fill_tune.jsonl)"
timeout 300 python tests/tools/verify_plans.py --plans $O/fill_tune.jsonl --top 3 --repeats 2 --out $O/fill_verify.jsonl > $O/fill_verify.log 2>&1; tail -1 $O/fill_verify.log
S=gpurun_out/sweep_r03; mkdir -p $S
awk 'NR % 8 == 1' cuda-l2_amd/tools/grid_shapes.txt > cuda-l2_amd/tools/.eighth.txt
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
bash cuda-l2_amd/tools/gpu_sweep.sh $S/autotune_1s fp32 offline tools/.eighth.txt $W --time_limit 420 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S/autotune_1s fp16 offline tools/.eighth.txt $W --time_limit 420 2>&1 | tail -1
export HGEMM_AUTOTUNE_MAX_SECONDS=0.05
printf "64_4096_64\n512_4096_4096\n4096_4096_4096\n1024_1024_1024\n256_16384_2048\n8192_8192_256\n128_128_8192\n2048_8192_8192\n" > cuda-l2_amd/tools/.subset8.txt
( cd cuda-l2_amd && python tools/sweep.py run --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset8.txt --warmup_seconds 1 --benchmark_seconds 2 \
  && python tools/sweep.py merge --out ../$S/process_per_baseline --acc_precise fp32 --mode offline --shapes-file tools/.subset8.txt > ../$S/process_per_baseline/merge_fp32_offline.json ) 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S/inprocess_same_boxes fp32 offline tools/.subset8.txt --warmup_seconds 1 --benchmark_seconds 2 2>&1 | tail -1
printf "512_4096_4096\n" > cuda-l2_amd/tools/.cfg4.txt
bash cuda-l2_amd/tools/gpu_sweep.sh $S/qps_10 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 100 --target_qps 10 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S/qps_100 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 12 --target_qps 100 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S/qps_1000 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 8 --target_qps 1000 2>&1 | tail -1
rm -f cuda-l2_amd/tools/.cfg4.txt cuda-l2_amd/tools/.eighth.txt cuda-l2_amd/tools/.subset8.txt
du -sh $S
