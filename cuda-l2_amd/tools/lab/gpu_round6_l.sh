#!/bin/bash
# Round-6 calls L1 / L2: the reference-metric sweeps of the SHIPPED (round-6) table on the whole grid, in-process driver, with the REAL
# hipBLASLt autotune: winners of the 1 s-per-layout search from the on-disk cache (the torch wheel's hipBLASLt build; problems the
# cache does not hold yet are searched here, 1 s per layout, and appended).  usage: gpu_round6_l.sh ACC [with_config4]
#   boxes: 0.05 s warm-up + 0.25 s recorded per shape (>= 3 rounds of all seven baselines); server mode: target_qps 100.
set -u
ACC=$1
S=gpurun_out/r6l; mkdir -p $S
export TMPDIR=/tmp
export HGEMM_AUTOTUNE_CACHE=$PWD/$S/r06_hipblaslt_autotune_cache.txt
[ -f $HGEMM_AUTOTUNE_CACHE ] || cp cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt $HGEMM_AUTOTUNE_CACHE
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
W="--warmup_seconds 0.05 --benchmark_seconds 0.25"
if [ "$ACC" == "fp32" ]; then CPU="--cpu_max_flops 2e10 --cpu_seconds 0.02"; else CPU=""; fi
bash cuda-l2_amd/tools/gpu_sweep.sh $S $ACC offline tools/grid_shapes.txt $W $CPU --time_limit 1300 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S $ACC server tools/grid_shapes.txt $W --target_qps 100 --time_limit 1300 2>&1 | tail -1
if [ "${2:-}" == "with_config4" ]; then
  # BASELINE config 4 (512x4096x4096 fp32, server mode) at qps 10 / 100 / 1000 with >= 1000 samples of cuda_l2 each
  echo 512_4096_4096 > cuda-l2_amd/tools/.cfg4.txt
  bash cuda-l2_amd/tools/gpu_sweep.sh $S/config4/qps_10 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 100 --target_qps 10 2>&1 | tail -1
  bash cuda-l2_amd/tools/gpu_sweep.sh $S/config4/qps_100 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 12 --target_qps 100 2>&1 | tail -1
  bash cuda-l2_amd/tools/gpu_sweep.sh $S/config4/qps_1000 fp32 server tools/.cfg4.txt --warmup_seconds 0.5 --benchmark_seconds 8 --target_qps 1000 2>&1 | tail -1
  rm -f cuda-l2_amd/tools/.cfg4.txt
fi
find $S -name "*.so" -delete 2>/dev/null; find $S -name "*.o" -delete 2>/dev/null; du -sh $S; wc -l $HGEMM_AUTOTUNE_CACHE
