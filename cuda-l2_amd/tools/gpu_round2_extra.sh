#!/bin/bash
# Round-2 extras: PMC passes of the streaming kernel on 16384x64x16384, ragged shapes and off-grid shapes vs hipBLASLt.
set -u
O=gpurun_out/r2x; mkdir -p $O/pmc_16384_64_16384
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
bash cuda-l2_amd/tools/pmc_sweep.sh $O/pmc_16384_64_16384 -- $T bench --shape 16384_64_16384 --lib --reps 10 > /dev/null 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete
$T tune --plan-only --baselines --shapes 1000_520_200,65_30_100,33_17_40,2100_2050_328,4000_4000_4000,1000_520_192 --out $O/ragged_vs_baselines.jsonl > $O/ragged.log 2>&1; echo "ragged rc=$?"
$T tune --plan-only --baselines --shapes 3072_5120_7168,3584_18944_3584,7168_7168_7168,6144_6144_6144,4352_4352_4096,5120_5120_5120,1536_6144_4096,2560_10240_2560,10000_10000_1024,768_3072_768,3072_768_3072,8192_28672_8192,1280_1280_8192,20480_64_5120 --out $O/offgrid_vs_baselines.jsonl > $O/offgrid.log 2>&1; echo "offgrid rc=$?"
