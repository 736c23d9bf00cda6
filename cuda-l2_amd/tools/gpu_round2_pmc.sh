#!/bin/bash
# PMC passes (tools/pmc_sweep.sh) for the three BASELINE shapes at their shipped plans.
set -u
O=gpurun_out/r2g; T=cuda-l2_amd/bin/hgemm_tune
for mnk in 4096_4096_4096 512_4096_4096 64_4096_64; do
  mkdir -p $O/pmc_$mnk
  bash cuda-l2_amd/tools/pmc_sweep.sh $O/pmc_$mnk -- $T bench --shape $mnk --lib --reps 12 2>&1 | tail -2
  ls $O/pmc_$mnk | head -12
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; du -sh $O
