#!/bin/bash
# Round-2 profiles: bench.py (full JSON line), rocprofv3 kernel stats of the same command, PMC passes for the three
# BASELINE shapes at their shipped plans.  Summaries are copied into profiles/ afterwards.
set -u
O=gpurun_out/r2g; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
for mnk in 4096_4096_4096 512_4096_4096 64_4096_64; do
  bash cuda-l2_amd/tools/pmc_sweep.sh $O/pmc_$mnk -- $T bench --shape $mnk --lib --reps 12 > /dev/null 2>&1
  mkdir -p $O/pmc_$mnk; ls $O/pmc_$mnk | head -3
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O/prof -name "*kernel_trace.csv" -size +20M -delete; du -sh $O
