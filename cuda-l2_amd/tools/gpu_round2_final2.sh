#!/bin/bash
# Round-2 final artefacts with the shipped binary + table: whole-grid parity, bench.py, rocprofv3 kernel stats,
# PMC passes for the BASELINE shapes, then the reference-metric sweeps.
set -u
O=gpurun_out/r2y; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 600 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl > $O/verify.log 2>&1; echo "verify rc=$?"; tail -1 $O/verify.log
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
for mnk in 4096_4096_4096 512_4096_4096 64_4096_64; do
  mkdir -p $O/pmc_$mnk
  bash cuda-l2_amd/tools/pmc_sweep.sh $O/pmc_$mnk -- $T bench --shape $mnk --lib --reps 12 > /dev/null 2>&1
done
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O/prof -name "*kernel_trace.csv" -delete
rm -rf gpurun_out/sweep_r02
bash cuda-l2_amd/tools/gpu_round2_sweeps.sh
