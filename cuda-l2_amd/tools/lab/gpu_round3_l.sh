#!/bin/bash
# Round-3 call L: the two GPU tests touched after call K (sample selection fixed; off-grid planner restricted to its fitted
# domains), the off-grid plan report of the final planner, and an A/B of the per-XCD K stagger of family q (-DHGEMM_SQ_XSTAGGER=1,
# lib_xs/) on the HBM-streaming shapes whose time varied 160 -> 240 us between boxes while hipBLASLt's did not.
set -u
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid.py -m gpu -q -k "sample_of_grid or off_grid or race_screen" 2>&1 | tail -3
timeout 300 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
{
for sh in 16384_256_16384 256_16384_16384 12288_128_8192 12288_128_16384 128_12288_16384 4096_4096_4096 8192_8192_8192 16384_16384_256 4096_4096_8192; do
  for lib in lib lib_xs; do
    echo "# $lib isolated"; LD_LIBRARY_PATH=$P/$lib timeout 30 $T bench --shape $sh --lib --reps 20
    echo "# $lib stream";   LD_LIBRARY_PATH=$P/$lib timeout 30 $T bench --shape $sh --lib --power --seconds 0.3
  done
  echo "# hipblaslt stream"; timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.3
done
} > $O/xstagger_ab.jsonl 2>&1
du -sh $O
