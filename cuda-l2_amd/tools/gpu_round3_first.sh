#!/bin/bash
# Round-3 opening GPU call.  Build the variant library on the CPU side first (it travels with the snapshot):
#   (cd cuda-l2_amd && HGEMM_LIB_SUFFIX=fd HGEMM_EXTRA_HIPFLAGS="-DHGEMM_FASTDIV=1" python build.py)
# 1. the shipped state (GPU suite), 2. the multiplier raster map: exactness of every geometry x
# split-K form, whole-grid parity with the variant library, 3. A/B against the shipped library in both timing modes
# (stream + telemetry, one launch at a time) on the small-K class, a launch-bound shape and the bench shapes.
set -u
O=gpurun_out/r3a; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -5
# off-grid shapes at the plans the neighbour planner gives them: parity against the oracle, then device time against the vendor libraries
timeout 300 python tests/tools/verify_plans.py --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/parity_offgrid.jsonl | tail -1
timeout 300 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; tail -2 $O/offgrid_plan_report.log
if [ -f cuda-l2_amd/lib_fd/libhgemm_mi355x.so ]; then
  LD_LIBRARY_PATH=$PWD/cuda-l2_amd/lib_fd timeout 120 $T check --shapes 328_456_1024,4352_4352_320,512_768_128,256_256_8192 | tail -3
  HGEMM_LIB_DIR=$PWD/cuda-l2_amd/lib_fd timeout 300 python tests/tools/verify_plans.py --out $O/parity_fastdiv.jsonl | tail -1
  for sh in 8192_16384_256 16384_16384_256 4096_8192_128 8192_8192_128 64_4096_64 1024_1024_64 512_4096_4096 4096_4096_4096; do
    for v in lib lib_fd; do
      echo "# $v"
      LD_LIBRARY_PATH=$PWD/cuda-l2_amd/$v timeout 20 $T bench --shape $sh --lib --power --seconds 0.5
      LD_LIBRARY_PATH=$PWD/cuda-l2_amd/$v timeout 20 $T bench --shape $sh --lib --reps 200
    done
  done 2>&1 | tee $O/fastdiv_ab.jsonl
fi
