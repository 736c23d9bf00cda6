#!/bin/bash
# Round-3 GPU call A: where does the compute-bound kernel's time go?
#  1. in-kernel timeline (lib_tl*: head / K loop / epilogue / drain per workgroup, launch gaps) of the shipped q kernel and of
#     its ablations (no barrier / no vmcnt / no syncs / no DMA / no fragment reads / MFMA only), NT stores, multiplier raster map
#  2. back-to-back stream A/B of the variant libraries against hipBLASLt
#  3. PMC passes of our kernel and of hipBLASLt's on the same shapes
#  4. what round 2 shipped blind: FASTDIV exactness + whole-grid parity, off-grid planner parity + report
set -u
O=gpurun_out/r3a; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
export TMPDIR=/tmp
echo "== check"; timeout 120 $T check --configs q256x256_w2x2,q128x128_w2x2_k128 --shapes 512_768_512,4096_4096_256 2>&1 | tail -2
echo "== timeline"
for sh in 4096_4096_4096 8192_8192_8192; do
  for v in tl tl_a1 tl_a2 tl_a7 tl_a8 tl_a16 tl_a31 tl_nt tl_fd; do
    echo "# $v $sh"
    LD_LIBRARY_PATH=$P/lib_$v timeout 60 $T bench --shape $sh --config q256x256_w2x2 --group 8 --timeline
  done
done > $O/timeline.jsonl 2>&1
grep -c timeline $O/timeline.jsonl
echo "== stream A/B"
for rep in 1 2; do
  for sh in 4096_4096_4096 8192_8192_8192 4096_4096_1024 2048_8192_8192; do
    for v in lib lib_nt lib_fd lib_ntfd; do
      echo "# $v"
      LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --lib --power --seconds 1.0
    done
    echo "# hipblaslt"
    timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 1.0
  done
done > $O/stream_ab.jsonl 2>&1
grep -c stream $O/stream_ab.jsonl
echo "== pmc"
for sh in 4096_4096_4096 8192_8192_8192; do
  mkdir -p $O/pmc_ours_$sh $O/pmc_hbl_$sh
  bash cuda-l2_amd/tools/pmc_sweep_mem.sh $O/pmc_ours_$sh -- $T bench --shape $sh --lib --reps 12
  bash cuda-l2_amd/tools/pmc_sweep_mem.sh $O/pmc_hbl_$sh -- $T bench --shape $sh --baseline hipblaslt_tn --isolated --reps 12
done
find $O -name "*_counter_collection.csv" | wc -l
echo "== fastdiv"
LD_LIBRARY_PATH=$P/lib_fd timeout 120 $T check --shapes 328_456_1024,4352_4352_320,512_768_128,256_256_8192 2>&1 | tail -3
HGEMM_LIB_DIR=$P/lib_fd timeout 300 python tests/tools/verify_plans.py --out $O/parity_fastdiv.jsonl 2>&1 | tail -1
echo "== offgrid"
timeout 300 python tests/tools/verify_plans.py --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/parity_offgrid.jsonl 2>&1 | tail -1
timeout 300 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; tail -2 $O/offgrid_plan_report.log
# keep the merged-back payload small: counter CSVs only
find $O -name "*_agent_info.csv" -delete; du -sh $O
