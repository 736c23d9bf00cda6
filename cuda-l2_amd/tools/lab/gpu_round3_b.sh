#!/bin/bash
# Round-3 GPU call B: the spread LDS-DMA plan (HGEMM_SQ_SPREAD) and the LDS-staged epilogue (HGEMM_EPI_STAGED).
#  1. exactness of every geometry x split-K form with the new default library, the GPU test suite
#  2. timeline of new default / spread only / staged only / round-2 plan / NT stores
#  3. back-to-back stream A/B against the round-2 plan and hipBLASLt on the compute-bound and the small-K classes
set -u
O=gpurun_out/r3b; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
export TMPDIR=/tmp
echo "== check"; timeout 400 $T check 2>&1 | tail -2
echo "== timeline"
for sh in 4096_4096_4096 8192_8192_8192 8192_8192_256; do
  for v in tl tl_r2 tl_sp tl_st tl_nt; do
    echo "# $v $sh"
    LD_LIBRARY_PATH=$P/lib_$v timeout 60 $T bench --shape $sh --config q256x256_w2x2 --group 8 --timeline
  done
done > $O/timeline.jsonl 2>&1
grep -c timeline $O/timeline.jsonl
echo "== stream A/B"
for rep in 1 2; do
  for sh in 4096_4096_4096 8192_8192_8192 4096_4096_1024 2048_8192_8192 16384_16384_256 8192_16384_256 4096_8192_128 2048_2048_2048 1024_4096_4096; do
    for v in lib lib_r2 lib_sp lib_st lib_nt; do
      echo "# $v"
      LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --lib --power --seconds 0.7
    done
    echo "# hipblaslt"
    timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.7
  done
done > $O/stream_ab.jsonl 2>&1
grep -c stream $O/stream_ab.jsonl
echo "== pytest"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
du -sh $O
