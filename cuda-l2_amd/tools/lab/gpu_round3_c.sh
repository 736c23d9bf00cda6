#!/bin/bash
# Round-3 GPU call C: the one-instruction-per-gap form of family q's K loop (HGEMM_SQ_GAPS).
set -u
O=gpurun_out/r3c; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
export TMPDIR=/tmp
echo "== check"; timeout 300 $T check --configs q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q256x256_w2x2_m32 2>&1 | tail -2
echo "== timeline"
for sh in 4096_4096_4096 8192_8192_8192 8192_8192_256; do
  for v in tl tl_g0 tl_nt tl_a8 tl_a1 tl_a16; do
    echo "# $v $sh"
    LD_LIBRARY_PATH=$P/lib_$v timeout 60 $T bench --shape $sh --config q256x256_w2x2 --group 8 --timeline
  done
done > $O/timeline.jsonl 2>&1
grep -c timeline $O/timeline.jsonl
echo "== stream A/B"
for rep in 1 2; do
  for sh in 4096_4096_4096 8192_8192_8192 4096_4096_1024 2048_8192_8192 16384_16384_256 8192_16384_256 2048_2048_2048 1024_4096_4096; do
    for v in lib lib_g0 lib_r2 lib_nt; do
      echo "# $v"
      LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --lib --power --seconds 0.7
    done
    echo "# hipblaslt"
    timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.7
  done
done > $O/stream_ab.jsonl 2>&1
grep -c stream $O/stream_ab.jsonl
echo "== race screen + parity tests"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
du -sh $O
