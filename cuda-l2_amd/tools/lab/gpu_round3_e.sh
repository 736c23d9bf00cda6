#!/bin/bash
# Round-3 GPU call E: kernel-argument prefetch + interleaved prologue (A/B against the previous build), then the targeted
# re-tune of the whole grid (candidates from the round-2 runs + family q's members, NT-store trial for the winner) and the
# oracle verification of its fastest candidates.
set -u
O=gpurun_out/r3e; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
export TMPDIR=/tmp
NT1=131073
echo "== check"; timeout 300 $T check --configs q256x256_w2x2,q128x128_w2x2_k128,s256x256_w2x2,t128x128_w2x2_m16_s2,t64x64_w2x2_m16_s4,r64x64_k256 2>&1 | tail -1
echo "== A/B"
for rep in 1 2; do
  for sh in 4096_4096_4096 4096_4096_1024 1024_1024_1024 64_4096_64 256_256_256 2048_8192_8192; do
    for v in lib lib_prev; do
      echo "# $v lib"; LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --lib --power --seconds 0.5
      echo "# $v isolated"; LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --lib --reps 200
    done
  done
  for v in lib lib_prev; do echo "# $v q256nt"; LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape 4096_4096_4096 --config q256x256_w2x2 --group 8 --splits $NT1 --power --seconds 0.6; done
  echo "# hipblaslt"; timeout 30 $T bench --shape 4096_4096_4096 --baseline hipblaslt_tn --seconds 0.6
done > $O/ab.jsonl 2>&1
grep -c mnk $O/ab.jsonl
echo "# tl"; LD_LIBRARY_PATH=$P/lib_tl timeout 60 $T bench --shape 4096_4096_4096 --config q256x256_w2x2 --group 8 --splits $NT1 --timeline > $O/timeline.jsonl 2>&1
echo "== tune"
timeout 1500 $T tune --shape-file cuda-l2_amd/tools/grid_shapes.txt --cand-file cuda-l2_amd/tuning/r03_retune_candidates.txt --nt --out $O/grid_tune.jsonl > $O/tune.log 2>&1
echo "tune rc=$? lines=$(wc -l < $O/grid_tune.jsonl)"
timeout 600 python tests/tools/verify_plans.py --plans $O/grid_tune.jsonl --top 3 --repeats 2 --out $O/verify_candidates.jsonl > $O/verify.log 2>&1
echo "verify rc=$?"; tail -1 $O/verify.log
du -sh $O
