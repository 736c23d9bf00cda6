#!/bin/bash
# Round-3 GPU call D: NT-store plan flag, multiplier raster map, head split, the small-K / large-MN class
set -u
O=gpurun_out/r3d; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
export TMPDIR=/tmp
NT1=131073   # splits = 1 | HGEMM_PLAN_NT_STORE
echo "== check"; timeout 300 $T check --configs q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q256x256_w2x2_m32,s256x256_w2x2 2>&1 | tail -1
LD_LIBRARY_PATH=$P/lib_fd timeout 300 $T check --configs q256x256_w2x2,q128x128_w2x2_k128,s256x256_w2x2,t128x128_w2x2_m16_s2 2>&1 | tail -1
echo "== timeline (head split)"
for sh in 4096_4096_4096 16384_16384_256; do
  for v in tl tl_fd; do
    echo "# $v $sh"; LD_LIBRARY_PATH=$P/lib_$v timeout 60 $T bench --shape $sh --config q256x256_w2x2 --group 8 --timeline
    echo "# ${v}_nt $sh"; LD_LIBRARY_PATH=$P/lib_$v timeout 60 $T bench --shape $sh --config q256x256_w2x2 --group 8 --splits $NT1 --timeline
  done
done > $O/timeline.jsonl 2>&1
grep -c timeline $O/timeline.jsonl
echo "== stream A/B: plan flag NT, fastdiv, 2 workgroups per CU on the small-K class"
for rep in 1 2; do
  for sh in 4096_4096_4096 8192_8192_8192 4096_4096_1024 2048_8192_8192 16384_16384_256 8192_16384_256 4096_8192_128 16384_16384_512; do
    for v in lib lib_fd; do
      echo "# $v q256"; LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --config q256x256_w2x2 --group 8 --power --seconds 0.6
      echo "# $v q256_nt"; LD_LIBRARY_PATH=$P/$v timeout 30 $T bench --shape $sh --config q256x256_w2x2 --group 8 --splits $NT1 --power --seconds 0.6
    done
    echo "# lib q128"; timeout 30 $T bench --shape $sh --config q128x128_w2x2 --group 8 --power --seconds 0.6
    echo "# lib q128_nt"; timeout 30 $T bench --shape $sh --config q128x128_w2x2 --group 8 --splits $NT1 --power --seconds 0.6
    echo "# lib t128"; timeout 30 $T bench --shape $sh --config t128x128_w2x2_m16_s2 --group 4 --power --seconds 0.6
    echo "# lib t128_nt"; timeout 30 $T bench --shape $sh --config t128x128_w2x2_m16_s2 --group 4 --splits $NT1 --power --seconds 0.6
    echo "# hipblaslt"; timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.6
  done
done > $O/stream_ab.jsonl 2>&1
grep -c stream $O/stream_ab.jsonl
echo "== tune --nt on a sample (isolated time of the plan, back-to-back plain vs NT)"
timeout 400 $T tune --nt --max-cand 4 --shapes 4096_4096_4096,4096_4096_1024,2048_2048_2048,8192_16384_256,16384_16384_256,1024_4096_4096,512_4096_4096,2048_8192_8192,8192_8192_64,4096_4096_256,1024_1024_1024,16384_4096_128 --out $O/tune_nt_sample.jsonl > $O/tune_nt_sample.log 2>&1; tail -1 $O/tune_nt_sample.log
echo "== pmc small-K"
sh=16384_16384_256
mkdir -p $O/pmc_ours_$sh $O/pmc_hbl_$sh
bash cuda-l2_amd/tools/pmc_sweep_mem.sh $O/pmc_ours_$sh -- $T bench --shape $sh --config q256x256_w2x2 --group 8 --reps 8
bash cuda-l2_amd/tools/pmc_sweep_mem.sh $O/pmc_hbl_$sh -- $T bench --shape $sh --baseline hipblaslt_tn --isolated --reps 8
find $O -name "*_agent_info.csv" -delete; du -sh $O
