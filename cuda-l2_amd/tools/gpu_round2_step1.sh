#!/bin/bash
# Round-2 GPU step 1: parity of the new kernels (pytest -m gpu, hgemm_tune check), A/B timings
# (SP 16x16 vs 32x32 MFMA; single-launch vs two-pass split-K), then the whole-grid plan verification.
# Usage (from the repo root, on the GPU box): bash cuda-l2_amd/tools/gpu_round2_step1.sh
set -u
O=gpurun_out/r2a; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
( timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ) 
tail -5 $O/pytest.log
timeout 300 $T check > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log; tail -3 $O/check.log
# A/B 1: SP family, 16x16x32 vs 32x32x16 MFMA, with the vendor baselines
timeout 400 $T tune --shapes 4096_4096_4096,8192_8192_8192,4096_4096_1024,8192_8192_1024,8192_4096_4096,4096_4096_16384,16384_16384_1024,2048_2048_8192,4096_8192_2048 \
   --configs s256x256_w2x2,s256x256_w2x2_m32 --keep 100 --max-cand 4 --baselines --out $O/ab_sp.jsonl > $O/ab_sp.log 2>&1
# A/B 2: split-K forms on a sample of the shipped split-K plans (all geometries compete)
timeout 500 $T tune --shapes 64_64_2048,128_512_2048,1024_128_2048,128_256_4096,128_2048_2048,256_128_8192,512_256_4096,256_64_16384,2048_128_4096,256_256_12288,128_1024_8192,128_2048_8192,4096_128_4096,2048_64_12288,128_4096_8192,2048_64_16384,1024_512_12288,256_2048_16384,64_12288_8192,1024_2048_8192,12288_64_8192,128_16384_8192,8192_256_12288,8192_128_16384,512_4096_4096,1024_4096_4096,2048_2048_2048,1024_1024_4096,512_512_8192,64_64_16384 \
   --two-pass --keep 2.0 --max-cand 14 --baselines --out $O/ab_splitk.jsonl > $O/ab_splitk.log 2>&1
# whole grid, shipped plans, both entry points, against the CPU oracle
timeout 600 python tests/tools/verify_plans.py --out $O/parity_grid.jsonl > $O/verify.log 2>&1; echo "verify rc=$?" >> $O/verify.log; tail -4 $O/verify.log
