#!/bin/bash
# Round-2 GPU step 5: family q 128x128 members (BK=128 / BK=64): exactness, race screen, A/B on the mid-size class.
set -u
O=gpurun_out/r2e; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 300 $T check --shapes 256_256_1024,320_448_512,1000_520_192,300_260_2048,1024_768_576,640_640_640 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log; tail -4 $O/check.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "race or geometry or guard or identity or hybrid" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 500 $T tune --shapes 1024_4096_4096,4096_1024_4096,2048_2048_2048,512_8192_4096,512_4096_4096,2048_2048_4096,1024_2048_4096,2048_4096_4096,1024_8192_4096,4096_4096_1024,2048_2048_8192,2048_2048_1024,1024_4096_1024,4096_2048_2048,512_4096_8192,256_16384_4096,1024_1024_4096,2048_4096_2048,4096_4096_4096 \
   --configs q128x128_w2x2_k128,q128x128_w2x2,t128x128_w2x2_m16_s3,t128x128_w2x2_m16_s2,t128x128_w2x4_m16_s4,q256x128_w2x2,q128x256_w2x2,t64x128_w2x2_m16_s4,q256x256_w2x2 --fused --keep 100 --max-cand 40 --baselines --out $O/ab_mid.jsonl > $O/ab_mid.log 2>&1
tail -2 $O/ab_mid.log
