#!/bin/bash
# Round-2 GPU step 2: parity of family "q" (early-A split) + A/B against family "s" and the vendor libraries.
set -u
O=gpurun_out/r2b; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
( timeout 900 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
timeout 300 $T check --shapes 256_256_1024,320_448_512,1000_520_192,300_260_2048,1024_768_576 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log; tail -3 $O/check.log
timeout 600 $T tune --shapes 4096_4096_4096,8192_8192_8192,4096_4096_1024,8192_8192_1024,8192_4096_4096,4096_4096_16384,16384_16384_1024,4096_8192_2048,16384_16384_16384,2048_2048_8192,16384_4096_256,8192_8192_512 \
   --configs s256x256_w2x2,q256x256_w2x2,s256x128_w2x2,q256x128_w2x2,s128x256_w2x2,q128x256_w2x2 --keep 100 --max-cand 6 --baselines --sweep-group --out $O/ab_q.jsonl > $O/ab_q.log 2>&1
tail -3 $O/ab_q.log
