#!/bin/bash
# Round-2 GPU step 4: single-launch split-K, write-through (sc1) protocol: exactness + A/B against the two-pass form.
set -u
O=gpurun_out/r2d; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 300 $T check --shapes 256_256_1024,320_448_512,1000_520_192,300_260_2048,1024_768_576 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log; tail -3 $O/check.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split or race or geometry or streams or lent" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 500 $T tune --shapes 64_64_2048,128_512_2048,1024_128_2048,128_256_4096,128_2048_2048,256_128_8192,512_256_4096,256_64_16384,2048_128_4096,256_256_12288,128_1024_8192,128_2048_8192,4096_128_4096,2048_64_12288,128_4096_8192,2048_64_16384,1024_512_12288,256_2048_16384,64_12288_8192,1024_2048_8192,12288_64_8192,128_16384_8192,8192_256_12288,8192_128_16384,512_4096_4096,1024_4096_4096,2048_2048_2048,1024_1024_4096,512_512_8192,64_64_16384,512_8192_4096,4096_512_4096,1024_2048_4096,512_2048_8192,256_4096_4096,2048_1024_8192 \
   --fused --keep 2.5 --max-cand 20 --baselines --out $O/ab_splitk.jsonl > $O/ab_splitk.log 2>&1
tail -2 $O/ab_splitk.log
