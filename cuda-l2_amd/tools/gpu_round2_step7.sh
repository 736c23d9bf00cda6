#!/bin/bash
# Round-2 GPU step 7: family r (register-staged streaming kernel): exactness + A/B on the skinny HBM-bound class.
set -u
O=gpurun_out/r2i; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
timeout 300 $T check --shapes 256_256_1024,320_448_512,300_260_2048,1024_768_768,200_136_1280 > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log; tail -3 $O/check.log
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "geometry or guard or identity" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 $T tune --shapes 16384_64_16384,8192_64_16384,16384_64_8192,12288_64_16384,16384_128_16384,12288_128_16384,8192_128_16384,128_16384_16384,64_16384_16384,64_12288_8192,16384_256_12288,256_16384_16384,4096_64_16384,4096_128_8192,2048_64_16384,16384_64_4096,8192_256_8192,1024_64_16384,64_4096_8192,16384_128_4096 \
   --fused --keep 100 --max-cand 60 --baselines --out $O/ab_skinny.jsonl > $O/ab_skinny.log 2>&1
tail -2 $O/ab_skinny.log
