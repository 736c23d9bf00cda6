#!/bin/bash
# Round-6 call A (no kernel changed since round 5's closing library; host-side changes only):
#  (1) exact check of the two-resident members' single-launch split-K (their launch grid changed: one wave of 256 on the 80 KiB members);
#  (2) the first-use-selection tests: C ABI, the harness path (eval_one_file.sh --insitu, BASELINE config 4 in server mode), default off;
#  (3) the on-disk cache of hipBLASLt autotune winners for the WHOLE grid: 1 s per layout, the reference's protocol
#      (cublas/fp32/hgemm_cublaslt_auto_tuning.cu:108-306), searched once per problem;
#  (4) the four reference-metric sweeps of the table as it ships at this commit, re-using that cache (boxes of rounds 3-4).
set -u
O=gpurun_out/r6a; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 300 $T check --configs q192x128_w2x2,q128x192_w2x2,q128x128_w2x2 > $O/check_two_resident.log 2>&1; echo "check rc=$? $(tail -1 $O/check_two_resident.log)"
timeout 900 python -m pytest tests -m gpu -q -k "first_use or insitu or extension_exports" > $O/pytest_insitu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_insitu.log)"
CACHE=cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
mkdir -p $O/cache
( cd cuda-l2_amd && HGEMM_AUTOTUNE_MAX_SECONDS=1.0 timeout ${1:-2100} python tools/build_autotune_cache.py --cache ../$O/cache/r06_hipblaslt_autotune_cache.txt \
    --shapes-file tools/grid_shapes_shuffled.txt --report ../$O/cache/build_report.json --time_limit ${2:-1900} > ../$O/cache/build.log 2>&1 ); echo "cache rc=$? $(tail -1 $O/cache/build.log | head -c 600)"
export HGEMM_AUTOTUNE_CACHE=$PWD/$O/cache/r06_hipblaslt_autotune_cache.txt
export HGEMM_AUTOTUNE_MAX_SECONDS=1.0
S=$O/sweep; mkdir -p $S
W="--warmup_seconds 0.04 --benchmark_seconds 0.15"
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 offline tools/grid_shapes.txt $W --cpu_max_flops 2e10 --cpu_seconds 0.02 --time_limit 560 2>&1 | tail -2
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 offline tools/grid_shapes.txt $W --time_limit 520 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp32 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
bash cuda-l2_amd/tools/gpu_sweep.sh $S fp16 server tools/grid_shapes.txt $W --target_qps 100 --time_limit 560 2>&1 | tail -1
find $O -name "*.so" -delete 2>/dev/null; find $O -name "*.o" -delete 2>/dev/null; du -sh $O
