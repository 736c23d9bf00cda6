#!/bin/bash
# Round-2 GPU step 3: family q slot-plan variants (interleaved A/B in one process each), small-K shapes after
# the one-coordinate-computation-per-item fix, PMC passes for q vs s at 8192^3.
set -u
O=gpurun_out/r2c; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
SH=4096_4096_4096,8192_8192_8192,8192_4096_4096,4096_4096_16384,8192_8192_1024
for rep in 1 2; do
for v in lib lib_slack2 lib_slack12 lib_rs1 lib_qord1; do
  LD_LIBRARY_PATH=$PWD/cuda-l2_amd/$v timeout 200 $T tune --shapes $SH --configs q256x256_w2x2 --keep 1.01 --max-cand 1 --out $O/var_${v}_$rep.jsonl > $O/var_${v}_$rep.log 2>&1
done; done
timeout 300 $T tune --shapes 16384_16384_1024,8192_8192_512,16384_4096_256,4096_16384_512,16384_16384_256,8192_8192_256,12288_12288_1024 \
   --configs s256x256_w2x2,q256x256_w2x2 --keep 100 --max-cand 2 --baselines --sweep-group --out $O/ab_smallk.jsonl > $O/ab_smallk.log 2>&1
# PMC: SQ pass + cache pass, q vs s at 8192^3
for c in q256x256_w2x2 s256x256_w2x2; do
  mkdir -p $O/pmc_$c
  export TMPDIR=/tmp
  timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_$c/pass0 -- $T bench --shape 8192_8192_8192 --config $c --group 4 --reps 6 > $O/pmc_$c/pass0.log 2>&1
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_$c/pass1 -- $T bench --shape 8192_8192_8192 --config $c --group 4 --reps 6 > $O/pmc_$c/pass1.log 2>&1
  timeout 120 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $O/pmc_$c/pass2 -- $T bench --shape 8192_8192_8192 --config $c --group 4 --reps 6 > $O/pmc_$c/pass2.log 2>&1
done
find $O -name "*.db" -delete 2>/dev/null; du -sh $O
