#!/bin/bash
# Collect rocprofv3 PMC counters in separate passes (few TCP/TCC slots per pass) for one command.
# Every pass runs under `timeout`: an over-subscribed pass makes rocprofv3 abort and then hang.
#   usage: pmc_sweep.sh OUTDIR -- cmd...
set -u
OUT=$1; shift; shift
export TMPDIR=/tmp
PASSES=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
"GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
"FETCH_SIZE TCC_HIT_sum"
"WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
)
i=0
for p in "${PASSES[@]}"; do
  timeout 90 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed/timeout"
  i=$((i+1))
done
