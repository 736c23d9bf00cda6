#!/bin/bash
# memory-side PMC passes only (see pmc_sweep.sh); usage: pmc_sweep_mem.sh OUTDIR -- cmd...
set -u
OUT=$1; shift; shift
export TMPDIR=/tmp
PASSES=(
"GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
"TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum"
"FETCH_SIZE TCC_HIT_sum"
"TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"
)
i=0
for p in "${PASSES[@]}"; do
  timeout 90 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $OUT/pass$i -- "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed/timeout"
  i=$((i+1))
done
