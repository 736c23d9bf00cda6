#!/bin/bash
# rocprofv3 PMC passes for tools/pmc_table.py: usage pmc_table.sh OUTDIR SHAPES.txt   (three passes of ONE bench process each)
set -u
OUT=$1; SH=$2
export TMPDIR=/tmp
LIST=$(paste -sd, $SH)
PASSES=(
"GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
"FETCH_SIZE TCC_HIT_sum"
"WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"
)
i=0
for p in "${PASSES[@]}"; do
  timeout 240 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $OUT/pass$i -- cuda-l2_amd/bin/hgemm_tune bench --shapes $LIST --lib --reps 6 > $OUT/pass$i.log 2>&1 || echo "pass $i failed/timeout"
  i=$((i+1))
done
find $OUT -name "*_agent_info.csv" -delete
