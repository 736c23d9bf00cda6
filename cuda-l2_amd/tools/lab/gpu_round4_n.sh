#!/bin/bash
# Round-4 call N (the last GPU seconds): VERDICT r3 item 6c asked for counters on why the 32x32x16 member of family q is slower than
# the 16x16x32 one although its micro-benchmark peak is higher.  One SQ / GRBM counter pass (the pass-0 list of tools/pmc_table.sh)
# of 4096^3 on q256x256_w2x2 and on q256x256_w2x2_m32, same plan flags (NT stores, raster group 4), six back-to-back launches each.
# Both kernels are in call K's check log (exact) -- this run only reads counters.
set -u
O=gpurun_out/r4n; mkdir -p $O
export TMPDIR=/tmp
C="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
for cfg in q256x256_w2x2 q256x256_w2x2_m32; do
  timeout 25 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$cfg -- cuda-l2_amd/bin/hgemm_tune bench --shape 4096_4096_4096 --config $cfg --splits 131073 --group 4 --reps 6 > $O/$cfg.log 2>&1; echo "$cfg rc=$?"
done
find $O -name "*agent_info.csv" -delete; du -sh $O
