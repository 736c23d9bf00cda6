#!/bin/bash
# Round-6 energy experiment: socket power, gfx clock and time per call of the compute-bound shapes, back to
# back for 2 s per variant (board at its power cap), two passes over the variant list so that each variant is read twice with
# different predecessors.  Variants: the shipped plan, the same geometry without non-temporal stores, its 32x32x16 sibling, the
# classic 8-wave 256x256 tile, family s, hipBLASLt-heuristic tn / nn (MT256x256x64 on these shapes) and rocBLAS tn.
# -> gpurun_out/r6p/power.jsonl -> tools/power_table.py -> profiles/r06_power_table.json
set -u
O=gpurun_out/r6p; mkdir -p $O
T=cuda-l2_amd/bin/hgemm_tune
: > $O/power.jsonl
for pass in 1 2; do
for mnk in 4096_4096_4096 8192_8192_8192 16384_16384_16384; do
  timeout 60 $T bench --shape $mnk --lib --power --seconds 2 >> $O/power.jsonl 2>> $O/power.err
  for cfg in q256x256_w2x2 q256x256_w2x2_m32 t256x256_w2x4_m16_s2 s256x256_w2x2 q256x128_w2x2 q192x256_w2x2; do
    timeout 60 $T bench --shape $mnk --config $cfg --splits 1 --group 4 --power --seconds 2 >> $O/power.jsonl 2>> $O/power.err
  done
  timeout 60 $T bench --shape $mnk --config q256x256_w2x2 --splits 131073 --group 4 --power --seconds 2 >> $O/power.jsonl 2>> $O/power.err
  for b in hipblaslt_tn hipblaslt_nn rocblas_tn; do
    timeout 60 $T bench --shape $mnk --baseline $b --seconds 2 >> $O/power.jsonl 2>> $O/power.err
  done
done
done
wc -l $O/power.jsonl; tail -3 $O/power.err
