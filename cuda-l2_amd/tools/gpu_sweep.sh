#!/bin/bash
# Reference-metric sweep on one GPU (in-process driver).  usage: gpu_sweep.sh OUTDIR ACC MODE SHAPEFILE [extra sweep.py args]
set -u
OUT=$1; ACC=$2; MODE=$3; SHAPES=$4; shift 4
cd cuda-l2_amd
export HGEMM_AUTOTUNE_MAX_SECONDS=${HGEMM_AUTOTUNE_MAX_SECONDS:-0.05}
python tools/sweep.py run --inprocess --out ../$OUT --acc_precise $ACC --mode $MODE --shapes-file $SHAPES "$@"
python tools/sweep.py merge --out ../$OUT --acc_precise $ACC --mode $MODE --shapes-file $SHAPES > ../$OUT/merge_${ACC}_${MODE}.json
python - <<PY
import json
d=json.load(open("../$OUT/merge_${ACC}_${MODE}.json"))
print("$ACC $MODE shapes", d["shapes"], "geomean vs autotune-max %.3f heuristic-max %.3f rocBLAS-max %.3f torch %.3f"%(d["geomean_speedup_vs_hipBLASLt-auto-tuning-max"], d["geomean_speedup_vs_hipBLASLt-heuristic-max"], d["geomean_speedup_vs_rocBLAS-max"], d["geomean_speedup_vs_torch.matmul"]), d["aggregate"])
PY
