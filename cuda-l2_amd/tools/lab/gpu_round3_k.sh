#!/bin/bash
# Round-3 final GPU call (after the late re-tune): extra exact checks of the late geometries on odd K-step counts, the whole
# `-m gpu` suite, smoke, bench.py + rocprofv3 kernel stats of the same command, whole-grid parity record, grid / off-grid plan
# reports, back-to-back comparison with hipBLASLt on the shapes the late geometries target.
set -u
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
NEW=t128x64_w4x2_m16_s4,t64x128_w2x4_m16_s4,t128x64_w4x2_m16_s3,t64x128_w2x4_m16_s3,q192x256_w2x2,q256x192_w2x2,r96x128_k128,r96x64_k128,r128x96_k128,r64x96_k128
timeout 200 $T check --configs $NEW --shapes 1000_520_192,1000_520_320,1000_520_448,584_1000_192,392_200_192,960_384_192,1152_768_192,3072_3072_192,200_1000_704,1920_1080_832,100_96_2176 > $O/check_odd.log 2>&1; tail -1 $O/check_odd.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
timeout 300 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl 2>&1 | tail -1
timeout 600 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/grid_shapes.txt --out $O/grid_plan_report.jsonl > $O/grid_plan_report.log 2>&1; echo "plan report lines=$(wc -l < $O/grid_plan_report.jsonl)"
timeout 300 $T tune --plan-only --baselines --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
{
for sh in 512_4096_4096 12288_1024_8192 1024_12288_12288 16384_16384_128 128_16384_2048 12288_128_8192 12288_64_12288 3072_3072_3072 1536_6144_6144 6144_6144_6144 4096_11008_4096; do
  echo "# lib"; timeout 30 $T bench --shape $sh --lib --power --seconds 0.4
  echo "# hipblaslt"; timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.4
done
} > $O/stream.jsonl 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O/prof -name "*kernel_trace.csv" -size +20M -delete; du -sh $O
