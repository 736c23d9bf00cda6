#!/bin/bash
# Round-6 call K: closing run on the FINAL tree (table after the fused / wide / flag re-tunes; kernels = calls F - J's, to be
# fingerprinted in profiles/r06_isa_fingerprint_closing_run_library.json):
#  (1) full exact check (the log names geometries and forms) + the walk / long-K shapes;
#  (2) the whole -m gpu suite with its grid passes kept as THE parity / tolerance records of the shipped table; smoke;
#  (3) off-grid plan report (80 never-tuned shapes, isolated + back to back, interleaved rounds);
#  (4) bench.py, then rocprofv3 --kernel-trace --stats of the same command (bench line + CSV of ONE tree);
#  (5) counters: the per-geometry PMC table of the final table (three --pmc passes) -> r06_pmc_table.json + r06_pmc_<BASELINE shape>.json.
set -u
O=gpurun_out/r6k; mkdir -p $O/pmc_table
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "48|fused" && strings $T | grep -q "with-shipped" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_final.log 2>&1; echo "check rc=$? $(tail -1 $O/check_final.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 500 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
WCFG=w64x64,w32x128,w128x32,w32x64,w64x32,w16x16_k4,w32x32_k4,w16x32_k4,w32x16_k4
timeout 500 $T check --shapes 64_64_4096,128_64_8192,80_48_2080,256_256_2048,33_17_1056,512_64_16384 --configs $WCFG > $O/check_w_deep.log 2>&1; echo "check_w_deep rc=$? $(tail -1 $O/check_w_deep.log)"
HGEMM_RECORD_DIR=$O/records timeout 2700 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
export HGEMM_AUTOTUNE_CACHE=$PWD/cuda-l2_amd/tuning/r06_hipblaslt_autotune_cache.txt
timeout 400 $T tune --plan-only --baselines --stream --interleave --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --out $O/offgrid_plan_report.jsonl > $O/offgrid_plan_report.log 2>&1; echo "offgrid report lines=$(wc -l < $O/offgrid_plan_report.jsonl)"
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
python cuda-l2_amd/tools/pmc_table.py shapes > $O/pmc_shapes.txt; wc -l $O/pmc_shapes.txt
sed -i 's/timeout 240 rocprofv3/timeout 150 rocprofv3/' cuda-l2_amd/tools/pmc_table.sh
bash cuda-l2_amd/tools/pmc_table.sh $O/pmc_table $O/pmc_shapes.txt
python cuda-l2_amd/tools/pmc_table.py table $O/pmc_table $O/pmc_shapes.txt > $O/pmc_table.json 2> $O/pmc_table.err; echo "pmc table rc=$? rows=$(grep -c '"mnk"' $O/pmc_table.json)"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O/prof -name "*kernel_trace.csv" -delete; du -sh $O
