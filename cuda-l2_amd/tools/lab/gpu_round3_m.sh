#!/bin/bash
# Round-3 closing GPU call, on the tree with the VALU -> asm-MFMA hazard fix (sq_settle): the diagnosis probes on the
# experiment build that still carries the 192 x 192 member, exact checks of the family-q members on odd K-step counts with the
# narrow epilogue (N % 8 = 4), the whole `-m gpu` suite, smoke, bench.py + rocprofv3 stats, whole-grid parity record.
set -u
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
HGEMM_LIB_DIR=cuda-l2_amd/lib_q192 timeout 150 python cuda-l2_amd/tools/diag_q192.py > $O/diag_after_fix.jsonl 2> $O/diag.err; echo "diag rc=$? wrong totals: $(grep -o '"wrong": [0-9]*' $O/diag_after_fix.jsonl | awk '{s+=$2} END{print s}')"
Q=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q256x256_w2x2_m32
timeout 200 $T check --configs $Q --shapes 1000_516_192,960_388_320,584_1004_448,1000_520_192,392_196_64,2304_2308_576,1000_516_256 > $O/check_q_odd_narrow.log 2>&1; tail -1 $O/check_q_odd_narrow.log
LD_LIBRARY_PATH=$P/lib_q192 timeout 100 $T check --configs q192x192_w2x2 --shapes 1000_516_192,1000_520_192,960_388_320,192_192_64,3072_3072_448 > $O/check_q192x192_after_fix.log 2>&1; tail -1 $O/check_q192x192_after_fix.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
timeout 300 python tests/tools/verify_plans.py --out $O/parity_1000.jsonl 2>&1 | tail -1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O/prof -name "*kernel_trace.csv" -size +20M -delete; du -sh $O
