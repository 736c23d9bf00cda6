#!/bin/bash
# Round-5 call E: the phase offset in eight groups (a prologue change of every family-q kernel):
#  (1) EXACT FIRST: every geometry x form (the tool now lists 1|phase-offset8), walk shapes, family w's long-K shapes;
#  (2) re-tune pass 3 (tools/make_round5_candidates.py --what pass3) + oracle parity of the three fastest per shape;
#  (3) the -m gpu suite on this tree (new tests: fp16 torch module, eval_one_file.sh end to end, whole-tile N(0,1) on the BASELINE
#      shapes, first-use selection) and a bench.py run.
set -u
O=gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "phase-offset8" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 400 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
for f in check_all check_q_walk; do grep -q " 0 failures" $O/$f.log || { echo "CHECK FAILED: $f"; grep FAIL $O/$f.log | head -30; exit 1; }; done
timeout 900 $T tune --shape-file cuda-l2_amd/tuning/r05_retune_pass3_shapes.txt --cand-file cuda-l2_amd/tuning/r05_retune_pass3_candidates.txt --rank both --baselines --stream --out $O/retune3.jsonl > $O/retune3.log 2>&1; echo "retune3 rc=$? lines=$(wc -l < $O/retune3.jsonl)"
timeout 900 python tests/tools/verify_plans.py --plans $O/retune3.jsonl --top 3 --out $O/retune3_parity.jsonl 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu.log)"
timeout 300 python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['launch_us'], d['roofline']['wall_per_call_us'], d.get('vs_hipblaslt_autotune_max'))"
du -sh $O
