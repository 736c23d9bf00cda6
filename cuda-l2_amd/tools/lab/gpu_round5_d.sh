#!/bin/bash
# Round-5 call D: (calls A-C ran a STALE bin/hgemm_tune -- `python build.py` rebuilds the library only, `--tools` was missing -- so
# their check logs enumerate the round-4 forms: the new GEOMETRIES were covered, family q's new plan FORMS (kstagger variants, phase
# flags, wave priority) were not; every plan adopted from call C is oracle-verified by verify_plans.py, tuning/r05_candidate_parity_pass1.jsonl.)
#  (1) EXACT FIRST, with the rebuilt tool: every geometry x every form incl. family q's new ones, the persistent-walk shapes, the long-K
#      shapes of family w's deep trips.  Nothing is timed when a check fails.
#  (2) re-tune pass 2 against the table as pass 1 left it (tools/make_round5_candidates.py --what pass2);
#  (3) oracle parity of the three fastest plans per shape.
set -u
O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
strings $T | grep -q "family q also" || { echo "STALE hgemm_tune"; exit 1; }
timeout 900 $T check > $O/check_all.log 2>&1; echo "check_all rc=$? $(tail -1 $O/check_all.log)"
QCFG=q256x256_w2x2,q256x128_w2x2,q128x256_w2x2,q128x128_w2x2_k128,q128x128_w2x2,q192x256_w2x2,q256x192_w2x2,q192x128_w2x2,q128x192_w2x2
timeout 400 $T check --shapes 2304_2304_1024,2496_2432_640,4224_4096_512 --configs $QCFG > $O/check_q_walk.log 2>&1; echo "check_q_walk rc=$? $(tail -1 $O/check_q_walk.log)"
WCFG=w64x64,w32x128,w128x32,w32x64,w64x32,w16x16_k4,w32x32_k4,w16x32_k4,w32x16_k4
timeout 400 $T check --shapes 64_64_4096,128_64_8192,80_48_2080,256_256_2048,33_17_1056,512_64_16384 --configs $WCFG > $O/check_w_deep.log 2>&1; echo "check_w_deep rc=$? $(tail -1 $O/check_w_deep.log)"
for f in check_all check_q_walk check_w_deep; do grep -q " 0 failures" $O/$f.log || { echo "CHECK FAILED: $f"; grep FAIL $O/$f.log | head -30; exit 1; }; done
timeout 2400 $T tune --shape-file cuda-l2_amd/tuning/r05_retune_pass2_shapes.txt --cand-file cuda-l2_amd/tuning/r05_retune_pass2_candidates.txt --rank both --baselines --stream --out $O/retune2.jsonl > $O/retune2.log 2>&1; echo "retune2 rc=$? lines=$(wc -l < $O/retune2.jsonl)"
timeout 1500 python tests/tools/verify_plans.py --plans $O/retune2.jsonl --top 3 --out $O/retune2_parity.jsonl 2>&1 | tail -2
du -sh $O
