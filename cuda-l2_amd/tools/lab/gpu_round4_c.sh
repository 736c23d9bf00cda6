#!/bin/bash
# Round-4 call C: second pass of the re-tune -- family w (checked exact in call B, profiles/r04_check_r_double_buffer_and_w*.log)
# beyond the first pass's domain, against the table as it stands after pass 1 -- and the oracle parity of its fastest plans.
set -u
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 900 $T tune --shape-file cuda-l2_amd/tuning/r04_retune_pass2_shapes.txt --cand-file cuda-l2_amd/tuning/r04_retune_pass2_candidates.txt --rank both --nt --out $O/r04_retune_pass2.jsonl > $O/r04_retune_pass2.log 2>&1; echo "tune rc=$? lines=$(wc -l < $O/r04_retune_pass2.jsonl)"
timeout 600 python tests/tools/verify_plans.py --plans $O/r04_retune_pass2.jsonl --top 2 --out $O/r04_candidate_parity_pass2.jsonl 2>&1 | tail -2
du -sh $O
