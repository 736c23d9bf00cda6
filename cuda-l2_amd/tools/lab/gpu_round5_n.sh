#!/bin/bash
# Round-5 call N (the last GPU seconds, a box none of the tuning ran on): the 53 rows that ship a phase offset of the persistent walk, each
# at its shipped plan and at the same plan without the flag (kernels and forms: the closing check of call G, profiles/r05_check_final.log) --
# does the flag's gain hold on another box, given that the variants traded places between the re-tune passes?
set -u
O=gpurun_out/r5n; mkdir -p $O
timeout 100 cuda-l2_amd/bin/hgemm_tune tune --shape-file cuda-l2_amd/tuning/r05_phase_rows_recheck_shapes.txt --cand-file cuda-l2_amd/tuning/r05_phase_rows_recheck_candidates.txt --rank both --stream --baselines --out $O/phase_rows_recheck.jsonl > $O/log.txt 2>&1
wc -l $O/phase_rows_recheck.jsonl
