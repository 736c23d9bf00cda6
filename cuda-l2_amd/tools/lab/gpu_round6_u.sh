#!/bin/bash
# Round-6 call U: the first-use selection off the grid (host code: its candidates are now the runners-up among the tuned plans of the
# surrounding grid shapes, in the model's order): (1) check of two off-grid shapes on every geometry; (2) the first-use-selection GPU tests
# (C ABI incl. an off-grid shape, the harness path) and the off-grid parity / tolerance tests on this library; (3) the selection on the 80
# off-grid shapes: the planner's plan against the chosen one, re-timed interleaved.
set -u
O=gpurun_out/r6u; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
timeout 600 $T check --shapes 1968_576_4096,1000_520_200,1336_3112_4440 > $O/check_offgrid.log 2>&1; echo "check rc=$? $(tail -1 $O/check_offgrid.log)"
grep -q " 0 failures" $O/check_offgrid.log || { echo "CHECK FAILED"; exit 1; }
timeout 1500 python -m pytest tests -m gpu -q -k "first_use or insitu or off_grid" > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
timeout 600 python cuda-l2_amd/tools/lab/insitu_demo_r06.py cuda-l2_amd/tools/offgrid_shapes.txt > $O/insitu_offgrid.jsonl 2> $O/insitu.err; echo "insitu offgrid rc=$? lines=$(wc -l < $O/insitu_offgrid.jsonl)"
