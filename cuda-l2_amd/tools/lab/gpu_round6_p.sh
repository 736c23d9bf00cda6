#!/bin/bash
# Round-6 call P (final tree): (1) the whole -m gpu suite + smoke on the commit that ships (host-side edits after call K: the off-grid sibling rule);
# (2) a 5-step bench.py; (3) the first-use selection on 64 rows of the box-sensitive classes on this (fresh) box.
set -u
O=gpurun_out/r6p2; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu_final.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest_gpu_final.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 5 --warmup 2 --no-shapes > $O/bench_final_quick.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench_final_quick.json; echo
timeout 600 python cuda-l2_amd/tools/lab/insitu_demo_r06.py > $O/insitu_demo_r06.jsonl 2> $O/insitu.err; echo "insitu rc=$? lines=$(wc -l < $O/insitu_demo_r06.jsonl)"
