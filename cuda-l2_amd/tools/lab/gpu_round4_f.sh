#!/bin/bash
# Round-4 call F: the lock-step family-q plans (16384 x 256 x 16384, 256 x 16384 x 16384: 161 ... 239 us across boxes, hipBLASLt 157).
# Is it the box or the physical placement of the operands?  The same candidates (all of them exact: profiles/r04_check_final.log)
# in five processes whose operand sets sit behind dummy allocations of different sizes, hipBLASLt beside them.
set -u
O=gpurun_out/r4f; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
for pad in 0 37 301 1111 4099; do
  timeout 120 $T tune --shapes 16384_256_16384,256_16384_16384,16384_512_16384,512_16384_16384 --cand-file cuda-l2_amd/tools/lab/r4_lockstep_candidates.txt --rank both --baselines --stream --pad-alloc $pad --out $O/lockstep_pad$pad.jsonl > $O/lockstep_pad$pad.log 2>&1; echo "pad $pad rc=$? lines=$(wc -l < $O/lockstep_pad$pad.jsonl)"
done
# (2) Family r on the skinny streaming class: the PMC table of call D shows 1.42x the algorithmic bytes on 16384 x 128 x 16384 (1.1x
# on its N = 64 sibling).  Reading: the per-TILE K stagger puts the 32 workgroups of an XCD at 32 different K offsets, so the slices
# of the small shared operand are evicted from the XCD's L2 by the streamed operand between their uses.  Experiment builds (never
# shipped, lib_<suffix>/): K stagger per XCD (an XCD's workgroups in lock-step), no stagger at all, NT loads on the streamed
# operand, and the combinations.  Exactness of each build first (the knobs change the summation order of a tile, not its value on
# 0/1 inputs), then the same candidates per build, hipBLASLt beside the shipping build.
# The two knobs are PLAN FLAGS of the shipping library now (HGEMM_PLAN_RS_XCD_STAGGER, HGEMM_PLAN_RS_NT_LOADS); "no stagger at all"
# stays an experiment build (lib_rs0/, lib_rs0nt/).  Exactness first, then per shape its r-family plans x {-, xcd, nt, xcd + nt}.
P=$PWD/cuda-l2_amd
RCFG=r64x64_k256,r64x128_k128,r128x64_k128,r128x128_k128,r96x128_k128,r96x64_k128,r128x96_k128,r64x96_k128,r128x128_k128_d,r64x128_k128_d,r128x64_k128_d,r64x64_k256_d
awk '{print $1}' cuda-l2_amd/tools/lab/r4_skinny_r_candidates.txt > $O/skinny_shapes.txt
timeout 200 $T check --configs $RCFG --shapes 256_256_1024,320_448_512,512_1024_2048,300_260_2048,1536_128_4096,1000_520_1280 > $O/check_r_flags.log 2>&1; echo "check rc=$? $(tail -1 $O/check_r_flags.log)"
timeout 300 $T tune --shape-file $O/skinny_shapes.txt --cand-file cuda-l2_amd/tools/lab/r4_skinny_r_flag_candidates.txt --rank both --baselines --stream --out $O/skinny_r_flags.jsonl > $O/skinny_r_flags.log 2>&1; echo "flags tune rc=$? lines=$(wc -l < $O/skinny_r_flags.jsonl)"
timeout 200 python tests/tools/verify_plans.py --plans $O/skinny_r_flags.jsonl --top 3 --out $O/skinny_r_flags_parity.jsonl 2>&1 | tail -1
for lib in lib_rs0 lib_rs0nt; do
  LD_LIBRARY_PATH=$P/$lib timeout 100 $T check --configs $RCFG --shapes 256_256_1024,512_1024_2048,1536_128_4096 > $O/check_r_$lib.log 2>&1; echo "$lib check rc=$? $(tail -1 $O/check_r_$lib.log)"
  LD_LIBRARY_PATH=$P/$lib timeout 120 $T tune --shape-file $O/skinny_shapes.txt --cand-file cuda-l2_amd/tools/lab/r4_skinny_r_candidates.txt --rank both --out $O/skinny_r_$lib.jsonl > $O/skinny_r_$lib.log 2>&1; echo "$lib tune rc=$? lines=$(wc -l < $O/skinny_r_$lib.jsonl)"
done
# (2b) the counters behind the reading: HBM + Infinity-Cache bytes of 16384 x 128 x 16384 with and without the flags (separate --pmc
# passes for FETCH_SIZE and WRITE_SIZE, MI355X_MICROARCH.md; one launch at a time)
for fl in 0 1572864; do   # 0x180000 = xcd stagger + nt loads
  sp=$((65538 + fl))
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 60 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_flags_${fl}_$c -- $T bench --shape 16384_128_16384 --config r128x128_k128_d --splits $sp --group 4 --reps 6 > $O/pmc_flags_${fl}_$c.log 2>&1 || echo "pmc $fl $c failed"
  done
done
find $O -name "*_agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
# (3) bench.py again (its `vs_hipblaslt_autotune_max.hipblaslt_tflops` field had the ratio inverted in call D's record), with the
# rocprofv3 kernel stats of the same command, so that the committed bench line, profiled line and CSV come from one tree
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --no-shapes --no-cpu-baseline > $O/bench_profiled.json 2> $O/prof.err; echo "rocprof rc=$?"
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -size +2M -delete
du -sh $O
