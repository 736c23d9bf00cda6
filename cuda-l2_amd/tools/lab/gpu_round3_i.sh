#!/bin/bash
# Round-3 diagnostics call (no shipped state changes): what hipBLASLt runs on the shapes we lose, the whole candidate
# landscape of our own geometries on them, the planner's regret on the off-grid list, timelines of the mid-size plans.
set -u
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
T=cuda-l2_amd/bin/hgemm_tune
P=$PWD/cuda-l2_amd
LOSERS=128_16384_16384,16384_128_16384,12288_128_8192,16384_16384_128,512_4096_2048,512_4096_4096,16384_128_4096,128_4096_16384,256_16384_16384,16384_256_16384,4096_16384_512,8192_128_12288,64_256_2048,2048_128_2048,12288_64_8192,256_256_1024,3072_3072_3072,1536_6144_6144,12032_2048_7152,4096_4096_4096
# 1. hipBLASLt's kernels (names carry macro tile, depth, split, store policy) + our kernels, per dispatch
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/hbl -- $T bench --shapes $LOSERS --baseline hipblaslt_tn --isolated --reps 4 > $O/hbl.log 2>&1; echo "hbl rc=$?"
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/ours -- $T bench --shapes $LOSERS --lib --reps 4 > $O/ours.log 2>&1; echo "ours rc=$?"
# 2. every candidate of ours on those shapes
timeout 200 $T tune --shapes $LOSERS --max-cand 56 --keep 8 --fused --baselines --out $O/losers_tune.jsonl > $O/losers_tune.log 2>&1; echo "losers tune rc=$? lines=$(wc -l < $O/losers_tune.jsonl)"
# 3. planner regret off the grid
timeout 240 $T tune --shape-file cuda-l2_amd/tools/offgrid_shapes.txt --max-cand 12 --keep 3 --fused --baselines --out $O/offgrid_tune.jsonl > $O/offgrid_tune.log 2>&1; echo "offgrid tune rc=$? lines=$(wc -l < $O/offgrid_tune.jsonl)"
# 4. timelines of the mid-size family-q plans
{
for spec in "512_4096_4096 --lib" "512_4096_4096 --config q128x128_w2x2 --splits 1" "512_4096_4096 --config q128x128_w2x2_k128 --splits 2" \
            "512_4096_4096 --config q256x128_w2x2 --splits 4" "512_4096_2048 --lib" "512_4096_2048 --config q128x128_w2x2 --splits 2" \
            "16384_16384_256 --lib" "4096_16384_512 --lib" "12288_128_8192 --lib" "256_16384_16384 --lib" "3072_3072_3072 --lib"; do
  echo "# tl $spec"; LD_LIBRARY_PATH=$P/lib_tl timeout 40 $T bench --shape $spec --timeline
done
} > $O/timeline.jsonl 2>&1
# 5. back to back, ours vs hipBLASLt, on the mid-size losers (isolated figures of 20 us kernels are noisy)
{
for sh in 512_4096_2048 512_4096_4096 128_16384_16384 16384_128_16384 12288_128_8192 16384_16384_128 256_16384_16384 4096_16384_512 3072_3072_3072 256_256_1024; do
  echo "# lib"; timeout 30 $T bench --shape $sh --lib --power --seconds 0.4
  echo "# hipblaslt"; timeout 30 $T bench --shape $sh --baseline hipblaslt_tn --seconds 0.4
done
} > $O/stream.jsonl 2>&1
# 6. explicit variants, back to back: NT C stores on the write-streaming shape; NT loads of the streamed operand (lib_rsnt)
{
for spec in "16384_16384_128 --config q256x128_w2x2 --splits 131073" "16384_16384_128 --config q128x256_w2x2 --splits 131073" \
            "16384_16384_128 --config q256x256_w2x2 --splits 131073" "16384_16384_128 --config t128x128_w2x2_m16_s2 --splits 131073" \
            "16384_16384_128 --config t256x256_w2x4_m16_s2 --splits 131073"; do
  echo "# $spec"; timeout 30 $T bench --shape $spec --power --seconds 0.3
done
for lib in lib lib_rsnt; do
for spec in "128_16384_16384 --config r128x128_k128 --splits 2" "16384_128_16384 --config r64x128_k128 --splits 65538" \
            "16384_128_16384 --config r128x128_k128 --splits 2" "12288_128_8192 --config r64x128_k128 --splits 65538" \
            "12288_64_8192 --config r128x64_k128 --splits 65538" "16384_64_16384 --config r128x64_k128 --splits 65538"; do
  echo "# $lib $spec"; LD_LIBRARY_PATH=$P/$lib timeout 30 $T bench --shape $spec --power --seconds 0.3
done
done
} > $O/variants.jsonl 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*agent_info.csv" -delete; du -sh $O
