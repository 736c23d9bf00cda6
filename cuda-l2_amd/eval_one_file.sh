#!/bin/bash
# Evaluate ONE (M,N,K): exact 0/1 correctness check, then the seven baselines against cuda_l2 in
# shuffled order (each in its own process), then the summary table.  Same flags as the reference's
# eval_one_file.sh (:16-59); --device_type is mi355x.
#   ./eval_one_file.sh --mnk 64_4096_64 --acc_precise fp32 --device_type mi355x --warmup_seconds 5 \
#       --benchmark_seconds 10 --base_dir ./results/64_4096_64 --gpu_device_id 0 --mode offline
#   ... --mode server --target_qps 100
#   ... --defense          additionally run the self-audit (defense.py) before the benchmarks
#   ... --insitu           first-use plan selection on this box (HGEMM_MI355X_INSITU=1): the first cuda_l2 call of every process
#                          -- it falls into the warm-up seconds -- times the shipped plan and its oracle-verified alternates and
#                          keeps the fastest, as the reference's H100 kernels tune on first invocation
#                          (kernels/h100_F32F16F16F32/64_4096_64.cu:623-690,702-721); benchmark_result_*.json then carry "insitu"
cd "$(dirname "$0")" || exit 1

MODE="offline"; TARGET_QPS=""; DEVICE_TYPE="mi355x"; GPU_DEVICE_ID=0; DEFENSE=0
while [[ $# -gt 0 ]]; do
    case $1 in
        --mnk) MNK="$2"; shift 2 ;;
        --acc_precise) ACC_PRECISE="$2"; shift 2 ;;
        --device_type) DEVICE_TYPE="$2"; shift 2 ;;
        --warmup_seconds) WARMUP_SECONDS="$2"; shift 2 ;;
        --benchmark_seconds) BENCHMARK_SECONDS="$2"; shift 2 ;;
        --base_dir) BASE_DIR="$2"; shift 2 ;;
        --gpu_device_id) GPU_DEVICE_ID="$2"; shift 2 ;;
        --mode) MODE="$2"; shift 2 ;;
        --target_qps) TARGET_QPS="$2"; shift 2 ;;
        --defense) DEFENSE=1; shift 1 ;;
        --insitu) export HGEMM_MI355X_INSITU=1; shift 1 ;;
        *) echo "Unknown option: $1"; exit 1 ;;
    esac
done
for v in MNK ACC_PRECISE WARMUP_SECONDS BENCHMARK_SECONDS BASE_DIR; do
    if [ -z "${!v}" ]; then echo "missing --${v,,}"; exit 1; fi
done
if [ "$MODE" == "server" ] && [ -z "$TARGET_QPS" ]; then echo "--mode server needs --target_qps"; exit 1; fi
echo "MNK: $MNK  ACC_PRECISE: $ACC_PRECISE  DEVICE_TYPE: $DEVICE_TYPE  MODE: $MODE"
echo "WARMUP_SECONDS: $WARMUP_SECONDS  BENCHMARK_SECONDS: $BENCHMARK_SECONDS  BASE_DIR: $BASE_DIR  GPU_DEVICE: $GPU_DEVICE_ID"

mkdir -p "$BASE_DIR"
rm -f "$BASE_DIR"/benchmark_result_*.json

COMMON=(--mnk "$MNK" --acc_precise "$ACC_PRECISE" --device_type "$DEVICE_TYPE" --base_dir "$BASE_DIR" --gpu_device_id "$GPU_DEVICE_ID")
python zero_one_correctness_check.py "${COMMON[@]}"
if [ $? -ne 0 ]; then
    echo "Error: Correctness Check failed or raised. Exiting..."
    exit 1
fi

if [ "$DEFENSE" == "1" ]; then
    python defense.py "${COMMON[@]}"
    if [ $? -ne 0 ]; then
        echo "Error: defense audit failed. Exiting..."
        exit 1
    fi
fi

PERF_FUNCS=(hgemm_cublas_tn hgemm_cublas_nn hgemm_cublaslt_heuristic_tn hgemm_cublaslt_heuristic_nn
            hgemm_cublaslt_auto_tuning_tn hgemm_cublaslt_auto_tuning_nn matmul)
echo "Executing hgemm benchmark with shuffled perf_funcs..."
for func in $(shuf -e "${PERF_FUNCS[@]}"); do
    echo "---------------------------------------------------------"
    echo ">>> Running benchmark for: $func"
    if [ "$MODE" == "server" ]; then
        python benchmarking_server.py "${COMMON[@]}" --warmup_seconds "$WARMUP_SECONDS" \
            --benchmark_seconds "$BENCHMARK_SECONDS" --perf_func "$func" --target_qps "$TARGET_QPS"
    else
        python benchmarking_offline.py "${COMMON[@]}" --warmup_seconds "$WARMUP_SECONDS" \
            --benchmark_seconds "$BENCHMARK_SECONDS" --perf_func "$func"
    fi
    if [ $? -ne 0 ]; then
        echo "Error: Benchmark failed at perf_func: $func. Exiting..."
        exit 1
    fi
done

python summarize_result.py --base_dir "$BASE_DIR" --acc_precise "$ACC_PRECISE" --device_type "$DEVICE_TYPE"
echo "All benchmarks completed successfully!"
