/*
 * hgemm_mi355x.h -- C ABI of libhgemm_mi355x.so, the MI355X (gfx950/CDNA4) HGEMM library that
 * replaces the hot path of deepreinforce-ai/CUDA-L2:
 *
 *     C[M,N] (fp16) = A[M,K] (fp16) x B[K,N] (fp16),  fp32 accumulate, alpha = 1, beta = 0
 *
 * Everything here is `extern "C"`, plain pointers and ints; no torch types.  The torch
 * extension `hgemm_lib` (cuda-l2_amd/pybind/hgemm_mi355x_{fp16,fp32}.cc) is a thin shim over
 * these entry points and exports the 15 Python names the reference harness binds
 * (reference pybind/hgemm_a100_fp32.cc:29-52).  INTEGRATION.md shows the binding.
 *
 * Conventions shared by every GEMM entry point
 *   a            device pointer, fp16 [M][K] row-major contiguous
 *   b            device pointer, fp16 [K][N] row-major contiguous
 *   b_col_major  device pointer, fp16 [N][K] row-major contiguous (= B transposed; the
 *                reference harness builds it with tools/utils.py:110-115 as_col_major)
 *   c            device pointer, fp16 [M][N] row-major contiguous, overwritten
 *   stream       hipStream_t (NULL = the legacy default stream, which is what the reference's
 *                <<<grid,block,smem>>> launches use, kernels/a100_F32F16F16F32/64_4096_64.cu:263)
 *   return       0 on success, a negative hgemm_status_t otherwise; launches are asynchronous,
 *                device faults surface at the caller's next synchronisation (same as the
 *                reference, zero_one_correctness_check.py:161-165).
 * Ownership: the caller owns all buffers; the library keeps no pointer past the call.
 */
#ifndef HGEMM_MI355X_H_
#define HGEMM_MI355X_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  HGEMM_OK = 0,
  HGEMM_ERR_BAD_ARG = -1,     /* null pointer, non-positive dimension, bad id */
  HGEMM_ERR_TOO_LARGE = -2,   /* a dimension exceeds the 32-bit addressing of one operand */
  HGEMM_ERR_HIP = -3,         /* a HIP runtime call failed (see hgemm_mi355x_last_hip_error) */
  HGEMM_ERR_BACKEND = -4,     /* rocBLAS / hipBLASLt returned an error */
  HGEMM_ERR_NOT_READY = -5,   /* baseline used before its init / find_best call */
  HGEMM_ERR_NO_ALGO = -6      /* hipBLASLt returned no usable algorithm */
} hgemm_status_t;

/* Accumulate mode names the reference's two kernel trees (F32F16F16F32 / F16F16F16F16).
 * CDNA4 has no fp16-accumulating MFMA (all f16 MFMA opcodes are v_mfma_f32_*), so both modes
 * run the same fp32-accumulate kernels here; for the rocBLAS / hipBLASLt baselines the mode
 * selects the compute type exactly as the reference's cublas/fp16 vs cublas/fp32 trees do. */
typedef enum { HGEMM_ACC_FP32 = 0, HGEMM_ACC_FP16 = 1 } hgemm_acc_t;

/* ------------------------------------------------------------------------------------------
 * The hot path.  Replaces cuda_l2_<dev>_fp32(a, b, b_col_major, c)
 * (reference kernels/a100_F32F16F16F32/64_4096_64.cu:275-287) and cuda_l2_<dev>_fp16
 * (kernels/a100_F16F16F16F16/4096_4096_4096.cu:280-295).  Picks the tuned kernel geometry /
 * split-K plan for (M,N,K) (tuned table first, analytic model otherwise) and launches it.
 * Any M,N,K >= 1 is accepted; shapes the MFMA path cannot take (K % 64 != 0, N % 4 != 0,
 * pointers not 16-byte aligned) run on a slow generic kernel. */
int hgemm_mi355x_fp32(const void* a, const void* b, const void* b_col_major, void* c,
                      int M, int N, int K, void* stream);
int hgemm_mi355x_fp16(const void* a, const void* b, const void* b_col_major, void* c,
                      int M, int N, int K, void* stream);

/* Explicit-plan launch: what a per-shape kernel file
 * (cuda-l2_amd/kernels/mi355x_<acc>/<M>_<N>_<K>.hip, the analogue of the reference's
 * kernels/<dev>_<acc>/<M>_<N>_<K>.cu) and the autotuner call.
 *   config_id  index into the geometry table (hgemm_mi355x_config_*), or -1 for the generic kernel
 *   splits     split-K factor >= 1 (fp32 slabs + deterministic combine kernel when > 1)
 *   group_m    rasterisation group height in tiles (>= 1)
 * lda/ldb/ldc are row strides in elements (ldb is the row stride of b_col_major, i.e. >= K). */
int hgemm_mi355x_launch(int config_id, int splits, int group_m,
                        const void* a, const void* b, const void* b_col_major, void* c,
                        int M, int N, int K, int lda, int ldb, int ldc, void* stream);

/* Plan the library would use for (M,N,K): outputs config id, split-K factor, raster group. */
int hgemm_mi355x_plan(int M, int N, int K, int* config_id, int* splits, int* group_m);

/* The analytic cost model behind hgemm_mi355x_plan, exposed for the autotuner's candidate
 * pruning and for reports: estimated microseconds of (config_id, splits) on (M,N,K). */
double hgemm_mi355x_model_us(int config_id, int splits, int M, int N, int K);

/* Raster group height the planner uses for (config, M, N) when no tuned value exists. */
int hgemm_mi355x_default_group(int config_id, int M, int N);

/* Geometry table introspection (ids are stable positions in csrc/hgemm_configs.def). */
int hgemm_mi355x_num_configs(void);
const char* hgemm_mi355x_config_name(int config_id);
/* out[0..7] = BM, BN, WM, WN, MI, NBUF, threads, lds_bytes */
int hgemm_mi355x_config_info(int config_id, int out[8]);
int hgemm_mi355x_config_by_name(const char* name);

/* Split-K workspace: by default the library grows a private device buffer on demand (first
 * use only, never in steady state).  A caller may instead lend its own buffer. */
int hgemm_mi355x_set_workspace(void* device_ptr, size_t bytes);
size_t hgemm_mi355x_workspace_bytes(int M, int N, int splits);

const char* hgemm_mi355x_strerror(int status);
int hgemm_mi355x_last_hip_error(void);
const char* hgemm_mi355x_version(void);
/* Ablation switches for the native tuner (bit 0: skip steady-state LDS-DMA -> wrong results,
 * timing only).  Returns the previous value.  Never set by the library itself. */
int hgemm_mi355x_set_debug(int flags);

/* ------------------------------------------------------------------------------------------
 * Vendor baselines (same tensors, same process, as in the reference's cublas/ tree).
 *
 * rocBLAS  <-  cublasGemmEx NN / TN (reference cublas/fp32/hgemm_cublas.cu:15-68):
 * row-major C = A.B computed as the column-major product C^T = B^T.A^T. */
int hgemm_rocblas_init(void);      /* init_cublas_handle    (hgemm_cublas.cu:15-28) */
int hgemm_rocblas_destroy(void);   /* destroy_cublas_handle (hgemm_cublas.cu:30-38) */
int hgemm_rocblas_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* stream);
int hgemm_rocblas_tn(const void* a, const void* b_col_major, void* c, int M, int N, int K, int acc, void* stream);

/* hipBLASLt heuristic  <-  cublasLtMatmulAlgoGetHeuristic top-1 of 4, cached
 * (reference cublas/fp32/hgemm_cublaslt_heuristic.cu:65-217). */
int hgemm_hipblaslt_heuristic_init(void);     /* init_cublaslt_handle_v1    */
int hgemm_hipblaslt_heuristic_destroy(void);  /* destroy_cublaslt_handle_v1 */
int hgemm_hipblaslt_heuristic_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* stream);
int hgemm_hipblaslt_heuristic_tn(const void* a, const void* b_col_major, void* c, int M, int N, int K, int acc, void* stream);

/* hipBLASLt autotune  <-  find_best_algo_{nn,tn}_v2 + cublaslt_tensor_op_{nn,tn}_v2
 * (reference cublas/fp32/hgemm_cublaslt_auto_tuning.cu:108-546): every candidate algorithm is
 * timed for 50 warm-up + 100 measured rounds, shuffled order, fresh N(0,1) inputs each round,
 * median per algorithm, best kept. */
int hgemm_hipblaslt_autotune_init(void);      /* init_cublaslt_handle_v2    */
int hgemm_hipblaslt_autotune_destroy(void);   /* destroy_cublaslt_handle_v2 */
int hgemm_hipblaslt_autotune_find_best_nn(int M, int N, int K, int acc);
int hgemm_hipblaslt_autotune_find_best_tn(int M, int N, int K, int acc);
int hgemm_hipblaslt_autotune_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* stream);
int hgemm_hipblaslt_autotune_tn(const void* a, const void* b_col_major, void* c, int M, int N, int K, int acc, void* stream);
/* Introspection for reports: candidates tried / median ms of the winner (nn = 0, tn = 1). */
int hgemm_hipblaslt_autotune_candidates(int tn);
double hgemm_hipblaslt_autotune_best_ms(int tn);

/* Device helper used by the autotune baseline and the native tools: fill `n` fp16 values with
 * N(0,1) samples (counter-based generator; `seed` makes runs reproducible). */
int hgemm_fill_normal_f16(void* device_ptr, size_t n, unsigned long long seed, void* stream);

/* ---- measurement helpers (bench.py; the reference times with torch events around each call,
 * benchmarking_utils.py:23-31) ------------------------------------------------------------------
 * hgemm_mi355x_time_next_launch arms a one-shot hook: the next GEMM call of the calling thread puts
 * the two events on its main kernel's own dispatch packet, so hgemm_mi355x_event_elapsed_us returns
 * that kernel's execution time as rocprofv3 reports it (event-record marker packets around a launch
 * add ~6 us of queue gaps on MI355X).  For split-K plans the reduce kernel is not included.
 * Pass (NULL, NULL) to disarm.  Events are created / destroyed with the two helpers below. */
void* hgemm_mi355x_event_create(void);
int hgemm_mi355x_event_destroy(void* event);
int hgemm_mi355x_time_next_launch(void* start_event, void* stop_event);
/* Waits for stop_event, returns microseconds between the two events (negative on error). */
double hgemm_mi355x_event_elapsed_us(void* start_event, void* stop_event);

#ifdef __cplusplus
}
#endif
#endif /* HGEMM_MI355X_H_ */
