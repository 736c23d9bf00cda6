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
  HGEMM_ERR_NO_ALGO = -6,     /* hipBLASLt returned no usable algorithm */
  HGEMM_ERR_NO_WORKSPACE = -7 /* hgemm_mi355x_reserve_workspace only: lent buffer too small, out of memory, or the
                                 stream is capturing (a GEMM call never returns it: it runs without split-K instead) */
} hgemm_status_t;

/* Accumulate mode names the reference's two kernel trees (F32F16F16F32 / F16F16F16F16).
 * CDNA4 has no fp16-accumulating MFMA (all f16 MFMA opcodes are v_mfma_f32_*), so both modes
 * run the same fp32-accumulate kernels here; for the rocBLAS / hipBLASLt baselines the mode
 * selects the compute type exactly as the reference's cublas/fp16 vs cublas/fp32 trees do. */
typedef enum { HGEMM_ACC_FP32 = 0, HGEMM_ACC_FP16 = 1 } hgemm_acc_t;

/* Special config ids of hgemm_mi355x_launch / hgemm_mi355x_plan (ids >= 0 index the geometry table of THIS library build:
 * resolve a geometry by name with hgemm_mi355x_config_by_name, ids shift when a build adds members). */
#define HGEMM_CONFIG_GENERIC (-1) /* one-output-per-thread reference kernel; reads b (row-major)        */
#define HGEMM_CONFIG_RAGGED  (-2) /* register-staged MFMA kernel for any M,N,K / alignment; reads b_col_major */

/* Split-K forms.  `splits` > 1 selects the two-launch form: fp32 slabs [splits][M][N] + a combine kernel that
 * adds them in split order.  OR-ing HGEMM_SPLITK_FUSED into `splits` selects the single-launch form instead
 * (fp32 partials + a per-tile arrival counter; the last workgroup to arrive adds the partials in split order
 * and writes the tile -- the deterministic counterpart of the reference's atomicAdd split-K,
 * kernels/a100_F32F16F16F32/64_256_16384.cu:149-152).  Measured on MI355X the single-launch form only wins
 * for the very smallest outputs (the last arriver reads the slabs alone, at 60-110 GB/s), so the tuner picks
 * it per shape.  Both forms are deterministic (fixed summation order). */
#define HGEMM_SPLITK_FUSED 0x10000
#define HGEMM_SPLITK_MASK  0x0ffff
/* Plan flag, OR-ed into `splits` like HGEMM_SPLITK_FUSED: the fp16 C stores of the launch are non-temporal (streaming).
 * C is written once and never re-read; streaming stores leave the device during the epilogue instead of sitting dirty in
 * the XCDs' write-back L2s until the end-of-kernel release.  Measured back to back on MI355X (round 3): the gap between
 * two launches shrinks from 3.7 to 2.0 us; 4096^3 -4 %, 8192 x 16384 x 256 -8 %, 16384^2 x 256 +1.5 %: the tuner decides
 * per shape.  Results are bit-identical either way.
 * Honoured by the direct fp16 epilogues of every family (whole tiles of a stream-K run and the last arriver of a single-launch
 * split-K included); ignored by the two-pass combine kernel, the hybrid tail's reduce and the any-shape kernels; the off-grid
 * planner does not propagate it from a corner plan. */
#define HGEMM_PLAN_NT_STORE 0x20000
/* Plan flag: STREAM-K (the reference's H100 tree: cutlass::gemm::StreamKScheduler, kernels/h100_F32F16F16F32/
 * 128_4096_16384.cu:79, 16384_512_2048.cu:71-73).  One persistent launch: the tiles x K-stages of the GEMM form one
 * tile-major sequence and each of G workgroups walks a contiguous run of it, so the chip is filled whatever the tile count
 * (12288 x 128 x 8192: 96 tiles of 128 x 128 on 256 CUs) and the workgroups of a cut tile start their K walks at different
 * offsets.  With the flag the low 16 bits of `splits` are G (0 or 1: one resident wave of workgroups, 256 x the geometry's
 * workgroups per CU).  At most the first and the last segment of a run are partial tiles; they go through compact fp32
 * slabs and per-tile arrival counters, the workgroup that completes a tile adds its parts in K order (deterministic, nobody
 * waits).  Geometries of the classic ("t") and register-staged ("r") families have the kernel
 * (hgemm_mi355x_config_streamk); elsewhere, or without workspace, the plan runs as the geometry's plain launch. */
#define HGEMM_PLAN_STREAMK 0x40000
/* Plan flags of the register-staged streaming family ("r" geometries; ignored elsewhere).  Results are exact either way (0/1
 * inputs) / differ only in the summation order of a tile's K walk (N(0,1) inputs), deterministically per plan.
 * HGEMM_PLAN_RS_XCD_STAGGER: the K stagger of the family (workgroups start their K walk at different stages so that the chip does
 *   not read one K offset of rows 16-32 KiB apart at the same time) is taken per XCD instead of per tile: the workgroups of an
 *   XCD walk K in lock-step and share the slices of the small operand in that XCD's L2; the eight XCDs are nk / 8 stages apart.
 *   (The summation order of a tile then depends on which workgroup computes it: "per plan" includes the raster group, as it does
 *   for a stream-K plan.)
 * HGEMM_PLAN_RS_NT_LOADS: the STREAMED operand (the one with more rows, read exactly once) is loaded non-temporally, so it does
 *   not push the shared operand's slices out of the L2 (what hipBLASLt's kernels do on the skinny shapes: NTA / NTB). */
#define HGEMM_PLAN_RS_XCD_STAGGER 0x80000
#define HGEMM_PLAN_RS_NT_LOADS    0x100000
/* The same bit for the "q" geometries (round 5; 16x16x32 members, K a whole number of pipeline stages -- ignored otherwise):
 * HGEMM_PLAN_XCD_STAGGER selects the "kstagger" kernel variant -- the workgroups of XCD x walk every work item's K stages in the
 * rotated order x nk / 8, ..., nk - 1, 0, ..., x nk / 8 - 1 (family q has no stagger without the flag).  For the one-round plans
 * whose 256 workgroups otherwise read the same K offset of rows 16-32 KiB apart at the same time: the reference's H100 tree
 * gets the same effect from stream-K start offsets (kernels/h100_F32F16F16F32/16384_512_2048.cu:71-73).  Exactness / determinism
 * as above. */
#define HGEMM_PLAN_XCD_STAGGER HGEMM_PLAN_RS_XCD_STAGGER
/* Plan flags of the persistent "q" geometries (round 5; ignored elsewhere).  Results are bit-identical with and without them.
 * HGEMM_PLAN_PHASE_OFFSET: every second workgroup of an XCD starts its walk over the work items half an item period late, so
 *   that the epilogues of an XCD's CUs (their C stores share the XCD's path into the fabric) no longer coincide.  For walks of
 *   several items per workgroup.
 * HGEMM_PLAN_WAVE_PRIORITY: two-resident members (two workgroups per CU: q128x128_w2x2, q192x128_w2x2, q128x192_w2x2): the two
 *   waves of a SIMD get different static priorities, so one workgroup's epilogue runs under the other's K loop. */
#define HGEMM_PLAN_PHASE_OFFSET  0x200000
#define HGEMM_PLAN_WAVE_PRIORITY 0x400000
#define HGEMM_PLAN_PHASE_OFFSET4 0x800000   /* four phase groups a quarter period apart instead of two half a period apart */
#define HGEMM_PLAN_PHASE_OFFSET8 (HGEMM_PLAN_PHASE_OFFSET | HGEMM_PLAN_PHASE_OFFSET4)   /* both bits: eight groups an eighth of a period apart */

/* ------------------------------------------------------------------------------------------
 * The hot path.  Replaces cuda_l2_<dev>_fp32(a, b, b_col_major, c)
 * (reference kernels/a100_F32F16F16F32/64_4096_64.cu:275-287) and cuda_l2_<dev>_fp16
 * (kernels/a100_F16F16F16F16/4096_4096_4096.cu:280-295).  Picks the tuned kernel geometry /
 * split-K plan for (M,N,K) and launches it: the tuned table for the 1000 grid shapes; for any other shape the
 * tuned plans of the grid shapes around it, ranked by the analytic model (the model alone when none of them fits);
 * off-grid plans are remembered per thread, so a repeated shape costs one table probe.
 * Any M,N,K >= 1 is accepted; shapes the LDS-DMA kernels cannot take (K % 8 != 0, N % 4 != 0,
 * pointers or strides not 16-byte aligned, operands beyond 32-bit tile offsets) run on a register-staged
 * MFMA kernel that pads on the way into LDS (the reference pads in the harness, tools/utils.py:8-36).
 * Thread / stream safety: calls may be issued from any thread on any stream and device; split-K plans use
 * a library-owned workspace that is private to the (device, stream) pair of the call. */
int hgemm_mi355x_fp32(const void* a, const void* b, const void* b_col_major, void* c,
                      int M, int N, int K, void* stream);
int hgemm_mi355x_fp16(const void* a, const void* b, const void* b_col_major, void* c,
                      int M, int N, int K, void* stream);

/* Explicit-plan launch: what a per-shape kernel file
 * (cuda-l2_amd/kernels/mi355x_<acc>/<M>_<N>_<K>.hip, the analogue of the reference's
 * kernels/<dev>_<acc>/<M>_<N>_<K>.cu) and the autotuner call.
 *   config_id  index into the geometry table (hgemm_mi355x_config_*), HGEMM_CONFIG_GENERIC or
 *              HGEMM_CONFIG_RAGGED; a table geometry whose alignment rules the operands do not meet is
 *              served by the ragged kernel
 *   splits     split-K factor >= 1, optionally | HGEMM_SPLITK_FUSED (see above); clamped to K / 64;
 *              degrades to 1 when no workspace is available (lent buffer too small, out of memory)
 *   group_m    rasterisation group height in tiles (>= 1)
 * lda/ldb/ldc are row strides in elements (ldb is the row stride of b_col_major, i.e. >= K). */
int hgemm_mi355x_launch(int config_id, int splits, int group_m,
                        const void* a, const void* b, const void* b_col_major, void* c,
                        int M, int N, int K, int lda, int ldb, int ldc, void* stream);

/* First-use plan selection on the box the library runs on (opt-in; the reference's H100 kernels time their variants on first
 * invocation, kernels/h100_F32F16F16F32/64_4096_64.cu:623-690).  Off by default: the entry points take the tuned table's plan.
 * With HGEMM_MI355X_INSITU=1 in the environment, or after hgemm_mi355x_set_insitu(1) (returns the previous setting; 0 also
 * forgets every recorded choice), the FIRST call of hgemm_mi355x_fp32 / _fp16 for a shape times up to three oracle-verified
 * plans -- the table's and its alternates (hgemm_mi355x_insitu_candidates lists them, the table's plan first; off the grid, round 6:
 * the planner's plan, then the runners-up among the tuned plans of the surrounding grid shapes in the model's order) -- on the call's
 * own operands and stream (interleaved rounds), keeps the fastest for the (process, device) (an alternate must win by 3 %) and runs it; that call
 * synchronises the stream, so it belongs in a warm-up phase.  Later calls, and calls on a capturing stream, time nothing.
 * hgemm_mi355x_insitu_choice returns 1 and the recorded plan once a shape has been measured. */
int hgemm_mi355x_set_insitu(int enable);
/* 1 when first-use selection is on (environment or hgemm_mi355x_set_insitu).  The per-shape kernel files
 * (csrc/hgemm_shape_entry.hpp) ask this before they launch their pinned plan: with the selection on they hand the call to
 * hgemm_mi355x_fp32 / _fp16 instead, so that `HGEMM_MI355X_INSITU=1 ./eval_one_file.sh ...` (or `--insitu`) measures on the
 * harness path -- where the reference's first-call autotune runs (kernels/h100_F32F16F16F32/64_4096_64.cu:702-721).
 * Choices are kept per (device, shape).  With the selection on, N(0,1) results may differ in their last bits from run to run and box
 * to box: a split-K / K-stagger alternate adds a tile's K stages up in another (fixed, deterministic per plan) order; 0/1 inputs
 * stay exact whatever is chosen. */
int hgemm_mi355x_insitu_enabled(void);
int hgemm_mi355x_insitu_candidates(int M, int N, int K, int config_id[3], int splits[3], int group_m[3]);
int hgemm_mi355x_insitu_choice(int M, int N, int K, int* config_id, int* splits, int* group_m);

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
/* The K multiple a geometry accepts: 8 for the families that take a K tail -- the classic "t" family zero-fills a partial
 * last K-step itself; families "q" (16x16x32 members) and "r" run the whole pipeline stages and accumulate the remainder
 * from fragments loaded straight from global memory (round 4; K must then hold at least one whole stage:
 * hgemm_mi355x_config_accepts_k) -- and the pipeline stage depth for the others (64; the reference pads K in the
 * harness instead, tools/utils.py:8-36).  hgemm_mi355x_launch returns HGEMM_ERR_BAD_ARG for a table geometry when K is a
 * multiple of 64 but not of its stage depth, and serves any other K it cannot take with the any-shape kernel (the planner
 * never picks such a geometry); 1 for the special ids. */
int hgemm_mi355x_config_k_granularity(int config_id);
/* 1 when hgemm_mi355x_launch runs this geometry's own kernel for a problem with this K (operands aligned), 0 when it
 * would fall back or refuse: K a multiple of the stage depth, or K % 8 == 0 on a geometry with a K tail (families q and
 * r: K >= one stage). */
int hgemm_mi355x_config_accepts_k(int config_id, int K);
/* > 0 when the geometry has a stream-K kernel (HGEMM_PLAN_STREAMK): workgroups of it one CU holds. */
int hgemm_mi355x_config_streamk(int config_id);
/* 1 when a HGEMM_PLAN_STREAMK plan of this geometry really runs as stream-K on (M, N, K) -- the family has the kernel, K has no
 * direct tail, the tile count fits the arrival-counter block -- 0 when hgemm_mi355x_launch would run the geometry's plain
 * data-parallel launch instead (it still returns HGEMM_OK: a degraded plan is a slower plan, not an error).  Tuners and
 * candidate generators ask this before they record a "stream-K" timing. */
int hgemm_mi355x_streamk_runs(int config_id, int M, int N, int K);

/* Split-K workspace.  Default: the library keeps one private device buffer per (device, stream) pair that
 * issued a split-K plan and grows it on first use of a bigger plan only (never in steady state; growing
 * frees the old buffer with hipFree, which synchronises the device).  Concurrent GEMMs on different streams
 * or devices therefore never share partial sums.
 * hgemm_mi355x_set_workspace lends ONE caller-owned buffer instead (NULL returns to the default): it is
 * bound to the device that is current at the call, used for every stream of that device -- so the caller
 * must not run split-K GEMMs concurrently on several streams while it is lent -- and its first 256 KiB are
 * zeroed here (tile arrival counters).  A plan that does not fit the lent buffer runs with splits = 1.
 * hgemm_mi355x_workspace_bytes gives a sufficient size for (M, N, splits); hgemm_mi355x_release_workspaces
 * frees the library-owned buffers (call with no GEMM in flight and no hipGraph that holds one still alive).
 *
 * hipGraph capture (launch-bound shapes: a 64x4096x64 call is ~3 us of kernel behind ~7 us of launch): every
 * entry point may be called on a capturing stream -- the call records kernel nodes only, nothing is allocated,
 * synchronised or memset during capture.  A split-K (or hybrid-tail) plan needs its workspace to exist before
 * the capture starts: call hgemm_mi355x_reserve_workspace(M, N, K, stream) for the largest shape that will be
 * captured on that stream (or run the shape once on it); otherwise the captured call runs with splits = 1.
 * A buffer that a capture has used is never freed by later growth (it is retired until
 * hgemm_mi355x_release_workspaces), so instantiated graphs stay valid.  Replays of graphs captured on one
 * stream share that stream's workspace: do not replay them concurrently on several streams.
 *
 * Cost and lifetime: a (device, stream) pair that ever ran a split-K / hybrid plan keeps its buffer (>= 8 MiB, grown
 * geometrically) until hgemm_mi355x_release_workspaces; destroying the stream frees nothing, so an application with many
 * short-lived streams calls hgemm_mi355x_release_stream_workspace(stream) (current device, nothing of that stream in
 * flight) before hipStreamDestroy.  Allocation and growth switch the calling thread's stream-capture mode to relaxed
 * around hipMalloc / hipFree, so a capture that ANOTHER stream runs in global mode is not invalidated by them. */
int hgemm_mi355x_set_workspace(void* device_ptr, size_t bytes);
int hgemm_mi355x_release_stream_workspace(void* stream);
size_t hgemm_mi355x_workspace_bytes(int M, int N, int splits);
/* Exactly what hgemm_mi355x_launch asks for with this plan on (M, N, K) (0: none) -- the config-aware form: a stream-K plan needs
 * 2 x workgroups x BM x BN floats (a few MiB), where the geometry-blind bound above must assume 256 x 128 tiles and 1024 workgroups. */
size_t hgemm_mi355x_plan_workspace_bytes(int config_id, int splits, int M, int N, int K);
int hgemm_mi355x_reserve_workspace(int M, int N, int K, void* stream);
int hgemm_mi355x_release_workspaces(void);

const char* hgemm_mi355x_strerror(int status);
int hgemm_mi355x_last_hip_error(void);   /* hipError_t behind the calling thread's last HGEMM_ERR_HIP */
const char* hgemm_mi355x_version(void);

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
/* On-disk cache of the search results (round 6).  The reference repeats the search in every benchmarking process
 * (benchmarking_offline.py:71-84: find_best_algo_* right after the handles are created), 7 + 1 processes per shape; over a
 * 1000-shape grid x two accumulate trees x two modes that is the same search eight thousand times.  With a cache file -- the
 * environment's HGEMM_AUTOTUNE_CACHE, or hgemm_hipblaslt_autotune_set_cache(path) (NULL / "" = none) -- find_best_* first looks the
 * problem (layout, M, N, K, compute type) up: a record searched with at least the current HGEMM_AUTOTUNE_MAX_SECONDS whose
 * hipBLASLt solution index resolves to the recorded solution name in the running library (an index is only valid for the build of
 * hipBLASLt that searched it: the torch wheel bundles its own, bin/hgemm_tune links /opt/rocm's -- records of both may share a
 * file) and which the library accepts for the problem is taken as the winner without timing anything;
 * otherwise the search runs and appends its winner (text, one line per problem: tn M N K compute16 algo_index best_ms candidates
 * warm timed budget_s solution_name).  hgemm_hipblaslt_autotune_from_cache(tn): 1 when the last find_best of that layout was a
 * cache hit; _cache_stats: records held, hits / misses of this process. */
int hgemm_hipblaslt_autotune_set_cache(const char* path);
int hgemm_hipblaslt_autotune_from_cache(int tn);
int hgemm_hipblaslt_autotune_cache_stats(int* hits, int* misses);
/* Introspection for reports: candidates tried / median ms of the winner (nn = 0, tn = 1). */
int hgemm_hipblaslt_autotune_candidates(int tn);
double hgemm_hipblaslt_autotune_best_ms(int tn);
/* 1 when the last hipBLASLt problem prepared with acc = HGEMM_ACC_FP16 for this layout found no
 * HIPBLAS_COMPUTE_16F kernel and runs with 32F compute instead (result files report it as
 * "hipblaslt_compute16_fallback"), 0 when 16F compute is in use or acc was FP32, -1 before any call.
 * which: 0 = heuristic, 1 = autotune; tn: 0 = nn, 1 = tn. */
int hgemm_hipblaslt_compute16_fallback(int which, int tn);

/* Device helper used by the autotune baseline and the native tools: fill `n` fp16 values with
 * N(0,1) samples (counter-based generator; `seed` makes runs reproducible). */
int hgemm_fill_normal_f16(void* device_ptr, size_t n, unsigned long long seed, void* stream);

/* ---- measurement helpers (bench.py; the reference times with torch events around each call,
 * benchmarking_utils.py:23-31) ------------------------------------------------------------------
 * hgemm_mi355x_time_next_launch arms a one-shot hook: the next GEMM call of the calling thread puts
 * the two events on its main kernel's own dispatch packet, so hgemm_mi355x_event_elapsed_us returns
 * that plan's device time as rocprofv3 reports it (event-record marker packets around a launch
 * add ~6 us of queue gaps on MI355X).  Plans with several kernels (two-pass split-K, hybrid tail) carry the
 * start event on their first dispatch and the stop event on their last one, so the combine is included.
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
