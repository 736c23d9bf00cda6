// Vendor baselines behind the C ABI: rocBLAS rocblas_gemm_ex, hipBLASLt heuristic, hipBLASLt
// autotune.  They stand where the reference has cuBLAS / cuBLASLt
// (cublas/{fp16,fp32}/hgemm_cublas.cu, hgemm_cublaslt_heuristic.cu, hgemm_cublaslt_auto_tuning.cu)
// and follow the same protocol so that "speedup vs baseline" means the same thing:
//   * row-major C = A.B is issued as the column-major product C^T = B^T.A^T
//     (NN: B as [N x K] ld N, A as [K x M] ld K; TN: b_col_major as op(T) with ld K);
//   * heuristic = top-1 of 4 requested, cached per problem;
//   * autotune  = up to 100 heuristic candidates, 50 warm-up + 100 timed rounds, candidate order
//     shuffled every round, fresh N(0,1) operands every round, median per candidate.
// Differences, on purpose: the workspace is an explicit size_t (the reference's heuristic path
// overflows `20 * 1024 * 1024 * 1024` to 0, hgemm_cublaslt_heuristic.cu:18), and the autotune is
// time-boxed for very large shapes (HGEMM_AUTOTUNE_MAX_SECONDS, default 30 s per layout).
// Round 6: the search result can be kept on disk (HGEMM_AUTOTUNE_CACHE=<file> or hgemm_hipblaslt_autotune_set_cache): one line per
// (layout, M, N, K, compute type) with the winner's hipBLASLt solution index, so that a 1000-shape sweep pays the search once per
// problem instead of once per (sweep, process) -- the reference re-runs it in every benchmarking process
// (benchmarking_offline.py:71-84), which is what boxed rounds 2-4's sweeps to a 0.05 s search.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../../include/hgemm_mi355x.h"

namespace {

typedef _Float16 f16;

constexpr size_t kLtWorkspaceBytes = (size_t)256 << 20;  // 256 MiB, explicit size_t

// ---------------------------------------------------------------------------------------------
// N(0,1) fp16 fill: counter-based (splitmix64 -> Box-Muller), 2 values per 64-bit draw.
__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) fill_normal_f16_kernel(f16* out, size_t n, uint64_t seed) {
  const size_t pairs = (n + 1) / 2;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < pairs;
       p += (size_t)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix64(seed ^ splitmix64(p));
    const float u1 = ((uint32_t)(r >> 32) + 1.0f) * (1.0f / 4294967296.0f);  // (0,1]
    const float u2 = (uint32_t)r * (1.0f / 4294967296.0f);
    const float rad = sqrtf(-2.0f * __logf(u1));
    float s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    out[2 * p] = (f16)(rad * c);
    if (2 * p + 1 < n) out[2 * p + 1] = (f16)(rad * s);
  }
}

// ---------------------------------------------------------------------------------------------
rocblas_handle g_rocblas = nullptr;

int rocblas_run(bool tn, const void* a, const void* b, void* c, int M, int N, int K, int acc,
                void* stream) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  if (!g_rocblas) {
    int st = hgemm_rocblas_init();
    if (st != HGEMM_OK) return st;
  }
  if (rocblas_set_stream(g_rocblas, (hipStream_t)stream) != rocblas_status_success)
    return HGEMM_ERR_BACKEND;
  const float alpha32 = 1.0f, beta32 = 0.0f;
  const f16 alpha16 = (f16)1.0f, beta16 = (f16)0.0f;
  const bool h = (acc == HGEMM_ACC_FP16);
  const void* alpha = h ? (const void*)&alpha16 : (const void*)&alpha32;
  const void* beta  = h ? (const void*)&beta16 : (const void*)&beta32;
  const rocblas_datatype ct = h ? rocblas_datatype_f16_r : rocblas_datatype_f32_r;
  rocblas_status rs = rocblas_gemm_ex(
      g_rocblas, tn ? rocblas_operation_transpose : rocblas_operation_none, rocblas_operation_none,
      N, M, K, alpha, b, rocblas_datatype_f16_r, tn ? K : N, a, rocblas_datatype_f16_r, K, beta, c,
      rocblas_datatype_f16_r, N, c, rocblas_datatype_f16_r, N, ct, rocblas_gemm_algo_standard, 0, 0);
  return rs == rocblas_status_success ? HGEMM_OK : HGEMM_ERR_BACKEND;
}

// ---------------------------------------------------------------------------------------------
// One hipBLASLt problem (layout x shape x compute type) with its descriptors and chosen algo.
struct LtProblem {
  hipblasLtMatmulDesc_t   op = nullptr;
  hipblasLtMatrixLayout_t a_desc = nullptr, b_desc = nullptr, c_desc = nullptr;
  hipblasLtMatmulAlgo_t   algo;
  size_t ws_needed = 0;
  bool   have_algo = false;
  int M = 0, N = 0, K = 0, acc = -1;
  bool tn = false;
  bool compute16 = false;   // descriptors built with HIPBLAS_COMPUTE_16F (alpha/beta are fp16 then)
  int  candidates = 0;
  double best_ms = 0.0;

  void destroy() {
    if (op) hipblasLtMatmulDescDestroy(op);
    if (a_desc) hipblasLtMatrixLayoutDestroy(a_desc);
    if (b_desc) hipblasLtMatrixLayoutDestroy(b_desc);
    if (c_desc) hipblasLtMatrixLayoutDestroy(c_desc);
    op = nullptr; a_desc = b_desc = c_desc = nullptr;   // nulled, unlike the reference (:51-60)
    have_algo = false; acc = -1; M = N = K = 0;
  }
  bool matches(bool tn_, int M_, int N_, int K_, int acc_) const {
    return op && tn == tn_ && M == M_ && N == N_ && K == K_ && acc == acc_;
  }
};

struct LtContext {
  hipblasLtHandle_t handle = nullptr;
  void* workspace = nullptr;
  LtProblem nn, tn;
};

LtContext g_heur, g_auto;

int lt_init(LtContext& ctx) {
  if (ctx.handle) return HGEMM_OK;
  if (hipblasLtCreate(&ctx.handle) != HIPBLAS_STATUS_SUCCESS) {
    ctx.handle = nullptr;
    return HGEMM_ERR_BACKEND;
  }
  if (hipMalloc(&ctx.workspace, kLtWorkspaceBytes) != hipSuccess) {
    hipblasLtDestroy(ctx.handle);
    ctx.handle = nullptr; ctx.workspace = nullptr;
    return HGEMM_ERR_HIP;
  }
  return HGEMM_OK;
}

int lt_destroy(LtContext& ctx) {
  ctx.nn.destroy();
  ctx.tn.destroy();
  if (ctx.handle) hipblasLtDestroy(ctx.handle);
  if (ctx.workspace) hipFree(ctx.workspace);
  ctx.handle = nullptr; ctx.workspace = nullptr;
  return HGEMM_OK;
}

// Build descriptors for the column-major product C^T[N x M] = op(B')[N x K] * A^T[K x M].
int lt_describe(LtProblem& p, bool tn, int M, int N, int K, int acc, hipblasComputeType_t compute) {
  p.destroy();
  const hipDataType scale = (compute == HIPBLAS_COMPUTE_16F) ? HIP_R_16F : HIP_R_32F;
  if (hipblasLtMatmulDescCreate(&p.op, compute, scale) != HIPBLAS_STATUS_SUCCESS) return HGEMM_ERR_BACKEND;
  const hipblasOperation_t opa = tn ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.op, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa));
  hipblasLtMatmulDescSetAttribute(p.op, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb));
  hipblasStatus_t s1 = tn ? hipblasLtMatrixLayoutCreate(&p.b_desc, HIP_R_16F, K, N, K)
                          : hipblasLtMatrixLayoutCreate(&p.b_desc, HIP_R_16F, N, K, N);
  hipblasStatus_t s2 = hipblasLtMatrixLayoutCreate(&p.a_desc, HIP_R_16F, K, M, K);
  hipblasStatus_t s3 = hipblasLtMatrixLayoutCreate(&p.c_desc, HIP_R_16F, N, M, N);
  if (s1 != HIPBLAS_STATUS_SUCCESS || s2 != HIPBLAS_STATUS_SUCCESS || s3 != HIPBLAS_STATUS_SUCCESS) {
    p.destroy();
    return HGEMM_ERR_BACKEND;
  }
  p.tn = tn; p.M = M; p.N = N; p.K = K; p.acc = acc;
  p.compute16 = (compute == HIPBLAS_COMPUTE_16F);
  return HGEMM_OK;
}

int lt_candidates(LtContext& ctx, LtProblem& p, int requested,
                  std::vector<hipblasLtMatmulHeuristicResult_t>& out) {
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return HGEMM_ERR_BACKEND;
  uint64_t ws = kLtWorkspaceBytes;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
  out.assign(requested, hipblasLtMatmulHeuristicResult_t());
  int returned = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(ctx.handle, p.op, p.b_desc, p.a_desc, p.c_desc,
                                                       p.c_desc, pref, requested, out.data(), &returned);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS) returned = 0;
  out.resize(returned);
  out.erase(std::remove_if(out.begin(), out.end(),
                           [](const hipblasLtMatmulHeuristicResult_t& r) {
                             return r.state != HIPBLAS_STATUS_SUCCESS || r.workspaceSize > kLtWorkspaceBytes;
                           }),
            out.end());
  return out.empty() ? HGEMM_ERR_NO_ALGO : HGEMM_OK;
}

// Describe + enumerate; fp16-accumulate falls back to fp32 compute when hipBLASLt has no
// COMPUTE_16F kernels for this problem (CDNA4 MFMA accumulates in fp32 regardless).
int lt_prepare(LtContext& ctx, LtProblem& p, bool tn, int M, int N, int K, int acc, int requested,
               std::vector<hipblasLtMatmulHeuristicResult_t>& cands) {
  if (!ctx.handle) return HGEMM_ERR_NOT_READY;
  int st = HGEMM_ERR_NO_ALGO;
  if (acc == HGEMM_ACC_FP16) {
    st = lt_describe(p, tn, M, N, K, acc, HIPBLAS_COMPUTE_16F);
    if (st == HGEMM_OK) st = lt_candidates(ctx, p, requested, cands);
  }
  if (st != HGEMM_OK) {
    st = lt_describe(p, tn, M, N, K, acc, HIPBLAS_COMPUTE_32F);
    if (st == HGEMM_OK) st = lt_candidates(ctx, p, requested, cands);
  }
  if (st != HGEMM_OK) p.destroy();
  return st;
}

int lt_matmul(LtContext& ctx, LtProblem& p, const hipblasLtMatmulAlgo_t* algo, const void* a,
              const void* b, void* c, hipStream_t stream) {
  const float alpha32 = 1.0f, beta32 = 0.0f;
  const f16 alpha16 = (f16)1.0f, beta16 = (f16)0.0f;
  const bool h = p.compute16;
  hipblasStatus_t st = hipblasLtMatmul(ctx.handle, p.op, h ? (const void*)&alpha16 : (const void*)&alpha32, b,
                                       p.b_desc, a, p.a_desc, h ? (const void*)&beta16 : (const void*)&beta32,
                                       c, p.c_desc, c, p.c_desc, algo, ctx.workspace, kLtWorkspaceBytes, stream);
  return st == HIPBLAS_STATUS_SUCCESS ? HGEMM_OK : HGEMM_ERR_BACKEND;
}

int heuristic_run(bool tn, const void* a, const void* b, void* c, int M, int N, int K, int acc,
                  void* stream) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  if (!g_heur.handle) {
    int st = lt_init(g_heur);
    if (st != HGEMM_OK) return st;
  }
  LtProblem& p = tn ? g_heur.tn : g_heur.nn;
  if (!p.matches(tn, M, N, K, acc) || !p.have_algo) {
    std::vector<hipblasLtMatmulHeuristicResult_t> cands;
    int st = lt_prepare(g_heur, p, tn, M, N, K, acc, 4, cands);  // top-1 of 4, as the reference
    if (st != HGEMM_OK) return st;
    p.algo = cands[0].algo;
    p.have_algo = true;
    p.candidates = (int)cands.size();
  }
  return lt_matmul(g_heur, p, &p.algo, a, b, c, (hipStream_t)stream);
}

double median_of(std::vector<float>& v) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  const size_t mid = v.size() / 2;
  return (v.size() % 2 == 0) ? 0.5 * (v[mid] + v[mid - 1]) : v[mid];
}

// ---- on-disk cache of autotune winners ----------------------------------------------------------------------------------------
// Text, one record per line:  tn M N K compute16 algo_index best_ms candidates warm timed budget_s solution_name
// The solution index is hipBLASLt's own (hipblaslt_ext::getIndexFromAlgo / getAlgosFromIndex): valid for the library build that
// wrote it -- a record whose index does not resolve to the recorded solution name in the running library, or that the library does
// not accept for the problem (matmulIsAlgoSupported), is ignored and the search runs again.  A record is reused only if it was searched with at least the budget the caller asks for now.
struct AutotuneRecord {
  int tn, M, N, K, compute16, algo_index, candidates, warm, timed;
  double best_ms, budget_s;
  std::string solution;
};
std::mutex g_cache_mutex;
std::string g_cache_path;
bool g_cache_path_set = false;     // hgemm_hipblaslt_autotune_set_cache was called (overrides the environment)
bool g_cache_loaded = false;
std::vector<AutotuneRecord> g_cache;
int g_cache_hits = 0, g_cache_misses = 0;
int g_last_from_cache[2] = {0, 0};   // [tn]: the last find_best of this layout was served from the cache

double autotune_budget_s() {
  const char* env = getenv("HGEMM_AUTOTUNE_MAX_SECONDS");
  return env ? atof(env) : 30.0;
}

void cache_load_locked() {
  if (g_cache_loaded) return;
  g_cache_loaded = true;
  g_cache.clear();
  if (!g_cache_path_set) {
    const char* env = getenv("HGEMM_AUTOTUNE_CACHE");
    g_cache_path = env ? env : "";
  }
  if (g_cache_path.empty()) return;
  FILE* f = fopen(g_cache_path.c_str(), "r");
  if (!f) return;
  static char line[8192];   // (hipBLASLt solution names run to ~700 characters)
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    AutotuneRecord r;
    static char name[4096];
    name[0] = 0;
    if (sscanf(line, "%d %d %d %d %d %d %lf %d %d %d %lf %4095s", &r.tn, &r.M, &r.N, &r.K, &r.compute16, &r.algo_index, &r.best_ms,
               &r.candidates, &r.warm, &r.timed, &r.budget_s, name) < 11) continue;
    r.solution = name;
    g_cache.push_back(r);   // (a later line of the same problem supersedes an earlier one: lookups scan from the back)
  }
  fclose(f);
}

// every record of the problem that was searched with at least the budget asked for now, newest first (a file may hold records
// of several hipBLASLt builds -- the torch wheel bundles its own, `hgemm_tune` links /opt/rocm's: cache_apply tells them apart)
std::vector<AutotuneRecord> cache_lookup(bool tn, int M, int N, int K, bool compute16, double budget_s) {
  std::vector<AutotuneRecord> out;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  cache_load_locked();
  for (auto it = g_cache.rbegin(); it != g_cache.rend(); ++it)
    if (it->tn == (int)tn && it->M == M && it->N == N && it->K == K && it->compute16 == (int)compute16 && it->budget_s + 1e-9 >= budget_s)
      out.push_back(*it);
  return out;
}

void cache_store(const AutotuneRecord& r) {
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  cache_load_locked();
  g_cache.push_back(r);
  if (g_cache_path.empty()) return;
  FILE* f = fopen(g_cache_path.c_str(), "a");
  if (!f) return;
  if (ftell(f) == 0)
    fprintf(f, "# hipBLASLt autotune winners (hgemm_baselines.hip): tn M N K compute16 algo_index best_ms candidates warm timed budget_s solution\n");
  fprintf(f, "%d %d %d %d %d %d %.6f %d %d %d %.3f %s\n", r.tn, r.M, r.N, r.K, r.compute16, r.algo_index, r.best_ms, r.candidates,
          r.warm, r.timed, r.budget_s, r.solution.empty() ? "-" : r.solution.c_str());
  fclose(f);
}

// the cached winner as a usable algo for the prepared problem p, or false
bool cache_apply(LtProblem& p, const AutotuneRecord& r) {
  if (r.algo_index < 0) return false;
  std::vector<int> idx{r.algo_index};
  std::vector<hipblasLtMatmulHeuristicResult_t> res;
  if (hipblaslt_ext::getAlgosFromIndex(g_auto.handle, idx, res) != HIPBLAS_STATUS_SUCCESS || res.empty()) return false;
  // A solution index belongs to ONE build of the hipBLASLt kernel library: the same number names another kernel in another build
  // (measured in round 6: a cache searched by a process that linked /opt/rocm's hipBLASLt was useless to the torch processes of the
  // sweep, whose wheel bundles its own).  The record is only taken when the index resolves to the solution NAME that was searched.
  if (!r.solution.empty() && r.solution != "-") {
    std::string now = hipblaslt_ext::getSolutionNameFromAlgo(g_auto.handle, res[0].algo);
    for (char& ch : now) if (ch == ' ' || ch == '\n' || ch == '\t') ch = '_';
    if (now.size() > 700) now.resize(700);
    if (now != r.solution) return false;
  }
  const float alpha32 = 1.0f, beta32 = 0.0f;
  const f16 alpha16 = (f16)1.0f, beta16 = (f16)0.0f;
  const bool h = p.compute16;
  size_t ws = 0;
  if (hipblaslt_ext::matmulIsAlgoSupported(g_auto.handle, p.op, h ? (const void*)&alpha16 : (const void*)&alpha32, p.b_desc, p.a_desc,
                                           h ? (const void*)&beta16 : (const void*)&beta32, p.c_desc, p.c_desc, res[0].algo,
                                           ws) != HIPBLAS_STATUS_SUCCESS || ws > kLtWorkspaceBytes)
    return false;
  p.algo = res[0].algo;
  p.have_algo = true;
  p.candidates = r.candidates;
  p.best_ms = r.best_ms;
  return true;
}

int autotune_find(bool tn, int M, int N, int K, int acc) {
  if (M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  if (!g_auto.handle) return HGEMM_ERR_NOT_READY;
  LtProblem& p = tn ? g_auto.tn : g_auto.nn;
  std::vector<hipblasLtMatmulHeuristicResult_t> cands;
  int st = lt_prepare(g_auto, p, tn, M, N, K, acc, 100, cands);
  if (st != HGEMM_OK) return st;
  g_last_from_cache[tn ? 1 : 0] = 0;
  for (const AutotuneRecord& rec : cache_lookup(tn, M, N, K, p.compute16, autotune_budget_s()))
    if (cache_apply(p, rec)) {
      g_last_from_cache[tn ? 1 : 0] = 1;
      ++g_cache_hits;
      return HGEMM_OK;
    }
  ++g_cache_misses;
  int n_algo = (int)cands.size();
  {
    // Time box, part 1 (before anything is allocated): when one round over all candidates would already take more
    // than a fifth of HGEMM_AUTOTUNE_MAX_SECONDS (estimated at 400 TFLOP/s), only the heuristic's top candidates
    // compete (never fewer than 4).  With the default box of 30 s this does not trigger below ~1e13 flop.
    const char* env = getenv("HGEMM_AUTOTUNE_MAX_SECONDS");
    const double box = env ? atof(env) : 30.0;
    const double est_s = 2.0 * M * N * (double)K / 4.0e14;
    if (box > 0 && est_s * n_algo > box / 5.0) {
      n_algo = std::max(4, std::min(n_algo, (int)(box / 5.0 / est_s)));
      cands.resize(n_algo);
    }
  }

  f16 *a = nullptr, *b = nullptr, *c = nullptr;
  if (hipMalloc(&a, (size_t)M * K * 2) != hipSuccess || hipMalloc(&b, (size_t)K * N * 2) != hipSuccess ||
      hipMalloc(&c, (size_t)M * N * 2) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
    p.destroy();
    return HGEMM_ERR_HIP;
  }
  hipStream_t stream;
  hipStreamCreate(&stream);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);

  // The reference runs 50 warm-up + 100 timed rounds over every candidate (hgemm_cublaslt_auto_tuning.cu:108-306).
  // HGEMM_AUTOTUNE_MAX_SECONDS (default 30) is a REAL time box: the first round is timed on the host clock
  // (it includes the operand refill, the per-call event sync and hipBLASLt's own host path, which dominate for
  // small problems) and the round counts shrink to fit, never below 2 + 3.
  int warm = 50, timed = 100;
  const char* box_env = getenv("HGEMM_AUTOTUNE_MAX_SECONDS");
  const double budget = box_env ? atof(box_env) : 30.0;
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<std::vector<float>> times(n_algo);
  std::vector<bool> failed(n_algo, false);
  std::mt19937 rng(std::random_device{}());
  uint64_t seed = ((uint64_t)std::random_device{}() << 32) | std::random_device{}();
  std::vector<int> order(n_algo);

  for (int round = 0; round < warm + timed; ++round) {
    hgemm_fill_normal_f16(a, (size_t)M * K, seed++, stream);
    hgemm_fill_normal_f16(b, (size_t)K * N, seed++, stream);
    hipStreamSynchronize(stream);
    std::iota(order.begin(), order.end(), 0);
    std::shuffle(order.begin(), order.end(), rng);
    // untimed launch of the last candidate in this round's order (reference :211-225)
    lt_matmul(g_auto, p, &cands[order[n_algo - 1]].algo, a, b, c, stream);
    hipStreamSynchronize(stream);
    for (int i = 0; i < n_algo; ++i) {
      const int idx = order[i];
      if (failed[idx]) continue;
      hipEventRecord(e0, stream);
      int rs = lt_matmul(g_auto, p, &cands[idx].algo, a, b, c, stream);
      hipEventRecord(e1, stream);
      hipEventSynchronize(e1);
      if (rs != HGEMM_OK) { failed[idx] = true; continue; }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      if (round >= warm) times[idx].push_back(ms);
    }
    if (round == 0 && budget > 0) {
      const double t_round = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
      const double full = t_round * (warm + timed);
      if (full > budget) {
        const double f = budget / full;
        warm = std::max(2, (int)(warm * f));
        timed = std::max(3, (int)(timed * f));
      }
    }
  }
  int best = -1;
  double best_ms = 1e30;
  for (int i = 0; i < n_algo; ++i) {
    if (failed[i] || times[i].empty()) continue;
    const double med = median_of(times[i]);
    if (med < best_ms) { best_ms = med; best = i; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(stream);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  if (best < 0) { p.destroy(); return HGEMM_ERR_NO_ALGO; }
  p.algo = cands[best].algo;
  p.have_algo = true;
  p.candidates = n_algo;
  p.best_ms = best_ms;
  {
    AutotuneRecord rec;
    rec.tn = tn; rec.M = M; rec.N = N; rec.K = K; rec.compute16 = p.compute16;
    rec.algo_index = hipblaslt_ext::getIndexFromAlgo(p.algo);
    rec.best_ms = best_ms; rec.candidates = n_algo; rec.warm = warm; rec.timed = timed; rec.budget_s = budget;
    rec.solution = hipblaslt_ext::getSolutionNameFromAlgo(g_auto.handle, p.algo);
    for (char& ch : rec.solution) if (ch == ' ' || ch == '\n' || ch == '\t') ch = '_';
    if (rec.solution.size() > 700) rec.solution.resize(700);
    if (rec.algo_index >= 0) cache_store(rec);
  }
  return HGEMM_OK;
}

int autotune_run(bool tn, const void* a, const void* b, void* c, int M, int N, int K, int acc,
                 void* stream) {
  if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  LtProblem& p = tn ? g_auto.tn : g_auto.nn;
  // Like the reference (hgemm_cublaslt_auto_tuning.cu:466-546) the algorithm must have been
  // selected by find_best_* for this very problem.
  if (!g_auto.handle || !p.matches(tn, M, N, K, acc) || !p.have_algo) return HGEMM_ERR_NOT_READY;
  return lt_matmul(g_auto, p, &p.algo, a, b, c, (hipStream_t)stream);
}

}  // namespace

extern "C" {

int hgemm_fill_normal_f16(void* device_ptr, size_t n, unsigned long long seed, void* stream) {
  if (!device_ptr && n) return HGEMM_ERR_BAD_ARG;
  if (n == 0) return HGEMM_OK;
  size_t blocks = ((n + 1) / 2 + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(fill_normal_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (f16*)device_ptr, n, (uint64_t)seed);
  return hipGetLastError() == hipSuccess ? HGEMM_OK : HGEMM_ERR_HIP;
}

int hgemm_rocblas_init(void) {
  if (g_rocblas) return HGEMM_OK;
  if (rocblas_create_handle(&g_rocblas) != rocblas_status_success) {
    g_rocblas = nullptr;
    return HGEMM_ERR_BACKEND;
  }
  return HGEMM_OK;
}

int hgemm_rocblas_destroy(void) {
  if (g_rocblas) {
    rocblas_destroy_handle(g_rocblas);
    g_rocblas = nullptr;
  }
  return HGEMM_OK;
}

int hgemm_rocblas_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* stream) {
  return rocblas_run(false, a, b, c, M, N, K, acc, stream);
}
int hgemm_rocblas_tn(const void* a, const void* bt, void* c, int M, int N, int K, int acc, void* stream) {
  return rocblas_run(true, a, bt, c, M, N, K, acc, stream);
}

int hgemm_hipblaslt_heuristic_init(void) { return lt_init(g_heur); }
int hgemm_hipblaslt_heuristic_destroy(void) { return lt_destroy(g_heur); }
int hgemm_hipblaslt_heuristic_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* s) {
  return heuristic_run(false, a, b, c, M, N, K, acc, s);
}
int hgemm_hipblaslt_heuristic_tn(const void* a, const void* bt, void* c, int M, int N, int K, int acc, void* s) {
  return heuristic_run(true, a, bt, c, M, N, K, acc, s);
}

int hgemm_hipblaslt_autotune_init(void) { return lt_init(g_auto); }
int hgemm_hipblaslt_autotune_destroy(void) { return lt_destroy(g_auto); }
int hgemm_hipblaslt_autotune_find_best_nn(int M, int N, int K, int acc) { return autotune_find(false, M, N, K, acc); }
int hgemm_hipblaslt_autotune_find_best_tn(int M, int N, int K, int acc) { return autotune_find(true, M, N, K, acc); }
int hgemm_hipblaslt_autotune_nn(const void* a, const void* b, void* c, int M, int N, int K, int acc, void* s) {
  return autotune_run(false, a, b, c, M, N, K, acc, s);
}
int hgemm_hipblaslt_autotune_tn(const void* a, const void* bt, void* c, int M, int N, int K, int acc, void* s) {
  return autotune_run(true, a, bt, c, M, N, K, acc, s);
}
int hgemm_hipblaslt_autotune_set_cache(const char* path) {
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  g_cache_path = path ? path : "";
  g_cache_path_set = true;
  g_cache_loaded = false;   // (re)read on the next find_best
  return HGEMM_OK;
}
int hgemm_hipblaslt_autotune_from_cache(int tn) { return g_last_from_cache[tn ? 1 : 0]; }
int hgemm_hipblaslt_autotune_cache_stats(int* hits, int* misses) {
  if (hits) *hits = g_cache_hits;
  if (misses) *misses = g_cache_misses;
  std::lock_guard<std::mutex> lk(g_cache_mutex);
  cache_load_locked();
  return (int)g_cache.size();
}
int hgemm_hipblaslt_autotune_candidates(int tn) { return (tn ? g_auto.tn : g_auto.nn).candidates; }
double hgemm_hipblaslt_autotune_best_ms(int tn) { return (tn ? g_auto.tn : g_auto.nn).best_ms; }
int hgemm_hipblaslt_compute16_fallback(int which, int tn) {
  LtContext& ctx = which ? g_auto : g_heur;
  const LtProblem& p = tn ? ctx.tn : ctx.nn;
  if (p.M == 0) return -1;
  return (p.acc == HGEMM_ACC_FP16 && !p.compute16) ? 1 : 0;
}

}  // extern "C"
