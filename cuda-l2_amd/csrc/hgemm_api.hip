// C ABI of the hot path (include/hgemm_mi355x.h): planning (tuned table -> analytic model),
// split-K workspace, launch.  Host-side cost per call is one table probe + one (or two)
// hipLaunchKernelGGL: the reference harness times host wall-clock around each call
// (benchmarking_utils.py:23-31), so a lean host path is part of the hot path.
#include "hgemm_launch.hpp"
#include "../../include/hgemm_mi355x.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

using namespace hgemm_mi355x;

namespace {

thread_local int g_last_hip_error = 0;   // per calling thread, like hipGetLastError (calls may come from any thread)
#ifdef HGEMM_ABLATION
int g_debug_flags = 0;  // tuner-only build: hgemm_mi355x_set_debug
#endif
#ifdef HGEMM_TIMELINE
// measurement build (lib_tl/): hgemm_mi355x_set_timeline lends a device buffer of `slots` records, launch n writes record
// n % slots (kTimelineWgs workgroups x HGEMM_TL_WORDS words each)
unsigned long long* g_timeline = nullptr;
int g_timeline_slots = 0;
unsigned g_timeline_count = 0;
constexpr int kTimelineWgs = 1024;
#endif

// ---- split-K workspace ------------------------------------------------------------------------
// One buffer per (device, stream): two GEMMs on different streams (or devices) never share slabs or
// arrival counters.  Layout: [kCounterBytes of tile arrival counters, zero between launches][fp32 slabs].
// Growth happens on first use of a bigger plan only (hipFree of the old buffer synchronises the device,
// so no kernel can still be using it); steady state is a mutex + a short linear search.
// hipGraph capture: a captured launch bakes the workspace address into its kernel node, so (i) nothing is
// allocated while the stream is capturing (a plan that does not fit what is there degrades to splits = 1,
// like a lent buffer that is too small; hgemm_mi355x_reserve_workspace sizes it beforehand) and (ii) a buffer
// that a capture has seen is never freed by growth: it is retired and lives until
// hgemm_mi355x_release_workspaces, so graphs instantiated earlier stay valid.
constexpr size_t kCounterBytes = (size_t)256 << 10;             // 65536 tiles
constexpr size_t kMaxFusedTiles = kCounterBytes / sizeof(unsigned);
struct Workspace {
  int device; hipStream_t stream;
  char* ptr; size_t bytes;      // whole allocation (counters + slabs)
  bool captured;                // some hipGraph holds ptr
};
std::mutex g_ws_mutex;
std::vector<Workspace> g_ws;
std::vector<void*> g_ws_retired;   // buffers replaced by growth while a graph may still reference them
// caller-lent buffer (hgemm_mi355x_set_workspace): used for every stream of the device that was current
// when it was lent; the caller promises not to run GEMMs concurrently on several streams then.
char*  g_lent_ptr = nullptr;
size_t g_lent_bytes = 0;
int    g_lent_device = -1;

// slab_bytes of fp32 slab space (+ counters) for a launch on `stream`; HGEMM_ERR_NO_WORKSPACE when a lent
// buffer is too small or the allocation fails -- the caller then degrades to a plan that needs none.
constexpr int HGEMM_ERR_NO_WORKSPACE_INTERNAL = -100;
int ensure_workspace(size_t slab_bytes, hipStream_t stream, float** slabs, unsigned** counters) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HGEMM_ERR_HIP;
  const size_t need = kCounterBytes + slab_bytes;
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (g_lent_ptr) {
    if (dev != g_lent_device || need > g_lent_bytes) return HGEMM_ERR_NO_WORKSPACE_INTERNAL;
    *counters = (unsigned*)g_lent_ptr;
    *slabs = (float*)(g_lent_ptr + kCounterBytes);
    return HGEMM_OK;
  }
  Workspace* w = nullptr;
  for (Workspace& e : g_ws)
    if (e.device == dev && e.stream == stream) { w = &e; break; }
  if (!w) {
    g_ws.push_back({dev, stream, nullptr, 0, false});
    w = &g_ws.back();
  }
  // (the legacy default stream cannot capture, and querying it while another stream captures is an error)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive;
  if (capturing) {
    if (need > w->bytes) return HGEMM_ERR_NO_WORKSPACE_INTERNAL;
    w->captured = true;
  }
  if (need > w->bytes) {
    // hipMalloc / hipFree are "unsafe" calls for stream capture: made while ANOTHER stream of the process captures in
    // global mode (this call's own stream does not, see above; the legacy stream cannot even be asked) they would
    // invalidate that capture.  The thread's capture mode is switched to relaxed around them, which is the documented
    // way for a library to allocate next to somebody else's capture.
    struct RelaxedCapture {
      hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
      bool ok;
      RelaxedCapture() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
      ~RelaxedCapture() { if (ok && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
    } relaxed;
    const size_t old = w->bytes;   // (read before the buffer is let go: the growth below is geometric in the OLD size)
    if (w->ptr && w->captured) {
      g_ws_retired.push_back(w->ptr);
      w->ptr = nullptr; w->bytes = 0; w->captured = false;
    } else if (w->ptr) {
      hipError_t e = hipFree(w->ptr);   // device-synchronising: nothing in flight still reads the old slabs
      if (e != hipSuccess) { g_last_hip_error = (int)e; return HGEMM_ERR_HIP; }
      w->ptr = nullptr; w->bytes = 0;
    }
    // Grow geometrically (first buffer: at least 8 MiB) so a sweep over shapes re-allocates O(log) times, while a
    // process with many short-lived streams does not pin 64 MiB for each of them (round 2 did).
    const size_t want = std::max(need, std::max<size_t>(old * 2, (size_t)8 << 20));
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { (void)hipGetLastError(); g_last_hip_error = (int)e; return HGEMM_ERR_NO_WORKSPACE_INTERNAL; }
    e = hipMemsetAsync(p, 0, kCounterBytes, stream);   // ordered in front of the first launch that uses them
    if (e != hipSuccess) { (void)hipFree(p); g_last_hip_error = (int)e; return HGEMM_ERR_HIP; }
    w->ptr = (char*)p; w->bytes = want;
  }
  *counters = (unsigned*)w->ptr;
  *slabs = (float*)(w->ptr + kCounterBytes);
  return HGEMM_OK;
}

// ---- tuned plans --------------------------------------------------------------------------------
struct TunedRow { int M, N, K; const char* cfg; int splits, group_m; };
const TunedRow g_tuned_rows[] = {
#include "hgemm_tuned_table.inc"
    {0, 0, 0, nullptr, 0, 0}};

struct TunedPlan { uint64_t key; int M, N, K; int cfg, splits, group_m; };   // key orders the index; M, N, K decide a hit
TunedPlan* g_tuned = nullptr;
int g_num_tuned = 0;
std::once_flag g_tuned_once;

// Sort / hash key only: the three fields overlap once a dimension reaches 2^21, so two shapes may share a key --
// every probe below compares M, N, K themselves (round 2 compared the key alone: plan(64, 68, 8388864) returned
// the tuned row of (64, 64, 256)).
inline uint64_t shape_key(int M, int N, int K) {
  return ((uint64_t)(uint32_t)M << 42) ^ ((uint64_t)(uint32_t)N << 21) ^ (uint64_t)(uint32_t)K;
}

void build_tuned_index() {
  const int rows = (int)(sizeof(g_tuned_rows) / sizeof(g_tuned_rows[0])) - 1;
  g_tuned = new TunedPlan[rows > 0 ? rows : 1];
  for (int i = 0; i < rows; ++i) {
    const int id = hgemm_mi355x_config_by_name(g_tuned_rows[i].cfg);
    if (id < 0) continue;  // stale row (geometry removed): fall back to the model
    g_tuned[g_num_tuned++] = {shape_key(g_tuned_rows[i].M, g_tuned_rows[i].N, g_tuned_rows[i].K), g_tuned_rows[i].M,
                              g_tuned_rows[i].N, g_tuned_rows[i].K, id, g_tuned_rows[i].splits, g_tuned_rows[i].group_m};
  }
  std::sort(g_tuned, g_tuned + g_num_tuned,
            [](const TunedPlan& a, const TunedPlan& b) { return a.key < b.key; });
}

// ---- analytic plan model --------------------------------------------------------------------------
// MI355X constants (MI355X_MICROARCH.md): 256 CUs, 4 SIMDs/CU, 160 KiB LDS/CU, 512 regs/lane/SIMD.
constexpr int    kCUs          = 256;
constexpr double kLaunchUs     = 2.0;    // host launch + dispatch of one kernel
constexpr double kBoundaryUs   = 3.7;    // dependent kernel boundary + launch of the split-K combine (fitted)
constexpr double kHbmBytesUs   = 4.6e6;  // ~4.6 TB/s sustained for mixed read/write (fitted)
constexpr double kCuFlopUs     = 4069.0 * 2000.0;  // fp16 MFMA flop per CU per us at ~2.0 GHz

// persistent workgroups of a stream-K plan: the plan's low 16 bits, or (0 / 1) one resident wave of them
inline int streamk_grid(const KernelEntry& e, int g_plan) {
  return g_plan > 1 ? std::min(g_plan, 4096) : kCUs * std::max(1, e.sk_wgs_per_cu);
}
// stages closer than this to a tile boundary are not worth a cut (the prologue of a segment is 2-3 stages deep)
inline int streamk_min_steps(const KernelEntry& e) { return e.kgran >= 256 ? 2 : e.kgran >= 128 ? 3 : 4; }

// Raster group height.  What matters for L2 reuse is the set of tiles an XCD runs CONCURRENTLY
// (32 CUs x workgroups per CU), not all the tiles it will ever get: consecutive logical ids fill a
// column of `g` tiles, so `c` concurrent tiles touch g A-panels and c/g B-panels; panel bytes are
// g*BM + (c/g)*BN rows, minimal at g = sqrt(c*BN/BM).  (Measured at 8192^3: g=4 1289 TF, g=16 1211.)
int default_group_m(const KernelEntry& e, int tiles_m, int tiles_n) {
  const int nw = e.wm * e.wn;
  const int wg_per_cu = std::max(1, std::min(160 * 1024 / e.lds_bytes, std::max(1, 8 / nw)));
  const long per_xcd_total = std::max<long>(1, ((long)tiles_m * tiles_n + NUM_XCD - 1) / NUM_XCD);
  const double conc = (double)std::min<long>(per_xcd_total, 32L * wg_per_cu);
  const double ideal = std::sqrt(conc * e.bn / e.bm);
  int g = 1;
  while (g * 2 <= ideal * 1.42 && g * 2 <= tiles_m) g *= 2;   // nearest power of two
  return std::max(1, std::min(g, tiles_m));
}

double model_us(const KernelEntry& e, int M, int N, int K, int splits) {
  const int tiles_m = (M + e.bm - 1) / e.bm, tiles_n = (N + e.bn - 1) / e.bn;
  const long wgs = (long)tiles_m * tiles_n * splits;
  const int nw = e.wm * e.wn;
  const int tm = e.bm / e.wm, tn = e.bn / e.wn;
  const int vgprs = tm * tn / 64 + 48 + (tm + tn) / e.mi * 4;
  const int waves_simd = std::max(1, std::min(8, 512 / std::max(vgprs, 64)));
  int conc = std::min(160 * 1024 / e.lds_bytes, std::max(1, waves_simd * 4 / nw));
  conc = (int)std::max<long>(1, std::min<long>(conc, (wgs + kCUs - 1) / kCUs));
  const long rounds = (wgs + (long)kCUs * conc - 1) / ((long)kCUs * conc);
  const int ksteps = (K / splits + BK - 1) / BK;
  // Constants fitted to the measured candidates of the 1000-shape tune (tuning/r01_grid_tune_*.jsonl):
  // geomean regret of the model's pick against the measured best 2.0 % (3.9 % before the fit).
  // MFMA efficiency falls with the wave tile's operand reuse (LDS bytes per flop); the software-
  // pipelined family ('s', one wave per SIMD) sustains ~1.5x the classic schedule's rate.
  // Round 2 (families q, r added; tuning/r02_grid_tune_run{A,B}, r02_skinny_tune_run1): the 128x128 members of
  // family q sustain the classic rate per flop (their gain is the pipelining, modelled by step_lat), and q beats
  // s by a few percent once a work item has >= 16 K-steps, s wins below (one coordinate computation per item).
  // Regret of the model's pick among the measured candidates, ties broken in table order: 5.5 % (6.8 % before).
  if (e.name[0] == 'w') {
    // family "w" (wave-direct, no LDS staging): a trip of four K = 32 slices is one round trip to memory (~0.9 us cold, less once
    // the rows stream), the "_k4" members walk K with four waves; MFMA and load issue are never what bounds these shapes
    const bool k4 = e.wm * e.wn == 1;
    const double trips = std::ceil((double)(K / splits) / 32.0 / (k4 ? 16.0 : 4.0));
    const double main_w = rounds * (0.9 + 0.45 * std::max(0.0, trips - 1.0) + (k4 ? 0.25 : 0.0));
    double bytes_w = 2.0 * ((double)M * K + (double)N * K + (double)M * N), extra_w = 0.0;
    if (splits > 1) {
      bytes_w += 8.0 * (double)M * N * splits;
      extra_w = kBoundaryUs + 4.0 * (double)M * N * (splits + 0.5) / kHbmBytesUs + 0.17 * splits;
    }
    // every wave fetches its own fragments: L2 -> CU traffic is (BM + BN) rows per wave tile, not per workgroup tile
    const double l2_bytes = 2.0 * (double)K * ((double)tiles_m * tiles_n) * (e.wm * e.wn) * (e.bm / e.wm + e.bn / e.wn);
    return kLaunchUs - 0.7 + std::max(main_w, std::max(bytes_w / kHbmBytesUs, l2_bytes / 2.0e7)) + extra_w;
  }
  const bool is_q = e.name[0] == 'q', is_q128 = is_q && e.bm == 128 && e.bn == 128;
  const char family = is_q ? 's' : e.name[0];   // 'q' = 's' with the early-A split
  const double reuse = (double)tm * tn / (tm + tn);
  // Round 3 (tuning/r03_late_tune_mi355x.jsonl, 706 candidates): the 8-wave 128x64 / 64x128 members of the classic family run at
  // 0.78 of this model's time where their 4-wave counterparts run at 1.07 -- two waves per SIMD hide the LDS-DMA issue stalls
  // the per-step latency term charges; the 192-wide q members sit on the family's common ratio (1.41 vs 1.39-1.42).
  const bool w8_mid = family == 't' && nw == 8 && e.bm * e.bn <= 128 * 64 && wgs <= 2L * kCUs;   // (the fitted domain: <= 512 workgroups)
  // a 192-wide q tile costs 0.87 of a 256 x 256 one for 0.75 of its flops (12288 x 1024 x 16384: 396 -> 342 us with as many
  // rounds; 1024 x 12288 x 12288: 256 -> 225): fewer flops per LDS-DMA piece and per fragment read, one of them without the staged
  // epilogue -- without this term the off-grid ranking takes them whenever they save a fraction of a round
  const bool q192 = is_q && (e.bm == 192 || e.bn == 192);
  const double eff = 0.62 * std::min(1.0, reuse / 51.0) * (is_q128 ? 1.0 : family == 's' ? (q192 ? 1.47 / 1.16 : 1.47) : w8_mid ? 1.4 : 1.0);
  const double step_tp = conc * (2.0 * e.bm * e.bn * BK) / (kCuFlopUs * eff);
  // per-K-step latency floor: barrier + LDS-DMA round trip (double-buffered rings expose all of it)
  const double step_lat = family == 's' ? 0.40 : (e.nbuf >= 3 ? 0.33 : 0.74);
  double main_us = rounds * (1.0 + ksteps * std::max(step_tp, step_lat));
  // Rows that are not a multiple of 128 bytes apart (K % 64 != 0: only shapes off the grid): every 128-byte row segment of an
  // LDS-DMA piece straddles two cache lines, and the kernels that are bound by piece issue rather than by MFMA time pay for it in
  // proportion to their pieces per MFMA cycle, x = 4 (BM + BN) / (BM BN).  Round 4, first measurements of families q with a K tail
  // (tuning/r04_ktail_candidates_mi355x.jsonl, K = 4440 / 7152 / 520): 128 x 256 tiles 1.40 us per K-step against 0.85 on the
  // grid (+65 %, x = 0.047), 256 x 192 +24 % (x = 0.037), 256 x 256 +7 % (x = 0.031); capped where latency, not issue, bounds the
  // small tiles.  Family r streams 256-512 contiguous bytes per row and is not charged.
  if ((2 * K) % 128 != 0 && e.name[0] != 'r') {
    const double x = 4.0 * (e.bm + e.bn) / ((double)e.bm * e.bn);
    main_us *= 1.0 + std::min(0.5, std::max(0.0, 25.0 * x - 0.675));
  }
  double bytes = 2.0 * ((double)M * K + (double)N * K + (double)M * N);
  double extra = 0.0;
  if (splits > 1) {
    bytes += 8.0 * (double)M * N * splits;
    extra = kBoundaryUs + 4.0 * (double)M * N * (splits + 0.5) / kHbmBytesUs + 0.17 * splits;
  }
  const double q_bias = (is_q && !is_q128) ? (K / splits >= 1024 ? 0.99 : 1.01) : 1.0;
  return (kLaunchUs + std::max(main_us, bytes / kHbmBytesUs) + extra) * q_bias;
}

// Stream-K plan (EPI_STREAMK) of G persistent workgroups: every workgroup walks ~tiles * stages / G pipeline stages at the
// family's step cost, a cut tile costs one slab round trip (write-through store, arrival, the completer reads the parts).
// Constants are the data-parallel model's; the tuner measures, this only prunes candidates and ranks off-grid corners.
double model_us_streamk(const KernelEntry& e, int M, int N, int K, int G) {
  const long tiles = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn);
  const int stages = (K + e.kgran - 1) / e.kgran;
  const long total = tiles * stages;
  G = (int)std::max<long>(1, std::min<long>(G, total));
  const int conc = (int)std::max<long>(1, std::min<long>(e.sk_wgs_per_cu, (G + kCUs - 1) / kCUs));
  const int tm = e.bm / e.wm, tn = e.bn / e.wn;
  const double reuse = (double)tm * tn / (tm + tn);
  const double eff = 0.62 * std::min(1.0, reuse / 51.0) * (e.wm * e.wn == 8 && e.bm * e.bn <= 128 * 64 ? 1.4 : 1.0);
  const double step_tp = conc * (2.0 * e.bm * e.bn * e.kgran) / (kCuFlopUs * eff);
  const double step_lat = (e.name[0] == 'r' ? 0.45 : (e.nbuf >= 3 ? 0.33 : 0.74)) * e.kgran / 64.0 * (e.name[0] == 'r' ? 0.5 : 1.0);
  const double per_wg = (double)((total + G - 1) / G);
  const long rounds = (G + (long)kCUs * conc - 1) / ((long)kCUs * conc);
  const double main_us = rounds * (1.0 + per_wg * std::max(step_tp, step_lat));
  const long cuts = total % G == 0 && (total / G) % stages == 0 ? 0 : std::min<long>(G, tiles * 2);   // partial segments
  const double fix_bytes = 2.0 * cuts * e.bm * e.bn * 4.0;
  const double bytes = 2.0 * ((double)M * K + (double)N * K + (double)M * N) + fix_bytes;
  return kLaunchUs + std::max(main_us, bytes / kHbmBytesUs) + (cuts ? 1.5 : 0.0);
}

// K the geometry accepts: a multiple of its stage depth, or any multiple of 8 for the families that zero-fill a
// partial last K-step themselves
// ... (classic family: the last LDS-DMA step is padded, so its step count rounds up), or accumulate the remainder from
// fragments loaded straight from global memory behind at least one whole stage (families q and r, "direct" tail: the step
// count rounds down and the remainder rides on the last split)
inline bool direct_tail(const KernelEntry& e) { return e.ktail && e.name[0] != 't'; }
inline bool k_ok(const KernelEntry& e, int K) { return K % e.kgran == 0 || (e.ktail && K % 8 == 0 && (!direct_tail(e) || K >= e.kgran)); }

// How a geometry cuts K for a split count the caller asked for: `steps` pipeline stages (rounded up for the classic family, whose
// last step may be partial; rounded DOWN for a direct tail, which rides on the last split), at most one split per stage, no empty
// split, chunks of whole stages.  Split s covers [s * k_chunk, min(K, (s + 1) * k_chunk)), the last one of a direct tail up to K.
struct KSplit { int steps, splits, k_chunk; bool tail_direct; };
inline KSplit split_k(const KernelEntry& e, int K, int splits) {
  KSplit r;
  r.tail_direct = direct_tail(e) && K % e.kgran != 0;
  r.steps = r.tail_direct ? K / e.kgran : (K + e.kgran - 1) / e.kgran;
  r.splits = std::max(1, std::min(splits, r.steps));
  const int per = (r.steps + r.splits - 1) / r.splits;
  r.splits = (r.steps + per - 1) / per;
  r.k_chunk = per * e.kgran;
  return r;
}

// Whether a HGEMM_PLAN_STREAMK plan of geometry e REALLY runs as stream-K on (M, N, K), workspace aside -- one predicate for the
// launch, the off-grid planner, the query below (tuner, candidate generators): the family has the kernel, the tile count fits the
// counter block, the stage sequence fits 30 bits, and K has no direct tail (families q / r: no stream-K kernel with one).
static bool streamk_really_runs(const KernelEntry& e, int M, int N, int K) {
  if (e.sk_wgs_per_cu <= 0 || !k_ok(e, K)) return false;
  const KSplit ks = split_k(e, K, 1);
  const long tiles = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn);
  return !ks.tail_direct && tiles <= (long)kMaxFusedTiles && tiles * ks.steps < (1L << 30);
}


void model_plan(int M, int N, int K, int* cfg, int* splits, int* group_m) {
  double best = 1e30;
  int bc = 0, bs = 1;
  const int ksteps = K / BK;
  for (int c = 0; c < g_num_kernels; ++c) {
    const KernelEntry& e = g_kernel_table[c];
    // Do not pick tiles that mostly compute padding.
    if (e.bm > M * 2 && e.bm > 32) continue;
    if (e.bn > N * 2 && e.bn > 32) continue;
    if (!k_ok(e, K)) continue;
    if ((e.name[0] == 's' || e.name[0] == 'q') && e.mi == 32) continue;   // experimental 32x32x16 members: explicit plans only
    // family "w" inside its domain only: a K of one or two pipeline steps, or a tiny output with a long K
    if (e.name[0] == 'w' && !(K <= 128 || (long)M * N <= 128L * 128L)) continue;
    for (int s = 1; s <= 64; s *= 2) {
      if (s > 1 && ksteps / s < 4) break;
      const double t = model_us(e, M, N, K, s);
      if (t < best) { best = t; bc = c; bs = s; }
    }
  }
  *cfg = bc; *splits = bs;
  const KernelEntry& e = g_kernel_table[bc];
  *group_m = default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
}

// Off-grid shapes, first choice: the tuned plans of the surrounding grid shapes (the 2 x 2 x 2 lattice corners around
// (M, N, K)), ranked for THIS shape by the analytic model -- the reference's advice for unlisted sizes is "use the
// nearest larger configuration" (README.md:83-86).  Leave-one-out on the round-2 tuning runs (tools/eval_planner_loo.py:
// every grid shape planned from its four nearest neighbours' winners, judged by its own measured candidates;
// tuning/r02_planner_loo.json): geomean regret 3.3 %, 90th percentile 10.7 %, against 3.9 % / 14.7 % for the model
// choosing among all geometries.
const int kLattice[] = {64, 128, 256, 512, 1024, 2048, 4096, 8192, 12288, 16384};
constexpr int kLatticeN = (int)(sizeof(kLattice) / sizeof(kLattice[0]));

const TunedPlan* find_tuned(int M, int N, int K) {
  const uint64_t key = shape_key(M, N, K);
  const TunedPlan* lo = std::lower_bound(g_tuned, g_tuned + g_num_tuned, key,
                                         [](const TunedPlan& p, uint64_t k) { return p.key < k; });
  for (; lo != g_tuned + g_num_tuned && lo->key == key; ++lo)
    if (lo->M == M && lo->N == N && lo->K == K) return lo;
  return nullptr;
}

// (round 6) every admissible corner plan with its modelled time, for the first-use selection off the grid: the runners-up of the
// ranking below are the plans whose order the model is least sure about -- what the box should decide (13272 x 512 x 4440: the
// model ranks q128x256 first, q256x256 at two splits measures 11 % faster)
struct RankedPlan { double us; int cfg, splits, group_m; };
bool neighbour_plan(int M, int N, int K, int* cfg, int* splits, int* group_m, RankedPlan* ranked = nullptr, int* n_ranked = nullptr) {
  auto record = [&](double us, int c, int sp, int g) {
    if (ranked && n_ranked && *n_ranked < 8) ranked[(*n_ranked)++] = RankedPlan{us, c, sp, g};
  };
  int br[3][2];
  const int dims[3] = {M, N, K};
  for (int d = 0; d < 3; ++d) {
    int lo = kLattice[0], hi = kLattice[kLatticeN - 1];
    for (int i = 0; i < kLatticeN; ++i) {
      if (kLattice[i] <= dims[d]) lo = kLattice[i];
      if (kLattice[kLatticeN - 1 - i] >= dims[d]) hi = kLattice[kLatticeN - 1 - i];
    }
    br[d][0] = lo; br[d][1] = hi;
  }
  double best = 1e30;
  bool found = false;
  for (int c = 0; c < 8; ++c) {
    const TunedPlan* p = find_tuned(br[0][c & 1], br[1][(c >> 1) & 1], br[2][(c >> 2) & 1]);
    if (!p || p->cfg < 0) continue;
    const KernelEntry& e = g_kernel_table[p->cfg];
    if (!k_ok(e, K)) continue;
    if ((e.bm > M * 2 && e.bm > 32) || (e.bn > N * 2 && e.bn > 32)) continue;   // mostly padding
    const int ksteps = std::max(1, K / e.kgran);
    // a stream-K corner plan keeps its form (the low bits are its workgroup count, not a split count) and is priced as such
    const bool sk_usable = streamk_really_runs(e, M, N, K);   // (direct K tail, > 65536 tiles: the launch would run data-parallel)
    if ((p->splits & HGEMM_PLAN_STREAMK) && sk_usable) {
      const double t = model_us_streamk(e, M, N, K, streamk_grid(e, p->splits & HGEMM_SPLITK_MASK));
      record(t, p->cfg, HGEMM_PLAN_STREAMK | (p->splits & HGEMM_SPLITK_MASK), default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn));
      if (t < best) {
        best = t; found = true;
        *cfg = p->cfg;
        *splits = HGEMM_PLAN_STREAMK | (p->splits & HGEMM_SPLITK_MASK);
        *group_m = default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
      }
      continue;
    }
    const int s = (p->splits & HGEMM_PLAN_STREAMK) ? 1 : std::max(1, std::min(p->splits & HGEMM_SPLITK_MASK, ksteps));
    const long wgs_here = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn) * s;
    // A "_k4" member of family w spends a whole four-wave workgroup on a 32 x 32 (or smaller) tile: it wins where the grid gave it
    // at most one workgroup per CU (256 x 1024 x 1024, 512 x 512 x 512: by 2 % over t32x64) and loses as soon as there are more
    // (512 x 1024 x 512: 7.2 us against 5.7; off the grid 256 x 1600 x 1024 at 400 workgroups 12.2 against 9.0 for t64x64,
    // 640 x 640 x 640 9.2 against 7.4 -- tuning/r04_retune_pass2_mi355x.jsonl, r03_offgrid_tune_mi355x.jsonl, r04 off-grid reports)
    if (e.name[0] == 'w' && e.wm * e.wn == 1 && wgs_here > kCUs) continue;
    // Family r is bound by what a CU can stream: its corner plans were tuned with one or two workgroups on EVERY CU.  A count
    // between one and 1.75 rounds of the chip leaves most CUs idle while a few run a second workgroup (64 x 14928 x 10624: 156 tiles
    // of 64 x 96 at two splits = 312 workgroups, 83.4 us, where the 64 x 128 corner plan's 234 take 63.8)
    if (e.name[0] == 'r' && wgs_here > kCUs && wgs_here < kCUs * 7 / 4) continue;
    // the 8-wave mid tiles were tuned (and the model fitted) for at most two workgroups per CU: beyond that the larger tiles of
    // another corner win (1332 x 3108 x 4440: 525 tiles of 64 x 128 measured 101 us against 82 us for the 256 x 256 corner plan)
    if (e.name[0] == 't' && e.wm * e.wn == 8 && e.bm * e.bn <= 128 * 64 &&
        (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn) * s > 2L * kCUs) continue;
    const double t = model_us(e, M, N, K, s);
    {
      int sp = s > 1 ? (s | (p->splits & HGEMM_SPLITK_FUSED)) : 1;
      if (e.name[0] == 'r' && K % 64 == 0) sp |= p->splits & (HGEMM_PLAN_RS_XCD_STAGGER | HGEMM_PLAN_RS_NT_LOADS);
      if (e.name[0] == 'q') sp |= p->splits & (HGEMM_PLAN_XCD_STAGGER | HGEMM_PLAN_PHASE_OFFSET | HGEMM_PLAN_PHASE_OFFSET4);
      record(t, p->cfg, sp, default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn));
    }
    if (t < best) {
      best = t; found = true;
      *cfg = p->cfg;
      *splits = s > 1 ? (s | (p->splits & HGEMM_SPLITK_FUSED)) : 1;
      // family r's load flags travel with the corner plan (they belong to the access pattern of the shape class, not to the shape)
      // -- while the rows stay 128-byte aligned (K % 64 == 0).  With rows that straddle cache lines a non-temporal load drops the
      // half line the next K stage of the same row needs again, and the flags cost instead of paying: 64 x 16384 x 9160 (stride
      // 18320 B) r64x128_k128 split 2 64.0 us plain / 70.6 with both flags, r64x64_k256 66.8 / 70.3, the corner plan itself
      // (r64x128_k128_d, NT loads) 73.6 -> 67.7 without; with aligned rows off the grid they keep paying (16000 x 128 x 16000
      // 0.81 -> 0.79 of hipBLASLt without them, 128 x 16000 x 16000 0.99 -> 0.95, 64 x 14928 x 10624 0.77 -> 0.73:
      // tuning/r04_ktail_candidates_mi355x.jsonl, r04_offgrid_plan_report_call_j3_no_r_flags_mi355x.jsonl)
      if (e.name[0] == 'r' && K % 64 == 0) *splits |= p->splits & (HGEMM_PLAN_RS_XCD_STAGGER | HGEMM_PLAN_RS_NT_LOADS);
      // family q's schedule flags (round 5) travel with the corner plan as well: they belong to the shape class (a one-round plan of
      // long rows wants the K stagger, a walk of many short-K items the phase offset), cannot change a result, and fall away by
      // themselves where they do not apply (a K tail takes the ktail variant, a single round has nothing to offset)
      if (e.name[0] == 'q') *splits |= p->splits & (HGEMM_PLAN_XCD_STAGGER | HGEMM_PLAN_PHASE_OFFSET | HGEMM_PLAN_PHASE_OFFSET4);
      *group_m = default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
    }
  }
  // The grid's only multiple of 192 is 12288: a shape with another one (3072, 1536, 6144 ...) finds no corner that uses the 192-wide
  // persistent tiles although they may fit it exactly (3072^2: 144 tiles of 256 x 256 on 256 CUs, 192 of 192 x 256).  They join the
  // ranking with the split count of the best corner and unsplit (the model prices the members of family q on one scale:
  // measured / modelled 1.39-1.42 for all of them, tuning/r03_late_tune_mi355x.jsonl).
  if (found) {
    const int best_s = (*splits & HGEMM_PLAN_STREAMK) ? 1 : std::max(1, *splits & HGEMM_SPLITK_MASK), best_fused = *splits & HGEMM_SPLITK_FUSED;
    const char* extra[2] = {(M % 192 == 0 && N >= 128) ? "q192x256_w2x2" : nullptr,
                            (N % 192 == 0 && M >= 128) ? "q256x192_w2x2" : nullptr};
    for (const char* name : extra) {
      if (!name) continue;
      const int c = hgemm_mi355x_config_by_name(name);
      if (c < 0) continue;
      const KernelEntry& e = g_kernel_table[c];
      // (only where the 192-wide tiles fill at least half the chip: 1968 x 576 has 24 of them and measured 0.72x of its corner plan)
      if ((long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn) < kCUs / 2 || !k_ok(e, K)) continue;
      for (int s : {1, best_s}) {
        if (s > std::max(1, K / e.kgran)) continue;
        const double t = model_us(e, M, N, K, s);
        if (t < best) {
          best = t;
          *cfg = c;
          *splits = s > 1 ? (s | best_fused) : 1;
          *group_m = default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
        }
      }
    }
  }
  // A corner plan of family q was tuned on a shape whose tiles fill the resident workgroups; off the grid the same tile may leave
  // much of the last (or only) round empty (1332 x 3108 x 4440: 143 tiles of 128 x 256 on 256 workgroups = 84.7 us, where 204 items
  // of 256 x 192 at two splits take 60.0 and 156 of 256 x 256 at two splits 66.0 -- tuning/r04_ktail_candidates_mi355x.jsonl).
  // When the chosen q plan is a single round that fills less than 80 % of the resident workgroups, its siblings join the ranking
  // at one, two and four splits -- inside the family the model prices on one scale -- provided they do not fill their rounds worse.  (More
  // than one round is the hybrid tail schedule's case, hgemm_mi355x_launch; the 192-wide members keep their own, measured rule
  // above.)
  if (found && !(*splits & HGEMM_PLAN_STREAMK) && g_kernel_table[*cfg].name[0] == 'q' && g_kernel_table[*cfg].mi == 16) {
    auto fill_of = [&](const KernelEntry& e, int s) {
      const long items = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn) * s, cap = std::max(1, e.persistent_wgs);
      return (double)items / (double)(((items + cap - 1) / cap) * cap);
    };
    const KernelEntry& e0 = g_kernel_table[*cfg];
    const int s0 = std::max(1, *splits & HGEMM_SPLITK_MASK);
    const double fill0 = fill_of(e0, s0);
    // (not for a two-resident corner plan: its 512 slots are two per CU, 256 items of it already occupy every CU)
    if (fill0 < 0.8 && e0.persistent_wgs <= kCUs && (long)((M + e0.bm - 1) / e0.bm) * ((N + e0.bn - 1) / e0.bn) * s0 <= e0.persistent_wgs) {
      const int fused0 = *splits & HGEMM_SPLITK_FUSED;
      for (const char* name : {"q256x256_w2x2", "q256x128_w2x2", "q128x256_w2x2", "q128x128_w2x2_k128"}) {
        const int c = hgemm_mi355x_config_by_name(name);
        if (c < 0) continue;
        const KernelEntry& e = g_kernel_table[c];
        if (!k_ok(e, K) || (e.bm > M * 2 && e.bm > 32) || (e.bn > N * 2 && e.bn > 32)) continue;
        for (int s : {1, 2, 4}) {
          // (round 6: a split of a sibling needs >= 2048 of K per slice -- with 1024 the model took 40 tiles of 256 x 128 at four
          // single-launch splits for 1968 x 576 x 4096 when its corner moved to the K = 128 stages: 39.1 us against 26.9 for the corner plan)
          if (s > 1 && K / s < 2048) break;
          if (fill_of(e, s) < fill0) continue;
          const double t = model_us(e, M, N, K, s);
          if (t < best) {
            best = t;
            *cfg = c;
            *splits = s > 1 ? (s | fused0) : 1;
            *group_m = default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
          }
        }
      }
    }
  }
  return found;
}

// Plans of off-grid shapes are remembered per thread (the search above / model_plan cost 3-7 us of host time, as much
// as a launch-bound GEMM itself; a harness calls the same shape over and over).
struct PlanMemo { int M, N, K; int cfg, splits, group_m; };
thread_local PlanMemo t_plan_memo[32] = {};

// alignment rules of the LDS-DMA path (the K multiple depends on the geometry: k_ok())
bool mfma_path_ok(const void* a, const void* bt, const void* c, int M, int N, int K, int lda,
                  int ldb, int ldc) {
  if (K % 8 != 0 || (N & 3) != 0) return false;
  if ((lda & 7) || (ldb & 7) || (ldc & 3)) return false;
  if (((uintptr_t)a & 15) || ((uintptr_t)bt & 15) || ((uintptr_t)c & 7)) return false;
  (void)M;
  return true;
}

// ---- first-use plan selection on the box the library runs on (opt-in) ---------------------------------------------------------
// The reference's H100 tree re-tunes in situ: a kernel file times its variants on first invocation and keeps the fastest
// (kernels/h100_F32F16F16F32/64_4096_64.cu:623-690, 702-721).  This library ships ONE measured table, and rounds 3-4 showed rows
// whose ranking flips between boxes (the lock-step plans of family q: 40 % across boxes).  With
// HGEMM_MI355X_INSITU=1 in the environment (or hgemm_mi355x_set_insitu(1)) the FIRST call of hgemm_mi355x_fp32 / _fp16 for a
// shape times up to three plans on the call's own operands and stream -- the table's (or the planner's) plan and its
// alternates: for a grid shape the oracle-verified runners-up of the last re-tune (hgemm_tuned_alternates.inc), for any shape the
// plan with family q's K stagger / phase offset toggled (flags that cannot change a result's exactness) -- each once to warm up and
// kInsituReps times under dispatch-attached events, in interleaved rounds, and keeps the fastest for the (process, device); an alternate must beat the plan by
// 3 % to replace it.  The call then runs the winner, so C holds exactly one plan's result.  That first call synchronises the
// stream (it is meant for a warm-up phase: the harness's warm-up seconds, a model's first step); calls on a capturing stream and
// every later call take the recorded choice without timing anything.  Default: off -- the hot path is the table probe.
struct AltRow { int M, N, K; const char* cfg; int splits, group_m; };
const AltRow g_alt_rows[] = {
#include "hgemm_tuned_alternates.inc"
    {0, 0, 0, nullptr, 0, 0}};
struct PlanTriple { int cfg, splits, group_m; };
// One record per (device, shape): boxes rank plans differently and so may the devices of one box.
struct InsituChoice { int dev, M, N, K; PlanTriple plan; };
std::mutex g_insitu_mutex;
std::vector<InsituChoice> g_insitu;      // the process's choices (guarded by g_insitu_mutex)
std::atomic<unsigned> g_insitu_epoch{1}; // bumped when set_insitu(0) forgets them: invalidates the per-thread memos below
std::atomic<int> g_insitu_mode{-1};      // -1: environment not read yet (atomic: calls may come from any thread)
constexpr int kInsituReps = 5;
// Steady state must not take a process-wide lock per GEMM: a thread remembers the choices it has used (direct-mapped on the
// shape key like t_plan_memo; the mutex-guarded vector is only consulted on a memo miss).
struct InsituMemo { unsigned epoch; int dev, M, N, K; PlanTriple plan; };
thread_local InsituMemo t_insitu_memo[32] = {};

bool insitu_enabled() {
  int mode = g_insitu_mode.load(std::memory_order_relaxed);
  if (mode < 0) {
    const char* e = getenv("HGEMM_MI355X_INSITU");
    int expected = -1;
    g_insitu_mode.compare_exchange_strong(expected, (e && *e && *e != '0') ? 1 : 0);   // (a concurrent set_insitu wins)
    mode = g_insitu_mode.load(std::memory_order_relaxed);
  }
  return mode == 1;
}

// the plan + its alternates (at most three, no duplicates, every one launchable on this K)
int insitu_candidates(int M, int N, int K, PlanTriple out[3]) {
  int n = 0;
  PlanTriple p0;
  if (hgemm_mi355x_plan(M, N, K, &p0.cfg, &p0.splits, &p0.group_m) != HGEMM_OK) return 0;
  out[n++] = p0;
  auto add = [&](PlanTriple p) {
    if (n >= 3 || p.cfg < 0 || p.cfg >= g_num_kernels || !k_ok(g_kernel_table[p.cfg], K)) return;
    for (int i = 0; i < n; ++i)
      if (out[i].cfg == p.cfg && out[i].splits == p.splits && out[i].group_m == p.group_m) return;
    out[n++] = p;
  };
  for (const AltRow* r = g_alt_rows; r->cfg; ++r)
    if (r->M == M && r->N == N && r->K == K) add({hgemm_mi355x_config_by_name(r->cfg), r->splits, r->group_m});
  // off the grid (round 6): the runners-up among the tuned plans of the surrounding grid shapes, in the model's order -- every one of
  // them an oracle-verified plan of its own corner, run here through the same edge predication / K tail as the planner's choice
  std::call_once(g_tuned_once, build_tuned_index);
  if (p0.cfg >= 0 && !find_tuned(M, N, K) && K % 8 == 0 && (N & 3) == 0) {
    RankedPlan ranked[8];
    int nr = 0, c0, s0, g0;
    if (neighbour_plan(M, N, K, &c0, &s0, &g0, ranked, &nr)) {
      std::sort(ranked, ranked + nr, [](const RankedPlan& a, const RankedPlan& b) { return a.us < b.us; });
      for (int i = 0; i < nr; ++i) add({ranked[i].cfg, ranked[i].splits, ranked[i].group_m});
    }
  }
  if (p0.cfg >= 0 && !(p0.splits & HGEMM_PLAN_STREAMK)) {
    const KernelEntry& e = g_kernel_table[p0.cfg];
    if (e.name[0] == 'q' && e.mi == 16 && K % e.kgran == 0) {
      const int s = std::max(1, p0.splits & HGEMM_SPLITK_MASK);
      const long items = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn) * s;
      if (K / e.kgran / s >= 8) add({p0.cfg, p0.splits ^ HGEMM_PLAN_XCD_STAGGER, p0.group_m});
      if (items >= 2L * e.persistent_wgs && K <= 1024) add({p0.cfg, p0.splits ^ HGEMM_PLAN_PHASE_OFFSET, p0.group_m});
    }
  }
  return n;
}

inline InsituMemo& insitu_memo_slot(int dev, int M, int N, int K) {
  const uint64_t key = shape_key(M, N, K) * 0x9E3779B97F4A7C15ull + (uint64_t)dev;
  return t_insitu_memo[(key ^ (key >> 21) ^ (key >> 42)) & 31];
}

bool insitu_lookup(int dev, int M, int N, int K, PlanTriple* p) {
  const unsigned epoch = g_insitu_epoch.load(std::memory_order_acquire);
  InsituMemo& m = insitu_memo_slot(dev, M, N, K);
  if (m.epoch == epoch && m.dev == dev && m.M == M && m.N == N && m.K == K) { *p = m.plan; return true; }
  std::lock_guard<std::mutex> lk(g_insitu_mutex);
  for (const InsituChoice& c : g_insitu)
    if (c.dev == dev && c.M == M && c.N == N && c.K == K) {
      *p = c.plan;
      m = InsituMemo{epoch, dev, M, N, K, c.plan};
      return true;
    }
  return false;
}

// times the candidates on the call's operands; false (nothing recorded) when the stream captures or an event call fails
bool insitu_select(int dev, const void* a, const void* b, const void* bt, void* c, int M, int N, int K, void* stream, PlanTriple* choice) {
  hipStream_t s = (hipStream_t)stream;
  // A capture on THIS stream means nothing may be timed.  (The legacy stream is not asked: the query itself would invalidate a
  // global-mode capture of another stream, and so would the caller's launch on it.)  The event calls below would invalidate a
  // global-mode capture that ANOTHER stream of this thread is recording: they run under the relaxed capture mode, as
  // ensure_workspace's hipMalloc does.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (s && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return false;
  struct RelaxedCapture {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCapture() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
    ~RelaxedCapture() { if (ok && hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError(); }
  } relaxed;
  PlanTriple cand[3];
  const int n = insitu_candidates(M, N, K, cand);
  if (n == 0) return false;
  int best = 0;
  if (n > 1) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); (void)hipGetLastError(); return false; }
    // Interleaved rounds: one launch of every candidate per round, so that all of them see the same clock /
    // thermal history -- timing them one after the other right after a cold start favours whoever comes later.
    float t[3][kInsituReps];
    bool ok[3] = {true, true, true};
    for (int r = -1; r < kInsituReps; ++r)   // round -1 warms every candidate up untimed
      for (int i = 0; i < n; ++i) {
        if (!ok[i]) continue;
        if (r >= 0) { hgemm_mi355x::t_launch_timing.start = e0; hgemm_mi355x::t_launch_timing.stop = e1; }   // ride on the plan's own dispatch packets
        ok[i] = hgemm_mi355x_launch(cand[i].cfg, cand[i].splits, cand[i].group_m, a, b, bt, c, M, N, K, K, K, N, stream) == HGEMM_OK;
        if (ok[i] && r >= 0) ok[i] = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&t[i][r], e0, e1) == hipSuccess;
        if (!ok[i]) (void)hipGetLastError();
      }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (!ok[0]) return false;
    double us[3] = {1e30, 1e30, 1e30};
    for (int i = 0; i < n; ++i) {
      if (!ok[i]) continue;
      std::sort(t[i], t[i] + kInsituReps);
      us[i] = (double)t[i][kInsituReps / 2] * 1e3;   // median
    }
    for (int i = 1; i < n; ++i)
      if (us[i] < 0.97 * us[0] && us[i] < us[best]) best = i;   // an alternate has to beat the shipped plan by 3 %
  }
  *choice = cand[best];
  std::lock_guard<std::mutex> lk(g_insitu_mutex);
  for (const InsituChoice& ch : g_insitu)
    if (ch.dev == dev && ch.M == M && ch.N == N && ch.K == K) { *choice = ch.plan; return true; }   // another thread was faster: one choice per (process, device)
  g_insitu.push_back({dev, M, N, K, *choice});
  return true;
}

int run(int acc, const void* a, const void* b, const void* bt, void* c, int M, int N, int K,
        void* stream) {
  (void)acc;  // both accumulate modes use the fp32-accumulating MFMA (header comment)
  if (insitu_enabled() && a && c && bt && M > 0 && N > 0 && K > 0) {
    PlanTriple p;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    if (insitu_lookup(dev, M, N, K, &p) || insitu_select(dev, a, b, bt, c, M, N, K, stream, &p))
      return hgemm_mi355x_launch(p.cfg, p.splits, p.group_m, a, b, bt, c, M, N, K, K, K, N, stream);
  }
  int cfg, splits, group_m;
  int st = hgemm_mi355x_plan(M, N, K, &cfg, &splits, &group_m);
  if (st != HGEMM_OK) return st;
  return hgemm_mi355x_launch(cfg, splits, group_m, a, b, bt, c, M, N, K, K, K, N, stream);
}

}  // namespace

extern "C" {

int hgemm_mi355x_num_configs(void) { return g_num_kernels; }

const char* hgemm_mi355x_config_name(int id) {
  return (id >= 0 && id < g_num_kernels) ? g_kernel_table[id].name : nullptr;
}

int hgemm_mi355x_config_info(int id, int out[8]) {
  if (id < 0 || id >= g_num_kernels || !out) return HGEMM_ERR_BAD_ARG;
  const KernelEntry& e = g_kernel_table[id];
  out[0] = e.bm; out[1] = e.bn; out[2] = e.wm; out[3] = e.wn;
  out[4] = e.mi; out[5] = e.nbuf; out[6] = e.threads; out[7] = e.lds_bytes;
  return HGEMM_OK;
}

int hgemm_mi355x_config_k_granularity(int id) {
  if (id < 0 || id >= g_num_kernels) return 1;
  return g_kernel_table[id].ktail ? 8 : g_kernel_table[id].kgran;
}

int hgemm_mi355x_config_accepts_k(int id, int K) {
  if (id < 0 || id >= g_num_kernels) return K > 0 ? 1 : 0;   // the special ids take any K
  return K > 0 && k_ok(g_kernel_table[id], K) ? 1 : 0;
}

int hgemm_mi355x_config_by_name(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < g_num_kernels; ++i)
    if (std::strcmp(name, g_kernel_table[i].name) == 0) return i;
  return -1;
}

int hgemm_mi355x_plan(int M, int N, int K, int* config_id, int* splits, int* group_m) {
  if (M <= 0 || N <= 0 || K <= 0 || !config_id || !splits || !group_m) return HGEMM_ERR_BAD_ARG;
  std::call_once(g_tuned_once, build_tuned_index);
  if (const TunedPlan* hit = find_tuned(M, N, K)) {
    *config_id = hit->cfg; *splits = hit->splits; *group_m = hit->group_m;
    return HGEMM_OK;
  }
  if (K % 8 != 0 || (N & 3) != 0) {  // register-staged any-shape MFMA kernel (hgemm_kernel_rg.hpp)
    *config_id = HGEMM_CONFIG_RAGGED; *splits = 1; *group_m = 1;
    return HGEMM_OK;
  }
  const uint64_t key = shape_key(M, N, K);
  PlanMemo& memo = t_plan_memo[(key ^ (key >> 21) ^ (key >> 42)) & 31];
  if (memo.M != M || memo.N != N || memo.K != K) {   // (an empty slot holds M = 0, never a valid shape)
    PlanMemo m{M, N, K, 0, 1, 1};
    if (!neighbour_plan(M, N, K, &m.cfg, &m.splits, &m.group_m)) model_plan(M, N, K, &m.cfg, &m.splits, &m.group_m);
    memo = m;
  }
  *config_id = memo.cfg; *splits = memo.splits; *group_m = memo.group_m;
  return HGEMM_OK;
}

double hgemm_mi355x_model_us(int config_id, int splits, int M, int N, int K) {
  if (config_id < 0 || config_id >= g_num_kernels || splits < 1 || M <= 0 || N <= 0 || K <= 0) return -1.0;
  const KernelEntry& e = g_kernel_table[config_id];
  // `splits` as hgemm_mi355x_launch takes it: the plan flags are not a split count (round 3 passed them through: a row with
  // HGEMM_PLAN_NT_STORE was priced as 131073 splits)
  if ((splits & HGEMM_PLAN_STREAMK) && e.sk_wgs_per_cu > 0)
    return model_us_streamk(e, M, N, K, streamk_grid(e, splits & HGEMM_SPLITK_MASK));
  return model_us(e, M, N, K, std::max(1, splits & HGEMM_SPLITK_MASK));
}

int hgemm_mi355x_streamk_runs(int config_id, int M, int N, int K) {
  if (config_id < 0 || config_id >= g_num_kernels || M <= 0 || N <= 0 || K <= 0) return 0;
  return streamk_really_runs(g_kernel_table[config_id], M, N, K) ? 1 : 0;
}

int hgemm_mi355x_config_streamk(int config_id) {
  return (config_id >= 0 && config_id < g_num_kernels) ? g_kernel_table[config_id].sk_wgs_per_cu : 0;
}

int hgemm_mi355x_default_group(int config_id, int M, int N) {
  if (config_id < 0 || config_id >= g_num_kernels || M <= 0 || N <= 0) return 1;
  const KernelEntry& e = g_kernel_table[config_id];
  return default_group_m(e, (M + e.bm - 1) / e.bm, (N + e.bn - 1) / e.bn);
}

size_t hgemm_mi355x_workspace_bytes(int M, int N, int splits) {
  // upper bound over both split-K forms and every tile size (<= 256): counters + tile-padded fp32 slabs
  if (splits & HGEMM_PLAN_STREAMK) {   // two compact slabs per persistent workgroup; no stream-K kernel beyond 256 x 128 tiles, and
    // the launch never asks for more workgroups than the largest default grid (1024) unless the plan names more
    const int G = std::min(4096, std::max(1024, splits & HGEMM_SPLITK_MASK));
    return kCounterBytes + (size_t)2 * (size_t)G * 256 * 128 * sizeof(float);
  }
  if ((splits & HGEMM_SPLITK_MASK) <= 1) return 0;
  const size_t mp = ((size_t)M + 255) / 256 * 256, np = ((size_t)N + 255) / 256 * 256;
  return kCounterBytes + (size_t)(splits & HGEMM_SPLITK_MASK) * mp * np * sizeof(float);
}

size_t hgemm_mi355x_plan_workspace_bytes(int config_id, int splits, int M, int N, int K) {
  // exactly what hgemm_mi355x_launch will ask ensure_workspace for with this plan (0: the plan needs none)
  if (config_id < 0 || config_id >= g_num_kernels || M <= 0 || N <= 0 || K <= 0) return 0;
  const KernelEntry& e = g_kernel_table[config_id];
  if (!k_ok(e, K)) return 0;
  const long tiles = (long)((M + e.bm - 1) / e.bm) * ((N + e.bn - 1) / e.bn);
  if (splits & HGEMM_PLAN_STREAMK) {
    if (!streamk_really_runs(e, M, N, K)) return 0;
    const long total = tiles * split_k(e, K, 1).steps;
    const long G = std::max<long>(1, std::min<long>(streamk_grid(e, splits & HGEMM_SPLITK_MASK), std::max<long>(1, total / streamk_min_steps(e))));
    return kCounterBytes + (size_t)2 * G * e.bm * e.bn * sizeof(float);
  }
  const int sp = split_k(e, K, std::max(1, splits & HGEMM_SPLITK_MASK)).splits;
  if (sp <= 1) return 0;
  const bool fused = (splits & HGEMM_SPLITK_FUSED) && tiles <= (long)kMaxFusedTiles && e.has_fused &&
                     (double)tiles * sp * e.bm * e.bn * sizeof(float) < 2147483648.0;
  return kCounterBytes + (fused ? (size_t)tiles * sp * e.bm * e.bn : (size_t)sp * M * N) * sizeof(float);
}

int hgemm_mi355x_set_workspace(void* device_ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  if (device_ptr) {
    if (bytes < kCounterBytes) return HGEMM_ERR_BAD_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HGEMM_ERR_HIP;
    hipError_t e = hipMemset(device_ptr, 0, kCounterBytes);   // arrival counters start at zero
    if (e != hipSuccess) { g_last_hip_error = (int)e; return HGEMM_ERR_HIP; }
    g_lent_device = dev;
  }
  g_lent_ptr = (char*)device_ptr; g_lent_bytes = device_ptr ? bytes : 0;
  return HGEMM_OK;
}

int hgemm_mi355x_release_workspaces(void) {
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  int rc = HGEMM_OK;
  for (Workspace& w : g_ws)
    if (w.ptr && hipFree(w.ptr) != hipSuccess) rc = HGEMM_ERR_HIP;
  for (void* p : g_ws_retired)
    if (hipFree(p) != hipSuccess) rc = HGEMM_ERR_HIP;
  g_ws.clear();
  g_ws_retired.clear();
  return rc;
}

int hgemm_mi355x_release_stream_workspace(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return HGEMM_ERR_HIP;
  std::lock_guard<std::mutex> lk(g_ws_mutex);
  for (size_t i = 0; i < g_ws.size(); ++i) {
    if (g_ws[i].device != dev || g_ws[i].stream != (hipStream_t)stream) continue;
    int rc = HGEMM_OK;
    if (g_ws[i].ptr) {
      if (g_ws[i].captured) g_ws_retired.push_back(g_ws[i].ptr);   // a graph may still hold it: lives until release_workspaces
      else if (hipFree(g_ws[i].ptr) != hipSuccess) rc = HGEMM_ERR_HIP;
    }
    g_ws.erase(g_ws.begin() + (long)i);
    return rc;
  }
  return HGEMM_OK;
}

int hgemm_mi355x_reserve_workspace(int M, int N, int K, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  int cfg = 0, splits = 1, group = 1;
  const int st = hgemm_mi355x_plan(M, N, K, &cfg, &splits, &group);
  if (st != HGEMM_OK) return st;
  // split-K slabs of the plan, or the hybrid tail's compact slabs (one 256x256 fp32 tile per resident workgroup at most)
  size_t slab = cfg >= 0 ? hgemm_mi355x_plan_workspace_bytes(cfg, splits, M, N, K) : 0;   // exactly what the launch will ask for
  slab = std::max(slab, kCounterBytes + (size_t)256 * 256 * 256 * sizeof(float)) - kCounterBytes;
  float* slabs = nullptr; unsigned* counters = nullptr;
  const int rc = ensure_workspace(slab, (hipStream_t)stream, &slabs, &counters);
  return rc == HGEMM_ERR_NO_WORKSPACE_INTERNAL ? HGEMM_ERR_NO_WORKSPACE : rc;
}

int hgemm_mi355x_launch(int config_id, int splits_arg, int group_m, const void* a, const void* b,
                        const void* b_col_major, void* c, int M, int N, int K, int lda, int ldb,
                        int ldc, void* stream) {
  struct DisarmTiming {  // the timing hook is one-shot whatever path (or error return) this call takes
    ~DisarmTiming() { hgemm_mi355x::t_launch_timing = hgemm_mi355x::LaunchTiming{}; }
  } disarm_timing;
  if (!a || !c || M <= 0 || N <= 0 || K <= 0) return HGEMM_ERR_BAD_ARG;
  if (config_id >= g_num_kernels || config_id < HGEMM_CONFIG_RAGGED) return HGEMM_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool want_fused = (splits_arg & HGEMM_SPLITK_FUSED) != 0;
  const bool want_streamk = (splits_arg & HGEMM_PLAN_STREAMK) != 0;
  int splits = splits_arg & HGEMM_SPLITK_MASK;

  GemmArgs g;
  g.A = (const f16*)a; g.Bt = (const f16*)b_col_major; g.C = (f16*)c; g.partial = nullptr; g.counters = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.k_chunk = K; g.splits = 1; g.tiles_m = g.tiles_n = 1; g.group_m = 1; g.items = 1;
  g.tail_first = 0; g.tail_tiles = 0;
  g.flags = ((splits_arg & HGEMM_PLAN_NT_STORE) ? 1 : 0) | ((splits_arg & HGEMM_PLAN_RS_XCD_STAGGER) ? 2 : 0) | ((splits_arg & HGEMM_PLAN_RS_NT_LOADS) ? 4 : 0) |
            ((splits_arg & HGEMM_PLAN_PHASE_OFFSET) ? 8 : 0) | ((splits_arg & HGEMM_PLAN_WAVE_PRIORITY) ? 16 : 0) | ((splits_arg & HGEMM_PLAN_PHASE_OFFSET4) ? 32 : 0);
  g.sk = StreamK{1, 0, 0, 1, FastDiv{0u, 0u, 0u}};
#ifdef HGEMM_ABLATION
  g.debug = g_debug_flags;
#endif
#ifdef HGEMM_TIMELINE
  g.timeline = (g_timeline && g_timeline_slots > 0)
                   ? g_timeline + (size_t)(g_timeline_count++ % (unsigned)g_timeline_slots) * kTimelineWgs * HGEMM_TL_WORDS
                   : nullptr;
#endif

  bool fast = config_id >= 0 && b_col_major && mfma_path_ok(a, b_col_major, c, M, N, K, lda, ldb, ldc);
  if (fast) {
    // 32-bit LDS-DMA offsets: (BM-1) rows * ld * 2 B + K * 2 B must stay below 4 GiB; beyond that the
    // register-staged kernel (64-bit addressing) takes over instead of an error.
    const KernelEntry& e = g_kernel_table[config_id];
    // 2 GiB where bit 31 of an offset is the out-of-range mark: the classic family (its descriptors end at 2 GiB), and a launch that
    // runs a direct K tail (families q / r, K % stage != 0); their whole-stage launches address 4 GiB
    const bool mark31 = e.name[0] == 't' || (e.ktail && k_ok(e, K) && split_k(e, K, 1).tail_direct);
    const double reach = mark31 ? 2147483648.0 : 4294967296.0;
    if ((double)e.bm * lda * 2.0 + K * 2.0 >= reach || (double)e.bn * ldb * 2.0 + K * 2.0 >= reach) fast = false;
    if (K % BK != 0 && !k_ok(e, K)) fast = false;   // a partial last K-step on a geometry that cannot take it: any-shape kernel
    // the LDS-staged epilogue addresses a wave tile through one buffer descriptor with 32-bit offsets
    if ((double)e.bm * ldc * 2.0 + (double)N * 2.0 >= 2147483648.0) fast = false;
  }
  if (!fast) {
    if (config_id != HGEMM_CONFIG_GENERIC && b_col_major) {
      if ((long)((M + 63) / 64) * ((N + 63) / 64) > 0x7fffffffL) return HGEMM_ERR_TOO_LARGE;
      launch_ragged(g, s, timing_slot(true, true));
    } else {
      if (!b) return HGEMM_ERR_BAD_ARG;
      launch_generic((const f16*)a, (const f16*)b, (f16*)c, M, N, K, lda, N, ldc, s, timing_slot(true, true));
    }
  } else {
    const KernelEntry& e = g_kernel_table[config_id];
    g.tiles_m = (M + e.bm - 1) / e.bm;
    g.tiles_n = (N + e.bn - 1) / e.bn;
    const long tiles = (long)g.tiles_m * g.tiles_n;
    // pipeline stages of this geometry along K (BK = 64, or 128 for the "_k128" members)
    if (!k_ok(e, K)) return HGEMM_ERR_BAD_ARG;
    const int kgran = e.kgran;
    const KSplit ks0 = split_k(e, K, 1);
    const bool tail_direct = ks0.tail_direct;
    const int ksteps = ks0.steps;   // (direct tail: whole stages; the rest rides on the last split)
    // Stream-K (HGEMM_PLAN_STREAMK; the low bits of `splits` are then the number of persistent workgroups): one launch, no
    // combine kernel.  Not available (family without the kernel, too many tiles for the counter block, no workspace): the
    // plan degrades to the geometry's plain data-parallel launch, like a split-K plan without workspace.
    if (want_streamk) {
      const long total = tiles * ksteps;
      if (streamk_really_runs(e, M, N, K)) {   // (else: the plain launch below; hgemm_mi355x_streamk_runs tells a caller beforehand)
        const int min_steps = streamk_min_steps(e);
        const int G = (int)std::max<long>(1, std::min<long>(streamk_grid(e, splits), std::max<long>(1, total / min_steps)));
        const size_t slab_bytes = (size_t)2 * G * e.bm * e.bn * sizeof(float);
        unsigned* counters = nullptr;
        const int st = slab_bytes < 2147483648ull ? ensure_workspace(slab_bytes, s, &g.partial, &counters) : HGEMM_ERR_NO_WORKSPACE_INTERNAL;
        if (st == HGEMM_OK) {
          g.counters = counters;
          g.group_m = std::max(1, std::min(group_m, g.tiles_m));
          g.items = (int)tiles;
          g.sk = make_streamk((int)tiles, ksteps, G, min_steps);
          set_raster_div(g);
          e.launch(g, G, s, EPI_STREAMK, timing_slot(true, true));
          hipError_t err2 = hipGetLastError();
          if (err2 != hipSuccess) { g_last_hip_error = (int)err2; return HGEMM_ERR_HIP; }
          return HGEMM_OK;
        }
        if (st != HGEMM_ERR_NO_WORKSPACE_INTERNAL) return st;
        g.partial = nullptr;
      }
      splits = 1;
    }
    splits = split_k(e, K, splits).splits;   // at most one per stage, no empty split
    g.group_m = std::max(1, std::min(group_m, g.tiles_m));
    if (tiles * splits > 0x7fffffffL) return HGEMM_ERR_TOO_LARGE;
    // Split-K: two-pass (slabs + combine kernel) by default; single-launch ("fused") on request, unless the
    // tile count exceeds the counter block or the kernel family has no fused epilogue.  No workspace (lent buffer too small, allocation failed) means
    // no split-K: the plan degrades to splits = 1 instead of failing.
    int epi = EPI_C16;
    if (splits > 1) {
      const bool fused = want_fused && tiles <= (long)kMaxFusedTiles && e.has_fused &&
                         (double)tiles * splits * e.bm * e.bn * sizeof(float) < 2147483648.0;   // 32-bit slab offsets
      const size_t slab_bytes = fused ? (size_t)tiles * splits * e.bm * e.bn * sizeof(float)
                                      : (size_t)splits * M * N * sizeof(float);
      unsigned* counters = nullptr;
      const int st = ensure_workspace(slab_bytes, s, &g.partial, &counters);
      if (st == HGEMM_OK) {
        epi = fused ? EPI_FUSED : EPI_SLAB;
        if (fused) g.counters = counters;
      } else if (st == HGEMM_ERR_NO_WORKSPACE_INTERNAL) {
        splits = 1; g.partial = nullptr;
      } else {
        return st;
      }
    }
    g.k_chunk = split_k(e, K, splits).k_chunk;   // (splits may have dropped to 1 above: no workspace)
    g.splits = splits;
    const long grid = tiles * splits;
    g.items = (int)grid;
    // Hybrid schedule for the persistent family (stream-K's data-parallel + tail form): when the tile
    // count is not a multiple of the resident workgroups, the last partial round would keep most CUs
    // idle for a whole tile time.  Instead the full rounds run as they are and the `tail` leftover
    // tiles are cut along K into floor(G / tail) slices each, one slice per workgroup, combined by a
    // small reduce over compact fp32 slabs.  (7168^3 with 256x256 tiles: 784 = 3 x 256 + 16.)
    const long G = e.persistent_wgs;
#ifdef HGEMM_ABLATION
    const bool hybrid_ok = !(g_debug_flags & 32);
#else
    const bool hybrid_ok = true;
#endif
    if (G > 0 && splits == 1 && tiles > G && tiles % G != 0 && hybrid_ok) {
      const long tail = tiles % G, full = tiles - tail;
      int S = (int)std::min<long>(G / tail, ksteps / 4);   // >= 4 K-steps per slice
      if (S >= 2) {
        const int per_t = (ksteps + S - 1) / S;
        S = (ksteps + per_t - 1) / per_t;
        // Worth it when it beats the partial round it replaces.  Measured on MI355X: a round with few
        // tiles runs at ~0.6x of a full round's tile time (no contention), and the tail pass costs its
        // K slice plus ~25 us of prologue / epilogue / two extra launches / combine
        // (7168^3: 702 -> 661 us, 4352^2 x 4096: 175 -> 149 us; 10000^2 x 1024 would lose 7 %).
        const double tile_us = model_us(e, e.bm, e.bn, K, 1) - kLaunchUs;
        if (0.6 * tile_us - tile_us / S > 25.0) {
          GemmArgs t = g;
          t.tail_first = (int)full; t.tail_tiles = (int)tail; t.splits = S; t.k_chunk = per_t * kgran;
          t.items = (int)tail * S;
          unsigned* unused = nullptr;
          // (no workspace just means: no hybrid schedule)
          if (ensure_workspace((size_t)t.items * e.bm * e.bn * sizeof(float), s, &t.partial, &unused) == HGEMM_OK) {
            g.items = (int)full;
            set_raster_div(g); set_raster_div(t);
            e.launch(g, (int)std::min<long>(full, G), s, EPI_C16, timing_slot(true, false));
            e.launch(t, (int)std::min<long>(t.items, G), s, EPI_SLAB, timing_slot(false, false));
            launch_tail_reduce(t, e.bm, e.bn, s, timing_slot(false, true));
            hipError_t err2 = hipGetLastError();
            if (err2 != hipSuccess) { g_last_hip_error = (int)err2; return HGEMM_ERR_HIP; }
            return HGEMM_OK;
          }
        }
      }
    }
    // persistent families walk their work items themselves: one resident wave of workgroups
    long launch_grid = (e.persistent_wgs > 0) ? std::min<long>(grid, e.persistent_wgs) : grid;
    // A two-resident member (persistent_wgs = 512) whose single-launch split-K variant no longer fits twice into a CU's 160 KiB
    // -- the fused epilogue adds the 64-byte vote word to the stage area: q192x128 / q128x192 with 80 KiB of stages -- is resident
    // once per CU: 512 workgroups would run as two sequential waves of 256, each paying its own prologue.  One wave of 256 walks
    // the same items.
    if (epi == EPI_FUSED && e.persistent_wgs > kCUs && 2L * (e.lds_bytes + 64) > 160L * 1024) launch_grid = std::min<long>(launch_grid, kCUs);
    set_raster_div(g);
    e.launch(g, (int)launch_grid, s, epi, timing_slot(true, epi != EPI_SLAB));
    if (epi == EPI_SLAB) launch_splitk_reduce(g.partial, g.C, M, N, ldc, splits, s, timing_slot(false, true));
  }
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) { g_last_hip_error = (int)err; return HGEMM_ERR_HIP; }
  return HGEMM_OK;
}

int hgemm_mi355x_set_insitu(int enable) {
  const int old = insitu_enabled() ? 1 : 0;
  g_insitu_mode.store(enable ? 1 : 0);
  if (!enable) {   // a later enable measures again (the epoch retires every thread's memo of the old choices)
    std::lock_guard<std::mutex> lk(g_insitu_mutex);
    g_insitu.clear();
    g_insitu_epoch.fetch_add(1, std::memory_order_release);
  }
  return old;
}

int hgemm_mi355x_insitu_enabled(void) { return insitu_enabled() ? 1 : 0; }

int hgemm_mi355x_insitu_candidates(int M, int N, int K, int config_id[3], int splits[3], int group_m[3]) {
  if (M <= 0 || N <= 0 || K <= 0 || !config_id || !splits || !group_m) return 0;
  PlanTriple c[3];
  const int n = insitu_candidates(M, N, K, c);
  for (int i = 0; i < n; ++i) { config_id[i] = c[i].cfg; splits[i] = c[i].splits; group_m[i] = c[i].group_m; }
  return n;
}

int hgemm_mi355x_insitu_choice(int M, int N, int K, int* config_id, int* splits, int* group_m) {
  PlanTriple p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  if (!config_id || !splits || !group_m || !insitu_lookup(dev, M, N, K, &p)) return 0;
  *config_id = p.cfg; *splits = p.splits; *group_m = p.group_m;
  return 1;
}

int hgemm_mi355x_fp32(const void* a, const void* b, const void* bt, void* c, int M, int N, int K,
                      void* stream) {
  return run(HGEMM_ACC_FP32, a, b, bt, c, M, N, K, stream);
}

int hgemm_mi355x_fp16(const void* a, const void* b, const void* bt, void* c, int M, int N, int K,
                      void* stream) {
  return run(HGEMM_ACC_FP16, a, b, bt, c, M, N, K, stream);
}

// Host-side self-check hook (tests/test_host_logic.py; not part of the public header): the raster map of the
// kernels evaluated on the host, with true divisions (use_fast = 0) or with the multipliers the launch path would
// pass (use_fast = 1).  out = {split, tile, tile_m, tile_n}.
int hgemm_mi355x_selfcheck_raster(int tiles_m, int tiles_n, int group_m, int tail_first, int tail_tiles, int bid, int use_fast,
                                  int out[4]) {
  if (tiles_m < 1 || tiles_n < 1 || group_m < 1 || bid < 0 || !out) return HGEMM_ERR_BAD_ARG;
  const RasterPos r = use_fast ? raster_fast(bid, tiles_m, tiles_n, group_m, tail_first, tail_tiles,
                                             make_raster_div(tiles_m, tiles_n, group_m, tail_tiles))
                               : raster_ref(bid, tiles_m, tiles_n, group_m, tail_first, tail_tiles);
  out[0] = r.split; out[1] = r.tile; out[2] = r.tile_m; out[3] = r.tile_n;
  return HGEMM_OK;
}
// the stream-K partition as the kernels evaluate it: out[0] = first stage of workgroup w's run (w = G: the total), out[1] = the
// workgroup that owns stage x
int hgemm_mi355x_selfcheck_streamk(int tiles, int steps, int G, int min_steps, int w, int x, int out[2]) {
  if (tiles < 1 || steps < 1 || G < 1 || min_steps < 1 || w < 0 || !out || (long)tiles * steps >= (1L << 30)) return HGEMM_ERR_BAD_ARG;
  const StreamK sk = make_streamk(tiles, steps, G, min_steps);
  out[0] = sk_start(sk, w, G);
  out[1] = (x >= 0 && (long)x < (long)tiles * steps) ? sk_owner(sk, x, G) : -1;
  return HGEMM_OK;
}
// how hgemm_mi355x_launch cuts K for (geometry, K, requested split count): out = {stages, splits, k_chunk, direct tail (0 / 1)}
int hgemm_mi355x_selfcheck_ksplit(int config_id, int K, int splits, int out[4]) {
  if (config_id < 0 || config_id >= g_num_kernels || K <= 0 || splits < 1 || !out || !k_ok(g_kernel_table[config_id], K)) return HGEMM_ERR_BAD_ARG;
  const KSplit r = split_k(g_kernel_table[config_id], K, splits);
  out[0] = r.steps; out[1] = r.splits; out[2] = r.k_chunk; out[3] = r.tail_direct ? 1 : 0;
  return HGEMM_OK;
}
unsigned hgemm_mi355x_selfcheck_fastdiv(unsigned n, unsigned d) { return d ? fast_div(n, make_fast_div(d)) : 0xFFFFFFFFu; }

const char* hgemm_mi355x_strerror(int status) {
  switch (status) {
    case HGEMM_OK: return "ok";
    case HGEMM_ERR_BAD_ARG: return "bad argument";
    case HGEMM_ERR_TOO_LARGE: return "operand too large for 32-bit tile addressing";
    case HGEMM_ERR_HIP: return "HIP runtime error";
    case HGEMM_ERR_BACKEND: return "rocBLAS/hipBLASLt error";
    case HGEMM_ERR_NOT_READY: return "baseline not initialised / no algorithm selected";
    case HGEMM_ERR_NO_ALGO: return "hipBLASLt returned no usable algorithm";
    case HGEMM_ERR_NO_WORKSPACE: return "no split-K workspace (lent buffer too small, out of memory, or stream capturing)";
    default: return "unknown status";
  }
}

int hgemm_mi355x_last_hip_error(void) { return g_last_hip_error; }

// ---- measurement helpers (bench.py): kernel-exact timing without marker packets between launches ----
void* hgemm_mi355x_event_create(void) {
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
int hgemm_mi355x_event_destroy(void* ev) { return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? HGEMM_OK : HGEMM_ERR_HIP; }
int hgemm_mi355x_time_next_launch(void* start_event, void* stop_event) {
  if ((start_event == nullptr) != (stop_event == nullptr)) return HGEMM_ERR_BAD_ARG;
  hgemm_mi355x::t_launch_timing.start = (hipEvent_t)start_event;
  hgemm_mi355x::t_launch_timing.stop = (hipEvent_t)stop_event;
  return HGEMM_OK;
}
double hgemm_mi355x_event_elapsed_us(void* start_event, void* stop_event) {
  float ms = 0.f;
  if (hipEventSynchronize((hipEvent_t)stop_event) != hipSuccess) return -1.0;
  if (hipEventElapsedTime(&ms, (hipEvent_t)start_event, (hipEvent_t)stop_event) != hipSuccess) return -1.0;
  return (double)ms * 1e3;
}

#ifdef HGEMM_ABLATION
// tuner-only library build (lib_ablation/): results are garbage by construction, never shipped
int hgemm_mi355x_set_debug(int flags) { const int old = g_debug_flags; g_debug_flags = flags; return old; }
#endif

#ifdef HGEMM_TIMELINE
// measurement library build (lib_tl/), never shipped: see HGEMM_TL_* in hgemm_kernel.hpp.  The buffer must hold
// slots * 1024 * 16 eight-byte words; returns the number of launches recorded so far.
int hgemm_mi355x_set_timeline(void* device_ptr, int slots) {
  g_timeline = (unsigned long long*)device_ptr;
  g_timeline_slots = device_ptr ? slots : 0;
  const int n = (int)g_timeline_count;
  g_timeline_count = 0;
  return n;
}
#endif

const char* hgemm_mi355x_version(void) { return "hgemm_mi355x 0.2 (gfx950)"; }

}  // extern "C"
