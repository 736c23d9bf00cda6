// MI355X (gfx950 / CDNA4) HGEMM device code:  C[M,N] (fp16) = A[M,K] (fp16) * B[K,N] (fp16)
//
// This is the hot path that replaces the reference's per-shape CUDA kernels
// (reference: kernels/a100_F32F16F16F32/64_4096_64.cu:8-169 device kernel, :171-267 launcher).
// It is NOT a translation of that CuTe/cp.async/ldmatrix/mma.sync design. CDNA4 design:
//
//   * "TN" operand form: A is [M][K] row-major and the kernel reads B through the harness's
//     b_col_major tensor, i.e. Bt = [N][K] row-major (reference tools/utils.py:110-115).  Both
//     operands are therefore K-contiguous and every MFMA fragment is ONE 16-byte LDS read.
//   * HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): one wave instruction moves
//     8 tile rows x 128 B (BK = 64 halfs) = 1 KiB straight into LDS, no VGPR round trip.
//   * LDS image: row-major [rows][128 B]; the 16-byte chunk c of row r is stored at slot
//     c ^ ((r >> 1) & 7).  The DMA destination is lane-linear, so the permutation is applied to
//     the per-lane *source* address (inside one 128-B line: coalescing is preserved) and the
//     same XOR is applied on the fragment read.  With it every ds_read_b128 lane group hits 16
//     distinct 16-B slots of the 256-B bank row (conflict-free) for both MFMA shapes used.
//   * v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16, fp32 accumulate.  CDNA4 has no
//     fp16-accumulating MFMA, so the F16F16F16F16 and F32F16F16F32 entry points share kernels
//     (fp32 accumulate is a superset of the fp16-accumulate accuracy contract).
//   * Operands are swapped in the MFMA (D^T = Bt_frag x A_frag) so that each lane ends up with
//     4 consecutive N elements of one C row: the epilogue stores 8 B (fp16) / 16 B (fp32
//     split-K partial) per lane without a transposition pass.
//   * NBUF-deep LDS ring, ONE barrier per K-step, counted s_waitcnt vmcnt(N) so that NBUF-2
//     tiles stay in flight across the barrier.
//   * 1-D grid with an XCD-aware bijective block remap (8 XCDs, private 4 MiB L2 each) and
//     grouped rasterisation, so the tiles that share A/B panels run on the same XCD.
//   * M/N edges: rows are clamped on load and predicated on store -> no harness padding needed
//     (the reference needs harness-side zero padding, tools/utils.py:8-36).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace hgemm_mi355x {

using f16    = _Float16;
using f16x4  = __attribute__((ext_vector_type(4))) _Float16;
using f16x8  = __attribute__((ext_vector_type(8))) _Float16;
using f32x4  = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK        = 64;       // K elements per pipeline stage
constexpr int ROW_BYTES = BK * 2;   // one tile row in LDS = 128 B
constexpr int NUM_XCD   = 8;

// cache-policy bits of the LDS-DMA loads (buffer_load ... lds aux operand: 1 = sc0, 2 = nt, 16 = sc1).
// 0 = default policy; other values are build-time experiments (build.py HGEMM_EXTRA_HIPFLAGS).
// C stores: each C element is written once and never re-read by the kernel.  HGEMM_NT_STORE=1 marks the
// fp16 output stores non-temporal (streaming), so they do not displace the A/B panels from L2.
#ifndef HGEMM_NT_STORE
#define HGEMM_NT_STORE 0
#endif
// (HGEMM_NT_STORE=1 forces them for every plan: experiment builds.  Shipping builds take the plan's HGEMM_PLAN_NT_STORE bit,
// GemmArgs::flags bit 0 -- a wave-uniform branch around the store.)
#if HGEMM_NT_STORE
#define HGEMM_STORE_C(g, ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define HGEMM_STORE_C(g, ptr, val) \
  do { if ((g).flags & 1) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
#endif

#ifndef HGEMM_DMA_AUX
#define HGEMM_DMA_AUX 0
#endif

typedef __attribute__((address_space(3))) void lds_void_t;

// ---- exact unsigned division by a launch constant (Granlund-Montgomery, 32-bit): the raster map below divides a
// work-item id by three launch constants; a hardware-less integer division is ~40 instructions with a v_rcp in its
// dependency chain, five of them are most of a kernel's prologue and ~700 cycles per work item of a persistent
// kernel.  With HGEMM_FASTDIV the host passes multipliers (GemmArgs::rd) and a division is s_mul_hi + 4 ALU ops.
// tests/test_host_logic.py compares raster_fast with raster_ref on the host for every id of many launch shapes.
#ifndef HGEMM_FASTDIV
#define HGEMM_FASTDIV 1
#endif
struct FastDiv { uint32_t mul, sh1, sh2; };   // n / d = (t + ((n - t) >> sh1)) >> sh2,  t = mulhi(n, mul)
__host__ __device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t t = __umulhi(n, f.mul);
#else
  const uint32_t t = (uint32_t)(((uint64_t)n * f.mul) >> 32);
#endif
  return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
inline FastDiv make_fast_div(uint32_t d) {   // host side; d >= 1
  FastDiv f{0u, 0u, 0u};                     // d == 1: t = 0, n >> 0 >> 0
  if (d <= 1) return f;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;               // l = ceil(log2 d), 1 <= l <= 32
  f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
  f.sh1 = 1; f.sh2 = l - 1;
  return f;
}
// divisors of the raster map: ids per split (tiles, or tail_tiles in the hybrid tail pass), ids per raster group
// (group_m * tiles_n), group height (group_m) and the height of the last, partial group (tiles_m % group_m)
struct RasterDiv { FastDiv per_split, per_group, gm_full, gm_last; };
struct RasterPos { int split, tile, tile_m, tile_n; };   // K slice, output-tile id in raster order, tile coordinates

// id -> (split, tile, tile row, tile column), grouped raster: groups of group_m tile rows, column-major inside a group
__host__ __device__ __forceinline__ RasterPos raster_ref(int bid, int tiles_m, int tiles_n, int group_m, int tail_first, int tail_tiles) {
  const int per = tail_tiles > 0 ? tail_tiles : tiles_m * tiles_n;
  RasterPos r;
  r.split = bid / per;
  r.tile = (tail_tiles > 0 ? tail_first : 0) + (bid - r.split * per);
  const int gsz = group_m * tiles_n, grp = r.tile / gsz, first_m = grp * group_m;
  const int rest = tiles_m - first_m, gm = rest < group_m ? rest : group_m, tin = r.tile - grp * gsz;
  r.tile_m = first_m + tin % gm;
  r.tile_n = tin / gm;
  return r;
}
__host__ __device__ __forceinline__ RasterPos raster_fast(int bid, int tiles_m, int tiles_n, int group_m, int tail_first, int tail_tiles,
                                                          const RasterDiv& d) {
  const int per = tail_tiles > 0 ? tail_tiles : tiles_m * tiles_n;
  RasterPos r;
  r.split = (int)fast_div((uint32_t)bid, d.per_split);
  r.tile = (tail_tiles > 0 ? tail_first : 0) + (bid - r.split * per);
  const int gsz = group_m * tiles_n, grp = (int)fast_div((uint32_t)r.tile, d.per_group), first_m = grp * group_m;
  const int rest = tiles_m - first_m, gm = rest < group_m ? rest : group_m, tin = r.tile - grp * gsz;
  r.tile_n = (int)fast_div((uint32_t)tin, gm == group_m ? d.gm_full : d.gm_last);
  r.tile_m = first_m + (tin - r.tile_n * gm);
  return r;
}
inline RasterDiv make_raster_div(int tiles_m, int tiles_n, int group_m, int tail_tiles) {   // host side
  RasterDiv d;
  d.per_split = make_fast_div((uint32_t)(tail_tiles > 0 ? tail_tiles : tiles_m * tiles_n));
  d.per_group = make_fast_div((uint32_t)(group_m * tiles_n));
  d.gm_full = make_fast_div((uint32_t)group_m);
  d.gm_last = make_fast_div((uint32_t)(tiles_m % group_m ? tiles_m % group_m : 1));
  return d;
}

// ---- stream-K partition (EPI_STREAMK; the reference's H100 tree schedules 48 shapes with cutlass::gemm::StreamKScheduler,
// kernels/h100_F32F16F16F32/128_4096_16384.cu:79, 16384_512_2048.cu:71-73; every hipBLASLt kernel on this chip is one).
// The tiles x steps pipeline stages of a GEMM form ONE sequence, tile-major (tiles in raster order, a tile's K stages in
// order).  Workgroup w of G (after the XCD remap: an XCD owns a contiguous run of w) takes [sk_start(w), sk_start(w + 1)):
//   base(w) = w * q + min(w, r)          total = tiles * steps = q * G + r: contiguous runs that differ by at most one stage
//   a boundary closer than `min_steps` to a tile boundary is snapped onto it (a segment shorter than the pipeline prologue
//   costs more than the imbalance it removes); a tile with fewer than 2 * min_steps stages is never cut.
// snap() is monotone and so is base(): the starts are non-decreasing without a fix-up pass, some workgroups may get nothing.
// A run decomposes into segments (tile, [k0, k1)): at most the first and the last are partial.  A partial segment writes its
// fp32 partial to the compact slab 2 * w + (0: it opens the run, 1: it does not) and adds its stage count to the tile's arrival
// counter; whoever completes the count adds the tile's slabs in K order (= workgroup order: deterministic) and stores the
// tile.  Nobody waits for anybody: no co-residency assumption, no deadlock.  The closed form is evaluated by the host (slab
// sizing, tests: hgemm_mi355x_selfcheck_streamk), by every producer and by the combiner, so they agree by construction;
// tests/kernel_layout_model.py::streamk_partition is the same map in Python with its invariants.
struct StreamK {
  int steps;       // pipeline stages per tile (ceil(K / stage depth))
  int q, r;        // tiles * steps = q * G + r
  int min_steps;
  FastDiv div_steps;
};
__host__ __device__ __forceinline__ int sk_start(const StreamK& sk, int w, int G) {
  if (w <= 0) return 0;
  const int total = sk.q * G + sk.r;
  if (w >= G) return total;
  int x = w * sk.q + (w < sk.r ? w : sk.r);
  const int t = (int)fast_div((uint32_t)x, sk.div_steps), rem = x - t * sk.steps;
  if (sk.steps < 2 * sk.min_steps) x += (rem * 2 < sk.steps) ? -rem : sk.steps - rem;
  else if (rem < sk.min_steps) x -= rem;
  else if (sk.steps - rem < sk.min_steps) x += sk.steps - rem;
  return x;
}
// the workgroup whose run holds stage x (0 <= x < total): the LAST w with sk_start(w) <= x (its run is not empty then)
__host__ __device__ __forceinline__ int sk_owner(const StreamK& sk, int x, int G) {
  int lo = 0, hi = G;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sk_start(sk, mid, G) <= x) lo = mid; else hi = mid;
  }
  return lo;
}
inline StreamK make_streamk(int tiles, int steps, int G, int min_steps) {   // host side
  StreamK sk;
  const long total = (long)tiles * steps;
  sk.steps = steps; sk.q = (int)(total / G); sk.r = (int)(total % G); sk.min_steps = min_steps;
  sk.div_steps = make_fast_div((uint32_t)steps);
  return sk;
}

struct GemmArgs {
  const f16* A;    // [M][lda]
  const f16* Bt;   // [N][ldb]   (b_col_major: B transposed, K contiguous)
  f16*       C;    // [M][ldc]
  float*     partial;  // split-K fp32 slabs [splits][M][N]; unused when splits == 1
  int M, N, K;
  int lda, ldb, ldc;
  int k_chunk;     // K elements per split, multiple of BK
  int splits;
  int tiles_m, tiles_n;
  int group_m;     // rasterisation group height in tiles
  // Work-item space of this launch.  Normal launch: items = splits * tiles, item = split * tiles + tile.
  // Tail pass of the hybrid schedule (tail_tiles > 0): only tiles [tail_first, tail_first + tail_tiles)
  // are processed, each cut into `splits` K slices; item = slice * tail_tiles + (tile - tail_first) and
  // its fp32 partial goes to the COMPACT slab partial[item][BM][BN].
  int items;       // number of work items of this launch
  int tail_first, tail_tiles;
  // Single-launch split-K (EPI_FUSED): one arrival counter per output tile (zero before the launch, reset
  // by the last arriver) and compact per-item slabs partial[item][BM*BN] in the kernel's own lane order.
  unsigned* counters;
  int flags;       // bit 0: non-temporal fp16 C stores (HGEMM_PLAN_NT_STORE); bit 1 (HGEMM_PLAN_XCD_STAGGER) family r: K stagger per XCD instead of per tile, family q: the kstagger variant
                   // (HGEMM_PLAN_RS_XCD_STAGGER), bit 2 non-temporal loads of the streamed operand (HGEMM_PLAN_RS_NT_LOADS)
#if HGEMM_FASTDIV
  RasterDiv rd;    // multipliers for the raster map's divisions (set_raster_div on the host, after the fields above are final)
#endif
  StreamK sk;      // stream-K launches (EPI_STREAMK) only: the partition of the tile-major K-stage sequence (appended: the
                   // kernarg offsets of every field above are what the round-3 kernels were validated with)
#ifdef HGEMM_ABLATION
  int debug;       // tuner-only build (results are garbage): bit 0 skip steady-state LDS-DMA, bit 1 skip the
                   // epilogue stores, bit 2 cut every tile's K loop to two steps, bit 3 every tile stores to tile (0,0)
#endif
#ifdef HGEMM_TIMELINE
  unsigned long long* timeline;   // measurement build (lib_tl/): 16 words per workgroup, see HGEMM_TL_* below; may be null
#endif
};

// Measurement build only (-DHGEMM_TIMELINE, lib_tl/; hgemm_tune bench --timeline): wave 0 of every workgroup stamps the
// shader clock (s_memtime) at the seams of the kernel -- entry, pipeline primed, last MFMA of the last work item, last
// store issued, stores acknowledged -- and the 100 MHz wall clock (s_memrealtime) at entry and exit, so the head / K loop /
// epilogue / drain split of a launch and the gap between back-to-back launches can be read per workgroup.
#ifdef HGEMM_TIMELINE
#define HGEMM_TL_WORDS 16
// stamp -> 8-byte slot `slot` of the kernel's LDS scratch (lane 0 of wave 0 only): no SGPRs stay live across the K loop.
// The LDS accesses are written as asm on the 32-bit LDS address: a volatile C++ store through the generic pointer
// becomes a flat store with a vmcnt(0) behind it, which would drain the LDS-DMA pipeline at every stamp.
#define HGEMM_TL_PUT_(INSTR, slots, slot, tid)                                                            \
  do {                                                                                                    \
    unsigned long long v_;                                                                                \
    asm volatile(INSTR " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_) :: "memory");                            \
    if ((tid) == 0) {                                                                                     \
      const unsigned a_ = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(slots) + 8u * (slot); \
      asm volatile("ds_write_b64 %0, %1" :: "v"(a_), "v"(v_) : "memory");                                 \
    }                                                                                                     \
  } while (0)
#define HGEMM_TL_STAMP(slots, slot, tid) HGEMM_TL_PUT_("s_memtime", slots, slot, tid)
#define HGEMM_TL_REALTIME(slots, slot, tid) HGEMM_TL_PUT_("s_memrealtime", slots, slot, tid)
__device__ __forceinline__ unsigned long long hgemm_tl_get(const char* slots, int slot) {
  unsigned long long v;
  const unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(slots) + 8u * slot;
  asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
#else
#define HGEMM_TL_STAMP(slots, slot, tid) ((void)0)
#define HGEMM_TL_REALTIME(slots, slot, tid) ((void)0)
#endif
inline void set_raster_div(GemmArgs& g) {   // host: after tiles_m / tiles_n / group_m / tail_tiles are final
#if HGEMM_FASTDIV
  g.rd = make_raster_div(g.tiles_m, g.tiles_n, g.group_m, g.tail_tiles);
#else
  (void)g;
#endif
}

// Kernel-argument prefetch.  GemmArgs is passed by value: the kernel reads it from the launch's kernarg buffer with scalar
// loads, and hipcc places those where the values are first needed -- three dependent round trips for the three 64-byte
// lines of the struct, each a cold miss (every launch gets a fresh kernarg buffer).  Round-3 timeline: 2.6k cycles (1.6 us)
// pass between kernel entry and the first LDS-DMA piece with or without the multiplier raster map, i.e. the arithmetic is
// not what takes the time.  One dword of every line is requested at entry instead, all at once (ONE round trip): the
// compiler's own loads then hit the scalar cache.  (By-value structs cannot use the CP's kernarg preload.)
#ifndef HGEMM_KERNARG_PREFETCH
#define HGEMM_KERNARG_PREFETCH 1
#endif
template <int BYTES>
__device__ __forceinline__ void prefetch_kernargs() {
#if defined(__HIP_DEVICE_COMPILE__) && HGEMM_KERNARG_PREFETCH
  // one statement: the loads AND their wait (a scalar load writes its destination when it returns; the compiler would
  // consider an unused destination free again right behind the statement and the late write would corrupt its new value)
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned d0, d1, d2, d3;
  if constexpr (BYTES > 192)
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3) : "s"(ka));
  else if constexpr (BYTES > 128)
    asm volatile("s_load_dword %0, %3, 0x0\n\ts_load_dword %1, %3, 0x40\n\ts_load_dword %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2) : "s"(ka));
  else if constexpr (BYTES > 64)
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0), "=&s"(d1) : "s"(ka));
  else
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(d0) : "s"(ka));
  (void)d0; (void)d1; (void)d2; (void)d3;
#endif
}

// Ablation switches exist only in the tuner's -DHGEMM_ABLATION build of the library (lib_ablation/);
// in the shipping library they fold to `false` and the branches disappear.
#ifdef HGEMM_ABLATION
#define HGEMM_DBG(g, bit) (((g).debug & (bit)) != 0)
#else
#define HGEMM_DBG(g, bit) false
#endif

// Epilogue modes shared by the kernel families.
constexpr int EPI_C16    = 0;  // fp16 C, written directly (splits == 1)
constexpr int EPI_SLAB   = 1;  // fp32 partials to [splits][M][N] (or compact tail slabs); a second kernel combines
constexpr int EPI_FUSED  = 2;  // single-launch split-K: fp32 partials + arrival counter, the last arriver combines
constexpr int EPI_STREAMK = 3; // stream-K: one persistent launch over the tile-major K-stage sequence (StreamK above)
// Kernel-template flag on top of an epilogue id (families q and r): the "ktail" variant of the kernel, for a K that is not a
// multiple of the geometry's stage depth -- whole stages through the pipeline, the rest by direct_k_tail.  A variant of its own,
// so the kernels every K % stage == 0 launch runs keep their instruction streams.
constexpr int EPI_KTAIL = 8;
// Kernel-template flag (family q, round 5): the "kstagger" variant -- the workgroups of XCD x start every work item's K walk at
// stage x * nk / 8 and wrap (plan flag HGEMM_PLAN_XCD_STAGGER).  A variant of its own for the same reason as EPI_KTAIL.
constexpr int EPI_KSTAGGER = 16;

// Compile-time geometry of one kernel instantiation.
template <int BM_, int BN_, int WM_, int WN_, int MI_, int NBUF_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, MI = MI_, NBUF = NBUF_;
  static constexpr int NW          = WM * WN;            // waves per workgroup
  static constexpr int THREADS     = NW * 64;
  static constexpr int TM          = BM / WM;             // wave tile
  static constexpr int TN          = BN / WN;
  static constexpr int FM          = TM / MI;             // MFMA fragments per wave tile
  static constexpr int FN          = TN / MI;
  static constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
  static constexpr int LDS_BYTES   = STAGE_BYTES * NBUF;
  static constexpr int NI_A        = BM / 8;              // 1-KiB DMA pieces of the A tile
  static constexpr int NI          = (BM + BN) / 8;       // ... of the A+B stage
  static constexpr int NJ          = (NI + NW - 1) / NW;  // pieces per wave
  static_assert(MI == 16 || MI == 32, "MFMA shape");
  static_assert(BM % (WM * MI) == 0 && BN % (WN * MI) == 0, "wave tile must be MFMA-aligned");
  static_assert(BM % 8 == 0 && BN % 8 == 0, "tile rows come in 8-row DMA pieces");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(NBUF >= 2, "need at least double buffering");
  // counted vmcnt needs every wave to own the same number of DMA pieces per tile
  static_assert(NBUF == 2 || NI % NW == 0, "deep ring needs an even piece split");
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Block id -> (split, tile origin, K range): XCD-bijective remap + grouped rasterisation.
// The dispatcher places block b on XCD b % 8 (observed, used for speed only); the remap hands
// each XCD one contiguous run of logical tile ids, and the grouped raster makes that run a
// compact patch of the tile grid, so the A/B panels a patch shares are hit in that XCD's L2.
struct TileCoord {
  int split, m0, n0, k_begin, nk;
  int tile, item;   // output-tile id (raster order) and work-item id (split * tiles + tile)
  float* slab;   // fp32 partial destination of element (m0, n0) for split-K / tail items
  int slab_ld;   // its row stride in floats
};

// logical work-item id (after the XCD remap) -> (split, tile origin, K range), grouped raster
__device__ __forceinline__ TileCoord map_logical(const GemmArgs& g, int bid, int BM, int BN) {
#if HGEMM_FASTDIV
  const RasterPos rp = raster_fast(bid, g.tiles_m, g.tiles_n, g.group_m, g.tail_first, g.tail_tiles, g.rd);
  const int split = rp.split;
  TileCoord tc;
  tc.split = split;
  tc.tile = rp.tile;
  tc.item = bid;
  tc.m0 = rp.tile_m * BM;
  tc.n0 = rp.tile_n * BN;
#else   // (the same map as raster_ref, kept in this form: the shipped binaries were validated with it)
  const int tiles = g.tiles_m * g.tiles_n;
  int split, t_id;
  if (g.tail_tiles > 0) {
    split = bid / g.tail_tiles;
    t_id  = g.tail_first + (bid - split * g.tail_tiles);
  } else {
    split = bid / tiles;
    t_id  = bid - split * tiles;
  }
  const int gsz   = g.group_m * g.tiles_n;
  const int grp   = t_id / gsz;
  const int first_m = grp * g.group_m;
  const int gm    = min(g.tiles_m - first_m, g.group_m);
  const int tin   = t_id - grp * gsz;
  TileCoord tc;
  tc.split = split;
  tc.tile = t_id;
  tc.item = bid;
  tc.m0 = (first_m + tin % gm) * BM;
  tc.n0 = (tin / gm) * BN;
#endif
  tc.k_begin = split * g.k_chunk;
  tc.nk = (min(g.K, tc.k_begin + g.k_chunk) - tc.k_begin + BK - 1) / BK;   // the last K-step may be partial (classic family)
  if (HGEMM_DBG(g, 4)) tc.nk = min(tc.nk, 2);
  if (g.tail_tiles > 0 || g.counters != nullptr) {   // compact per-item slabs (tail pass, fused split-K)
    tc.slab = g.partial + (size_t)bid * ((size_t)BM * BN);
    tc.slab_ld = BN;
  } else {
    tc.slab = g.partial + ((size_t)split * g.M + tc.m0) * g.N + tc.n0;
    tc.slab_ld = g.N;
  }
  return tc;
}

__device__ __forceinline__ TileCoord map_block(const GemmArgs& g, int BM, int BN) {
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
    const int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  return map_logical(g, bid, BM, BN);
}

// Persistent grids (fewer workgroups than work items): the workgroups that run on XCD x
// (blockIdx % 8 == x) share the contiguous logical range XCD x would also get from map_block, and
// walk it round by round -- the r-th item of workgroup j of that XCD is base_x + r * nwg_x + j -- so
// the items an XCD processes concurrently stay a compact patch of the tile grid.
struct ItemWalk { int base, stride, first, count; };   // item r -> logical id base + first + r*stride

__device__ __forceinline__ ItemWalk persistent_walk(int total_items) {
  const int G = gridDim.x, bid = blockIdx.x;
  const int xcd = bid % NUM_XCD, j = bid / NUM_XCD;
  const int nwg_x = G / NUM_XCD + (xcd < G % NUM_XCD ? 1 : 0);
  const int q = total_items / NUM_XCD, r = total_items % NUM_XCD;
  const int items_x = q + (xcd < r ? 1 : 0);
  ItemWalk w;
  w.base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  w.stride = nwg_x;
  w.first = j;
  w.count = (j < items_x) ? (items_x - j + nwg_x - 1) / nwg_x : 0;
  return w;
}

// Epilogue shared by the kernel families (MI = 16 or 32 accumulator layout, operands swapped):
// lane holds C[m][n .. n+3] (4 consecutive N) per accumulator quad.
//   MI=16: m = (lane & 15),  n = (lane >> 4) * 4 + e                       (e = 0..3)
//   MI=32: m = (lane & 31),  n = 8 * q + 4 * (lane >> 5) + e,  reg = 4*q+e (q = 0..3)
// The host only takes the MFMA path when N % 4 == 0, ldc % 4 == 0 and C is 8-byte aligned, so a
// quad is either fully inside or fully outside the matrix.
// WIDE: -1 = pick the path at run time; 0 / 1 = only the narrow / wide path is compiled (the host chose).
// Families whose accumulators fill the register file need the single-path forms: with both paths
// present the compiler hoists their common fp32->fp16 conversions above the branch, which makes all
// FM*FN*4 values live at once.
//
// One fragment row (fixed i, all FN tiles) of the epilogue; store_tile below walks the rows.
template <int MI, int FN, int TM, int TN, bool SPLITK, int WIDE = -1, class ACC>
__device__ __forceinline__ void store_tile_row(const GemmArgs& g, const TileCoord& tc, int wave_m, int wave_n,
                                               int lane, int i, ACC (&row)[FN]) {
  // Wide path (fp16 output, 16x16 MFMA tiles, FN even): v_permlane16_swap exchanges the odd 16-lane
  // rows of tile j with the even rows of tile j+1, after which every lane owns 8 consecutive N
  // (16 bytes) of its C row: half the store instructions, 64 contiguous bytes per row per store.
  //   row q = lane >> 4 ends up with n = 16*(j + (q & 1)) + 8*(q >> 1) + 0..7
  if constexpr (!SPLITK && MI == 16 && (FN % 2 == 0) && WIDE != 0) {
    const bool wide = (WIDE == 1) ||
                      (((g.N & 7) == 0) && ((g.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0));
    if (wide) {
      const int q = lane >> 4;
      const int m = (HGEMM_DBG(g, 8) ? 0 : tc.m0) + wave_m * TM + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < FN; j += 2) {
        using h2 = __attribute__((ext_vector_type(2))) _Float16;
        const h2 a01 = {(f16)row[j][0], (f16)row[j][1]}, a23 = {(f16)row[j][2], (f16)row[j][3]};
        const h2 b01 = {(f16)row[j + 1][0], (f16)row[j + 1][1]}, b23 = {(f16)row[j + 1][2], (f16)row[j + 1][3]};
        const auto r0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a01), __builtin_bit_cast(unsigned, b01), false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a23), __builtin_bit_cast(unsigned, b23), false, false);
        const int n = (HGEMM_DBG(g, 8) ? 0 : tc.n0) + wave_n * TN + 16 * (j + (q & 1)) + 8 * (q >> 1);
        if (m < g.M && n < g.N) {
          using u4 = __attribute__((ext_vector_type(4))) unsigned;
          const u4 o = {r0[0], r1[0], r0[1], r1[1]};
          HGEMM_STORE_C(g, (u4*)(g.C + (size_t)m * g.ldc + n), o);
        }
      }
      return;
    }
  }
  const int lm = (MI == 16) ? (lane & 15) : (lane & 31);
  const int ln = (MI == 16) ? ((lane >> 4) * 4) : ((lane >> 5) * 4);
  constexpr int NQ = (MI == 16) ? 1 : 4;
  const int m = tc.m0 + wave_m * TM + i * MI + lm;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int n = tc.n0 + wave_n * TN + j * MI + q * 8 + ln;
      if (m < g.M && n < g.N) {
        if constexpr (SPLITK) {
          float* dst = tc.slab + (size_t)(m - tc.m0) * tc.slab_ld + (n - tc.n0);
          f32x4 o = {row[j][q * 4 + 0], row[j][q * 4 + 1], row[j][q * 4 + 2], row[j][q * 4 + 3]};
          *(f32x4*)dst = o;
        } else {
          f16* dst = g.C + (size_t)m * g.ldc + n;
          f16x4 o = {(f16)row[j][q * 4 + 0], (f16)row[j][q * 4 + 1], (f16)row[j][q * 4 + 2], (f16)row[j][q * 4 + 3]};
          HGEMM_STORE_C(g, (f16x4*)dst, o);
        }
      }
    }
  }
}

template <int MI, int FM, int FN, int TM, int TN, bool SPLITK, int WIDE = -1, class ACC>
__device__ __forceinline__ void store_tile(const GemmArgs& g, const TileCoord& tc, int wave_m,
                                           int wave_n, int lane, ACC (&acc)[FM][FN]) {
  if (HGEMM_DBG(g, 2)) return;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    // keep the accumulator -> VGPR traffic of one fragment row together: without the fence the
    // scheduler hoists all FM*FN*4 accumulator reads ahead of the first store
    __builtin_amdgcn_sched_barrier(0);
    store_tile_row<MI, FN, TM, TN, SPLITK, WIDE>(g, tc, wave_m, wave_n, lane, i, acc[i]);
  }
}

// ---- single-launch split-K ("fused"; replaces the reference's atomicAdd split-K, a100_F32F16F16F32/
// 64_256_16384.cu:149-152, and its per-call memset + convert kernels) ---------------------------------------
// Every (split, tile) work item writes its fp32 partial to its own compact slab in the kernel's LANE ORDER
// (quad x of thread tid at slab[(x * THREADS + tid) * 4]: every store instruction is one contiguous 1 KiB
// per wave) and draws a ticket from the tile's arrival counter.  The workgroup that draws the last ticket
// adds the `splits` slabs IN SPLIT ORDER (so the result does not depend on arrival order: deterministic,
// unlike atomics) and writes the fp16 tile.  Nobody waits for anybody, so the scheme needs no co-residency
// and cannot deadlock; placement only affects speed.  The last arriver resets the counter, so the counters
// are zero between launches (the host zeroes them once at allocation).
// Visibility (cdna_hip_programming.md section 6, guideline 16, write-through form): the slabs are written
// with sc1 (write-through) stores and read with sc1 loads, so they bypass the non-coherent per-XCD L2s; each
// wave drains its stores (vmcnt(0)), a workgroup barrier collects the waves, then ONE relaxed agent-scope
// fetch_add publishes the arrival.  No release / acquire fences: an agent-scope release is a whole-L2
// write-back (buffer_wbl2) per workgroup -- measured +7..12 us per GEMM with the fence form.
#if defined(__HIP_DEVICE_COMPILE__)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
constexpr int kAuxSc1 = 16;   // buffer aux operand: bit 4 = sc1 (system-coherent / write-through)

// slab byte offset of quad x of thread tid in work item `item` (32-bit: the host only takes the fused form
// when all slabs together stay below 2 GiB)
template <int THREADS>
__device__ __forceinline__ uint32_t fused_off(int item, int slab_elems, int x, int tid) {
  return ((uint32_t)item * (uint32_t)slab_elems + ((uint32_t)x * THREADS + tid) * 4u) * 4u;
}
__device__ __forceinline__ void fused_store(__amdgpu_buffer_rsrc_t rs, uint32_t off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, kAuxSc1);
}
__device__ __forceinline__ f32x4 fused_load(__amdgpu_buffer_rsrc_t rs, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, kAuxSc1));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fused_rsrc(const GemmArgs& g) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)g.partial, 0, 0xFFFFFFFFu, 0x00020000);
}

// Called by every thread of the workgroup after its slab stores were issued.  `lds_flag` is a word of the
// kernel's (single) LDS array that no in-flight LDS-DMA targets.  True in every thread of the last arriver.
__device__ __forceinline__ bool fused_publish_and_vote(const GemmArgs& g, int tile, volatile unsigned* lds_flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my write-through slab stores are acknowledged
  __syncthreads();                                    // ... and so are every other wave's
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(g.counters + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (old + 1u == (unsigned)g.splits);
    if (last) __hip_atomic_store(g.counters + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *lds_flag != 0u;
}

// The last arriver's combine: out[x] = slab(0)[x] + slab(1)[x] + ... + slab(splits - 1)[x], added IN THAT ORDER (so the bits do not
// depend on who arrives last), for the NQ quads quad_of(0 .. NQ-1) of this thread.
// Round 6: up to UMAX slabs' loads are in flight behind ONE wait.  The loop this replaces loaded a split's quads,
// waited for them, added them, and only then asked for the next split's -- `splits` dependent round trips to the memory side of the
// fabric (the slabs are sc1 / write-through data: every load misses the XCD's L2 by design), 16 of them for a 16-way split of a
// 64 x 64 output.  That serial walk, not the arrival protocol, is why the single-launch form lost to the two-pass form (a second
// dispatch, ~2.5 us back to back) on all but the smallest splits; the reference's one-launch split-K meets in L2 atomics
// (kernels/a100_F32F16F16F32/64_256_16384.cu:24-31,149-152) and pays no such walk.  With the loads batched the combine is
// ceil(splits / UMAX) round trips.  Slab 0 initialises (no "+ 0.0": a sum of -0.0 partials stays -0.0 as before).
template <int THREADS, int NQ, int UMAX, class QuadOf>
__device__ __forceinline__ void fused_combine(__amdgpu_buffer_rsrc_t rsP, int splits, int tiles, int tile, int slab_elems, int tid,
                                              QuadOf quad_of, f32x4 (&out)[NQ]) {
  static_assert(UMAX == 1 || UMAX == 2 || UMAX == 4 || UMAX == 8 || UMAX == 16 || UMAX == 32, "batch depth");
  int s = 0;
  auto batch = [&](auto u_tag, auto first_tag) {
    constexpr int U = decltype(u_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    f32x4 v[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) v[u][q] = fused_load(rsP, fused_off<THREADS>((s + u) * tiles + tile, slab_elems, quad_of(q), tid));
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) out[q] = (FIRST && u == 0) ? v[u][q] : out[q] + v[u][q];
    s += U;
  };
  using std::integral_constant;
  using T = std::true_type;
  using F = std::false_type;
  // first batch: the deepest that fits (splits >= 1 always)
  if (UMAX >= 32 && splits >= 32) batch(integral_constant<int, (UMAX >= 32 ? 32 : 1)>{}, T{});
  else if (UMAX >= 16 && splits >= 16) batch(integral_constant<int, (UMAX >= 16 ? 16 : 1)>{}, T{});
  else if (UMAX >= 8 && splits >= 8) batch(integral_constant<int, (UMAX >= 8 ? 8 : 1)>{}, T{});
  else if (UMAX >= 4 && splits >= 4) batch(integral_constant<int, (UMAX >= 4 ? 4 : 1)>{}, T{});
  else if (UMAX >= 2 && splits >= 2) batch(integral_constant<int, (UMAX >= 2 ? 2 : 1)>{}, T{});
  else batch(integral_constant<int, 1>{}, T{});
  if constexpr (UMAX >= 32) { while (s + 32 <= splits) batch(integral_constant<int, 32>{}, F{}); }
  if constexpr (UMAX >= 16) { while (s + 16 <= splits) batch(integral_constant<int, 16>{}, F{}); }
  if constexpr (UMAX >= 8) { while (s + 8 <= splits) batch(integral_constant<int, 8>{}, F{}); }
  if constexpr (UMAX >= 4) { while (s + 4 <= splits) batch(integral_constant<int, 4>{}, F{}); }
  if constexpr (UMAX >= 2) { while (s + 2 <= splits) batch(integral_constant<int, 2>{}, F{}); }
  while (s < splits) batch(integral_constant<int, 1>{}, F{});
}
// batch depth for a thread that owns NQ quads: at most 32 sixteen-byte loads (128 registers) in flight
template <int NQ>
struct FusedBatch { static constexpr int U = NQ <= 1 ? 32 : NQ <= 2 ? 16 : NQ <= 4 ? 8 : NQ <= 8 ? 4 : NQ <= 16 ? 2 : 1; };
#endif  // __HIP_DEVICE_COMPILE__

// The buffer-resource builtins only exist in the device pass; the host pass just needs the
// kernel stubs, so device bodies are compiled under __HIP_DEVICE_COMPILE__ only.
#if defined(__HIP_DEVICE_COMPILE__)
// Issue the LDS-DMA pieces of one K-step (tile rows x 128 B) owned by this wave.
template <class CFG>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                           const uint32_t (&voff)[CFG::NJ], char* lds_stage,
                                           int wave, uint32_t kbyte) {
#pragma unroll
  for (int j = 0; j < CFG::NJ; ++j) {
    const int i = wave + j * CFG::NW;  // wave-uniform piece index
    if (CFG::NI % CFG::NW == 0 || i < CFG::NI) {
      lds_void_t* dst = (lds_void_t*)(lds_stage + i * 1024);
      if (i < CFG::NI_A)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, voff[j], kbyte, 0, HGEMM_DMA_AUX);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, voff[j], kbyte, 0, HGEMM_DMA_AUX);
    }
  }
}

// The last, PARTIAL K-step of a work item (K % 64 != 0, K % 8 == 0): lanes whose 16-byte source chunk lies at
// k >= K get bit 31 set in their offset, which is beyond the descriptors' 2 GiB range: the load returns zeros, so
// the tail of every LDS row is zero-filled by the DMA itself (the reference pads K in the harness instead,
// tools/utils.py:8-36).  Bit j of tailmask = piece j of this lane is past K.
template <class CFG>
__device__ __forceinline__ void stage_tile_tail(__amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB,
                                                const uint32_t (&voff)[CFG::NJ], uint32_t tailmask, char* lds_stage,
                                                int wave, uint32_t kbyte) {
#pragma unroll
  for (int j = 0; j < CFG::NJ; ++j) {
    const int i = wave + j * CFG::NW;
    if (CFG::NI % CFG::NW == 0 || i < CFG::NI) {
      lds_void_t* dst = (lds_void_t*)(lds_stage + i * 1024);
      const uint32_t vo = voff[j] | (((tailmask >> j) & 1u) << 31);
      if (i < CFG::NI_A)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, dst, 16, vo, kbyte, 0, HGEMM_DMA_AUX);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, dst, 16, vo, kbyte, 0, HGEMM_DMA_AUX);
    }
  }
}


// K tail by DIRECT fragment loads (families q and r, their "ktail" kernel variants; the classic family pads its last LDS-DMA
// step instead, stage_tile_tail): the last k_end - k0 elements (> 0, a multiple of 8, less than one pipeline stage) of a work
// item's K range never enter LDS.  Every wave loads the MFMA fragments of its own wave tile straight from global memory, the
// way family w does for whole problems (hgemm_kernel_wd.hpp): lane l holds the 8 consecutive K of row l & 15 at
// k = 8 (l >> 4) of a K = 32 slice, 16 rows x 64 contiguous bytes per wave instruction.  The reference pads K in the
// harness instead (tools/utils.py:8-36).
// The loads are buffer loads through descriptors that start at the tile's first row and end with the matrix (at most 2 GiB,
// and every real offset below that: host check): rows past the M / N edge are out of range and read as zeros (their
// products are never stored), and a lane whose 8 halfs lie at k >= k_end gets bit 31 into its offset -- out of range as
// well, so the last, partial slice is zero-filled per lane (the mark of stage_tile_tail).  U slices, U (FM + FN) loads,
// are in flight behind one wait.  mfma(i, j, b, a) accumulates the fragment pair of A row block i and B row block j.
// Buffer descriptor of an operand's rows from `base` on: `bytes` long, at most `cap`.  Base and range are made PROVABLY uniform
// (readfirstlane), or hipcc wraps every buffer instruction that uses the descriptor in a waterfall loop.
template <unsigned long long CAP>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const f16* base, size_t bytes) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(base);
  const uintptr_t uni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
  const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bytes > CAP ? CAP : bytes));
  return __builtin_amdgcn_make_buffer_rsrc((void*)uni, 0, nrec, 0x00020000);
}

template <int FM, int FN, class F>
__device__ __forceinline__ void direct_k_tail(const GemmArgs& g, int m0, int n0, int row_a, int row_b, int k0, int k_end, int lane,
                                              F&& mfma) {
  const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc<0x80000000ull>(g.A + (size_t)m0 * g.lda, ((size_t)(g.M - m0) * g.lda) * 2);
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc<0x80000000ull>(g.Bt + (size_t)n0 * g.ldb, ((size_t)(g.N - n0) * g.ldb) * 2);
  const int l15 = lane & 15, lq = lane >> 4;
  uint32_t va[FM], vb[FN];   // byte offset of this lane's row of every 16-row block of the wave tile
#pragma unroll
  for (int i = 0; i < FM; ++i) va[i] = (uint32_t)(row_a + i * 16 + l15) * (uint32_t)g.lda * 2u;
#pragma unroll
  for (int j = 0; j < FN; ++j) vb[j] = (uint32_t)(row_b + j * 16 + l15) * (uint32_t)g.ldb * 2u;
  const int nslices = (k_end - k0 + 31) / 32;
  int s = 0;
  auto trip = [&](auto u_tag) __attribute__((always_inline)) {
    constexpr int U = decltype(u_tag)::value;
    f16x8 af[U][FM], bf[U][FN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (s + u) * 32 + lq * 8;
      const uint32_t kb = k < k_end ? (uint32_t)k * 2u : 0x80000000u;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[u][i] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA, va[i] + kb, 0, 0));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[u][j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB, vb[j] + kb, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mfma(i, j, bf[u][j], af[u][i]);
    s += U;
  };
  constexpr int UMAX = (FM + FN) <= 4 ? 4 : (FM + FN) <= 8 ? 2 : 1;   // at most 16 fragments (64 registers) per trip
  if constexpr (UMAX > 1) {
#pragma clang loop unroll(disable)
    while (s + UMAX <= nslices) trip(std::integral_constant<int, UMAX>{});
  }
#pragma clang loop unroll(disable)
  while (s < nslices) trip(std::integral_constant<int, 1>{});
}

#endif  // __HIP_DEVICE_COMPILE__

// The classic family's main loop for ONE work item: tile (m0, n0), K range [k_begin, k_begin + k_items) in nk pipeline
// stages (the last one partial when k_items % 64 != 0), accumulators cleared here.  Shared by hgemm_tn_kernel (one item per
// workgroup) and hgemm_tn_sk_kernel (a run of stream-K segments per workgroup).
#if defined(__HIP_DEVICE_COMPILE__)
template <class CFG, class ACC>
__device__ __forceinline__ void classic_mainloop(const GemmArgs& g, int m0, int n0, int k_begin, int k_items, int nk, char* smem, int lane,
                                                 int wave, int wave_m, int wave_n, ACC (&acc)[CFG::FM][CFG::FN]) {
  constexpr int BM = CFG::BM, MI = CFG::MI, NBUF = CFG::NBUF;
  constexpr int FM = CFG::FM, FN = CFG::FN, NW = CFG::NW, NJ = CFG::NJ;
  // ---- LDS-DMA source addressing ------------------------------------------------------------
  // One descriptor per operand, based at the tile's first row; rows past the matrix edge are
  // clamped to the last valid row (their products are never stored).
  const f16* a_base = g.A + (size_t)m0 * g.lda;
  const f16* b_base = g.Bt + (size_t)n0 * g.ldb;
  // (range 2 GiB: every real offset is below it -- host check -- and bit 31 marks a lane as out of range)
  __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc((void*)a_base, 0, 0x80000000u, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc((void*)b_base, 0, 0x80000000u, 0x00020000);
  // valid 16-byte chunks of the last K-step of this work item (8 = it is a full step)
  const int tail_chunks = (k_items % BK) ? (k_items % BK) / 8 : 8;
  uint32_t tailmask = 0;

  uint32_t voff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i    = wave + j * NW;
    const bool isA = i < CFG::NI_A;
    const int il   = isA ? i : i - CFG::NI_A;      // piece index inside its operand tile
    const int r    = il * 8 + (lane >> 3);         // tile row written by this lane
    const int rmax = isA ? (g.M - 1 - m0) : (g.N - 1 - n0);
    const int rc   = min(r, rmax);
    const int ld   = isA ? g.lda : g.ldb;
    // LDS slot (lane & 7) of row r holds source chunk slot ^ ((r >> 1) & 7); r & 15 = (il&1)*8 + lane>>3
    const int chunk = (lane & 7) ^ (((il & 1) << 2) | (lane >> 4));
    voff[j] = ((uint32_t)rc * (uint32_t)ld + (uint32_t)chunk * 8u) * 2u;
    tailmask |= (chunk >= tail_chunks ? 1u : 0u) << j;
  }

  // ---- fragment read offsets (bytes inside a stage) -----------------------------------------
  constexpr int KS = (MI == 16) ? 2 : 4;  // MFMA k-slices per stage (K=32 / K=16 each)
  int frag_off[KS];
  {
    const int lr = (MI == 16) ? (lane & 15) : (lane & 31);
    const int lq = (MI == 16) ? (lane >> 4) : (lane >> 5);
    const int sw = (lr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = (MI == 16) ? (ks * 4 + lq) : (ks * 2 + lq);
      frag_off[ks] = lr * ROW_BYTES + ((c ^ sw) << 4);
    }
  }
  const int a_row_base = wave_m * CFG::TM * ROW_BYTES;
  const int b_row_base = BM * ROW_BYTES + wave_n * CFG::TN * ROW_BYTES;

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < (MI == 16 ? 4 : 16); ++e) acc[i][j][e] = 0.0f;

  // ---- pipeline --------------------------------------------------------------------------------
  uint32_t kbyte = (uint32_t)k_begin * 2u;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) {
    if (s < nk) {
      if (s == nk - 1 && tail_chunks < 8) stage_tile_tail<CFG>(rsA, rsB, voff, tailmask, smem + s * CFG::STAGE_BYTES, wave, kbyte);
      else stage_tile<CFG>(rsA, rsB, voff, smem + s * CFG::STAGE_BYTES, wave, kbyte);
      kbyte += ROW_BYTES;
    }
  }

  int rd = 0;             // stage being consumed
  int wr = NBUF - 1;      // stage being refilled
  for (int t = 0; t < nk; ++t) {
    // Tile t must have landed: allow the NBUF-2 younger tiles to stay in flight.
    if (t + NBUF - 2 < nk)
      wait_vmcnt<NJ*(NBUF - 2)>();
    else
      wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // all waves' pieces of tile t landed; stage `wr` is free again

    if (t + NBUF - 1 < nk && !HGEMM_DBG(g, 1)) {
      if (t + NBUF == nk && tail_chunks < 8) stage_tile_tail<CFG>(rsA, rsB, voff, tailmask, smem + wr * CFG::STAGE_BYTES, wave, kbyte);
      else stage_tile<CFG>(rsA, rsB, voff, smem + wr * CFG::STAGE_BYTES, wave, kbyte);
      kbyte += ROW_BYTES;
    }

    const char* st = smem + rd * CFG::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f16x8 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *(const f16x8*)(st + a_row_base + i * MI * ROW_BYTES + frag_off[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bf[j] = *(const f16x8*)(st + b_row_base + j * MI * ROW_BYTES + frag_off[ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if constexpr (MI == 16)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
          else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
    rd = (rd + 1 == NBUF) ? 0 : rd + 1;
    wr = (wr + 1 == NBUF) ? 0 : wr + 1;
  }
}
#endif  // __HIP_DEVICE_COMPILE__

template <class CFG, int EPI>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI;
  constexpr int FM = CFG::FM, FN = CFG::FN;

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;

  const TileCoord tc = map_block(g, BM, BN);

  using acc_t = typename std::conditional<MI == 16, f32x4, f32x16>::type;
  acc_t acc[FM][FN];
  classic_mainloop<CFG>(g, tc.m0, tc.n0, tc.k_begin, min(g.K, tc.k_begin + g.k_chunk) - tc.k_begin, tc.nk, smem, lane, wave, wave_m, wave_n, acc);

  // ---- epilogue ----------------------------------------------------------------------------------
  if constexpr (EPI == EPI_FUSED) {
    constexpr int NQ = (MI == 16) ? 1 : 4;   // f32x4 quads per accumulator tile
    constexpr int SLAB = BM * BN;
    const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
          fused_store(rsP, fused_off<CFG::THREADS>(tc.item, SLAB, (i * FN + j) * NQ + q, tid), v);
        }
    if (!fused_publish_and_vote(g, tc.tile, (volatile unsigned*)smem, tid)) return;
    // last arriver: slabs of this tile are item = s * tiles + tile, s = 0 .. splits-1, added in that order
    {
      constexpr int NQT = FM * FN * NQ;
      f32x4 sum[NQT];
      fused_combine<CFG::THREADS, NQT, FusedBatch<NQT>::U>(rsP, g.splits, g.tiles_m * g.tiles_n, tc.tile, SLAB, tid, [](int x) { return x; }, sum);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = sum[(i * FN + j) * NQ + q][e];
    }
    store_tile<MI, FM, FN, CFG::TM, CFG::TN, false>(g, tc, wave_m, wave_n, lane, acc);
  } else {
    store_tile<MI, FM, FN, CFG::TM, CFG::TN, EPI == EPI_SLAB>(g, tc, wave_m, wave_n, lane, acc);
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// ---- stream-K for the families whose accumulators are C++ values (classic "t", register-staged "r") ------------------
#if defined(__HIP_DEVICE_COMPILE__)
// One segment of a workgroup's run.
struct SkSegment {
  int tile, k0, k1;   // output tile (raster order), stage range inside it
  int m0, n0;
  int slab;           // compact slab of a partial segment (2 * w + 0 / 1)
  bool whole;         // covers the whole tile: stored directly
};
// logical workgroup id: the XCD-bijective remap of map_block (an XCD's workgroups own a contiguous part of the sequence,
// so the tiles an XCD works on concurrently are a compact patch of the tile grid and share A / B panels in its L2)
__device__ __forceinline__ int sk_logical_wg() {
  const int bid = blockIdx.x, nwg = gridDim.x;
  const int xcd = bid % NUM_XCD, idx = bid / NUM_XCD;
  const int q = nwg / NUM_XCD, r = nwg % NUM_XCD;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// segment that starts at stage x of workgroup w's run [run_begin, run_end)
__device__ __forceinline__ SkSegment sk_segment(const GemmArgs& g, int w, int run_begin, int run_end, int x, int BM, int BN) {
  SkSegment s;
  s.tile = (int)fast_div((uint32_t)x, g.sk.div_steps);
  s.k0 = x - s.tile * g.sk.steps;
  s.k1 = min(g.sk.steps, s.k0 + (run_end - x));
  s.whole = s.k0 == 0 && s.k1 == g.sk.steps;
  s.slab = 2 * w + (x == run_begin ? 0 : 1);
#if HGEMM_FASTDIV
  const RasterPos rp = raster_fast(s.tile, g.tiles_m, g.tiles_n, g.group_m, 0, 0, g.rd);
#else
  const RasterPos rp = raster_ref(s.tile, g.tiles_m, g.tiles_n, g.group_m, 0, 0);
#endif
  s.m0 = rp.tile_m * BM;
  s.n0 = rp.tile_n * BN;
  return s;
}
// Arrival of a partial segment (its slab stores were issued by every thread): true in every thread of the workgroup that
// completes the tile's stage count.  Same visibility protocol as fused_publish_and_vote (write-through slabs, vmcnt(0),
// barrier, one relaxed agent-scope fetch_add); `lds_flag` is a word no in-flight LDS traffic targets.
__device__ __forceinline__ bool sk_publish_and_vote(const GemmArgs& g, const SkSegment& s, volatile unsigned* lds_flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned mine = (unsigned)(s.k1 - s.k0);
    const unsigned old = __hip_atomic_fetch_add(g.counters + s.tile, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (old + mine == (unsigned)g.sk.steps);
    if (last) __hip_atomic_store(g.counters + s.tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *lds_flag != 0u;
}

// Epilogue of one segment: a whole tile is stored directly; a partial one goes to its slab, and the workgroup that completes
// the tile adds the tile's slabs in K order and stores it.
template <class CFG, class ACC>
__device__ __forceinline__ void sk_epilogue(const GemmArgs& g, const SkSegment& s, ACC (&acc)[CFG::FM][CFG::FN], volatile unsigned* lds_flag,
                                            int tid, int lane, int wave_m, int wave_n) {
  constexpr int MI = CFG::MI, FM = CFG::FM, FN = CFG::FN;
  constexpr int NQ = (MI == 16) ? 1 : 4, SLAB = CFG::BM * CFG::BN;
  TileCoord tc;
  tc.split = 0; tc.m0 = s.m0; tc.n0 = s.n0; tc.k_begin = 0; tc.nk = 0; tc.tile = s.tile; tc.item = s.tile; tc.slab = nullptr; tc.slab_ld = 0;
  if (s.whole) {
    store_tile<MI, FM, FN, CFG::TM, CFG::TN, false>(g, tc, wave_m, wave_n, lane, acc);
    return;
  }
  const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const f32x4 v = {acc[i][j][q * 4 + 0], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3]};
        fused_store(rsP, fused_off<CFG::THREADS>(s.slab, SLAB, (i * FN + j) * NQ + q, tid), v);
      }
  if (!sk_publish_and_vote(g, s, lds_flag, tid)) return;
  // the tile's parts are the runs of consecutive workgroups, in K order
  const int G = gridDim.x, t0 = s.tile * g.sk.steps, t1 = t0 + g.sk.steps;
  bool first = true;
  for (int w = sk_owner(g.sk, t0, G); w < G; ++w) {
    const int b = sk_start(g.sk, w, G), e = sk_start(g.sk, w + 1, G);
    if (b >= t1) break;
    if (min(e, t1) <= max(b, t0)) continue;             // (an empty run)
    const int slab = 2 * w + (b >= t0 ? 0 : 1);         // the part opens w's run, or closes it
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const f32x4 v = fused_load(rsP, fused_off<CFG::THREADS>(slab, SLAB, (i * FN + j) * NQ + q, tid));
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) acc[i][j][q * 4 + e2] = first ? v[e2] : acc[i][j][q * 4 + e2] + v[e2];
        }
    first = false;
  }
  store_tile<MI, FM, FN, CFG::TM, CFG::TN, false>(g, tc, wave_m, wave_n, lane, acc);
}
#endif  // __HIP_DEVICE_COMPILE__

// Stream-K on the classic family: one persistent launch of G workgroups, each walking its run of the sequence.
template <class CFG>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_sk_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI;
  constexpr int FM = CFG::FM, FN = CFG::FN;

  // the ring + the vote word behind it (ONE LDS object, see hgemm_kernel_sp.hpp)
  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES + 64];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;

  const int G = gridDim.x, w = sk_logical_wg();
  const int run_begin = sk_start(g.sk, w, G), run_end = sk_start(g.sk, w + 1, G);
  using acc_t = typename std::conditional<MI == 16, f32x4, f32x16>::type;
#pragma clang loop unroll(disable)
  for (int x = run_begin; x < run_end;) {
    const SkSegment s = sk_segment(g, w, run_begin, run_end, x, BM, BN);
    // every wave is done with the previous segment's last stage (and with the vote word) before the ring is refilled
    if (x != run_begin) __syncthreads();
    const int k_begin = s.k0 * BK, k_items = min(g.K, s.k1 * BK) - k_begin;
    acc_t acc[FM][FN];
    classic_mainloop<CFG>(g, s.m0, s.n0, k_begin, k_items, s.k1 - s.k0, smem, lane, wave, wave_m, wave_n, acc);
    sk_epilogue<CFG>(g, s, acc, (volatile unsigned*)(smem + CFG::LDS_BYTES), tid, lane, wave_m, wave_n);
    x += s.k1 - s.k0;
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
