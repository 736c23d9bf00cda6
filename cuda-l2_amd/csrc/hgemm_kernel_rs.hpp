// Family "r": register-staged STREAMING kernel for the HBM-bound skinny shapes (min(M, N) <= 256, K >= 4096).
//
// Why (round-2 measurements): on 16384 x 64 x 16384 the LDS-DMA families stream A at 4.5-4.8 TB/s,
// hipBLASLt's MT64x64x256 kernel at 5.9 TB/s.  An LDS-DMA instruction moves 8 tile rows x 128 B: every request
// of a K-step opens another DRAM page, and a 4-deep ring of BK=64 tiles keeps only 24 KiB of A per workgroup in
// flight.  This family gives up the DMA for the access shape that matters to HBM:
//   * BKS = 256 (or 128) halfs of K per stage: one global_load_dwordx4 of a wave covers 2 (4) tile rows x 512 (256)
//     contiguous bytes;
//   * TWO whole stages are prefetched in VGPRs (stage t+1 and t+2: 2 x (BM + BN) x BKS x 2 B per workgroup in
//     flight -- 128 KiB for 64 x 64 x 256) while the MFMAs run on stage t out of a SINGLE LDS buffer;
//     per stage: MFMAs(t) | barrier | ds_write stage t+1 | issue the loads of stage t+3 into the freed registers |
//     barrier.  The kernel is HBM-bound by a factor of ~4 against its LDS and MFMA work, so nothing else is
//     overlapped; the loads are ordinary VGPR loads, so hipcc's own counted vmcnt keeps stage t+2 in flight
//     across the wait for stage t+1;
//   * LDS image [BM + BN rows][BKS x 2 B], 16-byte chunk c of row r at slot c ^ (r & 15): conflict-free for the
//     8-lane groups of ds_write_b128 (8 consecutive chunks of one row) and the 16-lane groups of the fragment
//     ds_read_b128 (16 rows, one chunk column); the XOR of the K slice folds into one v_xor per read;
//   * MFMA loop, operand swap and epilogues (direct / split-K slabs / single-launch split-K) are hgemm_tn_kernel's.
// Geometry: 4 waves (2 x 2), 16x16x32 MFMA.  Every split-K chunk is a multiple of BKS; so is K, or (round 4) the kernel's "ktail"
// variant walks the whole stages and accumulates the remaining K % BKS (a multiple of 8) from fragments loaded straight from
// global memory (hgemm_kernel.hpp: direct_k_tail).
#pragma once

#include "hgemm_kernel.hpp"

namespace hgemm_mi355x {

// LB = LDS buffers: 1 (round 2) or 2 ("_d" members, round 4).  With one buffer a stage is two phases between barriers
// (MFMAs on stage t | ds_write stage t+1, issue the loads of stage t+3): nothing of the CU's LDS-write, load-issue and MFMA time
// overlaps, and on the N = 128 / M = 128 streaming shapes the sum of the three is what a stage takes (16384 x 128 x 16384 at
// 3.9 TB/s where the N = 64 shapes, with half the MFMA and fragment work per streamed byte, reach 5.6-6.3).  With two buffers
// stage t+1 is written to the OTHER buffer while stage t is computed: one barrier per stage, and the ds_writes, the global
// loads of stage t+3 and the MFMAs of stage t are one basic block the hardware overlaps.
template <int BM_, int BN_, int BKS_, int LB_ = 1>
struct CfgRS : Cfg<BM_, BN_, 2, 2, 16, 2> {
  using Base = Cfg<BM_, BN_, 2, 2, 16, 2>;
  static constexpr int LB  = LB_;
  static constexpr int BKS = BKS_;                        // K halfs per stage
  static constexpr int RB  = BKS * 2;                     // LDS row bytes
  static constexpr int NCH = RB / 16;                     // 16-byte chunks per row
  static constexpr int KS  = BKS / 32;                    // MFMA K=32 slices per stage
  static constexpr int CA  = BM_ * NCH / Base::THREADS;   // chunks per thread per stage, A / B
  static constexpr int CB  = BN_ * NCH / Base::THREADS;
  static constexpr int STAGE_BYTES = (BM_ + BN_) * RB;
  static constexpr int LDS_BYTES = LB_ * STAGE_BYTES;
  static constexpr int WGS_PER_CU = (BM_ * BN_ <= 64 * 128 && LDS_BYTES <= 64 * 1024) ? 2 : 1;   // two wherever tile and LDS allow it
  static_assert(LB_ == 1 || LB_ == 2, "one or two LDS buffers");
  static_assert(BKS == 128 || BKS == 256, "stage depth");
  static_assert((BM_ * NCH) % Base::THREADS == 0 && (BN_ * NCH) % Base::THREADS == 0, "whole chunks per thread");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert((CA + CB) * 4 * 2 + Base::FM * Base::FN * 4 <= 232, "two stages in registers + accumulators");
};

// The epilogue of hgemm_tn_kernel for accumulators held as C++ values (shared with family r).
template <class CFG, int EPI, class ACC>
__device__ __forceinline__ void classic_epilogue(const GemmArgs& g, const TileCoord& tc, ACC (&acc)[CFG::FM][CFG::FN], char* smem,
                                                 int tid, int lane, int wave_m, int wave_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, MI = CFG::MI, FM = CFG::FM, FN = CFG::FN;
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
#endif
}

// cache policy of the operand loads (experiment knob HGEMM_RS_NT: 0 = default, 1 = the STREAMED operand -- the one
// with more rows, read exactly once -- is loaded non-temporally, 2 = both)
#ifndef HGEMM_RS_NT
#define HGEMM_RS_NT 0
#endif
// K stagger (Tensile's StaggerU): workgroup w starts its K walk at stage (w * HGEMM_RS_STAGGER) mod nk and wraps
// around, so the workgroups of a launch do not all read the same K offset of their rows at the same time (row
// stride = K * 2 B is a large power of two on the grid shapes: identical low address bits = the same few HBM
// channels; measured: 16384 x 64 x 16384 115 -> 96 us).  The summation order of a tile changes with its
// coordinates, deterministically.
#ifndef HGEMM_RS_STAGGER
#define HGEMM_RS_STAGGER 3
#endif
#ifndef HGEMM_RS_STAGGER_MODE
#define HGEMM_RS_STAGGER_MODE 0   // 0: per work item; 1: per XCD (blockIdx % 8): an XCD's workgroups stay in lock-step; 2: item % 8
#endif

// Family r's main loop for ONE work item: tile (m0, n0), nk stages of BKS halfs from K offset k_begin, accumulators cleared
// here; `tile_id` seeds the K stagger (built from the item's COORDINATES, not from its raster position: the summation order
// of an output tile must not depend on the raster group the caller or the tuner picked).  Shared by hgemm_tn_rs_kernel (one
// item per workgroup) and hgemm_tn_rs_sk_kernel (a run of stream-K segments per workgroup).
#if defined(__HIP_DEVICE_COMPILE__)
template <class CFG>
__device__ __forceinline__ void rs_mainloop(const GemmArgs& g, int m0, int n0, int k_begin, int nk, unsigned tile_id, char* smem, int tid,
                                            int lane, int wave_m, int wave_n, f32x4 (&acc)[CFG::FM][CFG::FN]) {
  constexpr int BM = CFG::BM, FM = CFG::FM, FN = CFG::FN, THREADS = CFG::THREADS;
  constexpr int RB = CFG::RB, NCH = CFG::NCH, KS = CFG::KS, CA = CFG::CA, CB = CFG::CB, BKS = CFG::BKS;
  // ---- addressing of this thread's chunks ------------------------------------------------------------------------
  // chunk id = tid + p * THREADS: row id / NCH, chunk id % NCH -> a wave covers 64 / NCH rows x RB contiguous bytes.
  // THREADS is a multiple of NCH, so every chunk p of a thread has the same column c and row r0 + p * RP: ONE
  // per-lane byte offset per operand; the row step p * RP * ld * 2 is a wave-uniform add on top of it and the K
  // position rides in the scalar offset of the buffer load.  The descriptors start at the tile's first row and end
  // with the matrix, and the whole ROW part of the address is in the vector offset (the part the hardware range
  // check always covers): rows past the edge are out of range and read as zeros (their products are never stored).
  constexpr int RP = THREADS / NCH;                       // tile rows between two chunks of a thread (8 or 16)
  const int r0 = tid / NCH, c0 = tid % NCH;
  const __amdgpu_buffer_rsrc_t rsA = uniform_rsrc<0xFFFFFFFFull>(g.A + (size_t)m0 * g.lda, ((size_t)(g.M - m0) * g.lda) * 2);
  const __amdgpu_buffer_rsrc_t rsB = uniform_rsrc<0xFFFFFFFFull>(g.Bt + (size_t)n0 * g.ldb, ((size_t)(g.N - n0) * g.ldb) * 2);
  const uint32_t voff_a = ((uint32_t)r0 * (uint32_t)g.lda + (uint32_t)c0 * 8u) * 2u;
  const uint32_t voff_b = ((uint32_t)r0 * (uint32_t)g.ldb + (uint32_t)c0 * 8u) * 2u;
  const uint32_t step_a = (uint32_t)RP * (uint32_t)g.lda * 2u, step_b = (uint32_t)RP * (uint32_t)g.ldb * 2u;
  const uint32_t kbyte0 = (uint32_t)k_begin * 2u;
  // LDS slot of chunk p: row r0 + p*RP, slot c0 ^ (row & 15).  RP = 16: the XOR term is the same for every p;
  // RP = 8: it flips bit 3 of the slot (= byte bit 7) for odd p.  So: one base, one XOR constant, immediates.
  const int lds_base = r0 * RB + ((c0 ^ (r0 & 15)) << 4);
  auto lds_of = [&](int p) { return (lds_base ^ ((RP == 8 && (p & 1)) ? 128 : 0)) + p * RP * RB; };

  // fragment addressing: row i*16 + l15, chunk (4 ks + lq) ^ l15  ->  lane constant ^ (ks << 6)
  const int l15 = lane & 15, lq = lane >> 4;
  const int frag_lane = l15 * RB + ((lq ^ l15) << 4);      // ks = 0: chunk lq ^ l15 (l15 < 16 <= NCH)
  const int a_row_base = wave_m * CFG::TM * RB;
  const int b_row_base = (BM + wave_n * CFG::TN) * RB;

#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  f16x8 ra0[CA], rb0[CB], ra1[CA], rb1[CB];      // two stages in flight
  constexpr int kAuxNt = 2;   // buffer aux operand: bit 1 = nt
  // (plan flags, GemmArgs::flags bits 1 and 2: wave-uniform; the build-time knobs force them for experiment builds)
  const bool nt_streamed = HGEMM_RS_NT == 1 || (g.flags & 4) != 0;
  const bool nt_a = HGEMM_RS_NT == 2 || (nt_streamed && g.M >= g.N);
  const bool nt_b = HGEMM_RS_NT == 2 || (nt_streamed && g.M < g.N);
  const unsigned stg_id = (HGEMM_RS_STAGGER_MODE == 1 || (g.flags & 2) != 0) ? (blockIdx.x % NUM_XCD) * (unsigned)max(1, nk / NUM_XCD)
                        : HGEMM_RS_STAGGER_MODE == 2 ? (tile_id % 8u) * (unsigned)max(1, nk / 8) : tile_id * (unsigned)HGEMM_RS_STAGGER;
  const int stage0 = HGEMM_RS_STAGGER ? (int)(stg_id % (unsigned)nk) : 0;
#define RS_STAGE(T) ((stage0 + (T)) >= nk ? (stage0 + (T)) - nk : (stage0 + (T)))
#define RS_LD1(RS, VOFF, SOFF, NT) \
  __builtin_bit_cast(f16x8, (NT) ? __builtin_amdgcn_raw_buffer_load_b128(RS, VOFF, SOFF, kAuxNt) : __builtin_amdgcn_raw_buffer_load_b128(RS, VOFF, SOFF, 0))
#define RS_LOAD(RA, RBV, STAGE)                                                                     \
  do {                                                                                              \
    const uint32_t kb_ = kbyte0 + (uint32_t)RS_STAGE(STAGE) * (uint32_t)(BKS * 2);                  \
    if (nt_a) { _Pragma("unroll") for (int p = 0; p < CA; ++p) RA[p] = RS_LD1(rsA, voff_a + p * step_a, kb_, true); }   \
    else      { _Pragma("unroll") for (int p = 0; p < CA; ++p) RA[p] = RS_LD1(rsA, voff_a + p * step_a, kb_, false); }  \
    if (nt_b) { _Pragma("unroll") for (int p = 0; p < CB; ++p) RBV[p] = RS_LD1(rsB, voff_b + p * step_b, kb_, true); }  \
    else      { _Pragma("unroll") for (int p = 0; p < CB; ++p) RBV[p] = RS_LD1(rsB, voff_b + p * step_b, kb_, false); } \
  } while (0)
#define RS_WRITE_TO(BUF, RA, RBV)                                                                   \
  do {                                                                                              \
    _Pragma("unroll") for (int p = 0; p < CA; ++p) *(f16x8*)((BUF) + lds_of(p)) = RA[p];            \
    _Pragma("unroll") for (int p = 0; p < CB; ++p) *(f16x8*)((BUF) + BM * RB + lds_of(p)) = RBV[p]; \
  } while (0)
#define RS_WRITE(RA, RBV) RS_WRITE_TO(smem, RA, RBV)
#define RS_COMPUTE() RS_COMPUTE_FROM(smem)
#define RS_COMPUTE_FROM(BUF)                                                                        \
  do {                                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                             \
      const int fo = frag_lane ^ (ks << 6);                                                         \
      f16x8 af[FM], bf[FN];                                                                         \
      _Pragma("unroll") for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)((BUF) + a_row_base + i * 16 * RB + fo); \
      _Pragma("unroll") for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)((BUF) + b_row_base + j * 16 * RB + fo); \
      _Pragma("unroll") for (int i = 0; i < FM; ++i)                                                \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                              \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);     \
    }                                                                                               \
  } while (0)

  // prologue: stages 0 and 1 in flight, stage 0 into LDS, stage 2 into the freed registers.  The loads are
  // UNCONDITIONAL (stage index clamped to the last one): a load behind a branch makes hipcc count its vmcnt for
  // the path without it, i.e. wait for the younger stage as well, which would halve the bytes in flight.
  const int last = nk - 1;
  RS_LOAD(ra0, rb0, 0);
  RS_LOAD(ra1, rb1, min(1, last));
  RS_WRITE(ra0, rb0);
  RS_LOAD(ra0, rb0, min(2, last));
  __syncthreads();
  // invariant at the top of step t (t even): LDS = stage t, ra1/rb1 = stage t+1, ra0/rb0 = stage t+2 (in flight)
  int t = 0;
  if constexpr (CFG::LB == 2) {
    // two LDS buffers: stage t lives in buffer t & 1.  Step t writes stage t+1 into the other buffer (last read by the MFMAs of
    // stage t-1, which every wave finished in front of the barrier that closed step t-1), refills its registers with stage
    // t+3 and computes stage t; ONE barrier per stage makes stage t+1 visible and retires the reads of stage t.
    char* const buf0 = smem;
    char* const buf1 = smem + CFG::STAGE_BYTES;
    for (; t + 4 < nk; t += 2) {           // stages t+3 and t+4 exist: every load is real
      RS_WRITE_TO(buf1, ra1, rb1);         // stage t+1
      RS_LOAD(ra1, rb1, t + 3);
      RS_COMPUTE_FROM(buf0);               // stage t
      __syncthreads();
      RS_WRITE_TO(buf0, ra0, rb0);         // stage t+2
      RS_LOAD(ra0, rb0, t + 4);
      RS_COMPUTE_FROM(buf1);               // stage t+1
      __syncthreads();
    }
    for (; t + 1 < nk; t += 2) {           // the last (up to four) stages
      RS_WRITE_TO(buf1, ra1, rb1);
      if (t + 3 < nk) RS_LOAD(ra1, rb1, t + 3);
      RS_COMPUTE_FROM(buf0);
      __syncthreads();
      if (t + 2 < nk) RS_WRITE_TO(buf0, ra0, rb0);
      RS_COMPUTE_FROM(buf1);
      if (t + 2 < nk) __syncthreads();
    }
    if (t < nk) RS_COMPUTE_FROM(buf0);     // odd stage count: the last stage sits in buffer 0 (t is even)
    return;
  }
  // main loop: stages t+3 and t+4 exist, every load is real
  for (; t + 4 < nk; t += 2) {
    RS_COMPUTE();
    __syncthreads();                       // every wave is done with stage t
    RS_WRITE(ra1, rb1);                    // stage t+1 (hipcc waits for exactly these loads; stage t+2 flies on)
    RS_LOAD(ra1, rb1, t + 3);
    __syncthreads();
    RS_COMPUTE();                          // stage t+1
    __syncthreads();
    RS_WRITE(ra0, rb0);                    // stage t+2
    RS_LOAD(ra0, rb0, t + 4);
    __syncthreads();
  }
  // tail: the last (up to four) stages; no further loads are needed beyond stage nk-1
  for (; t + 1 < nk; t += 2) {
    RS_COMPUTE();
    __syncthreads();
    RS_WRITE(ra1, rb1);
    if (t + 3 < nk) RS_LOAD(ra1, rb1, t + 3);
    __syncthreads();
    RS_COMPUTE();
    if (t + 2 < nk) {
      __syncthreads();
      RS_WRITE(ra0, rb0);
      __syncthreads();
    }
  }
  if (t < nk) RS_COMPUTE();                // odd stage count: the last stage is already in LDS
#undef RS_LOAD
#undef RS_LD1
#undef RS_STAGE
#undef RS_WRITE
#undef RS_COMPUTE
#undef RS_WRITE_TO
#undef RS_COMPUTE_FROM
}
#endif  // __HIP_DEVICE_COMPILE__

// EPI_: EPI_C16 / EPI_SLAB / EPI_FUSED, + EPI_KTAIL: the variant for K % BKS != 0 (K % 8 == 0): the stage loop walks the whole
// stages of the work item (the last split runs to K), the remaining < BKS elements are accumulated by direct_k_tail.
template <class CFG, int EPI_>
// two workgroups per CU wherever the tile allows it (second argument = waves per SIMD: <= 256 registers)
__global__ void __launch_bounds__(CFG::THREADS, CFG::WGS_PER_CU) hgemm_tn_rs_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int EPI = EPI_ & 7;
  constexpr bool KTAIL = (EPI_ & EPI_KTAIL) != 0;
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, BKS = CFG::BKS;

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN, wave_n = wave % CFG::WN;

  const TileCoord tc = map_block(g, BM, BN);
  // K range of this work item; ktail variant: the host counts chunks on floor(K / BKS), the remainder rides on the last split
  const int k_items = (KTAIL && tc.split + 1 == g.splits ? g.K : min(g.K, tc.k_begin + g.k_chunk)) - tc.k_begin;
  const int nk = KTAIL ? k_items / BKS : tc.nk / (BKS / BK);   // whole stages (host: K chunk % BKS == 0)
  const unsigned tile_id = (unsigned)(tc.m0 / BM) + (unsigned)(tc.n0 / BN) * (unsigned)g.tiles_m + (unsigned)tc.split * 5u;

  f32x4 acc[FM][FN];
  rs_mainloop<CFG>(g, tc.m0, tc.n0, tc.k_begin, nk, tile_id, smem, tid, lane, wave_m, wave_n, acc);
  if constexpr (KTAIL) {
    if (nk * BKS < k_items)   // (workgroup-uniform)
      direct_k_tail<FM, FN>(g, tc.m0, tc.n0, wave_m * CFG::TM, wave_n * CFG::TN, tc.k_begin + nk * BKS, tc.k_begin + k_items, lane,
                            [&](int i, int j, const f16x8& b, const f16x8& a) __attribute__((always_inline)) {
                              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[i][j], 0, 0, 0);
                            });
  }

  if constexpr (EPI == EPI_FUSED) __syncthreads();   // the vote word lives at smem[0]: every wave must be done reading
  classic_epilogue<CFG, EPI>(g, tc, acc, smem, tid, lane, wave_m, wave_n);
#endif  // __HIP_DEVICE_COMPILE__
}

// Stream-K on family r (hgemm_kernel.hpp: StreamK): the skinny streaming shapes get their exact chip fill (12288 x 128 x 8192:
// 96 tiles of 128 x 128 on 256 CUs) and every workgroup of a cut tile starts its K walk somewhere else.
template <class CFG>
__global__ void __launch_bounds__(CFG::THREADS, CFG::WGS_PER_CU) hgemm_tn_rs_sk_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, BKS = CFG::BKS;

  // the stage buffer + the vote word behind it (ONE LDS object)
  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES + 64];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN, wave_n = wave % CFG::WN;

  const int G = gridDim.x, w = sk_logical_wg();
  const int run_begin = sk_start(g.sk, w, G), run_end = sk_start(g.sk, w + 1, G);
#pragma clang loop unroll(disable)
  for (int x = run_begin; x < run_end;) {
    const SkSegment s = sk_segment(g, w, run_begin, run_end, x, BM, BN);
    // every wave is done with the previous segment's last stage (and with the vote word) before the buffer is rewritten
    if (x != run_begin) __syncthreads();
    // (the stagger seed depends on the tile's coordinates and on where the segment starts, not on the raster group)
    const unsigned tile_id = (unsigned)(s.m0 / BM) + (unsigned)(s.n0 / BN) * (unsigned)g.tiles_m + (unsigned)s.k0 * 5u;
    f32x4 acc[FM][FN];
    rs_mainloop<CFG>(g, s.m0, s.n0, s.k0 * BKS, s.k1 - s.k0, tile_id, smem, tid, lane, wave_m, wave_n, acc);
    sk_epilogue<CFG>(g, s, acc, (volatile unsigned*)(smem + CFG::LDS_BYTES), tid, lane, wave_m, wave_n);
    x += s.k1 - s.k0;
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
