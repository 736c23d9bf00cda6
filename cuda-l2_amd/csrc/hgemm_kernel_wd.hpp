// Family "w": wave-direct kernel for the launch-bound end of the grid -- no LDS staging, no barrier in the K walk.
//
// Counterpart of the reference's tiny-shape kernels, one warp per 16 x 16 output tile straight from global memory
// (kernels/a100_F32F16F16F32/64_64_64.cu:9-41,67-74).  Why here (round-3 measurements): 64 x 4096 x 64 ran a
// 4-deep-LDS-ring geometry for a K of ONE pipeline step -- LDS-DMA, wait, barrier, fragment reads, MFMAs, store: 3.3 us of
// kernel inside a 6.3 us call, against a 0.17 us HBM bound -- and the tiny-M*N / long-K shapes (64 x 256 x 2048 ...) lost to
// hipBLASLt's MT16x16x512 kernels (7.0 vs 5.5 us), whose waves each walk a slice of K.
//
//   * every wave owns an (FM x 16) x (FN x 16) output tile and loads its MFMA fragments DIRECTLY from global memory: the
//     operand layout of v_mfma_f32_16x16x32_f16 (lane l holds 8 consecutive K of row l & 15 at k = 8 (l >> 4)) is one
//     global_load_dwordx4 per fragment from a K-contiguous row, 16 rows x 64 contiguous bytes per wave instruction;
//   * KW = 1: the four waves of a workgroup form a 2 x 2 grid of wave tiles.  Nothing is shared between waves: no LDS,
//     no barrier; a K of 64 is two MFMA slices behind one round trip to memory;
//   * KW = 4: the four waves share ONE wave tile and take the K = 32 slices round-robin (slice s -> wave s mod 4), four
//     slices in flight per wave; their partials meet in LDS (one barrier) and the tile's quads are dealt out to the waves
//     for the epilogue.  Together with split-K across workgroups this is how a 64 x 64 x 4096 problem reaches all CUs;
//   * rows past the M / N edge are clamped on load (never stored); K must be a multiple of 32 per split;
//   * epilogues: fp16 C, fp32 slabs for the two-pass combine, single-launch split-K (own compact slab layout: quad x of
//     lane l at (x * 64 + l) * 4 -- the same arrival-counter protocol as the other families).
#pragma once

#include "hgemm_kernel.hpp"

namespace hgemm_mi355x {

template <int FM_, int FN_, int KW_>
struct CfgWD {
  static constexpr int FM = FM_, FN = FN_, KW = KW_;
  static constexpr int NW = 4, THREADS = 256;
  static constexpr int WM = KW == 1 ? 2 : 1, WN = KW == 1 ? 2 : 1;     // wave grid of the workgroup tile
  static constexpr int TM = FM * 16, TN = FN * 16;                      // wave tile
  static constexpr int BM = WM * TM, BN = WN * TN;                      // workgroup tile
  static constexpr int NQUAD = FM * FN;                                 // f32x4 accumulator quads per lane
  static constexpr int LDS_BYTES = KW == 1 ? 64 : KW * NQUAD * 64 * 16 + 64;   // KW partial tiles + the vote word
  static_assert(KW == 1 || KW == 4, "K walk by one wave or by all four");
  static_assert(FM * FN <= 16, "accumulators + one unrolled trip of fragments (at most 32 x 4 registers) stay below 256 registers");
};

template <class CFG, int EPI>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_wd_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int FM = CFG::FM, FN = CFG::FN, KW = CFG::KW, NQUAD = CFG::NQUAD;
  __shared__ __attribute__((aligned(16))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = KW == 1 ? wave / CFG::WN : 0, wave_n = KW == 1 ? wave % CFG::WN : 0;

  const TileCoord tc = map_block(g, CFG::BM, CFG::BN);
  const int l15 = lane & 15, lq = lane >> 4;
  // per-lane row pointers (rows past the edge clamped: their products are never stored); 64-bit addressing throughout
  const f16* pa[FM];
  const f16* pb[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) pa[i] = g.A + (size_t)min(tc.m0 + wave_m * CFG::TM + i * 16 + l15, g.M - 1) * g.lda + lq * 8;
#pragma unroll
  for (int j = 0; j < FN; ++j) pb[j] = g.Bt + (size_t)min(tc.n0 + wave_n * CFG::TN + j * 16 + l15, g.N - 1) * g.ldb + lq * 8;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // K = 32 slices of this work item; wave w of a KW = 4 workgroup takes slices w, w + 4, ...
  const int k_end = min(g.K, tc.k_begin + g.k_chunk);
  const int nslices = (k_end - tc.k_begin) / 32;
  // U slices per trip: U * (FM + FN) 16-byte loads in flight per lane behind ONE wait; trips of (16, 8,) 4, then one of 2, then one of 1
  // (a K of 64 on a KW = 1 member is a single trip of two slices: one round trip to memory for the whole kernel)
  int s = KW == 1 ? 0 : wave;
  auto trip = [&](auto u_tag) {
    constexpr int U = decltype(u_tag)::value;
    f16x8 af[U][FM], bf[U][FN];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = tc.k_begin + (s + u * KW) * 32;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[u][i] = *(const f16x8*)(pa[i] + k);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[u][j] = *(const f16x8*)(pb[j] + k);
    }
    // deep trips: every load goes out before the first MFMA (left alone, hipcc trades the loads in flight for a smaller register
    // count: 12 of 32 at a time, to keep 8 waves per SIMD that these launch-bound grids never have)
    if constexpr (U >= 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[u][j], af[u][i], acc[i][j], 0, 0, 0);
    s += U * KW;
  };
  // Round 5: deeper trips for the long K walks of the small wave tiles -- up to 32 sixteen-byte loads per lane (128 registers) in
  // flight: 16 slices for a 16 x 16 wave tile (K = 512 per wave and round trip, 2048 per `_k4` workgroup), 8 for 32 x 32 / 16 x 32.
  // With four slices a 64 x 64 x 4096 walk was 8 dependent round trips per wave (or a 4-way split-K and a second, serial
  // combine phase); hipBLASLt's MT16x16x512 kernels on these shapes have 512 of K in flight per wave.
  constexpr int UMAX = 32 / (FM + FN) >= 16 ? 16 : 32 / (FM + FN) >= 8 ? 8 : 4;
  if constexpr (UMAX >= 16) {
    while (s + 15 * KW < nslices) trip(std::integral_constant<int, 16>{});
  }
  if constexpr (UMAX >= 8) {
    while (s + 7 * KW < nslices) trip(std::integral_constant<int, 8>{});
  }
  while (s + 3 * KW < nslices) trip(std::integral_constant<int, 4>{});
  if (s + KW < nslices) trip(std::integral_constant<int, 2>{});
  if (s < nslices) trip(std::integral_constant<int, 1>{});

  if constexpr (KW == 1) {
    // four independent wave tiles: the classic epilogues (fp16 C / slabs / single-launch split-K with the vote word at smem[0])
    using ECFG = Cfg<CFG::BM, CFG::BN, 2, 2, 16, 2>;
    static_assert(ECFG::FM == FM && ECFG::FN == FN && ECFG::THREADS == CFG::THREADS, "epilogue geometry");
    if constexpr (EPI == EPI_FUSED) {
      constexpr int SLAB = CFG::BM * CFG::BN;
      const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) fused_store(rsP, fused_off<CFG::THREADS>(tc.item, SLAB, i * FN + j, tid), acc[i][j]);
      if (!fused_publish_and_vote(g, tc.tile, (volatile unsigned*)smem, tid)) return;
      {   // the slabs in split order, up to 8 of them in flight per round trip (fused_combine, hgemm_kernel.hpp)
        f32x4 sum[NQUAD];
        fused_combine<CFG::THREADS, NQUAD, FusedBatch<NQUAD>::U>(rsP, g.splits, g.tiles_m * g.tiles_n, tc.tile, SLAB, tid, [](int x) { return x; }, sum);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = sum[i * FN + j];
      }
      store_tile<16, FM, FN, CFG::TM, CFG::TN, false>(g, tc, wave_m, wave_n, lane, acc);
    } else {
      store_tile<16, FM, FN, CFG::TM, CFG::TN, EPI == EPI_SLAB>(g, tc, wave_m, wave_n, lane, acc);
    }
  } else {
    // ---- the four K walks meet in LDS: partial[w][x][lane] (16 bytes each), then quad x belongs to wave x mod 4 --------
    f32x4* part = (f32x4*)smem;
    volatile unsigned* flag = (volatile unsigned*)(smem + KW * NQUAD * 64 * 16);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) part[(wave * NQUAD + i * FN + j) * 64 + lane] = acc[i][j];
    __syncthreads();
    constexpr int MYQ = (NQUAD + KW - 1) / KW;          // quads per wave in the epilogue
    f32x4 sum[MYQ];
#pragma unroll
    for (int q = 0; q < MYQ; ++q) {
      const int x = wave + q * KW;
      sum[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (x < NQUAD) {
        sum[q] = part[(0 * NQUAD + x) * 64 + lane];
#pragma unroll
        for (int w = 1; w < KW; ++w) sum[q] += part[(w * NQUAD + x) * 64 + lane];   // in wave order: deterministic
      }
    }
    const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
    constexpr int SLAB = CFG::BM * CFG::BN;             // floats per work item = NQUAD * 64 * 4
    bool store_c = EPI == EPI_C16;
    if constexpr (EPI == EPI_FUSED) {
#pragma unroll
      for (int q = 0; q < MYQ; ++q) {
        const int x = wave + q * KW;
        if (x < NQUAD) fused_store(rsP, ((uint32_t)tc.item * SLAB + (uint32_t)(x * 64 + lane) * 4u) * 4u, sum[q]);
      }
      // (the vote word sits behind the partials; its own barriers order it against the reads above)
      if (!fused_publish_and_vote(g, tc.tile, flag, tid)) return;
      // (quad x of lane l sits at (x * 64 + l) * 4 floats of a slab: fused_off's layout with a "workgroup" of 64 threads)
#pragma unroll
      for (int q = 0; q < MYQ; ++q) {
        const int x = wave + q * KW;
        if (x < NQUAD) {
          f32x4 one[1];
          fused_combine<64, 1, 32>(rsP, g.splits, g.tiles_m * g.tiles_n, tc.tile, SLAB, lane, [x](int) { return x; }, one);
          sum[q] = one[0];
        }
      }
      store_c = true;
    }
#pragma unroll
    for (int q = 0; q < MYQ; ++q) {
      const int x = wave + q * KW;
      if (x >= NQUAD) continue;
      const int m = tc.m0 + (x / FN) * 16 + l15, n = tc.n0 + (x % FN) * 16 + lq * 4;
      if (m >= g.M || n >= g.N) continue;               // (N % 4 == 0 on this path: a quad is inside or outside)
      if (store_c) {
        const f16x4 o = {(f16)sum[q][0], (f16)sum[q][1], (f16)sum[q][2], (f16)sum[q][3]};
        HGEMM_STORE_C(g, (f16x4*)(g.C + (size_t)m * g.ldc + n), o);
      } else {
        *(f32x4*)(tc.slab + (size_t)(m - tc.m0) * tc.slab_ld + (n - tc.n0)) = sum[q];   // two-pass slabs [splits][M][N]
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
