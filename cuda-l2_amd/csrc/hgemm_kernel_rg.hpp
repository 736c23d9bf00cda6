// Any-shape MFMA kernel: C[M,N] = A[M,K] * B[K,N] for shapes the LDS-DMA families cannot take
// (K % 8 != 0, N % 4 != 0, row strides or base pointers that are not 16-byte aligned; K % 64 != 0 with K % 8 == 0 has its own
// paths since rounds 2 / 4: the classic family pads its last LDS-DMA step, families q and r run their ktail variants).
//
// The reference covers arbitrary sizes with harness-side zero padding to the tile size
// (tools/utils.py:8-36, README.md:83-86: "pad to the nearest larger config"); here the padding happens
// on the way into LDS instead, so the caller's tensors are used as they are:
//   * operands are staged global -> VGPR -> LDS (ds_write_b128) instead of by LDS-DMA, in pieces of
//     w = 8 / 4 / 2 / 1 halfs, w = the largest power of two that divides the row stride, K and the
//     pointer alignment: a piece never straddles the end of a row, so a piece is either loaded whole or
//     replaced by zeros (k >= K, row >= M / N).  Nothing outside the operand windows is ever read;
//   * the LDS image, the XOR swizzle, the fragment reads and the MFMA loop are those of hgemm_tn_kernel
//     (two stages, one barrier per K-step, the next tile's global loads issued ahead of the MFMAs);
//   * the epilogue stores 8 bytes per lane when N % 4 == 0, ldc % 4 == 0 and C is 8-byte aligned, single
//     halfs with per-element predication otherwise.
// It is a correctness-first kernel with matrix-core throughput, not a tuned one: the 1000 grid shapes never
// reach it.
#pragma once

#include "hgemm_kernel.hpp"

namespace hgemm_mi355x {

#if defined(__HIP_DEVICE_COMPILE__)
// 8 halfs starting at p, of which the first `valid` (a multiple of w) exist; the rest are zeros.
__device__ __forceinline__ f16x8 rg_load_chunk(const f16* p, int valid, int w) {
  f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (w == 8) {
    if (valid >= 8) v = *(const f16x8*)p;
  } else if (w == 4) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (valid >= 4 * (h + 1)) {
        const f16x4 q = *(const f16x4*)(p + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * h + e] = q[e];
      }
  } else if (w == 2) {
    using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
#pragma unroll
    for (int h = 0; h < 4; ++h)
      if (valid >= 2 * (h + 1)) {
        const f16x2 q = *(const f16x2*)(p + 2 * h);
        v[2 * h] = q[0];
        v[2 * h + 1] = q[1];
      }
  } else {
#pragma unroll
    for (int h = 0; h < 8; ++h)
      if (valid > h) v[h] = p[h];
  }
  return v;
}
#endif

// CFG = Cfg<BM, BN, WM, WN, 16, 2>.  wa / wb: piece width (halfs) of the A / Bt loads; vec_c: 8-byte C stores allowed.
template <class CFG>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_ragged_kernel(const GemmArgs g, int wa, int wb, int vec_c) {
  prefetch_kernargs<sizeof(GemmArgs) + 12>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, THREADS = CFG::THREADS;
  constexpr int CA = BM * 8 / THREADS, CB = BN * 8 / THREADS;   // 16-byte chunks per thread per operand
  static_assert(CFG::MI == 16 && CFG::NBUF == 2, "ragged kernel geometry");
  static_assert((BM * 8) % THREADS == 0 && (BN * 8) % THREADS == 0, "whole chunks per thread");

  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES];

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wave_m = wave / CFG::WN, wave_n = wave % CFG::WN;

  // plain row-major tile raster (no XCD remap: these shapes are small or rare)
  const int tile = blockIdx.x;
  const int m0 = (tile / g.tiles_n) * BM, n0 = (tile % g.tiles_n) * BN;
  const int nk = (g.K + BK - 1) / BK;

  f16x8 ra[CA], rb[CB];
  auto load_tile = [&](int t) {
    const int k0 = t * BK;
#pragma unroll
    for (int p = 0; p < CA; ++p) {
      const int id = tid + p * THREADS, r = id >> 3, c = id & 7;
      const int k = k0 + c * 8, row = m0 + r;
      const int valid = (row < g.M) ? max(0, min(8, g.K - k)) : 0;
      ra[p] = rg_load_chunk(g.A + (size_t)min(row, g.M - 1) * g.lda + min(k, g.K - 1), valid, wa);
    }
#pragma unroll
    for (int p = 0; p < CB; ++p) {
      const int id = tid + p * THREADS, r = id >> 3, c = id & 7;
      const int k = k0 + c * 8, row = n0 + r;
      const int valid = (row < g.N) ? max(0, min(8, g.K - k)) : 0;
      rb[p] = rg_load_chunk(g.Bt + (size_t)min(row, g.N - 1) * g.ldb + min(k, g.K - 1), valid, wb);
    }
  };
  auto write_tile = [&](char* st) {
#pragma unroll
    for (int p = 0; p < CA; ++p) {
      const int id = tid + p * THREADS, r = id >> 3, c = id & 7;
      *(f16x8*)(st + r * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4)) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < CB; ++p) {
      const int id = tid + p * THREADS, r = id >> 3, c = id & 7;
      *(f16x8*)(st + (BM + r) * ROW_BYTES + ((c ^ ((r >> 1) & 7)) << 4)) = rb[p];
    }
  };

  int frag_off[2];
  {
    const int lr = lane & 15, lq = lane >> 4, sw = (lr >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag_off[ks] = lr * ROW_BYTES + (((ks * 4 + lq) ^ sw) << 4);
  }
  const int a_row_base = wave_m * CFG::TM * ROW_BYTES;
  const int b_row_base = BM * ROW_BYTES + wave_n * CFG::TN * ROW_BYTES;

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  write_tile(smem);
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    if (t + 1 < nk) load_tile(t + 1);          // global loads fly behind this tile's MFMAs
    const char* st = smem + (t & 1) * CFG::STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *(const f16x8*)(st + a_row_base + i * 16 * ROW_BYTES + frag_off[ks]);
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(st + b_row_base + j * 16 * ROW_BYTES + frag_off[ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (t + 1 < nk) write_tile(smem + ((t + 1) & 1) * CFG::STAGE_BYTES);   // the stage read in step t-1
    __syncthreads();
  }

  // epilogue: lane holds C[m][n .. n+3], m = lane & 15, n = (lane >> 4) * 4 (operands swapped as in hgemm_tn_kernel)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wave_m * CFG::TM + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wave_n * CFG::TN + j * 16 + (lane >> 4) * 4;
      if (m >= g.M || n >= g.N) continue;
      f16* dst = g.C + (size_t)m * g.ldc + n;
      if (vec_c) {
        const f16x4 o = {(f16)acc[i][j][0], (f16)acc[i][j][1], (f16)acc[i][j][2], (f16)acc[i][j][3]};
        *(f16x4*)dst = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) dst[e] = (f16)acc[i][j][e];
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

}  // namespace hgemm_mi355x
