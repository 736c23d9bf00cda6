// Kernel registry (config id -> launch thunk) and the two helper kernels' launchers.
#include "hgemm_launch.hpp"
#include "hgemm_kernel_rg.hpp"

#include <algorithm>
#include <cstdio>

namespace hgemm_mi355x {

#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB) \
  extern template void launch_cfg<Cfg<BM, BN, WM, WN, MI, NB>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_SP(G, BM, BN, WM, WN, MI) \
  extern template void launch_sp<CfgSP<BM, BN, WM, WN, MI>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_SQ(G, BM, BN, WM, WN, KT, MI) \
  extern template void launch_sq<CfgSQ<BM, BN, WM, WN, KT, MI>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_RS(G, BM, BN, BKS, LB) \
  extern template void launch_rs<CfgRS<BM, BN, BKS, LB>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#define HGEMM_WD(G, FM, FN, KW) \
  extern template void launch_wd<CfgWD<FM, FN, KW>>(const GemmArgs&, int, hipStream_t, int, TimingSlot);
#include "hgemm_configs.def"
#undef HGEMM_CFG
#undef HGEMM_SP
#undef HGEMM_SQ
#undef HGEMM_RS
#undef HGEMM_WD

// The table holds host function pointers: keep it out of the device pass.
#if !defined(__HIP_DEVICE_COMPILE__)
thread_local LaunchTiming t_launch_timing;
// stream-K workgroups of a classic geometry one CU holds: LDS, 8 waves per SIMD, and the accumulator + fragment registers of a
// wave (512 per SIMD lane); at most 4 (more persistent workgroups only mean more slabs)
constexpr int sk_residency(int lds_bytes, int nw, int acc_regs) {
  const int by_lds = 160 * 1024 / lds_bytes, by_waves = 32 / nw, by_regs = 512 / (acc_regs + 64) * 4 / nw;
  const int r = by_lds < by_waves ? (by_lds < by_regs ? by_lds : by_regs) : (by_waves < by_regs ? by_waves : by_regs);
  return acc_regs * 64 * nw > 256 * 128 ? 0 : r < 1 ? 1 : r > 4 ? 4 : r;   // (no stream-K kernel beyond 256 x 128: hgemm_launch.hpp)
}
// "w<BM>x<BN>[_k4]" with the workgroup tile computed from the template arguments (static storage per instantiation)
template <int BM, int BN, int KW>
const char* wd_name() {
  static char buf[24];
  if (!buf[0]) snprintf(buf, sizeof buf, "w%dx%d%s", BM, BN, KW == 4 ? "_k4" : "");
  return buf;
}
#define HGEMM_STR2(x) #x
#define HGEMM_STR(x) HGEMM_STR2(x)
#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB)                                                    \
  {"t" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN) "_m" HGEMM_STR(MI)  \
   "_s" HGEMM_STR(NB),                                                                         \
   BM, BN, WM, WN, MI, NB, Cfg<BM, BN, WM, WN, MI, NB>::THREADS,                                \
   Cfg<BM, BN, WM, WN, MI, NB>::LDS_BYTES, &launch_cfg<Cfg<BM, BN, WM, WN, MI, NB>>, 0, true, 64, true,                      \
   sk_residency(Cfg<BM, BN, WM, WN, MI, NB>::LDS_BYTES + 64, Cfg<BM, BN, WM, WN, MI, NB>::NW, BM * BN / (64 * Cfg<BM, BN, WM, WN, MI, NB>::NW))},
#define HGEMM_SP(G, BM, BN, WM, WN, MI)
#define HGEMM_SQ(G, BM, BN, WM, WN, KT, MI)
#define HGEMM_RS(G, BM, BN, BKS, LB)
#define HGEMM_WD(G, FM, FN, KW)
const KernelEntry g_kernel_table[] = {
#include "hgemm_configs.def"
#undef HGEMM_CFG
#undef HGEMM_SP
#undef HGEMM_SQ
#undef HGEMM_RS
#define HGEMM_RS(G, BM, BN, BKS, LB)
#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB)
// MI = 16 members keep their round-1 names (tuned tables refer to plans by name); MI = 32 members add "_m32"
#define HGEMM_SP_NAME_16(BM, BN, WM, WN) "s" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN)
#define HGEMM_SP_NAME_32(BM, BN, WM, WN) "s" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN) "_m32"
#define HGEMM_SP(G, BM, BN, WM, WN, MI)                                                          \
  {HGEMM_SP_NAME_##MI(BM, BN, WM, WN), BM, BN, WM, WN, MI, 2,                                      \
   CfgSP<BM, BN, WM, WN, MI>::THREADS, CfgSP<BM, BN, WM, WN, MI>::LDS_BYTES + 64,                  \
   &launch_sp<CfgSP<BM, BN, WM, WN, MI>>, 256 * (160 * 1024 / (CfgSP<BM, BN, WM, WN, MI>::LDS_BYTES + 64)), true, 64, false, 0},
#define HGEMM_SQ_NAME_1_16(BM, BN, WM, WN) "q" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN)
#define HGEMM_SQ_NAME_2_16(BM, BN, WM, WN) "q" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN) "_k128"
#define HGEMM_SQ_NAME_1_32(BM, BN, WM, WN) "q" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_w" HGEMM_STR(WM) "x" HGEMM_STR(WN) "_m32"
#define HGEMM_SQ(G, BM, BN, WM, WN, KT, MI)                                                                     \
  {HGEMM_SQ_NAME_##KT##_##MI(BM, BN, WM, WN), BM, BN, WM, WN, MI, 2, CfgSQ<BM, BN, WM, WN, KT, MI>::THREADS,      \
   CfgSQ<BM, BN, WM, WN, KT, MI>::LDS_BYTES + (CfgSQ<BM, BN, WM, WN, KT, MI>::WGS == 2 ? 0 : 64), &launch_sq<CfgSQ<BM, BN, WM, WN, KT, MI>>, \
   256 * CfgSQ<BM, BN, WM, WN, KT, MI>::WGS, true, 64 * KT, MI == 16, 0},
#include "hgemm_configs.def"
#undef HGEMM_CFG
#undef HGEMM_SP
#undef HGEMM_SQ
#undef HGEMM_RS
#define HGEMM_CFG(G, BM, BN, WM, WN, MI, NB)
#define HGEMM_SP(G, BM, BN, WM, WN, MI)
#define HGEMM_SQ(G, BM, BN, WM, WN, KT, MI)
#define HGEMM_RS_SUFFIX_1 ""
#define HGEMM_RS_SUFFIX_2 "_d"
#define HGEMM_RS(G, BM, BN, BKS, LB)                                                                                   \
  {"r" HGEMM_STR(BM) "x" HGEMM_STR(BN) "_k" HGEMM_STR(BKS) HGEMM_RS_SUFFIX_##LB, BM, BN, 2, 2, 16, LB, CfgRS<BM, BN, BKS, LB>::THREADS, \
   CfgRS<BM, BN, BKS, LB>::LDS_BYTES, &launch_rs<CfgRS<BM, BN, BKS, LB>>, 0, true, BKS, true, CfgRS<BM, BN, BKS, LB>::WGS_PER_CU},
#include "hgemm_configs.def"
#undef HGEMM_RS
#undef HGEMM_WD
#define HGEMM_RS(G, BM, BN, BKS, LB)
// family "w": named by its WORKGROUP tile; "_k4" = the four waves split the K walk of one wave tile.  (K granularity 64: a
// split-K chunk is then a whole number of K = 32 slices whatever the split count.)
#define HGEMM_WD(G, FM, FN, KW)                                                                                          \
  {wd_name<CfgWD<FM, FN, KW>::BM, CfgWD<FM, FN, KW>::BN, KW>(), CfgWD<FM, FN, KW>::BM, CfgWD<FM, FN, KW>::BN, CfgWD<FM, FN, KW>::WM, \
   CfgWD<FM, FN, KW>::WN, 16, 1, CfgWD<FM, FN, KW>::THREADS, CfgWD<FM, FN, KW>::LDS_BYTES, &launch_wd<CfgWD<FM, FN, KW>>, 0, true, 64, false, 0},
#include "hgemm_configs.def"
};
#undef HGEMM_CFG
#undef HGEMM_SP
#undef HGEMM_SQ
#undef HGEMM_RS
#undef HGEMM_WD
const int g_num_kernels = (int)(sizeof(g_kernel_table) / sizeof(g_kernel_table[0]));
#endif  // !__HIP_DEVICE_COMPILE__

// Split-K combine: C[m][n] = fp16( sum_s partial[s][m][n] ), fp32 adds in split order
// (deterministic, unlike the reference's atomicAdd split-K, a100_F32F16F16F32/64_256_16384.cu:149-152).
__global__ void __launch_bounds__(256) hgemm_splitk_reduce_kernel(const float* __restrict__ partial,
                                                                  f16* __restrict__ C, int M, int N,
                                                                  int ldc, int splits) {
  const size_t total4 = ((size_t)M * N) >> 2;  // N % 4 == 0 on this path
  const size_t slab   = (size_t)M * N;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4;
       q += (size_t)gridDim.x * blockDim.x) {
    const size_t e = q << 2;
    f32x4 s = *(const f32x4*)(partial + e);
    int k = 1;
    // eight slabs' loads in flight per thread, added in split order: with tiny M*N this kernel is a
    // chain of dependent-looking global loads (measured: 64 splits cost +12 us before the unroll)
    for (; k + 8 <= splits; k += 8) {
      f32x4 p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = *(const f32x4*)(partial + (size_t)(k + u) * slab + e);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += p[u];
    }
    for (; k < splits; ++k) {
      const f32x4 p = *(const f32x4*)(partial + (size_t)k * slab + e);
      s += p;
    }
    const int m = (int)(e / N), n = (int)(e % N);
    f16x4 o = {(f16)s[0], (f16)s[1], (f16)s[2], (f16)s[3]};
    *(f16x4*)(C + (size_t)m * ldc + n) = o;
  }
}

// Combine of the hybrid schedule's tail pass: C tile t = fp16( sum_s partial[s * tail_tiles + t][BM][BN] ),
// slices added in K order.  One workgroup per (tail tile, 16-row band); the tile origin comes from the same
// raster map the GEMM kernel used.
__global__ void __launch_bounds__(256) hgemm_tail_reduce_kernel(const GemmArgs g, int BM, int BN) {
  const int bands = BM / 16;
  const int t = blockIdx.x / bands, band = blockIdx.x % bands;
  const TileCoord tc = map_logical(g, t, BM, BN);  // slice 0 of tail tile t: gives (m0, n0)
  const size_t slab = (size_t)BM * BN, stride = slab * g.tail_tiles;
  const float* base = g.partial + (size_t)t * slab;
  const int quads_per_row = BN / 4;
  for (int q = threadIdx.x; q < 16 * quads_per_row; q += blockDim.x) {
    const int r = band * 16 + q / quads_per_row, c = (q % quads_per_row) * 4;
    const int m = tc.m0 + r, n = tc.n0 + c;
    if (m >= g.M || n >= g.N) continue;
    const float* p = base + (size_t)r * BN + c;
    f32x4 s = *(const f32x4*)p;
    for (int k = 1; k < g.splits; ++k) s += *(const f32x4*)(p + (size_t)k * stride);
    f16x4 o = {(f16)s[0], (f16)s[1], (f16)s[2], (f16)s[3]};
    *(f16x4*)(g.C + (size_t)m * g.ldc + n) = o;
  }
}

// Any-shape / any-alignment fallback (one output per thread, fp32 accumulate).  Correctness
// net for shapes the MFMA paths do not accept (K % 8 != 0, unaligned views); never tuned.
__global__ void __launch_bounds__(256) hgemm_generic_kernel(const f16* __restrict__ A,
                                                            const f16* __restrict__ B,
                                                            f16* __restrict__ C, int M, int N, int K,
                                                            int lda, int ldb_rowmajor, int ldc) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  if (n >= N) return;
  // grid-stride over M: gridDim.y is capped at 65535 (M > 262140 used to fail the launch)
  for (int m = blockIdx.y * 4 + (threadIdx.x >> 6); m < M; m += gridDim.y * 4) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf((float)A[(size_t)m * lda + k], (float)B[(size_t)k * ldb_rowmajor + n], s);
    C[(size_t)m * ldc + n] = (f16)s;
  }
}

void launch_splitk_reduce(const float* partial, f16* C, int M, int N, int ldc, int splits,
                          hipStream_t stream, TimingSlot ts) {
  const size_t quads = ((size_t)M * N) >> 2;
  // small outputs: 64-thread blocks so the (latency-bound) slab reads spread over more CUs
  const int threads = quads <= 64 * 1024 ? 64 : 256;
  int grid = (int)((quads + threads - 1) / threads);
  if (grid > 256 * 8) grid = 256 * 8;  // grid-stride beyond 8 blocks per CU
  if (grid < 1) grid = 1;
  HGEMM_LAUNCH(hgemm_splitk_reduce_kernel, grid, threads, stream, ts, partial, C, M, N, ldc, splits);
}

void launch_tail_reduce(const GemmArgs& g, int BM, int BN, hipStream_t stream, TimingSlot ts) {
  HGEMM_LAUNCH(hgemm_tail_reduce_kernel, g.tail_tiles * (BM / 16), 256, stream, ts, g, BM, BN);
}

// largest power of two <= 8 (halfs) that divides the row stride, K and the pointer's alignment
static int piece_width(const void* p, int ld, int K) {
  int w = 8;
  while (w > 1 && ((ld % w) || (K % w) || (reinterpret_cast<uintptr_t>(p) % (2 * w)))) w >>= 1;
  return w;
}

void launch_ragged(const GemmArgs& g0, hipStream_t stream, TimingSlot ts) {
  GemmArgs g = g0;
  const int wa = piece_width(g.A, g.lda, g.K), wb = piece_width(g.Bt, g.ldb, g.K);
  const int vec_c = ((g.N & 3) == 0) && ((g.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 7) == 0);
  using Small = Cfg<64, 64, 2, 2, 16, 2>;
  using Big = Cfg<128, 128, 2, 2, 16, 2>;
  const long tiles_big = (long)((g.M + 127) / 128) * ((g.N + 127) / 128);
  if (tiles_big >= 256) {
    g.tiles_m = (g.M + 127) / 128; g.tiles_n = (g.N + 127) / 128;
    HGEMM_LAUNCH((hgemm_tn_ragged_kernel<Big>), g.tiles_m * g.tiles_n, Big::THREADS, stream, ts, g, wa, wb, vec_c);
  } else {
    g.tiles_m = (g.M + 63) / 64; g.tiles_n = (g.N + 63) / 64;
    HGEMM_LAUNCH((hgemm_tn_ragged_kernel<Small>), g.tiles_m * g.tiles_n, Small::THREADS, stream, ts, g, wa, wb, vec_c);
  }
}

void launch_generic(const f16* A, const f16* B, f16* C, int M, int N, int K, int lda, int ldb,
                    int ldc, hipStream_t stream, TimingSlot ts) {
  dim3 grid((N + 63) / 64, (unsigned)std::min<long>(((long)M + 3) / 4, 65535));
  HGEMM_LAUNCH(hgemm_generic_kernel, grid, 256, stream, ts, A, B, C, M, N, K, lda, ldb, ldc);
}

}  // namespace hgemm_mi355x
