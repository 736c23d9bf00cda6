// Host-side launch thunks for the kernel instantiations listed in hgemm_configs.def.
#pragma once
#include "hgemm_kernel_rs.hpp"
#include "hgemm_kernel_sq.hpp"
#include "hgemm_kernel_wd.hpp"

#include <hip/hip_ext.h>

namespace hgemm_mi355x {

// One thunk per geometry; `epi` selects the epilogue (EPI_C16 / EPI_SLAB / EPI_FUSED) or, for the classic and the
// register-staged family, the stream-K kernel (EPI_STREAMK: `grid` is then the number of persistent workgroups).  No per-call
// attribute setting, allocation or synchronisation happens here (the reference calls
// cudaFuncSetAttribute on every invocation, kernels/a100_F32F16F16F32/64_4096_64.cu:256-261).
//
// Timing hook (hgemm_mi355x_time_next_launch): when armed, the FIRST dispatch of the next GEMM call of this
// thread carries the start event and its LAST dispatch (the main kernel, or the combine kernel of a
// two-pass / hybrid plan) the stop event, both on the dispatches' own AQL packets
// (hipExtLaunchKernelGGL): their distance is the plan's device time as the profiler sees it, with no
// marker packets between kernels.
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchTiming t_launch_timing;

// which of the armed events this dispatch carries
struct TimingSlot { hipEvent_t start = nullptr, stop = nullptr; };
inline TimingSlot timing_slot(bool first, bool last) {
  TimingSlot s;
  if (first) { s.start = t_launch_timing.start; t_launch_timing.start = nullptr; }
  if (last) { s.stop = t_launch_timing.stop; t_launch_timing.stop = nullptr; }
  return s;
}

#define HGEMM_LAUNCH(KERNEL, GRID, THREADS, STREAM, SLOT, ...)                                              \
  do {                                                                                                      \
    if ((SLOT).start || (SLOT).stop)                                                                        \
      hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, STREAM, (SLOT).start, (SLOT).stop, 0,     \
                            __VA_ARGS__);                                                                   \
    else                                                                                                    \
      hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, STREAM, __VA_ARGS__);                        \
  } while (0)

template <class CFG>
void launch_cfg(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  if (epi == EPI_STREAMK) {
    // (the 256 x 256 members have no stream-K kernel: with 128 accumulator registers per lane the combine spills, and a
    // 256 KiB slab per cut is more than the imbalance it would remove -- family q's hybrid tail covers those shapes)
    if constexpr (CFG::BM * CFG::BN <= 256 * 128) HGEMM_LAUNCH((hgemm_tn_sk_kernel<CFG>), grid, CFG::THREADS, stream, ts, g);
  } else if (epi == EPI_FUSED)
    HGEMM_LAUNCH((hgemm_tn_kernel<CFG, EPI_FUSED>), grid, CFG::THREADS, stream, ts, g);
  else if (epi == EPI_SLAB)
    HGEMM_LAUNCH((hgemm_tn_kernel<CFG, EPI_SLAB>), grid, CFG::THREADS, stream, ts, g);
  else
    HGEMM_LAUNCH((hgemm_tn_kernel<CFG, EPI_C16>), grid, CFG::THREADS, stream, ts, g);
}

template <class CFG>
void launch_sp(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  const bool wide = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  if (epi == EPI_FUSED)
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, SP_EPI_FUSED>), grid, CFG::THREADS, stream, ts, g);
  else if (epi == EPI_SLAB)
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, SP_EPI_SLAB>), grid, CFG::THREADS, stream, ts, g);
  else if (wide)
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, SP_EPI_WIDE>), grid, CFG::THREADS, stream, ts, g);
  else
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, SP_EPI_NARROW>), grid, CFG::THREADS, stream, ts, g);
}

// FLAG = 0, or EPI_KTAIL for the "ktail" variants (whole stages through the pipeline + a direct tail: hgemm_kernel_sq.hpp / _rs.hpp)
template <class CFG, int FLAG>
void launch_sq_variant(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  const bool wide = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  if (epi == EPI_FUSED)
    HGEMM_LAUNCH((hgemm_tn_sq_kernel<CFG, SP_EPI_FUSED | FLAG>), grid, CFG::THREADS, stream, ts, g);
  else if (epi == EPI_SLAB)
    HGEMM_LAUNCH((hgemm_tn_sq_kernel<CFG, SP_EPI_SLAB | FLAG>), grid, CFG::THREADS, stream, ts, g);
  else if (wide)
    HGEMM_LAUNCH((hgemm_tn_sq_kernel<CFG, SP_EPI_WIDE | FLAG>), grid, CFG::THREADS, stream, ts, g);
  else
    HGEMM_LAUNCH((hgemm_tn_sq_kernel<CFG, SP_EPI_NARROW | FLAG>), grid, CFG::THREADS, stream, ts, g);
}

template <class CFG>
void launch_sq(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  if constexpr (CFG::MI == 16) {   // (the 32x32x16 members have no K tail: the host never sends them a K that is not whole stages)
    if (g.K % (BK * CFG::KT) != 0) return launch_sq_variant<CFG, EPI_KTAIL>(g, grid, stream, epi, ts);
    // plan flag HGEMM_PLAN_XCD_STAGGER (GemmArgs::flags bit 1): whole stages only, the 16x16x32 members only
    if (g.flags & 2) return launch_sq_variant<CFG, EPI_KSTAGGER>(g, grid, stream, epi, ts);
  }
  launch_sq_variant<CFG, 0>(g, grid, stream, epi, ts);
}

template <class CFG, int FLAG>
void launch_rs_variant(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  if (epi == EPI_FUSED)
    HGEMM_LAUNCH((hgemm_tn_rs_kernel<CFG, EPI_FUSED | FLAG>), grid, CFG::THREADS, stream, ts, g);
  else if (epi == EPI_SLAB)
    HGEMM_LAUNCH((hgemm_tn_rs_kernel<CFG, EPI_SLAB | FLAG>), grid, CFG::THREADS, stream, ts, g);
  else
    HGEMM_LAUNCH((hgemm_tn_rs_kernel<CFG, EPI_C16 | FLAG>), grid, CFG::THREADS, stream, ts, g);
}

template <class CFG>
void launch_rs(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  if (epi == EPI_STREAMK)   // (the host never asks for stream-K with a K tail)
    HGEMM_LAUNCH((hgemm_tn_rs_sk_kernel<CFG>), grid, CFG::THREADS, stream, ts, g);
  else if (g.K % CFG::BKS != 0)
    launch_rs_variant<CFG, EPI_KTAIL>(g, grid, stream, epi, ts);
  else
    launch_rs_variant<CFG, 0>(g, grid, stream, epi, ts);
}

template <class CFG>
void launch_wd(const GemmArgs& g, int grid, hipStream_t stream, int epi, TimingSlot ts) {
  if (epi == EPI_FUSED)
    HGEMM_LAUNCH((hgemm_tn_wd_kernel<CFG, EPI_FUSED>), grid, CFG::THREADS, stream, ts, g);
  else if (epi == EPI_SLAB)
    HGEMM_LAUNCH((hgemm_tn_wd_kernel<CFG, EPI_SLAB>), grid, CFG::THREADS, stream, ts, g);
  else
    HGEMM_LAUNCH((hgemm_tn_wd_kernel<CFG, EPI_C16>), grid, CFG::THREADS, stream, ts, g);
}

struct KernelEntry {
  const char* name;
  int bm, bn, wm, wn, mi, nbuf;
  int threads, lds_bytes;
  void (*launch)(const GemmArgs&, int, hipStream_t, int, TimingSlot);
  int persistent_wgs;  // > 0: the kernel walks its work items itself, launch at most this many workgroups
  bool has_fused;      // the family has a single-launch split-K epilogue (EPI_FUSED)
  int kgran;           // K granularity of one pipeline stage (64, or 128 / 256 for the deep-stage members): every
                       // split-K chunk is a multiple, and so is K unless `ktail`
  bool ktail;          // the kernel takes a K that is not a multiple of kgran (K % 8 == 0): the classic family zero-fills a partial
                       // last LDS-DMA step; families q (MI = 16) and r run their "ktail" variants: whole stages through the
                       // pipeline (so K >= kgran), the rest from fragments loaded straight from global memory (direct_k_tail)
  int sk_wgs_per_cu;   // > 0: the family has a stream-K kernel (EPI_STREAMK); workgroups of it one CU holds (default G = 256 x this)
};

extern const KernelEntry g_kernel_table[];
extern const int g_num_kernels;

void launch_splitk_reduce(const float* partial, f16* C, int M, int N, int ldc, int splits,
                          hipStream_t stream, TimingSlot ts);
void launch_tail_reduce(const GemmArgs& g, int BM, int BN, hipStream_t stream, TimingSlot ts);
// Any-shape kernels (hgemm_kernel_rg.hpp): MFMA with register staging for ragged K / N / unaligned views,
// and the one-output-per-thread reference kernel (config id -1).
void launch_ragged(const GemmArgs& g, hipStream_t stream, TimingSlot ts);
void launch_generic(const f16* A, const f16* B, f16* C, int M, int N, int K, int lda, int ldb,
                    int ldc, hipStream_t stream, TimingSlot ts);

}  // namespace hgemm_mi355x
