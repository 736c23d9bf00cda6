// Host-side launch thunks for the kernel instantiations listed in hgemm_configs.def.
#pragma once
#include "hgemm_kernel_pp.hpp"
#include "hgemm_kernel_sp.hpp"

#include <hip/hip_ext.h>

namespace hgemm_mi355x {

// One thunk per geometry; `splitk` selects the fp32-slab epilogue.  No per-call attribute
// setting, allocation or synchronisation happens here (the reference calls
// cudaFuncSetAttribute on every invocation, kernels/a100_F32F16F16F32/64_4096_64.cu:256-261).
// One-shot timing hook (hgemm_mi355x_time_next_launch): when armed, the next main-kernel dispatch of
// this thread carries the two events on its own AQL packet (hipExtLaunchKernelGGL), so their distance
// is the kernel's execution time as the profiler sees it -- no marker packets between kernels.
struct LaunchTiming { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local LaunchTiming t_launch_timing;

#define HGEMM_LAUNCH(KERNEL, GRID, THREADS, STREAM, ARGS)                                                   \
  do {                                                                                                      \
    if (t_launch_timing.start) {                                                                            \
      hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, STREAM, t_launch_timing.start,            \
                            t_launch_timing.stop, 0, ARGS);                                                 \
      t_launch_timing = LaunchTiming{};                                                                     \
    } else {                                                                                                \
      hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(THREADS), 0, STREAM, ARGS);                               \
    }                                                                                                       \
  } while (0)

template <class CFG>
void launch_cfg(const GemmArgs& g, int grid, hipStream_t stream, bool splitk) {
  if (splitk)
    HGEMM_LAUNCH((hgemm_tn_kernel<CFG, true>), grid, CFG::THREADS, stream, g);
  else
    HGEMM_LAUNCH((hgemm_tn_kernel<CFG, false>), grid, CFG::THREADS, stream, g);
}

template <class CFG>
void launch_pp(const GemmArgs& g, int grid, hipStream_t stream, bool splitk) {
  if constexpr (CFG::MODE == 0) {
    if (splitk)
      HGEMM_LAUNCH((hgemm_tn_pp_kernel<CFG, true>), grid, CFG::THREADS, stream, g);
    else
      HGEMM_LAUNCH((hgemm_tn_pp_kernel<CFG, false>), grid, CFG::THREADS, stream, g);
  } else {
    if (splitk)
      HGEMM_LAUNCH((hgemm_tn_cp_kernel<CFG, true>), grid, CFG::THREADS, stream, g);
    else
      HGEMM_LAUNCH((hgemm_tn_cp_kernel<CFG, false>), grid, CFG::THREADS, stream, g);
  }
}

template <class CFG>
void launch_sp(const GemmArgs& g, int grid, hipStream_t stream, bool splitk) {
  const bool wide = ((g.N & 7) == 0) && ((g.ldc & 7) == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0);
  if (splitk)
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, 2>), grid, CFG::THREADS, stream, g);
  else if (wide)
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, 1>), grid, CFG::THREADS, stream, g);
  else
    HGEMM_LAUNCH((hgemm_tn_sp_kernel<CFG, 0>), grid, CFG::THREADS, stream, g);
}

struct KernelEntry {
  const char* name;
  int bm, bn, wm, wn, mi, nbuf;
  int threads, lds_bytes;
  void (*launch)(const GemmArgs&, int, hipStream_t, bool);
  int persistent_wgs;  // > 0: the kernel walks its work items itself, launch at most this many workgroups
};

extern const KernelEntry g_kernel_table[];
extern const int g_num_kernels;

void launch_splitk_reduce(const float* partial, f16* C, int M, int N, int ldc, int splits,
                          hipStream_t stream);
void launch_tail_reduce(const GemmArgs& g, int BM, int BN, hipStream_t stream);
void launch_generic(const f16* A, const f16* B, f16* C, int M, int N, int K, int lda, int ldb,
                    int ldc, hipStream_t stream);

}  // namespace hgemm_mi355x
