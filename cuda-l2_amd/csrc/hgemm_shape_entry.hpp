// Glue for the per-shape kernel files kernels/mi355x_<acc>/<M>_<N>_<K>.hip (the analogue of the
// reference's one-file-per-shape kernels/<dev>_<acc>/<M>_<N>_<K>.cu).  A shape file pins the plan
// that was tuned for its (M,N,K) -- kernel geometry from csrc/hgemm_configs.def, split-K factor,
// raster group -- and exports the C symbol the torch shim calls.  The kernels themselves are
// instantiated once inside libhgemm_mi355x.so, so building a shape's extension compiles only this
// few-line translation unit (the reference re-JITs ~300 lines of CuTe per shape).
//
// All geometries predicate their M/N edges in-kernel, so no harness-side zero padding is needed and
// shape files deliberately contain nothing the reference's tile-size regex (tools/utils.py:8-36)
// could match: the harness then computes padding 0.
#pragma once

#include <stddef.h>

#include "hgemm_mi355x.h"

#ifndef HGEMM_SHAPE_FALLBACK
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#endif

#define HGEMM_MI355X_SHAPE_ENTRY(M_, N_, K_, CONFIG_NAME, SPLITS, GROUP_M)                              \
  extern "C" int cuda_l2_mi355x_shape_launch(const void* a, const void* b, const void* b_col_major,     \
                                             void* c, int M, int N, int K, void* stream) {              \
    static const int cfg = hgemm_mi355x_config_by_name(CONFIG_NAME);                                    \
    /* first-use selection on (HGEMM_MI355X_INSITU=1 / eval_one_file.sh --insitu): the library entry times this plan and */    \
    /* its alternates on the first call and keeps the winner; off (default): the pinned plan, one table-free launch      */    \
    if (M == (M_) && N == (N_) && K == (K_) && cfg >= 0 && !hgemm_mi355x_insitu_enabled())              \
      return hgemm_mi355x_launch(cfg, (SPLITS), (GROUP_M), a, b, b_col_major, c, M, N, K, K, K, N,      \
                                 stream);                                                               \
    /* tensors of another size (or a retired geometry name): let the library plan it */                 \
    return HGEMM_SHAPE_FALLBACK(a, b, b_col_major, c, M, N, K, stream);                                    \
  }
