// M=256 N=12288 K=12288  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x192_w2x2, split-K 4, K stagger per XCD, raster group 2  [tuned on MI355X (round 6): 97.9 us, 790.0 TFLOP/s two-pass split-K, K stagger per XCD (back to back 94.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 12288, 12288, "q256x192_w2x2", 524292, 2)
