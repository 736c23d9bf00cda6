// M=256 N=16384 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x128_w2x2, split-K 2 (single launch), K stagger per XCD, raster group 8  [tuned on MI355X (round 6): 148.0 us, 928.8 TFLOP/s fused split-K, K stagger per XCD (back to back 147.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 16384, 16384, "q256x128_w2x2", 589826, 8)
