// M=256 N=8192 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 2 (single launch), K stagger per XCD, raster group 8  [tuned on MI355X (round 6): 88.4 us, 777.5 TFLOP/s fused split-K, K stagger per XCD (back to back 86.3 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 8192, 16384, "q128x128_w2x2_k128", 589826, 8)
