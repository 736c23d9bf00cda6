// M=16384 N=256 K=4096  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 1, non-temporal C stores, K stagger per XCD, raster group 4  [tuned on MI355X (round 6): 43.6 us, 788.1 TFLOP/s K stagger per XCD (back to back 42.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 256, 4096, "q128x128_w2x2_k128", 655361, 4)
