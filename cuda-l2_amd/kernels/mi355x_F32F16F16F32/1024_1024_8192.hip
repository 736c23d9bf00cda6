// M=1024 N=1024 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 4, K stagger per XCD, raster group 32  [tuned on MI355X (round 6): 31.1 us, 551.7 TFLOP/s two-pass split-K, K stagger per XCD (back to back 28.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 1024, 8192, "q128x128_w2x2_k128", 524292, 32)
