// M=2048 N=2048 K=64  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x128_w2x2_m16_s2, split-K 1, raster group 16  [tuned on MI355X: 5.8 us, 93 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(2048, 2048, 64, "t64x128_w2x2_m16_s2", 1, 16)
