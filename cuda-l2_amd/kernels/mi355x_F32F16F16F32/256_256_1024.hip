// M=256 N=256 K=1024  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x32_w2x1_m16_s4, split-K 4 (single launch), raster group 1  [tuned on MI355X: 7.3 us, 18 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 256, 1024, "t64x32_w2x1_m16_s4", 65540, 1)
