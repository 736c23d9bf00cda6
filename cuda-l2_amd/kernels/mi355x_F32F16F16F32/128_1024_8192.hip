// M=128 N=1024 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x64_w2x2_m16_s4, split-K 8, raster group 32  [tuned on MI355X: 13.5 us, 159 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 1024, 8192, "t64x64_w2x2_m16_s4", 8, 32)
