// M=128 N=256 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry w32x32_k4, split-K 8 (single launch), raster group 2  [tuned on MI355X (round 6): 11.6 us, 46.1 TFLOP/s fused split-K (back to back 9.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 256, 8192, "w32x32_k4", 65544, 2)
