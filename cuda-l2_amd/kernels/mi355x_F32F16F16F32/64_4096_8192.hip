// M=64 N=4096 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x64_w2x2_m16_s4, split-K 8 (single launch), raster group 1  [tuned on MI355X (round 6): 22.1 us, 194.0 TFLOP/s fused split-K (back to back 19.2 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 4096, 8192, "t64x64_w2x2_m16_s4", 65544, 1)
