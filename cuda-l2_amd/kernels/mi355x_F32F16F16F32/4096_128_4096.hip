// M=4096 N=128 K=4096  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x64_w2x2_m16_s4, split-K 2 (single launch), raster group 4  [tuned on MI355X (round 6): 17.2 us, 250.0 TFLOP/s fused split-K (back to back 15.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(4096, 128, 4096, "t64x64_w2x2_m16_s4", 65538, 4)
