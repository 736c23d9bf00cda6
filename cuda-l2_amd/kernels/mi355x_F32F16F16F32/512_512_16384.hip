// M=512 N=512 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x128_w2x4_m16_s4, split-K 8, raster group 4  [tuned on MI355X (round 6): 23.9 us, 359.1 TFLOP/s two-pass split-K (back to back 21.2 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(512, 512, 16384, "t64x128_w2x4_m16_s4", 8, 4)
