// M=1024 N=256 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x64_w4x2_m16_s4, split-K 8, raster group 2  [tuned on MI355X (round 6): 25.3 us, 339.0 TFLOP/s two-pass split-K (back to back 22.6 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 256, 16384, "t128x64_w4x2_m16_s4", 8, 2)
