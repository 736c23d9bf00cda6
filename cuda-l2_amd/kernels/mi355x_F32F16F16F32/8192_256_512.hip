// M=8192 N=256 K=512  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x64_w4x2_m16_s3, split-K 1, raster group 4  [tuned on MI355X (round 6): 8.4 us, 254.4 TFLOP/s (back to back 6.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 256, 512, "t128x64_w4x2_m16_s3", 1, 4)
