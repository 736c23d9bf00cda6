// M=12288 N=12288 K=128  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x128_w2x2_m16_s2, stream-K on 512 workgroups, raster group 8  [tuned on MI355X (round 4): 92.3 us, 418.7 TFLOP/s stream-K, 512 workgroups (back to back 91.7 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 12288, 128, "t128x128_w2x2_m16_s2", 262656, 8)
