// M=12288 N=128 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x128_w2x4_m16_s4, stream-K on 256 workgroups, raster group 4  [tuned on MI355X (round 4): 55.9 us, 460.8 TFLOP/s stream-K, 256 workgroups (back to back 53.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 128, 8192, "t128x128_w2x4_m16_s4", 262400, 4)
