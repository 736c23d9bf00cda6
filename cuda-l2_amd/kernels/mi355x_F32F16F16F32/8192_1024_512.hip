// M=8192 N=1024 K=512  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x256_w2x4_m16_s2, stream-K on 256 workgroups, raster group 8  [tuned on MI355X (round 4): 16.8 us, 512.5 TFLOP/s stream-K, 256 workgroups (back to back 13.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 1024, 512, "t128x256_w2x4_m16_s2", 262400, 8)
