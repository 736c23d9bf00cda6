// M=256 N=8192 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x256_w2x4_m32_s3, stream-K on 256 workgroups, raster group 2  [tuned on MI355X (round 4): 100.2 us, 685.5 TFLOP/s stream-K, 256 workgroups (back to back 98.2 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 8192, 16384, "t128x256_w2x4_m32_s3", 262400, 2)
