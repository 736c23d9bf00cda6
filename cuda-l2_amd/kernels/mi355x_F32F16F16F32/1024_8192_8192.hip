// M=1024 N=8192 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry p256x256_w4x2_v1, split-K 2, raster group 4  [tuned on MI355X: 131.6 us, 1045 TFLOP/s]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 8192, 8192, "p256x256_w4x2_v1", 2, 4)
