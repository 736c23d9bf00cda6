// M=1024 N=64 K=512  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry w16x16_k4, split-K 1, raster group 8  [tuned on MI355X (round 6): 6.4 us, 10.6 TFLOP/s (back to back 2.9 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 64, 512, "w16x16_k4", 1, 8)
