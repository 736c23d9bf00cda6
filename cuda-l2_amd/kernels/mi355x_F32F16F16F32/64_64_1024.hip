// M=64 N=64 K=1024  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry w16x16_k4, split-K 1, raster group 2  [tuned on MI355X (round 6): 6.3 us, 1.3 TFLOP/s (back to back 3.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 64, 1024, "w16x16_k4", 1, 2)
