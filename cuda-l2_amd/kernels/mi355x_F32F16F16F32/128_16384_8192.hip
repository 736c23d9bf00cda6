// M=128 N=16384 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry r128x128_k128_d, split-K 2 (single launch), raster group 1  [tuned on MI355X (round 4): 64.0 us, 536.7 TFLOP/s fused split-K (back to back 61.7 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 16384, 8192, "r128x128_k128_d", 589826, 1)
