// M=8192 N=64 K=8192  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry r128x64_k128, split-K 4 (single launch), raster group 2  [tuned on MI355X (round 4): 35.1 us, 244.4 TFLOP/s fused split-K (back to back 31.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 64, 8192, "r128x64_k128", 1114116, 2)
