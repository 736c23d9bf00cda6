// M=64 N=2048 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry r64x64_k256, split-K 8 (single launch), raster group 1  [tuned on MI355X (round 4): 23.3 us, 184.5 TFLOP/s fused split-K (back to back 20.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 2048, 16384, "r64x64_k256", 1114120, 1)
