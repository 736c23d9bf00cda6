// M=64 N=16384 K=2048  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry r64x64_k256, split-K 1, raster group 1  [tuned on MI355X (round 4): 18.9 us, 227.0 TFLOP/s (back to back 16.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 16384, 2048, "r64x64_k256", 1572865, 1)
