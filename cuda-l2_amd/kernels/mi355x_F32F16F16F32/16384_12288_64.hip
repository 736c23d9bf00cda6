// M=16384 N=12288 K=64  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, raster group 4  [tuned on MI355X (round 6): 83.2 us, 309.8 TFLOP/s (back to back 82.6 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 12288, 64, "q256x256_w2x2", 1, 4)
