// M=512 N=2048 K=12288  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x128_w2x2, split-K 8, raster group 2  [tuned on MI355X (round 6): 40.8 us, 631.3 TFLOP/s two-pass split-K (back to back 38.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(512, 2048, 12288, "q256x128_w2x2", 8, 2)
