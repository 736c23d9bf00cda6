// M=256 N=128 K=8192  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry w32x32_k4, split-K 8 (single launch), raster group 2  [tuned on MI355X (round 6): 11.8 us, 45.7 TFLOP/s fused split-K (back to back 9.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 128, 8192, "w32x32_k4", 65544, 2)
