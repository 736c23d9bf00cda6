// M=128 N=12288 K=64  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t64x32_w2x1_m16_s4, split-K 1, raster group 2  [tuned on MI355X (round 6): 6.7 us, 30.0 TFLOP/s (back to back 3.6 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 12288, 64, "t64x32_w2x1_m16_s4", 1, 2)
