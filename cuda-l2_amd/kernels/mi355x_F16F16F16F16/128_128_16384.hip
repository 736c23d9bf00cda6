// M=128 N=128 K=16384  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t32x64_w1x2_m16_s4, split-K 32, raster group 2  [tuned on MI355X: 11.2 us, 48 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 128, 16384, "t32x64_w1x2_m16_s4", 32, 2)
