// M=2048 N=2048 K=256  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t128x128_w2x4_m16_s4, split-K 1, non-temporal C stores, raster group 4  [tuned on MI355X: 7.5 us, 287 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(2048, 2048, 256, "t128x128_w2x4_m16_s4", 131073, 4)
