// M=1024 N=16384 K=64  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t256x256_w2x4_m16_s2, split-K 1, non-temporal C stores, raster group 2  [tuned on MI355X: 11.5 us, 187 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 16384, 64, "t256x256_w2x4_m16_s2", 131073, 2)
