// M=1024 N=4096 K=256  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t128x128_w2x4_m16_s4, split-K 1, non-temporal C stores, raster group 16  [tuned on MI355X: 7.5 us, 288 TFLOP/s, verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 4096, 256, "t128x128_w2x4_m16_s4", 131073, 16)
