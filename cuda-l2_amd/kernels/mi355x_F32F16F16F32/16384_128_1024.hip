// M=16384 N=128 K=1024  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry t64x128_w2x4_m16_s4, split-K 1, non-temporal C stores, raster group 8  [tuned on MI355X (round 6): 12.8 us, 334.5 TFLOP/s (back to back 10.9 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 128, 1024, "t64x128_w2x4_m16_s4", 131073, 8)
