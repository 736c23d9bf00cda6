// M=12288 N=2048 K=512  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q192x256_w2x2, split-K 1, non-temporal C stores, K stagger per XCD, phase offset, raster group 8  [tuned on MI355X (round 5): 33.2 us, 775.7 TFLOP/s K stagger per XCD, phase offset (back to back 29.3 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 2048, 512, "q192x256_w2x2", 2752513, 8)
