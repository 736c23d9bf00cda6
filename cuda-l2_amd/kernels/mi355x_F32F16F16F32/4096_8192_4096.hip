// M=4096 N=8192 K=4096  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, non-temporal C stores, K stagger per XCD, raster group 8  [tuned on MI355X (round 6): 189.9 us, 1447.8 TFLOP/s K stagger per XCD (back to back 191.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(4096, 8192, 4096, "q256x256_w2x2", 655361, 8)
