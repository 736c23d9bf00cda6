// M=256 N=4096 K=8192  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 4, K stagger per XCD, raster group 4  [tuned on MI355X (round 6): 33.8 us, 508.0 TFLOP/s two-pass split-K, K stagger per XCD (back to back 31.9 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 4096, 8192, "q128x128_w2x2_k128", 524292, 4)
