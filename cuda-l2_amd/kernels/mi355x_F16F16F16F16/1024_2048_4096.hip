// M=1024 N=2048 K=4096  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 2 (single launch), K stagger per XCD, raster group 4  [tuned on MI355X (round 6): 29.2 us, 587.5 TFLOP/s fused split-K, K stagger per XCD (back to back 25.7 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(1024, 2048, 4096, "q128x128_w2x2_k128", 589826, 4)
