// M=256 N=16384 K=4096  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 1, non-temporal C stores, K stagger per XCD, raster group 8  [tuned on MI355X (round 6): 46.6 us, 737.6 TFLOP/s K stagger per XCD (back to back 42.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 16384, 4096, "q128x128_w2x2_k128", 655361, 8)
