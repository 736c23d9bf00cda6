// M=64 N=16384 K=8192  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry r64x128_k128_d, split-K 2 (single launch), raster group 1  [tuned on MI355X (round 4): 50.0 us, 343.6 TFLOP/s fused split-K (back to back 48.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 16384, 8192, "r64x128_k128_d", 1114114, 1)
