// M=64 N=12288 K=12288  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry r64x96_k128, split-K 2 (single launch), raster group 1  [tuned on MI355X (round 4): 59.3 us, 326.1 TFLOP/s fused split-K (back to back 57.4 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 12288, 12288, "r64x96_k128", 1114114, 1)
