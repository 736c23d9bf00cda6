// M=16384 N=12288 K=16384  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, raster group 2  [tuned on MI355X (round 6): 4499.1 us, 1466.3 TFLOP/s (back to back 4519.4 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 12288, 16384, "q256x256_w2x2", 1, 2)
