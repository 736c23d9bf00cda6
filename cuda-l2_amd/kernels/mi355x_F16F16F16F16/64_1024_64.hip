// M=64 N=1024 K=64  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry p128x128_w2x4_v1, split-K 1, raster group 16  [tuned on MI355X: 6.3 us, 1 TFLOP/s]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 1024, 64, "p128x128_w2x4_v1", 1, 16)
