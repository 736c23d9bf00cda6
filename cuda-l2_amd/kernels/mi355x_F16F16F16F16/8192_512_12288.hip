// M=8192 N=512 K=12288  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 4, raster group 8  [tuned on MI355X (round 6): 95.6 us, 1078.7 TFLOP/s two-pass split-K (back to back 96.6 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 512, 12288, "q256x256_w2x2", 4, 8)
