// M=512 N=2048 K=4096  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2, split-K 4, raster group 2  [tuned on MI355X (round 6): 24.6 us, 349.8 TFLOP/s two-pass split-K (back to back 19.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(512, 2048, 4096, "q128x128_w2x2", 4, 2)
