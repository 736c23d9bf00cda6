// M=8192 N=512 K=1024  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2, split-K 1, non-temporal C stores, raster group 8  [tuned on MI355X (round 6): 14.8 us, 582.0 TFLOP/s (back to back 12.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 512, 1024, "q128x128_w2x2", 131073, 8)
