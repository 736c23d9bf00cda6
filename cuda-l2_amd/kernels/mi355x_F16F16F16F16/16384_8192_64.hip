// M=16384 N=8192 K=64  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q256x128_w2x2, split-K 1, non-temporal C stores, phase offset, raster group 2  [tuned on MI355X (round 6): 56.9 us, 302.0 TFLOP/s phase offset (back to back 55.0 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 8192, 64, "q256x128_w2x2", 2228225, 2)
