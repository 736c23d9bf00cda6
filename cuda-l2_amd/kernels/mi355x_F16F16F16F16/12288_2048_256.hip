// M=12288 N=2048 K=256  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x256_w2x2, split-K 1, non-temporal C stores, phase offset, raster group 2  [tuned on MI355X (round 6): 23.1 us, 557.8 TFLOP/s phase offset (back to back 20.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 2048, 256, "q128x256_w2x2", 2228225, 2)
