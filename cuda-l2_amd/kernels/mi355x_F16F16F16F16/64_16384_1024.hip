// M=64 N=16384 K=1024  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t64x64_w2x2_m16_s4, split-K 1, non-temporal C stores, raster group 1  [tuned on MI355X (round 6): 10.6 us, 203.4 TFLOP/s (back to back 8.7 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 16384, 1024, "t64x64_w2x2_m16_s4", 131073, 1)
