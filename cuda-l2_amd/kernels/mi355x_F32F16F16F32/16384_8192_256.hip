// M=16384 N=8192 K=256  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, non-temporal C stores, phase offset x8, raster group 8  [tuned on MI355X (round 6): 89.1 us, 771.1 TFLOP/s phase offset x8 (back to back 88.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(16384, 8192, 256, "q256x256_w2x2", 10616833, 8)
