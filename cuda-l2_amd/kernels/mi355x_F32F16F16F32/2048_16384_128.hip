// M=2048 N=16384 K=128  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x128_w2x2, split-K 1, non-temporal C stores, phase offset x4, raster group 4  [tuned on MI355X (round 5): 23.1 us, 371.2 TFLOP/s phase offset x4 (back to back 20.6 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(2048, 16384, 128, "q256x128_w2x2", 8519681, 4)
