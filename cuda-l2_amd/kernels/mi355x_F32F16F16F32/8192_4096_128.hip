// M=8192 N=4096 K=128  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x128_w2x2, split-K 1, non-temporal C stores, raster group 2  [tuned on MI355X (round 6): 22.0 us, 389.7 TFLOP/s (back to back 19.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 4096, 128, "q256x128_w2x2", 131073, 2)
