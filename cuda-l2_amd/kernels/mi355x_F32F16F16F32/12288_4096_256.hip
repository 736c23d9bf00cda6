// M=12288 N=4096 K=256  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, non-temporal C stores, phase offset, raster group 8  [tuned on MI355X (round 6): 39.4 us, 654.4 TFLOP/s phase offset (back to back 35.3 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 4096, 256, "q256x256_w2x2", 2228225, 8)
