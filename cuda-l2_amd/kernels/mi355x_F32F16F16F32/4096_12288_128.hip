// M=4096 N=12288 K=128  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x256_w2x2, split-K 1, non-temporal C stores, phase offset, raster group 4  [tuned on MI355X (round 6): 29.8 us, 433.0 TFLOP/s phase offset (back to back 27.2 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(4096, 12288, 128, "q256x256_w2x2", 2228225, 4)
