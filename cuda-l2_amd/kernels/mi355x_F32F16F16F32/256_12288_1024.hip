// M=256 N=12288 K=1024  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 1, non-temporal C stores, raster group 2  [tuned on MI355X (round 6): 14.5 us, 443.1 TFLOP/s (back to back 10.9 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(256, 12288, 1024, "q128x128_w2x2_k128", 131073, 2)
