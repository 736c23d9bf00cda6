// M=64 N=16384 K=4096  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry r64x64_k256_d, split-K 1, raster group 1  [tuned on MI355X (round 4): 30.1 us, 285.2 TFLOP/s (back to back 27.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 16384, 4096, "r64x64_k256_d", 1572865, 1)
