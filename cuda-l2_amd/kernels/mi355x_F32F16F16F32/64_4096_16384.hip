// M=64 N=4096 K=16384  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry r64x64_k256_d, split-K 4 (single launch), NT loads of the streamed operand, raster group 1  [tuned on MI355X (round 6): 32.6 us, 263.8 TFLOP/s fused split-K, NT loads of the streamed operand (back to back 30.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(64, 4096, 16384, "r64x64_k256_d", 1114116, 1)
