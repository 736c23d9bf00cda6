// M=128 N=1024 K=16384  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry t64x64_w2x2_m16_s4, split-K 8 (single launch), raster group 2  [tuned on MI355X (round 6): 18.9 us, 226.8 TFLOP/s fused split-K (back to back 16.8 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(128, 1024, 16384, "t64x64_w2x2_m16_s4", 65544, 2)
