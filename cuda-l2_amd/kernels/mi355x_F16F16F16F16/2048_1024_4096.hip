// M=2048 N=1024 K=4096  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry q128x128_w2x2_k128, split-K 2 (single launch), raster group 8  [tuned on MI355X (round 6): 28.5 us, 603.2 TFLOP/s fused split-K (back to back 25.9 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(2048, 1024, 4096, "q128x128_w2x2_k128", 65538, 8)
