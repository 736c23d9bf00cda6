// M=12288 N=64 K=8192  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry r96x64_k128, split-K 2 (single launch), raster group 4  [tuned on MI355X (round 4): 44.1 us, 292.3 TFLOP/s fused split-K (back to back 41.1 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(12288, 64, 8192, "r96x64_k128", 1638402, 4)
