// M=8192 N=64 K=16384  F16F16F16F16 (fp16 in, fp32 MFMA accumulate [no fp16-accumulate MFMA on CDNA4], fp16 out)  MI355X / gfx950
// plan: geometry r128x64_k128, split-K 8 (single launch), raster group 2  [tuned on MI355X (round 4): 57.0 us, 301.3 TFLOP/s fused split-K (back to back 54.5 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp16
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(8192, 64, 16384, "r128x64_k128", 1114120, 2)
