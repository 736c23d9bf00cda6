// M=2048 N=12288 K=4096  F32F16F16F32 (fp16 in, fp32 MFMA accumulate, fp16 out)  MI355X / gfx950
// plan: geometry q256x192_w2x2, split-K 1, non-temporal C stores, K stagger per XCD, raster group 8  [tuned on MI355X (round 6): 151.9 us, 1357.5 TFLOP/s K stagger per XCD (back to back 154.3 us), verified against the CPU oracle]
// kernels: csrc/hgemm_kernel*.hpp (instantiated in libhgemm_mi355x.so); geometry table: csrc/hgemm_configs.def
#define HGEMM_SHAPE_FALLBACK hgemm_mi355x_fp32
#include "hgemm_shape_entry.hpp"
HGEMM_MI355X_SHAPE_ENTRY(2048, 12288, 4096, "q256x192_w2x2", 655361, 8)
