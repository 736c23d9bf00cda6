// Python extension `hgemm_lib` for --device_type mi355x --acc_precise fp16 (F16F16F16F16 tree).
// Same 15 names as pybind/hgemm_a100_fp16.cc:29-52 with cuda_l2_mi355x_fp16.  The baselines request
// fp16 compute (rocBLAS f16_r, hipBLASLt COMPUTE_16F with fp32 fall-back); the cuda_l2 entry uses the
// fp32-accumulating MFMA (CDNA4 has no fp16-accumulating MFMA), see include/hgemm_mi355x.h.
#define HGEMM_ACC_MODE HGEMM_ACC_FP16
#include "hgemm_mi355x_common.h"

HGEMM_DEFINE_CUDA_L2_ENTRY(cuda_l2_mi355x_fp16)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { HGEMM_BIND_ALL(m, cuda_l2_mi355x_fp16) }
