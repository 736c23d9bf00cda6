// Python extension `hgemm_lib` for --device_type mi355x --acc_precise fp32 (F32F16F16F32 tree).
// Exports the reference's 15 names (pybind/hgemm_a100_fp32.cc:29-52) with cuda_l2_mi355x_fp32 in
// place of cuda_l2_a100_fp32; the baseline names keep their cuBLAS spelling so the reference
// harness files work unchanged, but they run rocBLAS / hipBLASLt with fp32 compute.
#define HGEMM_ACC_MODE HGEMM_ACC_FP32
#include "hgemm_mi355x_common.h"

HGEMM_DEFINE_CUDA_L2_ENTRY(cuda_l2_mi355x_fp32)

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { HGEMM_BIND_ALL(m, cuda_l2_mi355x_fp32) }
