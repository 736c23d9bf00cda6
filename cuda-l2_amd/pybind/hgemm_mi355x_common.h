// Torch-facing shim shared by pybind/hgemm_mi355x_{fp16,fp32}.cc: converts torch::Tensor arguments
// into the plain-pointer C ABI of libhgemm_mi355x.so (include/hgemm_mi355x.h).  This file is the
// ONLY place where torch types appear; it plays the role of the reference's pybind shims
// (pybind/hgemm_a100_fp32.cc:1-52) plus the tensor checks the reference keeps next to each kernel
// (cublas/fp32/hgemm_cublas.cu:170-231: "values must be torch::kHalf", "Tensor size mismatch!").
#pragma once

#include <c10/hip/HIPStream.h>
#include <torch/extension.h>
#include <torch/types.h>

#include <stdexcept>
#include <string>

#include "hgemm_mi355x.h"

#ifndef HGEMM_ACC_MODE
#error "define HGEMM_ACC_MODE (HGEMM_ACC_FP32 or HGEMM_ACC_FP16) before including this header"
#endif

#define STRINGFY(str) #str
#define TORCH_BINDING_COMMON_EXTENSION(func) m.def(STRINGFY(func), &func, STRINGFY(func));

#define CHECK_TORCH_TENSOR_DTYPE(T, th_type)                    \
  if (((T).options().dtype() != (th_type))) {                   \
    std::cout << "Tensor Info:" << (T).options() << std::endl;  \
    throw std::runtime_error("values must be " #th_type);       \
  }

#define CHECK_TORCH_TENSOR_SHAPE(T, S0, S1)                     \
  if (((T).size(0) != (S0)) || ((T).size(1) != (S1))) {         \
    throw std::runtime_error("Tensor size mismatch!");          \
  }

namespace hgemm_shim {

inline void* current_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

inline void check_status(int st, const char* what) {
  if (st != HGEMM_OK)
    throw std::runtime_error(std::string(what) + ": " + hgemm_mi355x_strerror(st));
}

// a: [M,K], second operand: [K,N]-shaped tensor (b or b_col_major), c: [M,N]; all fp16.
struct Problem {
  int M, N, K;
};
inline Problem check(const torch::Tensor& a, const torch::Tensor& b2, const torch::Tensor& c) {
  CHECK_TORCH_TENSOR_DTYPE(a, torch::kHalf)
  CHECK_TORCH_TENSOR_DTYPE(b2, torch::kHalf)
  CHECK_TORCH_TENSOR_DTYPE(c, torch::kHalf)
  const int M = a.size(0), K = a.size(1), N = b2.size(1);
  CHECK_TORCH_TENSOR_SHAPE(a, M, K)
  CHECK_TORCH_TENSOR_SHAPE(b2, K, N)
  CHECK_TORCH_TENSOR_SHAPE(c, M, N)
  return {M, N, K};
}

}  // namespace hgemm_shim

// ---- rocBLAS (reference names kept: the harness looks the functions up by these strings) ---------
void init_cublas_handle() { hgemm_shim::check_status(hgemm_rocblas_init(), "rocblas init"); }
void destroy_cublas_handle() { hgemm_rocblas_destroy(); }
void hgemm_cublas_nn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b, c);
  hgemm_shim::check_status(hgemm_rocblas_nn(a.data_ptr(), b.data_ptr(), c.data_ptr(), p.M, p.N, p.K, HGEMM_ACC_MODE,
                                            hgemm_shim::current_stream()), "rocblas nn");
}
void hgemm_cublas_tn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b_col_major, c);
  hgemm_shim::check_status(hgemm_rocblas_tn(a.data_ptr(), b_col_major.data_ptr(), c.data_ptr(), p.M, p.N, p.K,
                                            HGEMM_ACC_MODE, hgemm_shim::current_stream()), "rocblas tn");
}

// ---- hipBLASLt heuristic ---------------------------------------------------------------------------
void init_cublaslt_handle_v1() { hgemm_shim::check_status(hgemm_hipblaslt_heuristic_init(), "hipblaslt init"); }
void destroy_cublaslt_handle_v1() { hgemm_hipblaslt_heuristic_destroy(); }
void hgemm_cublaslt_heuristic_nn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b, c);
  hgemm_shim::check_status(hgemm_hipblaslt_heuristic_nn(a.data_ptr(), b.data_ptr(), c.data_ptr(), p.M, p.N, p.K,
                                                        HGEMM_ACC_MODE, hgemm_shim::current_stream()),
                           "hipblaslt heuristic nn");
}
void hgemm_cublaslt_heuristic_tn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b_col_major, c);
  hgemm_shim::check_status(hgemm_hipblaslt_heuristic_tn(a.data_ptr(), b_col_major.data_ptr(), c.data_ptr(), p.M,
                                                        p.N, p.K, HGEMM_ACC_MODE, hgemm_shim::current_stream()),
                           "hipblaslt heuristic tn");
}

// ---- hipBLASLt autotune ----------------------------------------------------------------------------
void init_cublaslt_handle_v2() { hgemm_shim::check_status(hgemm_hipblaslt_autotune_init(), "hipblaslt init"); }
void destroy_cublaslt_handle_v2() { hgemm_hipblaslt_autotune_destroy(); }
void find_best_algo_nn_v2_torch(int M, int N, int K) {
  hgemm_shim::check_status(hgemm_hipblaslt_autotune_find_best_nn(M, N, K, HGEMM_ACC_MODE), "[V2] No algorithm found for NN");
}
void find_best_algo_tn_v2_torch(int M, int N, int K) {
  hgemm_shim::check_status(hgemm_hipblaslt_autotune_find_best_tn(M, N, K, HGEMM_ACC_MODE), "[V2] No algorithm found for TN");
}
void hgemm_cublaslt_auto_tuning_nn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b, c);
  hgemm_shim::check_status(hgemm_hipblaslt_autotune_nn(a.data_ptr(), b.data_ptr(), c.data_ptr(), p.M, p.N, p.K,
                                                       HGEMM_ACC_MODE, hgemm_shim::current_stream()),
                           "hipblaslt autotune nn");
}
void hgemm_cublaslt_auto_tuning_tn(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {
  auto p = hgemm_shim::check(a, b_col_major, c);
  hgemm_shim::check_status(hgemm_hipblaslt_autotune_tn(a.data_ptr(), b_col_major.data_ptr(), c.data_ptr(), p.M,
                                                       p.N, p.K, HGEMM_ACC_MODE, hgemm_shim::current_stream()),
                           "hipblaslt autotune tn");
}

// ---- the per-shape kernel file provides this C symbol (kernels/mi355x_<acc>/<M>_<N>_<K>.hip) --------
extern "C" int cuda_l2_mi355x_shape_launch(const void* a, const void* b, const void* b_col_major, void* c, int M,
                                           int N, int K, void* stream);

#define HGEMM_DEFINE_CUDA_L2_ENTRY(name)                                                                      \
  void name(torch::Tensor a, torch::Tensor b, torch::Tensor b_col_major, torch::Tensor c) {                   \
    CHECK_TORCH_TENSOR_DTYPE(a, torch::kHalf)                                                                 \
    CHECK_TORCH_TENSOR_DTYPE(b, torch::kHalf)                                                                 \
    CHECK_TORCH_TENSOR_DTYPE(c, torch::kHalf)                                                                 \
    const int M = a.size(0), K = a.size(1), N = b.size(1);                                                    \
    hgemm_shim::check_status(cuda_l2_mi355x_shape_launch(a.data_ptr(), b.data_ptr(), b_col_major.data_ptr(),  \
                                                         c.data_ptr(), M, N, K, hgemm_shim::current_stream()), \
                             #name);                                                                          \
  }

#define HGEMM_BIND_ALL(m, cuda_l2_name)                      \
  TORCH_BINDING_COMMON_EXTENSION(init_cublas_handle)         \
  TORCH_BINDING_COMMON_EXTENSION(destroy_cublas_handle)      \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublas_nn)            \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublas_tn)            \
  TORCH_BINDING_COMMON_EXTENSION(init_cublaslt_handle_v1)    \
  TORCH_BINDING_COMMON_EXTENSION(destroy_cublaslt_handle_v1) \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublaslt_heuristic_nn) \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublaslt_heuristic_tn) \
  TORCH_BINDING_COMMON_EXTENSION(init_cublaslt_handle_v2)    \
  TORCH_BINDING_COMMON_EXTENSION(destroy_cublaslt_handle_v2) \
  TORCH_BINDING_COMMON_EXTENSION(find_best_algo_nn_v2_torch) \
  TORCH_BINDING_COMMON_EXTENSION(find_best_algo_tn_v2_torch) \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublaslt_auto_tuning_nn) \
  TORCH_BINDING_COMMON_EXTENSION(hgemm_cublaslt_auto_tuning_tn) \
  TORCH_BINDING_COMMON_EXTENSION(cuda_l2_name)
