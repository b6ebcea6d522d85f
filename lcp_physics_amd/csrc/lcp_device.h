// lcp_device.h - small device-side helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcp_kernels.h"

namespace lcp {

template <typename T> __device__ __forceinline__ T nan_of();
template <> __device__ __forceinline__ float nan_of<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double nan_of<double>() { return __builtin_nan(""); }
template <typename T> __device__ __forceinline__ T inf_of();
template <> __device__ __forceinline__ float inf_of<float>() { return __builtin_huge_valf(); }
template <> __device__ __forceinline__ double inf_of<double>() { return __builtin_huge_val(); }

// `mu > 1e100` (pdipm.py:133): in fp32 only +inf compares greater than 1e100.
template <typename T> __device__ __forceinline__ T mu_limit();
template <> __device__ __forceinline__ float mu_limit<float>() { return 3.402823466e+38f; }
template <> __device__ __forceinline__ double mu_limit<double>() { return 1e100; }

// NaN-propagating min / max (Tensor.min()/max() and torch.min(a,b) semantics).
template <typename T> __device__ __forceinline__ T pmin(T a, T b) {
  return (a != a || b != b) ? nan_of<T>() : (a < b ? a : b);
}
template <typename T> __device__ __forceinline__ T pmax(T a, T b) {
  return (a != a || b != b) ? nan_of<T>() : (a > b ? a : b);
}

struct OpSum { template <typename T> __device__ T operator()(T a, T b) const { return a + b; } };
struct OpMin { template <typename T> __device__ T operator()(T a, T b) const { return pmin(a, b); } };
struct OpMax { template <typename T> __device__ T operator()(T a, T b) const { return pmax(a, b); } };

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int off);
template <> __device__ __forceinline__ float shfl_xor_t<float>(float v, int off) { return __shfl_xor(v, off, 64); }
template <> __device__ __forceinline__ double shfl_xor_t<double>(double v, int off) { return __shfl_xor(v, off, 64); }
template <typename T> __device__ __forceinline__ T shfl_t(T v, int src);
template <> __device__ __forceinline__ float shfl_t<float>(float v, int src) { return __shfl(v, src, 64); }
template <> __device__ __forceinline__ double shfl_t<double>(double v, int src) { return __shfl(v, src, 64); }

// In-place Gauss-Jordan inverse of an n x n matrix without pivoting (SPD inputs: Q, A Q^-1 A^T).
// Returns false (to all threads) if a zero / NaN pivot was met.
template <int NT, typename TC>
__device__ bool gj_inverse(TC* a, int n, int* flag) {
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const TC piv = a[k * n + k];
    if (!(piv != (TC)0) || piv != piv) { if (threadIdx.x == 0) *flag = 1; }
    const TC pinv = (TC)1 / piv;
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += NT) if (j != k) a[k * n + j] *= pinv;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * n; idx += NT) {
      const int i = idx / n, j = idx - i * n;
      if (i != k && j != k) a[idx] -= a[i * n + k] * a[k * n + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) a[i * n + k] = (i == k) ? pinv : -a[i * n + k] * pinv;
    __syncthreads();
  }
  return *flag == 0;
}

}  // namespace lcp
