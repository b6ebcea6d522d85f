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

// get_step (pdipm.py:182-186) from t_i = dv_i (1 / v_i) instead of the quotients a_i = -v_i / dv_i = -1 / t_i - the EXACT form, every case
// (tests/test_step_length_model.py holds the same algebra against the oracle's get_step on adversarial vectors):
//   an entry the fill replaces (dv_i > 0) has t_i > 0; the fill max(1, a.max()) is never below an entry it does not replace, so with a
//   decreasing entry (t_i < 0) the result is -1 / min_i t_i; an exact +0 among the dv_i is a_i = -inf (the reference's solve then dies on
//   NaN iterates), -0 is a_i = +inf; without a decreasing entry the result is the fill itself: 1, or +inf beside a -0; NaN propagates.
// step_flags: the five facts about one t_i; OR them over the vector (and the scene), take the NaN-ignoring minimum of the t_i beside them.
__device__ __forceinline__ uint32_t step_flags(double t) {
  return (__builtin_amdgcn_class(t, 0x3) ? 1u : 0u) | (__builtin_amdgcn_class(t, 0x40) ? 2u : 0u) | (__builtin_amdgcn_class(t, 0x20) ? 4u : 0u) |
         ((t < 0.0) ? 8u : 0u) | ((t > 0.0) ? 16u : 0u);
}
__device__ __forceinline__ double step_from_flags(uint32_t f, double tmin) {
  return (f & 1u) ? nan_of<double>() : ((f & 2u) ? -inf_of<double>() : ((f & 8u) ? -1.0 / tmin : ((f & 16u) ? ((f & 4u) ? inf_of<double>() : 1.0) : inf_of<double>())));
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

// ------------------------------------------------------------------------------------------
// One contact -> its rows of Jc / Jf, h entry and friction coefficient (physics/world.py:144-224,
// physics/engines.py:53), formed in I/O precision with FMA contraction OFF so that every kernel
// that assembles a scene (the stand-alone assembly kernel and both fused step kernels) produces
// bit-identical LCP data.
// ------------------------------------------------------------------------------------------
template <typename TI>
struct ContactRows {
  TI jn[6];   // normal row:   cols 3*b1..3*b1+2 then 3*b2..3*b2+2          (world.py:177-183)
  TI jf[6];   // friction row for direction 1 (the direction-2 row is its negative, world.py:191-210)
  TI h;       // (Jc v)_c * restitution_c                                     (engines.py:53, world.py:144-151)
  TI jv, rbar;  // (Jc v)_c and restitution_c by themselves (post-stabilisation: gc = jv + jv * -rbar, engines.py:87-89)
  TI mu;      // 0.5 (fric_b1 + fric_b2)                                       (world.py:213-224)
  int b1, b2;
};

// (the record already in registers: kernels that ask for it in their prologue, together with their other loads)
template <typename TI>
__device__ __forceinline__ ContactRows<TI> make_contact_rec(TI nx, TI ny, TI p1x, TI p1y, TI p2x, TI p2y, int b1, int b2,
                                                            const TI* rest, const TI* fric, const TI* vv) {
#pragma clang fp contract(off)
  ContactRows<TI> r;
  r.b1 = b1; r.b2 = b2;
  const TI tx = ny, ty = -nx;                                      // left_orthogonal, utils.py:99-102
  r.jn[0] = p1x * ny - p1y * nx; r.jn[1] = nx; r.jn[2] = ny;       // cross_2d, utils.py:93-96
  r.jn[3] = -(p2x * ny - p2y * nx); r.jn[4] = -nx; r.jn[5] = -ny;
  r.jf[0] = p1x * ty - p1y * tx; r.jf[1] = tx; r.jf[2] = ty;
  r.jf[3] = -(p2x * ty - p2y * tx); r.jf[4] = -tx; r.jf[5] = -ty;
  TI acc = (TI)0;                                                  // columns in ascending order
  const int lo = r.b1 < r.b2 ? 0 : 3, hi = 3 - lo;
  const int blo = r.b1 < r.b2 ? r.b1 : r.b2, bhi = r.b1 < r.b2 ? r.b2 : r.b1;
  for (int q = 0; q < 3; ++q) acc = acc + r.jn[lo + q] * vv[3 * blo + q];
  for (int q = 0; q < 3; ++q) acc = acc + r.jn[hi + q] * vv[3 * bhi + q];
  r.jv = acc;
  r.rbar = (TI)0.5 * (rest[r.b1] + rest[r.b2]);
  r.h = acc * r.rbar;
  r.mu = (TI)0.5 * (fric[r.b1] + fric[r.b2]);
  return r;
}

template <typename TI>
__device__ __forceinline__ ContactRows<TI> make_contact(const TI* cn, const TI* c1, const TI* c2, const int32_t* i1,
                                                        const int32_t* i2, const TI* rest, const TI* fric,
                                                        const TI* vv, int c) {
  return make_contact_rec<TI>(cn[2 * c], cn[2 * c + 1], c1[2 * c], c1[2 * c + 1], c2[2 * c], c2[2 * c + 1], i1[c], i2[c], rest, fric, vv);
}

// u = M v + dt f (engines.py:32) in I/O precision, contraction off.
template <typename TI>
__device__ __forceinline__ TI momentum_entry(TI md, TI v, TI dt, TI f) {
#pragma clang fp contract(off)
  return md * v + dt * f;
}

}  // namespace lcp
