// lcp_wave_scene.h - helpers of the kernels that give ONE wavefront the vector role of a scene (lane c = contact c with its
// four inequality components, lane j = x-space entry j): lcp_big.hip and lcp_primal.hip.
#pragma once
#include "lcp_wave_common.h"

namespace lcp {
namespace wsc {

using namespace w64;

template <typename TC> struct M4 { TC n, f1, f2, g; };
template <typename TC> __device__ __forceinline__ M4<TC> m4(TC a, TC b, TC c, TC d) { M4<TC> r; r.n = a; r.f1 = b; r.f2 = c; r.g = d; return r; }

// ---------------------------------------------------------------- wave-0 helpers
// LDS traffic inside ONE wave needs no barrier (the LDS serves a wave's instructions in order); the fence only stops the
// compiler from moving the accesses across it.
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Wave-wide reductions, every lane receives the result: four DPP steps inside each 16-lane row (quad_perm, row_half_mirror,
// row_mirror - as lcp_quad.hip), then the four row results through v_readlane.  (The butterfly of __shfl_xor they replace
// was six dependent ds_bpermute round trips per value: 12 k cycles per PDIPM iteration in the step-length code.)
template <int CTRL> __device__ __forceinline__ double dppx(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_lane(double v, int src) {       // src uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dppx<0xB1>(v); v += dppx<0x4E>(v); v += dppx<0x141>(v); v += dppx<0x140>(v);
  return (row_lane(v, 0) + row_lane(v, 16)) + (row_lane(v, 32) + row_lane(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {        // NaN-ignoring
  v = __builtin_fmax(v, dppx<0xB1>(v)); v = __builtin_fmax(v, dppx<0x4E>(v)); v = __builtin_fmax(v, dppx<0x141>(v)); v = __builtin_fmax(v, dppx<0x140>(v));
  return __builtin_fmax(__builtin_fmax(row_lane(v, 0), row_lane(v, 16)), __builtin_fmax(row_lane(v, 32), row_lane(v, 48)));
}
__device__ __forceinline__ double wave_min(double v) {
  v = __builtin_fmin(v, dppx<0xB1>(v)); v = __builtin_fmin(v, dppx<0x4E>(v)); v = __builtin_fmin(v, dppx<0x141>(v)); v = __builtin_fmin(v, dppx<0x140>(v));
  return __builtin_fmin(__builtin_fmin(row_lane(v, 0), row_lane(v, 16)), __builtin_fmin(row_lane(v, 32), row_lane(v, 48)));
}
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) {
  auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));
  v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
  const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
  return mx(mx(r0, r1), mx(r2, r3));
}
__device__ __forceinline__ uint32_t nan_key(double v) { return (uint32_t)__double2hiint(v) & 0x7fffffffu; }
__device__ __forceinline__ bool key_is_nan(uint32_t k) { return k > 0x7ff00000u; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ double bcast_lane(double v, int src) {     // src uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

}  // namespace wsc
}  // namespace lcp
