// lcp_wave_common.h - helpers shared by the wave-level kernel files (lcp_wave64.hip, lcp_quad.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "lcp_device.h"

namespace lcp {

// Scenes a launch of the four-scenes-per-wavefront kernels may hold while every wavefront still has a SIMD of its own: 4 scenes x 4 SIMDs
// per compute unit of the device that is CURRENT on the calling thread (the launch goes to it).  Read from the device - 1024 SIMDs =
// 4096 scenes on an MI355X - and kept per host thread and device ordinal (no process-global state; the attribute never changes).
inline int one_wave_per_simd_scenes() {
  thread_local int cached_dev = -1, cached = 4096;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 4096;
  if (dev != cached_dev) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) cached = 16 * cus;
    else cached = 4096;
    cached_dev = dev;
  }
  return cached;
}

namespace w64 {

constexpr int MP = 64;    // padded nineq (lanes)
constexpr int NZP = 16;   // padded nz
constexpr int EP = 8;     // padded neq
constexpr int NR = 32;    // reduced system size (2 nc <= 32)

// Compile-time loops: every index into the register-resident row t[] must be a constant the front end
// can see (hipcc demotes the array to scratch otherwise - measured), so the unrolling is done with
// templates rather than `#pragma unroll`.
template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}
#define LCP_INL __attribute__((always_inline))

// 1/x by v_rcp + Newton steps (the IEEE division sequence is ~12 dependent instructions and sat on the
// critical path of every elimination step).  Two steps for fp64 (v_rcp_f64 is ~26 bits), one for fp32.
#ifndef LCP_FAST_RCP_STEPS
#define LCP_FAST_RCP_STEPS 2
#endif
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
#if LCP_FAST_RCP_STEPS >= 2
  r = fma(fma(-x, r, 1.0), r, r);
#endif
  return r;
}
__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  r = fmaf(fmaf(-x, r, 1.0f), r, r);
  return r;
}

// 16-byte / 8-byte vector loads (the pointers are 16 B aligned by construction of the workspace)
__device__ __forceinline__ void load4(const float* src, float& a, float& b, float& c, float& d) {
  const float4 v = *reinterpret_cast<const float4*>(src); a = v.x; b = v.y; c = v.z; d = v.w;
}
__device__ __forceinline__ void load4(const double* src, double& a, double& b, double& c, double& d) {
  const double2 u = *reinterpret_cast<const double2*>(src), v = *reinterpret_cast<const double2*>(src + 2);
  a = u.x; b = u.y; c = v.x; d = v.y;
}
__device__ __forceinline__ void load2(const float* src, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(src); a = v.x; b = v.y;
}
__device__ __forceinline__ void load2(const double* src, double& a, double& b) {
  const double2 v = *reinterpret_cast<const double2*>(src); a = v.x; b = v.y;
}
__device__ __forceinline__ void store4(float* dst, float a, float b, float c, float d) { *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4(double* dst, double a, double b, double c, double d) {
  *reinterpret_cast<double2*>(dst) = make_double2(a, b); *reinterpret_cast<double2*>(dst + 2) = make_double2(c, d);
}
// the same as a streaming store (no reuse: gradients written once and read by another kernel) - A/B in lcp_bwd_quad: LCP_Q_NT_STORES
typedef float lcp_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store4_nt(float* dst, float a, float b, float c, float d) {
  lcp_f4v v = {a, b, c, d};
  __builtin_nontemporal_store(v, reinterpret_cast<lcp_f4v*>(dst));
}
__device__ __forceinline__ void store4_nt(double* dst, double a, double b, double c, double d) { store4(dst, a, b, c, d); }
__device__ __forceinline__ void store2(float* dst, float a, float b) { *reinterpret_cast<float2*>(dst) = make_float2(a, b); }
__device__ __forceinline__ void store2(double* dst, double a, double b) { *reinterpret_cast<double2*>(dst) = make_double2(a, b); }

// workspace per scene:  [TC] R2[MP*MP] Qit[256] GAc[512] S11i[64] x[16] s[64] z[64] y[8] pad -> 5120 TC
//                       [TI] Ft[MP*MP]   (lane-major copy of F: Ft[((j>>2)*MP + i)*4 + (j&3)] = F[i][j])
constexpr size_t WS_TC = 5120;
constexpr size_t WS_TI = MP * MP;
template <typename TI, typename TC> __host__ __device__ inline size_t ws_bytes() { return WS_TC * sizeof(TC) + WS_TI * sizeof(TI); }

template <typename TI, typename TC>
struct Ws {
  TC *R2, *Qit, *GAc, *S11i, *x, *s, *z, *y, *meta;     // meta[0] = structured flag, meta[1 + c] = mu of contact c
  TI* Ft;
  __device__ Ws(void* ws, int scene) {
    unsigned char* base = (unsigned char*)ws + (size_t)scene * ws_bytes<TI, TC>();
    TC* q = (TC*)base;
    R2 = q; q += MP * MP; Qit = q; q += NZP * NZP; GAc = q; q += EP * MP; S11i = q; q += EP * EP;
    x = q; q += NZP; s = q; q += MP; z = q; q += MP; y = q; q += EP; meta = q;   // 5080 + 19 <= WS_TC   (meta[18] = Q-is-diagonal flag)
    Ft = (TI*)(base + WS_TC * sizeof(TC));
  }
};

}  // namespace w64
}  // namespace lcp
