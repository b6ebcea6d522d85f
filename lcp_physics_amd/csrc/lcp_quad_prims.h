// lcp_quad_prims.h - DPP row primitives, NaN-key reductions and the hazard-safe v_fmac_f64_dpp blocks shared by the kernels that
// put one scene on the 16 lanes of a DPP row: lcp_quad.hip (four scenes per wavefront) and lcp_solo.hip (one scene per wavefront,
// small batches).  Moved verbatim out of lcp_quad.hip.
#pragma once
#include "lcp_wave_common.h"

namespace lcp {
namespace q16 {

using namespace w64;

constexpr int NCQ = 16;   // contacts (lanes) per scene
constexpr int EQ = 4;     // padded neq

// ---------------------------------------------------------------- DPP row primitives
template <int K> __device__ __forceinline__ float bc(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ double bc(double v) {
  return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true);     // one v_mov_b64_dpp (64-bit DPP: row_newbcast only)
}
// acc (+/-)= (lane K of the row's `src`) * mult: one 64-bit DPP move + FMA, hazards handled by the compiler.  (The
// single-instruction form v_fmac_f64_dpp exists but can only be emitted as inline asm, which the hazard recogniser
// does not see; it is used where a whole block of them can be made safe by construction - the LU trailing update below.)
template <int K, typename T> __device__ __forceinline__ void fmac_bc(T& acc, T src, T mult) { acc = fma(bc<K>(src), mult, acc); }
template <int K, typename T> __device__ __forceinline__ void fnmac_bc(T& acc, T src, T mult) { acc = fma(bc<K>(src), -mult, acc); }
#ifndef LCP_Q_ASM_LU
#define LCP_Q_ASM_LU 1      // 0: LU trailing update through the builtin as well (A/B aid)
#endif
// LCP_Q_LDSW = 1 keeps a packed symmetric copy of W in LDS and rebuilds T from it instead of re-reading the workspace
// (cuts the L2-miss traffic of the 11 factorisations).  Measured on MI355X, B = 4096 x 16 contacts: forward 0.260 ms
// against 0.227 ms with the plain 16-byte global loads (64 ds_read_b64 + address selects per lane cost more than 32
// L2/MALL-served dwordx4 loads), so it is off.
#ifndef LCP_Q_TS_GROUP
#define LCP_Q_TS_GROUP 4    // triangular-sweep steps per scalar guard (1, 2, 4, 8 or 16)
#endif
#ifndef LCP_Q_LDSW
#define LCP_Q_LDSW 0
#endif
// v where keep, else (numerically) zero: clears the high dword only - what is left is below 2^-1042, which vanishes
// in every accumulation it enters.  One v_cndmask instead of two for the masked triangular-solve multipliers.
__device__ __forceinline__ double keep_if(double v, bool keep) { return __hiloint2double(keep ? __double2hiint(v) : 0, __double2loint(v)); }
__device__ __forceinline__ float keep_if(float v, bool keep) { return keep ? v : 0.0f; }
template <int CTRL> __device__ __forceinline__ float dppx(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dppx(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// reductions over the 16 lanes of a row; every lane of the row receives the result
template <typename T> __device__ __forceinline__ T row_sum(T v) {
  v += dppx<0xB1>(v); v += dppx<0x4E>(v); v += dppx<0x141>(v); v += dppx<0x140>(v);
  return v;
}
template <typename T> __device__ __forceinline__ T row_max(T v) {
  T o;
  o = dppx<0xB1>(v); v = v > o ? v : o;
  o = dppx<0x4E>(v); v = v > o ? v : o;
  o = dppx<0x141>(v); v = v > o ? v : o;
  o = dppx<0x140>(v); v = v > o ? v : o;
  return v;
}
template <typename T> __device__ __forceinline__ T row_min(T v) {
  T o;
  o = dppx<0xB1>(v); v = v < o ? v : o;
  o = dppx<0x4E>(v); v = v < o ? v : o;
  o = dppx<0x141>(v); v = v < o ? v : o;
  o = dppx<0x140>(v); v = v < o ? v : o;
  return v;
}
// NaN handling without compare + select chains (on gfx950 every v_cmp -> v_cndmask pair costs an extra 2 wait states):
// the value reductions use v_max / v_min (which skip NaNs) and NaN-ness travels separately as the "key" = the value's
// high word without the sign bit, reduced with an unsigned max; only a NaN has a key above the infinity pattern.
__device__ __forceinline__ uint32_t nan_key(double v) { return (uint32_t)__double2hiint(v) & 0x7fffffffu; }
__device__ __forceinline__ uint32_t nan_key(float v) { return __float_as_uint(v) & 0x7fffffffu; }
template <typename T> __device__ __forceinline__ bool key_is_nan(uint32_t k);
template <> __device__ __forceinline__ bool key_is_nan<double>(uint32_t k) { return k > 0x7ff00000u; }
template <> __device__ __forceinline__ bool key_is_nan<float>(uint32_t k) { return k > 0x7f800000u; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ double fmax_(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fabs_(double a) { return __builtin_fabs(a); }
__device__ __forceinline__ float fabs_(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double fmin_(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fmin_(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ uint32_t row_umax(uint32_t k) {
  k = umax(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xf, 0xf, true));
  k = umax(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xf, 0xf, true));
  k = umax(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x141, 0xf, 0xf, true));
  k = umax(k, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x140, 0xf, 0xf, true));
  return k;
}
template <typename T> __device__ __forceinline__ T row_fmax(T v) {
  v = fmax_(v, dppx<0xB1>(v)); v = fmax_(v, dppx<0x4E>(v)); v = fmax_(v, dppx<0x141>(v)); v = fmax_(v, dppx<0x140>(v));
  return v;
}
template <typename T> __device__ __forceinline__ T row_fmin(T v) {
  v = fmin_(v, dppx<0xB1>(v)); v = fmin_(v, dppx<0x4E>(v)); v = fmin_(v, dppx<0x141>(v)); v = fmin_(v, dppx<0x140>(v));
  return v;
}
__device__ __forceinline__ bool row_any(bool p) { return row_umax(p ? 1u : 0u) != 0u; }
// NaN-propagating (Tensor.max()/min() semantics)
template <typename T> __device__ __forceinline__ T row_pmax(T v) { const uint32_t k = row_umax(nan_key(v)); v = row_fmax(v); return key_is_nan<T>(k) ? nan_of<T>() : v; }
template <typename T> __device__ __forceinline__ T row_pmin(T v) { const uint32_t k = row_umax(nan_key(v)); v = row_fmin(v); return key_is_nan<T>(k) ? nan_of<T>() : v; }

// Opaque pass-through: stops LICM from hoisting loop-invariant LDS loads / float->double conversions of the
// Jacobian rows out of the PDIPM loop (it did, and the ~200 extra live registers spilled to scratch).
// (never launder an LDS POINTER: it loses its address space and every access through it becomes a flat load followed
//  by a full s_waitcnt - launder an integer offset added to the pointer instead, see lds_opaque_zero)
__device__ __forceinline__ int lds_opaque_zero() { int z = 0; asm volatile("" : "+v"(z)); return z; }
__device__ __forceinline__ float launder(float f) { asm volatile("" : "+v"(f)); return f; }
// the lane-index compares (l16 > k, l16 == k ...) are loop-invariant: hoisted, their 64-bit masks overflow the SGPR file and
// come back through v_readlane spills; laundering the index per call keeps them local (one v_cmp each instead)
__device__ __forceinline__ int launder(int i) { asm volatile("" : "+v"(i)); return i; }
__device__ __forceinline__ double launder(double f) { asm volatile("" : "+v"(f)); return f; }

// an empty statement that needs `v` in a vector register HERE: loads issued early are not sunk into a predicated block further down
template <typename T> __device__ __forceinline__ void pin_vgpr(T& v) { asm volatile("" : "+v"(v)); }
// EXPERIMENT / round 6 (profiles/r06_ab_kernarg_warm.txt): the kernel-argument segment (FwdArgs + StepArgs: 0x1d0 bytes, eight 64-byte lines) is read by
// the compiler with s_load instructions placed near their uses - three to four of them wait one after the other in the prologue, each for its own
// line.  warm_kernarg() asks for one dword of every line at once and waits once: the compiler's own loads then hit the scalar cache.
#ifndef LCP_Q_KERNARG_WARM
#define LCP_Q_KERNARG_WARM 0
#endif
template <int NBYTES>
__device__ __forceinline__ void warm_kernarg() {
#if LCP_Q_KERNARG_WARM
  auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  int t0, t1, t2, t3, t4, t5, t6, t7;
  static_assert(NBYTES <= 512, "eight lines");
  asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\t"
               "s_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7) : "s"(kp) : "memory");
#endif
}
// Phase timing (build with -DLCP_Q_PROFILE; the dense forward then writes cycle totals to the debug trace buffer)
#ifdef LCP_Q_PROFILE
struct Prof { long long t[10]; long long last; };
// (the scheduling barriers keep the straight-line phases of the size-specialised kernels on their own side of the clock read)
#define LCP_QTICK(pr, i) { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); __builtin_amdgcn_sched_barrier(0); \
                           (pr).t[i] += now_ - (pr).last; (pr).last = now_; }
#define LCP_QPROF_ARG , Prof& pr
#define LCP_QPROF_PASS , pr
#else
#define LCP_QTICK(pr, i)
#define LCP_QPROF_ARG
#define LCP_QPROF_PASS
#endif

// a / b for the element-wise quotients of pdipm.py (s / z, -z / dz, rs / s ...).  LCP_Q_FAST_DIV = 1: a x (v_rcp_f64 + two Newton steps)
// - within an ulp or two of the IEEE quotient, 7 instructions instead of the 13 of the division sequence (a solve makes ~420 of them).
// `qdiv_x` also returns what IEEE division returns for b = 0, +-inf and NaN (the Newton steps turn those into NaN: the raw v_rcp value
// is taken then) - pdipm.py:182-186 relies on -v / 0 = -+inf.
#ifndef LCP_Q_FAST_DIV
#define LCP_Q_FAST_DIV 0
#endif
__device__ __forceinline__ double qdiv(double a, double b) {
#if LCP_Q_FAST_DIV
  return a * fast_rcp(b);
#else
  return a / b;
#endif
}
__device__ __forceinline__ double qdiv_x(double a, double b) {
#if LCP_Q_FAST_DIV
  const double r0 = __builtin_amdgcn_rcp(b);
  double r = fma(fma(-b, r0, 1.0), r0, r0);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * ((r != r) ? r0 : r);
#else
  return a / b;
#endif
}
__device__ __forceinline__ float qdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float qdiv_x(float a, float b) { return a / b; }

template <typename TC> struct M4 { TC n, f1, f2, g; };       // the four inequality rows of one contact
template <typename TC> __device__ __forceinline__ M4<TC> m4(TC a, TC b, TC c, TC d) { M4<TC> r; r.n = a; r.f1 = b; r.f2 = c; r.g = d; return r; }
template <typename TC> __device__ __forceinline__ TC sum4(const M4<TC>& a) { return (a.n + a.f1) + (a.f2 + a.g); }

// ---------------------------------------------------------------- LU column updates as hazard-safe asm blocks
// The trailing update of an LU step is `row[j] -= bcast_k(row_k[j]) * l` for every remaining column j.  Written with
// the builtin it costs v_mov_b64_dpp + v_fma_f64 per entry; v_fmac_f64_dpp folds the broadcast into the FMA (4.7 instead
// of 6.7 cycles per entry, tools/microbench/pair_cost.hip).  clang never forms that instruction itself (its DPP combiner
// skips FMAC), so it is emitted as inline asm - and inline asm is invisible to the hazard recogniser: gfx950 needs 2 wait
// states between a VALU write of a VGPR and a DPP read of it (tools/microbench/dpp_hazard.hip shows stale reads
// without them).  Every asm statement below therefore starts with `s_nop 1` (covers whatever the compiler put in front:
// copies, v_accvgpr_read ...) and inside a statement no DPP source is written before it is read; up to 14 column pairs
// (30 operands) share one s_nop.
#define LCP_DPP_FULL "row_mask:0xf bank_mask:0xf"
#define LCP_LUA_COL(A, U)                                                                   \
  "v_fmac_f64_dpp %[" #U "], %[" #A "], -%[lu] row_newbcast:%[k] " LCP_DPP_FULL "\n\t"      \
  "v_fmac_f64_dpp %[" #A "], %[" #A "], -%[la] row_newbcast:%[k] " LCP_DPP_FULL "\n\t"
#define LCP_LUU_COL(U) "v_fmac_f64_dpp %[" #U "], %[" #U "], -%[lu] row_newbcast:%[k] " LCP_DPP_FULL "\n\t"
#define LCP_LUA_S1 LCP_LUA_COL(a0, u0)
#define LCP_LUA_S2 LCP_LUA_S1 LCP_LUA_COL(a1, u1)
#define LCP_LUA_S3 LCP_LUA_S2 LCP_LUA_COL(a2, u2)
#define LCP_LUA_S4 LCP_LUA_S3 LCP_LUA_COL(a3, u3)
#define LCP_LUA_S5 LCP_LUA_S4 LCP_LUA_COL(a4, u4)
#define LCP_LUA_S6 LCP_LUA_S5 LCP_LUA_COL(a5, u5)
#define LCP_LUA_S7 LCP_LUA_S6 LCP_LUA_COL(a6, u6)
#define LCP_LUA_S8 LCP_LUA_S7 LCP_LUA_COL(a7, u7)
#define LCP_LUA_S9 LCP_LUA_S8 LCP_LUA_COL(a8, u8)
#define LCP_LUA_S10 LCP_LUA_S9 LCP_LUA_COL(a9, u9)
#define LCP_LUA_S11 LCP_LUA_S10 LCP_LUA_COL(a10, u10)
#define LCP_LUA_S12 LCP_LUA_S11 LCP_LUA_COL(a11, u11)
#define LCP_LUA_S13 LCP_LUA_S12 LCP_LUA_COL(a12, u12)
#define LCP_LUA_S14 LCP_LUA_S13 LCP_LUA_COL(a13, u13)
#define LCP_LUA_O1 [a0] "+v"(ta[J0 + 0]), [u0] "+v"(tu[J0 + 0])
#define LCP_LUA_O2 LCP_LUA_O1, [a1] "+v"(ta[J0 + 1]), [u1] "+v"(tu[J0 + 1])
#define LCP_LUA_O3 LCP_LUA_O2, [a2] "+v"(ta[J0 + 2]), [u2] "+v"(tu[J0 + 2])
#define LCP_LUA_O4 LCP_LUA_O3, [a3] "+v"(ta[J0 + 3]), [u3] "+v"(tu[J0 + 3])
#define LCP_LUA_O5 LCP_LUA_O4, [a4] "+v"(ta[J0 + 4]), [u4] "+v"(tu[J0 + 4])
#define LCP_LUA_O6 LCP_LUA_O5, [a5] "+v"(ta[J0 + 5]), [u5] "+v"(tu[J0 + 5])
#define LCP_LUA_O7 LCP_LUA_O6, [a6] "+v"(ta[J0 + 6]), [u6] "+v"(tu[J0 + 6])
#define LCP_LUA_O8 LCP_LUA_O7, [a7] "+v"(ta[J0 + 7]), [u7] "+v"(tu[J0 + 7])
#define LCP_LUA_O9 LCP_LUA_O8, [a8] "+v"(ta[J0 + 8]), [u8] "+v"(tu[J0 + 8])
#define LCP_LUA_O10 LCP_LUA_O9, [a9] "+v"(ta[J0 + 9]), [u9] "+v"(tu[J0 + 9])
#define LCP_LUA_O11 LCP_LUA_O10, [a10] "+v"(ta[J0 + 10]), [u10] "+v"(tu[J0 + 10])
#define LCP_LUA_O12 LCP_LUA_O11, [a11] "+v"(ta[J0 + 11]), [u11] "+v"(tu[J0 + 11])
#define LCP_LUA_O13 LCP_LUA_O12, [a12] "+v"(ta[J0 + 12]), [u12] "+v"(tu[J0 + 12])
#define LCP_LUA_O14 LCP_LUA_O13, [a13] "+v"(ta[J0 + 13]), [u13] "+v"(tu[J0 + 13])
#define LCP_LUU_S1 LCP_LUU_COL(u0)
#define LCP_LUU_S2 LCP_LUU_S1 LCP_LUU_COL(u1)
#define LCP_LUU_S3 LCP_LUU_S2 LCP_LUU_COL(u2)
#define LCP_LUU_S4 LCP_LUU_S3 LCP_LUU_COL(u3)
#define LCP_LUU_S5 LCP_LUU_S4 LCP_LUU_COL(u4)
#define LCP_LUU_S6 LCP_LUU_S5 LCP_LUU_COL(u5)
#define LCP_LUU_S7 LCP_LUU_S6 LCP_LUU_COL(u6)
#define LCP_LUU_S8 LCP_LUU_S7 LCP_LUU_COL(u7)
#define LCP_LUU_S9 LCP_LUU_S8 LCP_LUU_COL(u8)
#define LCP_LUU_S10 LCP_LUU_S9 LCP_LUU_COL(u9)
#define LCP_LUU_S11 LCP_LUU_S10 LCP_LUU_COL(u10)
#define LCP_LUU_S12 LCP_LUU_S11 LCP_LUU_COL(u11)
#define LCP_LUU_S13 LCP_LUU_S12 LCP_LUU_COL(u12)
#define LCP_LUU_S14 LCP_LUU_S13 LCP_LUU_COL(u13)
#define LCP_LUU_S15 LCP_LUU_S14 LCP_LUU_COL(u14)
#define LCP_LUU_O1 [u0] "+v"(tu[J0 + 0])
#define LCP_LUU_O2 LCP_LUU_O1 , [u1] "+v"(tu[J0 + 1])
#define LCP_LUU_O3 LCP_LUU_O2 , [u2] "+v"(tu[J0 + 2])
#define LCP_LUU_O4 LCP_LUU_O3 , [u3] "+v"(tu[J0 + 3])
#define LCP_LUU_O5 LCP_LUU_O4 , [u4] "+v"(tu[J0 + 4])
#define LCP_LUU_O6 LCP_LUU_O5 , [u5] "+v"(tu[J0 + 5])
#define LCP_LUU_O7 LCP_LUU_O6 , [u6] "+v"(tu[J0 + 6])
#define LCP_LUU_O8 LCP_LUU_O7 , [u7] "+v"(tu[J0 + 7])
#define LCP_LUU_O9 LCP_LUU_O8 , [u8] "+v"(tu[J0 + 8])
#define LCP_LUU_O10 LCP_LUU_O9 , [u9] "+v"(tu[J0 + 9])
#define LCP_LUU_O11 LCP_LUU_O10 , [u10] "+v"(tu[J0 + 10])
#define LCP_LUU_O12 LCP_LUU_O11 , [u11] "+v"(tu[J0 + 11])
#define LCP_LUU_O13 LCP_LUU_O12 , [u12] "+v"(tu[J0 + 12])
#define LCP_LUU_O14 LCP_LUU_O13 , [u13] "+v"(tu[J0 + 13])
#define LCP_LUU_O15 LCP_LUU_O14 , [u14] "+v"(tu[J0 + 14])
template <int K, int J0, int N> struct LuColsA;      // rows a and u of the lane, columns J0 .. J0+N-1, pivot row K
template <int K, int J0, int N> struct LuColsU;      // row u only
#define LCP_LUA_DEF(N)                                                                                                   \
  template <int K, int J0> struct LuColsA<K, J0, N> {                                                                    \
    static __device__ __forceinline__ void run(double (&ta)[32], double (&tu)[32], double la, double lu) {               \
      asm("s_nop 1\n\t" LCP_LUA_S##N : LCP_LUA_O##N : [la] "v"(la), [lu] "v"(lu), [k] "n"(K));                           \
    }                                                                                                                    \
  };
#define LCP_LUU_DEF(N)                                                                                                   \
  template <int K, int J0> struct LuColsU<K, J0, N> {                                                                    \
    static __device__ __forceinline__ void run(double (&tu)[32], double lu) {                                            \
      asm("s_nop 1\n\t" LCP_LUU_S##N : LCP_LUU_O##N : [lu] "v"(lu), [k] "n"(K));                                         \
    }                                                                                                                    \
  };
LCP_LUA_DEF(1)
LCP_LUA_DEF(2)
LCP_LUA_DEF(3)
LCP_LUA_DEF(4)
LCP_LUA_DEF(5)
LCP_LUA_DEF(6)
LCP_LUA_DEF(7)
LCP_LUA_DEF(8)
LCP_LUA_DEF(9)
LCP_LUA_DEF(10)
LCP_LUA_DEF(11)
LCP_LUA_DEF(12)
LCP_LUA_DEF(13)
LCP_LUA_DEF(14)
LCP_LUU_DEF(1)
LCP_LUU_DEF(2)
LCP_LUU_DEF(3)
LCP_LUU_DEF(4)
LCP_LUU_DEF(5)
LCP_LUU_DEF(6)
LCP_LUU_DEF(7)
LCP_LUU_DEF(8)
LCP_LUU_DEF(9)
LCP_LUU_DEF(10)
LCP_LUU_DEF(11)
LCP_LUU_DEF(12)
LCP_LUU_DEF(13)
LCP_LUU_DEF(14)
LCP_LUU_DEF(15)
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_a(double (&ta)[32], double (&tu)[32], double la, double lu) {
  if constexpr (N > 14) { LuColsA<K, J0, 14>::run(ta, tu, la, lu); lu_cols_a<K, J0 + 14, N - 14>(ta, tu, la, lu); }
  else if constexpr (N > 0) LuColsA<K, J0, N>::run(ta, tu, la, lu);
}
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_u(double (&tu)[32], double lu) {
  if constexpr (N > 0) LuColsU<K, J0, N>::run(tu, lu);
}
// fp32 arithmetic keeps the builtin form (32-bit DPP is folded by the compiler where it can be)
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_a(float (&ta)[32], float (&tu)[32], float la, float lu) {
  static_for<N>([&](auto JJ) LCP_INL { constexpr int j = J0 + JJ; const float sj = bc<K>(ta[j]); tu[j] = fmaf(-lu, sj, tu[j]); ta[j] = fmaf(-la, sj, ta[j]); });
}
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_u(float (&tu)[32], float lu) {
  static_for<N>([&](auto JJ) LCP_INL { constexpr int j = J0 + JJ; tu[j] = fmaf(-lu, bc<K>(tu[j]), tu[j]); });
}

// ---------------------------------------------------------------- asm blocks of the body-space variant
// (same rules as the LU blocks above: every statement opens with `s_nop 1`, no DPP source is written inside a statement)
// LU trailing update over the lane's x-row (xr) and equality row (er), 20 columns: the LuColsA pattern on 20-entry arrays
template <int K, int J0, int N> struct LuColsP;
#define LCP_LUP_DEF(N)                                                                                                   \
  template <int K, int J0> struct LuColsP<K, J0, N> {                                                                    \
    static __device__ __forceinline__ void run(double (&ta)[20], double (&tu)[20], double la, double lu) {               \
      asm("s_nop 1\n\t" LCP_LUA_S##N : LCP_LUA_O##N : [la] "v"(la), [lu] "v"(lu), [k] "n"(K));                           \
    }                                                                                                                    \
  };
LCP_LUP_DEF(1) LCP_LUP_DEF(2) LCP_LUP_DEF(3) LCP_LUP_DEF(4) LCP_LUP_DEF(5) LCP_LUP_DEF(6) LCP_LUP_DEF(7)
LCP_LUP_DEF(8) LCP_LUP_DEF(9) LCP_LUP_DEF(10) LCP_LUP_DEF(11) LCP_LUP_DEF(12) LCP_LUP_DEF(13) LCP_LUP_DEF(14)
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_p(double (&xr)[20], double (&er)[20], double lx, double le) {
  if constexpr (N > 14) { LuColsP<K, J0, 14>::run(xr, er, lx, le); lu_cols_p<K, J0 + 14, N - 14>(xr, er, lx, le); }
  else if constexpr (N > 0) LuColsP<K, J0, N>::run(xr, er, lx, le);
}
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_p(float (&xr)[20], float (&er)[20], float lx, float le) {
  static_for<N>([&](auto JJ) LCP_INL { constexpr int j = J0 + JJ; const float sj = bc<K>(xr[j]); er[j] = fmaf(-le, sj, er[j]); xr[j] = fmaf(-lx, sj, xr[j]); });
}
// single-row form (pinned variant: no equality rows), the LuColsU pattern on a 20-entry array
template <int K, int J0, int N> struct LuColsX;
#define LCP_LUX_DEF(N)                                                                                                   \
  template <int K, int J0> struct LuColsX<K, J0, N> {                                                                    \
    static __device__ __forceinline__ void run(double (&tu)[20], double lu) {                                            \
      asm("s_nop 1\n\t" LCP_LUU_S##N : LCP_LUU_O##N : [lu] "v"(lu), [k] "n"(K));                                         \
    }                                                                                                                    \
  };
LCP_LUX_DEF(1) LCP_LUX_DEF(2) LCP_LUX_DEF(3) LCP_LUX_DEF(4) LCP_LUX_DEF(5) LCP_LUX_DEF(6) LCP_LUX_DEF(7)
LCP_LUX_DEF(8) LCP_LUX_DEF(9) LCP_LUX_DEF(10) LCP_LUX_DEF(11) LCP_LUX_DEF(12) LCP_LUX_DEF(13) LCP_LUX_DEF(14)
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_x(double (&xr)[20], double lx) {
  if constexpr (N > 0) LuColsX<K, J0, N>::run(xr, lx);
}
template <int K, int J0, int N> __device__ __forceinline__ void lu_cols_x(float (&xr)[20], float lx) {
  static_for<N>([&](auto JJ) LCP_INL { constexpr int j = J0 + JJ; xr[j] = fmaf(-lx, bc<K>(xr[j]), xr[j]); });
}
// formation: xr[j] += a * (lane K's p0[j]) + b * (lane K's p1[j]) for eight columns
// (the eight p0 terms first, then the eight p1 terms: the two updates of a column are eight instructions apart)
#define LCP_PQF_ONE(X, P, M) "v_fmac_f64_dpp %[" #X "], %[" #P "], %[" #M "] row_newbcast:%[k] " LCP_DPP_FULL "\n\t"
template <int K, int J0> __device__ __forceinline__ void pq_form8(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x4, p4, a) LCP_PQF_ONE(x5, p5, a) LCP_PQF_ONE(x6, p6, a) LCP_PQF_ONE(x7, p7, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) LCP_PQF_ONE(x4, q4, b) LCP_PQF_ONE(x5, q5, b) LCP_PQF_ONE(x6, q6, b) LCP_PQF_ONE(x7, q7, b)
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3]), [x4] "+v"(xr[J0 + 4]), [x5] "+v"(xr[J0 + 5]), [x6] "+v"(xr[J0 + 6]), [x7] "+v"(xr[J0 + 7])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [p4] "v"(p0[J0 + 4]), [q4] "v"(p1[J0 + 4]), [p5] "v"(p0[J0 + 5]), [q5] "v"(p1[J0 + 5]), [p6] "v"(p0[J0 + 6]), [q6] "v"(p1[J0 + 6]), [p7] "v"(p0[J0 + 7]), [q7] "v"(p1[J0 + 7]), [a] "v"(a), [b] "v"(b), [k] "n"(K));
}
template <int K, int J0> __device__ __forceinline__ void pq_form8(float (&xr)[20], const float (&p0)[16], const float (&p1)[16], float a, float b) {
  static_for<8>([&](auto JJ) LCP_INL { constexpr int j = J0 + JJ; xr[j] = fmaf(bc<K>(p0[j]), a, xr[j]); xr[j] = fmaf(bc<K>(p1[j]), b, xr[j]); });
}

#ifndef LCP_DPP_FULL_EARLY
#define LCP_DPP_FULL_EARLY "row_mask:0xf bank_mask:0xf"
#endif
#ifndef LCP_GV_ONE
#define LCP_GV_ONE(ACC, M, K) "v_fmac_f64_dpp %[" #ACC "], %[v], %[" #M "] row_newbcast:%[" #K "] " LCP_DPP_FULL_EARLY "\n\t"
#endif
// ---------------------------------------------------------------- N-column forms of the two blocks above (compile-time sizes: only the
// columns of the free coordinates are formed / multiplied) - generated, same rules: `s_nop 1` first, DPP sources read-only
template <int K, int J0, int N> struct PqFormN;
template <int J0, int N> struct GvN;
template <int K, int J0> struct PqFormN<K, J0, 1> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x0, q0, b) 
      : [x0] "+v"(xr[J0 + 0])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 2> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 3> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 4> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 5> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x4, p4, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) LCP_PQF_ONE(x4, q4, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3]), [x4] "+v"(xr[J0 + 4])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [p4] "v"(p0[J0 + 4]), [q4] "v"(p1[J0 + 4]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 6> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x4, p4, a) LCP_PQF_ONE(x5, p5, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) LCP_PQF_ONE(x4, q4, b) LCP_PQF_ONE(x5, q5, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3]), [x4] "+v"(xr[J0 + 4]), [x5] "+v"(xr[J0 + 5])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [p4] "v"(p0[J0 + 4]), [q4] "v"(p1[J0 + 4]), [p5] "v"(p0[J0 + 5]), [q5] "v"(p1[J0 + 5]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 7> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x4, p4, a) LCP_PQF_ONE(x5, p5, a) LCP_PQF_ONE(x6, p6, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) LCP_PQF_ONE(x4, q4, b) LCP_PQF_ONE(x5, q5, b) LCP_PQF_ONE(x6, q6, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3]), [x4] "+v"(xr[J0 + 4]), [x5] "+v"(xr[J0 + 5]), [x6] "+v"(xr[J0 + 6])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [p4] "v"(p0[J0 + 4]), [q4] "v"(p1[J0 + 4]), [p5] "v"(p0[J0 + 5]), [q5] "v"(p1[J0 + 5]), [p6] "v"(p0[J0 + 6]), [q6] "v"(p1[J0 + 6]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int K, int J0> struct PqFormN<K, J0, 8> { static __device__ __forceinline__ void run(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  asm("s_nop 1\n\t" LCP_PQF_ONE(x0, p0, a) LCP_PQF_ONE(x1, p1, a) LCP_PQF_ONE(x2, p2, a) LCP_PQF_ONE(x3, p3, a) LCP_PQF_ONE(x4, p4, a) LCP_PQF_ONE(x5, p5, a) LCP_PQF_ONE(x6, p6, a) LCP_PQF_ONE(x7, p7, a) LCP_PQF_ONE(x0, q0, b) LCP_PQF_ONE(x1, q1, b) LCP_PQF_ONE(x2, q2, b) LCP_PQF_ONE(x3, q3, b) LCP_PQF_ONE(x4, q4, b) LCP_PQF_ONE(x5, q5, b) LCP_PQF_ONE(x6, q6, b) LCP_PQF_ONE(x7, q7, b) 
      : [x0] "+v"(xr[J0 + 0]), [x1] "+v"(xr[J0 + 1]), [x2] "+v"(xr[J0 + 2]), [x3] "+v"(xr[J0 + 3]), [x4] "+v"(xr[J0 + 4]), [x5] "+v"(xr[J0 + 5]), [x6] "+v"(xr[J0 + 6]), [x7] "+v"(xr[J0 + 7])
      : [p0] "v"(p0[J0 + 0]), [q0] "v"(p1[J0 + 0]), [p1] "v"(p0[J0 + 1]), [q1] "v"(p1[J0 + 1]), [p2] "v"(p0[J0 + 2]), [q2] "v"(p1[J0 + 2]), [p3] "v"(p0[J0 + 3]), [q3] "v"(p1[J0 + 3]), [p4] "v"(p0[J0 + 4]), [q4] "v"(p1[J0 + 4]), [p5] "v"(p0[J0 + 5]), [q5] "v"(p1[J0 + 5]), [p6] "v"(p0[J0 + 6]), [q6] "v"(p1[J0 + 6]), [p7] "v"(p0[J0 + 7]), [q7] "v"(p1[J0 + 7]), [a] "v"(a), [b] "v"(b), [k] "n"(K)); } };
template <int J0> struct GvN<J0, 1> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0)); } };
template <int J0> struct GvN<J0, 2> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1)); } };
template <int J0> struct GvN<J0, 3> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2)); } };
template <int J0> struct GvN<J0, 4> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2), [c3] "v"(c[3]), [t3_] "v"(t[3]), [k3] "n"(J0 + 3)); } };
template <int J0> struct GvN<J0, 5> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3) LCP_GV_ONE(n0, c4, k4) LCP_GV_ONE(t0, t4_, k4) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2), [c3] "v"(c[3]), [t3_] "v"(t[3]), [k3] "n"(J0 + 3), [c4] "v"(c[4]), [t4_] "v"(t[4]), [k4] "n"(J0 + 4)); } };
template <int J0> struct GvN<J0, 6> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3) LCP_GV_ONE(n0, c4, k4) LCP_GV_ONE(t0, t4_, k4) LCP_GV_ONE(n1, c5, k5) LCP_GV_ONE(t1, t5_, k5) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2), [c3] "v"(c[3]), [t3_] "v"(t[3]), [k3] "n"(J0 + 3), [c4] "v"(c[4]), [t4_] "v"(t[4]), [k4] "n"(J0 + 4), [c5] "v"(c[5]), [t5_] "v"(t[5]), [k5] "n"(J0 + 5)); } };
template <int J0> struct GvN<J0, 7> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3) LCP_GV_ONE(n0, c4, k4) LCP_GV_ONE(t0, t4_, k4) LCP_GV_ONE(n1, c5, k5) LCP_GV_ONE(t1, t5_, k5) LCP_GV_ONE(n0, c6, k6) LCP_GV_ONE(t0, t6_, k6) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2), [c3] "v"(c[3]), [t3_] "v"(t[3]), [k3] "n"(J0 + 3), [c4] "v"(c[4]), [t4_] "v"(t[4]), [k4] "n"(J0 + 4), [c5] "v"(c[5]), [t5_] "v"(t[5]), [k5] "n"(J0 + 5), [c6] "v"(c[6]), [t6_] "v"(t[6]), [k6] "n"(J0 + 6)); } };
template <int J0> struct GvN<J0, 8> { static __device__ __forceinline__ void run(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t" LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1) LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3) LCP_GV_ONE(n0, c4, k4) LCP_GV_ONE(t0, t4_, k4) LCP_GV_ONE(n1, c5, k5) LCP_GV_ONE(t1, t5_, k5) LCP_GV_ONE(n0, c6, k6) LCP_GV_ONE(t0, t6_, k6) LCP_GV_ONE(n1, c7, k7) LCP_GV_ONE(t1, t7_, k7) 
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [t0_] "v"(t[0]), [k0] "n"(J0 + 0), [c1] "v"(c[1]), [t1_] "v"(t[1]), [k1] "n"(J0 + 1), [c2] "v"(c[2]), [t2_] "v"(t[2]), [k2] "n"(J0 + 2), [c3] "v"(c[3]), [t3_] "v"(t[3]), [k3] "n"(J0 + 3), [c4] "v"(c[4]), [t4_] "v"(t[4]), [k4] "n"(J0 + 4), [c5] "v"(c[5]), [t5_] "v"(t[5]), [k5] "n"(J0 + 5), [c6] "v"(c[6]), [t6_] "v"(t[6]), [k6] "n"(J0 + 6), [c7] "v"(c[7]), [t7_] "v"(t[7]), [k7] "n"(J0 + 7)); } };
template <int K, int J0, int N> __device__ __forceinline__ void pq_formN(double (&xr)[20], const double (&p0)[16], const double (&p1)[16], double a, double b) {
  if constexpr (N > 0) PqFormN<K, J0, N>::run(xr, p0, p1, a, b);
}
template <int J0, int N> __device__ __forceinline__ void gvN_dpp(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  if constexpr (N > 0) GvN<J0, N>::run(n0, n1, t0, t1, v, c, t);
}

}  // namespace q16
}  // namespace lcp
