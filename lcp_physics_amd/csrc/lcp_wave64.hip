// lcp_wave64.hip - register-resident PDIPM LCP kernels for gfx950: ONE WAVEFRONT PER SCENE.
//
// Fast path for nineq <= 64, nz <= 16, neq <= 8, fp32 I/O (every BASELINE config except the
// nineq-256 pile).  Design (CDNA4-first):
//   * lane i of the wave owns inequality i: row i of T = R + diag(s/z) lives in 64 registers of
//     that lane (128 VGPRs in fp64), the whole LU and both triangular solves run out of the
//     register file; the pivot row is broadcast with v_readlane (SGPR operand of the FMA) - no
//     LDS traffic and no barrier inside the factorisation;
//   * partial pivoting is implicit: rows never move, a step just picks the not-yet-used lane with
//     the largest |T[i][k]| (one 32-bit key max-reduction) and reads that lane's row;
//   * x-space vectors (nz <= 16 entries) are replicated in the four 16-lane groups, e-space vectors
//     (neq <= 8) in the eight 8-lane groups; G is kept in LDS twice (column-major for G v,
//     row-major for G^T w) in I/O precision, which is exact for fp32 inputs;
//   * R and a lane-major copy of F stay in the HBM workspace and are streamed (coalesced, 16 B per
//     lane) once per PDIPM iteration; residual norms and step lengths are wavefront reductions.
// Reference lines implemented: see the table at the top of lcp_generic.hip (same algorithm).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcp_wave_common.h"

namespace lcp {
namespace w64 {

constexpr int GRS = 17;   // row stride of the row-major G copy (bank-conflict-free group reads)
// Scenes (waves) per workgroup.  The unrolled LU is far larger than the 64 KB instruction cache, so a
// wave on its own streams its code from L2 every PDIPM iteration (measured: fetch-bound).  Putting WPB
// waves in one workgroup and re-synchronising them with s_barrier at every elimination step keeps them
// in the same few cache lines, so the instruction stream is fetched once per workgroup.
#ifndef LCP_W64_WPB
#define LCP_W64_WPB 1
#endif
constexpr int WPB = LCP_W64_WPB;
#ifndef LCP_W64_OCC
#define LCP_W64_OCC 1     // waves per SIMD the structured forward kernel is compiled for
#endif
#ifndef LCP_W64_SYNC_EVERY
#define LCP_W64_SYNC_EVERY 1
#endif

// ---------------------------------------------------------------- lane primitives
template <typename T> __device__ __forceinline__ T rdlane(T v, int src);
template <> __device__ __forceinline__ float rdlane<float>(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
template <> __device__ __forceinline__ double rdlane<double>(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rdlane_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// Wavefront reductions with DPP row operations (quad_perm / row_half_mirror / row_mirror: a few cycles each)
// plus four v_readlane for the cross-row step - the ds_bpermute butterfly they replace was ~100 cycles of
// latency per step (micro-benchmark: ds_bpermute ~22 cycles issue, ~60+ latency).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

template <typename T, typename Op> __device__ __forceinline__ T wave_reduce(T v, Op op) {
  v = op(v, dpp_mov<DPP_XOR1>(v));
  v = op(v, dpp_mov<DPP_XOR2>(v));
  v = op(v, dpp_mov<DPP_HALF_MIRROR>(v));
  v = op(v, dpp_mov<DPP_ROW_MIRROR>(v));                 // every lane now holds its 16-lane row's result
  const T r0 = rdlane(v, 0), r1 = rdlane(v, 16), r2 = rdlane(v, 32), r3 = rdlane(v, 48);
  return op(op(r0, r1), op(r2, r3));
}
struct RSum { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; } };
struct RMax { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a > b ? a : b; } };
struct RMin { template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a < b ? a : b; } };

template <typename T> __device__ __forceinline__ T wave_sum(T v) { return wave_reduce(v, RSum()); }
template <typename T> __device__ __forceinline__ void wave_sum2(T& a, T& b) { a = wave_reduce(a, RSum()); b = wave_reduce(b, RSum()); }
// NaN-propagating max / min over the wave (Tensor.max()/min() semantics): plain max, then force NaN if any
// lane held one.
template <typename T> __device__ __forceinline__ void wave_pmax2(T& a, T& b) {
  const bool na = __any(a != a), nb = __any(b != b);
  a = wave_reduce(a, RMax()); b = wave_reduce(b, RMax());
  if (na) a = nan_of<T>();
  if (nb) b = nan_of<T>();
}
template <typename T> __device__ __forceinline__ void wave_pmin2(T& a, T& b) {
  const bool na = __any(a != a), nb = __any(b != b);
  a = wave_reduce(a, RMin()); b = wave_reduce(b, RMin());
  if (na) a = nan_of<T>();
  if (nb) b = nan_of<T>();
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = v > o ? v : o;
  }
  return v;
}

// ---------------------------------------------------------------- LU of T in registers
// t[j] of lane i = T[i][j].  Rows never move: step k eliminates with pivot lane p_k.
//   mystep : step at which this lane was the pivot (MP while unused)
//   porder : lane k holds p_k
//   udinv  : 1 / U[mystep][mystep] of this lane's row
// After the call: for lanes with mystep > k, t[k] holds the multiplier L[.][k]; for the lane with
// mystep == k, t[k..] is row k of U.  Returns true if an exact zero pivot was met.
//
// Code-size-aware blocking (the instruction cache is 64 KB; a fully unrolled 64x64 elimination is
// ~48 KB of code on its own and ran instruction-fetch-latency bound - measured 21 cycles/instruction):
//   * the matrix is processed in four 16-column panels;
//   * a panel is copied to LDS and factorised THERE with rolled loops (LDS addresses may be dynamic,
//     register names may not), then copied back;
//   * the trailing columns are updated by a rolled loop over the panel's 16 elimination steps: the
//     registers it touches do not depend on the step, the pivot lane and the multiplier do, and both
//     are run-time values (v_readlane lane select / an LDS read).
// Every v_readlane + FMA pair of the trailing update therefore exists once in the code per panel
// (96 pairs in total instead of 2016).
constexpr int PB = 16;          // panel width
constexpr int PS = PB + 1;      // LDS row stride of the panel (odd: conflict-free column walks)

template <typename TC, bool PIVOT>
__device__ __forceinline__ bool lu_factor(TC (&t)[MP], TC* pan, int m, int lane, int& mystep, int& porder, TC& udinv) {
  mystep = MP; porder = lane; udinv = (TC)1;
  bool singular = false;
  TC* myrow = pan + lane * PS;
  static_for<MP / PB>([&](auto BK) LCP_INL {
    constexpr int b = BK;
    if (PB * b < m) {
      // 1. panel -> LDS
      static_for<PB>([&](auto JJ) LCP_INL { myrow[JJ] = t[PB * b + JJ]; });
      __syncthreads();
      const int kend = (m - PB * b) < PB ? (m - PB * b) : PB;
      // 2. factorise the panel in LDS (partial pivoting over the lanes not used yet)
#pragma unroll 1
      for (int kk = 0; kk < kend; ++kk) {
        const int k = PB * b + kk;
        const TC a = myrow[kk];
        int p = k;
        if (PIVOT) {
          const bool cand = (mystep == MP) && (lane < m);
          unsigned key = cand ? ((__float_as_uint(fabsf((float)a)) & ~63u) | (unsigned)lane) : 0u;
          key = wave_max_u32(key);
          p = (int)(__builtin_amdgcn_readfirstlane((int)key) & 63);
        }
        const TC piv = pan[p * PS + kk];
        singular = singular || (piv == (TC)0);
        const TC inv = (TC)1 / piv;
        const bool isp = (lane == p);
        const bool act = (mystep == MP) && !isp && (lane < m);
        const TC l = act ? a * inv : (TC)0;
        if (act) myrow[kk] = l;
        mystep = isp ? k : mystep;
        udinv = isp ? inv : udinv;
        porder = (lane == k) ? p : porder;
        const TC* prow = pan + p * PS;
#pragma unroll 1
        for (int jj = kk + 1; jj < PB; ++jj) myrow[jj] = fma(-l, prow[jj], myrow[jj]);
        __syncthreads();
      }
      // 3. panel -> registers
      static_for<PB>([&](auto JJ) LCP_INL { t[PB * b + JJ] = myrow[JJ]; });
      // 4. trailing columns: rolled over the panel's steps, static over the columns
      if constexpr (b + 1 < MP / PB) {
        if (PB * (b + 1) < m) {
#pragma unroll 1
          for (int kk = 0; kk < kend; ++kk) {
            const int k = PB * b + kk;
            const int p = rdlane_i(porder, k);
            const TC lraw = myrow[kk];
            const TC l = (mystep > k && lane < m) ? lraw : (TC)0;
            static_for<MP / PB - 1 - b>([&](auto CB) LCP_INL {
              constexpr int cb = b + 1 + CB;
              if (PB * cb < m) {
                static_for<PB>([&](auto JJ) LCP_INL {
                  constexpr int j = PB * cb + JJ;
                  t[j] = fma(-l, rdlane(t[j], p), t[j]);
                });
              }
            });
          }
        }
      }
      __syncthreads();
    }
  });
  return singular;
}

// Solve T w = rhs with the factorisation above.  rhs: lane i holds component i (natural order);
// the result has component j in lane j.
template <typename TC, bool PIVOT>
__device__ __forceinline__ TC lu_solve(const TC (&t)[MP], TC w, int m, int lane, int mystep, int porder, TC udinv) {
  // (steps beyond m inside a partly used 16-block act on the identity padding of T and are harmless)
  static_for<MP / PB>([&](auto BK) LCP_INL {           // L y = P rhs
    constexpr int b = BK;
    if (PB * b < m) {
      static_for<PB>([&](auto KK) LCP_INL {
        constexpr int k = PB * b + KK;
        const int p = PIVOT ? rdlane_i(porder, k) : k;
        const TC yk = rdlane(w, p);
        const TC lk = (mystep > k) ? t[k] : (TC)0;
        w = fma(-lk, yk, w);
      });
    }
  });
  TC out = (TC)0;
  const TC wu = udinv;
  static_for<MP / PB>([&](auto BR) LCP_INL {           // U x = y
    constexpr int b = MP / PB - 1 - BR;
    if (PB * b < m) {
      static_for<PB>([&](auto JR) LCP_INL {
        constexpr int j = PB * b + PB - 1 - JR;
        const int p = PIVOT ? rdlane_i(porder, j) : j;
        const TC xj = rdlane(w * wu, p);
        const TC uj = (mystep < j) ? t[j] : (TC)0;
        w = fma(-uj, xj, w);
        out = (lane == j) ? xj : out;
      });
    }
  });
  return out;
}

// ---------------------------------------------------------------- contact-structured reduction
// For LCPs with the contact structure of physics/engines.py:67-73 (G = [Jc; Jf; 0] with the two friction
// rows of a contact being negatives of each other, F = [[0,0,0],[0,0,E],[mu,-E^T,0]]) the system
//     (R + diag(s/z)) dz = r,   dz = (a | b1,b2 per contact | g)
// is reduced EXACTLY to 2 nc unknowns: with u = b1 - b2, w = b1 + b2 and D = s/z,
//     rows f1+f2 and gamma of contact c involve only (w_c, g_c, a_c, u_c) and all their couplings are
//     diagonal, so (w_c, g_c) is eliminated per contact by a 2x2 solve:
//        Sp w + 2 g = (r1 + r2) - Sm u ,   -w + Dg g = rg - mu a ,   Sp = (D1+D2)/2, Sm = (D1-D2)/2
//     rows n and (f1 - f2)/2 then read
//        [ Wnn + Dn            Wnt                    ] [a]   [ rn                       ]
//        [ Wtn + Sm wa / 2     Wtt + (Sp + Sm wu) / 2 ] [u] = [ (r1 - r2)/2 - Sm w0 / 2  ]
//     with W = J P J^T for J = [Jc; Jt] (the rows n and f1 of R) and w = w0 + wa a + wu u.
// The reduced matrix is W plus positive diagonal terms (quasi-definite): elimination WITHOUT pivoting is
// backward stable at every PDIPM iterate (checked against the oracle: normwise backward error <= 1e-17
// up to cond(T) = 6e19, tests/test_reduction_algebra.py), because the +/- duplicated friction rows that
// make T numerically singular are removed analytically.  One eighth of the flops of the 4 nc system.

template <typename TC>
__device__ __forceinline__ bool lu32_factor(TC (&t)[MP], int nr, int lane, TC& udinv) {
  udinv = (TC)1;
  bool singular = false;
  static_for<NR / 8>([&](auto B8) LCP_INL {
    if (8 * B8 < nr) {
      static_for<8>([&](auto KK) LCP_INL {
        constexpr int k = 8 * B8 + KK;
        const TC piv = rdlane(t[k], k);
        singular = singular || (piv == (TC)0);
        const TC inv = fast_rcp(piv);
        const TC l = (lane > k) ? t[k] * inv : (TC)0;
        t[k] = (lane > k) ? l : t[k];
        udinv = (lane == k) ? inv : udinv;
        static_for<NR / 8>([&](auto C) LCP_INL {
          constexpr int c = C;
          if constexpr (c * 8 + 7 > k) {
            if (c * 8 < nr) {
              // broadcasts first, FMAs second: v_readlane has ~8 cycles of latency before a dependent VALU
              // can use the SGPR; batching hides it (measured 29 -> 21 cycles per pair)
              TC sv[8];
              static_for<8>([&](auto JJ) LCP_INL { constexpr int j = c * 8 + JJ; if constexpr (j > k) sv[JJ] = rdlane(t[j], k); });
              __builtin_amdgcn_sched_barrier(0);
              static_for<8>([&](auto JJ) LCP_INL { constexpr int j = c * 8 + JJ; if constexpr (j > k) t[j] = fma(-l, sv[JJ], t[j]); });
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        });
      });
    }
  });
  return singular;
}

template <typename TC>
__device__ __forceinline__ TC lu32_solve(const TC (&t)[MP], TC w, int nr, int lane, TC udinv) {
  static_for<NR / 8>([&](auto B8) LCP_INL {
    if (8 * B8 < nr) {
      static_for<8>([&](auto KK) LCP_INL {
        constexpr int k = 8 * B8 + KK;
        const TC yk = rdlane(w, k);
        const TC lk = (lane > k) ? t[k] : (TC)0;
        w = fma(-lk, yk, w);
      });
    }
  });
  static_for<NR / 8>([&](auto BR) LCP_INL {
    constexpr int b8 = NR / 8 - 1 - BR;
    if (8 * b8 < nr) {
      static_for<8>([&](auto JR) LCP_INL {
        constexpr int j = 8 * b8 + 7 - JR;
        const TC xj = rdlane(w * udinv, j);
        const TC uj = (lane < j) ? t[j] : (TC)0;
        w = fma(-uj, xj, w);
      });
    }
  });
  return w * udinv;
}

// Per-scene bookkeeping of the reduction.  Reduced lane r: r < nc -> a_r (normal), nc <= r < 2 nc -> u_{r-nc}.
template <typename TC>
struct Red {
  int nc, nr;
  bool isa, isu;
  int cu;                 // contact of this u-lane
  int i1, i2, ig;         // m-space lanes of f1, f2, gamma of contact cu (u-lanes)
  int back_src;           // m-space lane i takes its dz from this reduced lane
  int back_kind;          // 0: a (own lane)  1: b1  2: b2  3: g   4: padding
  TC mu;                  // friction coefficient of contact cu (u-lanes)
  // per factorisation (functions of D = s/z)
  TC Sp, Sm, Dg, idet, wa, wu;
  __device__ __forceinline__ void init(int nc_, int lane, int m) {
    nc = nc_; nr = 2 * nc_;
    isa = lane < nc; isu = (lane >= nc) && (lane < 2 * nc);
    cu = isu ? lane - nc : 0;
    i1 = nc + 2 * cu; i2 = i1 + 1; ig = 3 * nc + cu;
    if (lane < nc) { back_src = lane; back_kind = 0; }
    else if (lane < 3 * nc) { back_src = nc + ((lane - nc) >> 1); back_kind = 1 + ((lane - nc) & 1); }
    else if (lane < 4 * nc) { back_src = nc + (lane - 3 * nc); back_kind = 3; }
    else { back_src = lane; back_kind = 4; }
    mu = 0; Sp = Sm = Dg = idet = wa = wu = 0;
  }
  // D = s/z of this m-space lane
  __device__ __forceinline__ void prepare(TC D) {
    const TC D1 = shfl_t(D, i1), D2 = shfl_t(D, i2);
    Dg = shfl_t(D, ig);
    Sp = (TC)0.5 * (D1 + D2); Sm = (TC)0.5 * (D1 - D2);
    idet = (TC)1 / (Sp * Dg + (TC)2);
    wa = (TC)2 * mu * idet; wu = -Dg * Sm * idet;
  }
  // reduced matrix row of this lane from W (workspace) and the diagonal terms
  template <typename TI>
  __device__ __forceinline__ void build(TC (&t)[MP], const TC* W2, TC D, int lane) const {
    const TC addA = isa ? D : (isu ? (TC)0.5 * (Sp + Sm * wu) : (TC)1);
    const TC addB = isu ? (TC)0.5 * Sm * wa : (TC)0;
    const int colB = isu ? cu : -1;
    static_for<NR / 8>([&](auto B8) LCP_INL {
      if (8 * B8 < nr) {
        static_for<4>([&](auto JJ) LCP_INL {
          constexpr int j = 8 * B8 + 2 * JJ;
          TC a, c;
          load2(W2 + ((size_t)(j >> 1) * MP + lane) * 2, a, c);
          t[j] = a + ((lane == j) ? addA : (TC)0) + ((colB == j) ? addB : (TC)0);
          t[j + 1] = c + ((lane == j + 1) ? addA : (TC)0) + ((colB == j + 1) ? addB : (TC)0);
        });
      }
    });
  }
  // T^-1 hz through the reduced system (hz, result: m-space lanes)
  __device__ __forceinline__ TC solve(const TC (&t)[MP], TC hz, int lane, TC udinv) const {
    const TC r1 = shfl_t(hz, i1), r2 = shfl_t(hz, i2), rg = shfl_t(hz, ig);
    const TC w0 = (Dg * (r1 + r2) - (TC)2 * rg) * idet;
    const TC rhs = isa ? hz : (isu ? (TC)0.5 * (r1 - r2) - (TC)0.5 * Sm * w0 : (TC)0);
    const TC sol = lu32_solve<TC>(t, rhs, nr, lane, udinv);
    const TC ac = shfl_t(sol, cu);                       // a of this u-lane's contact
    const TC w = w0 + wa * ac + wu * sol;
    const TC gg = ((r1 + r2) - Sm * sol + Sp * (rg - mu * ac)) * idet;
    const TC b1 = (TC)0.5 * (w + sol), b2 = (TC)0.5 * (w - sol);
    const TC g1 = shfl_t(b1, back_src), g2 = shfl_t(b2, back_src), g3 = shfl_t(gg, back_src);
    return back_kind == 0 ? sol : (back_kind == 1 ? g1 : (back_kind == 2 ? g2 : (back_kind == 3 ? g3 : (TC)0)));
  }
};

// Gauss-Jordan inverse by one wave (same algorithm as gj_inverse<NT>, lane-indexed so that several waves
// of a workgroup can each invert their own matrix; the barriers are workgroup-wide and executed by all).
template <typename TC>
__device__ bool gj_inverse_wave(TC* a, int n, int* flag, int lane) {
  if (lane == 0) *flag = 0;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const TC piv = a[k * n + k];
    if (!(piv != (TC)0) || piv != piv) { if (lane == 0) *flag = 1; }
    const TC pinv = (TC)1 / piv;
    __syncthreads();
    for (int j = lane; j < n; j += 64) if (j != k) a[k * n + j] *= pinv;
    __syncthreads();
    for (int idx = lane; idx < n * n; idx += 64) {
      const int i = idx / n, j = idx - i * n;
      if (i != k && j != k) a[idx] -= a[i * n + k] * a[k * n + j];
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) a[i * n + k] = (i == k) ? pinv : -a[i * n + k] * pinv;
    __syncthreads();
  }
  return *flag == 0;
}

// ---------------------------------------------------------------- per-scene LDS block
template <typename TI, typename TC>
struct Lds {
  TI* Gc;    // [NZP][MP]   Gc[j*MP + i]  = G[i][j]
  TI* Gr;    // [MP][GRS]   Gr[i*GRS + j] = G[i][j]
  TI* Qt;    // [NZP][NZP]  Qt[k*NZP + j] = Q[j][k]
  TI* At;    // [EP][NZP]   At[a*NZP + k] = A[a][k]
  TC* Qit;   // [NZP][NZP]  Qit[k*NZP + j] = Qinv[j][k]
  TC* pan;   // [MP][PS]    LU panel work area (lu_factor)
  TC* Qrm;   // [nz][nz]    row-major Q^-1 (Gauss-Jordan work area, stride nz) - aliases `pan`
  TC* GAc;   // [EP][MP]    GAc[a*MP + i] = (G Q^-1 A^T)[i][a]
  TC* S11i;  // [e][e]      (A Q^-1 A^T)^-1, stride e
  TC* wbuf;  // [MP]
  int* flag;
};

// `with_panel`: the 8.7 KB LU panel is only needed by the general (pivoted 64x64) path; without it a
// scene needs 19 KB of LDS, i.e. 8 waves per CU instead of 6.
template <typename TI, typename TC>
__host__ __device__ inline size_t carve(Lds<TI, TC>& L, unsigned char* smem, bool with_panel = true) {
  unsigned char* q = smem;
  auto take = [&](size_t bytes) { unsigned char* r = q; q += (bytes + 15) & ~(size_t)15; return r; };
  L.Qit = (TC*)take(sizeof(TC) * NZP * NZP);
  L.pan = (TC*)take(with_panel ? sizeof(TC) * MP * (16 + 1) : sizeof(TC) * NZP * NZP);
  L.Qrm = L.pan;
  L.GAc = (TC*)take(sizeof(TC) * EP * MP);
  L.S11i = (TC*)take(sizeof(TC) * EP * EP);
  L.wbuf = (TC*)take(sizeof(TC) * MP);
  L.Gc = (TI*)take(sizeof(TI) * NZP * MP);
  L.Gr = (TI*)take(sizeof(TI) * MP * GRS);
  L.Qt = (TI*)take(sizeof(TI) * NZP * NZP);
  L.At = (TI*)take(sizeof(TI) * EP * NZP);
  L.flag = (int*)take(16);
  return (size_t)(q - smem);
}

// ---------------------------------------------------------------- F operators (wave level)
template <typename TI, typename TC>
struct FDenseW {
  const TI* F;     // [m,m] row-major input of this scene
  TI* Ft;          // lane-major copy in the workspace
  int m;
  __device__ __forceinline__ TC at(int i, int j) const { return (TC)F[(size_t)i * m + j]; }
  // (F z)_lane  using the lane-major copy (coalesced 16 B per lane)
  __device__ __forceinline__ TC Fz(TC z, int lane) const {
    TC acc = 0;
    const int nq = (m + 3) >> 2;
#pragma unroll 1
    for (int jj = 0; jj < nq; ++jj) {
      const TI* src = Ft + ((size_t)jj * MP + lane) * 4;
      TI f0, f1, f2, f3;
      load4(src, f0, f1, f2, f3);
      acc = fma((TC)f0, rdlane(z, jj * 4 + 0), acc);
      acc = fma((TC)f1, rdlane(z, jj * 4 + 1), acc);
      acc = fma((TC)f2, rdlane(z, jj * 4 + 2), acc);
      acc = fma((TC)f3, rdlane(z, jj * 4 + 3), acc);
    }
    return acc;
  }
  __device__ __forceinline__ void keep(int i, int j, TI v) const { Ft[((size_t)(j >> 2) * MP + i) * 4 + (j & 3)] = v; }
};

template <typename TC>
struct FContactW {
  int nc;
  TC mu;           // friction coefficient of the contact this lane's gamma row belongs to
  __device__ __forceinline__ TC at(int i, int j) const {       // engines.py:69-73
    if (i < nc) return (TC)0;
    if (i < 3 * nc) return (j >= 3 * nc && ((i - nc) >> 1) == (j - 3 * nc)) ? (TC)1 : (TC)0;
    const int c = i - 3 * nc;
    if (j < nc) return (j == c) ? mu : (TC)0;
    if (j < 3 * nc) return (((j - nc) >> 1) == c) ? (TC)-1 : (TC)0;
    return (TC)0;
  }
  __device__ __forceinline__ TC Fz(TC z, int lane) const {
    const int i = lane;
    const int c = i - 3 * nc;
    const int src_g = 3 * nc + ((i - nc) >> 1);                  // gamma of a friction row
    const TC zg = shfl_t(z, (i >= nc && i < 3 * nc) ? src_g : 0);
    const bool gam = (i >= 3 * nc) && (i < 4 * nc);
    const TC zn = shfl_t(z, gam ? c : 0);
    const TC zf0 = shfl_t(z, gam ? nc + 2 * c : 0);
    const TC zf1 = shfl_t(z, gam ? nc + 2 * c + 1 : 0);
    if (i >= nc && i < 3 * nc) return zg;
    if (gam) return mu * zn - (zf0 + zf1);
    return (TC)0;
  }
  __device__ __forceinline__ void keep(int, int, float) const {}
};

// ---------------------------------------------------------------- scene-level linear algebra
template <typename TI, typename TC>
struct Ops {
  Lds<TI, TC> L;
  int nz, m, e, lane;
  bool diagq;          // Q is diagonal (always true for assembled contact scenes): Q v and Q^-1 v are one multiply
  TC qd, qid;          // Q[j][j], Qinv[j][j] for j = lane & 15
  TI grow[NZP];        // row `lane` of G, register-resident (G v without LDS traffic)
  __device__ __forceinline__ void cache_rows() {
    static_for<NZP>([&](auto J) LCP_INL { grow[J] = L.Gc[J * MP + lane]; });
    const int j = lane & 15;
    qd = (TC)L.Qt[j * NZP + j]; qid = L.Qit[j * NZP + j];
  }
  // All loops below are ROLLED on purpose (run-time trip counts, v_readlane with a run-time lane select):
  // these products are small next to the LU, and compact code is what keeps the kernel in the I-cache.
  // m-space <- x-space : (G v)_i
  __device__ __forceinline__ TC Gv(TC v) const {
    TC acc = 0;                                        // columns >= nz are zero in grow and in v
    static_for<NZP>([&](auto J) LCP_INL { acc = fma((TC)grow[J], rdlane(v, J), acc); });
    return acc;
  }
  // x-space <- m-space : (G^T w)_j, j = lane & 15 (replicated in the four lane groups)
  __device__ __forceinline__ TC Gtw(TC w) const {
    __syncthreads();
    L.wbuf[lane] = (lane < m) ? w : (TC)0;
    __syncthreads();
    const int j = lane & 15, q = lane >> 4;
    TC acc = 0;
    const TI* g = L.Gr + (q * 16) * GRS + j;
    const TC* wb = L.wbuf + q * 16;
#pragma unroll 4
    for (int ii = 0; ii < 16; ++ii) acc = fma((TC)g[ii * GRS], wb[ii], acc);
    acc += shfl_xor_t(acc, 16);
    acc += shfl_xor_t(acc, 32);
    return acc;
  }
  __device__ __forceinline__ TC Qiv(TC v) const {
    if (diagq) return qid * v;
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll 1
    for (int k = 0; k < nz; ++k) acc = fma(L.Qit[k * NZP + j], rdlane(v, k), acc);
    return acc;
  }
  __device__ __forceinline__ TC Qv(TC v) const {
    if (diagq) return qd * v;
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll 1
    for (int k = 0; k < nz; ++k) acc = fma((TC)L.Qt[k * NZP + j], rdlane(v, k), acc);
    return acc;
  }
  // e-space <- x-space : (A v)_a, a = lane & 7
  __device__ __forceinline__ TC Av(TC v) const {
    const int a = lane & 7;
    TC acc = 0;
#pragma unroll 1
    for (int k = 0; k < nz; ++k) acc = fma((TC)L.At[a * NZP + k], rdlane(v, k), acc);
    return acc;
  }
  // x-space <- e-space : (A^T y)_j
  __device__ __forceinline__ TC Aty(TC y) const {
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll 1
    for (int a = 0; a < e; ++a) acc = fma((TC)L.At[a * NZP + j], rdlane(y, a), acc);
    return acc;
  }
  // m-space <- e-space : (GA t)_i
  __device__ __forceinline__ TC GAt(TC t) const {
    TC acc = 0;
#pragma unroll 1
    for (int a = 0; a < e; ++a) acc = fma(L.GAc[a * MP + lane], rdlane(t, a), acc);
    return acc;
  }
  // e-space <- m-space : (GA^T w)_a
  __device__ __forceinline__ TC GAtw(TC w) const {
    TC out = 0;
    const TC wm = (lane < m) ? w : (TC)0;
#pragma unroll 1
    for (int a = 0; a < e; ++a) {
      const TC sm = wave_sum(L.GAc[a * MP + lane] * wm);
      if ((lane & 7) == a) out = sm;
    }
    return out;
  }
  __device__ __forceinline__ TC S11v(TC v) const {
    const int a = lane & 7;
    TC acc = 0;
#pragma unroll 1
    for (int c = 0; c < e; ++c) acc = fma((a < e) ? L.S11i[a * e + c] : (TC)0, rdlane(v, c), acc);
    return acc;
  }
};

// get_step for (z,dz),(s,ds) at once: min(step(z,dz), step(s,ds)), pdipm.py:182-186 per scene.
template <typename TC>
__device__ __forceinline__ TC step_pair(TC z, TC dz, TC s, TC ds, bool valid) {
  TC rzz = valid ? -z / dz : -inf_of<TC>();
  TC rss = valid ? -s / ds : -inf_of<TC>();
  TC mz = rzz, ms = rss;
  wave_pmax2(mz, ms);
  const TC fz = (mz > (TC)1) ? mz : (TC)1;
  const TC fs = (ms > (TC)1) ? ms : (TC)1;
  TC az = valid ? ((dz > (TC)0) ? fz : rzz) : inf_of<TC>();
  TC as = valid ? ((ds > (TC)0) ? fs : rss) : inf_of<TC>();
  wave_pmin2(az, as);
  return pmin(az, as);
}

// ---------------------------------------------------------------- input stages
template <typename TI, typename TC>
__device__ __forceinline__ void zero_lds_inputs(const Lds<TI, TC>& L, int lane) {
  for (int i = lane; i < NZP * MP; i += 64) L.Gc[i] = (TI)0;
  for (int i = lane; i < MP * GRS; i += 64) L.Gr[i] = (TI)0;
  for (int i = lane; i < NZP * NZP; i += 64) { L.Qt[i] = (TI)0; L.Qit[i] = (TC)0; }
  for (int i = lane; i < EP * NZP; i += 64) L.At[i] = (TI)0;
  for (int i = lane; i < EP * MP; i += 64) L.GAc[i] = (TC)0;
  for (int i = lane; i < EP * EP; i += 64) L.S11i[i] = (TC)0;
}

// dense inputs -> LDS; returns p (x-space), h (m-space), b (e-space) in registers
template <typename TI, typename TC>
__device__ __forceinline__ void load_dense(const Lds<TI, TC>& L, const FwdArgs& P, int scene, int lane,
                                           TC& p, TC& h, TC& b) {
  const int nz = P.nz, m = P.m, e = P.e;
  zero_lds_inputs(L, lane);
  __syncthreads();
  const TI* Q = (const TI*)P.Q + (size_t)scene * nz * nz;
  const TI* G = (const TI*)P.G + (size_t)scene * m * nz;
  for (int idx = lane; idx < nz * nz; idx += 64) {
    const int r = idx / nz, c = idx - r * nz;
    const TI q = Q[idx];
    L.Qt[c * NZP + r] = q;               // Qt[k][j] = Q[j][k]
    L.Qrm[r * nz + c] = (TC)q;
  }
  for (int idx = lane; idx < m * nz; idx += 64) {
    const int i = idx / nz, j = idx - i * nz;
    const TI g = G[idx];
    L.Gc[j * MP + i] = g;
    L.Gr[i * GRS + j] = g;
  }
  if (e > 0) {
    const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
    for (int idx = lane; idx < e * nz; idx += 64) { const int a = idx / nz, k = idx - a * nz; L.At[a * NZP + k] = A[idx]; }
    b = ((lane & 7) < e) ? (TC)((const TI*)P.b)[(size_t)scene * e + (lane & 7)] : (TC)0;
  } else {
    b = (TC)0;
  }
  p = ((lane & 15) < nz) ? (TC)((const TI*)P.p)[(size_t)scene * nz + (lane & 15)] : (TC)0;
  h = (lane < m) ? (TC)((const TI*)P.h)[(size_t)scene * m + lane] : (TC)0;
  __syncthreads();
}

// Contact list -> LDS (engines.py:31-32,50-74; world.py:144-234).  All Jacobian entries are formed in
// I/O precision with the same expressions as lcp_assemble_kernel, so the fused and the
// assemble-then-solve paths see the same LCP.
template <typename TI, typename TC>
__device__ __forceinline__ void assemble_scene(const Lds<TI, TC>& L, const StepArgs& P, int scene, int lane,
                                               TC& p, TC& h, TC& b, TC& mu_lane) {
  const int nb = P.nb, nc = P.nc, nz = 3 * nb, e = P.e;
  zero_lds_inputs(L, lane);
  __syncthreads();
  const TI* Md = (const TI*)P.Mdiag + (size_t)scene * nz;
  const TI* vv = (const TI*)P.v + (size_t)scene * nz;
  const TI* ff = (const TI*)P.f + (size_t)scene * nz;
  const TI* rest = (const TI*)P.rest + (size_t)scene * nb;
  const TI* fric = (const TI*)P.fric + (size_t)scene * nb;
  const TI* cn = (const TI*)P.c_n + (size_t)scene * nc * 2;
  const TI* c1 = (const TI*)P.c_p1 + (size_t)scene * nc * 2;
  const TI* c2 = (const TI*)P.c_p2 + (size_t)scene * nc * 2;
  const int32_t* i1 = P.c_i1 + (size_t)scene * nc;
  const int32_t* i2 = P.c_i2 + (size_t)scene * nc;
  for (int idx = lane; idx < nz * nz; idx += 64) {
    const int r = idx / nz, c = idx - r * nz;
    const TI q = (r == c) ? Md[r] : (TI)0;
    L.Qt[c * NZP + r] = q;
    L.Qrm[idx] = (TC)q;
  }
  if (e > 0) {
    const TI* Je = (const TI*)P.Je + (size_t)scene * e * nz;
    for (int idx = lane; idx < e * nz; idx += 64) { const int a = idx / nz, k = idx - a * nz; L.At[a * NZP + k] = Je[idx]; }
  }
  b = (TC)0;
  {
    const int j = lane & 15;
    p = (j < nz) ? (TC)momentum_entry<TI>(Md[j], vv[j], (TI)P.dt, ff[j]) : (TC)0;      // engines.py:32
  }
  TI hrow = (TI)0;
  if (lane < nc) {
    const int c = lane;
    const ContactRows<TI> r = make_contact<TI>(cn, c1, c2, i1, i2, rest, fric, vv, c);
    const int rn = c, rf0 = nc + 2 * c, rf1 = nc + 2 * c + 1;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int col = (q < 3) ? 3 * r.b1 + q : 3 * r.b2 + (q - 3);
      L.Gc[col * MP + rn] = r.jn[q];    L.Gr[rn * GRS + col] = r.jn[q];
      L.Gc[col * MP + rf0] = r.jf[q];   L.Gr[rf0 * GRS + col] = r.jf[q];
      L.Gc[col * MP + rf1] = -r.jf[q];  L.Gr[rf1 * GRS + col] = -r.jf[q];
    }
    hrow = r.h;
  }
  h = (TC)hrow;
  {
    const int c = lane - 3 * nc;
    mu_lane = (c >= 0 && c < nc) ? (TC)((TI)0.5 * (fric[i1[c]] + fric[i2[c]])) : (TC)0;   // world.py:213-224
  }
  __syncthreads();
}

// Does the dense LCP have the contact structure of engines.py:67-73 ?  (wave-uniform answer)
//   G rows >= 3 nc are zero, friction rows come in +/- pairs, F = [[0,0,0],[0,0,E],[mu,-E^T,0]].
// mu_g: friction coefficient for this lane if it is a gamma row.
template <typename TI, typename TC>
__device__ __forceinline__ bool detect_structure(const Lds<TI, TC>& L, const TI* F, int nz, int m, int lane, TC& mu_g) {
  mu_g = 0;
  if ((m & 3) != 0 || m > 4 * (NR / 2)) return false;
  const int nc = m >> 2;
  bool ok = true;
  if (lane < m) {
    const int i = lane;
    const TI* g = L.Gr + i * GRS;
    if (i >= 3 * nc) { for (int j = 0; j < nz; ++j) ok = ok && (g[j] == (TI)0); }
    else if (i >= nc && ((i - nc) & 1)) { for (int j = 0; j < nz; ++j) ok = ok && (g[j] == -g[j - GRS]); }
    const TI* f = F + (size_t)i * m;
    const int cf = (i - nc) >> 1, cg = i - 3 * nc;
    for (int j = 0; j < m; ++j) {
      const TI v = f[j];
      TI want = (TI)0;
      if (i >= nc && i < 3 * nc) want = (j == 3 * nc + cf) ? (TI)1 : (TI)0;
      else if (i >= 3 * nc) {
        if (j == cg) { want = v; mu_g = (TC)v; }
        else if (j == nc + 2 * cg || j == nc + 2 * cg + 1) want = (TI)-1;
      }
      ok = ok && (v == want);
    }
  }
  return __all(ok);
}

// row `row` of G times Q^-1 (gq) and of (G Q^-1 A^T) S11^-1 (cc); ga_out (optional) receives G Q^-1 A^T of that row
template <typename TI, typename TC>
__device__ __forceinline__ void row_products(const Lds<TI, TC>& L, int row, bool valid, int nz, int e, bool have_s11,
                                             TC (&gq)[NZP], TC (&cc)[EP], TC (&ga)[EP]) {
#pragma unroll
  for (int k = 0; k < NZP; ++k) gq[k] = 0;
#pragma unroll
  for (int a = 0; a < EP; ++a) { cc[a] = 0; ga[a] = 0; }
  if (!valid) return;
  const TI* g = L.Gr + row * GRS;
#pragma unroll 1
  for (int j = 0; j < nz; ++j) {
    const TC gj = (TC)g[j];
    const TC* qrow = L.Qrm + j * nz;
#pragma unroll
    for (int k = 0; k < NZP; ++k) gq[k] = fma(gj, (k < nz) ? qrow[k] : (TC)0, gq[k]);
  }
  if (e > 0) {
#pragma unroll
    for (int a = 0; a < EP; ++a) {
      if (a < e) {
        TC acc = 0;
#pragma unroll
        for (int k = 0; k < NZP; ++k) acc = fma(gq[k], (TC)L.At[a * NZP + k], acc);
        ga[a] = acc;
      }
    }
    if (have_s11) {
#pragma unroll
      for (int a = 0; a < EP; ++a) {
        if (a < e) {
          TC acc = 0;
#pragma unroll
          for (int c = 0; c < EP; ++c) if (c < e) acc = fma(ga[c], L.S11i[c * e + a], acc);
          cc[a] = acc;
        }
      }
    }
  }
}

// pre_factor_kkt (pdipm.py:357-408): Q^-1, G Q^-1 A^T, (A Q^-1 A^T)^-1 into LDS; then either the dense
// R = G Q^-1 G^T + F - GA S11^-1 GA^T (+ the lane-major copy of F) or, for contact-structured problems,
// only W = J P J^T (rows / columns n and f1 of R) - both into the HBM workspace, two columns per lane-store.
template <typename TI, typename TC, typename FT>
__device__ __forceinline__ int prefactor(const Lds<TI, TC>& L, const Ws<TI, TC>& W, const FT& F, int nz, int m, int e,
                                         int lane, bool structured) {
  int status = 0;
  if (!gj_inverse_wave(L.Qrm, nz, L.flag, lane)) status |= LCP_ST_SINGULAR_Q;
  for (int idx = lane; idx < nz * nz; idx += 64) { const int r = idx / nz, c = idx - r * nz; L.Qit[c * NZP + r] = L.Qrm[idx]; }
  __syncthreads();
  TC gq[NZP], cc[EP], ga[EP];
  row_products<TI, TC>(L, lane, lane < m, nz, e, false, gq, cc, ga);
  if (e > 0) {
#pragma unroll
    for (int a = 0; a < EP; ++a) if (a < e) L.GAc[a * MP + lane] = ga[a];
    if (lane < e * e) {                               // S11 = A Q^-1 A^T, one entry per lane
      const int a = lane / e, c = lane - a * e;
      TC acc = 0;
      for (int k = 0; k < nz; ++k) {
        TC aq = 0;
        for (int j = 0; j < nz; ++j) aq = fma((TC)L.At[a * NZP + j], L.Qrm[j * nz + k], aq);
        acc = fma(aq, (TC)L.At[c * NZP + k], acc);
      }
      L.S11i[lane] = acc;
    }
    __syncthreads();
    if (!gj_inverse_wave(L.S11i, e, L.flag, lane)) status |= LCP_ST_SINGULAR_S11;
  }
  const int nc = m >> 2;
  // which full row does this lane produce ?  dense: its own; structured: reduced lane r -> n_r or f1_{r-nc}
  const int nrows = structured ? 2 * nc : m;
  const int myrow = structured ? ((lane < nc) ? lane : nc + 2 * (lane - nc)) : lane;
  row_products<TI, TC>(L, myrow, lane < nrows, nz, e, true, gq, cc, ga);
  const int ncols = structured ? NR : MP;
#pragma unroll 1
  for (int j0 = 0; j0 < ncols; j0 += 2) {
    TC r2[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = j0 + jj;
      const int col = structured ? ((j < nc) ? j : nc + 2 * (j - nc)) : j;      // full column index
      TC val = 0;
      if (lane < nrows && j < nrows) {
        const TI* gcol = L.Gr + col * GRS;
#pragma unroll
        for (int k = 0; k < NZP; ++k) val = fma(gq[k], (TC)gcol[k], val);
#pragma unroll
        for (int a = 0; a < EP; ++a) if (a < e) val = fma(-cc[a], L.GAc[a * MP + col], val);
        if (!structured) { const TC f = F.at(lane, j); F.keep(lane, j, (TI)f); val += f; }
      } else if (!structured) {
        F.keep(lane, j, (TI)0);
      }
      r2[jj] = val;
    }
    store2(W.R2 + ((size_t)(j0 >> 1) * MP + lane) * 2, r2[0], r2[1]);
  }
  // things the backward kernel needs
  for (int i = lane; i < NZP * NZP; i += 64) W.Qit[i] = L.Qit[i];
  for (int i = lane; i < EP * MP; i += 64) W.GAc[i] = L.GAc[i];
  for (int i = lane; i < EP * EP; i += 64) W.S11i[i] = L.S11i[i];
  if (lane == 0 && structured && W.meta[0] == (TC)0) W.meta[0] = (TC)1;   // (fused input: no classification ran)
  __threadfence_block();
  __syncthreads();
  return status;
}

// T = R + diag(1/d) from the workspace into registers (pdipm.py:427-429)
template <typename TI, typename TC>
__device__ __forceinline__ void load_T(TC (&t)[MP], const Ws<TI, TC>& W, TC dinv, int m, int lane) {
  static_for<MP / PB>([&](auto BK) LCP_INL {
    constexpr int b = BK;
    if (PB * b < m) {
      static_for<PB / 2>([&](auto JJ) LCP_INL {
        constexpr int j = PB * b + 2 * JJ;
        const TC* src = W.R2 + ((size_t)(j >> 1) * MP + lane) * 2;      // zero beyond m (prefactor pads)
        TC a, c;
        load2(src, a, c);
        t[j] = a + ((lane == j) ? dinv : (TC)0);
        t[j + 1] = c + ((lane == j + 1) ? dinv : (TC)0);
      });
    } else {
      static_for<PB>([&](auto JJ) LCP_INL { constexpr int j = PB * b + JJ; t[j] = (lane == j) ? dinv : (TC)0; });
    }
  });
}

// solve_kkt (pdipm.py:325-354) at wave level.  The T-solve goes through the reduced system when the
// scene is contact-structured, through the pivoted 64x64 factorisation otherwise.
template <typename TI, typename TC, bool PIVOT, bool ALWAYS_STRUCT>
__device__ __forceinline__ void solve_kkt(const Ops<TI, TC>& O, const TC (&t)[MP], int mystep, int porder, TC udinv,
                                          const Red<TC>& RD, bool structured,
                                          TC d, TC rx, TC rs, TC rz, TC ry, TC& ox, TC& os, TC& oz, TC& oy) {
  const int m = O.m, e = O.e, lane = O.lane;
  const TC v = O.Qiv(rx);                                                  // :333
  TC hz = (lane < m) ? (O.Gv(v) + rs / d - rz) : (TC)0;                    // :334-340
  TC hy = 0;
  if (e > 0) {
    hy = O.Av(v) - ry;
    hz -= O.GAt(O.S11v(hy));
    if (lane >= m) hz = 0;
  }
  TC wz;                                                                   // T^-1 (...)
  if constexpr (ALWAYS_STRUCT) wz = RD.solve(t, hz, lane, udinv);
  else wz = lu_solve<TC, PIVOT>(t, hz, m, lane, mystep, porder, udinv);
  TC dy = 0;
  if (e > 0) dy = -O.S11v(hy - O.GAtw(wz));                                // dy = -wy
  const TC dz = (lane < m) ? -wz : (TC)0;                                  // :342
  os = (lane < m) ? (-rs - dz) / d : (TC)0;                                // :347,350
  oz = dz; oy = dy;
  TC g1 = -rx - O.Gtw(dz);                                                 // :344-346
  if (e > 0) g1 -= O.Aty(dy);
  ox = O.Qiv(g1);                                                          // :349
}

// Build + factor T (or its reduction) for D^-1 = dinv.  Returns "exact zero pivot".
template <typename TI, typename TC, bool PIVOT, bool STRUCT>
__device__ __forceinline__ bool factor(TC (&t)[MP], const Ops<TI, TC>& O, const Ws<TI, TC>& W, Red<TC>& RD, bool,
                                       TC dinv, int& mystep, int& porder, TC& udinv) {
  const int m = O.m, lane = O.lane;
  if constexpr (STRUCT) {
    RD.prepare(dinv);
    RD.template build<TI>(t, W.R2, dinv, lane);
    mystep = 0; porder = lane;
    return lu32_factor<TC>(t, RD.nr, lane, udinv);
  } else {
    load_T<TI, TC>(t, W, dinv, m, lane);
    return lu_factor<TC, PIVOT>(t, O.L.pan, m, lane, mystep, porder, udinv);
  }
}

// ---------------------------------------------------------------- structure classification
// One wave per scene: does the dense LCP have the contact structure of engines.py:67-73 ?  Writes
// meta[0] = 1/0 and the per-contact friction coefficients meta[1 + c] into the scene's workspace; the
// single-path solver kernels below read the flag first and leave immediately when it is not theirs.
template <typename TI, typename TC>
__global__ void __launch_bounds__(256) lcp_classify_wave(FwdArgs P) {
  const int scene = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.tag) *P.tag = P.tag_value;   // workspace trailer: which kernel family laid it out
  if (scene >= P.B) return;
  const int nz = P.nz, m = P.m;
  Ws<TI, TC> W(P.ws, scene);
  bool ok = ((m & 3) == 0) && (m <= 4 * (NR / 2));
  const int nc = m >> 2;
  if (ok) {
    // G: lanes over columns, loop over rows (coalesced)
    const TI* G = (const TI*)P.G + (size_t)scene * m * nz;
    // (the tests are accumulated with `&`, not `&&`: a short-circuit would make every load wait for the previous compare)
    int good = 1;
    // (round 4: the kernel was bound by the LATENCY of its loads - eight in flight per wavefront, ~15 round trips, 24 us for the 88 MB
    //  of a 4096 x 16-contact batch; with 32 rows of F and 16 of G in flight it is a handful of round trips)
    if (lane < nz) {
#pragma unroll 8
      for (int i = nc; i < 3 * nc; i += 2) good &= (G[(i + 1) * nz + lane] == -G[i * nz + lane]) ? 1 : 0;
#pragma unroll 16
      for (int i = 3 * nc; i < m; ++i) good &= (G[i * nz + lane] == (TI)0) ? 1 : 0;
    }
    // F: lane j holds column j of every row in turn (one coalesced 4 m-byte read per row)
    const TI* F = (const TI*)P.F + (size_t)scene * m * m;
    if (lane < m) {
      const int j = lane;
#pragma unroll 32
      for (int i = 0; i < m; ++i) {
        const TI v = F[(size_t)i * m + j];
        TI want = (TI)0;
        if (i >= nc && i < 3 * nc) want = (j == 3 * nc + ((i - nc) >> 1)) ? (TI)1 : (TI)0;
        else if (i >= 3 * nc) {
          const int cg = i - 3 * nc;
          if (j == cg) want = v;
          else if (j == nc + 2 * cg || j == nc + 2 * cg + 1) want = (TI)-1;
        }
        good &= (v == want) ? 1 : 0;
      }
    }
    ok = good != 0;
  }
  bool qdiag = true;
  {
    const TI* Q = (const TI*)P.Q + (size_t)scene * nz * nz;
    int qd = 1;
    for (int idx = lane; idx < nz * nz; idx += 64) { const int r = idx / nz, c = idx - r * nz; qd &= (r == c || Q[idx] == (TI)0) ? 1 : 0; }
    qdiag = qd != 0;
  }
  qdiag = __all(qdiag);
  const bool all_ok = __all(ok);
  // meta[0]: 0 = general, 1 = contact-structured, 2 = contact-structured with diagonal Q (quad-kernel eligible)
  if (lane == 0) { W.meta[0] = all_ok ? (qdiag ? (TC)2 : (TC)1) : (TC)0; W.meta[18] = qdiag ? (TC)1 : (TC)0; }
  if (all_ok && lane < nc) W.meta[1 + lane] = (TC)((const TI*)P.F)[(size_t)scene * m * m + (size_t)(3 * nc + lane) * m + lane];
}

// ---------------------------------------------------------------- the forward kernel
// STRUCT = contact-structured path (reduced 2 nc system); FUSED implies STRUCT.  For dense inputs both
// instantiations are launched and each scene is served by the one its classification flag selects.
template <typename TI, typename TC, bool PIVOT, bool FUSED, bool STRUCT>
__device__ __forceinline__ void fwd_wave_body(const FwdArgs& P, const StepArgs& SP, int lds_per_wave, int skip2) {
  static_assert(STRUCT || !FUSED, "the fused step is always contact-structured");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int scene = blockIdx.x * WPB + wave;
  if (FUSED && blockIdx.x == 0 && threadIdx.x == 0 && SP.tag) *SP.tag = SP.tag_value;   // (dense inputs: lcp_classify_wave wrote it)
  if (scene >= (FUSED ? SP.B : P.B)) return;            // whole wave leaves; s_barrier ignores terminated waves
  unsigned char* smem = smem_all + (size_t)wave * lds_per_wave;
  const int nz = FUSED ? 3 * SP.nb : P.nz, m = FUSED ? 4 * SP.nc : P.m, e = FUSED ? SP.e : P.e;
  const int max_iter = FUSED ? SP.max_iter : P.max_iter, lim = FUSED ? SP.lim : P.lim;
  const TC eps = (TC)(FUSED ? SP.eps : P.eps);
  Ops<TI, TC> O;
  carve(O.L, smem, !STRUCT);
  O.nz = nz; O.m = m; O.e = e; O.lane = lane;
  Ws<TI, TC> W(FUSED ? SP.ws : P.ws, scene);
  const bool vm = lane < m;
  TC p, h, b, mu_lane = 0;
  int status;
  FDenseW<TI, TC> Fd{FUSED ? nullptr : (const TI*)P.F + (size_t)scene * m * m, W.Ft, m};
  constexpr bool structured = STRUCT;
  if (!FUSED) {                                                    // not this kernel's scene ?
    if ((W.meta[0] != (TC)0) != STRUCT) return;
    if (STRUCT && skip2 && W.meta[0] == (TC)2) return;               // served by the quad kernel
  }
  const int nc = m >> 2;
  Red<TC> RD;
  RD.init(nc, lane, m);
  if (FUSED) {
    assemble_scene<TI, TC>(O.L, SP, scene, lane, p, h, b, mu_lane);
    RD.mu = shfl_t(mu_lane, RD.isu ? RD.ig : 0);                    // mu of a u-lane's contact sits in its gamma lane
    if (!RD.isu) RD.mu = 0;
    if (RD.isu) W.meta[1 + RD.cu] = RD.mu;
  } else {
    load_dense<TI, TC>(O.L, P, scene, lane, p, h, b);
    if (STRUCT) {
      mu_lane = (lane >= 3 * nc && lane < m) ? W.meta[1 + lane - 3 * nc] : (TC)0;
      RD.mu = RD.isu ? W.meta[1 + RD.cu] : (TC)0;
    }
  }
  FContactW<TC> Fc{nc, mu_lane};
  if (STRUCT) status = prefactor<TI, TC>(O.L, W, Fc, nz, m, e, lane, true);
  else status = prefactor<TI, TC>(O.L, W, Fd, nz, m, e, lane, false);
  O.diagq = FUSED ? true : (W.meta[18] != (TC)0);
  O.cache_rows();

  TC t[MP];
  int mystep, porder;
  TC udinv;
  TC x = 0, s = 1, z = 1, y = 0, d = 1;
  TC bx = 0, bs = 1, bz = 1, by = 0;
  TC best_resid = inf_of<TC>();
  bool have_best = false;
  int n_not = 0, iters = 0;
  double* trace = (!FUSED && P.trace) ? P.trace + (size_t)scene * 4 * max_iter : nullptr;

#ifdef LCP_W64_PROFILE
  long long pc_res = 0, pc_fac = 0, pc_sol = 0, pc_mid = 0;
#define LCP_TICK(var) { const long long now_ = clock64(); var += now_ - tick_; tick_ = now_; }
  long long tick_ = clock64();
#else
#define LCP_TICK(var)
#endif
#pragma unroll 1
  for (int it = -1; it < max_iter; ++it) {
    TC rx, rs, rz, ry, mu = 0, resid = 0;
    if (it < 0) {                                                           // init: (p, 0, -h, -b), d = 1 (:57-63)
      rx = p; rs = 0; rz = -h; ry = -b; d = 1;
    } else {                                                                // residuals (:82-96)
      rx = O.Gtw(z) + O.Qv(x) + p;
      if (e > 0) rx += O.Aty(y);
      rs = z;
      TC fz;
      if (STRUCT) fz = Fc.Fz(z, lane);
      else fz = Fd.Fz(z, lane);
      rz = vm ? (O.Gv(x) + s - h - fz) : (TC)0;
      ry = (e > 0) ? (O.Av(x) - b) : (TC)0;
      TC n_rx = (lane < nz) ? rx * rx : (TC)0, n_rz = rz * rz;
      TC n_ry = (lane < e) ? ry * ry : (TC)0, sz = vm ? s * z : (TC)0;
      wave_sum2(n_rx, n_rz);
      wave_sum2(n_ry, sz);
      mu = sz / (TC)m; mu = mu < 0 ? -mu : mu;
      resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + (TC)m * mu;
      d = vm ? z / s : (TC)1;                                               // (:98)
    }
    LCP_TICK(pc_res)
    const bool singular = factor<TI, TC, PIVOT, STRUCT>(t, O, W, RD, structured, vm ? (TC)1 / d : (TC)1,
                                                       mystep, porder, udinv);                // (:99-100)
    LCP_TICK(pc_fac)
    if (it >= 0) {
      ++iters;
      if (trace && lane == 0) { trace[4 * it] = (double)resid; trace[4 * it + 1] = (double)mu; }
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; break; }      // except: return best (:99-102)
      const bool improved = !have_best || (resid < best_resid);            // (:107-132)
      if (improved) { best_resid = resid; n_not = 0; have_best = true; bx = x; bs = s; bz = z; by = y; }
      else ++n_not;
      if (n_not == lim || best_resid < eps || mu > mu_limit<TC>()) break;  // (:133)
      // the iterate this pass would produce is never evaluated (:176-179): dead work, skipped (kept when the
      // per-iteration trace is requested, which also records this pass's sigma and step length)
      if (it == max_iter - 1 && !trace) break;
    }
    // one call site for the KKT solve (code size): pass 0 = init / affine, pass 1 = corrector
    TC ax = 0, as_ = 0, az = 0, ay = 0;
    const int npass = (it < 0) ? 1 : 2;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      TC ox, os, oz, oy;
      LCP_TICK(pc_mid)
      solve_kkt<TI, TC, PIVOT, STRUCT>(O, t, mystep, porder, udinv, RD, structured, d, rx, rs, rz, ry, ox, os, oz, oy);
      LCP_TICK(pc_sol)
      if (it < 0) {
        x = ox; s = os; z = oz; y = oy;                                     // (:60-63)
        TC smin = vm ? s : inf_of<TC>(), zmin = vm ? z : inf_of<TC>();
        wave_pmin2(smin, zmin);
        if (smin <= (TC)0) s = s - smin + (TC)1;                            // (:66-75)
        if (zmin <= (TC)0) z = z - zmin + (TC)1;
        if (!vm) { s = 1; z = 1; }
      } else if (pass == 0) {
        ax = ox; as_ = os; az = oz; ay = oy;                                // affine direction (:138-139)
        const TC alpha = pmin(step_pair(z, az, s, as_, vm), (TC)1);        // (:142-144)
        TC t3 = vm ? (s + alpha * as_) * (z + alpha * az) : (TC)0, t4 = vm ? s * z : (TC)0;
        wave_sum2(t3, t4);
        const TC r3 = t3 / t4, sig = r3 * r3 * r3;                          // (:146-150)
        if (trace && lane == 0) trace[4 * it + 2] = (double)sig;
        rx = 0; rz = 0; ry = 0;
        rs = vm ? (-mu * sig + as_ * az) / s : (TC)0;                       // (:153)
      } else {
        const TC cx = ox + ax, cs = os + as_, cz = oz + az, cy = oy + ay;   // (:160-163)
        const TC alpha = pmin((TC)0.999 * step_pair(z, cz, s, cs, vm), (TC)1);   // (:164-166)
        if (trace && lane == 0) trace[4 * it + 3] = (double)alpha;
        x += alpha * cx; y += alpha * cy;                                   // (:171-174)
        if (vm) { s += alpha * cs; z += alpha * cz; }
      }
    }
  }

#ifdef LCP_W64_PROFILE
  if (trace && lane == 0) { trace[0] = (double)pc_res; trace[1] = (double)pc_fac; trace[2] = (double)pc_sol; trace[3] = (double)pc_mid; trace[4] = (double)iters; }
#endif
  // outputs: best iterate (x-space lanes < nz, m-space lanes < m, e-space lanes < e)
  int bad = 0;
  if (lane < nz) { W.x[lane] = bx; bad |= (bx != bx); }
  if (vm) { W.s[lane] = bs; W.z[lane] = bz; bad |= (bs != bs) | (bz != bz); }
  if (lane < e) W.y[lane] = by;
  if (__any(bad)) status |= LCP_ST_NAN;
  if (FUSED) {
    const TI* pos = (const TI*)SP.pos + (size_t)scene * nz;
    if (lane < nz) {
      const TC nv = -bx;                                                    // engines.py:76-77
      ((TI*)SP.v_new)[(size_t)scene * nz + lane] = (TI)nv;
      ((TI*)SP.p_new)[(size_t)scene * nz + lane] = (TI)((TC)pos[lane] + nv * (TC)SP.dt);   // bodies.py:81
    }
    if (vm && SP.z) ((TI*)SP.z)[(size_t)scene * m + lane] = (TI)bz;
    if (vm && SP.s) ((TI*)SP.s)[(size_t)scene * m + lane] = (TI)bs;
    if (lane < e && SP.y) ((TI*)SP.y)[(size_t)scene * e + lane] = (TI)by;
    if (lane == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
  } else {
    if (lane < nz) ((TI*)P.x)[(size_t)scene * nz + lane] = (TI)bx;
    if (vm) { ((TI*)P.z)[(size_t)scene * m + lane] = (TI)bz; ((TI*)P.s)[(size_t)scene * m + lane] = (TI)bs; }
    if (lane < e && P.y) ((TI*)P.y)[(size_t)scene * e + lane] = (TI)by;
    if (lane == 0) { if (P.iters) P.iters[scene] = iters; if (P.status) P.status[scene] = status; }
  }
}

template <typename TI, typename TC, bool PIVOT, bool FUSED, bool STRUCT>
__global__ void __launch_bounds__(64 * WPB, STRUCT ? LCP_W64_OCC : 1) lcp_fwd_wave(FwdArgs P, StepArgs SP, int lds_per_wave, int skip2) {
  fwd_wave_body<TI, TC, PIVOT, FUSED, STRUCT>(P, SP, lds_per_wave, skip2);
}
// Dense inputs: ONE launch for the scenes the four-scenes-per-wave kernels do not take - a wave reads its scene's class and runs the
// structured or the general body (round 4: the two were separate launches, and on the usual batches - every scene contact-structured
// with a diagonal Q - each of them cost 4.7 us to find nothing to do)
template <typename TI, typename TC>
__global__ void __launch_bounds__(64 * WPB) lcp_fwd_wave_any(FwdArgs P, StepArgs SP, int lds_per_wave, int skip2) {
  const int scene = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (scene >= P.B) return;
  Ws<TI, TC> W(P.ws, scene);
  if (W.meta[0] == (TC)0) fwd_wave_body<TI, TC, true, false, false>(P, SP, lds_per_wave, 0);
  else fwd_wave_body<TI, TC, true, false, true>(P, SP, lds_per_wave, skip2);
}

// ---------------------------------------------------------------- the backward kernel (lcp.py:37-64)
template <typename TI, typename TC, bool PIVOT, bool STRUCT>
__device__ __forceinline__ void bwd_wave_body(const BwdArgs& P, int lds_per_wave, int skip2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int scene = blockIdx.x * WPB + wave;
  if (scene >= P.B) return;
  if (P.tag && P.skip_tag && *P.tag == P.skip_tag) return;        // the forward fell back on the partner family: its backward serves this call
  unsigned char* smem = smem_all + (size_t)wave * lds_per_wave;
  const int nz = P.nz, m = P.m, e = P.e;
  Ops<TI, TC> O;
  carve(O.L, smem, !STRUCT);
  O.nz = nz; O.m = m; O.e = e; O.lane = lane;
  Ws<TI, TC> W(P.ws, scene);
  if ((W.meta[0] != (TC)0) != STRUCT) return;                     // served by the other instantiation
  if (STRUCT && skip2 && W.meta[0] == (TC)2) return;              // served by the quad kernel
  const bool vm = lane < m;
  zero_lds_inputs(O.L, lane);
  __syncthreads();
  const TI* G = (const TI*)P.G + (size_t)scene * m * nz;
  for (int idx = lane; idx < m * nz; idx += 64) {
    const int i = idx / nz, j = idx - i * nz;
    const TI g = G[idx];
    O.L.Gc[j * MP + i] = g; O.L.Gr[i * GRS + j] = g;
  }
  for (int i = lane; i < NZP * NZP; i += 64) O.L.Qit[i] = W.Qit[i];
  if (e > 0) {
    const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
    for (int idx = lane; idx < e * nz; idx += 64) { const int a = idx / nz, k = idx - a * nz; O.L.At[a * NZP + k] = A[idx]; }
    for (int i = lane; i < EP * MP; i += 64) O.L.GAc[i] = W.GAc[i];
    for (int i = lane; i < EP * EP; i += 64) O.L.S11i[i] = W.S11i[i];
  }
  __syncthreads();
  O.diagq = false;                                     // backward keeps the general Q^-1 product (one solve only)
  O.cache_rows();
  const int jx = lane & 15, ae = lane & 7;
  const TC x = (jx < nz) ? W.x[jx] : (TC)0;
  const TC z = vm ? W.z[lane] : (TC)1, s = vm ? W.s[lane] : (TC)1;
  const TC y = (ae < e) ? W.y[ae] : (TC)0;
  TC g = (jx < nz) ? (TC)((const TI*)P.dl_dx)[(size_t)scene * nz + jx] : (TC)0;
  if (P.tag && *P.tag != P.tag_value) g = nan_of<TC>();            // (another family's workspace: NaN gradients instead of a misread)
  TC d = vm ? z / s : (TC)1;                                                 // lcp.py:44
  TC t[MP];
  int mystep, porder;
  TC udinv;
  constexpr bool structured = STRUCT;
  Red<TC> RD;
  RD.init(m >> 2, lane, m);
  if (STRUCT && RD.isu) RD.mu = W.meta[1 + RD.cu];
  TC dinv = vm ? (TC)1 / d : (TC)1;
  const bool sing = factor<TI, TC, PIVOT, STRUCT>(t, O, W, RD, structured, dinv, mystep, porder, udinv);   // lcp.py:46
  {
    // (round 5) At an iterate that converged to rounding s / z ~ 1e-16 is lost against an R that redundant contact points make singular
    // and the elimination divides by rounding noise: multipliers of 1e15 .. 1e32, a dx that is their cancellation error (measured on
    // fp64 stacks with a non-diagonal Q: profiles/r05_own_iterate_probe.txt).  A pivot below 1e-13 of the largest diagonal entry of the
    // (reduced) R, or an exact zero, repeats the factorisation with s / z floored at 1e-12 x the row's own diagonal entry - as
    // factor_bwd_q (lcp_quad_kernels.inc) and lcp_bwd_kernel (lcp_generic.hip) do.
    auto rdiag = [&](int i) { const TC v = W.R2[(((size_t)(i >> 1)) * MP + i) * 2 + (i & 1)]; return v < 0 ? -v : v; };
    const int nrows = STRUCT ? RD.nr : m;
    TC scale = (lane < nrows) ? rdiag(lane) : (TC)0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const TC o = shfl_xor_t(scale, off); scale = o > scale ? o : scale; }
    const TC piv = (TC)1 / udinv, apiv = piv < 0 ? -piv : piv;
    const bool tiny = (lane < nrows) && !(apiv >= (TC)1e-13 * scale);
    static_assert(WPB == 1, "the repeated factorisation is a per-wavefront branch around lu_factor's __syncthreads(): with several wavefronts per workgroup the decision must be made workgroup-uniform (__syncthreads_or) first");
    if (__any(tiny || sing)) {
      const int nc = m >> 2;
      int idx = lane;                                                        // general scene: system row = m-space row
      bool fl = vm;
      if (STRUCT) { if (lane < nc) idx = lane; else if (lane < 3 * nc) idx = nc + ((lane - nc) >> 1); else fl = false; }   // (cone rows: no diagonal in R)
      if (fl) { const TC f = (TC)1e-12 * rdiag(idx); if (dinv < f) { dinv = f; d = (TC)1 / f; } }
      factor<TI, TC, PIVOT, STRUCT>(t, O, W, RD, structured, dinv, mystep, porder, udinv);
    }
  }
  TC dx, ds, dlam, dnu;
  solve_kkt<TI, TC, PIVOT, STRUCT>(O, t, mystep, porder, udinv, RD, structured, d, g, (TC)0, (TC)0, (TC)0,
                                  dx, ds, dlam, dnu);                                                    // lcp.py:47-50
  // outer products (lcp.py:52-61)
  if (P.dp && lane < nz) ((TI*)P.dp)[(size_t)scene * nz + lane] = (TI)dx;
  if (P.dh && vm) ((TI*)P.dh)[(size_t)scene * m + lane] = (TI)(-dlam);
  if (P.db && lane < e) ((TI*)P.db)[(size_t)scene * e + lane] = (TI)(-dnu);
  if (P.dQ) {
    TI* o = (TI*)P.dQ + (size_t)scene * nz * nz;
    for (int r = 0; r < nz; ++r) {                      // row r, lanes over columns
      const TC dxr = rdlane(dx, r), xr = rdlane(x, r);
      if (lane < nz) o[r * nz + lane] = (TI)((TC)0.5 * (dxr * x + xr * dx));
    }
  }
  if (P.dG) {
    TI* o = (TI*)P.dG + (size_t)scene * m * nz;
    for (int r = 0; r < m; ++r) {
      const TC dl = rdlane(dlam, r), zr = rdlane(z, r);
      if (lane < nz) o[r * nz + lane] = (TI)(dl * x + zr * dx);
    }
  }
  if (P.dA && e > 0) {
    TI* o = (TI*)P.dA + (size_t)scene * e * nz;
    for (int r = 0; r < e; ++r) {
      const TC dn = rdlane(dnu, r), yr = rdlane(y, r);
      if (lane < nz) o[r * nz + lane] = (TI)(dn * x + yr * dx);
    }
  }
  if (P.dF) {
    TI* o = (TI*)P.dF + (size_t)scene * m * m;
    for (int r = 0; r < m; ++r) {
      const TC dl = rdlane(dlam, r);
      if (vm) o[r * m + lane] = (TI)(-dl * z);
    }
  }
}

template <typename TI, typename TC, bool PIVOT, bool STRUCT>
__global__ void __launch_bounds__(64 * WPB) lcp_bwd_wave(BwdArgs P, int lds_per_wave, int skip2) {
  bwd_wave_body<TI, TC, PIVOT, STRUCT>(P, lds_per_wave, skip2);
}
template <typename TI, typename TC>
__global__ void __launch_bounds__(64 * WPB) lcp_bwd_wave_any(BwdArgs P, int lds_per_wave, int skip2) {   // (as lcp_fwd_wave_any)
  const int scene = blockIdx.x * WPB + (threadIdx.x >> 6);
  if (scene >= P.B) return;
  Ws<TI, TC> W(P.ws, scene);
  if (W.meta[0] == (TC)0) bwd_wave_body<TI, TC, true, false>(P, lds_per_wave, 0);
  else bwd_wave_body<TI, TC, true, true>(P, lds_per_wave, skip2);
}

}  // namespace w64

// ---------------------------------------------------------------- host-side launchers
bool wave64_supported(int nz, int m, int e) { return nz <= w64::NZP && m <= w64::MP && e <= w64::EP; }

size_t wave64_ws_bytes(int compute, int io_f64) {
  if (io_f64) return w64::ws_bytes<double, double>();
  return compute == LCP_COMPUTE_F64 ? w64::ws_bytes<float, double>() : w64::ws_bytes<float, float>();
}

template <typename TC, typename TI = float>
static size_t w64_lds(bool with_panel = true) { w64::Lds<TI, TC> L; return w64::carve<TI, TC>(L, nullptr, with_panel); }

static inline dim3 w64_grid(int B) { return dim3((B + w64::WPB - 1) / w64::WPB); }

// dynamic LDS above 64 KiB needs an explicit opt-in per kernel (gfx950 has 160 KiB per CU)
template <typename K>
static int w64_allow_lds(K kernel, size_t bytes) {
  // stateless on purpose: the opt-in is per device and per kernel, a cached "granted" size would be wrong on the second
  // GPU of a process and racy between host threads.  The attribute call is a host-side table update (no launch, no sync).
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return LCP_E_LAUNCH;
  return 0;
}
#define LCP_W64_LAUNCH(KERNEL, GRID, LW, ...)                                            \
  do {                                                                                   \
    auto kfn = KERNEL;                                                                   \
    if (w64_allow_lds(kfn, (size_t)(LW) * w64::WPB)) return LCP_E_LAUNCH;                \
    hipLaunchKernelGGL(kfn, GRID, blk, (size_t)(LW) * w64::WPB, st, __VA_ARGS__);        \
  } while (0)

int wave64_forward(const FwdArgs& P, int compute, void* stream, int io_f64, int body_space) {
  StepArgs SP = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 blk(64 * w64::WPB);
  const int quad = quad_supported(P.nz, P.m, P.e) ? 1 : 0;
  // classify, then: quad kernel (structured + diagonal Q), wave64 structured kernel, general kernel -
  // every scene is picked up by exactly one of them.
  if (io_f64) {                                  // fp64 I/O (the reference's native dtype): same kernels, fp64 loads / stores
    const int lw = (int)w64_lds<double, double>();
    hipLaunchKernelGGL((w64::lcp_classify_wave<double, double>), dim3((P.B + 3) / 4), dim3(256), 0, st, P);
    if (quad) { int rc = quad_forward(P, compute, 2, stream, 1); if (rc) return rc; }
    LCP_W64_LAUNCH((w64::lcp_fwd_wave_any<double, double>), w64_grid(P.B), lw, P, SP, lw, quad);      // (lw >= ls: the general body's LDS)
  } else if (compute == LCP_COMPUTE_F64) {
    const int lw = (int)w64_lds<double>();
    hipLaunchKernelGGL((w64::lcp_classify_wave<float, double>), dim3((P.B + 3) / 4), dim3(256), 0, st, P);
    if (quad) { int rc = quad_forward(P, compute, 2, stream, 0, body_space); if (rc) return rc; }
    LCP_W64_LAUNCH((w64::lcp_fwd_wave_any<float, double>), w64_grid(P.B), lw, P, SP, lw, quad);      // (lw >= ls: the general body's LDS)
  } else {
    const int lw = (int)w64_lds<float>();
    hipLaunchKernelGGL((w64::lcp_classify_wave<float, float>), dim3((P.B + 3) / 4), dim3(256), 0, st, P);
    if (quad) { int rc = quad_forward(P, compute, 2, stream); if (rc) return rc; }
    LCP_W64_LAUNCH((w64::lcp_fwd_wave_any<float, float>), w64_grid(P.B), lw, P, SP, lw, quad);      // (lw >= ls: the general body's LDS)
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int wave64_step(const StepArgs& SP, int compute, void* stream) {
  FwdArgs P = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 blk(64 * w64::WPB);
  if (quad_supported(3 * SP.nb, 4 * SP.nc, SP.e)) return quad_step(SP, compute, stream);
  if (compute == LCP_COMPUTE_F64) {
    const int lw = (int)w64_lds<double>(false);
    LCP_W64_LAUNCH((w64::lcp_fwd_wave<float, double, true, true, true>), w64_grid(SP.B), lw, P, SP, lw, 0);
  } else {
    const int lw = (int)w64_lds<float>(false);
    LCP_W64_LAUNCH((w64::lcp_fwd_wave<float, float, true, true, true>), w64_grid(SP.B), lw, P, SP, lw, 0);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int wave64_backward(const BwdArgs& P, int compute, bool all_quad, void* stream, int io_f64, int body_space) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 blk(64 * w64::WPB);
  const int quad = quad_supported(P.nz, P.m, P.e) ? 1 : 0;
  // (the class-2 scenes of a dense forward that ran in body space left no W: their backward factors in body space too)
  if (quad) { int rc = quad_backward(P, compute, 2, stream, io_f64, quad_dense_is_body_space(io_f64, compute, body_space) ? 1 : 0); if (rc || all_quad) return rc; }     // all_quad: LCP_HINT_ALL_CONTACT
  if (io_f64) {
    const int lw = (int)w64_lds<double, double>();
    LCP_W64_LAUNCH((w64::lcp_bwd_wave_any<double, double>), w64_grid(P.B), lw, P, lw, quad);
  } else if (compute == LCP_COMPUTE_F64) {
    const int lw = (int)w64_lds<double>();
    LCP_W64_LAUNCH((w64::lcp_bwd_wave_any<float, double>), w64_grid(P.B), lw, P, lw, quad);
  } else {
    const int lw = (int)w64_lds<float>();
    LCP_W64_LAUNCH((w64::lcp_bwd_wave_any<float, float>), w64_grid(P.B), lw, P, lw, quad);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
