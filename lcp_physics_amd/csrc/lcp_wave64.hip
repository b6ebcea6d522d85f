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

#include <utility>

#include "lcp_device.h"

namespace lcp {
namespace w64 {

constexpr int MP = 64;    // padded nineq (lanes)
constexpr int NZP = 16;   // padded nz
constexpr int EP = 8;     // padded neq
constexpr int GRS = 17;   // row stride of the row-major G copy (bank-conflict-free group reads)

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

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += shfl_xor_t(v, off);
  return v;
}
template <typename T> __device__ __forceinline__ void wave_sum2(T& a, T& b) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += shfl_xor_t(a, off); b += shfl_xor_t(b, off); }
}
// NaN-propagating max / min over the wave (Tensor.max()/min() semantics): plain IEEE max, then
// force NaN if any lane held one.
template <typename T> __device__ __forceinline__ void wave_pmax2(T& a, T& b) {
  const bool na = __any(a != a), nb = __any(b != b);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T oa = shfl_xor_t(a, off), ob = shfl_xor_t(b, off);
    a = a > oa ? a : oa; b = b > ob ? b : ob;
  }
  if (na) a = nan_of<T>();
  if (nb) b = nan_of<T>();
}
template <typename T> __device__ __forceinline__ void wave_pmin2(T& a, T& b) {
  const bool na = __any(a != a), nb = __any(b != b);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const T oa = shfl_xor_t(a, off), ob = shfl_xor_t(b, off);
    a = a < oa ? a : oa; b = b < ob ? b : ob;
  }
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
__device__ __forceinline__ void store2(float* dst, float a, float b) { *reinterpret_cast<float2*>(dst) = make_float2(a, b); }
__device__ __forceinline__ void store2(double* dst, double a, double b) { *reinterpret_cast<double2*>(dst) = make_double2(a, b); }

// ---------------------------------------------------------------- LU of T in registers
// t[j] of lane i = T[i][j].  Rows never move: step k eliminates with pivot lane p_k.
//   mystep : step at which this lane was the pivot (MP while unused)
//   porder : lane k holds p_k
//   udinv  : 1 / U[mystep][mystep] of this lane's row
// After the call: for lanes with mystep > k, t[k] holds the multiplier L[.][k]; for the lane with
// mystep == k, t[k..] is row k of U.  Returns true if an exact zero pivot was met.
template <typename TC, bool PIVOT>
__device__ __forceinline__ bool lu_factor(TC (&t)[MP], int m, int lane, int& mystep, int& porder, TC& udinv) {
  mystep = MP; porder = lane; udinv = (TC)1;
  bool singular = false;
  static_for<MP>([&](auto K) LCP_INL {
    constexpr int k = K;
    if (k < m) {
      int p = k;
      if (PIVOT) {
        const bool cand = (mystep == MP) && (lane < m);
        const float a = fabsf((float)t[k]);
        unsigned key = cand ? ((__float_as_uint(a) & ~63u) | (unsigned)lane) : 0u;
        key = wave_max_u32(key);
        p = (int)(__builtin_amdgcn_readfirstlane((int)key) & 63);
      }
      const TC piv = rdlane(t[k], p);
      singular = singular || (piv == (TC)0);
      const TC inv = (TC)1 / piv;
      const bool isp = (lane == p);
      const bool act = (mystep == MP) && !isp && (lane < m);
      const TC l = act ? t[k] * inv : (TC)0;
      t[k] = act ? l : t[k];
      mystep = isp ? k : mystep;
      udinv = isp ? inv : udinv;
      porder = (lane == k) ? p : porder;
      static_for<MP / 8>([&](auto C) LCP_INL {
        constexpr int c = C;
        if constexpr (c * 8 + 7 > k) {
          if (c * 8 < m) {
            static_for<8>([&](auto JJ) LCP_INL {
              constexpr int j = c * 8 + JJ;
              if constexpr (j > k) t[j] = fma(-l, rdlane(t[j], p), t[j]);
            });
          }
        }
      });
    }
  });
  return singular;
}

// Solve T w = rhs with the factorisation above.  rhs: lane i holds component i (natural order);
// the result has component j in lane j.
template <typename TC, bool PIVOT>
__device__ __forceinline__ TC lu_solve(const TC (&t)[MP], TC w, int m, int lane, int mystep, int porder, TC udinv) {
  static_for<MP>([&](auto K) LCP_INL {                 // L y = P rhs
    constexpr int k = K;
    if (k < m) {
      const int p = PIVOT ? rdlane_i(porder, k) : k;
      const TC yk = rdlane(w, p);
      const TC lk = (mystep > k && lane < m) ? t[k] : (TC)0;
      w = fma(-lk, yk, w);
    }
  });
  TC out = (TC)0;
  const TC wu = udinv;
  static_for<MP>([&](auto JR) LCP_INL {                // U x = y
    constexpr int j = MP - 1 - JR;
    if (j < m) {
      const int p = PIVOT ? rdlane_i(porder, j) : j;
      const TC xj = rdlane(w * wu, p);
      const TC uj = (mystep < j) ? t[j] : (TC)0;
      w = fma(-uj, xj, w);
      out = (lane == j) ? xj : out;
    }
  });
  return out;
}

// ---------------------------------------------------------------- per-scene LDS block
template <typename TI, typename TC>
struct Lds {
  TI* Gc;    // [NZP][MP]   Gc[j*MP + i]  = G[i][j]
  TI* Gr;    // [MP][GRS]   Gr[i*GRS + j] = G[i][j]
  TI* Qt;    // [NZP][NZP]  Qt[k*NZP + j] = Q[j][k]
  TI* At;    // [EP][NZP]   At[a*NZP + k] = A[a][k]
  TC* Qit;   // [NZP][NZP]  Qit[k*NZP + j] = Qinv[j][k]
  TC* Qrm;   // [nz][nz]    row-major Q^-1 (Gauss-Jordan work area, stride nz)
  TC* GAc;   // [EP][MP]    GAc[a*MP + i] = (G Q^-1 A^T)[i][a]
  TC* S11i;  // [e][e]      (A Q^-1 A^T)^-1, stride e
  TC* wbuf;  // [MP]
  int* flag;
};

template <typename TI, typename TC>
__host__ __device__ inline size_t carve(Lds<TI, TC>& L, unsigned char* smem) {
  unsigned char* q = smem;
  auto take = [&](size_t bytes) { unsigned char* r = q; q += (bytes + 15) & ~(size_t)15; return r; };
  L.Qit = (TC*)take(sizeof(TC) * NZP * NZP);
  L.Qrm = (TC*)take(sizeof(TC) * NZP * NZP);
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

// workspace per scene:  [TC] R2[MP*MP] Qit[256] GAc[512] S11i[64] x[16] s[64] z[64] y[8] pad -> 5120 TC
//                       [TI] Ft[MP*MP]   (lane-major copy of F: Ft[((j>>2)*MP + i)*4 + (j&3)] = F[i][j])
constexpr size_t WS_TC = 5120;
constexpr size_t WS_TI = MP * MP;
template <typename TI, typename TC> __host__ __device__ inline size_t ws_bytes() { return WS_TC * sizeof(TC) + WS_TI * sizeof(TI); }

template <typename TI, typename TC>
struct Ws {
  TC *R2, *Qit, *GAc, *S11i, *x, *s, *z, *y;
  TI* Ft;
  __device__ Ws(void* ws, int scene) {
    unsigned char* base = (unsigned char*)ws + (size_t)scene * ws_bytes<TI, TC>();
    TC* q = (TC*)base;
    R2 = q; q += MP * MP; Qit = q; q += NZP * NZP; GAc = q; q += EP * MP; S11i = q; q += EP * EP;
    x = q; q += NZP; s = q; q += MP; z = q; q += MP; y = q;
    Ft = (TI*)(base + WS_TC * sizeof(TC));
  }
};

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
#pragma unroll
    for (int jj = 0; jj < MP / 4; ++jj) {
      if (jj * 4 < m) {
        const TI* src = Ft + ((size_t)jj * MP + lane) * 4;
        TI f0, f1, f2, f3;
        load4(src, f0, f1, f2, f3);
        acc = fma((TC)f0, rdlane(z, jj * 4 + 0), acc);
        acc = fma((TC)f1, rdlane(z, jj * 4 + 1), acc);
        acc = fma((TC)f2, rdlane(z, jj * 4 + 2), acc);
        acc = fma((TC)f3, rdlane(z, jj * 4 + 3), acc);
      }
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
  // m-space <- x-space : (G v)_i
  __device__ __forceinline__ TC Gv(TC v) const {
    TC acc = 0;
#pragma unroll
    for (int j = 0; j < NZP; ++j) if (j < nz) acc = fma((TC)L.Gc[j * MP + lane], rdlane(v, j), acc);
    return acc;
  }
  // x-space <- m-space : (G^T w)_j, j = lane & 15 (replicated in the four lane groups)
  __device__ __forceinline__ TC Gtw(TC w) const {
    __syncthreads();
    L.wbuf[lane] = (lane < m) ? w : (TC)0;
    __syncthreads();
    const int j = lane & 15, q = lane >> 4;
    TC acc = 0;
#pragma unroll
    for (int ii = 0; ii < 16; ++ii) acc = fma((TC)L.Gr[(q * 16 + ii) * GRS + j], L.wbuf[q * 16 + ii], acc);
    acc += shfl_xor_t(acc, 16);
    acc += shfl_xor_t(acc, 32);
    return acc;
  }
  __device__ __forceinline__ TC Qiv(TC v) const {
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll
    for (int k = 0; k < NZP; ++k) if (k < nz) acc = fma(L.Qit[k * NZP + j], rdlane(v, k), acc);
    return acc;
  }
  __device__ __forceinline__ TC Qv(TC v) const {
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll
    for (int k = 0; k < NZP; ++k) if (k < nz) acc = fma((TC)L.Qt[k * NZP + j], rdlane(v, k), acc);
    return acc;
  }
  // e-space <- x-space : (A v)_a, a = lane & 7
  __device__ __forceinline__ TC Av(TC v) const {
    const int a = lane & 7;
    TC acc = 0;
#pragma unroll
    for (int k = 0; k < NZP; ++k) if (k < nz) acc = fma((TC)L.At[a * NZP + k], rdlane(v, k), acc);
    return acc;
  }
  // x-space <- e-space : (A^T y)_j
  __device__ __forceinline__ TC Aty(TC y) const {
    const int j = lane & 15;
    TC acc = 0;
#pragma unroll
    for (int a = 0; a < EP; ++a) if (a < e) acc = fma((TC)L.At[a * NZP + j], rdlane(y, a), acc);
    return acc;
  }
  // m-space <- e-space : (GA t)_i
  __device__ __forceinline__ TC GAt(TC t) const {
    TC acc = 0;
#pragma unroll
    for (int a = 0; a < EP; ++a) if (a < e) acc = fma(L.GAc[a * MP + lane], rdlane(t, a), acc);
    return acc;
  }
  // e-space <- m-space : (GA^T w)_a
  __device__ __forceinline__ TC GAtw(TC w) const {
    TC out = 0;
    const TC wm = (lane < m) ? w : (TC)0;
#pragma unroll
    for (int a = 0; a < EP; ++a) {
      if (a < e) {
        const TC sm = wave_sum(L.GAc[a * MP + lane] * wm);
        if ((lane & 7) == a) out = sm;
      }
    }
    return out;
  }
  __device__ __forceinline__ TC S11v(TC v) const {
    const int a = lane & 7;
    TC acc = 0;
#pragma unroll
    for (int c = 0; c < EP; ++c) if (c < e) acc = fma((a < e) ? L.S11i[a * e + c] : (TC)0, rdlane(v, c), acc);
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

// pre_factor_kkt (pdipm.py:357-408): Q^-1, G Q^-1 A^T, (A Q^-1 A^T)^-1 into LDS, R (and the lane-major F)
// into the workspace.
template <typename TI, typename TC, typename FT>
__device__ __forceinline__ int prefactor(const Lds<TI, TC>& L, const Ws<TI, TC>& W, const FT& F, int nz, int m, int e, int lane) {
  int status = 0;
  if (!gj_inverse<64>(L.Qrm, nz, L.flag)) status |= LCP_ST_SINGULAR_Q;
  for (int idx = lane; idx < nz * nz; idx += 64) { const int r = idx / nz, c = idx - r * nz; L.Qit[c * NZP + r] = L.Qrm[idx]; }
  __syncthreads();
  // gq = row `lane` of G Q^-1
  TC gq[NZP];
#pragma unroll
  for (int k = 0; k < NZP; ++k) {
    TC acc = 0;
    if (k < nz) for (int j = 0; j < nz; ++j) acc = fma((TC)L.Gr[lane * GRS + j], L.Qrm[j * nz + k], acc);
    gq[k] = acc;
  }
  TC cc[EP];
#pragma unroll
  for (int a = 0; a < EP; ++a) cc[a] = 0;
  if (e > 0) {
    // GA = (G Q^-1) A^T
#pragma unroll
    for (int a = 0; a < EP; ++a) {
      if (a < e) {
        TC acc = 0;
#pragma unroll
        for (int k = 0; k < NZP; ++k) acc = fma(gq[k], (TC)L.At[a * NZP + k], acc);
        L.GAc[a * MP + lane] = (lane < m) ? acc : (TC)0;
      }
    }
    // S11 = A Q^-1 A^T, one entry per lane (a = lane / e, c = lane % e)
    if (lane < e * e) {
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
    if (!gj_inverse<64>(L.S11i, e, L.flag)) status |= LCP_ST_SINGULAR_S11;
    // cc = row `lane` of GA S11^-1
#pragma unroll
    for (int a = 0; a < EP; ++a) {
      if (a < e) {
        TC acc = 0;
        for (int c = 0; c < e; ++c) acc = fma(L.GAc[c * MP + lane], L.S11i[c * e + a], acc);
        cc[a] = acc;
      }
    }
  }
  // R[i][j] = gq . G[j,:] + F[i][j] - cc . GA[j,:]   (two columns per trip, 16 B stores per lane)
  for (int j0 = 0; j0 < MP; j0 += 2) {
    TC r2[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = j0 + jj;
      TC val = 0;
      if (lane < m && j < m) {
#pragma unroll
        for (int k = 0; k < NZP; ++k) val = fma(gq[k], (TC)L.Gr[j * GRS + k], val);
        const TC f = F.at(lane, j);
        F.keep(lane, j, (TI)f);
        val += f;
#pragma unroll
        for (int a = 0; a < EP; ++a) if (a < e) val = fma(-cc[a], L.GAc[a * MP + j], val);
      } else if (lane < MP && j < MP) {
        F.keep(lane, j, (TI)0);
      }
      r2[jj] = val;
    }
    TC* dst = W.R2 + ((size_t)(j0 >> 1) * MP + lane) * 2;
    store2(dst, r2[0], r2[1]);
  }
  // things the backward kernel needs
  for (int i = lane; i < NZP * NZP; i += 64) W.Qit[i] = L.Qit[i];
  for (int i = lane; i < EP * MP; i += 64) W.GAc[i] = L.GAc[i];
  for (int i = lane; i < EP * EP; i += 64) W.S11i[i] = L.S11i[i];
  __threadfence_block();
  __syncthreads();
  return status;
}

// T = R + diag(1/d) from the workspace into registers (pdipm.py:427-429)
template <typename TI, typename TC>
__device__ __forceinline__ void load_T(TC (&t)[MP], const Ws<TI, TC>& W, TC dinv, int m, int lane) {
  static_for<MP / 2>([&](auto JJ) LCP_INL {
    constexpr int jj = JJ;
    if (jj * 2 < m) {
      const TC* src = W.R2 + ((size_t)jj * MP + lane) * 2;
      TC a, b;
      load2(src, a, b);
      t[2 * jj] = a + ((lane == 2 * jj) ? dinv : (TC)0);
      t[2 * jj + 1] = b + ((lane == 2 * jj + 1) ? dinv : (TC)0);
    } else {
      t[2 * jj] = (lane == 2 * jj) ? dinv : (TC)0;
      t[2 * jj + 1] = (lane == 2 * jj + 1) ? dinv : (TC)0;
    }
  });
}

// solve_kkt (pdipm.py:325-354) at wave level.  rs_over_d = rs / d (m-space).
template <typename TI, typename TC, bool PIVOT>
__device__ __forceinline__ void solve_kkt(const Ops<TI, TC>& O, const TC (&t)[MP], int mystep, int porder, TC udinv,
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
  const TC wz = lu_solve<TC, PIVOT>(t, hz, m, lane, mystep, porder, udinv);   // T^-1 (...)
  TC dy = 0;
  if (e > 0) dy = -O.S11v(hy - O.GAtw(wz));                                // dy = -wy
  const TC dz = (lane < m) ? -wz : (TC)0;                                  // :342
  os = (lane < m) ? (-rs - dz) / d : (TC)0;                                // :347,350
  oz = dz; oy = dy;
  TC g1 = -rx - O.Gtw(dz);                                                 // :344-346
  if (e > 0) g1 -= O.Aty(dy);
  ox = O.Qiv(g1);                                                          // :349
}

// ---------------------------------------------------------------- the forward kernel
template <typename TI, typename TC, bool PIVOT, bool FUSED>
__global__ void __launch_bounds__(64) lcp_fwd_wave(FwdArgs P, StepArgs SP) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x, lane = threadIdx.x;
  const int nz = FUSED ? 3 * SP.nb : P.nz, m = FUSED ? 4 * SP.nc : P.m, e = FUSED ? SP.e : P.e;
  const int max_iter = FUSED ? SP.max_iter : P.max_iter, lim = FUSED ? SP.lim : P.lim;
  const TC eps = (TC)(FUSED ? SP.eps : P.eps);
  Ops<TI, TC> O;
  carve(O.L, smem);
  O.nz = nz; O.m = m; O.e = e; O.lane = lane;
  Ws<TI, TC> W(FUSED ? SP.ws : P.ws, scene);
  const bool vm = lane < m;
  TC p, h, b, mu_lane = 0;
  int status;
  TC fz_dummy = 0; (void)fz_dummy;
  FDenseW<TI, TC> Fd{FUSED ? nullptr : (const TI*)P.F + (size_t)scene * m * m, W.Ft, m};
  if (FUSED) {
    assemble_scene<TI, TC>(O.L, SP, scene, lane, p, h, b, mu_lane);
    FContactW<TC> Fc{SP.nc, mu_lane};
    status = prefactor<TI, TC>(O.L, W, Fc, nz, m, e, lane);
  } else {
    load_dense<TI, TC>(O.L, P, scene, lane, p, h, b);
    status = prefactor<TI, TC>(O.L, W, Fd, nz, m, e, lane);
  }
  FContactW<TC> Fc{FUSED ? SP.nc : 0, mu_lane};

  TC t[MP];
  int mystep, porder;
  TC udinv;
  TC x = 0, s = 1, z = 1, y = 0, d = 1;
  TC bx = 0, bs = 1, bz = 1, by = 0;
  TC best_resid = inf_of<TC>();
  bool have_best = false;
  int n_not = 0, iters = 0;
  double* trace = (!FUSED && P.trace) ? P.trace + (size_t)scene * 4 * max_iter : nullptr;

#pragma unroll 1
  for (int it = -1; it < max_iter; ++it) {
    TC rx, rs, rz, ry, mu = 0, resid = 0;
    if (it < 0) {                                                           // init: (p, 0, -h, -b), d = 1 (:57-63)
      rx = p; rs = 0; rz = -h; ry = -b; d = 1;
    } else {                                                                // residuals (:82-96)
      rx = O.Gtw(z) + O.Qv(x) + p;
      if (e > 0) rx += O.Aty(y);
      rs = z;
      const TC fz = FUSED ? Fc.Fz(z, lane) : Fd.Fz(z, lane);
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
    load_T<TI, TC>(t, W, vm ? (TC)1 / d : (TC)1, m, lane);
    const bool singular = lu_factor<TC, PIVOT>(t, m, lane, mystep, porder, udinv);   // (:99-100)
    if (it >= 0) {
      ++iters;
      if (trace && lane == 0) { trace[4 * it] = (double)resid; trace[4 * it + 1] = (double)mu; }
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; break; }      // except: return best (:99-102)
      const bool improved = !have_best || (resid < best_resid);            // (:107-132)
      if (improved) { best_resid = resid; n_not = 0; have_best = true; bx = x; bs = s; bz = z; by = y; }
      else ++n_not;
      if (n_not == lim || best_resid < eps || mu > mu_limit<TC>()) break;  // (:133)
    }
    TC ax, as_, az, ay;
    solve_kkt<TI, TC, PIVOT>(O, t, mystep, porder, udinv, d, rx, rs, rz, ry, ax, as_, az, ay);
    if (it < 0) {
      x = ax; s = as_; z = az; y = ay;                                      // (:60-63)
      TC smin = vm ? s : inf_of<TC>(), zmin = vm ? z : inf_of<TC>();
      wave_pmin2(smin, zmin);
      if (smin <= (TC)0) s = s - smin + (TC)1;                              // (:66-75)
      if (zmin <= (TC)0) z = z - zmin + (TC)1;
      if (!vm) { s = 1; z = 1; }
      continue;
    }
    TC alpha = pmin(step_pair(z, az, s, as_, vm), (TC)1);                  // (:142-144)
    TC t3 = vm ? (s + alpha * as_) * (z + alpha * az) : (TC)0, t4 = vm ? s * z : (TC)0;
    wave_sum2(t3, t4);
    const TC r3 = t3 / t4, sig = r3 * r3 * r3;                              // (:146-150)
    const TC rsc = vm ? (-mu * sig + as_ * az) / s : (TC)0;                 // (:153)
    TC cx, cs, cz, cy;
    solve_kkt<TI, TC, PIVOT>(O, t, mystep, porder, udinv, d, (TC)0, rsc, (TC)0, (TC)0, cx, cs, cz, cy);
    cx += ax; cs += as_; cz += az; cy += ay;                                // (:160-163)
    alpha = pmin((TC)0.999 * step_pair(z, cz, s, cs, vm), (TC)1);          // (:164-166)
    if (trace && lane == 0) { trace[4 * it + 2] = (double)sig; trace[4 * it + 3] = (double)alpha; }
    x += alpha * cx; y += alpha * cy;                                       // (:171-174)
    if (vm) { s += alpha * cs; z += alpha * cz; }
  }

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

// ---------------------------------------------------------------- the backward kernel (lcp.py:37-64)
template <typename TI, typename TC, bool PIVOT>
__global__ void __launch_bounds__(64) lcp_bwd_wave(BwdArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x, lane = threadIdx.x;
  const int nz = P.nz, m = P.m, e = P.e;
  Ops<TI, TC> O;
  carve(O.L, smem);
  O.nz = nz; O.m = m; O.e = e; O.lane = lane;
  Ws<TI, TC> W(P.ws, scene);
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
  const int jx = lane & 15, ae = lane & 7;
  const TC x = (jx < nz) ? W.x[jx] : (TC)0;
  const TC z = vm ? W.z[lane] : (TC)1, s = vm ? W.s[lane] : (TC)1;
  const TC y = (ae < e) ? W.y[ae] : (TC)0;
  const TC g = (jx < nz) ? (TC)((const TI*)P.dl_dx)[(size_t)scene * nz + jx] : (TC)0;
  const TC d = vm ? z / s : (TC)1;                                           // lcp.py:44
  TC t[MP];
  int mystep, porder;
  TC udinv;
  load_T<TI, TC>(t, W, vm ? (TC)1 / d : (TC)1, m, lane);
  lu_factor<TC, PIVOT>(t, m, lane, mystep, porder, udinv);                  // lcp.py:46
  TC dx, ds, dlam, dnu;
  solve_kkt<TI, TC, PIVOT>(O, t, mystep, porder, udinv, d, g, (TC)0, (TC)0, (TC)0, dx, ds, dlam, dnu);   // lcp.py:47-50
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

}  // namespace w64

// ---------------------------------------------------------------- host-side launchers
bool wave64_supported(int nz, int m, int e) { return nz <= w64::NZP && m <= w64::MP && e <= w64::EP; }

size_t wave64_ws_bytes(int compute) {
  return compute == LCP_COMPUTE_F64 ? w64::ws_bytes<float, double>() : w64::ws_bytes<float, float>();
}

template <typename TC>
static size_t w64_lds() { w64::Lds<float, TC> L; return w64::carve<float, TC>(L, nullptr); }

int wave64_forward(const FwdArgs& P, int compute, void* stream) {
  StepArgs SP = {};
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64)
    hipLaunchKernelGGL((w64::lcp_fwd_wave<float, double, true, false>), dim3(P.B), dim3(64), w64_lds<double>(), st, P, SP);
  else
    hipLaunchKernelGGL((w64::lcp_fwd_wave<float, float, true, false>), dim3(P.B), dim3(64), w64_lds<float>(), st, P, SP);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int wave64_step(const StepArgs& SP, int compute, void* stream) {
  FwdArgs P = {};
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64)
    hipLaunchKernelGGL((w64::lcp_fwd_wave<float, double, true, true>), dim3(SP.B), dim3(64), w64_lds<double>(), st, P, SP);
  else
    hipLaunchKernelGGL((w64::lcp_fwd_wave<float, float, true, true>), dim3(SP.B), dim3(64), w64_lds<float>(), st, P, SP);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int wave64_backward(const BwdArgs& P, int compute, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64)
    hipLaunchKernelGGL((w64::lcp_bwd_wave<float, double, true>), dim3(P.B), dim3(64), w64_lds<double>(), st, P);
  else
    hipLaunchKernelGGL((w64::lcp_bwd_wave<float, float, true>), dim3(P.B), dim3(64), w64_lds<float>(), st, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
