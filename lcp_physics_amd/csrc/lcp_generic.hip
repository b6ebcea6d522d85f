// lcp_generic.hip - any-size batched PDIPM LCP kernels for gfx950 (one workgroup per scene).
//
// This is the GENERIC path: runtime sizes (nz, m, e), T = R + diag(s/z) kept in LDS when it
// fits (else in the HBM workspace), 256-thread workgroups.  The wave-per-scene register-resident
// kernels for m <= 64 live in lcp_wave64.hip; both implement the same algorithm:
//
//   reference (paths under /root/reference/lcp_physics)          here
//   lcp/solvers/pdipm.py:357-408  pre_factor_kkt                  prefactor()
//   lcp/solvers/pdipm.py:414-454  factor_kkt                      factor_T()
//   lcp/solvers/pdipm.py:325-354  solve_kkt                       solve_kkt()
//   lcp/solvers/pdipm.py:182-186  get_step                        step_pair()
//   lcp/solvers/pdipm.py:49-179   forward                         lcp_fwd_kernel
//   lcp/lcp.py:37-64              LCPFunction.backward            lcp_bwd_kernel
//   physics/engines.py:31-32,50-74, physics/world.py:144-234      assemble_scene()
//   physics/bodies.py:80-82       Body.move                       epilogue of the fused kernel
//
// Numerics: TC is the arithmetic type.  With TC = double the iterates follow the reference's
// fp64 trajectory (T is numerically singular in fp32 for sticking friction pairs).
// LU(T) uses partial pivoting like the reference's CPU path (pdipm.py:18, `pivot=not x.is_cuda`):
// at a converged iterate T is numerically singular along each sticking friction pair and a
// pivot-free elimination turns rounding noise into 1e16 multipliers (measured: the backward
// solve loses all digits on ~1 % of stack scenes without it).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcp_device.h"

namespace lcp {

constexpr int NT = 256;          // threads per workgroup (4 waves)
constexpr int NW = NT / 64;

// Workgroup reduction of two values at once (all threads receive the results).
template <typename T, typename Op>
__device__ void block_reduce2(T& a, T& b, Op op, T* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a = op(a, shfl_xor_t(a, off));
    b = op(b, shfl_xor_t(b, off));
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
  __syncthreads();
  a = red[0]; b = red[1];
#pragma unroll
  for (int i = 1; i < NW; ++i) { a = op(a, red[2 * i]); b = op(b, red[2 * i + 1]); }
}

// All per-scene state.  Pointers into LDS unless noted.
template <typename TC>
struct Scene {
  int nz, m, e, ldT;
  int nc;                 // contacts (fused path) : F is structured, m = 4 nc
  // matrices
  TC *Q, *Qi, *G, *A, *GA, *S11i, *scr;   // scr: m*nz scratch (G Q^-1) during prefactor
  TC *T;                                  // LU(T): LDS or workspace
  TC *R;                                  // HBM workspace
  TC *mu_c;                               // fused: friction coefficient per contact [nc]
  // vectors
  TC *p, *h, *b, *x, *s, *z, *y, *d;
  TC *rx, *rs, *rz, *ry;
  TC *ax, *as, *az, *ay;                  // affine direction
  TC *cx, *cs, *cz, *cy;                  // solve output / combined direction
  TC *bx, *bs, *bz, *by;                  // best iterate
  TC *v, *hy, *hz, *g1, *t, *u, *tmp;     // temporaries
  TC *red;                                // reduction scratch (2*NW)
  int *perm;                              // row permutation of LU(T) (pivoting)
  int *flag;                              // LDS status word
};

// Carve the per-scene LDS block.  Called with smem == nullptr on the host to size it.
// `level` (Plan::t_in_lds) - where the matrices live:
//   1  everything in LDS
//   0  the big-problem plan: T and the prefactor scratch (G Q^-1) in the HBM workspace (T_ws points at T, the scratch follows it)
//   2  the huge-problem plan: Q, Q^-1 and G follow them there too; LDS keeps A, G A-terms and the vectors (any World the reference
//      can hold: 32 bodies x 128 contacts in fp64 is 80 KB of vectors) - every access goes through the same pointers
template <typename TC>
__host__ __device__ inline size_t carve(Scene<TC>& S, unsigned char* smem, int nz, int m, int e, int ldT,
                                        int level, TC* T_ws) {
  const bool t_in_lds = level == 1, mats_in_lds = level != 2;
  TC* q = reinterpret_cast<TC*>(smem);
  auto take = [&](size_t n) { TC* r = q; q += n; return r; };
  S.nz = nz; S.m = m; S.e = e; S.ldT = ldT; S.nc = 0;
  TC* far = T_ws + (size_t)m * ldT + (size_t)m * nz;                     // (level 2) behind T and the scratch
  S.Q = mats_in_lds ? take((size_t)nz * nz) : far; S.Qi = mats_in_lds ? take((size_t)nz * nz) : far + (size_t)nz * nz;
  S.G = mats_in_lds ? take((size_t)m * nz) : far + 2 * (size_t)nz * nz;
  S.scr = t_in_lds ? take((size_t)m * nz) : T_ws + (size_t)m * ldT;
  S.A = take((size_t)e * nz); S.GA = take((size_t)m * e); S.S11i = take((size_t)e * e);
  S.T = t_in_lds ? take((size_t)m * ldT) : T_ws;
  S.mu_c = take(m);
  S.p = take(nz); S.x = take(nz); S.rx = take(nz); S.ax = take(nz); S.cx = take(nz); S.bx = take(nz);
  S.v = take(nz); S.g1 = take(nz);
  S.h = take(m); S.s = take(m); S.z = take(m); S.d = take(m); S.rs = take(m); S.rz = take(m);
  S.as = take(m); S.az = take(m); S.cs = take(m); S.cz = take(m); S.bs = take(m); S.bz = take(m);
  S.hz = take(m); S.tmp = take(m);
  S.b = take(e); S.y = take(e); S.ry = take(e); S.ay = take(e); S.cy = take(e); S.by = take(e);
  S.hy = take(e); S.t = take(e); S.u = take(e);
  S.red = take(2 * NW);
  S.perm = reinterpret_cast<int*>(q);
  S.flag = S.perm + m;
  size_t bytes = (size_t)(reinterpret_cast<unsigned char*>(S.flag + 4) - smem);
  return (bytes + 15) & ~(size_t)15;
}

inline size_t lds_bytes_for(int nz, int m, int e, int ldT, int t_in_lds, int csize) {
  if (csize == 8) { Scene<double> S; return carve<double>(S, nullptr, nz, m, e, ldT, t_in_lds, nullptr); }
  Scene<float> S; return carve<float>(S, nullptr, nz, m, e, ldT, t_in_lds, nullptr);
}

// ------------------------------------------------------------------------------------------
// F: dense (HBM, I/O precision) or structured (fused path: engines.py:69-73)
// ------------------------------------------------------------------------------------------
template <typename TI, typename TC>
struct FDense {
  const TI* F;   // [m,m] of this scene
  int m;
  __device__ TC at(int i, int j) const { return (TC)F[(size_t)i * m + j]; }
  // out[i] -= (F z)[i]  for all rows (workgroup-cooperative: one wave per row, coalesced)
  __device__ void sub_Fz(const TC* z, TC* out) const {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    for (int i = w; i < m; i += NW) {
      TC acc = 0;
      for (int j = l; j < m; j += 64) acc += (TC)F[(size_t)i * m + j] * z[j];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += shfl_xor_t(acc, off);
      if (l == 0) out[i] -= acc;
    }
  }
};

template <typename TC>
struct FContact {
  const TC* mu;  // [nc] in LDS
  int nc;
  __device__ TC at(int i, int j) const {
    if (i < nc) return (TC)0;
    if (i < 3 * nc) return (j >= 3 * nc && ((i - nc) >> 1) == (j - 3 * nc)) ? (TC)1 : (TC)0;
    const int c = i - 3 * nc;
    if (j < nc) return (j == c) ? mu[c] : (TC)0;
    if (j < 3 * nc) return (((j - nc) >> 1) == c) ? (TC)-1 : (TC)0;
    return (TC)0;
  }
  __device__ void sub_Fz(const TC* z, TC* out) const {
    for (int i = threadIdx.x; i < 4 * nc; i += NT) {
      TC fz = 0;
      if (i >= nc && i < 3 * nc) fz = z[3 * nc + ((i - nc) >> 1)];
      else if (i >= 3 * nc) { const int c = i - 3 * nc; fz = mu[c] * z[c] - (z[nc + 2 * c] + z[nc + 2 * c + 1]); }
      out[i] -= fz;
    }
  }
};

// F = 0: the frictionless LCP of post-stabilisation (engines.py:110)
template <typename TC>
struct FZero {
  __device__ TC at(int, int) const { return (TC)0; }
  __device__ void sub_Fz(const TC*, TC*) const {}
};

// ------------------------------------------------------------------------------------------
// small dense building blocks (workgroup-cooperative, operands in LDS)
// ------------------------------------------------------------------------------------------

// out[i] = sum_j M[i*ld + j] * v[j]   (rows over threads)
template <typename TC>
__device__ __forceinline__ TC row_dot(const TC* Mrow, const TC* v, int n) {
  TC acc = 0;
  for (int j = 0; j < n; ++j) acc += Mrow[j] * v[j];
  return acc;
}
// sum_i M[i*ld + col] * v[i]
template <typename TC>
__device__ __forceinline__ TC col_dot(const TC* M, int ld, int col, const TC* v, int n) {
  TC acc = 0;
  for (int i = 0; i < n; ++i) acc += M[(size_t)i * ld + col] * v[i];
  return acc;
}

// pre_factor_kkt (pdipm.py:357-408): Qi = Q^-1, GA = G Qi A^T, S11i = (A Qi A^T)^-1,
// R = G Qi G^T + F - GA S11i GA^T  (written to the HBM workspace).
template <typename TC, typename FT>
__device__ int prefactor(Scene<TC>& S, const FT& F) {
  const int nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  int status = 0;
  for (int i = tid; i < nz * nz; i += NT) S.Qi[i] = S.Q[i];
  __syncthreads();
  if (!gj_inverse<NT>(S.Qi, nz, S.flag)) status |= LCP_ST_SINGULAR_Q;
  // scr = G Qi   [m, nz]
  for (int idx = tid; idx < m * nz; idx += NT) {
    const int i = idx / nz, j = idx - i * nz;
    S.scr[idx] = col_dot(S.Qi, nz, j, S.G + (size_t)i * nz, nz);
  }
  __syncthreads();
  if (e > 0) {
    // GA = scr A^T [m, e]
    for (int idx = tid; idx < m * e; idx += NT) {
      const int i = idx / e, a = idx - i * e;
      S.GA[idx] = row_dot(S.scr + (size_t)i * nz, S.A + (size_t)a * nz, nz);
    }
    // S11 = A Qi A^T  [e, e]  (t-free: computed straight from Qi)
    for (int idx = tid; idx < e * e; idx += NT) {
      const int a = idx / e, c = idx - a * e;
      TC acc = 0;
      for (int k = 0; k < nz; ++k) acc += S.A[a * nz + k] * col_dot(S.Qi, nz, k, S.A + (size_t)c * nz, nz);
      S.S11i[idx] = acc;
    }
    __syncthreads();
    if (!gj_inverse<NT>(S.S11i, e, S.flag)) status |= LCP_ST_SINGULAR_S11;
  }
  // R = scr G^T + F - (GA S11i) GA^T
  for (int idx = tid; idx < m * m; idx += NT) {
    const int i = idx / m, j = idx - i * m;
    TC acc = row_dot(S.scr + (size_t)i * nz, S.G + (size_t)j * nz, nz) + F.at(i, j);
    if (e > 0) {
      TC corr = 0;
      for (int a = 0; a < e; ++a) {
        TC ca = 0;
        for (int c = 0; c < e; ++c) ca += S.GA[i * e + c] * S.S11i[c * e + a];
        corr += ca * S.GA[j * e + a];
      }
      acc -= corr;
    }
    S.R[idx] = acc;
  }
  __syncthreads();
  return status;
}

// factor_kkt (pdipm.py:414-454): T = R + diag(1/d) = R + diag(s/z); LU in place.
// Returns non-zero when an exact zero pivot was met (bit 0: the reference's `except` path, :99-102) or - backward only, `tiny` > 0 - a
// pivot below `tiny` (bit 1: what is left of it is rounding noise; see lcp_bwd_kernel).
// `transposed`: T^T = R^T + diag(s/z) - the opt-in adjoint backward (LCP_BWD_ADJOINT): K^T is K with F^T for F, and R = (symmetric) + F
template <typename TC, bool PIVOT>
__device__ int factor_T(Scene<TC>& S, const TC* dinv, TC tiny = (TC)0, bool transposed = false) {
  const int m = S.m, ld = S.ldT, tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  for (int idx = tid; idx < m * m; idx += NT) {
    const int i = idx / m, j = idx - i * m;
    TC val = transposed ? S.R[(size_t)j * m + i] : S.R[idx];
    if (i == j) val += dinv[i];
    S.T[(size_t)i * ld + j] = val;
  }
  for (int i = tid; i < m; i += NT) S.perm[i] = i;
  if (tid == 0) *S.flag = 0;
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    if (PIVOT) {
      // partial pivoting: wave 0 finds argmax_i>=k |T[i][k]| (first max wins), then rows swap
      if (w == 0) {
        TC best = -1; int bi = k;
        for (int i = k + l; i < m; i += 64) {
          TC a = S.T[(size_t)i * ld + k]; a = a < 0 ? -a : a;
          if (a > best) { best = a; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          TC ob = shfl_xor_t(best, off); int oi = __shfl_xor(bi, off, 64);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (l == 0) S.flag[1] = bi;
      }
      __syncthreads();
      const int pr = S.flag[1];
      if (pr != k) {
        for (int j = tid; j < m; j += NT) {
          TC a = S.T[(size_t)k * ld + j], b = S.T[(size_t)pr * ld + j];
          S.T[(size_t)k * ld + j] = b; S.T[(size_t)pr * ld + j] = a;
        }
        if (tid == 0) { int a = S.perm[k]; S.perm[k] = S.perm[pr]; S.perm[pr] = a; }
      }
      __syncthreads();
    }
    const TC piv = S.T[(size_t)k * ld + k];
    if (piv == (TC)0 && tid == 0) *S.flag |= 1;
    if (tiny > (TC)0 && !((piv < 0 ? -piv : piv) >= tiny) && tid == 0) *S.flag |= 2;
    const TC pinv = (TC)1 / piv;
    for (int i = k + 1 + w; i < m; i += NW) {
      const TC lik = S.T[(size_t)i * ld + k] * pinv;
      for (int j = k + 1 + l; j < m; j += 64) S.T[(size_t)i * ld + j] -= lik * S.T[(size_t)k * ld + j];
      if (l == 0) S.T[(size_t)i * ld + k] = lik;
    }
    __syncthreads();
  }
  return *S.flag;
}

// In-place solve T w = rhs with the LU above; rhs / result in S.hz.
template <typename TC, bool PIVOT>
__device__ void solve_T(Scene<TC>& S) {
  const int m = S.m, ld = S.ldT, tid = threadIdx.x;
  if (m <= 64) {
    // wave 0, one row per lane, right-hand side in a register, broadcasts by shuffle
    if (tid < 64) {
      const int i = tid;
      TC val = (i < m) ? S.hz[PIVOT ? S.perm[i] : i] : (TC)0;
      for (int k = 0; k + 1 < m; ++k) {
        const TC vk = shfl_t(val, k);
        if (i > k && i < m) val -= S.T[(size_t)i * ld + k] * vk;
      }
      for (int k = m - 1; k >= 0; --k) {
        if (i == k) val = val / S.T[(size_t)k * ld + k];
        const TC vk = shfl_t(val, k);
        if (i < k) val -= S.T[(size_t)i * ld + k] * vk;
      }
      if (i < m) S.hz[i] = val;
    }
    __syncthreads();
    return;
  }
  if (PIVOT) {
    for (int i = tid; i < m; i += NT) S.tmp[i] = S.hz[S.perm[i]];
    __syncthreads();
    for (int i = tid; i < m; i += NT) S.hz[i] = S.tmp[i];
    __syncthreads();
  }
  for (int k = 0; k + 1 < m; ++k) {
    const TC vk = S.hz[k];
    for (int i = k + 1 + tid; i < m; i += NT) S.hz[i] -= S.T[(size_t)i * ld + k] * vk;
    __syncthreads();
  }
  for (int k = m - 1; k >= 0; --k) {
    if (tid == 0) S.hz[k] = S.hz[k] / S.T[(size_t)k * ld + k];
    __syncthreads();
    const TC vk = S.hz[k];
    for (int i = tid; i < k; i += NT) S.hz[i] -= S.T[(size_t)i * ld + k] * vk;
    __syncthreads();
  }
}

// solve_kkt (pdipm.py:325-354).  rhs vectors: rx[nz], rs[m], rz[m], ry[e] (any may be nullptr =
// zero).  Outputs to ox[nz], os[m], oz[m], oy[e].  d in S.d.
template <typename TC, bool PIVOT>
__device__ void solve_kkt(Scene<TC>& S, const TC* rx, const TC* rs, const TC* rz, const TC* ry,
                          TC* ox, TC* os, TC* oz, TC* oy) {
  const int nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  // v = Q^-1 rx                                                            (:333)
  for (int j = tid; j < nz; j += NT) S.v[j] = rx ? row_dot(S.Qi + (size_t)j * nz, rx, nz) : (TC)0;
  __syncthreads();
  // hz = G v + rs/d - rz ; hy = A v - ry                                   (:334-340)
  for (int i = tid; i < m; i += NT) {
    TC a = rx ? row_dot(S.G + (size_t)i * nz, S.v, nz) : (TC)0;
    if (rs) a += rs[i] / S.d[i];
    if (rz) a -= rz[i];
    S.hz[i] = a;
  }
  for (int a = tid; a < e; a += NT) {
    TC acc = rx ? row_dot(S.A + (size_t)a * nz, S.v, nz) : (TC)0;
    if (ry) acc -= ry[a];
    S.hy[a] = acc;
  }
  __syncthreads();
  // w = S^-1 [hy; hz] by block elimination of the equality block           (:342)
  if (e > 0) {
    for (int a = tid; a < e; a += NT) S.t[a] = row_dot(S.S11i + (size_t)a * e, S.hy, e);
    __syncthreads();
    for (int i = tid; i < m; i += NT) S.hz[i] -= row_dot(S.GA + (size_t)i * e, S.t, e);
    __syncthreads();
  }
  solve_T<TC, PIVOT>(S);                  // hz <- T^-1 hz
  if (e > 0) {
    for (int a = tid; a < e; a += NT) S.u[a] = S.hy[a] - col_dot(S.GA, e, a, S.hz, m);
    __syncthreads();
    for (int a = tid; a < e; a += NT) oy[a] = -row_dot(S.S11i + (size_t)a * e, S.u, e);   // dy = -wy
  }
  for (int i = tid; i < m; i += NT) {
    const TC dz = -S.hz[i];                                               // dz = w_z = -S^-1 h
    oz[i] = dz;
    os[i] = (-(rs ? rs[i] : (TC)0) - dz) / S.d[i];                        // (:347,350)
  }
  __syncthreads();
  // g1 = -rx - G^T dz - A^T dy ; dx = Q^-1 g1                              (:344-349)
  for (int j = tid; j < nz; j += NT) {
    TC a = -(rx ? rx[j] : (TC)0) - col_dot(S.G, nz, j, oz, m);
    if (e > 0) a -= col_dot(S.A, nz, j, oy, e);
    S.g1[j] = a;
  }
  __syncthreads();
  for (int j = tid; j < nz; j += NT) ox[j] = row_dot(S.Qi + (size_t)j * nz, S.g1, nz);
  __syncthreads();
}

// get_step for the pair (z, dz), (s, ds) at once (pdipm.py:182-186, per-scene max):
// returns min(step(z,dz), step(s,ds)) with NaN propagation.
template <typename TC>
__device__ TC step_pair(Scene<TC>& S, const TC* z, const TC* dz, const TC* s, const TC* ds) {
  const int m = S.m, tid = threadIdx.x;
  TC mz = -inf_of<TC>(), ms = -inf_of<TC>();
  for (int i = tid; i < m; i += NT) {
    mz = pmax(mz, -z[i] / dz[i]);
    ms = pmax(ms, -s[i] / ds[i]);
  }
  block_reduce2(mz, ms, OpMax(), S.red);
  const TC fz = (mz > (TC)1) ? mz : (TC)1;       // max(1.0, a.max()): NaN -> 1.0
  const TC fs = (ms > (TC)1) ? ms : (TC)1;
  TC az = inf_of<TC>(), as = inf_of<TC>();
  for (int i = tid; i < m; i += NT) {
    az = pmin(az, (dz[i] > (TC)0) ? fz : (-z[i] / dz[i]));
    as = pmin(as, (ds[i] > (TC)0) ? fs : (-s[i] / ds[i]));
  }
  block_reduce2(az, as, OpMin(), S.red);
  return pmin(az, as);
}

// ------------------------------------------------------------------------------------------
// input stage
// ------------------------------------------------------------------------------------------
template <typename TI, typename TC>
__device__ void load_dense(Scene<TC>& S, const FwdArgs& P, int scene) {
  const int nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  const TI* Q = (const TI*)P.Q + (size_t)scene * nz * nz;
  const TI* G = (const TI*)P.G + (size_t)scene * m * nz;
  const TI* p = (const TI*)P.p + (size_t)scene * nz;
  const TI* h = (const TI*)P.h + (size_t)scene * m;
  for (int i = tid; i < nz * nz; i += NT) S.Q[i] = (TC)Q[i];
  for (int i = tid; i < m * nz; i += NT) S.G[i] = (TC)G[i];
  for (int i = tid; i < nz; i += NT) S.p[i] = (TC)p[i];
  for (int i = tid; i < m; i += NT) S.h[i] = (TC)h[i];
  if (e > 0) {
    const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
    const TI* b = (const TI*)P.b + (size_t)scene * e;
    for (int i = tid; i < e * nz; i += NT) S.A[i] = (TC)A[i];
    for (int i = tid; i < e; i += NT) S.b[i] = (TC)b[i];
  }
  __syncthreads();
}

// engines.py:31-32,50-74 + world.py:144-234: build Q (diag), p = M v + dt f, G = [Jc; Jf; 0],
// h = [(Jc v) * restitution; 0; 0], mu per contact, A = Je, b = 0 directly in LDS.
// POST: the frictionless LCP of PdipmEngine.post_stabilization instead (engines.py:80-116): p = 0, G = Jc,
// h = gc = Jc v + Jc v * -restitutions, A = Je, b = ge = Je v, F = 0; m = S.m = ncs.
template <typename TI, typename TC, bool POST = false>
__device__ void assemble_scene(Scene<TC>& S, const StepArgs& P, int scene, int ncs) {
  // `ncs` = contacts of this scene (<= P.nc, the capacity the contact arrays are strided by); m = S.m = 4 ncs
  const int nb = P.nb, ncap = P.nc, nc = ncs, nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  const TI* Md = (const TI*)P.Mdiag + (size_t)scene * nz;
  const TI* vv = (const TI*)P.v + (size_t)scene * nz;
  const TI* ff = (const TI*)P.f + (size_t)scene * nz;
  const TI* rest = (const TI*)P.rest + (size_t)scene * nb;
  const TI* fric = (const TI*)P.fric + (size_t)scene * nb;
  const TI* cn = (const TI*)P.c_n + (size_t)scene * ncap * 2;
  const TI* c1 = (const TI*)P.c_p1 + (size_t)scene * ncap * 2;
  const TI* c2 = (const TI*)P.c_p2 + (size_t)scene * ncap * 2;
  const int32_t* i1 = P.c_i1 + (size_t)scene * ncap;
  const int32_t* i2 = P.c_i2 + (size_t)scene * ncap;
  S.nc = nc;
  for (int i = tid; i < nz * nz; i += NT) { const int r = i / nz, c = i - r * nz; S.Q[i] = (r == c) ? (TC)Md[r] : (TC)0; }
  for (int i = tid; i < m * nz; i += NT) S.G[i] = 0;
  for (int j = tid; j < nz; j += NT) S.p[j] = POST ? (TC)0 : (TC)momentum_entry<TI>(Md[j], vv[j], (TI)P.dt, ff[j]);   // engines.py:32 | :84
  for (int i = tid; i < m; i += NT) S.h[i] = 0;
  if (e > 0) {
    const TI* Je = (const TI*)P.Je + (size_t)scene * e * nz;
    for (int i = tid; i < e * nz; i += NT) S.A[i] = (TC)Je[i];
    for (int i = tid; i < e; i += NT) {
      TC ge = 0;
      if (POST) for (int j = 0; j < nz; ++j) ge += (TC)Je[i * nz + j] * (TC)vv[j];                     // ge = Je v (engines.py:86)
      S.b[i] = ge;
    }
  }
  __syncthreads();
  if (POST) {
    for (int c = tid; c < nc; c += NT) {
      const ContactRows<TI> r = make_contact<TI>(cn, c1, c2, i1, i2, rest, fric, vv, c);
      TC* gn = S.G + (size_t)c * nz;
      for (int q = 0; q < 3; ++q) { gn[3 * r.b1 + q] = (TC)r.jn[q]; }
      for (int q = 0; q < 3; ++q) { gn[3 * r.b2 + q] = (TC)r.jn[3 + q]; }
      S.h[c] = (TC)r.jv + (TC)r.jv * -(TC)r.rbar;                                                      // engines.py:87-89
    }
    __syncthreads();
    return;
  }
  for (int c = tid; c < nc; c += NT) {
    const ContactRows<TI> r = make_contact<TI>(cn, c1, c2, i1, i2, rest, fric, vv, c);
    TC* gn = S.G + (size_t)c * nz;                                       // Jc row            world.py:177-183
    TC* g0 = S.G + (size_t)(nc + 2 * c) * nz;                            // Jf rows 2c, 2c+1  world.py:196-210
    TC* g1 = g0 + nz;
    for (int q = 0; q < 3; ++q) {                                        // body 1 first, body 2 second
      gn[3 * r.b1 + q] = (TC)r.jn[q]; g0[3 * r.b1 + q] = (TC)r.jf[q]; g1[3 * r.b1 + q] = (TC)(-r.jf[q]);
    }
    for (int q = 0; q < 3; ++q) {
      gn[3 * r.b2 + q] = (TC)r.jn[3 + q]; g0[3 * r.b2 + q] = (TC)r.jf[3 + q]; g1[3 * r.b2 + q] = (TC)(-r.jf[3 + q]);
    }
    S.mu_c[c] = (TC)r.mu;
    S.h[c] = (TC)r.h;
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------
// the PDIPM loop (pdipm.py:49-179), shared by the dense and the fused kernels
// ------------------------------------------------------------------------------------------
template <typename TC, bool PIVOT, typename FT>
__device__ void pdipm_loop(Scene<TC>& S, const FT& F, TC eps, int max_iter, int lim, int& iters_out,
                           int& status, double* trace) {
  const int nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  // init: d = 1, factor, solve with (p, 0, -h, -b)                          (:57-63)
  for (int i = tid; i < m; i += NT) { S.d[i] = 1; S.rs[i] = 1; S.rz[i] = -S.h[i]; }
  for (int a = tid; a < e; a += NT) S.ry[a] = -S.b[a];
  __syncthreads();
  factor_T<TC, PIVOT>(S, S.rs);
  solve_kkt<TC, PIVOT>(S, S.p, (const TC*)nullptr, S.rz, e > 0 ? S.ry : (const TC*)nullptr, S.x, S.s, S.z, S.y);
  {
    TC smin = inf_of<TC>(), zmin = inf_of<TC>();
    for (int i = tid; i < m; i += NT) { smin = pmin(smin, S.s[i]); zmin = pmin(zmin, S.z[i]); }
    block_reduce2(smin, zmin, OpMin(), S.red);
    for (int i = tid; i < m; i += NT) {                                   // (:66-75)
      if (smin <= (TC)0) S.s[i] = S.s[i] - smin + (TC)1;
      if (zmin <= (TC)0) S.z[i] = S.z[i] - zmin + (TC)1;
    }
    __syncthreads();
  }
  TC best_resid = inf_of<TC>();
  bool have_best = false;
  int n_not = 0, iters = 0;
  if (m == 0) {                              // no inequality at all: the initialisation solve IS the answer
    for (int j = tid; j < nz; j += NT) S.bx[j] = S.x[j];          // (engines.py:36-50, the no-contact branch)
    for (int a = tid; a < e; a += NT) S.by[a] = S.y[a];
    __syncthreads();
    iters_out = 0;
    return;
  }
  for (int it = 0; it < max_iter; ++it) {
    // residuals                                                             (:82-96)
    for (int j = tid; j < nz; j += NT) {
      TC a = col_dot(S.G, nz, j, S.z, m) + row_dot(S.Q + (size_t)j * nz, S.x, nz) + S.p[j];
      if (e > 0) a += col_dot(S.A, nz, j, S.y, e);
      S.rx[j] = a;
    }
    for (int i = tid; i < m; i += NT) S.rz[i] = row_dot(S.G + (size_t)i * nz, S.x, nz) + S.s[i] - S.h[i];
    for (int a = tid; a < e; a += NT) S.ry[a] = row_dot(S.A + (size_t)a * nz, S.x, nz) - S.b[a];
    __syncthreads();
    F.sub_Fz(S.z, S.rz);
    __syncthreads();
    TC n_rx = 0, n_rz = 0, n_ry = 0, sz = 0;
    for (int j = tid; j < nz; j += NT) n_rx += S.rx[j] * S.rx[j];
    for (int i = tid; i < m; i += NT) { n_rz += S.rz[i] * S.rz[i]; sz += S.s[i] * S.z[i]; }
    for (int a = tid; a < e; a += NT) n_ry += S.ry[a] * S.ry[a];
    block_reduce2(n_rx, n_rz, OpSum(), S.red);
    block_reduce2(n_ry, sz, OpSum(), S.red);
    TC mu = sz / (TC)m; mu = mu < 0 ? -mu : mu;                            // (:91)
    const TC resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + (TC)m * mu;    // (:92-96)
    // d = z / s ; factor                                                    (:98-100)
    for (int i = tid; i < m; i += NT) { const TC dd = S.z[i] / S.s[i]; S.d[i] = dd; S.rs[i] = (TC)1 / dd; }
    __syncthreads();
    const bool singular = factor_T<TC, PIVOT>(S, S.rs);
    ++iters;
    if (trace) { if (tid == 0) { trace[4 * it + 0] = (double)resid; trace[4 * it + 1] = (double)mu; } }
    if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; break; }       // except: return best (:99-102)
    // best-iterate bookkeeping                                              (:107-132)
    const bool improved = !have_best || (resid < best_resid);
    if (improved) {
      best_resid = resid; n_not = 0; have_best = true;
      for (int j = tid; j < nz; j += NT) S.bx[j] = S.x[j];
      for (int i = tid; i < m; i += NT) { S.bs[i] = S.s[i]; S.bz[i] = S.z[i]; }
      for (int a = tid; a < e; a += NT) S.by[a] = S.y[a];
    } else {
      ++n_not;
    }
    if (n_not == lim || best_resid < eps || mu > mu_limit<TC>()) break;                  // (:133)
    // the iterate this pass would produce is never evaluated (:176-179): dead work, skipped (kept when the per-iteration
    // trace is requested, which also records this pass's sigma and step length)
    if (it == max_iter - 1 && !trace) break;
    // affine direction                                                      (:138-139)
    for (int i = tid; i < m; i += NT) S.rs[i] = S.z[i];
    __syncthreads();
    solve_kkt<TC, PIVOT>(S, S.rx, S.rs, S.rz, e > 0 ? S.ry : (const TC*)nullptr, S.ax, S.as, S.az, S.ay);
    TC alpha = pmin(step_pair(S, S.z, S.az, S.s, S.as), (TC)1);          // (:142-144)
    TC t3 = 0, t4 = 0;
    for (int i = tid; i < m; i += NT) {
      t3 += (S.s[i] + alpha * S.as[i]) * (S.z[i] + alpha * S.az[i]);
      t4 += S.s[i] * S.z[i];
    }
    block_reduce2(t3, t4, OpSum(), S.red);
    const TC r3 = t3 / t4, sig = r3 * r3 * r3;                             // (:146-150)
    for (int i = tid; i < m; i += NT) S.rs[i] = (-mu * sig + S.as[i] * S.az[i]) / S.s[i];   // (:153)
    __syncthreads();
    solve_kkt<TC, PIVOT>(S, (const TC*)nullptr, S.rs, (const TC*)nullptr, (const TC*)nullptr, S.cx, S.cs, S.cz, S.cy);
    for (int j = tid; j < nz; j += NT) S.cx[j] += S.ax[j];                 // (:160-163)
    for (int i = tid; i < m; i += NT) { S.cs[i] += S.as[i]; S.cz[i] += S.az[i]; }
    for (int a = tid; a < e; a += NT) S.cy[a] += S.ay[a];
    __syncthreads();
    alpha = pmin((TC)0.999 * step_pair(S, S.z, S.cz, S.s, S.cs), (TC)1);  // (:164-166)
    if (trace) { if (tid == 0) { trace[4 * it + 2] = (double)sig; trace[4 * it + 3] = (double)alpha; } }
    for (int j = tid; j < nz; j += NT) S.x[j] += alpha * S.cx[j];         // (:171-174)
    for (int i = tid; i < m; i += NT) { S.s[i] += alpha * S.cs[i]; S.z[i] += alpha * S.cz[i]; }
    for (int a = tid; a < e; a += NT) S.y[a] += alpha * S.cy[a];
    __syncthreads();
  }
  __syncthreads();
  iters_out = iters;
}

// workspace layout per scene (TC elements): R[m*m] Qi[nz*nz] GA[m*e] S11i[e*e] x[nz] s[m] z[m] y[e] (T[m*ldT])
template <typename TC>
__host__ __device__ inline size_t ws_elems(int nz, int m, int e, int ldT, int level) {
  size_t n = (size_t)m * m + (size_t)nz * nz + (size_t)m * e + (size_t)e * e + nz + 2 * (size_t)m + e;
  if (level != 1) n += (size_t)m * ldT + (size_t)m * nz;   // T and the prefactor scratch
  if (level == 2) n += 2 * (size_t)nz * nz + (size_t)m * nz;   // Q, Q^-1, G
  n += 1;                                                       // the LAST element of a block: the live contact count of a contact-list forward (lcp_step_kernel -> lcp_step_bwd_kernel)
  return (n + 31) & ~(size_t)31;
}

template <typename TC>
struct WsView {
  TC *R, *Qi, *GA, *S11i, *x, *s, *z, *y, *T;
  __device__ WsView(void* ws, size_t stride_elems, int scene, int nz, int m, int e) {
    TC* q = reinterpret_cast<TC*>(ws) + stride_elems * (size_t)scene;
    R = q; q += (size_t)m * m; Qi = q; q += (size_t)nz * nz; GA = q; q += (size_t)m * e;
    S11i = q; q += (size_t)e * e; x = q; q += nz; s = q; q += m; z = q; q += m; y = q; q += e; T = q;
  }
};

template <typename TI, typename TC>
__device__ void store_solution(Scene<TC>& S, WsView<TC>& W, TI* x, TI* y, TI* z, TI* s, int& status) {
  const int nz = S.nz, m = S.m, e = S.e, tid = threadIdx.x;
  int bad = 0;
  for (int j = tid; j < nz; j += NT) { const TC a = S.bx[j]; W.x[j] = a; if (x) x[j] = (TI)a; bad |= (a != a); }
  for (int i = tid; i < m; i += NT) {
    const TC a = S.bz[i], c = S.bs[i];
    W.z[i] = a; W.s[i] = c; if (z) z[i] = (TI)a; if (s) s[i] = (TI)c; bad |= (a != a) | (c != c);
  }
  for (int a = tid; a < e; a += NT) { const TC yy = S.by[a]; W.y[a] = yy; if (y) y[a] = (TI)yy; }
  for (int i = tid; i < nz * nz; i += NT) W.Qi[i] = S.Qi[i];
  for (int i = tid; i < m * e; i += NT) W.GA[i] = S.GA[i];
  for (int i = tid; i < e * e; i += NT) W.S11i[i] = S.S11i[i];
  if (__syncthreads_or(bad)) status |= LCP_ST_NAN;
}

template <typename TI, typename TC, bool PIVOT>
__global__ void __launch_bounds__(NT) lcp_fwd_kernel(FwdArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.tag) *P.tag = P.tag_value;   // workspace trailer: which kernel family laid it out
  if (P.cls && P.cls[scene] >= 2) return;              // served by lcp_big.hip / lcp_primal.hip (contact-structured, diagonal Q)
  const int nz = P.nz, m = P.m, e = P.e;
  WsView<TC> W(P.ws, P.ws_stride, scene, nz, m, e);
  Scene<TC> S;
  carve(S, smem, nz, m, e, P.ldT, P.t_in_lds, W.T);
  S.R = W.R;
  load_dense<TI, TC>(S, P, scene);
  FDense<TI, TC> F{(const TI*)P.F + (size_t)scene * m * m, m};
  int status = prefactor(S, F);
  int iters = 0;
  double* trace = P.trace ? P.trace + (size_t)scene * 4 * P.max_iter : nullptr;
  pdipm_loop<TC, PIVOT>(S, F, (TC)P.eps, P.max_iter, P.lim, iters, status, trace);
  TI* x = (TI*)P.x + (size_t)scene * nz;
  TI* z = (TI*)P.z + (size_t)scene * m;
  TI* s = (TI*)P.s + (size_t)scene * m;
  TI* y = (e > 0 && P.y) ? (TI*)P.y + (size_t)scene * e : nullptr;
  store_solution<TI, TC>(S, W, x, y, z, s, status);
  if (threadIdx.x == 0) {
    if (P.iters) P.iters[scene] = iters;
    if (P.status) P.status[scene] = status;
  }
}

// PdipmEngine.post_stabilization (engines.py:80-116) for one scene per workgroup + the correction move of
// World.step_dt (world.py:109-117): dp = -x of the frictionless LCP; p_out = p + (dp / 2) dt_scene.
template <typename TI, typename TC, bool PIVOT>
__global__ void __launch_bounds__(NT) lcp_post_stab_kernel(StepArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.tag) *P.tag = P.tag_value;
  const int ncap = P.nc;
  int ncs = ncap;
  int truncated = 0;                                                       // the detection kernel found more contacts than the list holds
  if (P.c_count) { const int c = P.c_count[scene]; ncs = c < ncap ? (c < 0 ? 0 : c) : ncap; truncated = c > ncap ? LCP_ST_TRUNCATED : 0; }
  const int nz = 3 * P.nb, m = ncs, e = P.e;
  WsView<TC> W(P.ws, P.ws_stride, scene, nz, 4 * ncap, e);
  Scene<TC> S;
  carve(S, smem, nz, m, e, P.ldT, P.t_in_lds, W.T);
  S.R = W.R;
  assemble_scene<TI, TC, true>(S, P, scene, ncs);
  FZero<TC> F;
  int status = prefactor(S, F) | truncated;
  int iters = 0;
  pdipm_loop<TC, PIVOT>(S, F, (TC)P.eps, P.max_iter, P.lim, iters, status, nullptr);   // (m == 0: the direct KKT solve, :92-103)
  __syncthreads();
  // (round 6) the iterate of the ncs-row LCP and the contact count stay in the workspace for lcp_step_bwd_kernel<.., POST>
  if (threadIdx.x == 0) (reinterpret_cast<TC*>(P.ws) + P.ws_stride * (size_t)scene)[P.ws_stride - 1] = (TC)ncs;
  store_solution<TI, TC>(S, W, (TI*)nullptr, (TI*)nullptr, (TI*)nullptr, (TI*)nullptr, status);
  int bad = 0;
  TI* dpo = (TI*)P.v_new + (size_t)scene * nz;
  for (int j = threadIdx.x; j < nz; j += NT) {
    const TC dp = -S.bx[j];                                               // engines.py:115
    bad |= (dp != dp);
    dpo[j] = (TI)dp;
    if (P.p_out64) {                                                      // world.py:110-117: dp /= 2 ; body.move(dt)
      const double dts = P.dt_scene ? P.dt_scene[scene] : P.dt;
      P.p_out64[(size_t)scene * nz + j] = P.pos64[(size_t)scene * nz + j] + ((double)dp * 0.5) * dts;
    }
  }
  if (__syncthreads_or(bad)) status |= LCP_ST_NAN;
  if (threadIdx.x == 0) {
    if (P.iters) P.iters[scene] = iters;
    if (P.status) P.status[scene] = status;
  }
}

// Fused simulation step: assembly + solve + new_v = -x + p <- p + new_v dt (bodies.py:80-82).
template <typename TI, typename TC, bool PIVOT>
__global__ void __launch_bounds__(NT) lcp_step_kernel(StepArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x;
  if (blockIdx.x == 0 && threadIdx.x == 0 && P.tag) *P.tag = P.tag_value;
  const int ncap = P.nc, mcap = 4 * ncap;                                 // capacity: array strides, workspace layout
  int ncs = ncap;                                                         // contacts of this scene (engines.py:36,51)
  int truncated = 0;                                                       // the detection kernel found more contacts than the list holds
  if (P.c_count) { const int c = P.c_count[scene]; ncs = c < ncap ? (c < 0 ? 0 : c) : ncap; truncated = c > ncap ? LCP_ST_TRUNCATED : 0; }
  const int nz = 3 * P.nb, m = 4 * ncs, e = P.e;
  WsView<TC> W(P.ws, P.ws_stride, scene, nz, mcap, e);
  Scene<TC> S;
  carve(S, smem, nz, m, e, P.ldT, P.t_in_lds, W.T);
  S.R = W.R;
  assemble_scene<TI, TC>(S, P, scene, ncs);
  FContact<TC> F{S.mu_c, ncs};
  int status = prefactor(S, F) | truncated;
  int iters = 0;
  pdipm_loop<TC, PIVOT>(S, F, (TC)P.eps, P.max_iter, P.lim, iters, status, nullptr);
  TI* z = P.z ? (TI*)P.z + (size_t)scene * mcap : nullptr;
  TI* s = P.s ? (TI*)P.s + (size_t)scene * mcap : nullptr;
  TI* y = (e > 0 && P.y) ? (TI*)P.y + (size_t)scene * e : nullptr;
  if (threadIdx.x == 0) (reinterpret_cast<TC*>(P.ws) + P.ws_stride * (size_t)scene)[P.ws_stride - 1] = (TC)ncs;   // (for lcp_step_bwd_kernel)
  if (ncs == ncap) {
    store_solution<TI, TC>(S, W, (TI*)nullptr, y, z, s, status);
  } else {
    // fewer contacts than the capacity: the multipliers go to the row layout of a capacity-sized LCP
    // ([normal | friction pairs | gamma] blocks of ncap, 2 ncap, ncap rows), padded slots are 0.  The workspace keeps the iterate of the
    // 4 ncs-row LCP that was solved (round 6: lcp_step_bwd_kernel reads it; the dense backward of lcp_pdipm_backward_f32 still wants a
    // capacity-sized one and finds none here)
    store_solution<TI, TC>(S, W, (TI*)nullptr, (TI*)nullptr, (TI*)nullptr, (TI*)nullptr, status);
    __syncthreads();
    int bad = 0;
    for (int i = threadIdx.x; i < mcap; i += NT) { if (z) z[i] = (TI)0; if (s) s[i] = (TI)0; }
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += NT) {
      const int slot = (i < ncs) ? i : (i < 3 * ncs) ? ncap + (i - ncs) : 3 * ncap + (i - 3 * ncs);
      const TC a = S.bz[i], c = S.bs[i];
      if (z) z[slot] = (TI)a;
      if (s) s[slot] = (TI)c;
      bad |= (a != a) | (c != c);
    }
    for (int j = threadIdx.x; j < nz; j += NT) bad |= (S.bx[j] != S.bx[j]);
    for (int a = threadIdx.x; a < e; a += NT) if (y) y[a] = (TI)S.by[a];
    if (__syncthreads_or(bad)) status |= LCP_ST_NAN;
  }
  TI* vn = (TI*)P.v_new + (size_t)scene * nz;
  for (int j = threadIdx.x; j < nz; j += NT) {
    const TC nv = -S.bx[j];                                               // engines.py:76-77
    vn[j] = (TI)nv;
    if (P.p_new) ((TI*)P.p_new)[(size_t)scene * nz + j] = (TI)((TC)((const TI*)P.pos)[(size_t)scene * nz + j] + nv * (TC)P.dt);   // bodies.py:81
  }
  if (threadIdx.x == 0) {
    if (P.iters) P.iters[scene] = iters;
    if (P.status) P.status[scene] = status;
  }
}

// Backward of the fused step w.r.t. its PHYSICAL inputs for any size the generic plan holds (round 6, VERDICT r05 item 7: the reference has no
// size limit - world.py:139-142 - and a recorded step beyond the fused kernels used to leave the device path for torch: dense tensors, one
// host synchronisation per step, torch.linalg.solve).  LCPFunction.backward (lcp.py:37-64) at the iterate lcp_step_kernel kept - the
// 4 ncs-row LCP of the scene's own contact count, T = R + diag(s / z) factored with partial pivoting as there - and the rank-1 gradients
// contracted through the assembly (engines.py:31-32,50-74; world.py:144-234) as lcp_primal_step.inc does: dp = dx, dQ_jj = dx_j x_j,
// dG_row = dlam_row x + lam_row dx, dh = -dlam, dF[gamma_c, n_c] = -dlam_gamma lam_n, dA = dnu (x) x + nu (x) dx.
// POST: the backward of lcp_post_stab_kernel (engines.py:80-116: ncs rows, G = Jc, p = 0, F = 0, h = (Jc v)(1 - rbar), b = Je v; dp = -x)
// with respect to Mdiag, v, rest, the contact frame and Je - lcp_post_stabilization_backward_f32 beyond the one-wave sizes.
template <typename TI, typename TC, bool PIVOT, bool POST = false>
__global__ void __launch_bounds__(NT) lcp_step_bwd_kernel(StepArgs P, StepBwdArgs Gd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x, tid = threadIdx.x;
  const int nb = P.nb, ncap = P.nc, mcap = 4 * ncap, nz = 3 * nb, e = P.e;
  const TC* blk = reinterpret_cast<const TC*>(P.ws) + P.ws_stride * (size_t)scene;
  int ncs = (int)blk[P.ws_stride - 1];
  ncs = ncs < 0 ? 0 : (ncs > ncap ? ncap : ncs);
  const int m = POST ? ncs : 4 * ncs;
  WsView<TC> W(P.ws, P.ws_stride, scene, nz, mcap, e);
  Scene<TC> S;
  carve(S, smem, nz, m, e, P.ldT, P.t_in_lds, W.T);
  S.R = W.R;
  assemble_scene<TI, TC, POST>(S, P, scene, ncs);                          // G, A, Q, mu_c as the forward formed them
  for (int i = tid; i < nz * nz; i += NT) S.Qi[i] = W.Qi[i];
  for (int i = tid; i < m * e; i += NT) S.GA[i] = W.GA[i];
  for (int i = tid; i < e * e; i += NT) S.S11i[i] = W.S11i[i];
  for (int a = tid; a < e; a += NT) S.y[a] = W.y[a];
  const TI* g = (const TI*)Gd.dl_dv + (size_t)scene * nz;
  const bool tag_ok = !P.tag || *P.tag == P.tag_value;                     // (another family's workspace: NaN gradients instead of a misread)
  // v_new = -x (engines.py:76-77)  =>  d(loss)/dx = -d(loss)/d(v_new)
  for (int j = tid; j < nz; j += NT) { S.x[j] = W.x[j]; S.rx[j] = tag_ok ? -(TC)g[j] : nan_of<TC>(); }
  for (int i = tid; i < m; i += NT) {
    const TC zz = W.z[i], ss = W.s[i];
    S.z[i] = zz; S.s[i] = ss;
    const TC dd = zz / ss;                                                 // lcp.py:44
    S.d[i] = dd; S.rs[i] = (TC)1 / dd;
  }
  if (tid == 0) {                                                          // (the noise-pivot guard of lcp_bwd_kernel)
    TC mx = 0;
    for (int i = 0; i < m; ++i) { TC a = S.R[(size_t)i * m + i]; a = a < 0 ? -a : a; mx = a > mx ? a : mx; }
    S.cs[0] = mx;
  }
  __syncthreads();
  const TC rmax = S.cs[0];
  __syncthreads();
  if (m > 0 && factor_T<TC, PIVOT>(S, S.rs, (TC)1e-13 * rmax)) {           // lcp.py:46
    for (int i = tid; i < m; i += NT) {
      const TC f = (TC)1e-12 * S.R[(size_t)i * m + i];
      if (S.rs[i] < f) { S.rs[i] = f; S.d[i] = (TC)1 / f; }
    }
    __syncthreads();
    factor_T<TC, PIVOT>(S, S.rs);
  }
  solve_kkt<TC, PIVOT>(S, S.rx, (const TC*)nullptr, (const TC*)nullptr, (const TC*)nullptr, S.cx, S.cs, S.cz, S.cy);   // lcp.py:47-50: dx = cx, dlam = cz, dnu = cy
  __syncthreads();
  // ---- contraction through the assembly ---------------------------------------------------------------------------------------------
  const TI* Md = (const TI*)P.Mdiag + (size_t)scene * nz;
  const TI* vv = (const TI*)P.v + (size_t)scene * nz;
  const TI* rest = (const TI*)P.rest + (size_t)scene * nb;
  const TI* cn = (const TI*)P.c_n + (size_t)scene * ncap * 2;
  const TI* c1 = (const TI*)P.c_p1 + (size_t)scene * ncap * 2;
  const TI* c2 = (const TI*)P.c_p2 + (size_t)scene * ncap * 2;
  const int32_t* i1 = P.c_i1 + (size_t)scene * ncap;
  const int32_t* i2 = P.c_i2 + (size_t)scene * ncap;
  TC* CR = S.hz;                                                           // per contact: its share of d(loss)/d(restitution), d(loss)/d(friction),
  TC* CF = S.rz;                                                           // and dh_c rbar_c (the weight of Jc^T in d(loss)/dv) - m-space scratch, free now
  TC* GH = S.rs;
  for (int c = tid; c < ncs; c += NT) {
    const TC nx = (TC)cn[2 * c], ny = (TC)cn[2 * c + 1];
    const TC p1x = (TC)c1[2 * c], p1y = (TC)c1[2 * c + 1], p2x = (TC)c2[2 * c], p2y = (TC)c2[2 * c + 1];
    const int b1 = i1[c], b2 = i2[c];
    const TC rbar = (TC)0.5 * ((TC)rest[b1] + (TC)rest[b2]);
    const TC jnd[6] = {p1x * ny - p1y * nx, nx, ny, -(p2x * ny - p2y * nx), -nx, -ny};      // world.py:177-183
    const TC zn = S.z[c], dn = S.cz[c];
    TC zf1 = 0, zf2 = 0, df1 = 0, df2 = 0, dg = 0;
    if (!POST) { zf1 = S.z[ncs + 2 * c]; zf2 = S.z[ncs + 2 * c + 1]; df1 = S.cz[ncs + 2 * c]; df2 = S.cz[ncs + 2 * c + 1]; dg = S.cz[3 * ncs + c]; }
    const TC gh = -dn;                                                     // dh = -dlam (lcp.py:56)
    const TC af = df1 - df2, lf = zf1 - zf2;                               // Jf rows are +jt, -jt (world.py:191-192)
    const TC hw = POST ? gh * ((TC)1 - rbar) : gh * rbar;                  // h = (Jc v) rbar | POST: (Jc v) + (Jc v) * -rbar (engines.py:89)
    TC gjn[6], gjf[6], jnv = 0;
    for (int q = 0; q < 6; ++q) {
      const int col = (q < 3) ? 3 * b1 + q : 3 * b2 + (q - 3);
      const TC xq = S.x[col], dxq = S.cx[col], vq = (TC)vv[col];
      jnv += jnd[q] * vq;
      gjn[q] = dn * xq + zn * dxq + hw * vq;                               // dG row n (lcp.py:53) + h through Jc
      gjf[q] = af * xq + lf * dxq;
    }
    GH[c] = hw;
    CR[c] = (TC)0.5 * (POST ? -gh * jnv : gh * jnv);                       // rbar = (rest_b1 + rest_b2) / 2 (world.py:144-151)
    CF[c] = (TC)0.5 * (-dg * zn);                                          // dF[gamma_c, n_c] = -dlam_g lam_n (lcp.py:54), F = mu there
    const TC dnx = -gjn[0] * p1y + gjn[1] + gjn[3] * p2y - gjn[4] - gjf[0] * p1x - gjf[2] + gjf[3] * p2x + gjf[5];
    const TC dny = gjn[0] * p1x + gjn[2] - gjn[3] * p2x - gjn[5] - gjf[0] * p1y + gjf[1] + gjf[3] * p2y - gjf[4];
    const size_t cb = (size_t)scene * ncap + c;
    if (Gd.dcn) { ((TI*)Gd.dcn)[cb * 2] = (TI)dnx; ((TI*)Gd.dcn)[cb * 2 + 1] = (TI)dny; }
    if (Gd.dcp1) { ((TI*)Gd.dcp1)[cb * 2] = (TI)(gjn[0] * ny - gjf[0] * nx); ((TI*)Gd.dcp1)[cb * 2 + 1] = (TI)(-gjn[0] * nx - gjf[0] * ny); }
    if (Gd.dcp2) { ((TI*)Gd.dcp2)[cb * 2] = (TI)(-gjn[3] * ny + gjf[3] * nx); ((TI*)Gd.dcp2)[cb * 2 + 1] = (TI)(gjn[3] * nx + gjf[3] * ny); }
  }
  for (int c = ncs + tid; c < ncap; c += NT) {                             // padded contact slots: zero gradients
    const size_t cb = (size_t)scene * ncap + c;
    if (Gd.dcn) { ((TI*)Gd.dcn)[cb * 2] = 0; ((TI*)Gd.dcn)[cb * 2 + 1] = 0; }
    if (Gd.dcp1) { ((TI*)Gd.dcp1)[cb * 2] = 0; ((TI*)Gd.dcp1)[cb * 2 + 1] = 0; }
    if (Gd.dcp2) { ((TI*)Gd.dcp2)[cb * 2] = 0; ((TI*)Gd.dcp2)[cb * 2 + 1] = 0; }
  }
  __syncthreads();
  for (int j = tid; j < nz; j += NT) {                                     // Q = diag(M) (dQ, lcp.py:59-60), p = M v + dt f, h = (Jc v) rbar
    const size_t o = (size_t)scene * nz + j;
    const TC dx = S.cx[j], x = S.x[j], md = (TC)Md[j], v = (TC)vv[j];
    TC dvh = 0;
    for (int c = 0; c < ncs; ++c) dvh += S.G[(size_t)c * nz + j] * GH[c];   // Jc^T (dh rbar), contacts in list order
    if (POST) {                                                            // p = 0; v enters through gc and ge = Je v: db = -dnu (lcp.py:58)
      for (int a = 0; a < e; ++a) dvh -= S.A[(size_t)a * nz + j] * S.cy[a];
      if (Gd.dMdiag) ((TI*)Gd.dMdiag)[o] = (TI)(dx * x);
      if (Gd.dv) ((TI*)Gd.dv)[o] = (TI)dvh;
    } else {
      if (Gd.dMdiag) ((TI*)Gd.dMdiag)[o] = (TI)(dx * x + dx * v);
      if (Gd.dv) ((TI*)Gd.dv)[o] = (TI)(dx * md + dvh);
      if (Gd.df) ((TI*)Gd.df)[o] = (TI)(dx * (TC)P.dt);
    }
  }
  if (Gd.dJe && e > 0) {                                                    // dA = dnu (x) x + nu (x) dx (lcp.py:57; A = Je) | POST: + db (x) v
    TI* o = (TI*)Gd.dJe + (size_t)scene * e * nz;
    for (int i = tid; i < e * nz; i += NT) {
      const int a = i / nz, j = i - a * nz;
      TC t = S.cy[a] * S.x[j] + S.y[a] * S.cx[j];
      if (POST) t -= S.cy[a] * (TC)vv[j];
      o[i] = (TI)t;
    }
  }
  for (int b = tid; b < nb; b += NT) {                                      // per-body sums over the contacts, fixed order
    TC ar = 0, af = 0;
    for (int c = 0; c < ncs; ++c) {
      const TC w = ((i1[c] == b) ? (TC)1 : (TC)0) + ((i2[c] == b) ? (TC)1 : (TC)0);
      if (w != (TC)0) { ar += w * CR[c]; af += w * CF[c]; }
    }
    if (Gd.drest) ((TI*)Gd.drest)[(size_t)scene * nb + b] = (TI)ar;
    if (Gd.dfric) ((TI*)Gd.dfric)[(size_t)scene * nb + b] = (TI)af;
  }
}

// Stand-alone assembly to dense tensors (for LCPFunction users / parity of the assembly itself).
template <typename TI>
__global__ void __launch_bounds__(NT) lcp_assemble_kernel(StepArgs P, TI* Q, TI* p, TI* G, TI* h, TI* A, TI* b, TI* Fo) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x;
  const int nz = 3 * P.nb, nc = P.nc, m = 4 * nc, e = P.e, tid = threadIdx.x;
  Scene<TI> S;
  carve(S, smem, nz, m, e, m, false, (TI*)nullptr);
  assemble_scene<TI, TI>(S, P, scene, nc);
  FContact<TI> F{S.mu_c, nc};
  for (int i = tid; i < nz * nz; i += NT) Q[(size_t)scene * nz * nz + i] = S.Q[i];
  for (int i = tid; i < nz; i += NT) p[(size_t)scene * nz + i] = S.p[i];
  for (int i = tid; i < m * nz; i += NT) G[(size_t)scene * m * nz + i] = S.G[i];
  for (int i = tid; i < m; i += NT) h[(size_t)scene * m + i] = S.h[i];
  if (e > 0) {
    for (int i = tid; i < e * nz; i += NT) A[(size_t)scene * e * nz + i] = S.A[i];
    for (int i = tid; i < e; i += NT) b[(size_t)scene * e + i] = S.b[i];
  }
  for (int idx = tid; idx < m * m; idx += NT) Fo[(size_t)scene * m * m + idx] = F.at(idx / m, idx % m);
}

// LCPFunction.backward (lcp.py:37-64).
template <typename TI, typename TC, bool PIVOT>
__global__ void __launch_bounds__(NT) lcp_bwd_kernel(BwdArgs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x, tid = threadIdx.x;
  if (P.cls && P.cls[scene] >= 2) return;
  if (P.tag && P.skip_tag && *P.tag == P.skip_tag) return;        // (see BwdArgs::skip_tag: the partner family's backward serves this call)
  const int nz = P.nz, m = P.m, e = P.e;
  WsView<TC> W(P.ws, P.ws_stride, scene, nz, m, e);
  Scene<TC> S;
  carve(S, smem, nz, m, e, P.ldT, P.t_in_lds, W.T);
  S.R = W.R;
  const TI* G = (const TI*)P.G + (size_t)scene * m * nz;
  for (int i = tid; i < m * nz; i += NT) S.G[i] = (TC)G[i];
  for (int i = tid; i < nz * nz; i += NT) S.Qi[i] = W.Qi[i];
  if (e > 0) {
    const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
    for (int i = tid; i < e * nz; i += NT) S.A[i] = (TC)A[i];
    for (int i = tid; i < m * e; i += NT) S.GA[i] = W.GA[i];
    for (int i = tid; i < e * e; i += NT) S.S11i[i] = W.S11i[i];
    for (int a = tid; a < e; a += NT) S.y[a] = W.y[a];
  }
  const TI* g = (const TI*)P.dl_dx + (size_t)scene * nz;
  const bool tag_ok = !P.tag || *P.tag == P.tag_value;           // (another family's workspace: NaN gradients instead of a misread)
  for (int j = tid; j < nz; j += NT) { S.x[j] = W.x[j]; S.rx[j] = tag_ok ? (TC)g[j] : nan_of<TC>(); }
  for (int i = tid; i < m; i += NT) {
    const TC zz = W.z[i], ss = W.s[i];
    S.z[i] = zz; S.s[i] = ss;
    const TC dd = zz / ss;                                                 // lcp.py:44
    S.d[i] = dd; S.rs[i] = (TC)1 / dd;
  }
  // (round 5) A solve that converged to rounding leaves s / z ~ 1e-16 on its active rows, lost against an R that redundant contact
  // points make singular (two sticking points on one interface have identical tangential rows): the elimination then divides by
  // rounding noise - pivoting or not - and returns multipliers of 1e15 whose cancellation leaves dx off by O(1) with a residual of
  // 1e-16 of their size (one scene of 1024 on fp64 two-point stacks: profiles/r05_own_iterate_probe.txt; the reference's own solve has
  // the same exposure).  A pivot below 1e-13 of R's largest diagonal entry, or an exact zero, repeats the factorisation with s / z
  // floored at 1e-12 x the row's diagonal of R, as the four-scenes-per-wave contact-space backward does (factor_bwd_q).
  if (tid == 0) {
    TC mx = 0;
    for (int i = 0; i < m; ++i) { TC a = S.R[(size_t)i * m + i]; a = a < 0 ? -a : a; mx = a > mx ? a : mx; }
    S.cs[0] = mx;                                                          // (cs is written by solve_kkt below; free until then)
  }
  __syncthreads();
  const TC rmax = S.cs[0];
  __syncthreads();
  const bool adj = P.adjoint != 0;
  if (factor_T<TC, PIVOT>(S, S.rs, (TC)1e-13 * rmax, adj)) {               // lcp.py:46
    for (int i = tid; i < m; i += NT) {
      const TC f = (TC)1e-12 * S.R[(size_t)i * m + i];
      if (S.rs[i] < f) { S.rs[i] = f; S.d[i] = (TC)1 / f; }
    }
    __syncthreads();
    factor_T<TC, PIVOT>(S, S.rs, (TC)0, adj);
  }
  solve_kkt<TC, PIVOT>(S, S.rx, (const TC*)nullptr, (const TC*)nullptr, (const TC*)nullptr,
                       S.cx, S.cs, S.cz, S.cy);                           // lcp.py:47-50
  // outer products (lcp.py:52-61); dx = cx, dlam = cz, dnu = cy
  if (P.dp) { TI* o = (TI*)P.dp + (size_t)scene * nz; for (int j = tid; j < nz; j += NT) o[j] = (TI)S.cx[j]; }
  if (P.dh) { TI* o = (TI*)P.dh + (size_t)scene * m; for (int i = tid; i < m; i += NT) o[i] = (TI)(-S.cz[i]); }
  if (P.db && e > 0) { TI* o = (TI*)P.db + (size_t)scene * e; for (int a = tid; a < e; a += NT) o[a] = (TI)(-S.cy[a]); }
  if (P.dQ) {
    TI* o = (TI*)P.dQ + (size_t)scene * nz * nz;
    for (int idx = tid; idx < nz * nz; idx += NT) {
      const int i = idx / nz, j = idx - i * nz;
      o[idx] = (TI)((TC)0.5 * (S.cx[i] * S.x[j] + S.x[i] * S.cx[j]));
    }
  }
  if (P.dG) {
    TI* o = (TI*)P.dG + (size_t)scene * m * nz;
    for (int idx = tid; idx < m * nz; idx += NT) {
      const int i = idx / nz, j = idx - i * nz;
      o[idx] = (TI)(S.cz[i] * S.x[j] + S.z[i] * S.cx[j]);
    }
  }
  if (P.dA && e > 0) {
    TI* o = (TI*)P.dA + (size_t)scene * e * nz;
    for (int idx = tid; idx < e * nz; idx += NT) {
      const int a = idx / nz, j = idx - a * nz;
      o[idx] = (TI)(S.cy[a] * S.x[j] + S.y[a] * S.cx[j]);
    }
  }
  if (P.dF) {
    TI* o = (TI*)P.dF + (size_t)scene * m * m;
    for (int idx = tid; idx < m * m; idx += NT) {
      const int i = idx / m, j = idx - i * m;
      o[idx] = (TI)(-S.cz[i] * S.z[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host-side launch helpers (called from lcp_api.cpp)
// ------------------------------------------------------------------------------------------
constexpr size_t LDS_LIMIT = 160 * 1024;

Plan make_plan(int nz, int m, int e, int csize) {
  Plan pl;
  pl.ldT = m | 1;                      // odd leading dimension: conflict-free column walks
  pl.t_in_lds = 1;
  pl.lds_bytes = lds_bytes_for(nz, m, e, pl.ldT, 1, csize);
  for (int level = 0; pl.lds_bytes > LDS_LIMIT && level <= 2; level += 2) {       // 1 -> 0 -> 2: see carve()
    pl.t_in_lds = level;
    pl.lds_bytes = lds_bytes_for(nz, m, e, pl.ldT, level, csize);
  }
  pl.ok = pl.lds_bytes <= LDS_LIMIT;
  pl.ws_stride = (csize == 8) ? ws_elems<double>(nz, m, e, pl.ldT, pl.t_in_lds)
                              : ws_elems<float>(nz, m, e, pl.ldT, pl.t_in_lds);
  return pl;
}

// Raise the dynamic-LDS cap of a kernel once per (kernel, size) - gfx950 allows 160 KiB.
template <typename K>
static int set_lds(K kernel, size_t bytes) {
  // stateless on purpose: the opt-in is per device and per kernel, a cached "granted" size would be wrong on the second
  // GPU of a process and racy between host threads.  The attribute call is a host-side table update (no launch, no sync).
  if (bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
    return LCP_E_LAUNCH;
  return 0;
}

template <typename TI, typename TC, bool PIVOT>
static int launch_fwd_t(const FwdArgs& P, size_t lds, hipStream_t st) {
  auto k = lcp_fwd_kernel<TI, TC, PIVOT>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, st, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <typename TI, typename TC, bool PIVOT>
static int launch_bwd_t(const BwdArgs& P, size_t lds, hipStream_t st) {
  auto k = lcp_bwd_kernel<TI, TC, PIVOT>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, st, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <typename TI, typename TC, bool PIVOT>
static int launch_step_t(const StepArgs& P, size_t lds, hipStream_t st) {
  auto k = lcp_step_kernel<TI, TC, PIVOT>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, st, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int generic_forward(const FwdArgs& P, int io_f64, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (io_f64) return launch_fwd_t<double, double, true>(P, lds, st);
  if (compute == LCP_COMPUTE_F64) return launch_fwd_t<float, double, true>(P, lds, st);
  return launch_fwd_t<float, float, true>(P, lds, st);
}
int generic_backward(const BwdArgs& P, int io_f64, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (io_f64) return launch_bwd_t<double, double, true>(P, lds, st);
  if (compute == LCP_COMPUTE_F64) return launch_bwd_t<float, double, true>(P, lds, st);
  return launch_bwd_t<float, float, true>(P, lds, st);
}
template <typename TI, typename TC, bool PIVOT>
static int launch_post_stab_t(const StepArgs& P, size_t lds, hipStream_t st) {
  auto k = lcp_post_stab_kernel<TI, TC, PIVOT>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, st, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
int generic_post_stab(const StepArgs& P, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64) return launch_post_stab_t<float, double, true>(P, lds, st);
  return launch_post_stab_t<float, float, true>(P, lds, st);
}
template <typename TI, typename TC, bool PIVOT, bool POST = false>
static int launch_step_bwd_t(const StepArgs& P, const StepBwdArgs& Gd, size_t lds, hipStream_t st) {
  auto k = lcp_step_bwd_kernel<TI, TC, PIVOT, POST>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, st, P, Gd);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
int generic_step_backward(const StepArgs& P, const StepBwdArgs& Gd, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64) return launch_step_bwd_t<float, double, true>(P, Gd, lds, st);
  return launch_step_bwd_t<float, float, true>(P, Gd, lds, st);
}
int generic_post_stab_backward(const StepArgs& P, const StepBwdArgs& Gd, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64) return launch_step_bwd_t<float, double, true, true>(P, Gd, lds, st);
  return launch_step_bwd_t<float, float, true, true>(P, Gd, lds, st);
}
int generic_step(const StepArgs& P, int compute, size_t lds, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (compute == LCP_COMPUTE_F64) return launch_step_t<float, double, true>(P, lds, st);
  return launch_step_t<float, float, true>(P, lds, st);
}
int generic_assemble(const StepArgs& P, float* Q, float* p, float* G, float* h, float* A, float* b, float* F,
                     void* stream) {
  const int nz = 3 * P.nb, m = 4 * P.nc;
  size_t lds = lds_bytes_for(nz, m, P.e, m, 0, 4);
  if (lds > LDS_LIMIT) return LCP_E_TOOLARGE;
  auto k = lcp_assemble_kernel<float>;
  if (set_lds(k, lds)) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(P.B), dim3(NT), lds, (hipStream_t)stream, P, Q, p, G, h, A, b, F);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
