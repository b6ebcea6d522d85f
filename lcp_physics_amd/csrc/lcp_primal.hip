// lcp_primal.hip - contact-structured PDIPM for scenes of up to 64 contacts with the KKT systems solved in BODY space:
// one wavefront per scene, lane c = contact c, lane j = row j of an (nz + neq)-square system.
//
// The reference eliminates x first and factors T = G Q^-1 G^T + F + D^-1 in contact space (pdipm.py:325-454: nineq = 4 nc
// rows; lcp_quad.hip / lcp_big.hip reduce that exactly to 2 nc).  The same Newton step can be had by eliminating the
// inequality block first - the block (F + D^-1) of the mixed contact LCP (engines.py:67-73) is block diagonal, one 4 x 4 block
//        [ Dn  0   0   0 ]     rows / columns: normal, friction +, friction -, cone        D = s / z
//    M = [ 0   D1  0   1 ]     (F = [[0, 0, 0], [0, 0, E], [mu, -E^T, 0]])
//        [ 0   0   D2  1 ]
//        [ mu  -1  -1  Dg ]
// per contact, inverted in closed form inside the contact's lane:
//    G dx - M dz = q,  q = rs / d - rz                 (the inequality rows of the step equations behind pdipm.py:325-354)
//    (Q + G^T M^-1 G) dx + A^T dy = -rx + G^T M^-1 q,  A dx = -ry
//    dz = M^-1 (G dx - q),   ds = (-rs - dz) / d
// With G = [Jc; +Jt; -Jt; 0] only a 2 x 2 matrix per contact enters G^T M^-1 G (B00 Jc^T Jc + B10 Jt^T Jc + B11 Jt^T Jt), and a
// contact touches two bodies: its contribution is a 6 x 6 block.  The system is (nz + neq) square - 36 for BASELINE config 5
// (11 bodies, 64 contacts, nineq 256) instead of 256 (reference) or 128 (reduced contact space): 45 x fewer LU flops, and
// the whole scene fits one wavefront's registers.  Same iterates as the reference in exact arithmetic; in fp64 the final
// velocities agree with the contact-space solve to ~1e-12 (tools/experiments/primal_numerics.py, the parity tests).
//
// Mapping: lane c holds the contact's rows of Jc / Jt in compressed form (six entries each + the two body indices) and the
// four inequality components of every m-space vector; lane r < nz + neq holds entry r of the x / y vectors and ROW r of the
// system matrix in registers.  Formation: the lanes add their 6 x 6 blocks into an LDS image of the matrix with ds_add_f64
// (one wave owns the image, so the order of the additions - and the result - is the same on every run), rows then move to
// registers.  LU without pivoting (x rows first: Q + G^T M^-1 G has a positive definite symmetric part; then the equality
// rows, whose Schur complement -A S^-1 A^T is negative definite), pivot rows broadcast with v_readlane.
// Kernels: forward of the fused step (engines.py:26-78) and its backward w.r.t. the physical inputs (lcp.py:37-64 contracted
// through the assembly), behind lcp_solve_dynamics_f32 / lcp_step_backward_f32; the same solve behind the dense LCPFunction
// boundary (DENSE: lcp_pdipm_forward_f32 / _backward_f32 at 17..64 contacts); post-stabilisation (engines.py:80-116,
// lcp_poststab_primal_kernel behind lcp_post_stabilization_f32).
#include "lcp_wave_scene.h"

namespace lcp {
namespace primal {

using namespace w64;
using namespace wsc;

constexpr int LX = 64;           // lanes = stride of the stored iterate
constexpr int EQB = 4;           // padded neq
// workspace per scene (doubles): a 64-entry header (contact count), then the best iterate the backward needs, in the layout
// lcp_big.hip uses, with room for 24 equality multipliers: x[64] y[24] z[4][64] s[4][64] mu[64] diag(Q)[64]
struct WsLayout { static constexpr int IT = 64, YCAP = 24, ZO = 64 + YCAP, TOTAL = IT + ZO + 10 * LX; };
constexpr int ZO = WsLayout::ZO;     // offset of z in the iterate block

#ifdef LCP_PRIMAL_PROFILE
#define PR_TICK(i) { const long long now_ = clock64(); pc[i] += now_ - tk; tk = now_; }
#else
#define PR_TICK(i)
#endif

// keeps a wave-uniform value in scalar registers at this point (the batches of pivot-row broadcasts stay batches)
__device__ __forceinline__ void sgpr_pin(double& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);     // ds_add_f64 (no return)
}

// NCOL: capacity of the system (nz + neq <= NCOL <= 64), a multiple of 8
// DENSE: the same solve behind the dense LCPFunction boundary (lcp.py:22-64): the scene's (Q, p, G, h, A, b, F) is read instead of
//        a contact list - scenes lcp_classify_big marked 3: the mixed contact LCP of engines.py:50-74 with a diagonal Q whose
//        Jacobian rows touch at most two bodies -, the outputs are x, y, z, s (forward) or the seven dense gradients of
//        lcp.py:52-61 (backward).
template <int NCOL, bool BWD, bool DENSE, int EQC>
__global__ void __launch_bounds__(64, (NCOL <= 40 ? 2 : 1)) lcp_primal_kernel(StepArgs SP, StepBwdArgs Gd, DenseIO DN) {
  constexpr int LDK = NCOL + 1;
  constexpr int KSZ = (NCOL * LDK > 11 * LX) ? NCOL * LDK : 11 * LX;    // (the dense backward stages 144 + 8 nc <= 656 doubles here)
  __shared__ __attribute__((aligned(16))) double Kl[KSZ];          // image of the system matrix (formation); backward: staging
  __shared__ double xv[LX];                                        // x-space exchange / accumulation
  constexpr int AST = (EQC <= 4) ? LX : NCOL;                        // row stride of the A image (many rows: packed to the system's width)
  __shared__ float At[EQC * AST];                                  // A rows
  __shared__ int B12[2 * LX];
  __shared__ double stash[8 * LX];                                 // the affine direction, parked during the corrector solve
  const int scene = blockIdx.x, lane = threadIdx.x;
  if (DENSE && DN.cls[scene] != 3) return;                                 // (another family serves the scene)
  const int nb = SP.nb, nz = 3 * nb, ncap = SP.nc, e = SP.e, n = nz + e;
  double* Wg = (double*)SP.ws + (size_t)scene * (DENSE ? DN.ws_scene / sizeof(double) : (size_t)WsLayout::TOTAL);
  double* Wit = Wg + WsLayout::IT;
  int ncs = ncap;
  if (BWD) ncs = (int)Wg[0];                                               // the count the forward solved with
  else if (SP.c_count) ncs = SP.c_count[scene];
  const int truncated = (ncs > ncap) ? LCP_ST_TRUNCATED : 0;               // more contacts found than the list holds
  ncs = ncs < 0 ? 0 : (ncs > ncap ? ncap : ncs);
  if (!BWD && lane == 0) Wg[0] = (double)ncs;
  const bool vc = lane < ncs;                                              // this lane owns a live contact
  const bool vx = lane < nz, ve = lane >= nz && lane < n;                  // ... an x entry, an equality multiplier

  // ---- assembly (engines.py:31-32,50-74; world.py:144-234) ----------------------------------------------------------
  const float* Md = DENSE ? nullptr : (const float*)SP.Mdiag + (size_t)scene * nz;
  const float* vv = DENSE ? nullptr : (const float*)SP.v + (size_t)scene * nz;
  const float* ff = DENSE ? nullptr : (const float*)SP.f + (size_t)scene * nz;
  float jn[6] = {0, 0, 0, 0, 0, 0}, jf[6] = {0, 0, 0, 0, 0, 0};            // (fp32 inputs: exact, half the registers)
  int c0 = 0, c1 = 0;                                                      // first columns of the contact's two bodies
  double mu_c = 0, hn = 0;
  double qd = 0, p = 0, b_in = 0;
  for (int i = lane; i < EQC * AST; i += 64) At[i] = 0.0f;
  wsync();
  if constexpr (DENSE) {
    // dense boundary: G = [Jc; Jf; 0] with Jf rows (+jt, -jt) (engines.py:67-68, world.py:191-192), F[3nc + c][c] = mu_c
    // (engines.py:71), h = [h_n; 0; 0] (:74), at most two bodies per contact - all verified per scene by lcp_classify_big
    const int m = DN.m;
    if (vc) {
      const float* gc = DN.G + ((size_t)scene * m + lane) * nz;
      const float* gt = DN.G + ((size_t)scene * m + ncap + 2 * lane) * nz;
      int bf = -1, bl = -1;                                                // first / last body with a nonzero entry
      for (int bq = 0; bq < nb; ++bq) {
        const bool nzb = (gc[3 * bq] != 0.0f) || (gc[3 * bq + 1] != 0.0f) || (gc[3 * bq + 2] != 0.0f) ||
                         (gt[3 * bq] != 0.0f) || (gt[3 * bq + 1] != 0.0f) || (gt[3 * bq + 2] != 0.0f);
        if (nzb) { if (bf < 0) bf = bq; bl = bq; }
      }
      if (bf < 0) { bf = 0; bl = 0; }
      c0 = 3 * bf; c1 = 3 * bl;
#pragma unroll
      for (int q = 0; q < 3; ++q) { jn[q] = gc[c0 + q]; jf[q] = gt[c0 + q]; }
      if (bl != bf) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { jn[3 + q] = gc[c1 + q]; jf[3 + q] = gt[c1 + q]; }
      }
      if (BWD) mu_c = Wit[ZO + 8 * LX + lane];                              // (lcp_pdipm_backward_f32 gets G, A and the cotangent only)
      else {
        mu_c = (double)DN.F[(size_t)scene * m * m + (size_t)(3 * ncap + lane) * m + lane];
        hn = (double)DN.h[(size_t)scene * m + lane];
      }
    }
    if (BWD) { if (vx) qd = Wit[ZO + 9 * LX + lane]; }
    else {
      if (vx) { qd = (double)DN.Q[(size_t)scene * nz * nz + (size_t)lane * nz + lane]; p = (double)DN.p[(size_t)scene * nz + lane]; }
      if (ve) b_in = (double)DN.b[(size_t)scene * e + (lane - nz)];
      Wit[ZO + 8 * LX + lane] = mu_c; Wit[ZO + 9 * LX + lane] = qd;
    }
    for (int i = lane; i < e * nz; i += 64) { const int a = i / nz, k = i - a * nz; At[a * AST + k] = DN.A[(size_t)scene * e * nz + i]; }
  } else {
    if (vc) {
      const ContactRows<float> r = make_contact<float>((const float*)SP.c_n + (size_t)scene * ncap * 2, (const float*)SP.c_p1 + (size_t)scene * ncap * 2,
                                                       (const float*)SP.c_p2 + (size_t)scene * ncap * 2, SP.c_i1 + (size_t)scene * ncap,
                                                       SP.c_i2 + (size_t)scene * ncap, (const float*)SP.rest + (size_t)scene * nb,
                                                       (const float*)SP.fric + (size_t)scene * nb, vv, lane);
#pragma unroll
      for (int q = 0; q < 6; ++q) { jn[q] = r.jn[q]; jf[q] = r.jf[q]; }
      c0 = 3 * r.b1; c1 = 3 * r.b2;
      mu_c = (double)r.mu; hn = (double)r.h;
    }
    if (vx) {
      qd = (double)Md[lane];
      p = (double)momentum_entry<float>(Md[lane], vv[lane], (float)SP.dt, ff[lane]);          // engines.py:32
    }
    for (int i = lane; i < e * nz; i += 64) { const int a = i / nz, k = i - a * nz; At[a * AST + k] = ((const float*)SP.Je)[(size_t)scene * e * nz + i]; }
  }
  auto colq = [&](int q) { return q < 3 ? c0 + q : c1 + (q - 3); };
  wsync();
  int status = truncated;
  if (__any(vx && !(qd != 0.0))) status |= LCP_ST_SINGULAR_Q;
  // the lane's column / row of A: x lanes hold A[:, lane], the equality lane nz + a holds nothing extra (its row is read from At)
  auto acol = [&](int a) -> double { return (EQC <= 4 || lane < NCOL) ? (double)At[a * AST + lane] : 0.0; };    // (zero beyond nz: At is cleared)

  // ---- products ------------------------------------------------------------------------------------------------------
  auto Gv = [&](double v, double& gn, double& gt) {                       // m-space <- x-space (v on the x lanes)
    xv[lane] = vx ? v : 0.0; wsync();
    gn = 0; gt = 0;
    if (vc) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { const double xq = xv[colq(q)]; gn = fma((double)jn[q], xq, gn); gt = fma((double)jf[q], xq, gt); }
    }
    wsync();
  };
  auto Gtw = [&](double wn, double wt) -> double {                        // x-space <- m-space (Jc^T wn + Jt^T wt)
    xv[lane] = 0.0; wsync();
    if (vc) {
#pragma unroll
      for (int q = 0; q < 6; ++q) lds_add(&xv[colq(q)], fma((double)jf[q], wt, (double)jn[q] * wn));
    }
    wsync();
    const double r = vx ? xv[lane] : 0.0;
    wsync();
    return r;
  };
  auto Av = [&](double v) -> double {                                     // equality lanes <- x lanes
    double out = 0;
    if constexpr (EQC <= 4) {
#pragma unroll
      for (int a = 0; a < EQC; ++a) { if (a < e) { const double sm = wave_sum(acol(a) * (vx ? v : 0.0)); if (lane == nz + a) out = sm; } }
    } else {                                                              // many rows (chains of joints): every equality lane sums its own row
      xv[lane] = vx ? v : 0.0; wsync();
      if (ve) { const float* ar = At + (lane - nz) * AST; for (int k = 0; k < nz; ++k) out = fma((double)ar[k], xv[k], out); }
      wsync();
    }
    return out;
  };
  auto Aty = [&](double y) -> double {                                    // x lanes <- equality lanes
    double acc = 0;
    if constexpr (EQC <= 4) {
#pragma unroll
      for (int a = 0; a < EQC; ++a) { if (a < e) acc = fma(acol(a), bcast_lane(y, nz + a), acc); }
    } else {
      for (int a = 0; a < e; ++a) acc = fma(acol(a), bcast_lane(y, nz + a), acc);
    }
    return acc;
  };

  // ---- the contact's 4 x 4 block M = F_c + diag(s / z), inverted in closed form ------------------------------------------------
  double idn = 1, i1 = 1, i2 = 1, kap = 1.0 / 3.0;                         // 1 / Dn, 1 / D1, 1 / D2, 1 / (Dg + 1 / D1 + 1 / D2)
  double b00 = 0, b10 = 0, b11 = 0;                                        // the 2 x 2 matrix of G^T M^-1 G in (Jc, Jt) coordinates
  auto block_setup = [&](const M4<double>& D) {                           // D = 1 / d = s / z
    idn = 1.0 / D.n; i1 = 1.0 / D.f1; i2 = 1.0 / D.f2;
    kap = 1.0 / (D.g + (i1 + i2));
    b00 = idn;
    b10 = kap * (i1 - i2) * (mu_c * idn);
    b11 = kap * fma(i1 + i2, D.g, 4.0 * (i1 * i2));                        // = (i1 + i2) - kap (i1 - i2)^2, without the cancellation
  };
  auto minv = [&](const M4<double>& t) -> M4<double> {                    // M^-1 t
    M4<double> o;
    o.n = idn * t.n;
    o.g = kap * ((t.g - mu_c * o.n) + fma(i1, t.f1, i2 * t.f2));
    o.f1 = i1 * (t.f1 - o.g);
    o.f2 = i2 * (t.f2 - o.g);
    return o;
  };

#ifdef LCP_PRIMAL_PROFILE
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = clock64();
#endif
  // ---- formation + LU of K = [[Q + G^T M^-1 G, A^T], [A, 0]]: row `lane` in t[], 1 / U[lane][lane] in udinv ---------------------
  double t[NCOL];
  double udinv = 1.0;
  bool singular = false;
  auto factor = [&]() LCP_INL {
    // (the lane compares below are invariant over the PDIPM loop: hoisted, their ~100 masks overflow the scalar file and come
    //  back through v_readlane spills - an opaque copy of the lane index keeps them local to the factorisation)
    int ln = lane; asm volatile("" : "+v"(ln));
    for (int i = lane; i < NCOL * LDK; i += 64) Kl[i] = 0.0;
    wsync();
    if (lane < NCOL) Kl[lane * LDK + lane] = vx ? qd : (ve ? 0.0 : 1.0);   // rows beyond the system: identity
    if (vx) {
      if constexpr (EQC <= 4) {
#pragma unroll
        for (int a = 0; a < EQC; ++a) { if (a < e) { const double av = acol(a); Kl[lane * LDK + nz + a] = av; Kl[(nz + a) * LDK + lane] = av; } }
      } else {
        for (int a = 0; a < e; ++a) { const double av = acol(a); Kl[lane * LDK + nz + a] = av; Kl[(nz + a) * LDK + lane] = av; }
      }
    }
    wsync();
    if (vc) {
      double p0[6], p1[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) { p0[q] = b00 * (double)jn[q]; p1[q] = fma(b10, (double)jn[q], b11 * (double)jf[q]); }
#pragma unroll
      for (int pq = 0; pq < 6; ++pq) {
        double* row = Kl + colq(pq) * LDK;
#pragma unroll
        for (int q = 0; q < 6; ++q) lds_add(row + colq(q), fma((double)jf[pq], p1[q], (double)jn[pq] * p0[q]));
      }
    }
    wsync();
    PR_TICK(1)
    {
      const double* row = Kl + (lane < NCOL ? lane : 0) * LDK;
      static_for<NCOL>([&](auto J) LCP_INL { t[J] = row[J]; });
      if (lane >= NCOL) static_for<NCOL>([&](auto J) LCP_INL { t[J] = 0.0; });
    }
    wsync();
    singular = false;
    static_for<NCOL / 8>([&](auto G8) LCP_INL {
      if (8 * G8 < n) {
        static_for<8>([&](auto KK) LCP_INL {
          constexpr int k = 8 * G8 + KK;
          const double pk = bcast_lane(t[k], k);
          singular = singular || !(pk != 0.0) || (pk != pk);
          const double inv = fast_rcp(pk);
          if (ln == k) udinv = inv;
          const double l = (ln > k) ? t[k] * inv : 0.0;
          if (ln > k) t[k] = l;
          // pivot-row entries in batches of 8 scalar pairs, then the 8 FMAs: back to back, every v_readlane -> v_fma pair
          // costs two wait states more and the pairs serialise on one scalar register.  (Measured and dropped: the pivot row
          // through LDS - lane k stores it with ds_write_b128, every lane reads it back at a uniform address.  Fewer
          // instructions, but every read moves 1 KB through the CU's one LDS port and eight waves share it: 3.8 M against
          // 6.5 M sim steps/s on config 5.  Also measured and dropped: skipping the 3 x 3 column blocks a pivot row cannot reach
          // - block masks from the contact graph, symbolic elimination on scalars, bodies in reverse order to keep the fill
          // low (31 of 66 block visits on a config-5 pile) - with scalar branches between batches of three columns: the
          // shorter batches and the branches cost more than the skipped work saves, 6.3 M against 6.5 M; towers gained 3-5 %.)
          constexpr int NJ = NCOL - 1 - k;
          static_for<(NJ + 7) / 8>([&](auto C8) LCP_INL {
            constexpr int j0 = k + 1 + 8 * C8, nj = (NJ - 8 * C8) < 8 ? (NJ - 8 * C8) : 8;
            double pv[8];
            static_for<nj>([&](auto I) LCP_INL { pv[I] = bcast_lane(t[j0 + I], k); });
            static_for<nj>([&](auto I) LCP_INL { sgpr_pin(pv[I]); });
            static_for<nj>([&](auto I) LCP_INL { t[j0 + I] = fma(-l, pv[I], t[j0 + I]); });
          });
        });
      }
    });
  };
  // K^-1 w (w: entry `lane` of the right-hand side)
  auto ksolve = [&](double w) -> double {
    int ln = lane; asm volatile("" : "+v"(ln));
    static_for<NCOL / 8>([&](auto G8) LCP_INL {
      if (8 * G8 < n) {
        static_for<8>([&](auto KK) LCP_INL {
          constexpr int k = 8 * G8 + KK;
          const double yk = bcast_lane(w, k);
          w = fma(-((ln > k) ? t[k] : 0.0), yk, w);
        });
      }
    });
    static_for<NCOL / 8>([&](auto GR) LCP_INL {
      constexpr int g8 = NCOL / 8 - 1 - GR;
      if (8 * g8 < n) {
        static_for<8>([&](auto KR) LCP_INL {
          constexpr int k = 8 * g8 + 7 - KR;
          const double xk = bcast_lane(w * udinv, k);
          w = fma(-((ln < k) ? t[k] : 0.0), xk, w);
        });
      }
    });
    return w * udinv;
  };

  // solve_kkt (pdipm.py:325-354) in body space; di = 1 / d.  rx / ox: x lanes, ry / oy: equality lanes
  auto solve_kkt = [&](const M4<double>& di, double rx, const M4<double>& rs, const M4<double>& rz, double ry,
                       double& ox, M4<double>& os, M4<double>& oz, double& oy) {
    M4<double> q = m4<double>(rs.n * di.n - rz.n, rs.f1 * di.f1 - rz.f1, rs.f2 * di.f2 - rz.f2, rs.g * di.g - rz.g);
    if (!vc) q = m4<double>(0, 0, 0, 0);
    const M4<double> u = minv(q);
    const double gu = Gtw(vc ? u.n : 0.0, vc ? u.f1 - u.f2 : 0.0);
    const double rhs = vx ? (gu - rx) : (ve ? -ry : 0.0);
    const double sol = ksolve(rhs);
    ox = vx ? sol : 0.0; oy = ve ? sol : 0.0;
    double gn, gt;
    Gv(ox, gn, gt);
    oz = minv(m4<double>(gn - q.n, gt - q.f1, -gt - q.f2, -q.g));
    if (!vc) oz = m4<double>(0, 0, 0, 0);
    os = m4<double>((-rs.n - oz.n) * di.n, (-rs.f1 - oz.f1) * di.f1, (-rs.f2 - oz.f2) * di.f2, (-rs.g - oz.g) * di.g);   // :347,350
    if (!vc) os = m4<double>(0, 0, 0, 0);
  };

  // get_step for (z, dz), (s, ds) (pdipm.py:182-186), NaN semantics as in lcp_quad.hip step_pair_q
  auto step_pair = [&](const M4<double>& z, const M4<double>& dz, const M4<double>& s, const M4<double>& ds) -> double {
    const double ninf = -inf_of<double>(), pinf = inf_of<double>();
    const M4<double> az = m4<double>(-z.n / dz.n, -z.f1 / dz.f1, -z.f2 / dz.f2, -z.g / dz.g);
    const M4<double> as = m4<double>(-s.n / ds.n, -s.f1 / ds.f1, -s.f2 / ds.f2, -s.g / ds.g);
    auto key4 = [&](const M4<double>& a) { return umax(umax(nan_key(a.n), nan_key(a.f1)), umax(nan_key(a.f2), nan_key(a.g))); };
    auto max4 = [&](const M4<double>& a) { return __builtin_fmax(__builtin_fmax(a.n, a.f1), __builtin_fmax(a.f2, a.g)); };
    auto min4 = [&](const M4<double>& a) { return __builtin_fmin(__builtin_fmin(a.n, a.f1), __builtin_fmin(a.f2, a.g)); };
    const uint32_t kmz = wave_umax(vc ? key4(az) : 0u), kms = wave_umax(vc ? key4(as) : 0u);
    const double mz = wave_max(vc ? max4(az) : ninf), ms = wave_max(vc ? max4(as) : ninf);
    const double fz = key_is_nan(kmz) ? 1.0 : __builtin_fmax(mz, 1.0), fs = key_is_nan(kms) ? 1.0 : __builtin_fmax(ms, 1.0);
    auto pick = [&](double dv, double a, double fill) { return (dv > 0.0) ? fill : a; };
    const M4<double> pz = m4<double>(pick(dz.n, az.n, fz), pick(dz.f1, az.f1, fz), pick(dz.f2, az.f2, fz), pick(dz.g, az.g, fz));
    const M4<double> ps = m4<double>(pick(ds.n, as.n, fs), pick(ds.f1, as.f1, fs), pick(ds.f2, as.f2, fs), pick(ds.g, as.g, fs));
    const uint32_t kl = wave_umax(vc ? umax(key4(pz), key4(ps)) : 0u);
    const double l = wave_min(vc ? __builtin_fmin(min4(pz), min4(ps)) : pinf);
    return key_is_nan(kl) ? nan_of<double>() : l;
  };

  if (BWD) {
    // ---- backward: d(loss)/d(v_new) -> d(loss)/d(Mdiag, v, f, rest, fric, contact normal / arms) ----------------------------
    double x = vx ? Wit[lane] : 0.0, dx = 0, dnu = 0;
    M4<double> z = m4<double>(1, 1, 1, 1), s = z, dinv = z, ds, dl;
    if (vc) {
      z = m4<double>(Wit[ZO + lane], Wit[ZO + LX + lane], Wit[ZO + 2 * LX + lane], Wit[ZO + 3 * LX + lane]);
      s = m4<double>(Wit[ZO + 4 * LX + lane], Wit[ZO + 5 * LX + lane], Wit[ZO + 6 * LX + lane], Wit[ZO + 7 * LX + lane]);
      dinv = m4<double>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g);                 // 1 / d, d = z / s (lcp.py:44)
    }
    // At a converged iterate the ratios D = s / z of the active rows underflow against Q (1e-12 and below), and Q + G^T M^-1 G
    // would lose Q.  The factorisation therefore uses D floored at BWD_FLOOR x (the row's effective inverse mass
    // j Q^-1 j^T) - a perturbation of 1e-9 of the diagonal of the contact-space matrix - and one step of iterative refinement on
    // the UNREDUCED equations (residuals formed with M, not M^-1) takes the perturbation out again: 1e-8 of the natural scale
    // |g| / min Q against the contact-space solve (tools/experiments/primal_numerics.py).  The forward needs neither: its
    // right-hand sides keep the error of the stiff directions benign (same experiment).
    constexpr double BWD_FLOOR = 1e-9;
    M4<double> dfl = dinv;
    {
      xv[lane] = vx ? 1.0 / qd : 0.0; wsync();
      double wn = 0, wt = 0;
      if (vc) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { const double qi = xv[colq(q)]; wn = fma((double)jn[q] * (double)jn[q], qi, wn); wt = fma((double)jf[q] * (double)jf[q], qi, wt); }
        dfl.n = __builtin_fmax(dinv.n, BWD_FLOOR * wn);
        dfl.f1 = __builtin_fmax(dinv.f1, BWD_FLOOR * wt);
        dfl.f2 = __builtin_fmax(dinv.f2, BWD_FLOOR * wt);
      }
      wsync();
    }
    block_setup(dfl);
    factor();                                                               // lcp.py:46
    // v_new = -x (engines.py:76-77)  =>  d(loss)/dx = -d(loss)/d(v_new)
    const double g = !vx ? 0.0 : (DENSE ? (double)DN.dl_dx[(size_t)scene * nz + lane] : -(double)((const float*)Gd.dl_dv)[(size_t)scene * nz + lane]);
    const M4<double> zero = m4<double>(0, 0, 0, 0);
    solve_kkt(dfl, g, zero, zero, 0.0, dx, ds, dl, dnu);                     // lcp.py:47-50
    {
      // residuals of  Q dx + G^T dl + A^T dnu = -g ,  G dx - M dl = 0 ,  A dx = 0  with the TRUE D
      double r1 = -g - (qd * dx + Gtw(vc ? dl.n : 0.0, vc ? dl.f1 - dl.f2 : 0.0));
      if (e > 0) r1 -= Aty(dnu);
      if (!vx) r1 = 0.0;
      double gn, gt;
      Gv(dx, gn, gt);
      M4<double> r3 = m4<double>(-(gn - dinv.n * dl.n), -(gt - (dinv.f1 * dl.f1 + dl.g)), -(-gt - (dinv.f2 * dl.f2 + dl.g)),
                                 (mu_c * dl.n - (dl.f1 + dl.f2)) + dinv.g * dl.g);
      if (!vc) r3 = zero;
      const double r2 = (e > 0) ? -Av(dx) : 0.0;
      double cx, cnu;
      M4<double> cs, cl;
      solve_kkt(dfl, -r1, zero, m4<double>(-r3.n, -r3.f1, -r3.f2, -r3.g), -r2, cx, cs, cl, cnu);
      dx += cx; dnu += cnu;
      dl = m4<double>(dl.n + cl.n, dl.f1 + cl.f1, dl.f2 + cl.f2, dl.g + cl.g);
    }
    if constexpr (DENSE) {
      // ---- LCPFunction.backward (lcp.py:52-61): the outer products of (x, dx), (nu, dnu), (lam, dlam) --------------------------------
      double *X = Kl, *DX = Kl + 64, *NU = Kl + 128, *DNU = Kl + 136, *LAM = Kl + 144, *DLAM = Kl + 144 + 4 * LX;
      const int m = 4 * ncap;
      wsync();
      X[lane] = x; DX[lane] = vx ? dx : 0.0;
      if (lane < 8) { NU[lane] = (lane < e) ? Wit[64 + lane] : 0.0; DNU[lane] = 0.0; }
      wsync();
      if (ve) DNU[lane - nz] = dnu;
      if (lane < ncap) {                                                    // dense row order: [normal | friction pairs | cone]
        LAM[lane] = z.n; LAM[ncap + 2 * lane] = z.f1; LAM[ncap + 2 * lane + 1] = z.f2; LAM[3 * ncap + lane] = z.g;
        DLAM[lane] = dl.n; DLAM[ncap + 2 * lane] = dl.f1; DLAM[ncap + 2 * lane + 1] = dl.f2; DLAM[3 * ncap + lane] = dl.g;
      }
      wsync();
      if (DN.dp) for (int j = lane; j < nz; j += 64) DN.dp[(size_t)scene * nz + j] = (float)DX[j];                            // lcp.py:52
      if (DN.dh) for (int i = lane; i < m; i += 64) DN.dh[(size_t)scene * m + i] = (float)(-DLAM[i]);                          // :56
      if (DN.db) for (int a = lane; a < e; a += 64) DN.db[(size_t)scene * e + a] = (float)(-DNU[a]);                           // :58
      if (DN.dQ) for (int i = lane; i < nz * nz; i += 64) { const int j = i / nz, k = i - j * nz;
        DN.dQ[(size_t)scene * nz * nz + i] = (float)(0.5 * (DX[j] * X[k] + X[j] * DX[k])); }                                  // :59-60
      if (DN.dA) for (int i = lane; i < e * nz; i += 64) { const int a = i / nz, k = i - a * nz;
        DN.dA[(size_t)scene * e * nz + i] = (float)(DNU[a] * X[k] + NU[a] * DX[k]); }                                         // :57
      if (DN.dG) for (int i = lane; i < m * nz; i += 64) { const int r = i / nz, k = i - r * nz;
        DN.dG[(size_t)scene * m * nz + i] = (float)(DLAM[r] * X[k] + LAM[r] * DX[k]); }                                       // :53
      if (DN.dF) { float* o = DN.dF + (size_t)scene * m * m;
        for (int i = lane; i < m * m; i += 64) { const int r = i / m, c = i - r * m; o[i] = (float)(-DLAM[r] * LAM[c]); } }   // :54
      return;
    }
    // x-space vectors to LDS so that a contact lane can read the entries of its two bodies
    double* X = Kl; double* DX = Kl + LX; double* CR = Kl + 2 * LX; double* CF = Kl + 3 * LX;
    X[lane] = x; DX[lane] = dx; wsync();
    double gh_rbar = 0;
    {
      double cr = 0, cf = 0, dnx = 0, dny = 0, d1x = 0, d1y = 0, d2x = 0, d2y = 0;
      int b1 = 0, b2 = 0;
      if (vc) {
        const size_t cb = (size_t)scene * ncap + lane;
        const double nx = ((const float*)SP.c_n)[cb * 2], ny = ((const float*)SP.c_n)[cb * 2 + 1];
        const double p1x = ((const float*)SP.c_p1)[cb * 2], p1y = ((const float*)SP.c_p1)[cb * 2 + 1];
        const double p2x = ((const float*)SP.c_p2)[cb * 2], p2y = ((const float*)SP.c_p2)[cb * 2 + 1];
        b1 = SP.c_i1[cb]; b2 = SP.c_i2[cb];
        const double rbar = 0.5 * ((double)((const float*)SP.rest)[(size_t)scene * nb + b1] + (double)((const float*)SP.rest)[(size_t)scene * nb + b2]);
        const double jnd[6] = {p1x * ny - p1y * nx, nx, ny, -(p2x * ny - p2y * nx), -nx, -ny};     // world.py:177-183
        const double gh = -dl.n;                                              // dh = -dlam (lcp.py:56)
        const double af = dl.f1 - dl.f2, lf = z.f1 - z.f2;                    // Jf rows are +jt, -jt (world.py:191-192)
        double gjn[6], gjf[6], jnv = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int col = (q < 3) ? 3 * b1 + q : 3 * b2 + (q - 3);
          const double xq = X[col], dxq = DX[col], vq = (double)vv[col];
          jnv = fma(jnd[q], vq, jnv);
          gjn[q] = dl.n * xq + z.n * dxq + gh * rbar * vq;                    // dG row n (lcp.py:53) + h = (Jc v) rbar
          gjf[q] = af * xq + lf * dxq;
        }
        gh_rbar = gh * rbar;
        cr = 0.5 * gh * jnv;                                                  // rbar = (rest_b1 + rest_b2) / 2 (world.py:144-151)
        cf = 0.5 * (-dl.g * z.n);                                             // dF[gamma_c, n_c] = -dlam_g lam_n (lcp.py:54), F = mu there
        dnx = -gjn[0] * p1y + gjn[1] + gjn[3] * p2y - gjn[4] - gjf[0] * p1x - gjf[2] + gjf[3] * p2x + gjf[5];
        dny = gjn[0] * p1x + gjn[2] - gjn[3] * p2x - gjn[5] - gjf[0] * p1y + gjf[1] + gjf[3] * p2y - gjf[4];
        d1x = gjn[0] * ny - gjf[0] * nx; d1y = -gjn[0] * nx - gjf[0] * ny;
        d2x = -gjn[3] * ny + gjf[3] * nx; d2y = gjn[3] * nx + gjf[3] * ny;
      }
      wsync();
      CR[lane] = cr; CF[lane] = cf; B12[lane] = b1; B12[LX + lane] = b2;
      if (lane < ncap) {
        const size_t cb = (size_t)scene * ncap + lane;
        if (Gd.dcn) { ((float*)Gd.dcn)[cb * 2] = (float)dnx; ((float*)Gd.dcn)[cb * 2 + 1] = (float)dny; }
        if (Gd.dcp1) { ((float*)Gd.dcp1)[cb * 2] = (float)d1x; ((float*)Gd.dcp1)[cb * 2 + 1] = (float)d1y; }
        if (Gd.dcp2) { ((float*)Gd.dcp2)[cb * 2] = (float)d2x; ((float*)Gd.dcp2)[cb * 2 + 1] = (float)d2y; }
      }
    }
    const double dv_h = Gtw(gh_rbar, 0.0);                                   // Jc^T (dh rbar)
    if (vx) {
      const size_t o = (size_t)scene * nz + lane;
      const double md = (double)Md[lane], v = (double)vv[lane];
      if (Gd.dMdiag) ((float*)Gd.dMdiag)[o] = (float)(dx * x + dx * v);      // Q = diag(M) (dQ, lcp.py:59-60) and p = M v + dt f
      if (Gd.dv) ((float*)Gd.dv)[o] = (float)(dx * md + dv_h);
      if (Gd.df) ((float*)Gd.df)[o] = (float)(dx * (double)SP.dt);
    }
    if (Gd.dJe && e > 0) {                                                    // dA = dnu (x) x + nu (x) dx (lcp.py:57; A = Je)
      float* o = (float*)Gd.dJe + (size_t)scene * e * nz;
      for (int a = 0; a < e; ++a) {
        const double dn = bcast_lane(dnu, nz + a), nu = Wit[64 + a];
        if (vx) o[a * nz + lane] = (float)(dn * x + nu * dx);
      }
    }
    if (lane < nb) {                                                          // per-body sums over the contacts, fixed order
      double ar = 0, af = 0;
      for (int c = 0; c < ncs; ++c) {
        const double w = ((B12[c] == lane) ? 1.0 : 0.0) + ((B12[LX + c] == lane) ? 1.0 : 0.0);
        if (w != 0.0) { ar += w * CR[c]; af += w * CF[c]; }
      }
      if (Gd.drest) ((float*)Gd.drest)[(size_t)scene * nb + lane] = (float)ar;
      if (Gd.dfric) ((float*)Gd.dfric)[(size_t)scene * nb + lane] = (float)af;
    }
    return;
  }

  // ---- the PDIPM loop (pdipm.py:49-179) -----------------------------------------------------------------------------------
  const int max_iter = SP.max_iter, lim = SP.lim;
  const double eps = SP.eps;
  const double mf = (double)(4 * ncs);
  double x = 0, y = 0;                                                       // x on the x lanes, y on the equality lanes
  M4<double> s = m4<double>(1, 1, 1, 1), z = s, dinv = s;
  // the best iterate lives in the workspace block the backward reads (18 doubles per lane that would otherwise hold
  // registers for the whole loop): stored when an iterate improves, read back once for the outputs
  auto keep_best = [&](double x_, double y_, const M4<double>& z_, const M4<double>& s_) {
    if (vx) Wit[lane] = x_;
    if (ve) Wit[64 + (lane - nz)] = y_;
    Wit[ZO + lane] = z_.n; Wit[ZO + LX + lane] = z_.f1; Wit[ZO + 2 * LX + lane] = z_.f2; Wit[ZO + 3 * LX + lane] = z_.g;
    Wit[ZO + 4 * LX + lane] = s_.n; Wit[ZO + 5 * LX + lane] = s_.f1; Wit[ZO + 6 * LX + lane] = s_.f2; Wit[ZO + 7 * LX + lane] = s_.g;
  };
  double best_resid = inf_of<double>();
  bool have_best = false, done = false;
  int n_not = 0, iters = 0;
  for (int it = -1; it < max_iter; ++it) {
    double rx = 0, ry = 0, mu = 0, resid = 0;
    M4<double> rs = m4<double>(0, 0, 0, 0), rz = rs;
    if (it < 0) {                                                           // init: (p, 0, -h, -b), d = 1 (:57-63); b = 0 from a contact list (engines.py:74)
      rx = p; ry = -b_in; rz = m4<double>(-hn, 0, 0, 0); dinv = m4<double>(1, 1, 1, 1);
    } else {                                                                // residuals (:82-96)
      rx = Gtw(vc ? z.n : 0.0, vc ? z.f1 - z.f2 : 0.0) + qd * x + p;
      if (e > 0) rx += Aty(y);
      if (!vx) rx = 0.0;
      rs = z;
      double gn, gt;
      Gv(x, gn, gt);
      rz = m4<double>(gn + s.n - hn, gt + s.f1 - z.g, -gt + s.f2 - z.g, s.g - (mu_c * z.n - (z.f1 + z.f2)));
      if (!vc) rz = m4<double>(0, 0, 0, 0);
      ry = (e > 0) ? Av(x) - b_in : 0.0;
      const double n_rx = wave_sum(rx * rx);
      const double n_rz = wave_sum(rz.n * rz.n + rz.f1 * rz.f1 + rz.f2 * rz.f2 + rz.g * rz.g);
      const double n_ry = wave_sum(ry * ry);
      const double sz = wave_sum(vc ? (s.n * z.n + s.f1 * z.f1) + (s.f2 * z.f2 + s.g * z.g) : 0.0);
      mu = sz / mf; mu = mu < 0 ? -mu : mu;                                 // (:91)
      resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + mf * mu;               // (:92-96)
      dinv = vc ? m4<double>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g) : m4<double>(1, 1, 1, 1);   // 1 / d, d = z / s (:98)
    }
    block_setup(dinv);
    PR_TICK(0)                                                              // residuals
    factor();                                                               // (:99-100)
    PR_TICK(2)                                                              // LU
    if (it >= 0 && !done) {
      ++iters;
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; done = true; }   // except: return best (:99-102)
      else {
        const bool improved = !have_best || (resid < best_resid);             // (:107-132)
        if (improved) { best_resid = resid; n_not = 0; have_best = true; keep_best(x, y, z, s); }
        else ++n_not;
        if (n_not == lim || best_resid < eps || mu > mu_limit<double>()) done = true;   // (:133)
      }
    }
    // (the iterate the last pass would produce is never evaluated - pdipm.py:176-179 - so its solves are skipped)
    if (it >= 0 && it == max_iter - 1) done = true;
    if (done) break;
    double ax = 0, ay = 0;
    M4<double> as_ = m4<double>(0, 0, 0, 0), az = as_;
    const int npass = (it < 0) ? 1 : 2;
    for (int pass = 0; pass < npass; ++pass) {
      double ox, oy;
      M4<double> os, oz;
      PR_TICK(3)
      solve_kkt(dinv, rx, rs, rz, ry, ox, os, oz, oy);
      PR_TICK(4)                                                            // solve_kkt
      if (it < 0) {
        x = ox; s = os; z = oz; y = oy;                                     // (:60-63)
        auto min4 = [&](const M4<double>& a) { return pmin(pmin(a.n, a.f1), pmin(a.f2, a.g)); };
        const uint32_t ks = wave_umax(vc ? umax(umax(nan_key(s.n), nan_key(s.f1)), umax(nan_key(s.f2), nan_key(s.g))) : 0u);
        const uint32_t kz = wave_umax(vc ? umax(umax(nan_key(z.n), nan_key(z.f1)), umax(nan_key(z.f2), nan_key(z.g))) : 0u);
        double smin = wave_min(vc ? min4(s) : inf_of<double>()), zmin = wave_min(vc ? min4(z) : inf_of<double>());
        if (key_is_nan(ks)) smin = nan_of<double>();
        if (key_is_nan(kz)) zmin = nan_of<double>();
        if (smin <= 0.0) { const double sh = 1.0 - smin; s = m4<double>(s.n + sh, s.f1 + sh, s.f2 + sh, s.g + sh); }   // (:66-75)
        if (zmin <= 0.0) { const double sh = 1.0 - zmin; z = m4<double>(z.n + sh, z.f1 + sh, z.f2 + sh, z.g + sh); }
        if (!vc) { s = m4<double>(1, 1, 1, 1); z = s; }
        if (ncs == 0) { keep_best(x, y, z, s); done = true; }               // engines.py:36-50: x = P^-1 u, no LCP
      } else if (pass == 0) {
        ax = ox; ay = oy; as_ = os; az = oz;                                // affine direction (:138-139)
        const double alpha = pmin(step_pair(z, az, s, as_), 1.0);          // (:142-144)
        auto sc = [&](double sv, double dsv, double zv, double dzv) { return (sv + alpha * dsv) * (zv + alpha * dzv); };
        const double t3 = wave_sum(vc ? (sc(s.n, as_.n, z.n, az.n) + sc(s.f1, as_.f1, z.f1, az.f1)) + (sc(s.f2, as_.f2, z.f2, az.f2) + sc(s.g, as_.g, z.g, az.g)) : 0.0);
        const double t4 = wave_sum(vc ? (s.n * z.n + s.f1 * z.f1) + (s.f2 * z.f2 + s.g * z.g) : 0.0);
        const double r3 = t3 / t4, sig = r3 * r3 * r3;                      // (:146-150)
        const double ms = -mu * sig;
        rx = 0; ry = 0; rz = m4<double>(0, 0, 0, 0);
        rs = vc ? m4<double>((ms + as_.n * az.n) / s.n, (ms + as_.f1 * az.f1) / s.f1, (ms + as_.f2 * az.f2) / s.f2, (ms + as_.g * az.g) / s.g)
                : m4<double>(0, 0, 0, 0);                                   // (:153)
        stash[lane] = as_.n; stash[LX + lane] = as_.f1; stash[2 * LX + lane] = as_.f2; stash[3 * LX + lane] = as_.g;
        stash[4 * LX + lane] = az.n; stash[5 * LX + lane] = az.f1; stash[6 * LX + lane] = az.f2; stash[7 * LX + lane] = az.g;
      } else {
        const double cx = ox + ax, cy = oy + ay;                            // (:160-163)
        as_ = m4<double>(stash[lane], stash[LX + lane], stash[2 * LX + lane], stash[3 * LX + lane]);
        az = m4<double>(stash[4 * LX + lane], stash[5 * LX + lane], stash[6 * LX + lane], stash[7 * LX + lane]);
        const M4<double> cs = m4<double>(os.n + as_.n, os.f1 + as_.f1, os.f2 + as_.f2, os.g + as_.g);
        const M4<double> cz = m4<double>(oz.n + az.n, oz.f1 + az.f1, oz.f2 + az.f2, oz.g + az.g);
        const double alpha = pmin(0.999 * step_pair(z, cz, s, cs), 1.0);   // (:164-166)
        x += alpha * cx; y += alpha * cy;                                   // (:171-174)
        if (vc) {
          s = m4<double>(s.n + alpha * cs.n, s.f1 + alpha * cs.f1, s.f2 + alpha * cs.f2, s.g + alpha * cs.g);
          z = m4<double>(z.n + alpha * cz.n, z.f1 + alpha * cz.f1, z.f2 + alpha * cz.f2, z.g + alpha * cz.g);
        }
      }
      PR_TICK(5)                                                            // step lengths, update
    }
    if (done) break;
  }

  // ---- outputs (row layout of a capacity-sized LCP, padded slots 0) ---------------------------------------------------------
  if (!have_best && ncs > 0) keep_best(x, y, z, s);                         // (max_iter = 0: the initial point)
  const double bx = vx ? Wit[lane] : 0.0, by = ve ? Wit[64 + (lane - nz)] : 0.0;
  const M4<double> bz = m4<double>(Wit[ZO + lane], Wit[ZO + LX + lane], Wit[ZO + 2 * LX + lane], Wit[ZO + 3 * LX + lane]);
  const M4<double> bs = m4<double>(Wit[ZO + 4 * LX + lane], Wit[ZO + 5 * LX + lane], Wit[ZO + 6 * LX + lane], Wit[ZO + 7 * LX + lane]);
  bool bad = vx && (bx != bx);
  if (vc) bad = bad || (bz.n != bz.n) || (bs.n != bs.n) || (bz.f1 != bz.f1) || (bz.f2 != bz.f2) || (bz.g != bz.g) ||
                (bs.f1 != bs.f1) || (bs.f2 != bs.f2) || (bs.g != bs.g);
  if (__any(bad)) status |= LCP_ST_NAN;
  const int m = 4 * ncap;
  if (lane < ncap) {
    const float k = vc ? 1.0f : 0.0f;
    if (SP.z) { float* o = (float*)SP.z + (size_t)scene * m; o[lane] = k * (float)bz.n; o[ncap + 2 * lane] = k * (float)bz.f1; o[ncap + 2 * lane + 1] = k * (float)bz.f2; o[3 * ncap + lane] = k * (float)bz.g; }
    if (SP.s) { float* o = (float*)SP.s + (size_t)scene * m; o[lane] = k * (float)bs.n; o[ncap + 2 * lane] = k * (float)bs.f1; o[ncap + 2 * lane + 1] = k * (float)bs.f2; o[3 * ncap + lane] = k * (float)bs.g; }
  }
  if (ve && SP.y) ((float*)SP.y)[(size_t)scene * e + (lane - nz)] = (float)by;
  if (vx) {
    const double nv = DENSE ? bx : -bx;                                               // engines.py:76-77 (dense boundary: zhats = x itself, lcp.py:35)
    ((float*)SP.v_new)[(size_t)scene * nz + lane] = (float)nv;
    if (!DENSE && SP.p_new) ((float*)SP.p_new)[(size_t)scene * nz + lane] = (float)((double)((const float*)SP.pos)[(size_t)scene * nz + lane] + nv * SP.dt);   // bodies.py:81
  }
  if (lane == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
#ifdef LCP_PRIMAL_PROFILE
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && SP.s) { float* o = (float*)SP.s + (size_t)scene * 4 * ncap + (4 * ncap - 8); for (int i = 0; i < 6; ++i) o[i] = (float)pc[i]; }
#endif
}


// ---------------------------------------------------------------- post-stabilisation (engines.py:80-116; world.py:109-117)
// The frictionless LCP of PdipmEngine.post_stabilization - Q = M, p = 0, G = Jc, h = gc = Jc v + Jc v * -restitutions, A = Je,
// b = ge = Je v, F = 0: ONE inequality row per contact - solved in body space like the step above (the block M is the scalar
// D = s / z), then dp = -x and, when poses are given, the correction move p_out = p + (dp / 2) dt_scene.  No contact: the direct
// KKT solve of :92-103, which is what the initialisation solve computes.  One wave per scene; replaces the generic
// workgroup-per-scene kernel on this path (4.2 ms for 4096 x 16 contacts).
template <int NCOL, bool BWD, int EQC>
__global__ void __launch_bounds__(64, (NCOL <= 40 ? 2 : 1)) lcp_poststab_primal_kernel(StepArgs SP, StepBwdArgs Gd) {
  constexpr int LDK = NCOL + 1;
  __shared__ __attribute__((aligned(16))) double Kl[NCOL * LDK];
  __shared__ double xv[LX];
  constexpr int AST = (EQC <= 4) ? LX : NCOL;
  __shared__ float At[EQC * AST];
  __shared__ int B12[2 * LX];
  const int scene = blockIdx.x, lane = threadIdx.x;
  const int nb = SP.nb, nz = 3 * nb, ncap = SP.nc, e = SP.e, n = nz + e;
  // workspace per scene (when given): the count and the best iterate, for the backward: [ncs .. | x[64] y[16] z[64] s[64]]
  double* Wg = SP.ws ? (double*)SP.ws + (size_t)scene * (size_t)WsLayout::TOTAL : nullptr;
  double* Wit = Wg ? Wg + WsLayout::IT : nullptr;
  int ncs = ncap;
  if (BWD) ncs = (int)Wg[0];
  else if (SP.c_count) ncs = SP.c_count[scene];
  const int truncated = (ncs > ncap) ? LCP_ST_TRUNCATED : 0;
  ncs = ncs < 0 ? 0 : (ncs > ncap ? ncap : ncs);
  const bool vc = lane < ncs, vx = lane < nz, ve = lane >= nz && lane < n;
  const float* Md = (const float*)SP.Mdiag + (size_t)scene * nz;
  const float* vv = (const float*)SP.v + (size_t)scene * nz;
  float jn[6] = {0, 0, 0, 0, 0, 0};
  int c0 = 0, c1 = 0;
  double hn = 0;
  if (vc) {
    const ContactRows<float> r = make_contact<float>((const float*)SP.c_n + (size_t)scene * ncap * 2, (const float*)SP.c_p1 + (size_t)scene * ncap * 2,
                                                     (const float*)SP.c_p2 + (size_t)scene * ncap * 2, SP.c_i1 + (size_t)scene * ncap,
                                                     SP.c_i2 + (size_t)scene * ncap, (const float*)SP.rest + (size_t)scene * nb,
                                                     (const float*)SP.rest + (size_t)scene * nb, vv, lane);
#pragma unroll
    for (int q = 0; q < 6; ++q) jn[q] = r.jn[q];
    c0 = 3 * r.b1; c1 = 3 * r.b2;
    hn = (double)r.jv + (double)r.jv * -(double)r.rbar;                      // engines.py:87-89
  }
  auto colq = [&](int q) { return q < 3 ? c0 + q : c1 + (q - 3); };
  const double qd = vx ? (double)Md[lane] : 0.0;
  for (int i = lane; i < EQC * AST; i += 64) At[i] = 0.0f;
  wsync();
  for (int i = lane; i < e * nz; i += 64) { const int a = i / nz, k = i - a * nz; At[a * AST + k] = ((const float*)SP.Je)[(size_t)scene * e * nz + i]; }
  wsync();
  int status = truncated;
  if (__any(vx && !(qd != 0.0))) status |= LCP_ST_SINGULAR_Q;
  auto acol = [&](int a) -> double { return (EQC <= 4 || lane < NCOL) ? (double)At[a * AST + lane] : 0.0; };
  auto Gv = [&](double v) -> double {                                     // (Jc v)_c
    xv[lane] = vx ? v : 0.0; wsync();
    double gn = 0;
    if (vc) {
#pragma unroll
      for (int q = 0; q < 6; ++q) gn = fma((double)jn[q], xv[colq(q)], gn);
    }
    wsync();
    return gn;
  };
  auto Gtw = [&](double wn) -> double {                                   // (Jc^T w)_j
    xv[lane] = 0.0; wsync();
    if (vc) {
#pragma unroll
      for (int q = 0; q < 6; ++q) lds_add(&xv[colq(q)], (double)jn[q] * wn);
    }
    wsync();
    const double r = vx ? xv[lane] : 0.0;
    wsync();
    return r;
  };
  auto Av = [&](double v) -> double {
    double out = 0;
    if constexpr (EQC <= 4) {
#pragma unroll
      for (int a = 0; a < EQC; ++a) { if (a < e) { const double sm = wave_sum(acol(a) * (vx ? v : 0.0)); if (lane == nz + a) out = sm; } }
    } else {
      xv[lane] = vx ? v : 0.0; wsync();
      if (ve) { const float* ar = At + (lane - nz) * AST; for (int k = 0; k < nz; ++k) out = fma((double)ar[k], xv[k], out); }
      wsync();
    }
    return out;
  };
  auto Aty = [&](double y) -> double {
    double acc = 0;
    if constexpr (EQC <= 4) {
#pragma unroll
      for (int a = 0; a < EQC; ++a) { if (a < e) acc = fma(acol(a), bcast_lane(y, nz + a), acc); }
    } else {
      for (int a = 0; a < e; ++a) acc = fma(acol(a), bcast_lane(y, nz + a), acc);
    }
    return acc;
  };
  const double b_in = (e > 0) ? Av(vx ? (double)vv[lane] : 0.0) : 0.0;    // ge = Je v (engines.py:86), on the equality lanes

  double t[NCOL];
  double udinv = 1.0, idn = 1.0;
  bool singular = false;
  auto factor = [&]() LCP_INL {
    int ln = lane; asm volatile("" : "+v"(ln));
    for (int i = lane; i < NCOL * LDK; i += 64) Kl[i] = 0.0;
    wsync();
    if (lane < NCOL) Kl[lane * LDK + lane] = vx ? qd : (ve ? 0.0 : 1.0);
    if (vx) {
      if constexpr (EQC <= 4) {
#pragma unroll
        for (int a = 0; a < EQC; ++a) { if (a < e) { const double av = acol(a); Kl[lane * LDK + nz + a] = av; Kl[(nz + a) * LDK + lane] = av; } }
      } else {
        for (int a = 0; a < e; ++a) { const double av = acol(a); Kl[lane * LDK + nz + a] = av; Kl[(nz + a) * LDK + lane] = av; }
      }
    }
    wsync();
    if (vc) {
#pragma unroll
      for (int pq = 0; pq < 6; ++pq) {
        double* row = Kl + colq(pq) * LDK;
        const double a = idn * (double)jn[pq];
#pragma unroll
        for (int q = 0; q < 6; ++q) lds_add(row + colq(q), a * (double)jn[q]);
      }
    }
    wsync();
    {
      const double* row = Kl + (lane < NCOL ? lane : 0) * LDK;
      static_for<NCOL>([&](auto J) LCP_INL { t[J] = row[J]; });
      if (lane >= NCOL) static_for<NCOL>([&](auto J) LCP_INL { t[J] = 0.0; });
    }
    wsync();
    singular = false;
    static_for<NCOL / 8>([&](auto G8) LCP_INL {
      if (8 * G8 < n) {
        static_for<8>([&](auto KK) LCP_INL {
          constexpr int k = 8 * G8 + KK;
          const double pk = bcast_lane(t[k], k);
          singular = singular || !(pk != 0.0) || (pk != pk);
          const double inv = fast_rcp(pk);
          if (ln == k) udinv = inv;
          const double l = (ln > k) ? t[k] * inv : 0.0;
          if (ln > k) t[k] = l;
          constexpr int NJ = NCOL - 1 - k;
          static_for<(NJ + 7) / 8>([&](auto C8) LCP_INL {
            constexpr int j0 = k + 1 + 8 * C8, nj = (NJ - 8 * C8) < 8 ? (NJ - 8 * C8) : 8;
            double pv[8];
            static_for<nj>([&](auto I) LCP_INL { pv[I] = bcast_lane(t[j0 + I], k); });
            static_for<nj>([&](auto I) LCP_INL { sgpr_pin(pv[I]); });
            static_for<nj>([&](auto I) LCP_INL { t[j0 + I] = fma(-l, pv[I], t[j0 + I]); });
          });
        });
      }
    });
  };
  auto ksolve = [&](double w) -> double {
    int ln = lane; asm volatile("" : "+v"(ln));
    static_for<NCOL / 8>([&](auto G8) LCP_INL {
      if (8 * G8 < n) {
        static_for<8>([&](auto KK) LCP_INL {
          constexpr int k = 8 * G8 + KK;
          const double yk = bcast_lane(w, k);
          w = fma(-((ln > k) ? t[k] : 0.0), yk, w);
        });
      }
    });
    static_for<NCOL / 8>([&](auto GR) LCP_INL {
      constexpr int g8 = NCOL / 8 - 1 - GR;
      if (8 * g8 < n) {
        static_for<8>([&](auto KR) LCP_INL {
          constexpr int k = 8 * g8 + 7 - KR;
          const double xk = bcast_lane(w * udinv, k);
          w = fma(-((ln < k) ? t[k] : 0.0), xk, w);
        });
      }
    });
    return w * udinv;
  };
  // solve_kkt (pdipm.py:325-354) in body space: q = rs / d - rz, K [dx; dy] = [-rx + Jc^T (q / D); -ry], dz = (Jc dx - q) / D
  auto solve_kkt = [&](double di, double rx, double rs, double rz, double ry, double& ox, double& os, double& oz, double& oy) {
    const double q = vc ? rs * di - rz : 0.0;
    const double gu = Gtw(vc ? idn * q : 0.0);
    const double sol = ksolve(vx ? (gu - rx) : (ve ? -ry : 0.0));
    ox = vx ? sol : 0.0; oy = ve ? sol : 0.0;
    const double gx = Gv(ox);
    oz = vc ? idn * (gx - q) : 0.0;
    os = vc ? (-rs - oz) * di : 0.0;                                        // :347,350
  };
  // get_step for (z, dz), (s, ds) (pdipm.py:182-186), NaN semantics as in the step kernel
  auto step_pair = [&](double z, double dz, double s, double ds) -> double {
    const double ninf = -inf_of<double>(), pinf = inf_of<double>();
    const double az = -z / dz, as = -s / ds;
    const uint32_t kmz = wave_umax(vc ? nan_key(az) : 0u), kms = wave_umax(vc ? nan_key(as) : 0u);
    const double mz = wave_max(vc ? az : ninf), ms = wave_max(vc ? as : ninf);
    const double fz = key_is_nan(kmz) ? 1.0 : __builtin_fmax(mz, 1.0), fs = key_is_nan(kms) ? 1.0 : __builtin_fmax(ms, 1.0);
    const double pz = (dz > 0.0) ? fz : az, ps = (ds > 0.0) ? fs : as;
    const uint32_t kl = wave_umax(vc ? umax(nan_key(pz), nan_key(ps)) : 0u);
    const double l = wave_min(vc ? __builtin_fmin(pz, ps) : pinf);
    return key_is_nan(kl) ? nan_of<double>() : l;
  };

  if constexpr (BWD) {
    // ---- backward: d(loss)/d(dp) -> d(loss)/d(Mdiag, v, rest, contact normal / arms, Je): lcp.py:37-64 on the frictionless LCP,
    // contracted through h = gc = (Jc v)(1 - rbar), b = ge = Je v, G = Jc (engines.py:84-112) ---------------------------------
    const double x = vx ? Wit[lane] : 0.0, nu_l = ve ? Wit[64 + (lane - nz)] : 0.0;
    double z = 1, s = 1, dinv = 1;
    if (vc) { z = Wit[ZO + lane]; s = Wit[ZO + LX + lane]; dinv = s / z; }
    constexpr double BWD_FLOOR = 1e-9;                                       // (as in lcp_primal_kernel: floored D + one refinement step)
    double dfl = dinv;
    {
      xv[lane] = vx ? 1.0 / qd : 0.0; wsync();
      double wn = 0;
      if (vc) {
#pragma unroll
        for (int q = 0; q < 6; ++q) wn = fma((double)jn[q] * (double)jn[q], xv[colq(q)], wn);
        dfl = __builtin_fmax(dinv, BWD_FLOOR * wn);
      }
      wsync();
    }
    idn = vc ? 1.0 / dfl : 0.0;
    factor();
    const double g = vx ? -(double)((const float*)Gd.dl_dv)[(size_t)scene * nz + lane] : 0.0;     // dp = -x (engines.py:115)
    double dx, ds, dl, dnu;
    solve_kkt(dfl, g, 0.0, 0.0, 0.0, dx, ds, dl, dnu);
    if (ncs > 0) {                                                           // refinement on the unreduced equations, true D
      double r1 = -g - (qd * dx + Gtw(vc ? dl : 0.0));
      if (e > 0) r1 -= Aty(dnu);
      if (!vx) r1 = 0.0;
      const double gx = Gv(dx);
      const double r3 = vc ? -(gx - dinv * dl) : 0.0;
      const double r2 = (e > 0) ? -Av(dx) : 0.0;
      double cx, cs, cl, cnu;
      solve_kkt(dfl, -r1, 0.0, -r3, -r2, cx, cs, cl, cnu);
      dx += cx; dnu += cnu; dl += cl;
    }
    double* X = Kl; double* DX = Kl + LX; double* CR = Kl + 2 * LX;
    wsync();
    X[lane] = x; DX[lane] = vx ? dx : 0.0; wsync();
    double djv = 0;
    {
      double cr = 0, dnx = 0, dny = 0, d1x = 0, d1y = 0, d2x = 0, d2y = 0;
      int b1 = 0, b2 = 0;
      if (vc) {
        const size_t cb = (size_t)scene * ncap + lane;
        const double nx = ((const float*)SP.c_n)[cb * 2], ny = ((const float*)SP.c_n)[cb * 2 + 1];
        const double p1x = ((const float*)SP.c_p1)[cb * 2], p1y = ((const float*)SP.c_p1)[cb * 2 + 1];
        const double p2x = ((const float*)SP.c_p2)[cb * 2], p2y = ((const float*)SP.c_p2)[cb * 2 + 1];
        b1 = SP.c_i1[cb]; b2 = SP.c_i2[cb];
        const double rbar = 0.5 * ((double)((const float*)SP.rest)[(size_t)scene * nb + b1] + (double)((const float*)SP.rest)[(size_t)scene * nb + b2]);
        const double jnd[6] = {p1x * ny - p1y * nx, nx, ny, -(p2x * ny - p2y * nx), -nx, -ny};     // world.py:177-183
        const double gh = -dl;                                                // dh = -dlam (lcp.py:56)
        djv = gh * (1.0 - rbar);                                              // h = (Jc v) + (Jc v) * -rbar (engines.py:89)
        double gjn[6], jnv = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int col = (q < 3) ? 3 * b1 + q : 3 * b2 + (q - 3);
          const double xq = X[col], dxq = DX[col], vq = (double)vv[col];
          jnv = fma(jnd[q], vq, jnv);
          gjn[q] = dl * xq + z * dxq + djv * vq;                              // dG row (lcp.py:53) + h through Jc
        }
        cr = 0.5 * (-gh * jnv);                                               // rbar = (rest_b1 + rest_b2) / 2 (world.py:144-151)
        dnx = -gjn[0] * p1y + gjn[1] + gjn[3] * p2y - gjn[4];
        dny = gjn[0] * p1x + gjn[2] - gjn[3] * p2x - gjn[5];
        d1x = gjn[0] * ny; d1y = -gjn[0] * nx;
        d2x = -gjn[3] * ny; d2y = gjn[3] * nx;
      }
      wsync();
      CR[lane] = cr; B12[lane] = b1; B12[LX + lane] = b2;
      if (lane < ncap) {
        const size_t cb = (size_t)scene * ncap + lane;
        if (Gd.dcn) { ((float*)Gd.dcn)[cb * 2] = (float)dnx; ((float*)Gd.dcn)[cb * 2 + 1] = (float)dny; }
        if (Gd.dcp1) { ((float*)Gd.dcp1)[cb * 2] = (float)d1x; ((float*)Gd.dcp1)[cb * 2 + 1] = (float)d1y; }
        if (Gd.dcp2) { ((float*)Gd.dcp2)[cb * 2] = (float)d2x; ((float*)Gd.dcp2)[cb * 2 + 1] = (float)d2y; }
      }
      wsync();
    }
    // v enters through gc = (1 - rbar) Jc v and ge = Je v: dv = Jc^T djv + Je^T db, db = -dnu (lcp.py:58)
    double dv = Gtw(vc ? djv : 0.0);
    if (e > 0) dv += Aty(ve ? -dnu : 0.0);
    if (vx) {
      const size_t o = (size_t)scene * nz + lane;
      if (Gd.dMdiag) ((float*)Gd.dMdiag)[o] = (float)(dx * x);               // Q = diag(M): dQ_jj = dx_j x_j (lcp.py:59-60); p = 0
      if (Gd.dv) ((float*)Gd.dv)[o] = (float)dv;
    }
    if (Gd.dJe && e > 0) {                                                    // dA = dnu (x) x + nu (x) dx (lcp.py:57) + db (x) v
      float* o = (float*)Gd.dJe + (size_t)scene * e * nz;
      const double vl = vx ? (double)vv[lane] : 0.0;
      for (int a = 0; a < e; ++a) {
        const double dn = bcast_lane(dnu, nz + a), nu = bcast_lane(nu_l, nz + a);
        if (vx) o[a * nz + lane] = (float)(dn * x + nu * dx - dn * vl);
      }
    }
    if (lane < nb && Gd.drest) {
      double ar = 0;
      for (int c = 0; c < ncs; ++c) {
        const double w = ((B12[c] == lane) ? 1.0 : 0.0) + ((B12[LX + c] == lane) ? 1.0 : 0.0);
        if (w != 0.0) ar += w * CR[c];
      }
      ((float*)Gd.drest)[(size_t)scene * nb + lane] = (float)ar;
    }
    return;
  }

  const int max_iter = SP.max_iter, lim = SP.lim;
  const double eps = SP.eps;
  const double mf = (double)ncs;
  double x = 0, y = 0, s = 1, z = 1, dinv = 1, bx = 0, by = 0, bz = 1, bs = 1;
  double best_resid = inf_of<double>();
  bool have_best = false, done = false;
  int n_not = 0, iters = 0;
  for (int it = -1; it < max_iter; ++it) {
    double rx = 0, ry = 0, rs = 0, rz = 0, mu = 0, resid = 0;
    if (it < 0) {                                                           // init: (p, 0, -h, -b), d = 1 (:57-63); p = 0
      rx = 0.0; ry = -b_in; rz = -hn; dinv = 1.0;
    } else {                                                                // residuals (:82-96), F = 0
      rx = Gtw(vc ? z : 0.0) + qd * x;
      if (e > 0) rx += Aty(y);
      if (!vx) rx = 0.0;
      rs = z;
      const double gx = Gv(x);
      rz = vc ? gx + s - hn : 0.0;
      ry = (e > 0) ? Av(x) - b_in : 0.0;
      const double n_rx = wave_sum(rx * rx), n_rz = wave_sum(rz * rz), n_ry = wave_sum(ry * ry);
      const double sz = wave_sum(vc ? s * z : 0.0);
      mu = sz / mf; mu = mu < 0 ? -mu : mu;
      resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + mf * mu;
      dinv = vc ? s / z : 1.0;
    }
    idn = 1.0 / dinv;
    if (!vc) idn = 0.0;
    factor();
    if (it >= 0 && !done) {
      ++iters;
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; done = true; }
      else {
        const bool improved = !have_best || (resid < best_resid);
        if (improved) { best_resid = resid; n_not = 0; have_best = true; bx = x; by = y; bz = z; bs = s; }
        else ++n_not;
        if (n_not == lim || best_resid < eps || mu > mu_limit<double>()) done = true;
      }
    }
    if (it >= 0 && it == max_iter - 1) done = true;
    if (done) break;
    double ax = 0, ay = 0, as_ = 0, az = 0;
    const int npass = (it < 0) ? 1 : 2;
    for (int pass = 0; pass < npass; ++pass) {
      double ox, oy, os, oz;
      solve_kkt(dinv, rx, rs, rz, ry, ox, os, oz, oy);
      if (it < 0) {
        x = ox; s = os; z = oz; y = oy;
        const uint32_t ks = wave_umax(vc ? nan_key(s) : 0u), kz = wave_umax(vc ? nan_key(z) : 0u);
        double smin = wave_min(vc ? s : inf_of<double>()), zmin = wave_min(vc ? z : inf_of<double>());
        if (key_is_nan(ks)) smin = nan_of<double>();
        if (key_is_nan(kz)) zmin = nan_of<double>();
        if (smin <= 0.0) s += 1.0 - smin;                                   // (:66-75)
        if (zmin <= 0.0) z += 1.0 - zmin;
        if (!vc) { s = 1.0; z = 1.0; }
        if (ncs == 0) { bx = x; by = y; done = true; }                              // engines.py:92-103: the direct solve, no LCP
      } else if (pass == 0) {
        ax = ox; ay = oy; as_ = os; az = oz;
        const double alpha = pmin(step_pair(z, az, s, as_), 1.0);
        const double t3 = wave_sum(vc ? (s + alpha * as_) * (z + alpha * az) : 0.0);
        const double t4 = wave_sum(vc ? s * z : 0.0);
        const double r3 = t3 / t4, sig = r3 * r3 * r3;
        rx = 0; ry = 0; rz = 0;
        rs = vc ? (-mu * sig + as_ * az) / s : 0.0;                         // (:153)
      } else {
        const double cx = ox + ax, cy = oy + ay, cs = os + as_, cz = oz + az;
        const double alpha = pmin(0.999 * step_pair(z, cz, s, cs), 1.0);
        x += alpha * cx; y += alpha * cy;
        if (vc) { s += alpha * cs; z += alpha * cz; }
      }
    }
    if (done) break;
  }
  const double dp = -bx;                                                    // engines.py:115
  if (__any(vx && (dp != dp))) status |= LCP_ST_NAN;
  if (vx) {
    ((float*)SP.v_new)[(size_t)scene * nz + lane] = (float)dp;
    if (SP.p_out64) {                                                       // world.py:110-117: dp /= 2 ; body.move(dt)
      const double dts = SP.dt_scene ? SP.dt_scene[scene] : SP.dt;
      SP.p_out64[(size_t)scene * nz + lane] = SP.pos64[(size_t)scene * nz + lane] + (dp * 0.5) * dts;
    }
  }
  if (Wg) {                                                                 // the best iterate, for lcp_post_stabilization_backward_f32
    if (lane == 0) Wg[0] = (double)ncs;
    if (vx) Wit[lane] = bx;
    if (ve) Wit[64 + (lane - nz)] = by;
    Wit[ZO + lane] = vc ? bz : 1.0; Wit[ZO + LX + lane] = vc ? bs : 1.0;
  }
  if (lane == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
}

}  // namespace primal

// nz + neq rows on the lanes of one wave, a contact per lane
bool primal_supported(int nz, int m, int e) {
  return (m % 4) == 0 && m / 4 <= 64 && e <= primal::WsLayout::YCAP && (nz % 3) == 0 && nz + e <= 56;
}
// the dense boundary and post-stabilisation keep the four-row instantiations
bool primal_dense_supported(int nz, int m, int e) { return e <= primal::EQB && primal_supported(nz, m, e); }
size_t primal_ws_bytes() { return sizeof(double) * (size_t)primal::WsLayout::TOTAL; }

template <int NCOL, bool BWD, bool DENSE = false>
static int primal_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, const DenseIO& DN = DenseIO{}) {
  if constexpr (!DENSE) {
    if (SP.e > primal::EQB) {                                             // 5 .. 24 equality rows: chains of joints
      hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, false, primal::WsLayout::YCAP>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
      return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
    }
  }
  hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, DENSE, primal::EQB>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <bool BWD, bool DENSE = false>
static int primal_dispatch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, const DenseIO& DN = DenseIO{}) {
  const int n = 3 * SP.nb + SP.e;
  if (n <= 24) return primal_launch<24, BWD, DENSE>(SP, Gd, stream, DN);
  if (n <= 40) return primal_launch<40, BWD, DENSE>(SP, Gd, stream, DN);
  return primal_launch<56, BWD, DENSE>(SP, Gd, stream, DN);
}
int primal_step(const StepArgs& SP, void* stream) { StepBwdArgs Gd = {}; return primal_dispatch<false>(SP, Gd, stream); }
int primal_step_backward(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) { return primal_dispatch<true>(SP, Gd, stream); }

template <bool BWD>
static int primal_post_stab_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) {
  const int n = 3 * SP.nb + SP.e;
  hipStream_t st = (hipStream_t)stream;
  constexpr int E16 = primal::WsLayout::YCAP;                               // (24 rows)
  if (SP.e > primal::EQB) {                                                  // chains of joints
    if (n <= 24) hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<24, BWD, E16>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
    else if (n <= 40) hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<40, BWD, E16>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
    else hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<56, BWD, E16>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
  } else {
    if (n <= 24) hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<24, BWD, primal::EQB>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
    else if (n <= 40) hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<40, BWD, primal::EQB>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
    else hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<56, BWD, primal::EQB>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
int primal_post_stab(const StepArgs& SP, void* stream) { StepBwdArgs Gd = {}; return primal_post_stab_launch<false>(SP, Gd, stream); }
int primal_post_stab_backward(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) { return primal_post_stab_launch<true>(SP, Gd, stream); }

// dense boundary: the scenes lcp_classify_big marked 3 (launched next to the contact-space and generic kernels, which take 2 and 0)
int primal_dense_forward(const FwdArgs& P, int32_t* cls, size_t ws_scene, void* stream) {
  DenseIO DN = {};
  DN.nz = P.nz; DN.m = P.m; DN.cls = cls; DN.ws_scene = ws_scene;
  DN.Q = (const float*)P.Q; DN.p = (const float*)P.p; DN.G = (const float*)P.G; DN.h = (const float*)P.h;
  DN.A = (const float*)P.A; DN.b = (const float*)P.b; DN.F = (const float*)P.F;
  StepArgs SP = {};
  SP.B = P.B; SP.nb = P.nz / 3; SP.nc = P.m / 4; SP.e = P.e; SP.ws = P.ws;
  SP.eps = P.eps; SP.max_iter = P.max_iter; SP.lim = P.lim;
  SP.v_new = P.x; SP.z = P.z; SP.s = P.s; SP.y = P.y; SP.iters = P.iters; SP.status = P.status;
  StepBwdArgs Gd = {};
  return primal_dispatch<false, true>(SP, Gd, stream, DN);
}
int primal_dense_backward(const BwdArgs& P, int32_t* cls, size_t ws_scene, void* stream) {
  DenseIO DN = {};
  DN.nz = P.nz; DN.m = P.m; DN.cls = cls; DN.ws_scene = ws_scene;
  DN.G = (const float*)P.G; DN.A = (const float*)P.A; DN.dl_dx = (const float*)P.dl_dx;
  DN.dQ = (float*)P.dQ; DN.dp = (float*)P.dp; DN.dG = (float*)P.dG; DN.dh = (float*)P.dh; DN.dA = (float*)P.dA; DN.db = (float*)P.db; DN.dF = (float*)P.dF;
  StepArgs SP = {};
  SP.B = P.B; SP.nb = P.nz / 3; SP.nc = P.m / 4; SP.e = P.e; SP.ws = P.ws;
  StepBwdArgs Gd = {};
  return primal_dispatch<true, true>(SP, Gd, stream, DN);
}

}  // namespace lcp
