// lcp_primal_poststab.hip - PdipmEngine.post_stabilization (engines.py:80-116) and its backward in body space, one wave per scene
// (the mapping of lcp_primal.hip), behind lcp_post_stabilization_f32 / lcp_post_stabilization_backward_f32.
#include "lcp_primal_common.h"

namespace lcp {
namespace primal {
// (lcp_quad_kernels.inc's post-stabilisation forward writes this layout by its own constants: POST_WS_IT / _ZO / _TOTAL)
static_assert(WsLayout::IT == 64 && ZO == 88 && WsLayout::TOTAL == 792 && LX == 64, "workspace layout shared with lcp_fwd_quad<..., POST>");

// ---------------------------------------------------------------- post-stabilisation (engines.py:80-116; world.py:109-117)
// The frictionless LCP of PdipmEngine.post_stabilization - Q = M, p = 0, G = Jc, h = gc = Jc v + Jc v * -restitutions, A = Je,
// b = ge = Je v, F = 0: ONE inequality row per contact - solved in body space like the step above (the block M is the scalar
// D = s / z), then dp = -x and, when poses are given, the correction move p_out = p + (dp / 2) dt_scene.  No contact: the direct
// KKT solve of :92-103, which is what the initialisation solve computes.  One wave per scene; replaces the generic
// workgroup-per-scene kernel on this path (4.2 ms for 4096 x 16 contacts).
template <int NCOL, bool BWD, int EQC>
__global__ void __launch_bounds__(64, (NCOL <= 40 ? 2 : 1)) lcp_poststab_primal_kernel(StepArgs SP, StepBwdArgs Gd) {
  constexpr int LDK = ((NCOL + 1) % 32 == 31) ? NCOL + 3 : NCOL + 1;     // (never -1 mod 32: lcp_primal_step.inc)
  __shared__ __attribute__((aligned(16))) double Kl[NCOL * LDK];
  __shared__ double xv[LX];
  constexpr int ASZ = (EQC <= 4) ? EQC * LX : at_cap(NCOL, EQC);    // the A image: rows of 64 (few rows) or packed e x nz (many rows)
  __shared__ float At[ASZ];
  __shared__ int B12[2 * LX];
  const int scene = blockIdx.x, lane = threadIdx.x;
  if (!BWD && blockIdx.x == 0 && lane == 0 && SP.tag) *SP.tag = SP.tag_value;   // workspace trailer: which kernel family laid it out
  const int nb = SP.nb, nz = 3 * nb, ncap = SP.nc, e = SP.e, n = nz + e;
  const int AST = (EQC <= 4) ? LX : nz;                                      // row stride of the A image
  // workspace per scene (when given): the count and the best iterate, for the backward: [ncs .. | x[64] y[16] z[64] s[64]]
  double* Wg = SP.ws ? (double*)SP.ws + (size_t)scene * (size_t)WsLayout::TOTAL : nullptr;
  double* Wit = Wg ? Wg + WsLayout::IT : nullptr;
  int ncs = ncap;
  if (BWD) ncs = (int)Wg[0];
  else if (SP.c_count) ncs = SP.c_count[scene];
  const int truncated = (ncs > ncap) ? LCP_ST_TRUNCATED : 0;
  ncs = ncs < 0 ? 0 : (ncs > ncap ? ncap : ncs);
  const int ci = (LCP_PRIMAL_CPERM && ncap > 16) ? (4 * (lane & 15) + (lane >> 4)) : lane;   // the lane's contact (lcp_primal_step.inc: neighbours of the list in different 16-lane rows)
  const bool vc = ci < ncs, vx = lane < nz, ve = lane >= nz && lane < n;
  const float* Md = (const float*)SP.Mdiag + (size_t)scene * nz;
  const float* vv = (const float*)SP.v + (size_t)scene * nz;
  float jn[6] = {0, 0, 0, 0, 0, 0};
  int c0 = 0, c1 = 0;
  double hn = 0;
  if (vc) {
    const ContactRows<float> r = make_contact<float>((const float*)SP.c_n + (size_t)scene * ncap * 2, (const float*)SP.c_p1 + (size_t)scene * ncap * 2,
                                                     (const float*)SP.c_p2 + (size_t)scene * ncap * 2, SP.c_i1 + (size_t)scene * ncap,
                                                     SP.c_i2 + (size_t)scene * ncap, (const float*)SP.rest + (size_t)scene * nb,
                                                     (const float*)SP.rest + (size_t)scene * nb, vv, ci);
#pragma unroll
    for (int q = 0; q < 6; ++q) jn[q] = r.jn[q];
    c0 = 3 * r.b1; c1 = 3 * r.b2;
    hn = (double)r.jv + (double)r.jv * -(double)r.rbar;                      // engines.py:87-89
  }
  auto colq = [&](int q) { return q < 3 ? c0 + q : c1 + (q - 3); };
  const double qd = vx ? (double)Md[lane] : 0.0;
  for (int i = lane; i < ASZ; i += 64) At[i] = 0.0f;
  wsync();
  for (int i = lane; i < e * nz; i += 64) { const int a = i / nz, k = i - a * nz; At[a * AST + k] = ((const float*)SP.Je)[(size_t)scene * e * nz + i]; }
  wsync();
  int status = truncated;
  if (__any(vx && !(qd != 0.0))) status |= LCP_ST_SINGULAR_Q;
  auto acol = [&](int a) -> double { return (EQC <= 4 || lane < nz) ? (double)At[a * AST + lane] : 0.0; };
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
    {
      // (lcp_primal_step.inc: the fill cannot be the minimum where both vectors keep an entry and nothing is NaN - same value)
      const bool nn = key_is_nan(umax(nan_key(az), nan_key(as)));
      const uint32_t fz_ = wave_umax(vc ? (nn ? 8u : (!(dz > 0.0) ? 1u : 0u)) : 0u), fs_ = wave_umax((vc && !(ds > 0.0)) ? 1u : 0u);
      if (fz_ == 1u && fs_ == 1u) return wave_min(vc ? __builtin_fmin((dz > 0.0) ? pinf : az, (ds > 0.0) ? pinf : as) : pinf);
    }
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
    double g = vx ? -(double)((const float*)Gd.dl_dv)[(size_t)scene * nz + lane] : 0.0;     // dp = -x (engines.py:115)
    if (SP.tag && *SP.tag != SP.tag_value) g = nan_of<double>();          // (another family's workspace: NaN gradients instead of a misread)
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
        const size_t cb = (size_t)scene * ncap + ci;
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
      CR[ci] = cr; B12[ci] = b1; B12[LX + ci] = b2;
      if (ci < ncap) {
        const size_t cb = (size_t)scene * ncap + ci;
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
    else if (n <= 56) hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<56, BWD, primal::EQB>), dim3(SP.B), dim3(64), 0, st, SP, Gd);
    else hipLaunchKernelGGL((primal::lcp_poststab_primal_kernel<64, BWD, primal::EQB>), dim3(SP.B), dim3(64), 0, st, SP, Gd);   // (round 6: 18 .. 20 bodies, every lane a row)
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
int primal_post_stab(const StepArgs& SP, void* stream) { StepBwdArgs Gd = {}; return primal_post_stab_launch<false>(SP, Gd, stream); }
int primal_post_stab_backward(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) { return primal_post_stab_launch<true>(SP, Gd, stream); }

}  // namespace lcp
