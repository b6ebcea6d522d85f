// lcp_contacts.hip - batched narrow-phase contact generation + the position update of World.step_dt.
//
// SURVEY.md §8(f) rows 1-2 (the caller side of the LCP hot path):
//   reference (paths under /root/reference/lcp_physics/physics)            here
//   world.py:88-101   step_dt loop body: body.move(dt), find_contacts,       lcp_move_find_contacts_kernel
//                     penetration check, dt halving
//   bodies.py:80-82   Body.move  p <- p + v dt                               (prologue)
//   bodies.py:211-214 / :278-283 rotate_verts                                world_verts()
//   world.py:139-142  find_contacts (ODE broadphase -> here all pairs i<j)   pair loop
//   contacts.py:57-205 DiffContactHandler.__call__                           collide_pair()
//     :68-79 circle/circle   :80-141 circle/hull (GJK, SAT when inside)     circle_circle(), circle_hull()
//     :142-201 hull/hull (SAT both ways, incident edge, clipping)            hull_hull()
//   contacts.py:207-352 get_support, test_separations, get_incident_edge,    support(), test_separations(), incident_edge(),
//                     clip_segment_to_line, get_closest, barycentric         clip(), closest(), bary2(), bary3()
//
// One wavefront per scene, lane = body pair (i < j in body order, looping when there are more than 64
// pairs); the contacts of a scene are compacted in pair order with a wave prefix sum, so the list is the
// reference's list (tests compare index lists exactly).  Poses and all geometry are fp64 (the penetration
// test of step_dt uses tol = 1e-6 on coordinates of several hundred: fp32 cannot resolve it); the contact
// normals / arms handed to the LCP kernels are rounded to fp32.
// The two history-dependent tie-breakers of the reference (SAT warm start `last_sat_idx`, random GJK start
// vertex) are fixed to 0 as in the oracle; they only matter on exact ties.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcp_kernels.h"

namespace lcp {
namespace ct {

constexpr int NV = 8;          // max vertices of a hull
constexpr int MAXB = 32;       // max bodies per scene handled by the detection kernel (its LDS staging is sized by the NBMAX it is built for)

#define LCP_S double
#include "lcp_contacts_geom.inc"
#undef LCP_S

// ---- forward-mode derivative of the same geometry: value + ONE directional derivative ---------------------------------
// (the reference builds the contact tuple with differentiable torch operations, contacts.py:57-352; its autograd follows the
//  branches the forward pass took - so does this: every comparison looks at the values only)
namespace ad {
struct Dual {
  double v, d;
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(double a) : v(a), d(0.0) {}
  __device__ __forceinline__ Dual(double a, double b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return Dual(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual sqrt(Dual a) { const double r = ::sqrt(a.v); return Dual(r, 0.5 * a.d / r); }
__device__ __forceinline__ bool operator<(Dual a, Dual b) { return a.v < b.v; }
__device__ __forceinline__ bool operator<=(Dual a, Dual b) { return a.v <= b.v; }
__device__ __forceinline__ bool operator>(Dual a, Dual b) { return a.v > b.v; }
__device__ __forceinline__ bool operator>=(Dual a, Dual b) { return a.v >= b.v; }
__device__ __forceinline__ bool operator==(Dual a, Dual b) { return a.v == b.v; }
#define LCP_S Dual
#include "lcp_contacts_geom.inc"
#undef LCP_S
}  // namespace ad

// One launch = the whole position update of World.step_dt (world.py:88-101) for every scene: try the step with
// the current dt (Body.move), detect contacts, accept when no contact penetrates by more than `tol`, otherwise
// halve dt, go back to the start pose and retry.  The loop is per scene and runs on the device, so no host round trip.
// LPS = lanes per scene: 64 (one wave per scene, the pairs are walked 64 at a time) or 16 (four scenes per wave when a
// scene has at most 16 body pairs, i.e. nb <= 6 - the common case; the lanes of a scene are one DPP/shuffle row).
// Scenes that share a wave advance in lock step: a scene whose step is accepted keeps recomputing the same accepted
// trial (identical stores) until its neighbours are done.
template <int LPS, int NBMAX>
__global__ void __launch_bounds__(64) lcp_move_find_contacts_kernel(ContactArgs P) {
  constexpr int SPW = 64 / LPS;                                 // scenes per wave
  static_assert(NBMAX <= MAXB && (LPS == 64 || NBMAX * (NBMAX - 1) / 2 <= LPS), "bodies per scene");
  __shared__ V2 s_verts_all[SPW * NBMAX * NV];
  __shared__ V2 s_nrm_all[SPW * NBMAX * NV];
  __shared__ double s_elen_all[SPW * NBMAX * NV];
  __shared__ double s_pose_all[SPW * NBMAX * 3];
  __shared__ V2 s_vloc_all[SPW * NBMAX * NV];                   // body-frame vertices (constant over the trials)
  __shared__ V2 s_sc_all[SPW * NBMAX];                          // (sin, cos) of each body's rotation
  __shared__ double s_rad_all[SPW * NBMAX];
  __shared__ int s_kind_all[SPW * NBMAX], s_nv_all[SPW * NBMAX];
  const int lane = threadIdx.x, ll = lane % LPS, row = lane / LPS;
  const int scene_raw = blockIdx.x * SPW + row;
  const bool live = scene_raw < P.B;
  const int scene = live ? scene_raw : P.B - 1;                 // tail rows shadow the last scene and store nothing
  V2* s_verts = s_verts_all + row * NBMAX * NV;
  V2* s_nrm = s_nrm_all + row * NBMAX * NV;
  double* s_elen = s_elen_all + row * NBMAX * NV;
  V2* s_vloc = s_vloc_all + row * NBMAX * NV;
  V2* s_sc = s_sc_all + row * NBMAX;
  double* s_rad = s_rad_all + row * NBMAX;
  int* s_kind = s_kind_all + row * NBMAX;
  int* s_nv = s_nv_all + row * NBMAX;
  double* s_pose = s_pose_all + row * NBMAX * 3;
  const int nb = P.nb;
  const int npairs = nb * (nb - 1) / 2;
  double dt = P.dt;
  int base = 0, trial = 0;
  double maxpen = -1e300;
  bool done = false;
  // geometry does not change over the trials: stage it in LDS once
  for (int idx = ll; idx < nb * NV; idx += LPS) {
    const double* vl = P.verts_local + ((size_t)scene * nb) * NV * 2 + (size_t)idx * 2;
    s_vloc[idx] = v2(vl[0], vl[1]);
  }
  for (int b = ll; b < nb; b += LPS) {
    s_kind[b] = P.kind[(size_t)scene * nb + b]; s_nv[b] = P.nverts[(size_t)scene * nb + b]; s_rad[b] = P.radius[(size_t)scene * nb + b];
  }
  for (;;) {
    // bodies.py:80-82 (p <- p_start + v dt) and the vertex rotation of bodies.py:211-214
    for (int idx = ll; idx < nb * 3; idx += LPS) {
      double pv = P.p_start[(size_t)scene * nb * 3 + idx];
      if (P.v) pv += (double)P.v[(size_t)scene * nb * 3 + idx] * dt;
      s_pose[idx] = pv;
    }
    __syncthreads();
    for (int b = ll; b < nb; b += LPS) { const double rot = s_pose[b * 3]; s_sc[b] = v2(sin(rot), cos(rot)); }
    __syncthreads();
    for (int idx = ll; idx < nb * NV; idx += LPS) {
      const int bdy = idx / NV;
      const double sn = s_sc[bdy].x, cs = s_sc[bdy].y;
      const double lx = s_vloc[idx].x, ly = s_vloc[idx].y;
      s_verts[idx] = v2(cs * lx - sn * ly, sn * lx + cs * ly);                      // utils.py:105-112
    }
    __syncthreads();
    // edge normals and lengths of every hull, once per trial pose (every pair that touches the body re-uses them;
    // the reference recomputes them per pair: contacts.py:118-119,224-226,260-261)
    for (int idx = ll; idx < nb * NV; idx += LPS) {
      const int bdy = idx / NV, k = idx - bdy * NV;
      const int nvb = s_nv[bdy];
      if (k < nvb) {
        const V2 edge = s_verts[bdy * NV + (k + 1) % nvb] - s_verts[idx];
        const double en = norm(edge);
        s_elen[idx] = en;
        s_nrm[idx] = left_orth(edge) * (1.0 / en);
      }
    }
    __syncthreads();
    base = 0; maxpen = -1e300;
    for (int p0 = 0; p0 < npairs; p0 += LPS) {
      const int pr = p0 + ll;
      int cnt = 0, bi = 0, bj = 1;
      Pt pt0, pt1;
      pt0.n = v2(0, 0); pt0.p1 = pt0.n; pt0.p2 = pt0.n; pt0.pen = 0; pt1 = pt0;
      if (pr < npairs) {
        int rem = pr;                                           // pair index -> (i, j), i < j, lexicographic
        while (rem >= nb - 1 - bi) { rem -= nb - 1 - bi; ++bi; }
        bj = bi + 1 + rem;
        const bool skip = P.no_contact && P.no_contact[((size_t)scene * nb + bi) * nb + bj];
        if (!skip) {
          Body b1, b2;
          b1.kind = s_kind[bi]; b2.kind = s_kind[bj];
          b1.pos = v2(s_pose[bi * 3 + 1], s_pose[bi * 3 + 2]); b2.pos = v2(s_pose[bj * 3 + 1], s_pose[bj * 3 + 2]);
          b1.rad = s_rad[bi]; b2.rad = s_rad[bj];
          b1.nv = s_nv[bi]; b2.nv = s_nv[bj];
          b1.verts = s_verts + bi * NV; b2.verts = s_verts + bj * NV;
          b1.nrm = s_nrm + bi * NV; b2.nrm = s_nrm + bj * NV; b1.elen = s_elen + bi * NV; b2.elen = s_elen + bj * NV;
          cnt = collide_pair(b1, b2, P.eps, pt0, pt1);
        }
      }
      // exclusive prefix sum of cnt over the lanes of the scene (pair order = the reference's contact order)
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < LPS; off <<= 1) { const int o = __shfl_up(incl, off, LPS); if (ll >= off) incl += o; }
      const int excl = incl - cnt;
      const int total = __shfl(incl, LPS - 1, LPS);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const Pt& pt = q == 0 ? pt0 : pt1;
        const int slot = base + excl + q;
        if (q < cnt && slot < P.maxc && live) {
          const size_t o = (size_t)scene * P.maxc + slot;
          P.c_n[o * 2] = (float)pt.n.x; P.c_n[o * 2 + 1] = (float)pt.n.y;
          P.c_p1[o * 2] = (float)pt.p1.x; P.c_p1[o * 2 + 1] = (float)pt.p1.y;
          P.c_p2[o * 2] = (float)pt.p2.x; P.c_p2[o * 2 + 1] = (float)pt.p2.y;
          if (P.c_pen) P.c_pen[o] = pt.pen;
          P.c_i1[o] = bi; P.c_i2[o] = bj;
        }
        if (q < cnt) maxpen = pt.pen > maxpen ? pt.pen : maxpen;
      }
      base += total;
    }
#pragma unroll
    for (int off = LPS / 2; off > 0; off >>= 1) { const double o = __shfl_xor(maxpen, off, LPS); maxpen = o > maxpen ? o : maxpen; }
    if (!done) {
      ++trial;
      // world.py:95-101
      const bool ok = !(base > 0 && maxpen > P.tol);
      if (ok || (!P.strict && dt < P.dt_floor) || trial >= P.max_trials || !P.v) done = true;   // (max_trials: the reference would spin)
      else dt *= 0.5;
    }
    if (__all(done)) break;
    __syncthreads();
  }
  if (!live) return;
  // pad the unused contact slots with a harmless record (no normal, bodies 0/0)
  const int nfill = base < P.maxc ? base : P.maxc;
  for (int slot = nfill + ll; slot < P.maxc; slot += LPS) {
    const size_t o = (size_t)scene * P.maxc + slot;
    P.c_n[o * 2] = 0; P.c_n[o * 2 + 1] = 0; P.c_p1[o * 2] = 0; P.c_p1[o * 2 + 1] = 0; P.c_p2[o * 2] = 0; P.c_p2[o * 2 + 1] = 0;
    if (P.c_pen) P.c_pen[o] = 0;
    P.c_i1[o] = 0; P.c_i2[o] = 0;
  }
  if (P.p_out) for (int idx = ll; idx < nb * 3; idx += LPS) P.p_out[(size_t)scene * nb * 3 + idx] = s_pose[idx];
  if (ll == 0) {
    P.count[scene] = base;                                  // may exceed maxc: the caller checks
    if (P.max_pen) P.max_pen[scene] = base > 0 ? maxpen : 0.0;
    if (P.dt_used) P.dt_used[scene] = dt;
    if (P.t) P.t[scene] += dt;                              // world.py:122
    if (P.trials) P.trials[scene] = trial;
  }
}

// ---------------------------------------------------------------- backward of the contact frame
// The reference builds the contact tuple with differentiable torch operations (DiffContactHandler, contacts.py:57-205), so a
// loss after a roll-out reaches the poses through (normal, p1, p2) - `demos/grad_demo.py:45-50`, `experiments/inference.py`.
// This is that chain rule for every record type (circle / circle, circle / hull by GJK or SAT, hull / hull by SAT + clipping):
//     d(loss)/d(pose_b,q) = sum over the contacts of   g_n . dn/d(pose_b,q) + g_p1 . dp1/d(..) + g_p2 . dp2/d(..)
// with the partial derivatives of a pair's records taken in FORWARD mode: the pair's geometry code (lcp_contacts_geom.inc) is
// re-run on dual numbers, once per pose coordinate of its two bodies (6 passes), from the same pose the detection kernel saw.
// Hull vertices are R(rot) v_local, so d/d(rot) reaches normals, incident edges and clipped points the way the reference's
// incrementally rotated `verts` do (bodies.py:211-214).  One thread per scene walks the body pairs in the detection kernel's
// order (so the running contact index is the list's), pairs without a contact cost one value-only pass; fixed summation order.
constexpr int FB_T = 32;       // scenes (threads) per workgroup of the frame backward: its per-thread geometry lives in LDS (43 KB)
__global__ void __launch_bounds__(FB_T) lcp_contact_frame_backward_kernel(int B, int nb, int maxc, const int32_t* kind, const double* radius,
                                                                        const double* verts_local, const int32_t* nverts,
                                                                        const uint8_t* no_contact, const double* p, double eps,
                                                                        const int32_t* count, const float* g_n, const float* g_p1,
                                                                        const float* g_p2, double* dp) {
  using V2 = ad::V2; using Body = ad::Body; using Pt = ad::Pt; using Dual = ad::Dual;      // (functions: found through their arguments)
  // the pair's rotated vertices, edge normals and edge lengths (dual numbers, indexed by run-time vertex numbers): per-thread
  // slices of LDS - as local arrays they were 1296 B of scratch memory per lane.  (+1 element per slice: the threads' slices start
  // on different banks)
  __shared__ V2 s_v[4][FB_T][NV + 1];
  __shared__ Dual s_e[2][FB_T][NV + 1];
  const int scene = blockIdx.x * FB_T + threadIdx.x;
  if (scene >= B) return;
  double* out = dp + (size_t)scene * nb * 3;
  for (int i = 0; i < nb * 3; ++i) out[i] = 0.0;
  int ntot = count[scene];
  ntot = ntot < 0 ? 0 : (ntot > maxc ? maxc : ntot);
  const double* q = p + (size_t)scene * nb * 3;
  int base = 0;
  // one body of the pair at pose q with the derivative seeded on coordinate `seed` (0 rot, 1 x, 2 y; -1: none)
  auto build = [&](int b, int seed, V2* verts, V2* nrm, Dual* elen, Body& body) {
    body.kind = kind[(size_t)scene * nb + b];
    body.rad = Dual(radius[(size_t)scene * nb + b]);
    body.nv = nverts[(size_t)scene * nb + b];
    body.pos = v2(Dual(q[b * 3 + 1], seed == 1 ? 1.0 : 0.0), Dual(q[b * 3 + 2], seed == 2 ? 1.0 : 0.0));
    body.verts = verts; body.nrm = nrm; body.elen = elen;
    if (body.kind != 0) {
      const double rot = q[b * 3], sn = sin(rot), cs = cos(rot), dr = seed == 0 ? 1.0 : 0.0;
      const Dual S(sn, cs * dr), C(cs, -sn * dr);
      const double* vl = verts_local + ((size_t)scene * nb + b) * NV * 2;
      for (int k = 0; k < body.nv; ++k) { const Dual lx(vl[2 * k]), ly(vl[2 * k + 1]); verts[k] = v2(C * lx - S * ly, S * lx + C * ly); }    // utils.py:105-112
      for (int k = 0; k < body.nv; ++k) {
        const V2 edge = verts[(k + 1) % body.nv] - verts[k];
        const Dual en = norm(edge);
        elen[k] = en; nrm[k] = left_orth(edge) * (Dual(1.0) / en);
      }
    }
  };
  for (int bi = 0; bi < nb && base < ntot; ++bi) {
    for (int bj = bi + 1; bj < nb && base < ntot; ++bj) {
      if (no_contact && no_contact[((size_t)scene * nb + bi) * nb + bj]) continue;
      V2 *v1 = s_v[0][threadIdx.x], *n1 = s_v[1][threadIdx.x], *v2_ = s_v[2][threadIdx.x], *n2 = s_v[3][threadIdx.x];
      Dual *e1 = s_e[0][threadIdx.x], *e2 = s_e[1][threadIdx.x];
      Body b1, b2;
      Pt pt0, pt1;
      build(bi, -1, v1, n1, e1, b1); build(bj, -1, v2_, n2, e2, b2);
      const int cnt = collide_pair(b1, b2, eps, pt0, pt1);
      if (cnt == 0) continue;
      for (int s = 0; s < 6; ++s) {                                           // the six pose coordinates of the pair
        build(bi, s < 3 ? s : -1, v1, n1, e1, b1); build(bj, s >= 3 ? s - 3 : -1, v2_, n2, e2, b2);
        const int c2 = collide_pair(b1, b2, eps, pt0, pt1);                   // (same values, hence the same branches and count)
        double acc = 0.0;
        for (int c = 0; c < c2 && base + c < ntot; ++c) {
          const Pt& pt = c == 0 ? pt0 : pt1;
          const size_t o = ((size_t)scene * maxc + base + c) * 2;
          acc += (double)g_n[o] * pt.n.x.d + (double)g_n[o + 1] * pt.n.y.d + (double)g_p1[o] * pt.p1.x.d + (double)g_p1[o + 1] * pt.p1.y.d
               + (double)g_p2[o] * pt.p2.x.d + (double)g_p2[o + 1] * pt.p2.y.d;
        }
        out[(s < 3 ? bi : bj) * 3 + (s % 3)] += acc;
      }
      base += cnt;
    }
  }
}

// ---------------------------------------------------------------- joints whose Jacobian follows the pose
// World.Je (world.py:156-170) over the joints' J() and Joint.move (constraints.py:13-217) for B scenes; one thread per scene.
//   jtype: 1 Joint (revolute, 2 rows; anchor = body1.pos + r1 (cos rot1, sin rot1), constraints.py:13-50)
//          2 FixedJoint (3 rows, :56-92)   3 XConstraint   4 YConstraint   5 RotConstraint (1 row each, :95-172)
//          6 TotalConstraint (3 rows, :175-192)   0 empty slot
// With `v`: first rot1 += vscale * v[body1][0] * dt_k (Joint.move :39-43; dt_k = the dt the scene's step accepted - the
// retry loop of world.py:88-107 restores rot1 before every trial, so only the accepted dt counts), then Je at pose p.
__global__ void __launch_bounds__(64) lcp_joint_jacobian_kernel(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1,
                                                                const int32_t* jb2, const double* jr1, double* jrot1, const double* p,
                                                                const float* v, const double* dt_scene, double dt, double vscale, float* Je) {
  const int scene = blockIdx.x * 64 + threadIdx.x;
  if (scene >= B) return;
  const int nz = 3 * nb;
  float* J = Je + (size_t)scene * e * nz;
  for (int i = 0; i < e * nz; ++i) J[i] = 0.0f;
  const double* q = p + (size_t)scene * nb * 3;
  const double dtk = dt_scene ? dt_scene[scene] : dt;
  int row = 0;
  for (int k = 0; k < nj; ++k) {
    const size_t o = (size_t)scene * nj + k;
    const int t = jtype[o], b1 = jb1[o], b2 = jb2[o];
    if (t == 1 || t == 2) {
      double p1x = 0, p1y = 0;
      if (t == 1) {
        double rot = jrot1[o];
        if (v) { rot += vscale * (double)v[((size_t)scene * nb + b1) * 3] * dtk; jrot1[o] = rot; }
        p1x = jr1[o] * cos(rot); p1y = jr1[o] * sin(rot);                          // polar_to_cart (utils.py:85-90)
      }
      if (row + 1 < e) {
        J[(size_t)row * nz + 3 * b1] = (float)(-p1y); J[(size_t)row * nz + 3 * b1 + 1] = 1.0f;
        J[(size_t)(row + 1) * nz + 3 * b1] = (float)p1x; J[(size_t)(row + 1) * nz + 3 * b1 + 2] = 1.0f;
        if (b2 >= 0) {
          const double p2x = q[b1 * 3 + 1] + p1x - q[b2 * 3 + 1], p2y = q[b1 * 3 + 2] + p1y - q[b2 * 3 + 2];
          J[(size_t)row * nz + 3 * b2] = (float)p2y; J[(size_t)row * nz + 3 * b2 + 1] = -1.0f;
          J[(size_t)(row + 1) * nz + 3 * b2] = (float)(-p2x); J[(size_t)(row + 1) * nz + 3 * b2 + 2] = -1.0f;
        }
      }
      row += 2;
      if (t == 2) {
        if (row < e) { J[(size_t)row * nz + 3 * b1] = 1.0f; if (b2 >= 0) J[(size_t)row * nz + 3 * b2] = -1.0f; }
        row += 1;
      }
    } else if (t == 3) { if (row < e) J[(size_t)row * nz + 3 * b1 + 1] = 1.0f; row += 1; }
    else if (t == 4) { if (row < e) J[(size_t)row * nz + 3 * b1 + 2] = 1.0f; row += 1; }
    else if (t == 5) { if (row < e) J[(size_t)row * nz + 3 * b1] = 1.0f; row += 1; }
    else if (t == 6) { if (row + 2 < e) for (int i = 0; i < 3; ++i) J[(size_t)(row + i) * nz + 3 * b1 + i] = 1.0f; row += 3; }
  }
}

// Backward of lcp_joint_jacobian_kernel with respect to the pose and the revolute joints' angles - what the reference's autograd
// computes through Joint.J() / FixedJoint.J() and update_pos (constraints.py:26-50, 64-85): the four pose-dependent entries of a
// joint's rows are e0 = -pos1_y, e1 = pos1_x (columns of body 1's angle) and e2 = pos2_y, e3 = -pos2_x (body 2's), with
// pos1 = r1 (cos rot1, sin rot1), pos2 = body1.pos + pos1 - body2.pos.  jrot1: the angles Je was evaluated at.  One thread per scene.
__global__ void __launch_bounds__(64) lcp_joint_jacobian_backward_kernel(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1,
                                                                         const int32_t* jb2, const double* jr1, const double* jrot1,
                                                                         const float* gJe, double* g_p, double* g_rot) {
  const int scene = blockIdx.x * 64 + threadIdx.x;
  if (scene >= B) return;
  const int nz = 3 * nb;
  const float* G = gJe + (size_t)scene * e * nz;
  double* gq = g_p + (size_t)scene * nz;
  for (int i = 0; i < nz; ++i) gq[i] = 0.0;
  int row = 0;
  for (int k = 0; k < nj; ++k) {
    const size_t o = (size_t)scene * nj + k;
    const int t = jtype[o], b1 = jb1[o], b2 = jb2[o];
    double gr = 0.0;
    if (t == 1 || t == 2) {
      if (row + 1 < e) {
        const double a0 = (double)G[(size_t)row * nz + 3 * b1], a1 = (double)G[(size_t)(row + 1) * nz + 3 * b1];
        double a2 = 0, a3 = 0;
        if (b2 >= 0) {
          a2 = (double)G[(size_t)row * nz + 3 * b2]; a3 = (double)G[(size_t)(row + 1) * nz + 3 * b2];
          gq[b1 * 3 + 1] += -a3; gq[b1 * 3 + 2] += a2;                       // pos2 = body1.pos + pos1 - body2.pos
          gq[b2 * 3 + 1] += a3; gq[b2 * 3 + 2] += -a2;
        }
        if (t == 1) {                                                        // (a FixedJoint's anchor is body 1 itself: pos1 = 0)
          const double rot = jrot1[o];
          gr = jr1[o] * (-sin(rot) * (a1 - a3) + cos(rot) * (a2 - a0));
        }
      }
      row += (t == 2) ? 3 : 2;
    } else if (t == 3 || t == 4 || t == 5) row += 1;
    else if (t == 6) row += 3;
    g_rot[o] = gr;
  }
}

// Backward of the state update of a differentiable step - Body.move and Joint.move (bodies.py:80-82, 199-202;
// constraints.py:39-43) - with respect to the velocities the move used:
//   p_new   = p + scale v dt_k                                         cotangent g_p
//   geo_new = p_geo + [scale v dt_k where it is non-zero or the coordinate is x / y]   cotangent g_g (the reference turns a hull's
//             vertices by the increment and skips the turn when it is zero: no vertex path through such a step)
//   rot_new = rot + scale v[body1][0] dt_k  for revolute joints        cotangent g_rot
// out: g_v = d(loss)/dv (float32).  dt_k > 0 (the step_dt loop stops at dt / 4): the increment is zero exactly where v is.
// scale = 1 for the dynamics move, 0.5 for the post-stabilisation move (world.py:112).  One thread per scene.
__global__ void __launch_bounds__(64) lcp_state_update_backward_kernel(int B, int nb, int nj, const double* g_p, const double* g_g,
                                                                       const double* g_rot, const float* v, const double* dt_scene,
                                                                       double scale, const int32_t* jtype, const int32_t* jb1, float* g_v) {
  const int scene = blockIdx.x * 64 + threadIdx.x;
  if (scene >= B) return;
  const double k = scale * dt_scene[scene];
  const size_t base = (size_t)scene * nb * 3;
  for (int b = 0; b < nb; ++b) {
    double rot_extra = 0.0;
    if (g_rot) for (int j = 0; j < nj; ++j) { const size_t o = (size_t)scene * nj + j; if (jtype[o] == 1 && jb1[o] == b) rot_extra += g_rot[o]; }
    for (int c = 0; c < 3; ++c) {
      const size_t i = base + b * 3 + c;
      double tot = g_p ? g_p[i] : 0.0;
      if (g_g && (c > 0 || v[i] != 0.0f)) tot += g_g[i];
      if (c == 0) tot += rot_extra;
      g_v[i] = (float)(tot * k);
    }
  }
}

}  // namespace ct

int joint_jacobian_backward_launch(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                                   const double* jr1, const double* jrot1, const float* gJe, double* g_p, double* g_rot, void* stream) {
  hipLaunchKernelGGL(ct::lcp_joint_jacobian_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, nb, nj, e, jtype, jb1,
                     jb2, jr1, jrot1, gJe, g_p, g_rot);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int state_update_backward_launch(int B, int nb, int nj, const double* g_p, const double* g_g, const double* g_rot, const float* v,
                                 const double* dt_scene, double scale, const int32_t* jtype, const int32_t* jb1, float* g_v, void* stream) {
  hipLaunchKernelGGL(ct::lcp_state_update_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, nb, nj, g_p, g_g, g_rot, v,
                     dt_scene, scale, jtype, jb1, g_v);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int joint_jacobian_launch(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2, const double* jr1,
                          double* jrot1, const double* p, const float* v, const double* dt_scene, double dt, double vscale, float* Je,
                          void* stream) {
  hipLaunchKernelGGL(ct::lcp_joint_jacobian_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, nb, nj, e, jtype, jb1, jb2,
                     jr1, jrot1, p, v, dt_scene, dt, vscale, Je);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int contact_frame_backward_launch(int B, int nb, int maxc, const int32_t* kind, const double* radius, const double* verts_local,
                                  const int32_t* nverts, const uint8_t* no_contact, const double* p, double eps,
                                  const int32_t* count, const float* g_n, const float* g_p1, const float* g_p2, double* dp,
                                  void* stream) {
  if (nb > ct::MAXB) return LCP_E_TOOLARGE;
  hipLaunchKernelGGL(ct::lcp_contact_frame_backward_kernel, dim3((B + ct::FB_T - 1) / ct::FB_T), dim3(ct::FB_T), 0, (hipStream_t)stream, B, nb, maxc, kind,
                     radius, verts_local, nverts, no_contact, p, eps, count, g_n, g_p1, g_p2, dp);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int contacts_launch(const ContactArgs& P, void* stream) {
  if (P.nb > ct::MAXB) return LCP_E_TOOLARGE;
  if (P.nb <= 6) hipLaunchKernelGGL((ct::lcp_move_find_contacts_kernel<16, 6>), dim3((P.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, P);
  else if (P.nb <= 16) hipLaunchKernelGGL((ct::lcp_move_find_contacts_kernel<64, 16>), dim3(P.B), dim3(64), 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL((ct::lcp_move_find_contacts_kernel<64, ct::MAXB>), dim3(P.B), dim3(64), 0, (hipStream_t)stream, P);   // (15 KB of LDS per scene)
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
