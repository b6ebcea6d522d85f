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
constexpr int MAXB = 16;       // max bodies per scene handled by this kernel

struct V2 { double x, y; };
__device__ __forceinline__ V2 v2(double x, double y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ V2 operator-(V2 a) { return v2(-a.x, -a.y); }
__device__ __forceinline__ V2 operator*(V2 a, double s) { return v2(a.x * s, a.y * s); }
__device__ __forceinline__ V2 operator*(double s, V2 a) { return v2(a.x * s, a.y * s); }
__device__ __forceinline__ double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ double norm(V2 a) { return sqrt(a.x * a.x + a.y * a.y); }
__device__ __forceinline__ V2 left_orth(V2 v) { return v2(v.y, -v.x); }              // utils.py:99-102

struct Body {             // world frame
  int kind;               // 0 circle, 1 hull
  V2 pos;
  double rad;
  int nv;
  const V2* verts;        // hull vertices relative to pos, rotated (LDS)
  const V2* nrm;          // outward unit normal of edge k = (verts[k], verts[k+1])  (LDS, formed once per trial pose)
  const double* elen;     // length of edge k
};

struct Pt { V2 n, p1, p2; double pen; };

// contacts.py:207-217 (`>=`: last maximiser wins)
__device__ __forceinline__ int support(const V2* pts, int n, V2 dir) {
  int best = -1; double bn = -1.0;
  for (int i = 0; i < n; ++i) { const double c = dot(pts[i], dir); if (c >= bn) { bn = c; best = i; } }
  return best;
}

__device__ __forceinline__ int circle_circle(const Body& b1, const Body& b2, double eps, Pt& out0) {   // contacts.py:68-79
  const double r = b1.rad + b2.rad;
  V2 n = b1.pos - b2.pos;
  const double dist = norm(n);
  const double pen = r - dist;
  if (pen < -eps) return 0;
  n = n * (1.0 / dist);
  out0.n = n; out0.p1 = -n * (b1.rad - pen / 2); out0.p2 = n * (b2.rad - pen / 2); out0.pen = pen;
  return 1;
}

__device__ __forceinline__ void bary2(V2 p, V2 a, V2 b, double& u, double& v) {        // contacts.py:334-340
  const V2 d = b - a;
  const double n = norm(d);
  const V2 nd = d * (1.0 / n);
  u = dot(b - p, nd) / n; v = dot(p - a, nd) / n;
}
__device__ __forceinline__ void bary3(V2 p, V2 a, V2 b, V2 c, double& u, double& v, double& w) {   // contacts.py:341-350
  // inverse of [[ax,bx,cx],[ay,by,cy],[1,1,1]] applied to (px,py,1)
  const double det = a.x * (b.y - c.y) - b.x * (a.y - c.y) + c.x * (a.y - b.y);
  const double id = 1.0 / det;
  u = ((b.y - c.y) * p.x + (c.x - b.x) * p.y + (b.x * c.y - c.x * b.y)) * id;
  v = ((c.y - a.y) * p.x + (a.x - c.x) * p.y + (c.x * a.y - a.x * c.y)) * id;
  w = ((a.y - b.y) * p.x + (b.x - a.x) * p.y + (a.x * b.y - b.x * a.y)) * id;
}

// (no private arrays anywhere below: dynamically indexed locals live in scratch memory on this target, and a scratch
//  round trip costs as much as the whole arithmetic of a pair - the simplex, clip and manifold buffers are named scalars)
struct Simplex { V2 a, b, c; int ia, ib, ic; int n; };     // up to 3 points with their vertex indices

// contacts.py:295-330: closest point of the simplex to p; `keep` = the sub-simplex that supports it (the reference's
// `ids_used`, in its order), keep.n == 3 means p is inside the triangle
__device__ __forceinline__ V2 closest(V2 p, const Simplex& sx, Simplex& keep) {
  keep = sx;
  if (sx.n == 1) return sx.a;
  if (sx.n == 2) {
    double u, v; bary2(p, sx.a, sx.b, u, v);
    if (u <= 0) { keep.a = sx.b; keep.ia = sx.ib; keep.n = 1; return sx.b; }
    if (v <= 0) { keep.n = 1; return sx.a; }
    return u * sx.a + v * sx.b;
  }
  double uAB, vAB, uBC, vBC, uCA, vCA, uABC, vABC, wABC;
  bary2(p, sx.a, sx.b, uAB, vAB); bary2(p, sx.b, sx.c, uBC, vBC); bary2(p, sx.c, sx.a, uCA, vCA);
  bary3(p, sx.a, sx.b, sx.c, uABC, vABC, wABC);
  if (vAB <= 0 && uCA <= 0) { keep.n = 1; return sx.a; }
  if (vBC <= 0 && uAB <= 0) { keep.a = sx.b; keep.ia = sx.ib; keep.n = 1; return sx.b; }
  if (vCA <= 0 && uBC <= 0) { keep.a = sx.c; keep.ia = sx.ic; keep.n = 1; return sx.c; }
  if (uAB > 0 && vAB > 0 && wABC <= 0) { keep.n = 2; return uAB * sx.a + vAB * sx.b; }
  if (uBC > 0 && vBC > 0 && uABC <= 0) { keep.a = sx.b; keep.ia = sx.ib; keep.b = sx.c; keep.ib = sx.ic; keep.n = 2; return uBC * sx.b + vBC * sx.c; }
  if (uCA > 0 && vCA > 0 && vABC <= 0) { keep.a = sx.c; keep.ia = sx.ic; keep.b = sx.a; keep.ib = sx.ia; keep.n = 2; return uCA * sx.c + vCA * sx.a; }
  return p;                                                      // inside (the reference raises if nothing matched)
}

// contacts.py:80-141: `circ` plays b1, `hull` b2
__device__ __forceinline__ int circle_hull(const Body& circ, const Body& hull, double eps, bool circle_is_g2, Pt& out0) {
  const V2* verts = hull.verts;
  const int nv = hull.nv;
  const V2 tp = circ.pos - hull.pos;
  Simplex sx, keep;
  sx.a = verts[0]; sx.ia = 0; sx.b = sx.a; sx.c = sx.a; sx.ib = -1; sx.ic = -1; sx.n = 1;
  keep = sx;
  V2 cl = sx.a;
  for (int iter = 0; iter < 4 * NV; ++iter) {
    cl = closest(tp, sx, keep);
    if (keep.n == 3) break;
    V2 sd;
    if (keep.n == 2) {
      sd = left_orth(keep.a - keep.b);
      if (dot(sd, tp - keep.a) < 0) sd = -sd;
    } else {
      sd = tp - cl;
    }
    if (sd.x == 0 && sd.y == 0) break;
    const int si = support(verts, nv, sd);
    const bool in_simplex = (si == sx.ia) || (sx.n > 1 && si == sx.ib) || (sx.n > 2 && si == sx.ic);
    if (in_simplex) break;
    sx = keep;                                                   // the used points, then the new support point
    if (keep.n == 1) { sx.b = verts[si]; sx.ib = si; } else { sx.c = verts[si]; sx.ic = si; }
    sx.n = keep.n + 1;
  }
  V2 bn, bp1, bp2; double bd;
  if (keep.n < 3) {
    bp2 = cl;
    const V2 cw = cl + hull.pos;
    bp1 = cw - circ.pos;
    bd = norm(cw - circ.pos) - circ.rad;
    if (bd > eps) return 0;
    bn = -bp1 * (1.0 / norm(bp1));
  } else {                                     // centre inside the hull: SAT, contacts.py:114-137
    bd = -1e10; bn = v2(0, 0); bp1 = bn; bp2 = bn;
    for (int idx = 0; idx < nv; ++idx) {
      const V2 nrm = hull.nrm[idx];
      const V2 center = circ.pos - hull.pos;
      const double dist = dot(nrm, center - verts[idx]) - circ.rad;
      if (dist > bd) {
        if (dist > eps) return 0;
        bd = dist; bn = nrm;
        bp2 = center + nrm * -(dist + circ.rad);
        bp1 = bp2 + hull.pos - circ.pos;
      }
    }
  }
  if (circle_is_g2) { bn = -bn; const V2 t = bp1; bp1 = bp2; bp2 = t; }
  out0.n = bn; out0.p1 = bp1; out0.p2 = bp2; out0.pen = -bd;
  return 1;
}

struct Sep { double dist; V2 normal; int vertex; double edge_norm; int edge; };

__device__ __forceinline__ Sep test_separations(const Body& h1, const Body& h2, double eps) {      // contacts.py:220-250
  Sep best; best.dist = -1e10; best.normal = v2(0, 0); best.vertex = -1; best.edge_norm = 0; best.edge = 0;
  for (int idx = 0; idx < h1.nv; ++idx) {
    const double en = h1.elen[idx];
    const V2 nrm = h1.nrm[idx];
    const int si = support(h2.verts, h2.nv, -nrm);
    const V2 sp = h2.verts[si] + h2.pos - h1.pos;
    const double dist = dot(nrm, sp - h1.verts[idx]);
    if (dist > best.dist) {
      if (dist > eps) { best.dist = dist; best.edge = idx; return best; }
      best.dist = dist; best.normal = -nrm; best.vertex = si; best.edge_norm = en; best.edge = idx;
    }
  }
  return best;
}

__device__ __forceinline__ int incident_edge(V2 ref_normal, const Body& inc, int inc_vertex) {      // contacts.py:253-268
  double min_dot = 1e10; int best = -1;
  const int e0 = (inc_vertex - 1 + inc.nv) % inc.nv;
  for (int q = 0; q < 2; ++q) {
    const int i = q == 0 ? e0 : inc_vertex;
    const V2 inrm = inc.nrm[i];
    const double d = dot(ref_normal, inrm);
    if (d < min_dot) { min_dot = d; best = i; }
  }
  return best;
}

__device__ __forceinline__ int clip(V2 in0, V2 in1, V2 nrm, double offset, V2& o0, V2& o1) {       // contacts.py:270-292
  int n = 0;
  const double d0 = dot(nrm, in0) + offset, d1 = dot(nrm, in1) + offset;
  o0 = in0; o1 = in1;
  if (d0 >= 0.0) { o0 = in0; n = 1; }
  if (d1 >= 0.0) { if (n == 0) o0 = in1; else o1 = in1; ++n; }
  if (d0 * d1 < 0.0 || n < 2) {
    const double interp = d0 / (d0 - d1);
    const V2 x = in0 + interp * (in1 - in0);
    if (n == 0) o0 = x; else if (n == 1) o1 = x;             // (never a third point: two kept vertices imply d0 d1 >= 0)
    ++n;
  }
  return n;
}

__device__ __forceinline__ int hull_hull(const Body& b1, const Body& b2, double eps, Pt& out0, Pt& out1) {  // contacts.py:142-201
  const Sep c1 = test_separations(b1, b2, eps);
  if (c1.dist > eps) return 0;
  const Sep c2 = test_separations(b2, b1, eps);
  if (c2.dist > eps) return 0;
  const bool ref_is_b2 = c2.dist > c1.dist;
  const Body& ref = ref_is_b2 ? b2 : b1;
  const Body& inc = ref_is_b2 ? b1 : b2;
  const Sep& c = ref_is_b2 ? c2 : c1;
  const V2 nrm = -c.normal;
  const double half_edge = c.edge_norm / 2;
  const int ie = incident_edge(nrm, inc, c.vertex);
  const V2 iv0 = inc.verts[ie] + inc.pos - ref.pos;
  const V2 iv1 = inc.verts[(ie + 1) % inc.nv] + inc.pos - ref.pos;
  const V2 plane = left_orth(nrm);
  V2 a0, a1, q0, q1;
  const int n1 = clip(iv0, iv1, plane, half_edge, a0, a1);
  if (n1 < 2) return 0;
  const int n2 = clip(a0, a1, -plane, half_edge, q0, q1);
  int n = 0;
  const V2 refv = ref.verts[c.edge];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const V2 cq = q == 0 ? q0 : q1;
    const double dist = dot(nrm, cq - refv);
    if (q < n2 && dist <= eps) {
      const V2 pt1 = cq + nrm * -dist;
      const V2 pt2 = pt1 + ref.pos - inc.pos;
      Pt r;
      if (ref_is_b2) { r.n = nrm; r.p1 = pt2; r.p2 = pt1; }                        // contacts.py:170-175
      else { r.n = -nrm; r.p1 = pt1; r.p2 = pt2; }                                 // contacts.py:198-201
      r.pen = -dist;
      if (n == 0) out0 = r; else out1 = r;
      ++n;
    }
  }
  return n;
}

__device__ __forceinline__ int collide_pair(const Body& b1, const Body& b2, double eps, Pt& out0, Pt& out1) {   // contacts.py:57-205
  const bool c1 = b1.kind == 0, c2 = b2.kind == 0;
  if (c1 && c2) return circle_circle(b1, b2, eps, out0);
  if (c1) return circle_hull(b1, b2, eps, false, out0);
  if (c2) return circle_hull(b2, b1, eps, true, out0);
  return hull_hull(b1, b2, eps, out0, out1);
}

// One launch = the whole position update of World.step_dt (world.py:88-101) for every scene: try the step with
// the current dt (Body.move), detect contacts, accept when no contact penetrates by more than `tol`, otherwise
// halve dt, go back to the start pose and retry.  The loop is per scene and runs on the device, so no host round trip.
// LPS = lanes per scene: 64 (one wave per scene, the pairs are walked 64 at a time) or 16 (four scenes per wave when a
// scene has at most 16 body pairs, i.e. nb <= 6 - the common case; the lanes of a scene are one DPP/shuffle row).
// Scenes that share a wave advance in lock step: a scene whose step is accepted keeps recomputing the same accepted
// trial (identical stores) until its neighbours are done.
template <int LPS>
__global__ void __launch_bounds__(64) lcp_move_find_contacts_kernel(ContactArgs P) {
  constexpr int SPW = 64 / LPS;                                 // scenes per wave
  constexpr int NBMAX = (LPS == 64) ? MAXB : 6;
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

// ---------------------------------------------------------------- backward of the contact frame (circle / circle)
// The reference builds the contact tuple with differentiable torch operations (DiffContactHandler, contacts.py:57-205), so
// a loss after a roll-out reaches the poses through (normal, p1, p2).  This is that chain rule for the circle / circle
// record of contacts.py:68-79, with d = pos1 - pos2, dist = |d|, n = d / dist, r = rad1 + rad2:
//     normal = n        p1 = -n (rad1 - (r - dist) / 2)        p2 = n (rad2 - (r - dist) / 2)
//     d(loss)/dd = (I - n n^T) / dist . (g_n - a1 g_p1 + a2 g_p2)  +  n (n . (g_p2 - g_p1)) / 2,   a_i = rad_i - (r - dist) / 2
// and d(loss)/dpos1 = +that, d(loss)/dpos2 = -that (the rotations do not enter).  Contacts involving a hull are treated as
// constants of the step (their frame backward is not implemented: gradients stop there).  One thread per scene walks the
// scene's contact list in order: fixed summation order, no atomics.
__global__ void __launch_bounds__(64) lcp_contact_frame_backward_kernel(int B, int nb, int maxc, const int32_t* kind, const double* radius,
                                                                        const double* p, const int32_t* c_i1, const int32_t* c_i2,
                                                                        const int32_t* count, const float* g_n, const float* g_p1,
                                                                        const float* g_p2, double* dp) {
  const int scene = blockIdx.x * 64 + threadIdx.x;
  if (scene >= B) return;
  double* out = dp + (size_t)scene * nb * 3;
  for (int i = 0; i < nb * 3; ++i) out[i] = 0.0;
  int n = count[scene];
  n = n < 0 ? 0 : (n > maxc ? maxc : n);
  for (int c = 0; c < n; ++c) {
    const size_t o = (size_t)scene * maxc + c;
    const int i1 = c_i1[o], i2 = c_i2[o];
    if (kind[(size_t)scene * nb + i1] != 0 || kind[(size_t)scene * nb + i2] != 0) continue;
    const double* q1 = p + ((size_t)scene * nb + i1) * 3;
    const double* q2 = p + ((size_t)scene * nb + i2) * 3;
    const double r1 = radius[(size_t)scene * nb + i1], r2 = radius[(size_t)scene * nb + i2];
    const double dx = q1[1] - q2[1], dy = q1[2] - q2[2];
    const double dist = sqrt(dx * dx + dy * dy), inv = 1.0 / dist;
    const double nx = dx * inv, ny = dy * inv;
    const double half = 0.5 * ((r1 + r2) - dist);
    const double a1 = r1 - half, a2 = r2 - half;
    const double gnx = g_n[o * 2], gny = g_n[o * 2 + 1], g1x = g_p1[o * 2], g1y = g_p1[o * 2 + 1], g2x = g_p2[o * 2], g2y = g_p2[o * 2 + 1];
    const double wx = gnx - a1 * g1x + a2 * g2x, wy = gny - a1 * g1y + a2 * g2y;
    const double wn = wx * nx + wy * ny;
    const double k = 0.5 * (nx * (g2x - g1x) + ny * (g2y - g1y));
    const double gx = (wx - nx * wn) * inv + nx * k, gy = (wy - ny * wn) * inv + ny * k;
    out[i1 * 3 + 1] += gx; out[i1 * 3 + 2] += gy;
    out[i2 * 3 + 1] -= gx; out[i2 * 3 + 2] -= gy;
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

}  // namespace ct

int joint_jacobian_launch(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2, const double* jr1,
                          double* jrot1, const double* p, const float* v, const double* dt_scene, double dt, double vscale, float* Je,
                          void* stream) {
  hipLaunchKernelGGL(ct::lcp_joint_jacobian_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, nb, nj, e, jtype, jb1, jb2,
                     jr1, jrot1, p, v, dt_scene, dt, vscale, Je);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int contact_frame_backward_launch(int B, int nb, int maxc, const int32_t* kind, const double* radius, const double* p,
                                  const int32_t* c_i1, const int32_t* c_i2, const int32_t* count, const float* g_n,
                                  const float* g_p1, const float* g_p2, double* dp, void* stream) {
  hipLaunchKernelGGL(ct::lcp_contact_frame_backward_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, nb, maxc, kind,
                     radius, p, c_i1, c_i2, count, g_n, g_p1, g_p2, dp);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int contacts_launch(const ContactArgs& P, void* stream) {
  if (P.nb > ct::MAXB) return LCP_E_TOOLARGE;
  if (P.nb <= 6) hipLaunchKernelGGL(ct::lcp_move_find_contacts_kernel<16>, dim3((P.B + 3) / 4), dim3(64), 0, (hipStream_t)stream, P);
  else hipLaunchKernelGGL(ct::lcp_move_find_contacts_kernel<64>, dim3(P.B), dim3(64), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
