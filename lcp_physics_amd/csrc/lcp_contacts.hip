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
};

struct Pt { V2 n, p1, p2; double pen; };

// contacts.py:207-217 (`>=`: last maximiser wins)
__device__ __forceinline__ int support(const V2* pts, int n, V2 dir) {
  int best = -1; double bn = -1.0;
  for (int i = 0; i < n; ++i) { const double c = dot(pts[i], dir); if (c >= bn) { bn = c; best = i; } }
  return best;
}

__device__ __forceinline__ int circle_circle(const Body& b1, const Body& b2, double eps, Pt* out) {   // contacts.py:68-79
  const double r = b1.rad + b2.rad;
  V2 n = b1.pos - b2.pos;
  const double dist = norm(n);
  const double pen = r - dist;
  if (pen < -eps) return 0;
  n = n * (1.0 / dist);
  out[0].n = n; out[0].p1 = -n * (b1.rad - pen / 2); out[0].p2 = n * (b2.rad - pen / 2); out[0].pen = pen;
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

// contacts.py:295-330; simplex of 1..3 points, returns closest point and the ids used
__device__ __forceinline__ V2 closest(V2 p, const V2* sx, int ns, int* ids, int& nid) {
  if (ns == 1) { ids[0] = 0; nid = 1; return sx[0]; }
  if (ns == 2) {
    double u, v; bary2(p, sx[0], sx[1], u, v);
    if (u <= 0) { ids[0] = 1; nid = 1; return sx[1]; }
    if (v <= 0) { ids[0] = 0; nid = 1; return sx[0]; }
    ids[0] = 0; ids[1] = 1; nid = 2; return u * sx[0] + v * sx[1];
  }
  double uAB, vAB, uBC, vBC, uCA, vCA, uABC, vABC, wABC;
  bary2(p, sx[0], sx[1], uAB, vAB); bary2(p, sx[1], sx[2], uBC, vBC); bary2(p, sx[2], sx[0], uCA, vCA);
  bary3(p, sx[0], sx[1], sx[2], uABC, vABC, wABC);
  if (vAB <= 0 && uCA <= 0) { ids[0] = 0; nid = 1; return sx[0]; }
  if (vBC <= 0 && uAB <= 0) { ids[0] = 1; nid = 1; return sx[1]; }
  if (vCA <= 0 && uBC <= 0) { ids[0] = 2; nid = 1; return sx[2]; }
  if (uAB > 0 && vAB > 0 && wABC <= 0) { ids[0] = 0; ids[1] = 1; nid = 2; return uAB * sx[0] + vAB * sx[1]; }
  if (uBC > 0 && vBC > 0 && uABC <= 0) { ids[0] = 1; ids[1] = 2; nid = 2; return uBC * sx[1] + vBC * sx[2]; }
  if (uCA > 0 && vCA > 0 && vABC <= 0) { ids[0] = 2; ids[1] = 0; nid = 2; return uCA * sx[2] + vCA * sx[0]; }
  ids[0] = 0; ids[1] = 1; ids[2] = 2; nid = 3; return p;      // inside (the reference raises if nothing matched)
}

// contacts.py:80-141: `circ` plays b1, `hull` b2
__device__ __forceinline__ int circle_hull(const Body& circ, const Body& hull, double eps, bool circle_is_g2, Pt* out) {
  const V2* verts = hull.verts;
  const int nv = hull.nv;
  const V2 tp = circ.pos - hull.pos;
  V2 sx[3]; int sid[3];                        // simplex points and their vertex indices
  sx[0] = verts[0]; sid[0] = 0;
  int ns = 1, ids[3], nid = 1;
  V2 cl = sx[0];
  for (int iter = 0; iter < 4 * NV; ++iter) {
    cl = closest(tp, sx, ns, ids, nid);
    if (nid == 3) break;
    V2 sd;
    if (nid == 2) {
      sd = left_orth(sx[ids[0]] - sx[ids[1]]);
      if (dot(sd, tp - sx[ids[0]]) < 0) sd = -sd;
    } else {
      sd = tp - cl;
    }
    if (sd.x == 0 && sd.y == 0) break;
    const int si = support(verts, nv, sd);
    bool in_simplex = false;
    for (int q = 0; q < ns; ++q) in_simplex = in_simplex || (sid[q] == si);
    if (in_simplex) break;
    V2 nsx[3]; int nsid[3];
    for (int q = 0; q < nid; ++q) { nsx[q] = sx[ids[q]]; nsid[q] = sid[ids[q]]; }
    nsx[nid] = verts[si]; nsid[nid] = si;
    ns = nid + 1;
    for (int q = 0; q < ns; ++q) { sx[q] = nsx[q]; sid[q] = nsid[q]; }
  }
  V2 bn, bp1, bp2; double bd;
  if (nid < 3) {
    bp2 = cl;
    const V2 cw = cl + hull.pos;
    bp1 = cw - circ.pos;
    bd = norm(cw - circ.pos) - circ.rad;
    if (bd > eps) return 0;
    bn = -bp1 * (1.0 / norm(bp1));
  } else {                                     // centre inside the hull: SAT, contacts.py:114-137
    bd = -1e10; bn = v2(0, 0); bp1 = bn; bp2 = bn;
    for (int idx = 0; idx < nv; ++idx) {
      const V2 edge = verts[(idx + 1) % nv] - verts[idx];
      const V2 nrm = left_orth(edge) * (1.0 / norm(edge));
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
  out[0].n = bn; out[0].p1 = bp1; out[0].p2 = bp2; out[0].pen = -bd;
  return 1;
}

struct Sep { double dist; V2 normal; int vertex; double edge_norm; int edge; };

__device__ __forceinline__ Sep test_separations(const Body& h1, const Body& h2, double eps) {      // contacts.py:220-250
  Sep best; best.dist = -1e10; best.normal = v2(0, 0); best.vertex = -1; best.edge_norm = 0; best.edge = 0;
  for (int idx = 0; idx < h1.nv; ++idx) {
    const V2 edge = h1.verts[(idx + 1) % h1.nv] - h1.verts[idx];
    const double en = norm(edge);
    const V2 nrm = left_orth(edge) * (1.0 / en);
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
    const V2 edge = inc.verts[(i + 1) % inc.nv] - inc.verts[i];
    const V2 inrm = left_orth(edge) * (1.0 / norm(edge));
    const double d = dot(ref_normal, inrm);
    if (d < min_dot) { min_dot = d; best = i; }
  }
  return best;
}

__device__ __forceinline__ int clip(const V2* in, V2 nrm, double offset, V2* out) {                 // contacts.py:270-292
  int n = 0;
  const double d0 = dot(nrm, in[0]) + offset, d1 = dot(nrm, in[1]) + offset;
  if (d0 >= 0.0) out[n++] = in[0];
  if (d1 >= 0.0) out[n++] = in[1];
  if (d0 * d1 < 0.0 || n < 2) {
    const double interp = d0 / (d0 - d1);
    out[n++] = in[0] + interp * (in[1] - in[0]);
  }
  return n;
}

__device__ __forceinline__ int hull_hull(const Body& b1, const Body& b2, double eps, Pt* out) {       // contacts.py:142-201
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
  V2 iv[2];
  iv[0] = inc.verts[ie] + inc.pos - ref.pos;
  iv[1] = inc.verts[(ie + 1) % inc.nv] + inc.pos - ref.pos;
  const V2 plane = left_orth(nrm);
  V2 cl1[3], cl2[3];
  const int n1 = clip(iv, plane, half_edge, cl1);
  if (n1 < 2) return 0;
  const int n2 = clip(cl1, -plane, half_edge, cl2);
  int n = 0;
  for (int q = 0; q < n2 && n < 2; ++q) {
    const double dist = dot(nrm, cl2[q] - ref.verts[c.edge]);
    if (dist <= eps) {
      const V2 pt1 = cl2[q] + nrm * -dist;
      const V2 pt2 = pt1 + ref.pos - inc.pos;
      if (ref_is_b2) { out[n].n = nrm; out[n].p1 = pt2; out[n].p2 = pt1; }          // contacts.py:170-175
      else { out[n].n = -nrm; out[n].p1 = pt1; out[n].p2 = pt2; }                   // contacts.py:198-201
      out[n].pen = -dist;
      ++n;
    }
  }
  return n;
}

__device__ __forceinline__ int collide_pair(const Body& b1, const Body& b2, double eps, Pt* out) {   // contacts.py:57-205
  const bool c1 = b1.kind == 0, c2 = b2.kind == 0;
  if (c1 && c2) return circle_circle(b1, b2, eps, out);
  if (c1) return circle_hull(b1, b2, eps, false, out);
  if (c2) return circle_hull(b2, b1, eps, true, out);
  return hull_hull(b1, b2, eps, out);
}

// One launch = the whole position update of World.step_dt (world.py:88-101) for every scene: try the step with
// the current dt (Body.move), detect contacts, accept when no contact penetrates by more than `tol`, otherwise
// halve dt, go back to the start pose and retry.  The loop is per scene (one wave), so no host round trip.
__global__ void __launch_bounds__(64) lcp_move_find_contacts_kernel(ContactArgs P) {
  __shared__ V2 s_verts[MAXB * NV];
  __shared__ double s_pose[MAXB * 3];
  const int scene = blockIdx.x, lane = threadIdx.x;
  const int nb = P.nb;
  const int npairs = nb * (nb - 1) / 2;
  double dt = P.dt;
  int base = 0, trial = 0;
  double maxpen = -1e300;
  for (;;) {
    // bodies.py:80-82 (p <- p_start + v dt) and the vertex rotation of bodies.py:211-214
    for (int idx = lane; idx < nb * 3; idx += 64) {
      double pv = P.p_start[(size_t)scene * nb * 3 + idx];
      if (P.v) pv += (double)P.v[(size_t)scene * nb * 3 + idx] * dt;
      s_pose[idx] = pv;
    }
    __syncthreads();
    for (int idx = lane; idx < nb * NV; idx += 64) {
      const int bdy = idx / NV, k = idx - bdy * NV;
      const double rot = s_pose[bdy * 3];
      const double sn = sin(rot), cs = cos(rot);
      const double* vl = P.verts_local + ((size_t)scene * nb + bdy) * NV * 2 + k * 2;
      const double lx = vl[0], ly = vl[1];
      s_verts[idx] = v2(cs * lx - sn * ly, sn * lx + cs * ly);                      // utils.py:105-112
    }
    __syncthreads();
    base = 0; maxpen = -1e300;
    for (int p0 = 0; p0 < npairs; p0 += 64) {
      const int pr = p0 + lane;
      int cnt = 0, bi = 0, bj = 1;
      Pt pts[2];
      if (pr < npairs) {
        int rem = pr;                                           // pair index -> (i, j), i < j, lexicographic
        while (rem >= nb - 1 - bi) { rem -= nb - 1 - bi; ++bi; }
        bj = bi + 1 + rem;
        const bool skip = P.no_contact && P.no_contact[((size_t)scene * nb + bi) * nb + bj];
        if (!skip) {
          Body b1, b2;
          b1.kind = P.kind[(size_t)scene * nb + bi]; b2.kind = P.kind[(size_t)scene * nb + bj];
          b1.pos = v2(s_pose[bi * 3 + 1], s_pose[bi * 3 + 2]); b2.pos = v2(s_pose[bj * 3 + 1], s_pose[bj * 3 + 2]);
          b1.rad = P.radius[(size_t)scene * nb + bi]; b2.rad = P.radius[(size_t)scene * nb + bj];
          b1.nv = P.nverts[(size_t)scene * nb + bi]; b2.nv = P.nverts[(size_t)scene * nb + bj];
          b1.verts = s_verts + bi * NV; b2.verts = s_verts + bj * NV;
          cnt = collide_pair(b1, b2, P.eps, pts);
        }
      }
      // exclusive prefix sum of cnt over the lanes (pair order = the reference's contact order)
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
      const int excl = incl - cnt;
      const int total = __shfl(incl, 63, 64);
      for (int q = 0; q < cnt; ++q) {
        const int slot = base + excl + q;
        if (slot < P.maxc) {
          const size_t o = (size_t)scene * P.maxc + slot;
          P.c_n[o * 2] = (float)pts[q].n.x; P.c_n[o * 2 + 1] = (float)pts[q].n.y;
          P.c_p1[o * 2] = (float)pts[q].p1.x; P.c_p1[o * 2 + 1] = (float)pts[q].p1.y;
          P.c_p2[o * 2] = (float)pts[q].p2.x; P.c_p2[o * 2 + 1] = (float)pts[q].p2.y;
          if (P.c_pen) P.c_pen[o] = pts[q].pen;
          P.c_i1[o] = bi; P.c_i2[o] = bj;
        }
        maxpen = pts[q].pen > maxpen ? pts[q].pen : maxpen;
      }
      base += total;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(maxpen, off, 64); maxpen = o > maxpen ? o : maxpen; }
    ++trial;
    // world.py:95-101
    const bool ok = !(base > 0 && maxpen > P.tol);
    if (ok) break;
    if (!P.strict && dt < P.dt_floor) break;
    if (trial >= P.max_trials || !P.v) break;               // (the reference would not terminate here)
    dt *= 0.5;
    __syncthreads();
  }
  // pad the unused contact slots with a harmless record (no normal, bodies 0/0)
  const int nfill = base < P.maxc ? base : P.maxc;
  for (int slot = nfill + lane; slot < P.maxc; slot += 64) {
    const size_t o = (size_t)scene * P.maxc + slot;
    P.c_n[o * 2] = 0; P.c_n[o * 2 + 1] = 0; P.c_p1[o * 2] = 0; P.c_p1[o * 2 + 1] = 0; P.c_p2[o * 2] = 0; P.c_p2[o * 2 + 1] = 0;
    if (P.c_pen) P.c_pen[o] = 0;
    P.c_i1[o] = 0; P.c_i2[o] = 0;
  }
  if (P.p_out) for (int idx = lane; idx < nb * 3; idx += 64) P.p_out[(size_t)scene * nb * 3 + idx] = s_pose[idx];
  if (lane == 0) {
    P.count[scene] = base;                                  // may exceed maxc: the caller checks
    if (P.max_pen) P.max_pen[scene] = base > 0 ? maxpen : 0.0;
    if (P.dt_used) P.dt_used[scene] = dt;
    if (P.t) P.t[scene] += dt;                              // world.py:122
    if (P.trials) P.trials[scene] = trial;
  }
}

}  // namespace ct

int contacts_launch(const ContactArgs& P, void* stream) {
  if (P.nb > ct::MAXB) return LCP_E_TOOLARGE;
  hipLaunchKernelGGL(ct::lcp_move_find_contacts_kernel, dim3(P.B), dim3(64), 0, (hipStream_t)stream, P);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
