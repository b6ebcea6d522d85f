// lcp_solo.hip - the contact-list step of SMALL batches: ONE scene per wavefront, the body-space PDIPM of lcp_quad.hip (ALG = 2)
// spread over the four 16-lane DPP rows of the wave.
//
// Why: lcp_fwd_quad puts four scenes on a wavefront, and a scene's solve is a dependent chain of ~0.095 ms whatever the batch - at
// BASELINE configs[1] (1024 scenes) that is 256 wavefronts on a chip with 1024 SIMDs, three quarters of it idle.  Here the batch
// is 1024 wavefronts, and the 64 lanes of each work on one scene:
//   * m-space (the 4 nc inequality rows): lane (r, c) = COMPONENT r (normal, friction +, friction -, cone) of CONTACT c - DPP row r,
//     lane c of the row.  Every element-wise statement of pdipm.py (residuals, s / z, step lengths, updates) is one instruction
//     instead of four; the per-contact coupling (F z, the closed-form 4 x 4 block inverse M^-1) first gathers the contact's four
//     components into every row (gather4: v_permlane16_swap + v_permlane32_swap, gfx950) and then runs redundantly in the rows.
//   * x-space (nz <= 16) and the matrix Q + G^T M^-1 G: replicated in the four rows, lane j of a row = entry / matrix row j, exactly
//     the layout of lcp_quad.hip - the pivot-free LU and the triangular sweeps are its code (`row_newbcast` broadcasts,
//     v_fmac_f64_dpp blocks), executed identically by the four rows.
//   * formation: DPP row r accumulates COLUMNS 4 r .. 4 r + 3 of the matrix (a quarter of the FMAs), the four quarters are
//     exchanged with gather4; J v is ONE product for all four components (the lane holds its component's row of G); J^T w is
//     one product per row (its component's weights) followed by a sum over the rows.
//   * reductions over m-space: DPP row reduction, then two swap stages over the rows.
// Same equations, same exits (pdipm.py:49-186 per scene, best iterate, three strikes), same workspace fields as lcp_fwd_quad with
// ALG = 2: the backward kernels of lcp_quad.hip follow either forward.  Like ALG = 2 it serves scenes whose equality rows pin the
// leading coordinates (A = [I 0], b = 0: the TotalConstraint on the floor of the demo worlds) or that have none; another scene is
// marked (meta[21]) and left to lcp_fwd_quad<..., ALG = 1>, launched behind.
// Contact-list inputs, fp32 I/O, fp64 arithmetic, nz <= 16, <= 16 contacts, neq <= 4.
#include "lcp_quad_prims.h"

// register-resident Jacobian entries are read through an opaque statement (their conversions to fp64 must not be hoisted out of the PDIPM
// loop) - in place, not on a by-value copy, which cost a v_mov_b32 per use (LCP_Q_LAUNDER_IN_PLACE in lcp_quad_kernels.inc)
#define LCP_SOLO_OPAQUE(x) asm volatile("" : "+v"(x))
#ifndef LCP_SOLO_PEEL_INIT
#define LCP_SOLO_PEEL_INIT 1     // the initialisation pass (it = -1) as its own copy of the loop body (0: one loop, A/B)
#endif
#ifndef LCP_SOLO_RCP_STEP
#define LCP_SOLO_RCP_STEP 1      // step lengths, d = z / s, rs / s through one reciprocal per z_i, s_i (LCP_Q_RCP_STEP; 0: IEEE quotients, A/B)
#endif
#ifndef LCP_SOLO_UNROLL_PASS
#define LCP_SOLO_UNROLL_PASS 1   // the two KKT solves of an iteration as two copies of the code instead of a two-trip loop (0: the loop, A/B)
#endif

namespace lcp {
namespace solo {

using namespace w64;
using namespace q16;

// ---------------------------------------------------------------- moves between the four DPP rows (same lane of the row)
struct U2 { uint32_t a, b; };
__device__ __forceinline__ U2 swap16(uint32_t x, uint32_t y) {     // odd rows of x <-> even rows of y
  auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  return U2{r[0], r[1]};
}
__device__ __forceinline__ U2 swap32(uint32_t x, uint32_t y) {     // rows 2, 3 of x <-> rows 0, 1 of y
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  return U2{r[0], r[1]};
}
__device__ __forceinline__ void swap16(double v, double& even, double& odd) {   // even = [v0 v0 v2 v2], odd = [v1 v1 v3 v3] (by row)
  const U2 lo = swap16((uint32_t)__double2loint(v), (uint32_t)__double2loint(v));
  const U2 hi = swap16((uint32_t)__double2hiint(v), (uint32_t)__double2hiint(v));
  even = __hiloint2double((int)hi.a, (int)lo.a); odd = __hiloint2double((int)hi.b, (int)lo.b);
}
__device__ __forceinline__ void swap32(double v, double& low, double& high) {   // low = [v0 v1 v0 v1], high = [v2 v3 v2 v3]
  const U2 lo = swap32((uint32_t)__double2loint(v), (uint32_t)__double2loint(v));
  const U2 hi = swap32((uint32_t)__double2hiint(v), (uint32_t)__double2hiint(v));
  low = __hiloint2double((int)hi.a, (int)lo.a); high = __hiloint2double((int)hi.b, (int)lo.b);
}
// the values the four rows hold at this lane of the row, in every row: (row 0, row 1, row 2, row 3) = (n, f1, f2, g)
__device__ __forceinline__ M4<double> gather4(double v) {
  double ev, od, r0, r2, r1, r3;
  swap16(v, ev, od);
  swap32(ev, r0, r2);
  swap32(od, r1, r3);
  return m4<double>(r0, r1, r2, r3);
}
__device__ __forceinline__ double rows_sum(double v) {
  double a, b; swap32(v, a, b); v = a + b;
  swap16(v, a, b); return a + b;
}
__device__ __forceinline__ double rows_fmax(double v) {
  double a, b; swap32(v, a, b); v = fmax_(a, b);
  swap16(v, a, b); return fmax_(a, b);
}
__device__ __forceinline__ double rows_fmin(double v) {
  double a, b; swap32(v, a, b); v = fmin_(a, b);
  swap16(v, a, b); return fmin_(a, b);
}
__device__ __forceinline__ uint32_t rows_or(uint32_t k) {
  U2 t = swap32(k, k); k = t.a | t.b;
  t = swap16(k, k); return t.a | t.b;
}
__device__ __forceinline__ uint32_t row_or(uint32_t k) {
  k |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0xB1, 0xf, 0xf, true);
  k |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x4E, 0xf, 0xf, true);
  k |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x141, 0xf, 0xf, true);
  k |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)k, 0x140, 0xf, 0xf, true);
  return k;
}
// Four sums over all 64 lanes at once, results everywhere: the rows are folded FIRST (two swap stages without copies leave DPP row
// 0 / 1 / 2 / 3 with the row-folded values of v1 / v3 / v2 / v4), ONE DPP row reduction then serves all four, gather4 hands the
// totals round.  33 instructions instead of 4 x 22.
__device__ __forceinline__ void swap32_pair(double& a, double& b) {     // a = [a0 a1 b0 b1], b = [a2 a3 b2 b3] (by row)
  const U2 lo = swap32((uint32_t)__double2loint(a), (uint32_t)__double2loint(b));
  const U2 hi = swap32((uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b));
  a = __hiloint2double((int)hi.a, (int)lo.a); b = __hiloint2double((int)hi.b, (int)lo.b);
}
__device__ __forceinline__ void swap16_pair(double& a, double& b) {     // a = [a0 b0 a2 b2], b = [a1 b1 a3 b3]
  const U2 lo = swap16((uint32_t)__double2loint(a), (uint32_t)__double2loint(b));
  const U2 hi = swap16((uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b));
  a = __hiloint2double((int)hi.a, (int)lo.a); b = __hiloint2double((int)hi.b, (int)lo.b);
}
__device__ __forceinline__ void wave_sum4(double& v1, double& v2, double& v3, double& v4) {
  swap32_pair(v1, v2); swap32_pair(v3, v4);
  double t12 = v1 + v2, t34 = v3 + v4;          // rows: [v1(0+2) v1(1+3) v2(0+2) v2(1+3)], [v3 .. v4 ..]
  swap16_pair(t12, t34);
  const double u = row_sum(t12 + t34);          // rows: [V1 V3 V2 V4]
  const M4<double> g = gather4(u);
  v1 = g.n; v3 = g.f1; v2 = g.f2; v4 = g.g;
}
// two maxima over all 64 lanes at once (NaN-skipping v_max), results everywhere
__device__ __forceinline__ void wave_fmax2(double& a, double& b) {
  swap32_pair(a, b);
  double t = fmax_(a, b), ev, od;               // rows: [a(0,2) a(1,3) b(0,2) b(1,3)]
  swap16(t, ev, od);
  t = row_fmax(fmax_(ev, od));                  // rows: [A A B B]
  swap32(t, a, b);
}
// over all 64 lanes, result everywhere
__device__ __forceinline__ double wave_sum(double v) { return rows_sum(row_sum(v)); }
__device__ __forceinline__ double wave_fmax(double v) { return rows_fmax(row_fmax(v)); }
__device__ __forceinline__ double wave_fmin(double v) { return rows_fmin(row_fmin(v)); }
__device__ __forceinline__ uint32_t wave_or(uint32_t k) { return rows_or(row_or(k)); }

// acc0 / acc1 += sum over the 16 lanes k of the row of (lane k's `src`) * coef[k]  (even k -> acc0, odd k -> acc1): J v with the lane's
// row of G as coefficients, J^T w with its column.  Same rules as the blocks of lcp_quad_prims.h: `s_nop 1` first, DPP source read-only.
#define LCP_SOLO_DOT(ACC, K) "v_fmac_f64_dpp %[" #ACC "], %[v], %[c" #K "] row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void dot16_dpp(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t"
      LCP_SOLO_DOT(a0, 0) LCP_SOLO_DOT(a1, 1) LCP_SOLO_DOT(a0, 2) LCP_SOLO_DOT(a1, 3) LCP_SOLO_DOT(a0, 4) LCP_SOLO_DOT(a1, 5)
      LCP_SOLO_DOT(a0, 6) LCP_SOLO_DOT(a1, 7) LCP_SOLO_DOT(a0, 8) LCP_SOLO_DOT(a1, 9) LCP_SOLO_DOT(a0, 10) LCP_SOLO_DOT(a1, 11)
      LCP_SOLO_DOT(a0, 12) LCP_SOLO_DOT(a1, 13) LCP_SOLO_DOT(a0, 14) LCP_SOLO_DOT(a1, 15)
      : [a0] "+v"(a0), [a1] "+v"(a1)
      : [v] "v"(v), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]),
        [c7] "v"(c[7]), [c8] "v"(c[8]), [c9] "v"(c[9]), [c10] "v"(c[10]), [c11] "v"(c[11]), [c12] "v"(c[12]), [c13] "v"(c[13]),
        [c14] "v"(c[14]), [c15] "v"(c[15]));
}
// formation, one contact K: the lane's four matrix columns += a * (lane K's p0) + b * (lane K's p1)
#define LCP_SOLO_FORM(X, P, M, K) "v_fmac_f64_dpp %[" #X "], %[" #P "], %[" #M "] row_newbcast:%[" #K "] row_mask:0xf bank_mask:0xf\n\t"
template <int K> __device__ __forceinline__ void form4(double (&x)[4], const double (&p0)[4], const double (&p1)[4], double a, double b) {
  asm("s_nop 1\n\t"
      LCP_SOLO_FORM(x0, p0, a, k) LCP_SOLO_FORM(x1, p1, a, k) LCP_SOLO_FORM(x2, p2, a, k) LCP_SOLO_FORM(x3, p3, a, k)
      LCP_SOLO_FORM(x0, q0, b, k) LCP_SOLO_FORM(x1, q1, b, k) LCP_SOLO_FORM(x2, q2, b, k) LCP_SOLO_FORM(x3, q3, b, k)
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3])
      : [p0] "v"(p0[0]), [p1] "v"(p0[1]), [p2] "v"(p0[2]), [p3] "v"(p0[3]), [q0] "v"(p1[0]), [q1] "v"(p1[1]), [q2] "v"(p1[2]), [q3] "v"(p1[3]),
        [a] "v"(a), [b] "v"(b), [k] "n"(K));
}

// four contacts K0 .. K0 + 3 in one statement (one s_nop for 32 instructions)
#define LCP_SOLO_FORM1(A, B, KK) \
  LCP_SOLO_FORM(x0, p0, A, KK) LCP_SOLO_FORM(x1, p1, A, KK) LCP_SOLO_FORM(x2, p2, A, KK) LCP_SOLO_FORM(x3, p3, A, KK) \
  LCP_SOLO_FORM(x0, q0, B, KK) LCP_SOLO_FORM(x1, q1, B, KK) LCP_SOLO_FORM(x2, q2, B, KK) LCP_SOLO_FORM(x3, q3, B, KK)
template <int K0> __device__ __forceinline__ void form4x4(double (&x)[4], const double (&p0)[4], const double (&p1)[4], const double (&a)[4],
                                                         const double (&b)[4]) {
  asm("s_nop 1\n\t"
      LCP_SOLO_FORM1(a0, b0, k0) LCP_SOLO_FORM1(a1, b1, k1) LCP_SOLO_FORM1(a2, b2, k2) LCP_SOLO_FORM1(a3, b3, k3)
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3])
      : [p0] "v"(p0[0]), [p1] "v"(p0[1]), [p2] "v"(p0[2]), [p3] "v"(p0[3]), [q0] "v"(p1[0]), [q1] "v"(p1[1]), [q2] "v"(p1[2]), [q3] "v"(p1[3]),
        [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b0] "v"(b[0]), [b1] "v"(b[1]), [b2] "v"(b[2]), [b3] "v"(b[3]),
        [k0] "n"(K0), [k1] "n"(K0 + 1), [k2] "n"(K0 + 2), [k3] "n"(K0 + 3));
}

// ---- generated: the same two blocks over a WINDOW (compile-time sizes: columns neq .. nz-1 of x-space, contacts 0 .. nc-1)
template <int LO, int N> struct DotW;
template <int K0, int NS> struct FormW;
template <int LO> struct DotW<LO, 1> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0)); } };
template <int LO> struct DotW<LO, 2> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1)); } };
template <int LO> struct DotW<LO, 3> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2)); } };
template <int LO> struct DotW<LO, 4> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3)); } };
template <int LO> struct DotW<LO, 5> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4)); } };
template <int LO> struct DotW<LO, 6> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5)); } };
template <int LO> struct DotW<LO, 7> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6)); } };
template <int LO> struct DotW<LO, 8> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7)); } };
template <int LO> struct DotW<LO, 9> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8)); } };
template <int LO> struct DotW<LO, 10> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9)); } };
template <int LO> struct DotW<LO, 11> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10)); } };
template <int LO> struct DotW<LO, 12> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c11] row_newbcast:%[k11] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10), [c11] "v"(c[LO + 11]), [k11] "n"(LO + 11)); } };
template <int LO> struct DotW<LO, 13> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c11] row_newbcast:%[k11] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c12] row_newbcast:%[k12] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10), [c11] "v"(c[LO + 11]), [k11] "n"(LO + 11), [c12] "v"(c[LO + 12]), [k12] "n"(LO + 12)); } };
template <int LO> struct DotW<LO, 14> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c11] row_newbcast:%[k11] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c12] row_newbcast:%[k12] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c13] row_newbcast:%[k13] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10), [c11] "v"(c[LO + 11]), [k11] "n"(LO + 11), [c12] "v"(c[LO + 12]), [k12] "n"(LO + 12), [c13] "v"(c[LO + 13]), [k13] "n"(LO + 13)); } };
template <int LO> struct DotW<LO, 15> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c11] row_newbcast:%[k11] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c12] row_newbcast:%[k12] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c13] row_newbcast:%[k13] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c14] row_newbcast:%[k14] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10), [c11] "v"(c[LO + 11]), [k11] "n"(LO + 11), [c12] "v"(c[LO + 12]), [k12] "n"(LO + 12), [c13] "v"(c[LO + 13]), [k13] "n"(LO + 13), [c14] "v"(c[LO + 14]), [k14] "n"(LO + 14)); } };
template <int LO> struct DotW<LO, 16> { static __device__ __forceinline__ void run(double& a0, double& a1, double v, const double (&c)[16]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c4] row_newbcast:%[k4] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c5] row_newbcast:%[k5] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c6] row_newbcast:%[k6] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c7] row_newbcast:%[k7] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c8] row_newbcast:%[k8] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c9] row_newbcast:%[k9] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c10] row_newbcast:%[k10] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c11] row_newbcast:%[k11] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c12] row_newbcast:%[k12] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c13] row_newbcast:%[k13] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a0], %[v], %[c14] row_newbcast:%[k14] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[a1], %[v], %[c15] row_newbcast:%[k15] row_mask:0xf bank_mask:0xf\n\t"
      : [a0] "+v"(a0), [a1] "+v"(a1) : [v] "v"(v), [c0] "v"(c[LO + 0]), [k0] "n"(LO + 0), [c1] "v"(c[LO + 1]), [k1] "n"(LO + 1), [c2] "v"(c[LO + 2]), [k2] "n"(LO + 2), [c3] "v"(c[LO + 3]), [k3] "n"(LO + 3), [c4] "v"(c[LO + 4]), [k4] "n"(LO + 4), [c5] "v"(c[LO + 5]), [k5] "n"(LO + 5), [c6] "v"(c[LO + 6]), [k6] "n"(LO + 6), [c7] "v"(c[LO + 7]), [k7] "n"(LO + 7), [c8] "v"(c[LO + 8]), [k8] "n"(LO + 8), [c9] "v"(c[LO + 9]), [k9] "n"(LO + 9), [c10] "v"(c[LO + 10]), [k10] "n"(LO + 10), [c11] "v"(c[LO + 11]), [k11] "n"(LO + 11), [c12] "v"(c[LO + 12]), [k12] "n"(LO + 12), [c13] "v"(c[LO + 13]), [k13] "n"(LO + 13), [c14] "v"(c[LO + 14]), [k14] "n"(LO + 14), [c15] "v"(c[LO + 15]), [k15] "n"(LO + 15)); } };
template <int K0> struct FormW<K0, 1> { static __device__ __forceinline__ void run(double (&x)[4], const double (&p0)[4], const double (&p1)[4], const double (&a)[4], const double (&b)[4]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
      : [x0] "+v"(x[0]) : [p0] "v"(p0[0]), [q0] "v"(p1[0]), [a0] "v"(a[0]), [b0] "v"(b[0]), [k0] "n"(K0 + 0), [a1] "v"(a[1]), [b1] "v"(b[1]), [k1] "n"(K0 + 1), [a2] "v"(a[2]), [b2] "v"(b[2]), [k2] "n"(K0 + 2), [a3] "v"(a[3]), [b3] "v"(b[3]), [k3] "n"(K0 + 3)); } };
template <int K0> struct FormW<K0, 2> { static __device__ __forceinline__ void run(double (&x)[4], const double (&p0)[4], const double (&p1)[4], const double (&a)[4], const double (&b)[4]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]) : [p0] "v"(p0[0]), [q0] "v"(p1[0]), [p1] "v"(p0[1]), [q1] "v"(p1[1]), [a0] "v"(a[0]), [b0] "v"(b[0]), [k0] "n"(K0 + 0), [a1] "v"(a[1]), [b1] "v"(b[1]), [k1] "n"(K0 + 1), [a2] "v"(a[2]), [b2] "v"(b[2]), [k2] "n"(K0 + 2), [a3] "v"(a[3]), [b3] "v"(b[3]), [k3] "n"(K0 + 3)); } };
template <int K0> struct FormW<K0, 3> { static __device__ __forceinline__ void run(double (&x)[4], const double (&p0)[4], const double (&p1)[4], const double (&a)[4], const double (&b)[4]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]) : [p0] "v"(p0[0]), [q0] "v"(p1[0]), [p1] "v"(p0[1]), [q1] "v"(p1[1]), [p2] "v"(p0[2]), [q2] "v"(p1[2]), [a0] "v"(a[0]), [b0] "v"(b[0]), [k0] "n"(K0 + 0), [a1] "v"(a[1]), [b1] "v"(b[1]), [k1] "n"(K0 + 1), [a2] "v"(a[2]), [b2] "v"(b[2]), [k2] "n"(K0 + 2), [a3] "v"(a[3]), [b3] "v"(b[3]), [k3] "n"(K0 + 3)); } };
template <int K0> struct FormW<K0, 4> { static __device__ __forceinline__ void run(double (&x)[4], const double (&p0)[4], const double (&p1)[4], const double (&a)[4], const double (&b)[4]) {
  asm("s_nop 1\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[p3], %[a0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[q3], %[b0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[p3], %[a1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[q3], %[b1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[p3], %[a2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[q3], %[b2] row_newbcast:%[k2] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[p0], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[p1], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[p2], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[p3], %[a3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x0], %[q0], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x1], %[q1], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x2], %[q2], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t" "v_fmac_f64_dpp %[x3], %[q3], %[b3] row_newbcast:%[k3] row_mask:0xf bank_mask:0xf\n\t"
      : [x0] "+v"(x[0]), [x1] "+v"(x[1]), [x2] "+v"(x[2]), [x3] "+v"(x[3]) : [p0] "v"(p0[0]), [q0] "v"(p1[0]), [p1] "v"(p0[1]), [q1] "v"(p1[1]), [p2] "v"(p0[2]), [q2] "v"(p1[2]), [p3] "v"(p0[3]), [q3] "v"(p1[3]), [a0] "v"(a[0]), [b0] "v"(b[0]), [k0] "n"(K0 + 0), [a1] "v"(a[1]), [b1] "v"(b[1]), [k1] "n"(K0 + 1), [a2] "v"(a[2]), [b2] "v"(b[2]), [k2] "n"(K0 + 2), [a3] "v"(a[3]), [b3] "v"(b[3]), [k3] "n"(K0 + 3)); } };
template <int LO, int N> __device__ __forceinline__ void dot_dpp(double& a0, double& a1, double v, const double (&c)[16]) { if constexpr (N > 0) DotW<LO, N>::run(a0, a1, v, c); }

// ---------------------------------------------------------------- the kernel
#ifndef LCP_SOLO_OCC
#define LCP_SOLO_OCC 1      // wavefronts per SIMD the register allocation allows (2: at most 256 unified registers, fp32 tables)
#endif
// NZF, EF: the scene's nz and neq as compile-time constants (0, 0: read from the arguments) - the guards of the LU and the sweeps
// and the lane masks of the x-space fold; lcp_quad.hip (quad_step) has the measurement
// NCF: the contact count of every scene as a compile-time constant (full lists, no per-scene counts; 0: run time).
// With compile-time sizes only the columns neq .. nz-1 exist for the forward (x_p is identically zero, see lcp_quad_kernels.inc
// factor_pq): they are dealt round-robin to the four DPP rows (column neq + 4 j + r in slot j of row r), the products run over the
// window of live columns / contacts only.
template <int NZF, int EF, int NCF = 0>
__global__ void __launch_bounds__(64, LCP_SOLO_OCC) lcp_fwd_solo(StepArgs SP, int pinned_hint) {
  constexpr bool TRIM = NZF > 0;
  constexpr int CLO = TRIM ? EF : 0, CHI = TRIM ? NZF : 16, NCOLS = CHI - CLO, NS = TRIM ? (NCOLS + 3) / 4 : 4;
  constexpr int KHI = NCF > 0 ? NCF : 16;                              // contacts the products visit
  using TI = float;
  using TC = double;
  __shared__ TI GL[NCQ * 16], GTL[NCQ * 16], AtL[EQ * 16];
  const int lane = threadIdx.x, l16 = lane & 15, comp = lane >> 4;
  const int scene = blockIdx.x;
  const int nb = SP.nb, nz = NZF > 0 ? NZF : 3 * nb, nc = NCF > 0 ? NCF : SP.nc, e = NZF > 0 ? EF : SP.e, m = 4 * nc;
  const int max_iter = SP.max_iter, lim = SP.lim;
  const TC eps = SP.eps;
  Ws<TI, TC> W(SP.ws, scene);
#ifdef LCP_SOLO_PROFILE
  const long long t_start = clock64();
#endif
  if (blockIdx.x == 0 && lane == 0 && SP.tag) *SP.tag = SP.tag_value;
  int ncs = nc, truncated = 0;
  if (NCF == 0 && SP.c_count) { const int c = SP.c_count[scene]; ncs = c < nc ? (c < 0 ? 0 : c) : nc; truncated = c > nc ? LCP_ST_TRUNCATED : 0; }
  const bool vc = (NCF == 16) ? true : (l16 < ncs);                // this lane's contact is live
  const bool c0 = comp == 0, c1 = comp == 1, c2 = comp == 2, c3 = comp == 3;
  auto pick = [&](const M4<TC>& a) -> TC { return c0 ? a.n : (c1 ? a.f1 : (c2 ? a.f2 : a.g)); };

  // ---- contact list -> rows (physics/engines.py:31-32,50-74; physics/world.py:144-234), as assemble_q of lcp_quad.hip
  const TI* Md = (const TI*)SP.Mdiag + (size_t)scene * nz;
  const TI* vv = (const TI*)SP.v + (size_t)scene * nz;
  const TI* ff = (const TI*)SP.f + (size_t)scene * nz;
  TI hrow = 0, mu_f = 0;
  // (round 6: the loads that do not depend on the contact's body indices are issued up front and without lane predicates - address
  //  clamped, value masked later; see row16 in lcp_quad_kernels.inc: a predicated load is waited for at the join of its branch, and
  //  sixteen of them in a row were sixteen round trips to memory in front of the first iteration of a 30 us kernel)
  TI jer[16];
  static_for<16>([&](auto K) LCP_INL { jer[K] = (TI)0; });
  if (e > 0) {
    const TI* jr = (const TI*)SP.Je + ((size_t)scene * e + (l16 < e ? l16 : 0)) * nz;
    static_for<16>([&](auto K_) LCP_INL { constexpr int K = K_; jer[K] = jr[K < nz ? K : 0]; });
  }
  const int jx0 = l16 < nz ? l16 : 0;
  TI md_l = Md[jx0], vv_l = vv[jx0], ff_l = ff[jx0];
  if (c0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { GL[l16 * 16 + j] = 0; GTL[l16 * 16 + j] = 0; }
    if (vc) {
      const ContactRows<TI> r = make_contact<TI>((const TI*)SP.c_n + (size_t)scene * nc * 2, (const TI*)SP.c_p1 + (size_t)scene * nc * 2,
                                                 (const TI*)SP.c_p2 + (size_t)scene * nc * 2, SP.c_i1 + (size_t)scene * nc,
                                                 SP.c_i2 + (size_t)scene * nc, (const TI*)SP.rest + (size_t)scene * nb,
                                                 (const TI*)SP.fric + (size_t)scene * nb, vv, l16);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int col = (q < 3) ? 3 * r.b1 + q : 3 * r.b2 + (q - 3);
        GL[l16 * 16 + col] = r.jn[q];
        GTL[l16 * 16 + col] = r.jf[q];
      }
      hrow = r.h; mu_f = r.mu;
    }
  }
  static_for<16>([&](auto K) LCP_INL { pin_vgpr(jer[K]); });        // (here, not sunk into the predicated block below)
  pin_vgpr(md_l); pin_vgpr(vv_l); pin_vgpr(ff_l);
  if (c0 && l16 < EQ) {
    static_for<16>([&](auto K_) LCP_INL { constexpr int k = K_; AtL[l16 * 16 + k] = (l16 < e && k < nz) ? jer[k] : (TI)0; });
  }
  // h and mu of contact l16 to the other rows (row 0 computed them)
  {
    const M4<TC> hh = gather4((TC)hrow), mm = gather4((TC)mu_f);
    hrow = (TI)hh.n; mu_f = (TI)mm.n;
  }
  __syncthreads();
  const TC mu_c = (TC)mu_f;
  // G row of (component, contact): jc | +jt | -jt | 0, and the same matrix by columns (column l16 over the contacts) for J^T w
  const TC sgn = c2 ? (TC)-1 : (c3 ? (TC)0 : (TC)1);
  const TI* gsrc = c0 ? GL : GTL;
  // (LCP_SOLO_OCC = 2 keeps them as fp32 - they are exact - and widens them at every use: 249 registers, two wavefronts per SIMD;
  //  the default keeps fp64 copies: this kernel serves batches of at most one wavefront per SIMD, where its time is the length of
  //  one wave's dependent chains and the conversions are on them)
  using TT = std::conditional_t<(LCP_SOLO_OCC >= 2), TI, TC>;
  TT Grow[16], GTc[16];
  static_for<16>([&](auto J) LCP_INL { Grow[J] = (TT)(sgn * (TC)gsrc[l16 * 16 + J]); GTc[J] = (TT)(sgn * (TC)gsrc[J * 16 + l16]); });
  // formation operands: the lane's contact at the row's four columns 4 comp + jj; the weights jc_c[l16], jt_c[l16] of matrix row l16
  // over the contacts c are read from LDS per group of four contacts
  TI jcq[4], jtq[4];
  auto colof = [&](int j) { return TRIM ? CLO + 4 * j + comp : 4 * comp + j; };     // matrix column in slot j of this DPP row
  static_for<NS>([&](auto JJ) LCP_INL {
    const int col = colof(JJ);
    const bool okc = col < CHI;
    jcq[JJ] = okc ? GL[l16 * 16 + (okc ? col : 0)] : (TI)0; jtq[JJ] = okc ? GTL[l16 * 16 + (okc ? col : 0)] : (TI)0;
  });
  const TC qd = (l16 < nz) ? (TC)md_l : (TC)0;
  const TC p = (l16 < nz) ? (TC)momentum_entry<TI>(md_l, vv_l, (TI)SP.dt, ff_l) : (TC)0;   // engines.py:32
  // F z of the contact structure (engines.py:69-73) with the contact's gathered multipliers: (F z)_comp = fn z_n + f1 (z_f1 + z_f2) + fg z_g
  const TC fzn = c3 ? mu_c : (TC)0, fz12 = c3 ? (TC)-1 : (TC)0, fzg = (c1 || c2) ? (TC)1 : (TC)0;
  const TC hn = c0 ? (TC)hrow : (TC)0;                              // h = [Jc v rbar; 0; 0; 0]

  // ---- equality rows: pinned leading coordinates (A = [I 0]) or none; anything else goes to the general kernel behind
  bool okl = true;
  static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; if (a < e) okl = okl && ((TC)AtL[a * 16 + l16] == ((l16 == a) ? (TC)1 : (TC)0)); });
  const bool pin = __all(okl) != 0;
  if (lane == 0) {
    W.meta[21] = pin ? (TC)0 : (TC)1;
    W.meta[0] = (TC)2; W.meta[18] = (TC)1; W.meta[19] = (TC)ncs;
  }
  if (!pin) {                                                       // (LCP_HINT_PINNED: nothing is launched behind - the broken promise is loud)
    if (pinned_hint && c0) {
      if (l16 < nz) ((TI*)SP.v_new)[(size_t)scene * nz + l16] = nan_of<TI>();
      if (l16 == 0 && SP.status) SP.status[scene] = LCP_ST_NAN;
    }
    return;
  }
  if (c0 && vc) W.meta[1 + l16] = mu_c;
  int status = truncated;
  if (row_any(l16 < nz && !(qd != (TC)0))) status |= LCP_ST_SINGULAR_Q;
  if (c0) W.Qit[128 + l16] = qd;                                    // (Q's diagonal: the backward takes the reciprocals itself)

#ifdef LCP_SOLO_PROFILE
  // `make soloprof`: clock64 per phase of an iteration, written over the tail of the scene's `s` output
  // (tools/gpu_phase_profile_solo.py): 0 residuals + d, 1 formation, 2 LU, 3 products before the sweeps, 4 sweeps, 5 products after,
  // 6 bookkeeping, 7 step lengths / sigma / update, 8 prologue
  long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tk = 0;
#define SOLO_TICK(i) { __builtin_amdgcn_sched_barrier(0); const long long now_ = clock64(); __builtin_amdgcn_sched_barrier(0); pc[i] += now_ - tk; tk = now_; }
#else
#define SOLO_TICK(i)
#endif
  // ---- state: x-space entry l16 (replicated in the rows), m-space scalar of (comp, contact)
  TC x = 0, y = 0, s = 1, z = 1, dinv = 1;
  TC best_resid = inf_of<TC>(), bx = 0, by = 0, bz = 1, bs = 1;
  bool have_best = false, done = false;
  int n_not = 0, iters = 0;
  const TC mf = (TC)(4 * ncs);
  TC xr[20];                                                       // matrix row l16 of Q + G^T M^-1 G, then its LU (columns 0 .. 15)
  TC udx = 1, sp[EQ];
  TC idn = 1, i1 = 1, i2 = 1, kap = 1;

  // J v for the lane's (component, contact)
  auto Gv = [&](TC v) -> TC {
    TC a0 = 0, a1 = 0, cf[16];
    static_for<16>([&](auto J) LCP_INL { if constexpr (LCP_SOLO_OCC >= 2) { LCP_SOLO_OPAQUE(Grow[J]); cf[J] = (TC)Grow[J]; } else cf[J] = Grow[J]; });
    if constexpr (TRIM) dot_dpp<CLO, NCOLS>(a0, a1, v, cf);             // (v is zero on the pinned lanes, absent beyond nz)
    else dot16_dpp(a0, a1, v, cf);
    return a0 + a1;
  };
  // J^T w: every row sums its component over the contacts, then the rows are added (result replicated)
  auto Gtw = [&](TC w) -> TC {
    TC a0 = 0, a1 = 0, cf[16];
    static_for<16>([&](auto J) LCP_INL { if constexpr (LCP_SOLO_OCC >= 2) { LCP_SOLO_OPAQUE(GTc[J]); cf[J] = (TC)GTc[J]; } else cf[J] = GTc[J]; });
    if constexpr (NCF > 0) dot_dpp<0, KHI>(a0, a1, w, cf);
    else dot16_dpp(a0, a1, w, cf);
    return rows_sum(a0 + a1);
  };
  // M^-1 t for the contact's gathered components (lcp_quad.hip minv_pq)
  auto minv = [&](const M4<TC>& t) -> M4<TC> {
    M4<TC> o;
    o.n = idn * t.n;
    o.g = kap * ((t.g - mu_c * o.n) + fma(i1, t.f1, i2 * t.f2));
    o.f1 = i1 * (t.f1 - o.g);
    o.f2 = i2 * (t.f2 - o.g);
    return o;
  };

  // factor: formation of Q + G^T M^-1 G (this row's four columns, then exchanged) and the pivot-free LU over the free coordinates
  auto factor = [&](TC di) -> bool {
    const int ll = launder(l16), nzs = NZF > 0 ? NZF : __builtin_amdgcn_readfirstlane(nz), es = NZF > 0 ? EF : __builtin_amdgcn_readfirstlane(e);
    const int ncw = NCF > 0 ? NCF : __builtin_amdgcn_readfirstlane(ncs);
    const M4<TC> D = gather4(di);
    idn = fast_rcp(D.n); i1 = fast_rcp(D.f1); i2 = fast_rcp(D.f2);
    kap = fast_rcp(D.g + (i1 + i2));
    const TC b00 = vc ? idn : (TC)0;
    const TC b10 = vc ? kap * (i1 - i2) * (mu_c * idn) : (TC)0;
    const TC b11 = vc ? kap * fma(i1 + i2, D.g, (TC)4 * (i1 * i2)) : (TC)0;
    TC p0[4], p1[4], x4[4];
    static_for<NS>([&](auto JJ) LCP_INL {
      LCP_SOLO_OPAQUE(jcq[JJ]); LCP_SOLO_OPAQUE(jtq[JJ]);
      const TC jc_ = (TC)jcq[JJ], jt_ = (TC)jtq[JJ];
      p0[JJ] = b00 * jc_; p1[JJ] = fma(b10, jc_, b11 * jt_);
      x4[JJ] = (ll == colof(JJ)) ? ((ll < nzs) ? qd : (TC)1) : (TC)0;
    });
    {
      const int oz = lds_opaque_zero();
      const TI* gl = GL + ll + oz;
      const TI* gtl = GTL + ll + oz;
      static_for<4>([&](auto Gq) LCP_INL {
        if (4 * Gq < ncw) {
          TI av[4], bv[4];
          static_for<4>([&](auto Kq) LCP_INL { av[Kq] = gl[(4 * Gq + Kq) * 16]; bv[Kq] = gtl[(4 * Gq + Kq) * 16]; });
          TC ad[4], bd[4];
          static_for<4>([&](auto Kq) LCP_INL { ad[Kq] = (TC)av[Kq]; bd[Kq] = (TC)bv[Kq]; });
          if constexpr (NS == 4) form4x4<4 * Gq>(x4, p0, p1, ad, bd);
          else FormW<4 * Gq, NS>::run(x4, p0, p1, ad, bd);
        }
      });
    }
    SOLO_TICK(1)
    static_for<NS>([&](auto JJ) LCP_INL {
      const M4<TC> g = gather4(x4[JJ]);
      if constexpr (TRIM) {                                              // slot JJ of row r holds column CLO + 4 JJ + r
        constexpr int c0_ = CLO + 4 * JJ;
        if constexpr (c0_ < CHI) xr[c0_] = g.n;
        if constexpr (c0_ + 1 < CHI) xr[c0_ + 1] = g.f1;
        if constexpr (c0_ + 2 < CHI) xr[c0_ + 2] = g.f2;
        if constexpr (c0_ + 3 < CHI) xr[c0_ + 3] = g.g;
      } else { xr[JJ] = g.n; xr[4 + JJ] = g.f1; xr[8 + JJ] = g.f2; xr[12 + JJ] = g.g; }
    });
    bool singular = false;
    udx = 1;
    TC pivv = bc<0>(xr[0]), inv = (TC)1;
    bool primed = false;
    static_for<16>([&](auto K) LCP_INL {
      constexpr int k = K;
      if (k >= es && k < nzs) {
        if (!primed) { pivv = bc<k>(xr[k]); inv = fast_rcp(pivv); primed = true; }
        singular = singular || (pivv == (TC)0);
        const TC sk = xr[k] * inv;                                          // (column k scaled on every row but k: lcp_quad.hip factor_pq)
        const TC lx = keep_if(sk, ll > k);
        xr[k] = (ll == k) ? xr[k] : sk;
        udx = (ll == k) ? inv : udx;
        if constexpr (k + 1 < 16) {
          fnmac_bc<k>(xr[k + 1], xr[k + 1], lx);
          pivv = bc<(k + 1) & 15>(xr[k + 1]); inv = fast_rcp(pivv);
          lu_cols_x<k, k + 2, 14 - k>(xr, lx);
        }
      }
    });
    if constexpr (!TRIM) static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; sp[a] = xr[a]; xr[a] = (a < es) ? (TC)0 : xr[a]; });
    SOLO_TICK(2)
    return row_any(singular);
  };

  // solve_kkt (pdipm.py:325-354) in body space with pinned coordinates (lcp_quad.hip solve_kkt_pq): rs, rz, os, oz per (component, contact)
  auto solve = [&](TC di, TC rx, TC rs, TC rz, TC ry, TC& ox, TC& os, TC& oz, TC& oy) {
    const int nzs = NZF > 0 ? NZF : __builtin_amdgcn_readfirstlane(nz), es = NZF > 0 ? EF : __builtin_amdgcn_readfirstlane(e);
    const TC q = vc ? rs * di - rz : (TC)0;
    const M4<TC> u4 = minv(gather4(q));
    const TC gu = Gtw(vc ? pick(u4) : (TC)0);
    TC wx = (l16 < nzs) ? gu - rx : (TC)0;
    const TC we = (l16 < es) ? -ry : (TC)0;
    SOLO_TICK(3)
    if constexpr (TRIM) {                                                  // (we = -x_p is identically zero; free coordinates only)
      static_for<NCOLS>([&](auto Kq) LCP_INL { constexpr int k = CLO + Kq; fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 > k)); });
      static_for<NCOLS>([&](auto KR) LCP_INL { constexpr int k = CHI - 1 - KR; fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 < k)); });
    } else {
    if (es > 0) static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; fnmac_bc<a>(wx, we, keep_if(sp[a], a < es)); });
    static_for<4>([&](auto Gq) LCP_INL {                                   // L y = rhs (the steps of the pinned columns meet zeros)
      if (4 * Gq < nzs) static_for<4>([&](auto Kq) LCP_INL { constexpr int k = 4 * Gq + Kq; fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 > k)); });
    });
    static_for<4>([&](auto GR) LCP_INL {                                   // U x = y
      constexpr int Gq = 3 - GR;
      if (4 * Gq < nzs) static_for<4>([&](auto KR) LCP_INL {
        constexpr int k = 4 * Gq + 3 - KR;
        fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 < k));
      });
    });
    }
    SOLO_TICK(4)
    ox = (l16 < es) ? we : ((l16 < nzs) ? wx * udx : (TC)0);
    oy = (l16 < es) ? wx : (TC)0;
    const TC t = Gv(ox) - q;                                               // (cone row: G is zero there)
    const M4<TC> o4 = minv(gather4(t));
    oz = vc ? pick(o4) : (TC)0;
    os = vc ? (-rs - oz) * di : (TC)0;                                     // :347,350
    SOLO_TICK(5)
  };

#if !LCP_SOLO_RCP_STEP
  // get_step for (z, dz), (s, ds) of the scene (pdipm.py:182-186): min(step(z, dz), step(s, ds)), NaN semantics as step_pair_q
  auto step_pair = [&](TC zv, TC dz, TC sv, TC ds) -> TC {
    const TC ninf = -inf_of<TC>(), pinf = inf_of<TC>();
    const TC az = -zv / dz, as = -sv / ds;
    const uint32_t nz_ = key_is_nan<TC>(nan_key(az)) ? 1u : 0u, ns_ = key_is_nan<TC>(nan_key(as)) ? 2u : 0u;
    {
      // the fill max(1, a.max()) is at least every entry: with an entry it does not replace (dv <= 0) in both vectors and no NaN,
      // a.min() is the minimum over those entries - no maxima needed (lcp_quad_kernels.inc, step_pair_q: same value bit for bit)
      const uint32_t f = wave_or(vc ? ((!(dz > (TC)0) ? 1u : 0u) | (!(ds > (TC)0) ? 2u : 0u) | ((nz_ | ns_) ? 4u : 0u)) : 0u);
      if (f == 3u) return wave_fmin(vc ? fmin_((dz > (TC)0) ? pinf : az, (ds > (TC)0) ? pinf : as) : pinf);
    }
    const uint32_t kf = wave_or(vc ? (nz_ | ns_) : 0u);                    // a.max(): NaN if any entry is NaN ...
    TC mz = vc ? az : ninf, ms = vc ? as : ninf;
    wave_fmax2(mz, ms);
    const TC fz = (kf & 1u) ? (TC)1 : fmax_(mz, (TC)1), fs = (kf & 2u) ? (TC)1 : fmax_(ms, (TC)1);   // ... and max(1.0, .) maps NaN to 1.0
    const TC pz = (dz > (TC)0) ? fz : az, ps = (ds > (TC)0) ? fs : as;     // a[dv > 0] = fill
    const uint32_t kl = wave_or(vc ? ((key_is_nan<TC>(nan_key(pz)) || key_is_nan<TC>(nan_key(ps))) ? 1u : 0u) : 0u);
    const TC l = wave_fmin(vc ? fmin_(pz, ps) : pinf);                     // a.min(): NaN if any remaining entry is NaN
    return kl ? nan_of<TC>() : l;
  };

#else
  // get_step through the reciprocals of the iterate (LCP_SOLO_RCP_STEP; step_pair_rcp of lcp_quad_kernels.inc for one row per lane):
  // alpha = -1 / min_i(dv_i / v_i) where both vectors have a decreasing entry and no t_i is zero, NaN or infinite; the exact form of
  // lcp_device.h (step_flags / step_from_flags) otherwise
  auto step_pair_rcp = [&](TC rzv, TC dz, TC rsv, TC ds) -> TC {
    const TC tz = dz * rzv, ts = ds * rsv;
    const bool bad = __builtin_amdgcn_class(tz * ts, 0x2F7);               // NaN, +-inf, +-0, +-denormal
    if (__builtin_expect(!__any(vc && bad) && __any(vc && tz < (TC)0) && __any(vc && ts < (TC)0), 1))
      return -fast_rcp(wave_fmin(vc ? fmin_(tz, ts) : inf_of<TC>()));
    const uint32_t g = wave_or(vc ? (step_flags(tz) | (step_flags(ts) << 8)) : 0u);
    TC mz = vc ? tz : inf_of<TC>(), ms = vc ? ts : inf_of<TC>();
    mz = wave_fmin(mz); ms = wave_fmin(ms);
    return pmin(step_from_flags(g & 0xffu, mz), step_from_flags(g >> 8, ms));
  };

#endif
#ifdef LCP_SOLO_PROFILE
  tk = clock64(); pc[8] = tk - t_start;
#endif
#if LCP_SOLO_PEEL_INIT
  // (the initialisation pass - it = -1 - as its own copy of the loop body: LCP_Q_PEEL_INIT in lcp_quad_kernels.inc)
  auto iteration = [&](auto INIT_, const int it) LCP_INL -> bool {
    constexpr bool INIT = decltype(INIT_)::value;
    __builtin_assume(INIT == (it < 0));
#define LCP_SOLO_BREAK return false
#else
#pragma unroll 1
  for (int it = -1; it < max_iter; ++it) {
#define LCP_SOLO_BREAK break
#endif
    TC rx, ry, rs, rz, mu = 0, resid = 0, szsum = 0;
    if (it < 0) {                                                          // init: (p, 0, -h, -b), d = 1 (:57-63); b = 0 from a contact list
      rx = p; ry = 0; rs = 0; rz = -hn; dinv = 1;
    } else {                                                               // residuals (:82-96)
      rx = Gtw(vc ? z : (TC)0) + qd * x + p;
      if (e > 0) rx += (l16 < e) ? y : (TC)0;                              // A = [I 0]: A^T y is y on the pinned lanes
      rs = z;
      const M4<TC> z4 = gather4(z);
      const TC fzv = fma(fzn, z4.n, fma(fz12, z4.f1 + z4.f2, fzg * z4.g));
      const TC gxv = Gv(x);                                                // (a cross-lane product: never inside a per-lane conditional)
      rz = vc ? (gxv + s) - (hn + fzv) : (TC)0;
      ry = (e > 0 && l16 < e) ? x : (TC)0;                                 // A x = x_p, b = 0
      TC n_rz = rz * rz, sz = vc ? s * z : (TC)0;
      TC n_rx = (c0 && l16 < nz) ? rx * rx : (TC)0, n_ry = c0 ? ry * ry : (TC)0;    // (x-space is replicated in the rows: row 0 counts)
      wave_sum4(n_rz, sz, n_rx, n_ry);
      szsum = sz;
      mu = sz / mf; mu = mu < 0 ? -mu : mu;                                // (:91)
      resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + mf * mu;              // (:92-96)
#if LCP_SOLO_RCP_STEP
      dinv = vc ? s * fast_rcp(z) : (TC)1;
#else
      dinv = vc ? s / z : (TC)1;                                           // 1 / d, d = z / s (:98)
#endif
    }
    SOLO_TICK(0)
    const bool singular = factor(dinv);                                    // (:99-100)
    if (it < 0 && singular && e > 0) status |= LCP_ST_SINGULAR_S11;
    if (it >= 0 && !done) {
      ++iters;
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; done = true; }   // except: return best (:99-102)
      else {
        const bool improved = !have_best || (resid < best_resid);             // (:107-132)
        if (improved) { best_resid = resid; n_not = 0; have_best = true; bx = x; by = y; bz = z; bs = s; }
        else ++n_not;
        if (n_not == lim || best_resid < eps || mu > mu_limit<TC>()) done = true;   // (:133)
      }
    }
    SOLO_TICK(6)
    if (done) LCP_SOLO_BREAK;
    if (it >= 0 && it == max_iter - 1) LCP_SOLO_BREAK;                             // (the iterate the last pass would produce is never evaluated)
    TC ax = 0, ay = 0, as_ = 0, az = 0;
    const int npass = (it < 0) ? 1 : 2;
#if LCP_SOLO_UNROLL_PASS
    // (the two solves of an iteration as two copies of the code, `pass` a compile-time constant - as LCP_Q_UNROLL_PASS in lcp_quad_kernels.inc)
    auto one_pass = [&](auto PASS_) LCP_INL {
      constexpr int pass = PASS_;
#else
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
#endif
      TC ox, oy, os, oz;
      solve(dinv, rx, rs, rz, ry, ox, os, oz, oy);
      if (it < 0) {
        x = ox; s = os; z = oz; y = oy;                                       // (:60-63)
        const uint32_t kn = wave_or(vc ? ((s != s ? 1u : 0u) | (z != z ? 2u : 0u)) : 0u);
        TC smin = wave_fmin(vc ? s : inf_of<TC>()), zmin = wave_fmin(vc ? z : inf_of<TC>());
        if (kn & 1u) smin = nan_of<TC>();
        if (kn & 2u) zmin = nan_of<TC>();
        if (smin <= (TC)0) s += (TC)1 - smin;                                 // (:66-75)
        if (zmin <= (TC)0) z += (TC)1 - zmin;
        if (!vc) { s = 1; z = 1; }
        if (ncs == 0) { bx = x; by = y; done = true; }                        // engines.py:36-50: x = P^-1 u, no LCP
      } else if (pass == 0) {
        ax = ox; ay = oy; as_ = os; az = oz;                                  // affine direction (:138-139)
#if LCP_SOLO_RCP_STEP
        const TC irs = fast_rcp(s), irz = dinv * irs;                         // (1 / s_i: the step length and the corrector's rs / s; 1 / z_i = (s_i / z_i) (1 / s_i))
        const TC alpha = pmin(step_pair_rcp(irz, az, irs, as_), (TC)1);
#else
        const TC alpha = pmin(step_pair(z, az, s, as_), (TC)1);              // (:142-144)
#endif
        const TC t3 = wave_sum(vc ? (s + alpha * as_) * (z + alpha * az) : (TC)0);
        const TC r3 = t3 / szsum, sig = r3 * r3 * r3;                         // (:146-150)
        const TC ms = -mu * sig;
        rx = 0; ry = 0; rz = 0;
#if LCP_SOLO_RCP_STEP
        rs = vc ? (ms + as_ * az) * irs : (TC)0;
#else
        rs = vc ? (ms + as_ * az) / s : (TC)0;                                // (:153)
#endif
      } else {
        const TC cx = ox + ax, cy = oy + ay, cs = os + as_, cz = oz + az;     // (:160-163)
#if LCP_SOLO_RCP_STEP
        const TC irs = fast_rcp(s), irz = dinv * irs;
        const TC alpha = pmin((TC)0.999 * step_pair_rcp(irz, cz, irs, cs), (TC)1);
#else
        const TC alpha = pmin((TC)0.999 * step_pair(z, cz, s, cs), (TC)1);   // (:164-166)
#endif
        x += alpha * cx; y += alpha * cy;                                     // (:171-174)
        if (vc) { s += alpha * cs; z += alpha * cz; }
      }
      SOLO_TICK(7)
#if LCP_SOLO_UNROLL_PASS
    };
    one_pass(std::integral_constant<int, 0>{});
    if (npass > 1) one_pass(std::integral_constant<int, 1>{});
#else
    }
#endif
    if (done) LCP_SOLO_BREAK;
#if LCP_SOLO_PEEL_INIT
    return true;
  };
  if (iteration(std::true_type{}, -1)) {
#pragma unroll 1
    for (int it = 0; it < max_iter; ++it) { if (!iteration(std::false_type{}, it)) break; }
  }
#else
  }
#endif
#undef LCP_SOLO_BREAK

  // ---- outputs: natural m-space order [normal | friction pairs | cone]; the best iterate also goes to the workspace in fp64
  const int oi = c0 ? l16 : (c3 ? 3 * nc + l16 : nc + 2 * l16 + (comp - 1));
  if (c0) {
    if (l16 < nz) W.x[l16] = bx; else bx = 0;
    if (l16 < e) W.y[l16] = by; else by = 0;
  }
  if (vc) { W.z[oi] = bz; W.s[oi] = bs; } else { bz = 1; bs = 1; }
  bool bad = (bx != bx);
  if (vc) bad = bad || (bz != bz) || (bs != bs);
  if (__any(bad)) status |= LCP_ST_NAN;
  TI* zo = (TI*)SP.z; TI* so = (TI*)SP.s; TI* yo = (TI*)SP.y;
  if (l16 < nc) {
    const TI k = vc ? (TI)1 : (TI)0;                                      // padded slots report 0
    if (zo) zo[(size_t)scene * m + oi] = k * (TI)bz;
    if (so) so[(size_t)scene * m + oi] = k * (TI)bs;
  }
  if (c0) {
    if (l16 < e && yo) yo[(size_t)scene * e + l16] = (TI)by;
    if (l16 < nz) {
      const TC nv = -bx;                                                  // engines.py:76-77
      ((TI*)SP.v_new)[(size_t)scene * nz + l16] = (TI)nv;
      if (SP.p_new) ((TI*)SP.p_new)[(size_t)scene * nz + l16] = (TI)((TC)((const TI*)SP.pos)[(size_t)scene * nz + l16] + nv * (TC)SP.dt);   // bodies.py:81
    }
    if (l16 == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
  }
#ifdef LCP_SOLO_PROFILE
  __builtin_amdgcn_s_waitcnt(0);
  if (lane == 0 && so) { TI* o = so + (size_t)scene * m + (m - 11); for (int i = 0; i < 9; ++i) o[i] = (TI)pc[i]; o[9] = (TI)iters; o[10] = (TI)(clock64() - t_start); }
#endif
}

}  // namespace solo

// sizes lcp_fwd_solo takes (the body-space four-scenes-per-wave sizes with nz <= 16)
bool solo_supported(int nz, int m, int e) { return (m % 4 == 0) && (m / 4 <= q16::NCQ) && nz <= 16 && e <= q16::EQ; }
int solo_step(const StepArgs& SP, void* stream, bool pinned) {
  const int nz = 3 * SP.nb, e = SP.e, ph = pinned ? 1 : 0;
  const dim3 grid(SP.B), blk(64);
  hipStream_t st = (hipStream_t)stream;
  // the stack shapes of the BASELINE configs (floor pinned by a TotalConstraint: neq 3) get their sizes at compile time
  const bool full = !SP.c_count;                                   // full contact lists: the contact count is a constant too
  if (nz == 9 && e == 3 && full && SP.nc == 8) hipLaunchKernelGGL((solo::lcp_fwd_solo<9, 3, 8>), grid, blk, 0, st, SP, ph);
  else if (nz == 9 && e == 3) hipLaunchKernelGGL((solo::lcp_fwd_solo<9, 3>), grid, blk, 0, st, SP, ph);
  else if (nz == 15 && e == 3 && full && SP.nc == 16) hipLaunchKernelGGL((solo::lcp_fwd_solo<15, 3, 16>), grid, blk, 0, st, SP, ph);
  else if (nz == 15 && e == 3) hipLaunchKernelGGL((solo::lcp_fwd_solo<15, 3>), grid, blk, 0, st, SP, ph);
  else hipLaunchKernelGGL((solo::lcp_fwd_solo<0, 0>), grid, blk, 0, st, SP, ph);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
