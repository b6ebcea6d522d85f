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
// Translation units: this file (the step and the dense boundary, up to 4 equality rows), lcp_primal_pin.hip (the step when the
// equality rows are known to pin the leading coordinates: nz - neq pivots), lcp_primal_chain.hip (the step with 5 .. 24 equality
// rows), lcp_primal_poststab.hip (post-stabilisation); the kernel template lives in lcp_primal_step.inc.
#include "lcp_primal_common.h"

namespace lcp {
namespace primal {
#include "lcp_primal_step.inc"
}  // namespace primal

// nz + neq rows on the lanes of one wave, a contact per lane
bool primal_supported(int nz, int m, int e) {
  // (round 6: up to 64 rows - every lane of the wave a row - for the contact-list step and its backward with at most four equality rows:
  //  18 .. 20 bodies on a pinned floor no longer fall to the workgroup-per-scene generic kernels, 62-91 us per scene and step)
  return (m % 4) == 0 && m / 4 <= 64 && e <= primal::WsLayout::YCAP && (nz % 3) == 0 && (nz + e <= 56 || (e <= primal::EQB && nz + e <= 64));
}
// post-stabilisation (lcp_primal_poststab.hip) has the same instantiations; the dense boundary keeps the 56-row ones
bool primal_poststab_supported(int nz, int m, int e) { return primal_supported(nz, m, e); }
// the dense boundary keeps the four-row instantiations
bool primal_dense_supported(int nz, int m, int e) { return e <= primal::EQB && primal_supported(nz, m, e) && nz + e <= 56; }
static_assert(sizeof(double) * (size_t)primal::WsLayout::TOTAL <= DENSE_EXTRACT_OFF, "lcp_classify_big's per-contact records start behind the body-space kernels' iterate block");
size_t primal_ws_bytes() { return sizeof(double) * (size_t)primal::WsLayout::TOTAL; }

template <int NCOL, bool BWD, bool DENSE = false>
static int primal_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, const DenseIO& DN = DenseIO{}) {
  hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, DENSE, primal::EQB>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <bool BWD, bool DENSE = false>
static int primal_dispatch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, const DenseIO& DN = DenseIO{}) {
  const int n = 3 * SP.nb + SP.e;
  if constexpr (!DENSE) {
    if (SP.e > primal::EQB) return primal_chain_launch(SP, Gd, BWD ? 1 : 0, stream);     // 5 .. 24 equality rows: lcp_primal_chain.hip
  }
  if (n <= 24) return primal_launch<24, BWD, DENSE>(SP, Gd, stream, DN);
  if (n <= 40) return primal_launch<40, BWD, DENSE>(SP, Gd, stream, DN);
  if constexpr (!DENSE) { if (n > 56) return primal_launch<64, BWD, false>(SP, Gd, stream, DN); }
  return primal_launch<56, BWD, DENSE>(SP, Gd, stream, DN);
}
// pinned: LCP_HINT_PINNED came with the call (the forward's word travels with its backward): lcp_primal_pin.hip where its sizes allow
int primal_step(const StepArgs& SP, void* stream, bool pinned) {
  StepBwdArgs Gd = {};
  if (pinned && primal_pin_supported(3 * SP.nb, SP.e)) return primal_pin_launch(SP, Gd, 0, stream);
  return primal_dispatch<false>(SP, Gd, stream);
}
int primal_step_backward(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, bool pinned) {
  if (pinned && primal_pin_supported(3 * SP.nb, SP.e)) return primal_pin_launch(SP, Gd, 1, stream);
  return primal_dispatch<true>(SP, Gd, stream);
}

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
  SP.tag = P.tag; SP.tag_value = P.tag_value;
  StepBwdArgs Gd = {};
  if (primal_pin_supported(P.nz, P.e)) { const int rc = primal_pin_dense_launch(SP, DN, 0, stream); if (rc) return rc; }   // class 4: the pinned form
  return primal_dispatch<false, true>(SP, Gd, stream, DN);
}
int primal_dense_backward(const BwdArgs& P, int32_t* cls, size_t ws_scene, void* stream) {
  DenseIO DN = {};
  DN.nz = P.nz; DN.m = P.m; DN.cls = cls; DN.ws_scene = ws_scene;
  DN.G = (const float*)P.G; DN.A = (const float*)P.A; DN.dl_dx = (const float*)P.dl_dx;
  DN.dQ = (float*)P.dQ; DN.dp = (float*)P.dp; DN.dG = (float*)P.dG; DN.dh = (float*)P.dh; DN.dA = (float*)P.dA; DN.db = (float*)P.db; DN.dF = (float*)P.dF;
  StepArgs SP = {};
  SP.B = P.B; SP.nb = P.nz / 3; SP.nc = P.m / 4; SP.e = P.e; SP.ws = P.ws;
  SP.tag = (int32_t*)P.tag; SP.tag_value = P.tag_value;
  StepBwdArgs Gd = {};
  if (primal_pin_supported(P.nz, P.e)) { const int rc = primal_pin_dense_launch(SP, DN, 1, stream); if (rc) return rc; }
  return primal_dispatch<true, true>(SP, Gd, stream, DN);
}

}  // namespace lcp
