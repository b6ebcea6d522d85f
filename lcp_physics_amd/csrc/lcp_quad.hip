// lcp_quad.hip - contact-structured PDIPM kernels, FOUR SCENES PER WAVEFRONT (16 lanes per scene).
//
// The fastest path: contact-structured LCPs (engines.py:67-73) with diagonal Q, nc <= 16 contacts,
// nz <= 16 (dense inputs) or nz <= 32 (contact-list inputs, template parameter XH = 2), neq <= 4, fp32 I/O.
// Mapping (CDNA4-first):
//   * one 16-lane DPP row = one scene, lane c = contact c.  The lane holds everything that belongs to its
//     contact: the four inequality components (normal, friction +, friction -, gamma) of every m-space
//     vector, its rows of Jc and Jt, and BOTH of its rows of the reduced 2nc x 2nc system (a_c and u_c,
//     see lcp_wave64.hip `Red` for the algebra).  The per-contact 2x2 elimination, F z, the reduction's
//     right-hand side and back-substitution are therefore lane-local - no cross-lane traffic at all.
//   * x-space vectors live one entry per lane of the row (two with XH = 2), e-space (neq <= 4) likewise.
//   * every cross-lane move is a DPP row operation: `row_newbcast:k` broadcasts lane k of each row to its row
//     (pivot rows in the LU, vector entries in the products), quad_perm / row_mirror give row-local
//     reductions.  Micro-benchmark (tools/microbench/pair_cost.hip): 6.7 cycles per fp64 FMA fed by a 64-bit DPP
//     move, 4.7 with the broadcast folded into v_fmac_f64_dpp (the LU), versus 21-29 through v_readlane, and one
//     instruction serves four scenes.
//   * the pivot-free LU runs over the 2 x 32 register-resident row entries of each lane.
// Scenes of a wave advance in lock-step; a scene that terminates (pdipm.py:133) freezes its state and waits.  Scenes may
// have different contact counts (the fused step with `c_count`): lane masks are per row, loops run to the wave maximum
// over identity rows, a scene without contacts leaves after the initialisation solve (engines.py:36-50).
// Kernels: lcp_fwd_quad (dense or fused forward), lcp_bwd_quad (lcp.py:37-64, dense gradients), lcp_bwd_step_quad
// (gradients w.r.t. the physical inputs of the fused step).
// lcp_fwd_quad comes in three factorisations (template parameter ALG): 0 = the reduced 2 nc x 2 nc contact-space system described
// above (the dense boundary, nz > 16, fp32 arithmetic); 1 = the BODY-space system K = [[Q + G^T M^-1 G, A^T], [A, 0]] of nz + neq <= 20
// rows (x-row and equality row per lane; lcp_primal.hip has the algebra); 2 = the same with the pinned leading coordinates of a
// fixed floor (A = [I 0]) taken out - one row per lane, nz - neq pivots - alone in its kernel.  The contact-list entry points
// launch 2 and, behind it, 1 for the waves 2 declined; the backward kernels stay in contact space (factor_bwd_q).
// Same algorithm and the same reference lines as lcp_wave64.hip / lcp_generic.hip.
#include "lcp_quad_kernels.inc"

namespace lcp {

// ---------------------------------------------------------------- host-side launchers
bool quad_supported(int nz, int m, int e) { return (m % 4 == 0) && (m / 4 <= q16::NCQ) && nz <= 16 && e <= q16::EQ; }
// contact-list entry points only (lcp_solve_dynamics_f32 / lcp_step_backward_f32): up to ten bodies
bool quad_step_supported(int nz, int m, int e) { return (m % 4 == 0) && (m / 4 <= q16::NCQ) && nz <= 32 && e <= q16::EQ; }

// ---- size-specialised instantiations (lcp_quad_n*e*.hip): the pinned body-space kernels with nz / neq at compile time
#define LCP_QS_DECL(NZ, E)                                                                                       \
  int quad_sized_fwd_##NZ##_##E(const StepArgs& SP, int ls, size_t lds_bytes, int accept, void* stream);        \
  int quad_sized_dense_fwd_##NZ##_##E(const FwdArgs& P, int ls, size_t lds_bytes, int accept, void* stream);    \
  int quad_sized_bwd_##NZ##_##E(const BwdArgs& P, int ls, int accept, bool pinned, void* stream);               \
  int quad_sized_step_bwd_##NZ##_##E(const StepArgs& SP, const StepBwdArgs& Gd, int ls, bool pinned, void* stream);
LCP_QS_DECL(15, 3) LCP_QS_DECL(9, 3) LCP_QS_DECL(12, 3) LCP_QS_DECL(6, 3)
#ifndef LCP_Q_SIZED
#define LCP_Q_SIZED 1         // 0: always the run-time-size kernels (A/B aid)
#endif
#define LCP_QS_FOR_EACH(nz, e, CALL)                         \
  if (LCP_Q_SIZED && (e) == 3) {                             \
    if ((nz) == 15) return CALL(15, 3);                      \
    if ((nz) == 9) return CALL(9, 3);                        \
    if ((nz) == 12) return CALL(12, 3);                      \
    if ((nz) == 6) return CALL(6, 3);                        \
  }

// LDS bytes per scene.  The four scenes of a wave issue every LDS access together (16 lanes each, same offsets inside
// their blocks): the block stride is padded to 64 B modulo 256 B = 16 banks modulo 64, so that the four 16-lane groups
// land on four disjoint bank groups (a stride that is a multiple of 128 B - the unpadded 3456 B - puts them all on the
// same banks).  Measured: SQ_LDS_BANK_CONFLICT 273 k -> 169 k cycles per launch; no change of the kernel time - the
// single wave waits out the LDS latency either way.
template <typename TC, typename TI = float>
static size_t q16_lds(bool with_w, int xh = 1, bool with_gal = true) {
  q16::LdsQ<TI, TC> L;
  size_t n = q16::carve_q<TI, TC>(L, nullptr, with_w, xh, with_gal);
  while (n % 256 != 64) n += 16;
  return n;
}

// The dense LCPFunction boundary (fp32 tensors, fp64 arithmetic) runs the BODY-space kernels by default since round 4: the pinned
// variant (ALG = 2; size-specialised for the stack shapes) for the waves whose equality rows pin the leading coordinates, the
// general body-space kernel (ALG = 1) right behind it for the scenes those waves marked - exactly what the contact-list entry
// points launch, reading the rows of G instead of a contact list.  Nothing of the contact-space pre-factorisation is formed: no
// W = J P J^T written (45 MB per launch at 4096 x 16 contacts) and re-read at every factorisation, 12 pivots instead of 32.
// LCP_PATH_CONTACT_SPACE in the `compute` word keeps the contact-space factorisation - the reference's own formulation: with it
// the exit tests of pdipm.py:133 fall where the reference's fall even on solves that converge to rounding inside the ten
// iterations (the body-space kernels take the same Newton steps to ~1e-12 but can leave such a solve one iteration later; on
// BASELINE configs[2] / [3] the iteration counts of both equal the oracle's on every scene - tests/test_hip_headline_parity.py).
// fp64 tensors and fp32 arithmetic keep the contact-space kernels.
bool quad_dense_is_body_space(int io_f64, int compute, int body_space) { return body_space && !io_f64 && compute == LCP_COMPUTE_F64; }
int quad_forward(const FwdArgs& P, int compute, int accept, void* stream, int io_f64, int body_space) {
  StepArgs SP = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P.B + 3) / 4), blk(64);
  if (quad_dense_is_body_space(io_f64, compute, body_space)) {
    const int lb = (int)q16_lds<double>(LCP_Q_LDSW != 0, 1, false);          // body-space kernels: no contact-space tables
    const size_t lbytes = 4 * lb + 20 * 64 * sizeof(double);                 // (+ the parked best iterate and affine direction)
    auto first = [&]() -> int {
#define LCP_QS_CALL(NZ, E) quad_sized_dense_fwd_##NZ##_##E(P, lb, lbytes, accept, stream)
      if (LCP_Q_SIZED && P.e == 3) {
        int rc = 1;
        if (P.nz == 15) rc = LCP_QS_CALL(15, 3);
        else if (P.nz == 9) rc = LCP_QS_CALL(9, 3);
        else if (P.nz == 12) rc = LCP_QS_CALL(12, 3);
        else if (P.nz == 6) rc = LCP_QS_CALL(6, 3);
        if (rc <= 0) return rc;                                              // (1: not the shape's contact count)
      }
#undef LCP_QS_CALL
      hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, false, 1, 2>), grid, blk, lbytes, st, P, SP, lb, accept);
      return 0;
    };
    const int rc = first();
    if (rc) return rc;
    hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, false, 1, 1>), grid, blk, 4 * lb, st, P, SP, lb, accept | 16);   // whatever that one left
    return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
  }
  if (io_f64) {                                  // the reference's native dtype (physics/utils.py:34): fp64 loads / stores
    const int ls = (int)q16_lds<double, double>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<double, double, false, 1>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  } else if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, false, 1, 0>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  } else {
    const int ls = (int)q16_lds<float>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, false, 1>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

// does quad_step / quad_step_backward run the body-space kernels for these arguments ?  (fp64 arithmetic, nz <= 16, not forced off)
bool quad_step_is_body_space(int nz, int compute, int body_space) { return body_space && compute == LCP_COMPUTE_F64 && nz <= 16; }

// `body_space`: factor / solve the (nz + neq)-row body-space system instead of the 32-row contact-space one (fp64 arithmetic, nz <= 16)
// Below LCP_SOLO_MAX_B scenes the body-space step runs ONE scene per wavefront (lcp_solo.hip): four scenes per wavefront leave
// most SIMDs of the chip without a wave (B / 4 wavefronts on 1024 SIMDs) and a scene's dependent chain takes the same time
// whatever the batch.  Measured cross-over: see DESIGN.md (latency curve B = 256 .. 4096).
#ifndef LCP_SOLO_MAX_B
#define LCP_SOLO_MAX_B 1024
#endif
int quad_step(const StepArgs& SP, int compute, void* stream, int body_space, int solo, bool pinned) {
  FwdArgs P = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((SP.B + 3) / 4), blk(64);
  const bool wide = 3 * SP.nb > 16;
  if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(LCP_Q_LDSW != 0, wide ? 2 : 1);
    const int lb = (int)q16_lds<double>(LCP_Q_LDSW != 0, 1, false);          // body-space kernels: no contact-space tables
    if (wide) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 2>), grid, blk, 4 * ls, st, P, SP, ls, 2);
    else if (body_space && (solo > 0 || (solo < 0 && SP.B <= LCP_SOLO_MAX_B))) {
      int rc = solo_step(SP, stream, pinned);                                                                     // pinned leading coordinates, one scene per wave
      if (rc) return rc;
      if (!pinned) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 1>), grid, blk, 4 * lb, st, P, SP, lb, 3);   // whatever that one left
    } else if (body_space) {
      // (+ the parked best iterate and affine direction: 2 x 10 doubles per lane)
      // (accept 4 = LCP_HINT_PINNED: nothing is launched behind; a wave that does not qualify returns NaN velocities)
      const size_t lbytes = 4 * lb + 20 * 64 * sizeof(double);
      auto first = [&]() -> int {                                                                                 // pinned leading coordinates
#define LCP_QS_CALL(NZ, E) quad_sized_fwd_##NZ##_##E(SP, lb, lbytes, pinned ? 4 : 2, stream)
        LCP_QS_FOR_EACH(3 * SP.nb, SP.e, LCP_QS_CALL)
#undef LCP_QS_CALL
        hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 2>), grid, blk, lbytes, st, P, SP, lb, pinned ? 4 : 2);
        return 0;
      };
      const int rc = first();
      if (rc) return rc;
      if (!pinned) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 1>), grid, blk, 4 * lb, st, P, SP, lb, 3);   // whatever that one left
    }
    else hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1>), grid, blk, 4 * ls, st, P, SP, ls, 2);
  } else {
    const int ls = (int)q16_lds<float>(LCP_Q_LDSW != 0, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, true, 2>), grid, blk, 4 * ls, st, P, SP, ls, 2);
    else hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, true, 1>), grid, blk, 4 * ls, st, P, SP, ls, 2);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

// PdipmEngine.post_stabilization (engines.py:80-116) on the four-scenes-per-wavefront mapping (round 5): nz <= 16, <= 16 contacts, <= 4 joint
// rows, fp64 arithmetic.  The pinned body-space kernel first (equality rows [I 0] with Je v = 0: the fixed floor of the reference's worlds),
// the general body-space kernel behind it for the waves that one marked.  The best iterate is left in lcp_poststab_primal_kernel's
// workspace layout: lcp_post_stabilization_backward_f32 (one wave per scene) serves both forwards.
bool quad_post_supported(int nz, int m, int e) { return nz <= 16 && (m % 4) == 0 && m / 4 <= 16 && e <= q16::EQ; }
int quad_post_stab(const StepArgs& SP, void* stream) {
  FwdArgs P = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((SP.B + 3) / 4), blk(64);
  const int lb = (int)q16_lds<double>(LCP_Q_LDSW != 0, 1, false);
  const size_t lbytes = 4 * lb + 20 * 64 * sizeof(double);                   // (+ the parked best iterate and affine direction)
  hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 2, 0, 0, 0, false, true>), grid, blk, lbytes, st, P, SP, lb, 2);
  hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 1, 0, 0, 0, false, true>), grid, blk, 4 * lb, st, P, SP, lb, 3);   // whatever that one left
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

#ifndef LCP_Q_BWD_SPLIT
#define LCP_Q_BWD_SPLIT 0     // 1: body-space dense backward as solve kernel + streaming kernel for dG / dF.  Measured and NOT the default
                              // (profiles/r06_ab_bwd_split.txt): the streaming kernel alone - sixteen wavefronts per SIMD, 16-byte stores - takes 16.5 us for
                              // the 83 MB (5.0 TB/s), what the stores cost INSIDE the one-kernel form too (dG + dF: 16.5 us of its 25.2): the writes are
                              // at what the memory system takes, not starved by occupancy, and two launches add 3 us
#endif
static int quad_backward_solve_body(const BwdArgs& P, int ls, int accept, bool pinned, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P.B + 3) / 4), blk(64);
#define LCP_QS_CALL(NZ, E) quad_sized_bwd_##NZ##_##E(P, ls, accept, pinned, stream)
  LCP_QS_FOR_EACH(P.nz, P.e, LCP_QS_CALL)
#undef LCP_QS_CALL
  hipLaunchKernelGGL((q16::lcp_bwd_quad<float, double, true>), grid, blk, 4 * ls, st, P, ls, accept);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
int quad_backward(const BwdArgs& P, int compute, int accept, void* stream, int io_f64, int body, bool pinned) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P.B + 3) / 4), blk(64);
  if (body) {                                    // workspace of a body-space forward (fp32 I/O, fp64 arithmetic)
    if (io_f64 || compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
    const int ls = (int)q16_lds<double>(false);
    BwdArgs Ps = P;
    // dG / dF (20 of the 21.6 KB per scene at 16 contacts) from their own kernel: 16-byte stores need (m nz) % 4 == 0 and aligned outputs
    Ps.split = (LCP_Q_BWD_SPLIT && (P.dG || P.dF) && ((P.m * P.nz) & 3) == 0 && ((((uintptr_t)P.dG) | ((uintptr_t)P.dF)) & 15) == 0) ? 1 : 0;
    const int rc = quad_backward_solve_body(Ps, ls, accept, pinned, stream);
    if (rc || !Ps.split) return rc;
    hipLaunchKernelGGL((q16::lcp_bwd_stream_quad<float, double, 256>), dim3(P.B), dim3(256), 0, st, Ps, accept);
  } else if (io_f64) {
    const int ls = (int)q16_lds<double, double>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<double, double>), grid, blk, 4 * ls, st, P, ls, accept);
  } else if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<float, double>), grid, blk, 4 * ls, st, P, ls, accept);
  } else {
    const int ls = (int)q16_lds<float>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<float, float>), grid, blk, 4 * ls, st, P, ls, accept);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int quad_step_backward(const StepArgs& SP, const StepBwdArgs& Gd, int compute, void* stream, int body_space, bool pinned) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((SP.B + 3) / 4), blk(64);
  const bool wide = 3 * SP.nb > 16;
  if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(false, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 2>), grid, blk, 4 * ls, st, SP, Gd, ls);
    else if (body_space) {
#define LCP_QS_CALL(NZ, E) quad_sized_step_bwd_##NZ##_##E(SP, Gd, ls, pinned, stream)
      LCP_QS_FOR_EACH(3 * SP.nb, SP.e, LCP_QS_CALL)
#undef LCP_QS_CALL
      hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 1, true>), grid, blk, 4 * ls, st, SP, Gd, ls);
    }
    else hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 1>), grid, blk, 4 * ls, st, SP, Gd, ls);
  } else {
    const int ls = (int)q16_lds<float>(false, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, float, 2>), grid, blk, 4 * ls, st, SP, Gd, ls);
    else hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, float, 1>), grid, blk, 4 * ls, st, SP, Gd, ls);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
