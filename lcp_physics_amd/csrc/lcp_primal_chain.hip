// lcp_primal_chain.hip - the body-space step kernel (lcp_primal.hip, lcp_primal_step.inc) instantiated for 5 .. 24 equality rows:
// chains of joints (two rows per revolute `Joint`, constraints.py:13-50 - the reference's chain demo, `testChain`, the ten links
// of experiments/inference.py).  The rows of A live packed in LDS (e x nz floats), every equality lane sums its own row for A v.
// A translation unit of its own so that the build compiles the instantiations in parallel.
// Round 5 kept THIS unit's step lengths on IEEE quotients because its 56-column instantiation returned wrong velocities with the
// reciprocal form.  Round 6 found why (profiles/r06_chain_rootcause.txt): not the step lengths - the compiler had placed the VGPR -> AGPR
// spill of rz.n (two v_accvgpr_write_b32) at the top of the join block of `if (lane >= NROW) t[..] = 0`, in front of the
// `s_or_b64 exec, exec, s[0:1]` that re-enables lanes 0 .. 55: the spill saved lanes 56 .. 63 only and the first solve of every scene read
// stale accumulator registers for -h.  The build now moves such spill code behind the restore (csrc/compile_unit.sh,
// tools/isa_lint.py --fix) and the unit uses the reciprocal step lengths like every other (LCP_PRIMAL_RCP_STEP default 1).
#include "lcp_primal_common.h"

namespace lcp {
namespace primal {
#include "lcp_primal_step.inc"
}  // namespace primal

template <int NCOL, bool BWD>
static int chain_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) {
  DenseIO DN = {};
  hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, false, primal::WsLayout::YCAP>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <bool BWD>
static int chain_dispatch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) {
  const int n = 3 * SP.nb + SP.e;
  if (n <= 24) return chain_launch<24, BWD>(SP, Gd, stream);
  if (n <= 40) return chain_launch<40, BWD>(SP, Gd, stream);
  return chain_launch<56, BWD>(SP, Gd, stream);
}
int primal_chain_launch(const StepArgs& SP, const StepBwdArgs& Gd, int backward, void* stream) {
  return backward ? chain_dispatch<true>(SP, Gd, stream) : chain_dispatch<false>(SP, Gd, stream);
}

}  // namespace lcp
