// lcp_primal_chain.hip - the body-space step kernel (lcp_primal.hip, lcp_primal_step.inc) instantiated for 5 .. 24 equality rows:
// chains of joints (two rows per revolute `Joint`, constraints.py:13-50 - the reference's chain demo, `testChain`, the ten links
// of experiments/inference.py).  The rows of A live packed in LDS (e x nz floats), every equality lane sums its own row for A v.
// A translation unit of its own so that the build compiles the instantiations in parallel.
// The step lengths of THIS unit stay on the IEEE-quotient form (LCP_PRIMAL_RCP_STEP = 0).  Measured, round 5: with the reciprocal form the
// 56-column instantiation (one wavefront per SIMD, 90 accumulator registers of spill traffic, no inline asm anywhere in it) returns wrong
// velocities on five chain cases - while the same source with the fast branch disabled, or with a printf beside it, passes, and fast and
// exact form print identical step lengths (tools/gpu_calls/r05_e.sh): a code-generation fragility of that one instantiation, not
// arithmetic.  Chains are not where the divisions cost (five to twenty-four equality rows dominate their factorisation).
#define LCP_PRIMAL_RCP_STEP 0
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
