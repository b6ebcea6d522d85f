// lcp_primal_pin.hip - the body-space step kernel (lcp_primal.hip, lcp_primal_step.inc) in its PINNED form: the caller's
// LCP_HINT_PINNED says that the equality rows of every scene pin the leading coordinates (Je = [I 0]: the TotalConstraint that
// fixes the floor of the reference's worlds, physics/constraints.py:175-192), so the factored system is the (nz - neq)-square
// S_ff - 30 pivots over 32 columns on BASELINE config 5 (10-body pile + floor) instead of 36 over 40.
// A translation unit of its own so that the build compiles the instantiations in parallel.
#include "lcp_primal_common.h"

namespace lcp {
namespace primal {
#include "lcp_primal_step.inc"
}  // namespace primal

template <int NCOL, bool BWD>
static int pin_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) {
  DenseIO DN = {};
  hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, false, primal::EQB, 3>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <int NCOL, bool BWD>
static int pin_dense_launch(const StepArgs& SP, const DenseIO& DN, void* stream) {
  StepBwdArgs Gd = {};
  hipLaunchKernelGGL((primal::lcp_primal_kernel<NCOL, BWD, true, primal::EQB, 3>), dim3(SP.B), dim3(64), 0, (hipStream_t)stream, SP, Gd, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
// the pinned form behind the dense LCPFunction boundary: the scenes lcp_classify_big marked 4 (contact structure, two bodies per contact,
// A = [I 0], b = 0)
int primal_pin_dense_launch(const StepArgs& SP, const DenseIO& DN, int backward, void* stream) {
  const int n = 3 * SP.nb - SP.e;
  if (n <= 24) return backward ? pin_dense_launch<24, true>(SP, DN, stream) : pin_dense_launch<24, false>(SP, DN, stream);
  if (n == 30) return backward ? pin_dense_launch<32, true>(SP, DN, stream) : pin_dense_launch<30, false>(SP, DN, stream);
  if (n <= 32) return backward ? pin_dense_launch<32, true>(SP, DN, stream) : pin_dense_launch<32, false>(SP, DN, stream);
  return backward ? pin_dense_launch<40, true>(SP, DN, stream) : pin_dense_launch<40, false>(SP, DN, stream);
}
// pivots: the free coordinates.  One pinned body (three rows: its x, y and angle) - the count is a template argument of the kernel
bool primal_pin_supported(int nz, int e) { return e == 3 && nz - e <= 40; }
int primal_pin_launch(const StepArgs& SP, const StepBwdArgs& Gd, int backward, void* stream) {
  const int n = 3 * SP.nb - SP.e;
  if (n <= 24) return backward ? pin_launch<24, true>(SP, Gd, stream) : pin_launch<24, false>(SP, Gd, stream);
  if (n == 30) return backward ? pin_launch<32, true>(SP, Gd, stream) : pin_launch<30, false>(SP, Gd, stream);   // (BASELINE config 5: exactly its ten free bodies)
  if (n <= 32) return backward ? pin_launch<32, true>(SP, Gd, stream) : pin_launch<32, false>(SP, Gd, stream);
  return backward ? pin_launch<40, true>(SP, Gd, stream) : pin_launch<40, false>(SP, Gd, stream);
}

}  // namespace lcp
