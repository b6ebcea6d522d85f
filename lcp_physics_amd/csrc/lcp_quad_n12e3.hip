// lcp_quad_n12e3.hip - the pinned body-space kernels of lcp_quad.hip with nz = 12, neq = 3 known at compile time
// (a floor and three bodies; see lcp_quad_sized.inc)
#define LCP_QS_NZ 12
#define LCP_QS_E 3
#define LCP_QS_NC 12
#include "lcp_quad_sized.inc"
