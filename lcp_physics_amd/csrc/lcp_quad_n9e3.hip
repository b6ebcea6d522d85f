// lcp_quad_n9e3.hip - the pinned body-space kernels of lcp_quad.hip with nz = 9, neq = 3 known at compile time
// (a floor and two bodies: BASELINE configs[1]; see lcp_quad_sized.inc)
#define LCP_QS_NZ 9
#define LCP_QS_E 3
#define LCP_QS_NC 8
#include "lcp_quad_sized.inc"
