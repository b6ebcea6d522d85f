// lcp_quad_n15e3.hip - the pinned body-space kernels of lcp_quad.hip with nz = 15, neq = 3 known at compile time
// (a floor and four bodies: BASELINE configs[2], [3]; see lcp_quad_sized.inc)
#define LCP_QS_NZ 15
#define LCP_QS_E 3
#define LCP_QS_NC 16
#include "lcp_quad_sized.inc"
