// lcp_quad_n6e3.hip - the pinned body-space kernels of lcp_quad.hip with nz = 6, neq = 3 known at compile time
// (a floor and one body; see lcp_quad_sized.inc)
#define LCP_QS_NZ 6
#define LCP_QS_E 3
#define LCP_QS_NC 4
#include "lcp_quad_sized.inc"
