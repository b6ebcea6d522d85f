// lcp_kernels.h - internal launch-argument structs shared by the kernel translation units and
// the C-ABI layer (lcp_api.cpp).  Plain C++: no HIP types, no torch types.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/lcp_hip.h"

namespace lcp {

struct Plan {
  int ok;            // fits the LDS budget
  int ldT;           // leading dimension of T
  int t_in_lds;      // where the matrices live (lcp_generic.hip carve()): 1 = all in LDS, 0 = T = R + diag(s/z) and the prefactor scratch in the workspace, 2 = Q, Q^-1, G there too
  size_t lds_bytes;  // dynamic LDS per workgroup
  size_t ws_stride;  // workspace elements (compute precision) per scene
};

struct FwdArgs {
  int B, nz, m, e;
  const void *Q, *p, *G, *h, *A, *b, *F;
  void *x, *y, *z, *s;
  int32_t* iters;
  int32_t* status;
  void* ws;
  size_t ws_stride;
  double eps;
  int max_iter, lim;
  int ldT, t_in_lds;
  double* trace;     // optional [B, max_iter, 4] (resid, mu, sigma, alpha) - debugging aid
  const int32_t* cls;  // optional per-scene class (lcp_classify_big): the generic kernels leave scenes of class 2 alone
  int32_t* tag;        // workspace trailer word: every forward kernel writes `tag_value` (kernel family + layout) there
  int tag_value;
};

struct BwdArgs {
  int B, nz, m, e;
  const void *G, *A, *dl_dx;
  void *dQ, *dp, *dG, *dh, *dA, *db, *dF;
  void* ws;
  size_t ws_stride;
  int ldT, t_in_lds;
  const int32_t* cls;  // as FwdArgs::cls
  const int32_t* tag;  // workspace trailer word the forward left; a backward that finds another value than `tag_value` (a workspace
  int tag_value;       // laid out by another kernel family) returns NaN gradients instead of reading it
  int skip_tag;        // 0, or the tag of the OTHER kernel family the forward may have fallen back on (wave64 step -> generic step
                       // when contact counts are given): a backward that finds it leaves without writing, its partner serves the call
  int adjoint;         // LCP_BWD_ADJOINT (generic kernels): factor T^T - the solve with K^T instead of the reference's K
  int split;           // lcp_bwd_quad<..., BODY> (round 6): 1 = leave dG and dF to lcp_bwd_stream_quad - the solve kernel hands (x, dx, lam, dlam)
                       // over in the scene's workspace block and writes the small gradients only
};

// dense (Q, p, G, h, A, b, F) boundary of lcp_big.hip: LCPFunction sizes beyond the wave-per-scene kernels (nineq <= 256)
// where lcp_classify_big leaves the per-contact records of a contact-structured dense scene for the body-space kernels (bytes into the
// scene's workspace block; behind lcp_primal's 6.3 KB iterate block, 4 KB long at 64 contacts - the block is at least 32 KB at these sizes)
constexpr size_t DENSE_EXTRACT_OFF = 8192;

struct DenseIO {
  int nz, m;
  int32_t* tag;                                     // workspace trailer word (see FwdArgs::tag)
  int tag_value;
  const float *Q, *p, *G, *h, *A, *b, *F;           // forward inputs
  float *x, *y, *z, *s;                             // forward outputs
  int32_t *iters, *status;
  int32_t* cls;                                     // [B] class per scene (2 = contact-structured with diagonal Q)
  size_t ws_scene;                                  // workspace bytes per scene (common to the kernel families of a batch)
  const float* dl_dx;                               // backward: cotangent, then the seven gradients (any may be NULL)
  float *dQ, *dp, *dG, *dh, *dA, *db, *dF;
};

struct StepArgs {
  int B, nb, nc, e;
  const void *pos, *Mdiag, *v, *f, *rest, *fric, *c_n, *c_p1, *c_p2;
  const int32_t *c_i1, *c_i2;
  const int32_t* c_count;   // per-scene live contacts (<= nc) or NULL = nc everywhere
  const void* Je;
  double dt;
  double eps;
  int max_iter, lim;
  void *v_new, *p_new, *z, *s, *y;
  int32_t* iters;
  int32_t* status;
  void* ws;
  size_t ws_stride;
  int ldT, t_in_lds;
  // post-stabilisation (generic_post_stab only): pose before / after the correction move and the dt every scene used
  const double* pos64;
  const double* dt_scene;
  double* p_out64;
  int32_t* tag;             // workspace trailer word (see FwdArgs::tag): written by the forward kernels, compared by the backward ones
  int tag_value;
};

struct StepBwdArgs {
  const void* dl_dv;                                   // [B, nb, 3]  d(loss)/d(v_new)
  void *dMdiag, *dv, *df, *drest, *dfric, *dcn, *dcp1, *dcp2;
  void* dJe;                                           // [B, e, 3 nb]  d(loss)/d(Je) = dnu (x) x + nu (x) dx (lcp.py:57), or NULL
};

struct ContactArgs {
  int B, nb, maxc;
  const int32_t *kind, *nverts;         // [B, nb]  0 = circle, 1 = hull ; vertex count of a hull
  const double *radius, *verts_local;   // [B, nb], [B, nb, 8, 2] (body frame)
  const uint8_t* no_contact;            // [B, nb, nb] pairs to skip, or NULL
  const double* p_start;                // [B, nb, 3] (rot, x, y)
  const float* v;                       // [B, nb, 3] or NULL (detect at p_start)
  double dt, dt_floor, eps, tol;
  int strict, max_trials;
  double* p_out;
  float *c_n, *c_p1, *c_p2;
  double* c_pen;
  int32_t *c_i1, *c_i2, *count;
  double *max_pen, *dt_used, *t;
  int32_t* trials;
};

// generic (any size) path - lcp_generic.hip
Plan make_plan(int nz, int m, int e, int csize);
int generic_forward(const FwdArgs& P, int io_f64, int compute, size_t lds, void* stream);
int generic_backward(const BwdArgs& P, int io_f64, int compute, size_t lds, void* stream);
int generic_step_backward(const StepArgs& P, const StepBwdArgs& Gd, int compute, size_t lds, void* stream);   // lcp_step_backward_f32 at any size of the generic plan (round 6)
int generic_post_stab_backward(const StepArgs& P, const StepBwdArgs& Gd, int compute, size_t lds, void* stream);   // lcp_post_stabilization_backward_f32, same sizes
int generic_post_stab(const StepArgs& P, int compute, size_t lds, void* stream);
int generic_step(const StepArgs& P, int compute, size_t lds, void* stream);
int generic_assemble(const StepArgs& P, float* Q, float* p, float* G, float* h, float* A, float* b,
                     float* F, void* stream);

// wave-per-scene register-resident path (nz <= 16, nineq <= 64, neq <= 8, fp32 I/O) - lcp_wave64.hip
bool wave64_supported(int nz, int m, int e);
size_t wave64_ws_bytes(int compute, int io_f64 = 0);
int wave64_forward(const FwdArgs& P, int compute, void* stream, int io_f64 = 0, int body_space = 0);   // body_space: see quad_forward
int wave64_backward(const BwdArgs& P, int compute, bool all_quad, void* stream, int io_f64 = 0, int body_space = 0);
int wave64_step(const StepArgs& P, int compute, void* stream);

// body-space (primal) contact-structured path: one wave per scene, <= 64 contacts, nz + neq <= 56 - lcp_primal.hip
bool primal_supported(int nz, int m, int e);          // contact-list entry points: up to 24 equality rows (nz + neq <= 56), up to 64 rows with at most 4 (round 6)
bool primal_poststab_supported(int nz, int m, int e); // post-stabilisation: nz + neq <= 56
bool primal_dense_supported(int nz, int m, int e);    // dense boundary, post-stabilisation: up to 4
size_t primal_ws_bytes();
int primal_step(const StepArgs& P, void* stream, bool pinned = false);      // pinned: LCP_HINT_PINNED (lcp_primal_pin.hip)
int primal_step_backward(const StepArgs& P, const StepBwdArgs& G, void* stream, bool pinned = false);
bool primal_pin_supported(int nz, int e);             // lcp_primal_pin.hip: nz - neq pivots when the equality rows pin the leading coordinates
int primal_pin_launch(const StepArgs& P, const StepBwdArgs& G, int backward, void* stream);
int primal_pin_dense_launch(const StepArgs& P, const DenseIO& DN, int backward, void* stream);   // dense boundary, scenes of class 4
int primal_chain_launch(const StepArgs& P, const StepBwdArgs& G, int backward, void* stream);   // lcp_primal_chain.hip: 5 .. 24 equality rows
int primal_post_stab_backward(const StepArgs& P, const StepBwdArgs& G, void* stream);   // lcp.py:37-64 on that LCP, contracted through engines.py:84-112
int primal_post_stab(const StepArgs& P, void* stream);                                         // engines.py:80-116 in body space
int primal_dense_forward(const FwdArgs& P, int32_t* cls, size_t ws_scene, void* stream);      // scenes of class 3 (and 4: the pinned form)
int primal_dense_backward(const BwdArgs& P, int32_t* cls, size_t ws_scene, void* stream);

// four-scenes-per-wave contact-structured path (nc <= 16, neq <= 4, diagonal Q; nz <= 16, or nz <= 32 from a contact
// list) - lcp_quad.hip
// `accept`: classification flag value (workspace meta[0]) the launch serves
bool quad_post_supported(int nz, int m, int e);        // post-stabilisation on this mapping (m = 4 maxc as everywhere; the LCP has maxc rows)
int quad_post_stab(const StepArgs& P, void* stream);
bool quad_supported(int nz, int m, int e);
bool quad_step_supported(int nz, int m, int e);   // contact-list entry points: nz <= 32
int quad_forward(const FwdArgs& P, int compute, int accept, void* stream, int io_f64 = 0, int body_space = 0);   // body_space: the dense boundary on the body-space kernels (fp32 tensors, fp64 arithmetic)
bool quad_dense_is_body_space(int io_f64, int compute, int body_space);   // does quad_forward run the body-space kernels for these arguments ?
int quad_backward(const BwdArgs& P, int compute, int accept, void* stream, int io_f64 = 0, int body = 0, bool pinned = false);   // body: workspace of a body-space forward; pinned: LCP_HINT_PINNED
int quad_step(const StepArgs& P, int compute, void* stream, int body_space = 1, int solo = -1, bool pinned = false);   // solo: -1 by batch size, 0 never, 1 always; pinned: LCP_HINT_PINNED
// one scene per wavefront, small batches (the body-space sizes with nz <= 16) - lcp_solo.hip
bool solo_supported(int nz, int m, int e);
int solo_step(const StepArgs& P, void* stream, bool pinned = false);
int quad_step_backward(const StepArgs& P, const StepBwdArgs& G, int compute, void* stream, int body_space = 1, bool pinned = false);
bool quad_step_is_body_space(int nz, int compute, int body_space);   // does quad_step run the body-space kernels for these arguments ?

// workgroup-per-scene contact-structured forward for up to 64 contacts (fused step, forward only) - lcp_big.hip
bool big_supported(int nz, int m, int e);
size_t big_ws_bytes(int m);
int big_step(const StepArgs& P, void* stream);
int big_step_backward(const StepArgs& P, const StepBwdArgs& G, void* stream);
// dense boundary (lcp_pdipm_forward_f32 / _backward_f32) for 16 < nineq / 4 <= 64 contacts: classification, then the same kernel
bool big_dense_supported(int nz, int m, int e);
int big_dense_forward(const FwdArgs& P, int32_t* cls, size_t ws_scene, int primal_ok, void* stream);   // (classifies: 0 / 2 / 3 / 4; primal_ok: bit 0 = lcp_primal sizes, bit 1 = pinned sizes)
int big_dense_backward(const BwdArgs& P, int32_t* cls, size_t ws_scene, void* stream);

// narrow-phase contact generation + position update - lcp_contacts.hip
int contacts_launch(const ContactArgs& P, void* stream);
int joint_jacobian_launch(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2, const double* jr1,
                          double* jrot1, const double* p, const float* v, const double* dt_scene, double dt, double vscale, float* Je,
                          void* stream);
int joint_jacobian_backward_launch(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                                   const double* jr1, const double* jrot1, const float* gJe, double* g_p, double* g_rot, void* stream);
int state_update_backward_launch(int B, int nb, int nj, const double* g_p, const double* g_g, const double* g_rot, const float* v,
                                 const double* dt_scene, double scale, const int32_t* jtype, const int32_t* jb1, float* g_v, void* stream);
int contact_frame_backward_launch(int B, int nb, int maxc, const int32_t* kind, const double* radius, const double* verts_local,
                                  const int32_t* nverts, const uint8_t* no_contact, const double* p, double eps,
                                  const int32_t* count, const float* g_n, const float* g_p1, const float* g_p2, double* dp,
                                  void* stream);

}  // namespace lcp
