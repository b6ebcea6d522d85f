/*
 * lcp_hip.h - C ABI of the MI355X-native batched LCP contact solver (liblcp_hip.so).
 *
 * This is the drop-in boundary for the hot path of locuslab/lcp-physics.  The reference
 * is pure Python and has no FFI of its own; each entry point below names the reference
 * interface it replaces (paths relative to /root/reference/lcp_physics).  The reference-
 * side binding a maintainer would add (a ctypes stub inside lcp/lcp.py and
 * physics/engines.py) is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless stated otherwise; all tensors are
 *    row-major, contiguous, batch-major ([B, rows, cols]);
 *  - the caller owns every buffer, including the workspace (`lcp_workspace_bytes`);
 *    nothing is allocated, freed or synchronised inside; launches are ordered on `stream`
 *    (a hipStream_t passed as void*, NULL = the default stream);
 *  - return value: 0 = launched, <0 = invalid arguments / unsupported size (LCP_E_*);
 *    numerical trouble is NOT an error (the reference returns its best iterate silently,
 *    lcp/solvers/pdipm.py:99-102,133-136,176-179): it is reported per scene in `status`;
 *  - sizes: nz = number of primal variables, m = nineq (rows of G), e = neq (rows of A,
 *    may be 0: then A, b, y and their gradients may be NULL).
 *  - `compute`: arithmetic type used inside the kernels.  LCP_COMPUTE_F64 is the parity
 *    path (fp32 I/O, fp64 arithmetic); LCP_COMPUTE_F32 computes everything in fp32.
 *  - threading: the library holds no process-global mutable state.  Every launch is planned from its own arguments and
 *    ordered on the caller's stream of the CURRENT device, so host threads may drive different streams / devices
 *    concurrently; the two lcp_debug_* settings are per calling thread.
 */
#ifndef LCP_HIP_H
#define LCP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCP_COMPUTE_F32 0
#define LCP_COMPUTE_F64 1
/* May be OR-ed into the `compute` argument of lcp_pdipm_backward_f32: the workspace was left by a CONTACT-LIST forward
 * (lcp_step_fused_f32 / lcp_solve_dynamics_f32, sizes of the four-scenes-per-wave kernels: nz <= 16, <= 16 contacts, neq <= 4)
 * called with this same `compute` word - the dense backward (lcp.py:37-64) of a fused step.  One launch; it factors in body
 * space when that forward did (the workspace then holds no contact-space matrix at all). */
#define LCP_HINT_ALL_CONTACT 0x100
/* May be OR-ed into any `compute` argument: serve this call from the generic workgroup-per-scene kernels whatever the
 * sizes (A/B and debugging aid; a backward must carry the same flag as its forward - they share the workspace layout). */
#define LCP_PATH_GENERIC 0x200
/* Likewise (A/B aids): the contact-space kernels where the default is a body-space one; one wave per scene (lcp_primal) at every
 * size.  The kernel family of a call is a function of its sizes and its `compute` WORD: pass the word of the forward to the
 * backward and both plan the same workspace layout on any host threads.  The forward also leaves a layout tag in the workspace;
 * a backward planned for another layout returns NaN gradients instead of misreading it. */
#define LCP_PATH_CONTACT_SPACE 0x2000
#define LCP_PATH_PRIMAL 0x4000
/* Contact-list forwards of the four-scenes-per-wave sizes choose by batch size between four scenes per wavefront (lcp_quad.hip) and
 * one scene per wavefront (lcp_solo.hip, small batches); these force one of the two (A/B aids; same workspace layout, any backward
 * follows either). */
#define LCP_PATH_QUAD 0x8000
#define LCP_PATH_SOLO 0x10000
/* May be OR-ed into the `compute` argument of the contact-list forwards (lcp_step_fused_f32, lcp_solve_dynamics_f32): the caller
 * asserts that the equality rows of EVERY scene pin the leading coordinates - Je = [I 0], the TotalConstraint that fixes the floor of
 * the reference's demo worlds (physics/constraints.py:175-192) - or that there are none.  The four-scenes-per-wave family then skips
 * the launch that serves scenes with other equality rows (it finds nothing to do on such batches and costs 2-5 us); the
 * one-wave-per-scene family (17..64 contacts) forms and factors the free coordinates' system only (three pinned rows: 30 pivots
 * instead of 36 on BASELINE config 5).  The word travels with the op: the backward entry points take the same hint.  A scene that
 * breaks the promise is NOT solved: its new velocities are NaN and LCP_ST_NAN is set. */
#define LCP_HINT_PINNED 0x20000
/* OR-ed into the `compute` argument of lcp_workspace_bytes by callers of the fp64-I/O entry points (lcp_pdipm_forward_f64 /
 * lcp_pdipm_backward_f64): their workspace also keeps an fp64 copy of F. */
#define LCP_IO_F64 0x400

/* May be OR-ed into the `compute` argument of lcp_pdipm_backward_f32 / _f64 (round 6; SURVEY.md §0.5: "offer the adjoint-correct
 * backward as an opt-in flag, never as default"): solve with K^T instead of K.  The reference's LCPFunction.backward (lcp/lcp.py:46-50)
 * solves the KKT system itself - exact only where the LCP matrix F is symmetric, and the contact LCP's F is not (engines.py:69-73: +mu in
 * the cone rows, +1 / -1 between friction and cone rows).  K^T is K with F^T in place of F (every other block sits symmetrically), so the
 * transposed solve factors T^T = R^T + diag(s / z) from the R = G Q^-1 G^T - ... + F the forward left in the workspace.  Served by the
 * generic kernels only: the forward of such a call must have run with LCP_PATH_GENERIC (any other workspace layout returns NaN gradients
 * through the layout tag, as every mismatch does).  lcp_physics_amd.lcp.LCPFunction(adjoint_backward=True) sets both bits. */
#define LCP_BWD_ADJOINT 0x40000

#define LCP_E_BADARG   (-1)   /* null pointer / non-positive size                     */
#define LCP_E_TOOLARGE (-2)   /* problem does not fit the kernels' LDS/workspace plan (the generic kernels keep the matrices in the
                               * workspace when 160 KB of LDS do not hold them: e.g. 32 bodies x 128 contacts in fp64 plan; the limit
                               * is the vectors, ~(8 nz + 15 nineq + 9 neq) numbers in LDS)                                   */
#define LCP_E_LAUNCH   (-3)   /* hipLaunchKernel reported an error                     */

/* per-scene status bits written to `status[B]` */
#define LCP_ST_SINGULAR_Q   1  /* zero/NaN pivot while inverting Q   (pdipm.py:361-368 raises) */
#define LCP_ST_SINGULAR_S11 2  /* zero/NaN pivot in A Q^-1 A^T                                  */
#define LCP_ST_SINGULAR_T   4  /* exact zero pivot in LU(T): best iterate returned (pdipm.py:99-102) */
#define LCP_ST_NAN          8  /* the returned iterate contains NaN                             */
#define LCP_ST_TRUNCATED   16  /* c_count[k] > maxc: the scene was solved with its first maxc contacts only (the reference's
                                  list is unbounded, world.py:139-142) - enlarge maxc and redo the step      */

/* Library / build identification: returns a static string such as
 * "lcp_hip 0.1.0 gfx950".  Host-only, never touches the GPU. */
const char* lcp_version(void);

/* Bytes of workspace the forward/backward pair needs for a batch (caller allocates,
 * 256-byte aligned).  Holds what the reference keeps on the op instance between forward
 * and backward (lcp/lcp.py:28-29: Q_LU, S_LU, R, nus, lams, slacks): R, Q^-1, G Q^-1 A^T,
 * (A Q^-1 A^T)^-1 and the best iterate in compute precision.  Host-only. */
size_t lcp_workspace_bytes(int B, int nz, int m, int e, int compute);

/* Replaces LCPFunction.forward (lcp/lcp.py:22-35) = pdipm.pre_factor_kkt
 * (lcp/solvers/pdipm.py:357-408) + pdipm.forward (pdipm.py:49-179) with factor_kkt
 * (:414-454), solve_kkt (:325-354) and get_step (:182-186), per-scene (batch-1) semantics.
 *   in : Q[B,nz,nz] p[B,nz] G[B,m,nz] h[B,m] A[B,e,nz] b[B,e] F[B,m,m]
 *   out: x[B,nz] (zhats)  y[B,e] (nus)  z[B,m] (lams)  s[B,m] (slacks)
 *        iters[B]  (PDIPM loop iterations executed = factorisations inside the loop)
 *        status[B] (LCP_ST_* bits)
 * eps / max_iter / not_improved_lim: lcp/lcp.py:12-13 (1e-12, 10, 3). */
int lcp_pdipm_forward_f32(int B, int nz, int m, int e,
                          const float* Q, const float* p, const float* G, const float* h,
                          const float* A, const float* b, const float* F,
                          double eps, int max_iter, int not_improved_lim, int compute,
                          float* x, float* y, float* z, float* s,
                          int32_t* iters, int32_t* status, void* ws, void* stream);

/* Same, fp64 I/O and fp64 arithmetic (the reference's default dtype, physics/utils.py:34). */
int lcp_pdipm_forward_f64(int B, int nz, int m, int e,
                          const double* Q, const double* p, const double* G, const double* h,
                          const double* A, const double* b, const double* F,
                          double eps, int max_iter, int not_improved_lim,
                          double* x, double* y, double* z, double* s,
                          int32_t* iters, int32_t* status, void* ws, void* stream);

/* Replaces LCPFunction.backward (lcp/lcp.py:37-64): d = lams/slacks, factor_kkt, one
 * solve_kkt with rhs (dl_dx, 0, 0, 0), then the outer products.  Uses the workspace left
 * by the matching forward call (same B, nz, m, e, compute).  The reference solves with K,
 * not K^T (exact only for symmetric F); that formula is reproduced.
 *   in : G[B,m,nz] A[B,e,nz] dl_dx[B,nz]
 *   out: dQ[B,nz,nz] dp[B,nz] dG[B,m,nz] dh[B,m] dA[B,e,nz] db[B,e] dF[B,m,m]
 *        (any output pointer may be NULL to skip that gradient)
 * At an iterate that converged to rounding (d_i = lams_i / slacks_i of 1e16 and more) the matrix of lcp.py:46 is singular to working
 * precision wherever contact points are redundant; an elimination that meets a pivot of rounding noise there (or an exact zero) is
 * repeated with slacks_i / lams_i floored at 1e-12 x the row's diagonal of G Q^-1 G^T - the body-space kernels floor always and refine
 * once - so that dl/dp, dQ, dA, db stay the gradients of the converged solution instead of the cancellation error of multipliers of
 * 1e15 (the reference's own pivoted LU has that exposure; tests/parity.py::own_iterate_backward is the gate).
 * Alignment (m of 65 .. 256 rows, the dense route of 17 .. 64 contacts): scenes the forward classified as contact LCPs with at most two
 * bodies per contact are served by kernels that move F, dG and dF in 16-byte pieces.  The forward only hands that class out when F and G
 * are 16-byte aligned (torch allocations are; a view with an odd storage offset is not - such a call is solved by the contact-space
 * kernels instead); the backward of a forward that did returns LCP_E_BADARG for a dG or dF that is not 16-byte aligned.  For those scenes
 * the backward reads the Jacobian rows from the records the forward left in the workspace: the G passed here must be the forward's G. */
int lcp_pdipm_backward_f32(int B, int nz, int m, int e,
                           const float* G, const float* A, const float* dl_dx, int compute,
                           float* dQ, float* dp, float* dG, float* dh,
                           float* dA, float* db, float* dF,
                           void* ws, void* stream);

int lcp_pdipm_backward_f64(int B, int nz, int m, int e,
                           const double* G, const double* A, const double* dl_dx,
                           double* dQ, double* dp, double* dG, double* dh,
                           double* dA, double* db, double* dF,
                           void* ws, void* stream);

/* Replaces the contact branch of PdipmEngine.solve_dynamics (physics/engines.py:26-78)
 * for B independent scenes: assembles u = M v + dt f (engines.py:31-32), Jc / Jf / mu / E /
 * restitutions (physics/world.py:144-234) into the dense (Q,p,G,h,A,b,F) the reference hands
 * to LCPFunction (engines.py:67-76).  nb bodies (3 DoF each, nz = 3 nb), nc contacts
 * (m = 4 nc, two friction directions), e joint rows.
 *   in : Mdiag[B,nb,3] v[B,nb,3] f[B,nb,3] rest[B,nb] fric[B,nb]
 *        c_n[B,nc,2] c_p1[B,nc,2] c_p2[B,nc,2] c_i1[B,nc] c_i2[B,nc] (int32)  Je[B,e,nz]
 *   out: Q[B,nz,nz] p[B,nz] G[B,m,nz] h[B,m] A[B,e,nz] b[B,e] F[B,m,m] */
int lcp_assemble_contacts_f32(int B, int nb, int nc, int e,
                              const float* Mdiag, const float* v, const float* f,
                              const float* rest, const float* fric,
                              const float* c_n, const float* c_p1, const float* c_p2,
                              const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                              float* Q, float* p, float* G, float* h, float* A, float* b, float* F,
                              void* stream);

/* One fused simulation step for B scenes in a single launch: the assembly above, the LCP
 * solve (as lcp_pdipm_forward_f32), new_v = -x (engines.py:76-77) and the semi-implicit
 * integrator p <- p + new_v dt of Body.move (physics/bodies.py:80-82).  Dense LCP data
 * never leaves the chip.  Leaves the same workspace as the forward so that
 * lcp_pdipm_backward_f32 can follow (with G from lcp_assemble_contacts_f32).
 *   out: v_new[B,nb,3]  p_new[B,nb,3]  z[B,m]  s[B,m]  y[B,e]  iters[B]  status[B]
 *        z, s, y may be NULL: the multipliers are then not written out (the reference's step returns new_v only, engines.py:76-77;
 *        the backward reads them, in fp64, from the workspace either way). */
int lcp_step_fused_f32(int B, int nb, int nc, int e,
                       const float* pos, const float* Mdiag, const float* v, const float* f,
                       const float* rest, const float* fric,
                       const float* c_n, const float* c_p1, const float* c_p2,
                       const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                       double eps, int max_iter, int not_improved_lim, int compute,
                       float* v_new, float* p_new, float* z, float* s, float* y,
                       int32_t* iters, int32_t* status, void* ws, void* stream);

/* Backward of lcp_step_fused_f32 / lcp_solve_dynamics_f32 with respect to the PHYSICAL inputs of the step: what the
 * reference obtains by autograd through PdipmEngine.solve_dynamics (physics/engines.py:31-32,50-77) and the World
 * Jacobian builders (physics/world.py:144-234) on top of LCPFunction.backward (lcp/lcp.py:37-64).  The rank-1 LCP
 * gradients (dQ, dp, dG, dh, dF of lcp.py:52-61) are contracted on chip and never written.  Must follow the forward on
 * the same stream with the same workspace and the same (unchanged) inputs.
 *   in : the inputs of the forward, dl_dv[B,nb,3] = d(loss)/d(v_new)
 *   out: dMdiag[B,nb,3] dv[B,nb,3] df[B,nb,3] drest[B,nb] dfric[B,nb] dc_n[B,nc,2] dc_p1[B,nc,2] dc_p2[B,nc,2]
 *        (any may be NULL; padded contact slots get 0).  The joint Jacobian Je is treated as a constant here: see
 *        lcp_step_backward_je_f32 for its gradient.
 * Sizes: 3 nb <= 16, nc <= 16, e <= 4 after either forward; 3 nb <= 32 with nc <= 16, and up to nc <= 64, 3 nb <= 43,
 * e <= 4 (fp64 arithmetic), after lcp_solve_dynamics_f32 only (its kernels own the workspace layout); 5 <= e <= 24 equality rows
 * (chains of joints) with nc <= 64 and 3 nb + e <= 56, or e <= 4 with 3 nb + e <= 64 (18 .. 20 bodies; round 6) (fp64 arithmetic) after either
 * forward; (round 6) every other size the generic
 * kernels step forward - more than 64 contacts, 3 nb + e > 56, fp32 arithmetic beyond 16 contacts: lcp_step_bwd_kernel on the iterate
 * lcp_step_kernel leaves, each scene at its own contact count - after either forward, in either arithmetic.  LCP_E_TOOLARGE only for
 * the wave64 step family (fp32 arithmetic, 3 nb <= 16, 5 <= e <= 8: its kernel keeps no iterate) and beyond the generic plan. */
int lcp_step_backward_f32(int B, int nb, int nc, int e,
                          const float* Mdiag, const float* v, const float* f,
                          const float* rest, const float* fric,
                          const float* c_n, const float* c_p1, const float* c_p2,
                          const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                          const float* dl_dv, int compute,
                          float* dMdiag, float* dv, float* df, float* drest, float* dfric,
                          float* dc_n, float* dc_p1, float* dc_p2, void* ws, void* stream);

/* Host-only: 1 when lcp_step_backward_f32 / _je_f32 can follow a contact-list forward (lcp_step_fused_f32,
 * lcp_solve_dynamics_f32) called with these sizes and this `compute` word - since round 6 every size the forward itself accepts
 * except the wave64 step family (fp32 arithmetic, 3 nb <= 16, 5 <= e <= 8) -, else 0: the backward entry points return
 * LCP_E_TOOLARGE there.  The reference differentiates any size through LCPFunction.backward (lcp/lcp.py:37-64): the dense boundary
 * (lcp_physics_amd/physics/dense_step.py) is what the host falls back on for a 0. */
int lcp_step_has_backward(int nb, int maxc, int e, int compute);

/* The same with the gradient of the joint Jacobian as a ninth output: dJe[B,e,3 nb] = dnu (x) x + nu (x) dx (lcp.py:57, A = Je
 * in engines.py:75) - what the reference back-propagates into Joint.J() / FixedJoint.J() (constraints.py:26-36, 64-73: the
 * anchor arms follow the bodies' poses) when a world with joints is differentiated through its steps.  dJe may be NULL. */
int lcp_step_backward_je_f32(int B, int nb, int nc, int e,
                             const float* Mdiag, const float* v, const float* f,
                             const float* rest, const float* fric,
                             const float* c_n, const float* c_p1, const float* c_p2,
                             const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                             const float* dl_dv, int compute,
                             float* dMdiag, float* dv, float* df, float* drest, float* dfric,
                             float* dc_n, float* dc_p1, float* dc_p2, float* dJe, void* ws, void* stream);

/* Replaces PdipmEngine.solve_dynamics (physics/engines.py:26-78) for B scenes whose contact lists have
 * DIFFERENT lengths (what contact detection produces): scene k uses the first c_count[k] <= maxc records of
 * its padded contact list and solves the mixed LCP of exactly that size (nineq = 4 c_count[k], engines.py:51-76);
 * a scene without contacts takes the direct KKT solve of engines.py:36-50.  One launch, no position update
 * (that is lcp_move_find_contacts_f64).  Arguments as lcp_step_fused_f32; z, s use the row layout of a
 * maxc-contact LCP ([normal | friction pairs | gamma] blocks of maxc, 2 maxc, maxc rows), padded slots are 0.
 * Served by the four-scenes-per-wave kernel when 3 nb <= 32, maxc <= 16, e <= 4: the workspace it leaves then feeds
 * lcp_step_backward_f32 (padded slots get zero gradients) and, for 3 nb <= 16, lcp_pdipm_backward_f32 (m = 4 maxc).
 * Larger scenes - maxc <= 64 with 3 nb + e <= 56 and e <= 24 (chains of joints: two rows per revolute joint), with 3 nb + e <= 64 and
 * e <= 4 (up to 20 bodies on a pinned floor: the 64-row instantiation, every lane of the wave a row; round 6), or 3 nb <= 43,
 * e <= 4 - run (fp64 arithmetic) on the wave-per-scene body-space kernel (BASELINE config 5) or the workgroup-per-scene
 * contact-space kernel, and can be followed by lcp_step_backward_f32 (not by the dense backward); anything else runs on the
 * generic kernels (LCP_E_TOOLARGE beyond their LDS / workspace plan), which keep each scene's iterate and its contact count for
 * lcp_step_backward_f32 as well (round 6).
 *   out: v_new[B,nb,3]  z[B,4 maxc]  s[B,4 maxc]  y[B,e]  iters[B]  status[B] */
int lcp_solve_dynamics_f32(int B, int nb, int maxc, int e, const int32_t* c_count,
                           const float* Mdiag, const float* v, const float* f,
                           const float* rest, const float* fric,
                           const float* c_n, const float* c_p1, const float* c_p2,
                           const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                           double eps, int max_iter, int not_improved_lim, int compute,
                           float* v_new, float* z, float* s, float* y,
                           int32_t* iters, int32_t* status, void* ws, void* stream);

/* Replaces PdipmEngine.post_stabilization (physics/engines.py:80-116) and the correction move World.step_dt makes with
 * its result (physics/world.py:109-117), for B scenes in one launch.  Per scene the frictionless LCP
 *   Q = M, p = 0, G = Jc (world.py:172-184), h = gc = Jc v + Jc v * -restitutions (:87-89), A = Je, b = ge = Je v (:86),
 *   F = 0 (:110), solver defaults (lcp.py:12-13: pass eps = 1e-12, max_iter = 10, not_improved_lim = 3)
 * over the first c_count[k] entries of the padded contact list; dp = -x (:115).  A scene without contacts takes the
 * direct solve [[M, -Je^T], [Je, 0]]^-1 [0; ge] (:92-103).  `v` are the velocities AFTER the dynamics solve and the
 * contacts those found after the move (world.py:87-94).  When p / p_out are given, p_out = p + (dp / 2) dt_k with
 * dt_k = dt_scene[k] (the dt the scene's step ended up using; NULL: the scalar `dt`) - world.py:110-117; the caller
 * re-detects contacts at p_out (world.py:121: lcp_move_find_contacts_f64 with v = NULL); p_out may be p itself.
 * Runs on the wave-per-scene body-space kernel (fp64 arithmetic, maxc <= 64, e <= 24, 3 nb + e <= 56, or e <= 4 with 3 nb + e <= 64; it leaves its best
 * iterate in the workspace for lcp_post_stabilization_backward_f32) or on the workgroup-per-scene generic kernels (any size
 * their plan takes: LCP_E_TOOLARGE beyond); workspace of lcp_workspace_bytes(B, 3 nb, 4 maxc, e, compute).
 *   out: dp[B,nb,3]  p_out[B,nb,3] (optional)  iters[B]  status[B] */
int lcp_post_stabilization_f32(int B, int nb, int maxc, int e, const int32_t* c_count,
                               const float* Mdiag, const float* v, const float* rest,
                               const float* c_n, const float* c_p1, const float* c_p2,
                               const int32_t* c_i1, const int32_t* c_i2, const float* Je,
                               double eps, int max_iter, int not_improved_lim, int compute,
                               const double* p, const double* dt_scene, double dt, double* p_out,
                               float* dp, int32_t* iters, int32_t* status, void* ws, void* stream);

/* Backward of lcp_post_stabilization_f32 with respect to its physical inputs - what the reference obtains by autograd
 * through PdipmEngine.post_stabilization (engines.py:80-116: ge = Je v, gc = Jc v + Jc v * -restitutions, the LCPFunction
 * call and its backward lcp.py:37-64, dp = -x) when a World with post_stab=True is differentiated (experiments/inference.py).
 * Must follow the forward on the same stream with the same workspace and unchanged inputs.  Body-space kernel where the forward
 * ran there (fp64 arithmetic, maxc <= 64, e <= 24, 3 nb + e <= 56, or e <= 4 with 3 nb + e <= 64), else (round 6) the generic kernels on the iterate
 * lcp_post_stab_kernel keeps - any size of their plan, either arithmetic; LCP_E_TOOLARGE beyond it.
 *   in : the forward's inputs, dl_ddp[B,nb,3] = d(loss)/d(dp)
 *   out: dMdiag[B,nb,3] dv[B,nb,3] drest[B,nb] dc_n[B,maxc,2] dc_p1[B,maxc,2] dc_p2[B,maxc,2] dJe[B,e,3nb]
 *        (any may be NULL; padded contact slots get 0) */
int lcp_post_stabilization_backward_f32(int B, int nb, int maxc, int e,
                                        const float* Mdiag, const float* v, const float* rest,
                                        const float* c_n, const float* c_p1, const float* c_p2,
                                        const int32_t* c_i1, const int32_t* c_i2, const float* Je,
                                        const float* dl_ddp, int compute,
                                        float* dMdiag, float* dv, float* drest,
                                        float* dc_n, float* dc_p1, float* dc_p2, float* dJe,
                                        void* ws, void* stream);

/* Host-only: 1 when lcp_post_stabilization_backward_f32 can follow lcp_post_stabilization_f32 at these sizes and this `compute`
 * word - every size the forward accepts, since round 6 -, else 0 (the host would then differentiate the correction through the dense
 * boundary, the way engines.py:80-116 itself does: lcp_physics_amd/physics/dense_step.py). */
int lcp_post_stabilization_has_backward(int nb, int maxc, int e, int compute);

/* Replaces the position update of World.step_dt (physics/world.py:88-101,122) together with the
 * contact generation it calls, for B independent scenes in one launch:
 *   Body.move (physics/bodies.py:80-82)         p_try = p_start + v dt
 *   World.find_contacts (world.py:139-142)      all body pairs i < j (the reference delegates the
 *                                               broadphase to ODE; `no_contact[B,nb,nb]` != 0 skips a pair)
 *   DiffContactHandler.__call__ (physics/contacts.py:57-205) circle/circle, circle/hull (GJK + SAT),
 *                                               hull/hull (SAT, incident edge, clipping), with its helpers
 *                                               (contacts.py:207-352) and rotate_verts (bodies.py:211-214)
 *   the penetration test and dt halving (world.py:95-101): while any contact penetrates by more than
 *   `tol`, dt <- dt / 2 and retry from p_start; `strict` = strict_no_penetration, `dt_floor` = world.dt / 4
 *   (non-strict worlds accept once dt < dt_floor).  The loop is per scene and runs on the device;
 *   `max_trials` bounds it (the reference would spin forever on a pose that penetrates at dt -> 0).
 * Geometry and poses are fp64 (tol = 1e-6 against coordinates of several hundred cannot be resolved in
 * fp32); the contact frame handed to the LCP kernels is fp32.
 *   in : kind[B,nb] (0 circle, 1 hull)  radius[B,nb]  verts_local[B,nb,8,2] (body frame, CCW as the
 *        reference's Hull.verts)  nverts[B,nb]  p_start[B,nb,3] (rot,x,y)  v[B,nb,3] (NULL: detect at p_start)
 *   out: p_out[B,nb,3]  c_n/c_p1/c_p2[B,maxc,2]  c_pen[B,maxc]  c_i1/c_i2[B,maxc]  count[B] (contacts found,
 *        in the reference's order; > maxc means the list was truncated)  max_pen[B]  dt_used[B]
 *        t[B] (+= dt_used)  trials[B]   (c_pen, max_pen, dt_used, t, trials, p_out may be NULL)
 * nb <= 32, hulls of <= 8 vertices. */
int lcp_move_find_contacts_f64(int B, int nb, int maxc,
                               const int32_t* kind, const double* radius, const double* verts_local,
                               const int32_t* nverts, const uint8_t* no_contact,
                               const double* p_start, const float* v,
                               double dt, double dt_floor, int strict, int max_trials,
                               double eps, double tol,
                               double* p_out, float* c_n, float* c_p1, float* c_p2, double* c_pen,
                               int32_t* c_i1, int32_t* c_i2, int32_t* count, double* max_pen,
                               double* dt_used, double* t, int32_t* trials, void* stream);

/* Replaces World.Je (physics/world.py:156-170) over the joints' J() (physics/constraints.py:13-217) and Joint.move /
 * update_pos (constraints.py:39-50) for B scenes: joints whose Jacobian follows the pose.
 *   jtype[B,nj]: 1 Joint (revolute, 2 rows)  2 FixedJoint (3 rows)  3 XConstraint  4 YConstraint  5 RotConstraint (1 row each)
 *                6 TotalConstraint (3 rows)  0 empty;   jb1 / jb2 [B,nj] body indices (jb2 = -1: no second body)
 *   jr1 / jrot1 [B,nj]: polar coordinates of a Joint's anchor relative to body 1 (cart_to_polar with the positive-angle rule,
 *                utils.py:75-82); jrot1 is state: with `v` != NULL it is first advanced by vscale * v[body1][0] * dt_k,
 *                dt_k = dt_scene[k] (NULL: `dt`) - the dt the scene's step accepted (world.py:88-107 restores the joints before
 *                every retry); vscale = 1 for the dynamics move, 0.5 for the post-stabilisation move (world.py:112).
 *   p[B,nb,3] the pose the Jacobian is wanted at;  out: Je[B,e,3 nb] float32, e = the rows the joint list adds up to. */
int lcp_joint_jacobian_f64(int B, int nb, int nj, int e,
                           const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                           const double* jr1, double* jrot1, const double* p,
                           const float* v, const double* dt_scene, double dt, double vscale,
                           float* Je, void* stream);

/* Backward of lcp_joint_jacobian_f64 with respect to the pose and the revolute joints' angles: what the reference obtains by autograd
 * through Joint.J() / FixedJoint.J() with update_pos (physics/constraints.py:26-50, 64-85) when a roll-out with joints is
 * back-propagated (experiments/inference.py:55-61).  jrot1[B,nj]: the angles the Jacobian was evaluated at (a copy taken then - the
 * forward advances its jrot1 in place);  gJe[B,e,3 nb] float32 = d(loss)/dJe (lcp_step_backward_je_f32's dJe).
 *   out: g_p[B,nb,3] float64 (written, not accumulated; only the x / y columns of the joints' bodies are non-zero), g_rot[B,nj]. */
int lcp_joint_jacobian_backward_f64(int B, int nb, int nj, int e,
                                    const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                                    const double* jr1, const double* jrot1, const float* gJe,
                                    double* g_p, double* g_rot, void* stream);

/* Backward of the state update that ends a differentiable step - Body.move (physics/bodies.py:80-82), the vertex turn it skips for a
 * zero rotation increment (bodies.py:199-202) and Joint.move (constraints.py:39-43) - with respect to the velocities moved by:
 *   p_new = p + scale v dt_k (cotangent g_p);  the pose the geometry is differentiated at, geo_new = p_geo + the same increment where
 *   it is non-zero or the coordinate is x / y (cotangent g_g);  rot_new = rot + scale v[body1][0] dt_k for revolute joints (g_rot).
 * g_p / g_g [B,nb,3] float64 and g_rot [B,nj] float64 may be NULL (no such cotangent);  v[B,nb,3] float32 the velocities of the move
 * (new_v, or the post-stabilisation dp with scale = 0.5, world.py:112);  dt_scene[B] > 0 the dt each scene's step accepted;
 * jtype / jb1 [B,nj] as in lcp_joint_jacobian_f64 (needed with g_rot).   out: g_v[B,nb,3] float32 = d(loss)/dv. */
int lcp_state_update_backward_f64(int B, int nb, int nj,
                                  const double* g_p, const double* g_g, const double* g_rot,
                                  const float* v, const double* dt_scene, double scale,
                                  const int32_t* jtype, const int32_t* jb1, float* g_v, void* stream);

/* Backward of the contact frame with respect to the poses: what the reference obtains by autograd through
 * DiffContactHandler (physics/contacts.py:57-352 - every operation of the contact tuple is a differentiable torch op,
 * including the rotated hull vertices of bodies.py:211-214), needed to back-propagate through a roll-out
 * (demos/grad_demo.py:45-50, experiments/inference.py:55-61).  Every record type: circle / circle, circle / hull (GJK or
 * SAT), hull / hull (SAT, incident edge, clipping); the derivative follows the branches the detection took (as autograd does).
 *   in : the geometry and the pose `p` lcp_move_find_contacts_f64 detected the contact list at (its p_out), its `eps`,
 *        count[B], and g_n / g_p1 / g_p2 [B,maxc,2] = d(loss)/d(c_n, c_p1, c_p2) (what lcp_step_backward_f32 returns)
 *   out: dp[B,nb,3] = d(loss)/d(pose) through the contact frame (overwritten).   nb <= 32. */
int lcp_contact_frame_backward_f64(int B, int nb, int maxc,
                                   const int32_t* kind, const double* radius, const double* verts_local,
                                   const int32_t* nverts, const uint8_t* no_contact,
                                   const double* p, double eps, const int32_t* count,
                                   const float* g_n, const float* g_p1, const float* g_p2,
                                   double* dp, void* stream);

/* ---- debugging / A-B aids (not part of the drop-in surface) ----
 * lcp_debug_set_trace: when non-NULL, the dense forward writes trace[B, max_iter, 4] =
 *   (resid, mu, sigma, alpha) per PDIPM iteration (device pointer to doubles).
 * lcp_debug_set_path : 0 = automatic kernel selection, 1 = generic (workgroup-per-scene) kernels only,
 *   2 = wave-per-scene kernels whenever the sizes allow (the default behaviour of 0 today),
 *   3 = contact-space factorisation everywhere: lcp_big.hip instead of lcp_primal.hip for 17..64 contacts, the 32-row reduced
 *       system instead of the body-space one in lcp_quad.hip's contact-list forward (the formulation of pdipm.py:325-454;
 *       same answers).  A forward and its backward must run under the same setting.
 *   4 = lcp_primal.hip (one wave per scene) also where lcp_quad.hip would serve (A/B: small batches).
 * Both settings are thread_local: they affect the calls of the thread that made them, nobody else's. */
void lcp_debug_set_trace(double* device_trace);
/* LCP_BWD_ADJOINT for lcp_pdipm_backward_f64 (that entry has no `compute` word): per calling thread, like the two settings above;
 * non-zero = the next fp64-I/O backward calls of this thread solve with K^T.  Reset it to 0 afterwards. */
void lcp_set_backward_adjoint(int on);
void lcp_debug_set_path(int path);

#ifdef __cplusplus
}
#endif
#endif /* LCP_HIP_H */
