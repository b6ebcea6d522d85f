// lcp_api.cpp - the C ABI declared in include/lcp_hip.h: argument checking, launch planning and
// dispatch to the kernel translation units.  No torch types, no allocation, no synchronisation.
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "lcp_kernels.h"

namespace {

// The library keeps NO process-global mutable state: launches are planned from their arguments alone and ordered on the
// caller's stream, so any number of host threads may drive any number of streams / devices concurrently.  The two
// debugging aids below are per calling thread (thread_local), and the kernel family can also be forced per call with
// LCP_PATH_GENERIC in the `compute` word.
thread_local double* g_trace = nullptr;   // debugging aid, see lcp_debug_set_trace
thread_local int g_adjoint = 0;           // lcp_set_backward_adjoint: LCP_BWD_ADJOINT for the fp64-I/O backward (it has no `compute` word)
thread_local int g_path = 0;              // this thread's DEFAULT kernel path for calls whose `compute` word names none:
                                          // 0 = automatic, 1 = generic kernels, 3 = contact-space kernels instead of the body-space ones,
                                          // 4 = one wave per scene (lcp_primal.hip) at every size (A/B aids)

constexpr int FLAG_BITS = LCP_BWD_ADJOINT | LCP_PATH_GENERIC | LCP_HINT_ALL_CONTACT | LCP_IO_F64 | LCP_PATH_CONTACT_SPACE | LCP_PATH_PRIMAL | LCP_PATH_QUAD | LCP_PATH_SOLO |
                          LCP_HINT_PINNED;

// `compute` word of an entry point -> arithmetic type and kernel path.  The path is a function of the WORD whenever the word
// names one (LCP_PATH_*): a forward and its backward that carry the same word pick the same kernel family on any two host
// threads.  Only a word without path bits falls back on the calling thread's lcp_debug_set_path default.
// *path: 0 automatic, 1 generic, 3 contact space, 4 primal;  *generic = (path == 1)
inline int split_compute(int compute, bool* generic, int* path = nullptr, int* solo = nullptr) {
  // (same workspace layout either way: forward only; bit 2 of the value: LCP_HINT_PINNED)
  if (solo) *solo = ((compute & LCP_PATH_SOLO) ? 1 : ((compute & LCP_PATH_QUAD) ? 0 : -1)) + ((compute & LCP_HINT_PINNED) ? 16 : 0);
  int p = g_path;
  if (compute & LCP_PATH_GENERIC) p = 1;
  else if (compute & LCP_PATH_CONTACT_SPACE) p = 3;
  else if (compute & LCP_PATH_PRIMAL) p = 4;
  if (path) *path = p;
  *generic = p == 1;
  return compute & ~FLAG_BITS;
}

// Workspace trailer (one 256-byte block behind the scene blocks and the class words): tag[0] = which forward laid the workspace
// out.  Every forward kernel writes it, every backward kernel compares it with what ITS launch plan expects and returns NaN
// gradients on a mismatch (a backward planned for another kernel family would otherwise misread the layout silently).
enum WsTag {
  TAG_DENSE_WAVE = 1, TAG_DENSE_BIG = 2, TAG_DENSE_GENERIC = 3,                       // lcp_pdipm_forward_*
  TAG_STEP_QUAD_BODY = 4, TAG_STEP_QUAD_CS = 5, TAG_STEP_PRIMAL = 6, TAG_STEP_BIG = 7, TAG_STEP_WAVE64 = 8, TAG_STEP_GENERIC = 9,
  TAG_POSTSTAB_PRIMAL = 10, TAG_POSTSTAB_GENERIC = 11,
  TAG_DENSE_WAVE_BODY = 12                                                            // lcp_pdipm_forward_f32, class-2 scenes solved in body space (no W)
};
constexpr size_t TRAILER_BYTES = 256;

// Which kernel family serves a problem.  Deterministic in (sizes, io type) so that forward and
// backward of one op agree on the workspace layout.
inline bool use_wave64(int io_f64, int nz, int m, int e, bool generic) {
  if (generic) return false;
  return lcp::wave64_supported(nz, m, e);       // (fp64 I/O runs the same kernels with fp64 loads / stores)
}

// Kernel family of the contact-list entry points (lcp_step_fused_f32, lcp_solve_dynamics_f32, lcp_step_backward_f32):
// ONE function of (sizes, arithmetic, forced path), so that a backward always reads the workspace layout its forward wrote.
enum StepFamily { FAM_QUAD, FAM_PRIMAL, FAM_BIG, FAM_WAVE64, FAM_GENERIC };
inline StepFamily step_family(int nz, int m, int e, int compute, int path) {
  if (path == 1) return FAM_GENERIC;
  if (path == 4 && compute == LCP_COMPUTE_F64 && lcp::primal_supported(nz, m, e)) return FAM_PRIMAL;   // (A/B: one wave per scene at every size)
  if (lcp::quad_step_supported(nz, m, e)) return FAM_QUAD;                     // <= 16 contacts, <= 10 bodies, e <= 4
  if (compute == LCP_COMPUTE_F64 && path != 3 && lcp::primal_supported(nz, m, e)) return FAM_PRIMAL;   // <= 64 contacts, body-space systems
  if (compute == LCP_COMPUTE_F64 && lcp::big_supported(nz, m, e)) return FAM_BIG;   // <= 64 contacts (fp64 arithmetic)
  if (lcp::wave64_supported(nz, m, e)) return FAM_WAVE64;                      // nz <= 16, e 5..8
  return FAM_GENERIC;
}

inline int csize_of(int io_f64, int compute) { return (io_f64 || compute == LCP_COMPUTE_F64) ? 8 : 4; }

// The dense entry points serve 17 .. 64 contact LCPs (nineq <= 256) from lcp_big.hip when the scene has the contact structure
// (classified per scene on the device); the other scenes of the batch stay on the generic kernels.  Both families then share
// ONE per-scene workspace stride, and the per-scene classes live behind the B scene blocks.
inline bool use_big_dense(int io_f64, int nz, int m, int e, int compute, bool generic) {
  return !io_f64 && !generic && compute == LCP_COMPUTE_F64 && !lcp::wave64_supported(nz, m, e) && lcp::big_dense_supported(nz, m, e);
}
// workspace tag a contact-list forward of this family leaves (the quad family has two layouts: with and without W)
inline int step_tag(StepFamily fam, int nz, int compute, int path) {
  switch (fam) {
    case FAM_QUAD: return lcp::quad_step_is_body_space(nz, compute, path != 3) ? TAG_STEP_QUAD_BODY : TAG_STEP_QUAD_CS;
    case FAM_PRIMAL: return TAG_STEP_PRIMAL;
    case FAM_BIG: return TAG_STEP_BIG;
    case FAM_WAVE64: return TAG_STEP_WAVE64;
    default: return TAG_STEP_GENERIC;
  }
}

inline size_t scene_bytes(int nz, int m, int e, int compute, int io_f64) {
  const int cs = (compute == LCP_COMPUTE_F64) ? 8 : 4;
  lcp::Plan pl = lcp::make_plan(nz, m, e, cs);
  size_t per_scene = pl.ws_stride * cs;
  if (lcp::wave64_supported(nz, m, e) || lcp::quad_step_supported(nz, m, e)) {   // (lcp_quad.hip uses the wave64 layout)
    const size_t w = lcp::wave64_ws_bytes(compute, io_f64);
    if (w > per_scene) per_scene = w;
  }
  if (!lcp::quad_step_supported(nz, m, e) && lcp::big_supported(nz, m, e) && lcp::big_ws_bytes(m) > per_scene)
    per_scene = lcp::big_ws_bytes(m);      // (the sizes the quad kernel takes never reach lcp_big.hip)
  if (lcp::primal_supported(nz, m, e) && lcp::primal_ws_bytes() > per_scene) per_scene = lcp::primal_ws_bytes();
  return (per_scene + 15) & ~(size_t)15;
}
inline size_t cls_bytes_of(int B) { return (((size_t)B * sizeof(int32_t)) + 255) & ~(size_t)255; }
inline int32_t* trailer_of(void* ws, int B, size_t per_scene) { return (int32_t*)((unsigned char*)ws + (size_t)B * per_scene + cls_bytes_of(B)); }

}  // namespace

extern "C" {

const char* lcp_version(void) { return "lcp_hip 0.2.0 gfx950"; }

size_t lcp_workspace_bytes(int B, int nz, int m, int e, int compute) {
  if (B <= 0 || nz <= 0 || m <= 0 || e < 0) return 0;
  const int io_f64 = (compute & LCP_IO_F64) ? 1 : 0;
  compute &= ~FLAG_BITS;
  if (io_f64) compute = LCP_COMPUTE_F64;
  const size_t per_scene = scene_bytes(nz, m, e, compute, io_f64);
  // scene blocks | per-scene classes of the dense lcp_big path | trailer (the layout tag)
  return (size_t)B * per_scene + cls_bytes_of(B) + TRAILER_BYTES;
}

// Debugging / A-B aid: the calling thread's default path for calls whose `compute` word carries no LCP_PATH_* bit
// (0 automatic, 1 generic kernels, 3 contact-space kernels, 4 one wave per scene).  Prefer the per-call bits: they travel with
// the word from a forward to its backward, whatever threads the two run on.
void lcp_debug_set_path(int path) { g_path = path; }

// Debugging aid (not part of the drop-in surface): when set, the dense forward writes
// trace[B, max_iter, 4] = (resid, mu, sigma, alpha) per PDIPM iteration.  Pass NULL to disable.
void lcp_debug_set_trace(double* device_trace) { g_trace = device_trace; }
void lcp_set_backward_adjoint(int on) { g_adjoint = on ? 1 : 0; }

static int forward_common(int io_f64, int B, int nz, int m, int e, const void* Q, const void* p, const void* G,
                          const void* h, const void* A, const void* b, const void* F, double eps, int max_iter,
                          int lim, int compute, void* x, void* y, void* z, void* s, int32_t* iters,
                          int32_t* status, void* ws, void* stream) {
  if (B <= 0 || nz <= 0 || m <= 0 || e < 0 || max_iter < 0) return LCP_E_BADARG;
  if (!Q || !p || !G || !h || !F || !x || !z || !s || !ws) return LCP_E_BADARG;
  if (e > 0 && (!A || !b)) return LCP_E_BADARG;
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  const int cs = csize_of(io_f64, compute);
  const bool w64 = use_wave64(io_f64, nz, m, e, generic);
  lcp::Plan pl = lcp::make_plan(nz, m, e, cs);
  if (!w64 && !pl.ok) return LCP_E_TOOLARGE;
  const size_t per_scene_all = scene_bytes(nz, m, e, io_f64 ? LCP_COMPUTE_F64 : compute, io_f64);
  const bool bigd = use_big_dense(io_f64, nz, m, e, compute, generic);
  lcp::FwdArgs P;
  memset(&P, 0, sizeof(P));
  // the wave-per-scene sizes whose contact-structured scenes the four-scenes-per-wave kernels take: in body space unless the word
  // asks for the contact-space formulation (LCP_PATH_CONTACT_SPACE) - a function of (sizes, word), the same in the backward
  const int dense_body = (w64 && path != 3 && lcp::quad_supported(nz, m, e) && lcp::quad_dense_is_body_space(io_f64, compute, 1)) ? 1 : 0;
  P.tag = trailer_of(ws, B, per_scene_all);
  P.tag_value = w64 ? (dense_body ? TAG_DENSE_WAVE_BODY : TAG_DENSE_WAVE) : (bigd ? TAG_DENSE_BIG : TAG_DENSE_GENERIC);
  P.B = B; P.nz = nz; P.m = m; P.e = e;
  P.Q = Q; P.p = p; P.G = G; P.h = h; P.A = A; P.b = b; P.F = F;
  P.x = x; P.y = y; P.z = z; P.s = s; P.iters = iters; P.status = status;
  P.ws = ws; P.ws_stride = pl.ws_stride; P.eps = eps; P.max_iter = max_iter; P.lim = lim;
  P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds; P.trace = g_trace;
  if (w64) return lcp::wave64_forward(P, compute, stream, io_f64, dense_body);
  if (bigd) {
    const size_t per_scene = per_scene_all;
    int32_t* cls = (int32_t*)((unsigned char*)ws + (size_t)B * per_scene);
    // classes per scene: 4 = as 3 with equality rows that pin the leading coordinates (lcp_primal_pin.hip); 3 = contact structure, at
    // most two bodies per contact, sizes of lcp_primal.hip; 2 = contact structure (lcp_big.hip); 0 = anything else (the generic kernels)
    // (bit 1: the pinned form's sizes - lcp_classify_big then also looks at A and b and marks the scenes whose rows pin the leading coordinates 4)
    // (the body-space kernels read lcp_classify_big's per-contact records - 16 floats per contact, DENSE_EXTRACT_OFF into the scene's block -
    //  and move F, dG, dF with 16-byte accesses: both are preconditions of that route, checked here; without them the scenes stay with the
    //  contact-space / generic kernels, which make neither assumption)
    const bool rec_fits = lcp::DENSE_EXTRACT_OFF + (size_t)16 * (m / 4) * sizeof(float) <= per_scene;
    const bool aligned16 = (((uintptr_t)F | (uintptr_t)G) & 15) == 0;
    const int primal_ok = (path != 3 && rec_fits && aligned16 && (nz % 3) == 0 && lcp::primal_dense_supported(nz, m, e)) ? (1 | (lcp::primal_pin_supported(nz, e) ? 2 : 0)) : 0;
    int rc = lcp::big_dense_forward(P, cls, per_scene, primal_ok, stream);
    if (rc) return rc;
    if (primal_ok) { rc = lcp::primal_dense_forward(P, cls, per_scene, stream); if (rc) return rc; }
    P.cls = cls; P.ws_stride = per_scene / cs;                            // the rest of the batch, same stride
  }
  return lcp::generic_forward(P, io_f64, compute, pl.lds_bytes, stream);
}

int lcp_pdipm_forward_f32(int B, int nz, int m, int e, const float* Q, const float* p, const float* G,
                          const float* h, const float* A, const float* b, const float* F, double eps,
                          int max_iter, int not_improved_lim, int compute, float* x, float* y, float* z,
                          float* s, int32_t* iters, int32_t* status, void* ws, void* stream) {
  return forward_common(0, B, nz, m, e, Q, p, G, h, A, b, F, eps, max_iter, not_improved_lim, compute, x, y,
                        z, s, iters, status, ws, stream);
}

int lcp_pdipm_forward_f64(int B, int nz, int m, int e, const double* Q, const double* p, const double* G,
                          const double* h, const double* A, const double* b, const double* F, double eps,
                          int max_iter, int not_improved_lim, double* x, double* y, double* z, double* s,
                          int32_t* iters, int32_t* status, void* ws, void* stream) {
  return forward_common(1, B, nz, m, e, Q, p, G, h, A, b, F, eps, max_iter, not_improved_lim,
                        LCP_COMPUTE_F64, x, y, z, s, iters, status, ws, stream);
}

static int backward_common(int io_f64, int B, int nz, int m, int e, const void* G, const void* A,
                           const void* dl_dx, int compute, void* dQ, void* dp, void* dG, void* dh, void* dA,
                           void* db, void* dF, void* ws, void* stream) {
  if (B <= 0 || nz <= 0 || m <= 0 || e < 0) return LCP_E_BADARG;
  if (!G || !dl_dx || !ws) return LCP_E_BADARG;
  if (e > 0 && !A) return LCP_E_BADARG;
  const int hint = compute & LCP_HINT_ALL_CONTACT;
  const bool pinned = (compute & LCP_HINT_PINNED) != 0;      // (the forward's word: its promise holds for the backward too)
  const bool adjoint = (compute & LCP_BWD_ADJOINT) != 0 || (io_f64 && g_adjoint);   // opt-in: solve with K^T (generic kernels; the forward ran with LCP_PATH_GENERIC)
  if (adjoint) { if (hint) return LCP_E_BADARG; compute |= LCP_PATH_GENERIC; }
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  const int cs = csize_of(io_f64, compute);
  const bool w64 = use_wave64(io_f64, nz, m, e, generic);
  lcp::Plan pl = lcp::make_plan(nz, m, e, cs);
  if (!w64 && !pl.ok) return LCP_E_TOOLARGE;
  const size_t per_scene_all = scene_bytes(nz, m, e, io_f64 ? LCP_COMPUTE_F64 : compute, io_f64);
  const bool bigd = use_big_dense(io_f64, nz, m, e, compute, generic);
  lcp::BwdArgs P;
  memset(&P, 0, sizeof(P));
  const int dense_body = (w64 && path != 3 && lcp::quad_supported(nz, m, e) && lcp::quad_dense_is_body_space(io_f64, compute, 1)) ? 1 : 0;   // (as in forward_common)
  P.tag = trailer_of(ws, B, per_scene_all);
  P.tag_value = w64 ? (dense_body ? TAG_DENSE_WAVE_BODY : TAG_DENSE_WAVE) : (bigd ? TAG_DENSE_BIG : TAG_DENSE_GENERIC);
  P.B = B; P.nz = nz; P.m = m; P.e = e; P.G = G; P.A = A; P.dl_dx = dl_dx;
  P.dQ = dQ; P.dp = dp; P.dG = dG; P.dh = dh; P.dA = dA; P.db = db; P.dF = dF;
  P.ws = ws; P.ws_stride = pl.ws_stride; P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds;
  P.adjoint = adjoint ? 1 : 0;
  if (adjoint && !pl.ok) return LCP_E_TOOLARGE;
  if (hint) {
    // LCP_HINT_ALL_CONTACT: the workspace was left by a contact-list forward (lcp_step_fused_f32 / lcp_solve_dynamics_f32) called with
    // this `compute` word.  Three of its kernel families keep a workspace a dense backward can read - the four-scenes-per-wave one
    // (in body space, no W in it: lcp_bwd_quad<..., BODY>, or in contact space, exactly as that forward decided), the wave64 step
    // kernel and the generic one (their dense layouts) -, the tag says which one it was.
    if (io_f64) return LCP_E_BADARG;
    const StepFamily fam = step_family(nz, m, e, compute, path);
    P.tag_value = step_tag(fam, nz, compute, path);
    if (fam == FAM_QUAD) {
      if (!lcp::quad_supported(nz, m, e)) return LCP_E_TOOLARGE;             // (nz 17..32: the physical backward only)
      return lcp::quad_backward(P, compute, 2, stream, 0, P.tag_value == TAG_STEP_QUAD_BODY, pinned);
    }
    if (fam == FAM_WAVE64) {
      // lcp_solve_dynamics_f32 (contact counts per scene) runs these sizes on the generic step kernel (launch_step), the full-list
      // lcp_step_fused_f32 on the wave64 one: the `compute` word cannot tell, the tag the forward left can.  Both backwards are
      // launched; the one whose family did not run the forward finds its partner's tag and leaves without writing.
      P.skip_tag = TAG_STEP_GENERIC;
      int rc = lcp::wave64_backward(P, compute, false, stream, 0);
      if (rc) return rc;
      if (!pl.ok) return 0;                                                   // (no generic plan at these sizes: the generic step cannot have run either)
      P.tag_value = TAG_STEP_GENERIC; P.skip_tag = TAG_STEP_WAVE64;
      return lcp::generic_backward(P, io_f64, compute, pl.lds_bytes, stream);
    }
    if (fam != FAM_GENERIC) return LCP_E_TOOLARGE;                            // (lcp_primal / lcp_big: lcp_step_backward_f32 is their backward)
    return lcp::generic_backward(P, io_f64, compute, pl.lds_bytes, stream);
  }
  if (w64) return lcp::wave64_backward(P, compute, false, stream, io_f64, dense_body);
  if (bigd) {
    const size_t per_scene = per_scene_all;
    int32_t* cls = (int32_t*)((unsigned char*)ws + (size_t)B * per_scene);
    int rc = lcp::big_dense_backward(P, cls, per_scene, stream);           // (the classes the forward left behind the scene blocks)
    if (rc) return rc;
    if ((nz % 3) == 0 && lcp::primal_dense_supported(nz, m, e)) {
      // (classes 3 / 4 - the forward only hands them out for 16-byte aligned F and G - write dG and dF with 16-byte stores)
      if (((((uintptr_t)dG) | ((uintptr_t)dF)) & 15) != 0) return LCP_E_BADARG;
      rc = lcp::primal_dense_backward(P, cls, per_scene, stream);
      if (rc) return rc;
    }
    P.cls = cls; P.ws_stride = per_scene / cs;
  }
  return lcp::generic_backward(P, io_f64, compute, pl.lds_bytes, stream);
}

int lcp_pdipm_backward_f32(int B, int nz, int m, int e, const float* G, const float* A, const float* dl_dx,
                           int compute, float* dQ, float* dp, float* dG, float* dh, float* dA, float* db,
                           float* dF, void* ws, void* stream) {
  return backward_common(0, B, nz, m, e, G, A, dl_dx, compute, dQ, dp, dG, dh, dA, db, dF, ws, stream);
}

int lcp_pdipm_backward_f64(int B, int nz, int m, int e, const double* G, const double* A,
                           const double* dl_dx, double* dQ, double* dp, double* dG, double* dh, double* dA,
                           double* db, double* dF, void* ws, void* stream) {
  return backward_common(1, B, nz, m, e, G, A, dl_dx, LCP_COMPUTE_F64, dQ, dp, dG, dh, dA, db, dF, ws,
                         stream);
}

static int launch_step(lcp::StepArgs& P, int nz, int m, int e, int compute, int path, void* stream, int solo = -1);

static int fill_step(lcp::StepArgs& P, int B, int nb, int nc, int e, const float* pos, const float* Mdiag,
                     const float* v, const float* f, const float* rest, const float* fric, const float* c_n,
                     const float* c_p1, const float* c_p2, const int32_t* c_i1, const int32_t* c_i2,
                     const float* Je, float dt) {
  if (B <= 0 || nb <= 0 || nc <= 0 || e < 0) return LCP_E_BADARG;
  if (!Mdiag || !v || !f || !rest || !fric || !c_n || !c_p1 || !c_p2 || !c_i1 || !c_i2) return LCP_E_BADARG;
  if (e > 0 && !Je) return LCP_E_BADARG;
  memset(&P, 0, sizeof(P));
  P.B = B; P.nb = nb; P.nc = nc; P.e = e;
  P.pos = pos; P.Mdiag = Mdiag; P.v = v; P.f = f; P.rest = rest; P.fric = fric;
  P.c_n = c_n; P.c_p1 = c_p1; P.c_p2 = c_p2; P.c_i1 = c_i1; P.c_i2 = c_i2; P.Je = Je; P.dt = dt;
  return 0;
}

int lcp_assemble_contacts_f32(int B, int nb, int nc, int e, const float* Mdiag, const float* v, const float* f,
                              const float* rest, const float* fric, const float* c_n, const float* c_p1,
                              const float* c_p2, const int32_t* c_i1, const int32_t* c_i2, const float* Je,
                              float dt, float* Q, float* p, float* G, float* h, float* A, float* b, float* F,
                              void* stream) {
  lcp::StepArgs P;
  int rc = fill_step(P, B, nb, nc, e, nullptr, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt);
  if (rc) return rc;
  if (!Q || !p || !G || !h || !F) return LCP_E_BADARG;
  if (e > 0 && (!A || !b)) return LCP_E_BADARG;
  return lcp::generic_assemble(P, Q, p, G, h, A, b, F, stream);
}

int lcp_step_fused_f32(int B, int nb, int nc, int e, const float* pos, const float* Mdiag, const float* v,
                       const float* f, const float* rest, const float* fric, const float* c_n,
                       const float* c_p1, const float* c_p2, const int32_t* c_i1, const int32_t* c_i2,
                       const float* Je, float dt, double eps, int max_iter, int not_improved_lim, int compute,
                       float* v_new, float* p_new, float* z, float* s, float* y, int32_t* iters,
                       int32_t* status, void* ws, void* stream) {
  bool generic;
  int path, solo;
  compute = split_compute(compute, &generic, &path, &solo);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  lcp::StepArgs P;
  int rc = fill_step(P, B, nb, nc, e, pos, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt);
  if (rc) return rc;
  if (!pos || !v_new || !p_new || !ws || max_iter < 0) return LCP_E_BADARG;
  const int nz = 3 * nb, m = 4 * nc;
  P.eps = eps; P.max_iter = max_iter; P.lim = not_improved_lim;
  P.v_new = v_new; P.p_new = p_new; P.z = z; P.s = s; P.y = y; P.iters = iters; P.status = status;
  P.ws = ws;
  return launch_step(P, nz, m, e, compute, path, stream, solo);
}

int lcp_step_backward_f32(int B, int nb, int nc, int e, const float* Mdiag, const float* v, const float* f,
                          const float* rest, const float* fric, const float* c_n, const float* c_p1,
                          const float* c_p2, const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                          const float* dl_dv, int compute, float* dMdiag, float* dv, float* df, float* drest,
                          float* dfric, float* dc_n, float* dc_p1, float* dc_p2, void* ws, void* stream) {
  return lcp_step_backward_je_f32(B, nb, nc, e, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt, dl_dv, compute, dMdiag,
                                  dv, df, drest, dfric, dc_n, dc_p1, dc_p2, nullptr, ws, stream);
}

int lcp_step_backward_je_f32(int B, int nb, int nc, int e, const float* Mdiag, const float* v, const float* f,
                             const float* rest, const float* fric, const float* c_n, const float* c_p1,
                             const float* c_p2, const int32_t* c_i1, const int32_t* c_i2, const float* Je, float dt,
                             const float* dl_dv, int compute, float* dMdiag, float* dv, float* df, float* drest,
                             float* dfric, float* dc_n, float* dc_p1, float* dc_p2, float* dJe, void* ws, void* stream) {
  bool generic;
  int path;
  const bool pinned = (compute & LCP_HINT_PINNED) != 0;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  lcp::StepArgs P;
  int rc = fill_step(P, B, nb, nc, e, nullptr, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt);
  if (rc) return rc;
  if (!dl_dv || !ws) return LCP_E_BADARG;
  P.ws = ws;
  lcp::StepBwdArgs G;
  G.dl_dv = dl_dv; G.dMdiag = dMdiag; G.dv = dv; G.df = df; G.drest = drest; G.dfric = dfric;
  G.dcn = dc_n; G.dcp1 = dc_p1; G.dcp2 = dc_p2; G.dJe = (e > 0) ? dJe : nullptr;
  // the same family decision as the forward entry points (launch_step): a function of the sizes and the `compute` word; the
  // kernels check the tag that forward left in the workspace trailer
  const StepFamily fam = step_family(3 * nb, 4 * nc, e, compute, path);
  P.tag = trailer_of(ws, B, scene_bytes(3 * nb, 4 * nc, e, compute, 0));
  P.tag_value = step_tag(fam, 3 * nb, compute, path);
  switch (fam) {
    case FAM_QUAD: return lcp::quad_step_backward(P, G, compute, stream, path != 3, pinned);
    case FAM_PRIMAL: return lcp::primal_step_backward(P, G, stream, pinned);
    case FAM_BIG: return lcp::big_step_backward(P, G, stream);
    case FAM_GENERIC: {                      // (round 6) any size the generic plan holds: lcp_step_bwd_kernel on the iterate lcp_step_kernel kept
      const int cs = (compute == LCP_COMPUTE_F64) ? 8 : 4;
      lcp::Plan pl = lcp::make_plan(3 * nb, 4 * nc, e, cs);
      if (!pl.ok) return LCP_E_TOOLARGE;
      P.ws_stride = pl.ws_stride; P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds;
      return lcp::generic_step_backward(P, G, compute, pl.lds_bytes, stream);
    }
    default: return LCP_E_TOOLARGE;          // the wave64 step kernel keeps no workspace this backward can read
  }
}

int lcp_step_has_backward(int nb, int maxc, int e, int compute) {
  if (nb <= 0 || maxc <= 0 || e < 0) return 0;
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return 0;
  const StepFamily fam = step_family(3 * nb, 4 * maxc, e, compute, path);
  if (fam == FAM_GENERIC) return lcp::make_plan(3 * nb, 4 * maxc, e, (compute == LCP_COMPUTE_F64) ? 8 : 4).ok ? 1 : 0;
  return (fam == FAM_QUAD || fam == FAM_PRIMAL || fam == FAM_BIG) ? 1 : 0;
}

// forward of the contact-list entry points, by family
static int launch_step(lcp::StepArgs& P, int nz, int m, int e, int compute, int path, void* stream, int solo) {
  StepFamily fam = step_family(nz, m, e, compute, path);
  if (fam == FAM_WAVE64 && P.c_count) fam = FAM_GENERIC;                    // (its kernel takes full lists only)
  P.tag = trailer_of(P.ws, P.B, scene_bytes(nz, m, e, compute, 0));
  P.tag_value = step_tag(fam, nz, compute, path);
  switch (fam) {
    case FAM_QUAD: return lcp::quad_step(P, compute, stream, path != 3, solo >= 8 ? solo - 16 : solo, solo >= 8);
    case FAM_PRIMAL: return lcp::primal_step(P, stream, solo >= 8);
    case FAM_BIG: return lcp::big_step(P, stream);
    case FAM_WAVE64: return lcp::wave64_step(P, compute, stream);
    default: break;
  }
  const int cs = (compute == LCP_COMPUTE_F64) ? 8 : 4;
  lcp::Plan pl = lcp::make_plan(nz, m, e, cs);
  if (!pl.ok) return LCP_E_TOOLARGE;
  P.ws_stride = pl.ws_stride; P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds;
  return lcp::generic_step(P, compute, pl.lds_bytes, stream);
}

int lcp_solve_dynamics_f32(int B, int nb, int maxc, int e, const int32_t* c_count, const float* Mdiag,
                           const float* v, const float* f, const float* rest, const float* fric,
                           const float* c_n, const float* c_p1, const float* c_p2, const int32_t* c_i1,
                           const int32_t* c_i2, const float* Je, float dt, double eps, int max_iter,
                           int not_improved_lim, int compute, float* v_new, float* z, float* s, float* y,
                           int32_t* iters, int32_t* status, void* ws, void* stream) {
  bool generic;
  int path, solo;
  compute = split_compute(compute, &generic, &path, &solo);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  lcp::StepArgs P;
  int rc = fill_step(P, B, nb, maxc, e, nullptr, Mdiag, v, f, rest, fric, c_n, c_p1, c_p2, c_i1, c_i2, Je, dt);
  if (rc) return rc;
  if (!c_count || !v_new || !ws || max_iter < 0) return LCP_E_BADARG;
  P.c_count = c_count;
  P.eps = eps; P.max_iter = max_iter; P.lim = not_improved_lim;
  P.v_new = v_new; P.p_new = nullptr; P.z = z; P.s = s; P.y = y; P.iters = iters; P.status = status;
  P.ws = ws;
  return launch_step(P, 3 * nb, 4 * maxc, e, compute, path, stream, solo);
}

int lcp_post_stabilization_f32(int B, int nb, int maxc, int e, const int32_t* c_count, const float* Mdiag,
                               const float* v, const float* rest, const float* c_n, const float* c_p1,
                               const float* c_p2, const int32_t* c_i1, const int32_t* c_i2, const float* Je,
                               double eps, int max_iter, int not_improved_lim, int compute, const double* p,
                               const double* dt_scene, double dt, double* p_out, float* dp, int32_t* iters,
                               int32_t* status, void* ws, void* stream) {
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  lcp::StepArgs P;
  // (forces and friction do not enter this LCP: engines.py:80-116 reads M, v, Je, Jc and the restitutions only)
  int rc = fill_step(P, B, nb, maxc, e, nullptr, Mdiag, v, /*f*/ v, rest, /*fric*/ rest, c_n, c_p1, c_p2, c_i1, c_i2, Je,
                     (float)dt);
  if (rc) return rc;
  if (!c_count || !dp || !ws || max_iter < 0) return LCP_E_BADARG;
  if ((p_out != nullptr) != (p != nullptr)) return LCP_E_BADARG;
  const int nz = 3 * nb, m = 4 * maxc;
  P.c_count = c_count;
  P.dt = dt;
  P.eps = eps; P.max_iter = max_iter; P.lim = not_improved_lim;
  P.v_new = dp; P.iters = iters; P.status = status;
  P.pos64 = p; P.dt_scene = dt_scene; P.p_out64 = p_out;
  // body space, one wave per scene (lcp_primal.hip) where the sizes allow; the generic workgroup-per-scene kernel otherwise
  P.ws = ws;                                                                // (lcp_primal.hip leaves the best iterate there for the backward)
  // (LCP_PATH_PRIMAL: the one-wave-per-scene kernel at every size - the A/B partner of the four-scenes-per-wave form of round 5)
  const bool body = (path == 0 || path == 4) && compute == LCP_COMPUTE_F64 && lcp::primal_poststab_supported(nz, m, e);
  P.tag = trailer_of(ws, B, scene_bytes(nz, m, e, compute, 0));
  P.tag_value = body ? TAG_POSTSTAB_PRIMAL : TAG_POSTSTAB_GENERIC;
  if (body && path == 0 && lcp::quad_post_supported(nz, m, e)) return lcp::quad_post_stab(P, stream);   // (same workspace layout, same tag: one backward)
  if (body) return lcp::primal_post_stab(P, stream);
  const int cs = (compute == LCP_COMPUTE_F64) ? 8 : 4;
  lcp::Plan pl = lcp::make_plan(nz, m, e, cs);
  if (!pl.ok) return LCP_E_TOOLARGE;
  P.ws_stride = pl.ws_stride; P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds;
  return lcp::generic_post_stab(P, compute, pl.lds_bytes, stream);
}

int lcp_post_stabilization_has_backward(int nb, int maxc, int e, int compute) {
  if (nb <= 0 || maxc <= 0 || e < 0) return 0;
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return 0;
  // (the routing test of lcp_post_stabilization_backward_f32 below: body space where the forward ran there, else the generic kernels)
  if (compute == LCP_COMPUTE_F64 && (path == 0 || path == 4) && lcp::primal_poststab_supported(3 * nb, 4 * maxc, e)) return 1;
  return lcp::make_plan(3 * nb, 4 * maxc, e, (compute == LCP_COMPUTE_F64) ? 8 : 4).ok ? 1 : 0;
}

int lcp_post_stabilization_backward_f32(int B, int nb, int maxc, int e, const float* Mdiag, const float* v,
                                        const float* rest, const float* c_n, const float* c_p1, const float* c_p2,
                                        const int32_t* c_i1, const int32_t* c_i2, const float* Je, const float* dl_ddp,
                                        int compute, float* dMdiag, float* dv, float* drest, float* dc_n, float* dc_p1,
                                        float* dc_p2, float* dJe, void* ws, void* stream) {
  bool generic;
  int path;
  compute = split_compute(compute, &generic, &path);
  if (compute != LCP_COMPUTE_F32 && compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
  lcp::StepArgs P;
  int rc = fill_step(P, B, nb, maxc, e, nullptr, Mdiag, v, /*f*/ v, rest, /*fric*/ rest, c_n, c_p1, c_p2, c_i1, c_i2, Je, 0.0f);
  if (rc) return rc;
  if (!dl_ddp || !ws) return LCP_E_BADARG;
  // the same routing test as the forward: the body-space kernels where they ran, lcp_step_bwd_kernel<.., POST> on the iterate
  // lcp_post_stab_kernel kept otherwise (round 6: any size of the generic plan, fp32 arithmetic included)
  const bool body = (path == 0 || path == 4) && compute == LCP_COMPUTE_F64 && lcp::primal_poststab_supported(3 * nb, 4 * maxc, e);
  P.ws = ws;
  P.tag = trailer_of(ws, B, scene_bytes(3 * nb, 4 * maxc, e, compute, 0));
  P.tag_value = body ? TAG_POSTSTAB_PRIMAL : TAG_POSTSTAB_GENERIC;
  lcp::StepBwdArgs G = {};
  G.dl_dv = dl_ddp; G.dMdiag = dMdiag; G.dv = dv; G.drest = drest; G.dcn = dc_n; G.dcp1 = dc_p1; G.dcp2 = dc_p2;
  G.dJe = (e > 0) ? dJe : nullptr;
  if (body) return lcp::primal_post_stab_backward(P, G, stream);
  const int cs = (compute == LCP_COMPUTE_F64) ? 8 : 4;
  lcp::Plan pl = lcp::make_plan(3 * nb, 4 * maxc, e, cs);
  if (!pl.ok) return LCP_E_TOOLARGE;
  P.ws_stride = pl.ws_stride; P.ldT = pl.ldT; P.t_in_lds = pl.t_in_lds;
  return lcp::generic_post_stab_backward(P, G, compute, pl.lds_bytes, stream);
}

int lcp_move_find_contacts_f64(int B, int nb, int maxc, const int32_t* kind, const double* radius,
                               const double* verts_local, const int32_t* nverts, const uint8_t* no_contact,
                               const double* p_start, const float* v, double dt, double dt_floor, int strict,
                               int max_trials, double eps, double tol, double* p_out, float* c_n, float* c_p1,
                               float* c_p2, double* c_pen, int32_t* c_i1, int32_t* c_i2, int32_t* count,
                               double* max_pen, double* dt_used, double* t, int32_t* trials, void* stream) {
  if (B <= 0 || nb <= 0 || maxc <= 0 || max_trials <= 0) return LCP_E_BADARG;
  if (!kind || !radius || !verts_local || !nverts || !p_start) return LCP_E_BADARG;
  if (!c_n || !c_p1 || !c_p2 || !c_i1 || !c_i2 || !count) return LCP_E_BADARG;
  lcp::ContactArgs P;
  memset(&P, 0, sizeof(P));
  P.B = B; P.nb = nb; P.maxc = maxc; P.kind = kind; P.nverts = nverts; P.radius = radius;
  P.verts_local = verts_local; P.no_contact = no_contact; P.p_start = p_start; P.v = v;
  P.dt = dt; P.dt_floor = dt_floor; P.eps = eps; P.tol = tol; P.strict = strict; P.max_trials = max_trials;
  P.p_out = p_out; P.c_n = c_n; P.c_p1 = c_p1; P.c_p2 = c_p2; P.c_pen = c_pen; P.c_i1 = c_i1; P.c_i2 = c_i2;
  P.count = count; P.max_pen = max_pen; P.dt_used = dt_used; P.t = t; P.trials = trials;
  return lcp::contacts_launch(P, stream);
}

int lcp_joint_jacobian_f64(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                           const double* jr1, double* jrot1, const double* p, const float* v, const double* dt_scene, double dt,
                           double vscale, float* Je, void* stream) {
  if (B <= 0 || nb <= 0 || nj <= 0 || e <= 0) return LCP_E_BADARG;
  if (!jtype || !jb1 || !jb2 || !jr1 || !jrot1 || !p || !Je) return LCP_E_BADARG;
  return lcp::joint_jacobian_launch(B, nb, nj, e, jtype, jb1, jb2, jr1, jrot1, p, v, dt_scene, dt, vscale, Je, stream);
}

int lcp_joint_jacobian_backward_f64(int B, int nb, int nj, int e, const int32_t* jtype, const int32_t* jb1, const int32_t* jb2,
                                    const double* jr1, const double* jrot1, const float* gJe, double* g_p, double* g_rot, void* stream) {
  if (B <= 0 || nb <= 0 || nj <= 0 || e <= 0) return LCP_E_BADARG;
  if (!jtype || !jb1 || !jb2 || !jr1 || !jrot1 || !gJe || !g_p || !g_rot) return LCP_E_BADARG;
  return lcp::joint_jacobian_backward_launch(B, nb, nj, e, jtype, jb1, jb2, jr1, jrot1, gJe, g_p, g_rot, stream);
}

int lcp_state_update_backward_f64(int B, int nb, int nj, const double* g_p, const double* g_g, const double* g_rot, const float* v,
                                  const double* dt_scene, double scale, const int32_t* jtype, const int32_t* jb1, float* g_v, void* stream) {
  if (B <= 0 || nb <= 0 || nj < 0) return LCP_E_BADARG;
  if (!v || !dt_scene || !g_v) return LCP_E_BADARG;
  if (g_rot && (nj <= 0 || !jtype || !jb1)) return LCP_E_BADARG;
  return lcp::state_update_backward_launch(B, nb, nj, g_p, g_g, g_rot, v, dt_scene, scale, jtype, jb1, g_v, stream);
}

int lcp_contact_frame_backward_f64(int B, int nb, int maxc, const int32_t* kind, const double* radius, const double* verts_local,
                                   const int32_t* nverts, const uint8_t* no_contact, const double* p, double eps,
                                   const int32_t* count, const float* g_n, const float* g_p1, const float* g_p2, double* dp,
                                   void* stream) {
  if (B <= 0 || nb <= 0 || maxc <= 0) return LCP_E_BADARG;
  if (!kind || !radius || !verts_local || !nverts || !p || !count || !g_n || !g_p1 || !g_p2 || !dp) return LCP_E_BADARG;
  return lcp::contact_frame_backward_launch(B, nb, maxc, kind, radius, verts_local, nverts, no_contact, p, eps, count, g_n, g_p1, g_p2,
                                            dp, stream);
}

}  // extern "C"
