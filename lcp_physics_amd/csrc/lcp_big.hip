// lcp_big.hip - contact-structured PDIPM forward for LARGE scenes: up to 64 contacts (nineq 256), nz <= 43, neq <= 4,
// diagonal Q, one 256-thread workgroup per scene.  BASELINE config 5 (4096 x 64 contacts) and every ContactWorld scene
// beyond the four-scenes-per-wave kernel's 16 contacts / 5 bodies.
//
// Same algorithm and reference lines as lcp_quad.hip (pdipm.py:49-186, 325-454; engines.py:26-78; the exact 4nc -> 2nc
// reduction of lcp_wave64.hip `Red`), different mapping because a 128 x 128 reduced system no longer fits one wave:
//   * vector role - wave 0, lane c = contact c (all four inequality components of every m-space vector in the lane, as in
//     the quad kernel), lane j = x-space entry j, lane a = e-space entry a.  Cross-lane traffic goes through small LDS
//     buffers (a 64-lane scene has no DPP row to broadcast in).
//   * matrix role - all 256 threads own the reduced matrix T (rows / columns 0..63 = a_c, 64..127 = u_c) as 8 x 8 register
//     tiles in a G x G cyclic layout (G = 16): thread (ti, tj) holds the entries (ti + G p, tj + G q).  The right-looking LU
//     (no pivoting, as in the quad kernel) broadcasts the pivot row and the multiplier column through LDS, two barriers
//     per pivot; the load stays balanced to the last step because the layout is cyclic.
//   * the finished factors are parked in LDS column-major (128 KB - this is what sizes the kernel: one scene per CU) and
//     wave 0 runs the triangular sweeps off conflict-free column reads.
// W = J P J^T (the part of T that does not change over the iterations) is formed once, tile by tile, and kept in the
// workspace in tile order (each thread re-reads its own 512 contiguous bytes per factorisation).
// Served through lcp_solve_dynamics_f32 (forward) and lcp_step_backward_f32 (gradients w.r.t. the physical inputs of the
// step); the workspace it leaves is its own (W tiles + the best iterate), not the dense LCPFunction backward's.
#include "lcp_wave_scene.h"
#include "lcp_big_dpp.h"

namespace lcp {
namespace big {

using namespace w64;
using namespace wsc;

// threads per scene: a G x G grid over the reduced matrix.  64 contacts: 16 x 16 = 256 threads (four waves, 8 x 8 tiles).
// 32 contacts: 8 x 8 = ONE wave (again 8 x 8 tiles): the workgroup barriers of the pivot loop become wave-local and four
// scenes share a CU, one per SIMD (measured against the 256-thread form: 1.0 M -> see DESIGN.md).  16 contacts: 16 x 16.
template <int NCB> struct Grid { static constexpr int G = (NCB == 32) ? 8 : 16, NT = G * G; };
constexpr int LX = 64;           // lanes of wave 0 = stride of the exchange buffers and of the stored iterate
// The kernel is instantiated for a contact capacity NCB of 64, 32 or 16: the reduced system has NRD = 2 NCB rows, the
// register tile of a thread is TP x TP with TP = NRD / G (8, 8, 2), and the LDS footprint (NRD x (NRD + 1) doubles of
// factors) lets 1, 4 or ~8 scenes share a CU.
constexpr int EQB = 4;           // padded neq
// LCP_BIG_MFMA = 1 (default): the 64- and 32-contact classes factor T with the BLOCKED right-looking LU below (16-wide
// panels, trailing rank-16 updates on v_mfma_f64_16x16x4_f64, two barriers per PANEL); 0 keeps the round-1 rank-1 LU
// (one barrier per pivot) for A/B runs.  The 16-contact class (2 x 2 register tiles) always uses the rank-1 form.
#ifndef LCP_BIG_MFMA
#define LCP_BIG_MFMA 1
#endif
typedef double d4 __attribute__((ext_vector_type(4)));
// lane K of every 16-lane DPP row, broadcast to its row (one v_mov_b64_dpp row_newbcast:K)
template <int K> __device__ __forceinline__ double bc16(double v) { return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xf, 0xf, true); }
// Pins a value where it is computed.  The panel results are stored under `if (valid)`: without this LLVM SINKS the whole
// dependent chain of a tile into that branch and keeps all 120 broadcast pivot-row values alive for it (measured: 256 + 256
// registers and 1.4 KB of scratch per lane instead of ~90 registers).
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
constexpr int NZB = 64;          // x-space capacity (lanes of wave 0)

struct Lds {
  double* LU;        // [128][128] column-major: LU[j * 128 + i]; before the first factorisation also prefactor scratch
  double* prow;      // [2][128] pivot row, double-buffered over the pivot steps
  double* pcol;      // [2][128] multiplier column
  double* dU;        // [128] 1 / U[i][i]
  double* pinv;      // [2][2] (pivot, 1 / pivot) of the coming step, written by the owner of the diagonal entry
  double* add;       // [3][64] addA, addB, addU of the current factorisation
  double* xv;        // [64] x-space exchange
  double* wv;        // [2][64] m-space exchange (normal part, tangential part)
  double* ev;        // [EQB] e-space exchange
  double* qid;       // [64]
  double* S11;       // [EQB][EQB]
  float* Jc;         // [64][nzs]
  float* Jt;         // [64][nzs]
  float* At;         // [EQB][nzs]
  int* flag;         // [4]: 0 singular pivot, 1 all done, 2 singular S11
  double* dt;        // [16][17] raw diagonal tile of the current panel step (blocked LU)
};
template <int NCB>
__host__ __device__ inline size_t carve(Lds& L, unsigned char* smem, int nzs) {
  unsigned char* q = smem;
  auto take = [&](size_t bytes) { unsigned char* r = q; q += (bytes + 15) & ~(size_t)15; return r; };
  constexpr int NRD = 2 * NCB, LDU = NRD + 1;
  L.LU = (double*)take(sizeof(double) * (NRD * LDU > 2 * NRD * EQB ? NRD * LDU : 2 * NRD * EQB));   // (also the prefactor scratch)
  L.prow = (double*)take(sizeof(double) * 2 * NRD);
  L.pcol = (double*)take(sizeof(double) * 2 * NRD);
  L.dU = (double*)take(sizeof(double) * (NRD > 2 * LX ? NRD : 2 * LX));
  L.pinv = (double*)take(sizeof(double) * 4);
  L.add = (double*)take(sizeof(double) * 3 * LX);
  L.xv = (double*)take(sizeof(double) * NZB);
  L.wv = (double*)take(sizeof(double) * 2 * LX);
  L.ev = (double*)take(sizeof(double) * EQB);
  L.qid = (double*)take(sizeof(double) * NZB);
  L.S11 = (double*)take(sizeof(double) * EQB * EQB);
  L.Jc = (float*)take(sizeof(float) * NCB * nzs);
  L.Jt = (float*)take(sizeof(float) * NCB * nzs);
  L.At = (float*)take(sizeof(float) * EQB * nzs);
  L.flag = (int*)take(sizeof(int) * 4);
  L.dt = (double*)take(sizeof(double) * 16 * 17);
  return (size_t)(q - smem);
}

// row r of the stacked Jacobian [Jc; Jt] (r < 64: Jc row r, else Jt row r - 64)
template <int NCB>
__device__ __forceinline__ const float* jrow(const Lds& L, int r, int nzs) { return (r < NCB ? L.Jc + (size_t)r * nzs : L.Jt + (size_t)(r - NCB) * nzs); }

// workspace per scene (doubles): W tiles, a 64-entry header (contact count), then the best iterate the backward needs:
// x[64] y[8] z[4][64] s[4][64], then mu[64] and diag(Q)[64] (what the dense backward cannot read from its arguments)
template <int NCB> struct WsLayout { static constexpr int W = 4 * NCB * NCB, IT = W + 64, TOTAL = IT + 64 + 8 + 10 * LX; };

// ---------------------------------------------------------------- the kernel
// BWD = false: fused step (contact list in, v_new out; engines.py:26-78).
// BWD = true : backward of that step w.r.t. its physical inputs (what lcp_bwd_step_quad does for the small scenes: one
//              factorisation at the stored iterate, one KKT solve - lcp.py:37-64 - and the contraction of the rank-1
//              LCP gradients through the engine assembly), reading W and the iterate the forward left in the workspace.
// (second launch bound = waves per SIMD the register allocation must allow: the small classes share a CU)
// DENSE: the same solve behind the dense LCPFunction boundary (lcp.py:22-64): the scene's (Q, p, G, h, A, b, F) is read
//        instead of a contact list (scenes lcp_classify_big marked contact-structured with diagonal Q), the outputs are
//        x, y, z, s (forward) or the seven dense gradients of lcp.py:52-61 (backward).
template <int NCB, bool BWD, bool DENSE>
__global__ void __launch_bounds__(Grid<NCB>::NT, NCB == 16 ? 4 : 1) lcp_big_kernel(StepArgs SP, StepBwdArgs Gd, int nzs, DenseIO DN) {
  constexpr int G = Grid<NCB>::G, NT = Grid<NCB>::NT, GSH = (G == 8) ? 3 : 4;
  constexpr int NRD = 2 * NCB, LDU = NRD + 1, TP = NRD / G, TH = TP / 2;
  constexpr int WS_W = WsLayout<NCB>::W, WS_IT = WsLayout<NCB>::IT, WS_TOTAL = WsLayout<NCB>::TOTAL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int scene = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // The wave that plays the vector role rotates with the block index: the dispatcher puts wave i of every workgroup on
  // SIMD i, so with a fixed choice the (serial, busy) vector waves of all the scenes sharing a CU would queue up on one
  // SIMD while the other three idle.  Blocks 256 apart are the ones that tend to share a CU.
  const bool w0 = wave == ((NCB == 64 || NT == 64) ? 0 : (int)((blockIdx.x >> 8) & 3));
  const int ti = tid >> GSH, tj = tid & (G - 1);                          // tile coordinates of the matrix role
  if (!BWD && blockIdx.x == 0 && threadIdx.x == 0 && SP.tag) *SP.tag = SP.tag_value;   // workspace trailer: which kernel family laid it out
  const bool tag_ok = !BWD || !SP.tag || *SP.tag == SP.tag_value;          // (a backward on another family's workspace returns NaN gradients)
  if (DENSE && DN.cls[blockIdx.x] != 2) return;                            // (a general scene: the generic kernels serve it)
  const int nb = SP.nb, nz = DENSE ? DN.nz : 3 * nb, ncap = SP.nc, e = SP.e;
  Lds L;
  carve<NCB>(L, smem, nzs);
  const int lc = lane < NCB ? lane : NCB - 1;                             // (lanes beyond the contact capacity read in bounds, results unused)
  // ---- blocked LU (MF): T as NTL x NTL tiles of 16 x 16, each tile one MFMA accumulator (4 doubles per lane:
  // register r of lane l = element (hg + 4 r, lo), hg = l >> 4, lo = l & 15).  64 contacts: four waves in a 2 x 2 grid,
  // cyclic over the tiles (wave (wr, wc) owns the tiles (2a + wr, 2b + wc)); 32 contacts: one wave owns all 4 x 4 tiles.
  // Tiles BELOW the diagonal are held TRANSPOSED (register r of lane l = element (lo, hg + 4 r)): then every trailing
  // update is an MFMA whose A / B operands are plain lane-linear reads of the finished panels (see `trailing`).
  constexpr bool MF = (LCP_BIG_MFMA != 0) && (NCB != 16);
  constexpr int NTL = NRD / 16, NWV = NT / 64;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);                 // (scalar: tile ownership tests become scalar branches)
  const int wr = (NWV == 4) ? (wave_u >> 1) : 0, wc = (NWV == 4) ? (wave_u & 1) : 0;
  const int lo = lane & 15, hg = lane >> 4;
  auto gI = [&](int a) { return (NWV == 4) ? 2 * a + wr : a; };          // global tile row / column of a local index
  auto gJ = [&](int b) { return (NWV == 4) ? 2 * b + wc : b; };
  double* Wg = (double*)SP.ws + (size_t)scene * (DENSE ? DN.ws_scene / sizeof(double) : (size_t)WS_TOTAL);   // W tiles, header, iterate
  double* Wit = Wg + WS_IT;                                               // best iterate
  int ncs = ncap;
  if (BWD) ncs = (int)Wg[WS_W];                                           // the count the forward solved with
  else if (SP.c_count) ncs = SP.c_count[scene];
  const int truncated = (ncs > ncap) ? LCP_ST_TRUNCATED : 0;               // more contacts found than the list holds
  ncs = ncs < 0 ? 0 : (ncs > ncap ? ncap : ncs);
  if (!BWD && tid == 0) Wg[WS_W] = (double)ncs;
  const bool vc = w0 && lane < ncs;                                       // this lane owns a live contact

  // ---- assembly (engines.py:31-32,50-74; world.py:144-234) ----------------------------------------------------------
  for (int i = tid; i < NCB * nzs; i += NT) { L.Jc[i] = 0.0f; L.Jt[i] = 0.0f; }
  for (int i = tid; i < EQB * nzs; i += NT) L.At[i] = 0.0f;
  if (tid < 4) L.flag[tid] = 0;
  __syncthreads();
  const float* Md = DENSE ? nullptr : (const float*)SP.Mdiag + (size_t)scene * nz;
  const float* vv = DENSE ? nullptr : (const float*)SP.v + (size_t)scene * nz;
  const float* ff = DENSE ? nullptr : (const float*)SP.f + (size_t)scene * nz;
  double mu_c = 0, hn = 0, p = 0, qd = 0, qid = 0, b_in = 0;
  if constexpr (DENSE) {
    // dense boundary: G = [Jc; Jf; 0] with Jf rows (+jt, -jt) (engines.py:67-68, world.py:191-192), F[3nc + c][c] = mu_c
    // (engines.py:71), h = [h_n; 0; 0] (:74) - all verified per scene by lcp_classify_big
    const int m = DN.m;
    const float* Gs = DN.G + (size_t)scene * m * nz;
    for (int i = tid; i < ncap * nz; i += NT) {
      const int c = i / nz, k = i - c * nz;
      L.Jc[(size_t)c * nzs + k] = Gs[(size_t)c * nz + k];
      L.Jt[(size_t)c * nzs + k] = Gs[(size_t)(ncap + 2 * c) * nz + k];
    }
    if (BWD) {                                                            // (lcp_pdipm_backward_f32 gets G, A and the cotangent only)
      if (vc) mu_c = Wit[72 + 8 * LX + lane];
      if (w0 && lane < nz) { qd = Wit[72 + 9 * LX + lane]; qid = 1.0 / qd; }
    } else {
      if (vc) {
        mu_c = (double)DN.F[(size_t)scene * m * m + (size_t)(3 * ncap + lane) * m + lane];
        hn = (double)DN.h[(size_t)scene * m + lane];
      }
      if (w0 && lane < nz) {
        const float q = DN.Q[(size_t)scene * nz * nz + (size_t)lane * nz + lane];
        qd = (double)q; qid = 1.0 / (double)q;
        p = (double)DN.p[(size_t)scene * nz + lane];
      }
      if (w0 && lane < e) b_in = (double)DN.b[(size_t)scene * e + lane];
      if (w0) { Wit[72 + 8 * LX + lane] = mu_c; Wit[72 + 9 * LX + lane] = qd; }
    }
    if (w0) L.qid[lane] = qid;
    for (int i = tid; i < e * nz; i += NT) { const int a = i / nz, k = i - a * nz; L.At[(size_t)a * nzs + k] = DN.A[(size_t)scene * e * nz + i]; }
  } else {
    if (vc) {
      const ContactRows<float> r = make_contact<float>((const float*)SP.c_n + (size_t)scene * ncap * 2, (const float*)SP.c_p1 + (size_t)scene * ncap * 2,
                                                       (const float*)SP.c_p2 + (size_t)scene * ncap * 2, SP.c_i1 + (size_t)scene * ncap,
                                                       SP.c_i2 + (size_t)scene * ncap, (const float*)SP.rest + (size_t)scene * nb,
                                                       (const float*)SP.fric + (size_t)scene * nb, vv, lane);
  #pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int col = (q < 3) ? 3 * r.b1 + q : 3 * r.b2 + (q - 3);
        L.Jc[(size_t)lane * nzs + col] = r.jn[q];
        L.Jt[(size_t)lane * nzs + col] = r.jf[q];
      }
      mu_c = (double)r.mu; hn = (double)r.h;
    }
    if (w0 && lane < nz) {
      const float q = Md[lane];
      qd = (double)q; qid = 1.0 / (double)q;
      p = (double)momentum_entry<float>(Md[lane], vv[lane], (float)SP.dt, ff[lane]);          // engines.py:32
    }
    if (w0) L.qid[lane] = qid;
    for (int i = tid; i < e * nz; i += NT) { const int a = i / nz, k = i - a * nz; L.At[(size_t)a * nzs + k] = ((const float*)SP.Je)[(size_t)scene * e * nz + i]; }
  }
  int status = truncated;
  __syncthreads();

  // ---- pre_factor_kkt for diagonal Q (pdipm.py:357-408): GA = J Q^-1 A^T, S11 = (A Q^-1 A^T)^-1, W = J P J^T -----------
  double* GA = L.LU;                       // [128][EQB]   (scratch inside the LU area, dead before the first factorisation)
  double* CC = L.LU + NRD * EQB;           // [128][EQB]   GA S11
  if (e > 0) {
    for (int r = tid; r < NRD; r += NT) {
      const float* jr = jrow<NCB>(L, r, nzs);
      for (int a = 0; a < EQB; ++a) {
        double acc = 0;
        for (int k = 0; k < nz; ++k) acc = fma((double)jr[k] * L.qid[k], (double)L.At[(size_t)a * nzs + k], acc);
        GA[r * EQB + a] = acc;
      }
    }
    if (tid < EQB * EQB) {
      const int a = tid >> 2, c = tid & 3;
      double acc = 0;
      for (int k = 0; k < nz; ++k) acc = fma((double)L.At[(size_t)a * nzs + k] * L.qid[k], (double)L.At[(size_t)c * nzs + k], acc);
      L.S11[tid] = acc;
    }
    __syncthreads();
    if (tid == 0) {                        // tiny e x e Gauss-Jordan
      double* a = L.S11;
      bool bad = false;
      for (int k = 0; k < e; ++k) {
        const double piv = a[k * EQB + k];
        bad = bad || !(piv != 0.0) || (piv != piv);
        const double pinv = 1.0 / piv;
        for (int j = 0; j < e; ++j) if (j != k) a[k * EQB + j] *= pinv;
        for (int i = 0; i < e; ++i) if (i != k) { const double f = a[i * EQB + k]; for (int j = 0; j < e; ++j) if (j != k) a[i * EQB + j] -= f * a[k * EQB + j]; a[i * EQB + k] = -f * pinv; }
        a[k * EQB + k] = pinv;
      }
      for (int i = 0; i < EQB; ++i) for (int j = 0; j < EQB; ++j) if (i >= e || j >= e) a[i * EQB + j] = 0;
      if (bad) L.flag[2] = 1;
    }
    __syncthreads();
    for (int r = tid; r < NRD; r += NT) for (int a = 0; a < EQB; ++a) {
      double acc = 0;
      for (int c = 0; c < EQB; ++c) acc = fma(GA[r * EQB + c], L.S11[c * EQB + a], acc);
      CC[r * EQB + a] = acc;
    }
    __syncthreads();
    if (L.flag[2]) status |= LCP_ST_SINGULAR_S11;
  }
  // the rows of GA this contact needs in the solve, and the row of S11 this e-lane needs
  double gan[EQB], gat[EQB], s11row[EQB];
#pragma unroll
  for (int a = 0; a < EQB; ++a) {
    gan[a] = (vc && e > 0) ? GA[lc * EQB + a] : 0.0;
    gat[a] = (vc && e > 0) ? GA[(NCB + lc) * EQB + a] : 0.0;
    s11row[a] = (w0 && lane < EQB && e > 0) ? L.S11[lane * EQB + a] : 0.0;
  }
  if (!BWD) {
    if constexpr (MF) {
      // W = J P J^T in tile order: the thread's four entries of tile (a, b) pair its `lo` index with one tile index and
      // its four `hg + 4 r` indices with the other (W[row][col] = sum_k (J[row][k] / q_k) J[col][k], as the rank-1 form)
      static_for<4>([&](auto A_) LCP_INL {
        static_for<4>([&](auto B_) LCP_INL {
          constexpr int a = A_, b = B_;
          const int I = gI(a), J = gJ(b);
          const bool up = I <= J;                                             // stored as is (else transposed)
          const int il = 16 * (up ? J : I) + lo;                              // the matrix index that runs along `lo`
          const int ih0 = 16 * (up ? I : J) + hg;                             // ... along `hg + 4 r`: ih0 + 4 r
          const float* jl = jrow<NCB>(L, il, nzs);
          const float* jh0 = jrow<NCB>(L, ih0, nzs); const float* jh1 = jrow<NCB>(L, ih0 + 4, nzs);
          const float* jh2 = jrow<NCB>(L, ih0 + 8, nzs); const float* jh3 = jrow<NCB>(L, ih0 + 12, nzs);
          double w0_ = 0, w1_ = 0, w2_ = 0, w3_ = 0;
          for (int k = 0; k < nz; ++k) {
            const double qk = L.qid[k], cl = (double)jl[k];
            const double h0 = (double)jh0[k], h1 = (double)jh1[k], h2 = (double)jh2[k], h3 = (double)jh3[k];
            if (up) { w0_ = fma(h0 * qk, cl, w0_); w1_ = fma(h1 * qk, cl, w1_); w2_ = fma(h2 * qk, cl, w2_); w3_ = fma(h3 * qk, cl, w3_); }
            else { const double rl = cl * qk; w0_ = fma(rl, h0, w0_); w1_ = fma(rl, h1, w1_); w2_ = fma(rl, h2, w2_); w3_ = fma(rl, h3, w3_); }
          }
          double wv[4] = {w0_, w1_, w2_, w3_};
          if (e > 0) {
            for (int q = 0; q < EQB; ++q) {
              static_for<4>([&](auto R_) LCP_INL {
                const int ih = ih0 + 4 * R_;
                const int row = up ? ih : il, col = up ? il : ih;
                wv[R_] = fma(-CC[row * EQB + q], GA[col * EQB + q], wv[R_]);
              });
            }
          }
          static_for<4>([&](auto R_) LCP_INL {
            const int cl_ = il & (NCB - 1), ch_ = (ih0 + 4 * R_) & (NCB - 1);
            Wg[((size_t)(a * 4 + b) * NT + tid) * 4 + R_] = (cl_ < ncs && ch_ < ncs) ? wv[R_] : 0.0;   // (32 B per lane and tile)
          });
        });
      });
    } else {
      // W tile of this thread: entries (ti + G p, tj + G q)
      double wt_[TP][TP];
      static_for<TP>([&](auto P) LCP_INL { static_for<TP>([&](auto Q) LCP_INL { wt_[P][Q] = 0; }); });
      for (int k = 0; k < nz; ++k) {
        const double qk = L.qid[k];
        double ri[TP], cj[TP];
        static_for<TP>([&](auto P) LCP_INL { ri[P] = (double)jrow<NCB>(L, ti + G * P, nzs)[k] * qk; cj[P] = (double)jrow<NCB>(L, tj + G * P, nzs)[k]; });
        static_for<TP>([&](auto P) LCP_INL { static_for<TP>([&](auto Q) LCP_INL { wt_[P][Q] = fma(ri[P], cj[Q], wt_[P][Q]); }); });
      }
      if (e > 0) {
        for (int a = 0; a < EQB; ++a) {
          double ci[TP], gj[TP];
          static_for<TP>([&](auto P) LCP_INL { ci[P] = CC[(ti + G * P) * EQB + a]; gj[P] = GA[(tj + G * P) * EQB + a]; });
          static_for<TP>([&](auto P) LCP_INL { static_for<TP>([&](auto Q) LCP_INL { wt_[P][Q] = fma(-ci[P], gj[Q], wt_[P][Q]); }); });
        }
      }
      // rows / columns of contacts the scene does not have are identity in T: zero here, 1 arrives through addA / addU
      static_for<TP>([&](auto P) LCP_INL {
        static_for<TP>([&](auto Q) LCP_INL {
          const int ci = (ti + G * P) & (NCB - 1), cj_ = (tj + G * Q) & (NCB - 1);
          const double v = (ci < ncs && cj_ < ncs) ? wt_[P][Q] : 0.0;
          Wg[(size_t)(P * TP + Q) * NT + tid] = v;                            // entry-major: every store / load instruction is coalesced
        });
      });
    }
  }
  if (!(qd != 0.0) && w0 && lane < nz) L.flag[2] = 2;
  __syncthreads();                                                        // GA / CC scratch dead from here on
  if (L.flag[2] == 2) status |= LCP_ST_SINGULAR_Q;

  // ---- products of the vector role (wave 0 only) ----------------------------------------------------------------------
  auto Gv = [&](double v, double& gn, double& gt) {                       // m-space <- x-space
    L.xv[lane] = v; wsync();
    gn = 0; gt = 0;
    if (lane < ncs) {                                                     // (x-space entries >= nz are 0 in xv, J columns >= nz are 0)
      const float* jc = L.Jc + (size_t)lane * nzs; const float* jt = L.Jt + (size_t)lane * nzs;
      for (int k0 = 0; k0 < nz; k0 += 8) {
        float a_[8], b_[8]; double xk[8];
        static_for<8>([&](auto I) LCP_INL { const int k = (k0 + I < nz) ? k0 + I : nz - 1; a_[I] = jc[k]; b_[I] = jt[k]; xk[I] = (k0 + I < nz) ? L.xv[k] : 0.0; });
        static_for<8>([&](auto I) LCP_INL { gn = fma((double)a_[I], xk[I], gn); gt = fma((double)b_[I], xk[I], gt); });
      }
    }
    wsync();
  };
  auto Gtw = [&](double wn, double wt) -> double {                        // x-space <- m-space
    L.wv[lane] = wn; L.wv[LX + lane] = wt; wsync();
    double acc = 0;
    if (lane < nz) {                                                      // (contacts >= ncs have zero rows in Jc / Jt)
      const int n8 = (ncs + 7) & ~7;
      double a0 = 0, a1 = 0;
      for (int c0 = 0; c0 < n8; c0 += 8) {
        float a_[8], b_[8]; double wn_[8], wt_[8];
        static_for<8>([&](auto I) LCP_INL { const int c = c0 + I; a_[I] = L.Jc[(size_t)c * nzs + lane]; b_[I] = L.Jt[(size_t)c * nzs + lane]; wn_[I] = L.wv[c]; wt_[I] = L.wv[LX + c]; });
        static_for<8>([&](auto I) LCP_INL { a0 = fma((double)a_[I], wn_[I], a0); a1 = fma((double)b_[I], wt_[I], a1); });
      }
      acc = a0 + a1;
    }
    wsync();
    return acc;
  };
  auto Av = [&](double v) -> double {                                     // e-space <- x-space
    L.xv[lane] = v; wsync();
    double acc = 0;
    if (lane < e) for (int k = 0; k < nz; ++k) acc = fma((double)L.At[(size_t)lane * nzs + k], L.xv[k], acc);
    wsync();
    return acc;
  };
  auto Aty = [&](double y) -> double {                                    // x-space <- e-space
    if (lane < EQB) L.ev[lane] = y; wsync();
    double acc = 0;
    if (lane < nz) for (int a = 0; a < e; ++a) acc = fma((double)L.At[(size_t)a * nzs + lane], L.ev[a], acc);
    wsync();
    return acc;
  };
  auto GAt = [&](double t, double& gn, double& gt) {                      // m-space <- e-space
    if (lane < EQB) L.ev[lane] = t; wsync();
    gn = 0; gt = 0;
#pragma unroll
    for (int a = 0; a < EQB; ++a) { const double tb = L.ev[a]; gn = fma(gan[a], tb, gn); gt = fma(gat[a], tb, gt); }
    wsync();
  };
  auto GAtw = [&](double wn, double wt) -> double {                       // e-space <- m-space
    double out = 0;
#pragma unroll
    for (int a = 0; a < EQB; ++a) { const double sm = wave_sum(gan[a] * wn + gat[a] * wt); if (lane == a) out = sm; }
    return out;
  };
  auto S11v = [&](double v) -> double {
    if (lane < EQB) L.ev[lane] = v; wsync();
    double acc = 0;
#pragma unroll
    for (int c = 0; c < EQB; ++c) acc = fma(s11row[c], L.ev[c], acc);
    wsync();
    return acc;
  };

  // ---- per-factorisation quantities of the 4nc -> 2nc reduction (lcp_wave64.hip `Red`) ------------------------------------
  double rSp = 0, rSm = 0, rDg = 1, ridet = 0.5, rwa = 0, rwu = 0, ua = 1, uu = 1;
  auto reduce_setup = [&](const M4<double>& D) {                          // D = 1 / d
    rDg = D.g; rSp = 0.5 * (D.f1 + D.f2); rSm = 0.5 * (D.f1 - D.f2);
    ridet = 1.0 / (rSp * rDg + 2.0);
    rwa = 2.0 * mu_c * ridet; rwu = -rDg * rSm * ridet;
    L.add[lane] = vc ? D.n : 1.0;
    L.add[LX + lane] = vc ? 0.5 * rSm * rwa : 0.0;
    L.add[2 * LX + lane] = vc ? 0.5 * (rSp + rSm * rwu) : 1.0;
    if (lane == 0) L.flag[0] = 0;                                         // "singular pivot" is per factorisation
  };

#ifdef LCP_BIG_PROFILE
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk = clock64();
  long long pcm[5] = {0, 0, 0, 0, 0}, tm = 0;      // blocked LU: publish + barrier, panel, barrier, trailing update
#define MF_TICK(i) { const long long now_ = clock64(); pcm[i] += now_ - tm; tm = now_; }
#define BIG_TICK(i) { const long long now_ = clock64(); pc[i] += now_ - tk; tk = now_; }
#else
#define BIG_TICK(i)
#define MF_TICK(i)
#endif
  // ---- factorisation: T = W + diag terms, LU in register tiles, factors to LDS (all 256 threads) ---------------------------
  auto factor = [&]() LCP_INL {
    if constexpr (MF) {
      // ================= blocked right-looking LU, no pivoting (same elimination order as the rank-1 form) ===============
      // per panel step k:  (1) the owners publish the diagonal tile (k, k) and the panel tiles (k, J > k), (I > k, k)
      //                    (2) every 16-lane DPP row takes ONE panel tile: lane = row of an L tile (or column of a U tile),
      //                        the tile's 16 entries in registers, and eliminates the 16 pivots of the diagonal tile -
      //                        which every DPP row factors redundantly, bit-identically - with row_newbcast broadcasts:
      //                        L_Ik = A_Ik U_kk^-1 and U_kJ = L_kk^-1 A_kJ by SUBSTITUTION (backward stable; the
      //                        explicit-inverse TRSM is not, and T is singular to working precision near convergence)
      //                    (3) trailing update A_IJ -= L_Ik U_kJ on v_mfma_f64_16x16x4_f64, four chained MFMAs per tile
      // two barriers per PANEL (the rank-1 form: one per pivot); the finished factors are left in LDS column-major,
      // exactly where the triangular sweeps of `tsolve` read them.
      d4 acc[4][4];
      static_for<4>([&](auto A_) LCP_INL { static_for<4>([&](auto B_) LCP_INL {       // 32 B per lane and tile: two dwordx4 loads
        acc[A_][B_] = *reinterpret_cast<const d4*>(Wg + ((size_t)(A_ * 4 + B_) * NT + tid) * 4);
      }); });
      // T = W + the diagonal terms of the reduction.  A lane holds at most ONE diagonal element of a tile: lo == hg + 4 rd.
      // Diagonal tiles (a, a) carry (a_c, a_c) / (u_c, u_c); the tiles (a, a - 2) - rows u_c, columns a_c, held transposed -
      // carry (u_c, a_c) on their diagonal.  With four waves they belong to the waves with wr == wc.
      if (wr == wc) {
        const bool dgl = (((lo - hg) & 3) == 0) && lo >= hg;
        const int rd = (lo - hg) >> 2;
        static_for<4>([&](auto A_) LCP_INL {
          const int row = 16 * gI(A_) + lo;
          double av = L.add[(row < NCB) ? row : 2 * LX + row - NCB];
          av = dgl ? av : 0.0;
          // (backward: W's own diagonal entry of the row is kept - the scale a pivot is held against afterwards; prow is free in the blocked LU)
          if constexpr (BWD) static_for<4>([&](auto R_) LCP_INL { if (dgl && rd == R_) L.prow[row] = acc[A_][A_][(int)R_]; });
          static_for<4>([&](auto R_) LCP_INL { acc[A_][A_][(int)R_] += (rd == R_) ? av : 0.0; });
        });
        static_for<2>([&](auto Q_) LCP_INL {
          constexpr int a = 2 + Q_, b = Q_;
          double av = L.add[LX + 16 * gJ(b) + lo];
          av = dgl ? av : 0.0;
          static_for<4>([&](auto R_) LCP_INL { acc[a][b][(int)R_] += (rd == R_) ? av : 0.0; });
        });
      }
      BIG_TICK(5)                                                             // (profile: W load + diagonal)
      auto sync = [&]() { if constexpr (NT == 64) wsync(); else __syncthreads(); };
      bool singular_seen = false;
      // ---- panel pass over ONE tile per 16-lane row.  Lane = row `lo`: d = its row of the diagonal tile (every DPP row factors
      // that tile redundantly, bit-identically), x = its row of an L tile (I, k), y = its row of a U tile (k, J).  Per pivot j:
      //     l = d[j] / u_jj (rows below j) ;  d[c] -= l u_jc, x[c] -= (x[j] / u_jj) u_jc  (c > j) ;  y[c] -= l y_j[c]  (all c)
      // i.e. L_Ik = A_Ik U_kk^-1 and U_kJ = L_kk^-1 A_kJ by substitution, the pivot row (u_jc, y_j[c]) arriving through
      // row_newbcast folded into the FMA (lcp_big_dpp.h).  Software-pipelined over the pivots: a step first updates column j + 1 -
      // it holds the NEXT pivot - and starts that pivot's reciprocal chain, then sweeps the remaining columns.
      auto panel = [&](int k, int I, bool vL, int J, bool vU, bool keeper, auto HASL_, auto HASU_) LCP_INL {
        constexpr bool HASL = HASL_, HASU = HASU_;
        double d[16], x[16], y[16], myinv = 1.0;
        // (unconditional loads - a row without a tile re-reads the diagonal tile's place - then selects: written as conditional
        //  loads they become eight exec-masked branches, each waiting out its own LDS round trip)
        const int Ie = vL ? I : k, Je = vU ? J : k;
        static_for<16>([&](auto C_) LCP_INL {
          d[C_] = L.dt[lo * 17 + C_];
          if constexpr (HASL) x[C_] = L.LU[(size_t)(16 * k + C_) * LDU + 16 * Ie + lo];
          if constexpr (HASU) y[C_] = L.LU[(size_t)(16 * Je + C_) * LDU + 16 * k + lo];
        });
        static_for<16>([&](auto C_) LCP_INL {
          if constexpr (HASL) x[C_] = vL ? x[C_] : 0.0;
          if constexpr (HASU) y[C_] = vU ? y[C_] : 0.0;
        });
        double piv = bc16<0>(d[0]);
        double inv = fast_rcp(piv);
        static_for<16>([&](auto J_) LCP_INL {
          constexpr int j = J_;
          singular_seen = singular_seen || (piv == 0.0);
          const double ld = (lo > j) ? d[j] * inv : 0.0;
          double lx = 0.0;
          if constexpr (HASL) lx = x[j] * inv;
          myinv = (lo == j) ? inv : myinv;
          if constexpr (j < 15) {
            const double u1 = bc16<j>(d[j + 1]);
            d[j + 1] = fma(-ld, u1, d[j + 1]);
            if constexpr (HASL) x[j + 1] = fma(-lx, u1, x[j + 1]);
            piv = bc16<j + 1>(d[j + 1]);
            inv = fast_rcp(piv);
            if constexpr (HASL) panel_dx<j, j + 2>(d, x, ld, lx); else panel_d<j, j + 2>(d, ld);
            if constexpr (HASU) panel_y<j>(y, ld);
          }
          d[j] = (lo > j) ? ld : d[j];
          if constexpr (HASL) x[j] = lx;
        });
        if constexpr (HASL) { if (vL) static_for<16>([&](auto C_) LCP_INL { L.LU[(size_t)(16 * k + C_) * LDU + 16 * I + lo] = x[C_]; }); }
        if constexpr (HASU) { if (vU) static_for<16>([&](auto C_) LCP_INL { L.LU[(size_t)(16 * J + C_) * LDU + 16 * k + lo] = y[C_]; }); }
        if (keeper) {                                                         // one DPP row stores the factored diagonal tile
          static_for<16>([&](auto C_) LCP_INL { L.LU[(size_t)(16 * k + C_) * LDU + 16 * k + lo] = d[C_]; });
          L.dU[16 * k + lo] = myinv;
        }
      };
      // (3) trailing update of the owned tiles (I > k, J > k) = the local tiles [A0, 4) x [B0, 4): one straight-line
      //     specialisation per (A0, B0), no branch between the MFMAs.  Operand reads, lane-linear in the finished panels:
      //       lop[a][c] = -L[16 I + lo][16 k + 4 c + hg]       uop[b][c] = U[16 k + 4 c + hg][16 J + lo]
      //     tile on / above the diagonal:  D -= L_Ik U_kJ       = mfma(A = lop, B = uop)
      //     tile below (held transposed):  D^T -= U_kJ^T L_Ik^T = mfma(A = uop, B = lop)
      //     local tiles a < b are above, a > b below; a == b is above iff wr <= wc (wave-uniform)
      const bool dup = (NWV == 1) || (wr <= wc);
      auto trailing = [&](auto A0_, auto B0_, int k, bool upd, bool pub) LCP_INL {
        constexpr int A0 = A0_, B0 = B0_;
        if (upd) {
        double lop[4][4], uop[4][4];
        static_for<4 - A0>([&](auto AA_) LCP_INL {
          constexpr int a = A0 + AA_;
          const int I = gI(a);
          static_for<4>([&](auto C_) LCP_INL { lop[a][C_] = -L.LU[(size_t)(16 * k + 4 * C_ + hg) * LDU + 16 * I + lo]; });
        });
        static_for<4 - B0>([&](auto BB_) LCP_INL {
          constexpr int b = B0 + BB_;
          const int J = gJ(b);
          static_for<4>([&](auto C_) LCP_INL { uop[b][C_] = L.LU[(size_t)(16 * J + lo) * LDU + 16 * k + 4 * C_ + hg]; });
        });
        static_for<4>([&](auto C_) LCP_INL {                                  // chunk-outer: independent accumulators in flight
          static_for<4 - A0>([&](auto AA_) LCP_INL { static_for<4 - B0>([&](auto BB_) LCP_INL {
            constexpr int a = A0 + AA_, b = B0 + BB_;
            double av, bv;
            if constexpr (a < b) { av = lop[a][C_]; bv = uop[b][C_]; }
            else if constexpr (a > b) { av = uop[b][C_]; bv = lop[a][C_]; }
            else { av = dup ? lop[a][C_] : uop[b][C_]; bv = dup ? uop[b][C_] : lop[a][C_]; }
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[a][b], 0, 0, 0);
          }); });
        });
        }
        // publish panel k + 1 (its tiles are all among the ones just updated): local row A0 is tile row k + 1 iff rp,
        // local column B0 is tile column k + 1 iff cp.  (Not for a panel without live contacts: nobody reads it.)
        const bool rp = pub && gI(A0) == k + 1, cp = pub && gJ(B0) == k + 1;
        if (rp && cp) static_for<4>([&](auto R_) LCP_INL { L.dt[(hg + 4 * R_) * 17 + lo] = acc[A0][B0][(int)R_]; });
        if (rp) static_for<4 - B0>([&](auto BB_) LCP_INL {
          constexpr int b = B0 + BB_;
          const int J = gJ(b);
          if (J > k + 1) static_for<4>([&](auto R_) LCP_INL { L.LU[(size_t)(16 * J + lo) * LDU + 16 * (k + 1) + hg + 4 * R_] = acc[A0][b][(int)R_]; });
        });
        if (cp) static_for<4 - A0>([&](auto AA_) LCP_INL {
          constexpr int a = A0 + AA_;
          const int I = gI(a);
          if (I > k + 1) static_for<4>([&](auto R_) LCP_INL { L.LU[(size_t)(16 * (k + 1) + hg + 4 * R_) * LDU + 16 * I + lo] = acc[a][B0][(int)R_]; });
        });
      };
      // (1) publish the tiles of panel 0 (registers -> LDS; below the diagonal the registers hold the transpose); the
      //     tiles of panel k + 1 are published by the trailing update of step k, straight from its accumulators
      {
        constexpr int k = 0;
        static_for<4>([&](auto A_) LCP_INL { static_for<4>([&](auto B_) LCP_INL {
          const int I = gI(A_), J = gJ(B_);
          if (I == k && J == k) {
            static_for<4>([&](auto R_) LCP_INL { L.dt[(hg + 4 * R_) * 17 + lo] = acc[A_][B_][(int)R_]; });
          } else if (I == k && J > k) {
            static_for<4>([&](auto R_) LCP_INL { L.LU[(size_t)(16 * J + lo) * LDU + 16 * I + hg + 4 * R_] = acc[A_][B_][(int)R_]; });
          } else if (J == k && I > k) {
            static_for<4>([&](auto R_) LCP_INL { L.LU[(size_t)(16 * J + hg + 4 * R_) * LDU + 16 * I + lo] = acc[A_][B_][(int)R_]; });
          }
        }); });
      }
#ifdef LCP_BIG_PROFILE
      tm = clock64();
#endif
      // A panel whose 16 pivots all belong to contacts the scene does not have is the identity: nothing to eliminate, nothing
      // to update with, and `tsolve` never reads its columns - the step is skipped (per-scene counts: ContactWorld).
      auto live_step = [&](int k) { return 16 * (k % (NTL / 2)) < ncs; };
#pragma unroll 1
      for (int k = 0; k < NTL; ++k) {
        const bool live = live_step(k);
        MF_TICK(4)
        if (live) {
        sync();
        MF_TICK(0)
        // (2) panel: L tiles k+1 .. on the first DPP rows, U tiles on the others
        if constexpr (NWV == 4) {
          const int slot = wave_u * 4 + hg;
          using T_ = std::integral_constant<bool, true>; using F_ = std::integral_constant<bool, false>;
          if (wave_u < 2) panel(k, k + 1 + slot, k + 1 + slot < NTL, 0, false, slot == 0, T_{}, F_{});      // L tiles k+1 .. on waves 0, 1
          else panel(k, 0, false, k + 1 + (slot - 8), k + 1 + (slot - 8) < NTL, false, F_{}, T_{});        // U tiles on waves 2, 3
        } else {
          using T_ = std::integral_constant<bool, true>;
          panel(k, k + 1 + hg, k + 1 + hg < NTL, k + 1 + hg, k + 1 + hg < NTL, hg == 0, T_{}, T_{});       // one wave: both tiles of index k+1+hg
        }
        MF_TICK(1)
        sync();
        } else if (wave_u == 0 && hg == 0) L.dU[16 * k + lo] = 1.0;           // (1 / U_ii of identity rows: `tsolve` scales by it)
        MF_TICK(2)
        // (3) trailing update of the owned tiles (I > k, J > k).  Operand reads, lane-linear in the finished panels:
        //       lop[a][c] = -L[16 I + lo][16 k + 4 c + hg]       uop[b][c] = U[16 k + 4 c + hg][16 J + lo]
        //     tile on / above the diagonal:  D -= L_Ik U_kJ       = mfma(A = lop, B = uop)
        //     tile below (held transposed):  D^T -= U_kJ^T L_Ik^T = mfma(A = uop, B = lop)
        {
          const int a0 = (NWV == 4) ? ((k + 2 - wr) >> 1) : k + 1, b0 = (NWV == 4) ? ((k + 2 - wc) >> 1) : k + 1;
          static_for<4>([&](auto A0_) LCP_INL { static_for<4>([&](auto B0_) LCP_INL {
            constexpr int A0 = A0_, B0 = B0_;
            if constexpr ((NWV == 4) ? (A0 - B0 <= 1 && B0 - A0 <= 1) : (A0 == B0)) {
              if (a0 == A0 && b0 == B0) trailing(A0_, B0_, k, live, k + 1 < NTL && live_step(k + 1));
            }
          }); });
        }
#ifdef LCP_BIG_PROFILE
        static_for<4>([&](auto A_) LCP_INL { static_for<4>([&](auto B_) LCP_INL { double t_ = acc[A_][B_][0]; pin(t_); }); });   // (wait for the MFMAs)
#endif
        MF_TICK(3)
      }
#ifdef LCP_BIG_PROFILE
      if (lane == 0) for (int i = 0; i < 5; ++i) L.prow[5 * wave_u + i] = (double)pcm[i];      // (prow is unused by the blocked LU)
#endif
      if (singular_seen && lo == 0) L.flag[0] = 1;
      BIG_TICK(6)                                                             // (profile: LU loop)
    } else {
      double t[TP][TP];
      static_for<TP>([&](auto P) LCP_INL { static_for<TP>([&](auto Q) LCP_INL { t[P][Q] = Wg[(size_t)(P * TP + Q) * NT + tid]; }); });
      if (ti == tj) {
        static_for<TH>([&](auto P) LCP_INL {
          const int c = ti + G * P;
          t[P][P] += L.add[c];                                              // (a_c, a_c)
          t[P + TH][P] += L.add[LX + c];                                    // (u_c, a_c)
          t[P + TH][P + TH] += L.add[2 * LX + c];                           // (u_c, u_c)
        });
      }
      int buf = 0;
      bool singular_seen = false;
      BIG_TICK(5)                                                             // (profile: W load + diagonal)
      const int nblk = (ncs + G - 1) >> GSH;                                  // G-pivot blocks that hold live contacts
      auto sync = [&]() { if constexpr (NT == 64) wsync(); else __syncthreads(); };   // (one wave: the LDS is in order)
      // Right-looking LU, no pivoting.  Pivot row, RAW multiplier column and (pivot, 1 / pivot) travel through LDS, double
      // buffered.  The step is software-pipelined: it first updates the border of the trailing tile - which holds the NEXT
      // pivot's row and column -, lets their owners publish them (and the owner of the diagonal entry run the reciprocal
      // chain), and only then sweeps the interior: the LDS round trip and the reciprocal of step k + 1 run under the
      // interior FMAs of step k, and one barrier per step remains.
      static_for<TP>([&](auto KB) LCP_INL {
        constexpr int kb = KB;
        // rows / columns of contacts the scene does not have are identity: a block made of them only has nothing to eliminate
        const int steps = ((kb % TH) < nblk) ? G : 0;
        if (steps) {                                                          // prologue: the block's first pivot
          double* prow = L.prow + buf * NRD;
          double* pcol = L.pcol + buf * NRD;
          if (ti == 0) static_for<TP - kb>([&](auto QQ) LCP_INL { constexpr int q = kb + QQ; prow[tj + G * q] = t[kb][q]; });
          if (tj == 0) static_for<TP - kb>([&](auto PP) LCP_INL { constexpr int pp = kb + PP; pcol[ti + G * pp] = t[pp][kb]; });
          if (tid == 0) { L.pinv[2 * buf] = t[kb][kb]; L.pinv[2 * buf + 1] = fast_rcp(t[kb][kb]); }
          sync();
        }
#pragma unroll 1
        for (int kk = 0; kk < steps; ++kk) {
          const double* prow = L.prow + buf * NRD;                            // (pivot k = G * kb + kk)
          const double* pcol = L.pcol + buf * NRD;
          // every LDS read of the step is issued here, ahead of any use
          // (unconditional loads, then selects: written as conditional loads they become sixteen exec-masked branches per step)
          double lm[TP - kb], rv[TP - kb];
          const double piv = L.pinv[2 * buf], inv = L.pinv[2 * buf + 1];       // (the reciprocal chain runs once, in the owner)
          static_for<TP - kb>([&](auto PP) LCP_INL { constexpr int pp = kb + PP; lm[PP] = pcol[ti + G * pp]; rv[PP] = prow[tj + G * pp]; });
          __builtin_amdgcn_sched_barrier(0);
          singular_seen = singular_seen || (piv == 0.0);
          // rows / columns of the LATER tile blocks (pp > kb) lie below / right of the pivot whatever the thread: only the
          // pivot's own block needs the "strictly below / right of k" masks (one compare each instead of TP - kb)
          const bool colk = tj == kk;
          static_for<TP - kb>([&](auto PP) LCP_INL {
            constexpr int pp = kb + PP;
            const double l = lm[PP] * inv;
            if constexpr (PP == 0) {
              lm[0] = (ti > kk) ? l : 0.0;
              rv[0] = (tj > kk) ? rv[0] : 0.0;
              t[pp][kb] = (colk && ti > kk) ? l : t[pp][kb];
            } else {
              lm[PP] = l;
              t[pp][kb] = colk ? l : t[pp][kb];
            }
          });
          // border of the trailing tile: its first row and first column
          static_for<TP - kb>([&](auto QQ) LCP_INL { t[kb][kb + QQ] = fma(-lm[0], rv[QQ], t[kb][kb + QQ]); });
          static_for<TP - kb - 1>([&](auto P1) LCP_INL { constexpr int PP = 1 + P1; t[kb + PP][kb] = fma(-lm[PP], rv[0], t[kb + PP][kb]); });
          buf ^= 1;
          if (kk + 1 < G) {                                                   // publish pivot k + 1 (same tile block)
            double* nrow = L.prow + buf * NRD;
            double* ncol = L.pcol + buf * NRD;
            if (ti == kk + 1) static_for<TP - kb>([&](auto QQ) LCP_INL { constexpr int q = kb + QQ; nrow[tj + G * q] = t[kb][q]; });
            if (tj == kk + 1) static_for<TP - kb>([&](auto PP) LCP_INL { constexpr int pp = kb + PP; ncol[ti + G * pp] = t[pp][kb]; });
            if (ti == kk + 1 && tj == kk + 1) { L.pinv[2 * buf] = t[kb][kb]; L.pinv[2 * buf + 1] = fast_rcp(t[kb][kb]); }
          }
          __builtin_amdgcn_sched_barrier(0);
          // interior
          static_for<TP - kb - 1>([&](auto P1) LCP_INL {
            constexpr int PP = 1 + P1;
            static_for<TP - kb - 1>([&](auto Q1) LCP_INL { constexpr int QQ = 1 + Q1; t[kb + PP][kb + QQ] = fma(-lm[PP], rv[QQ], t[kb + PP][kb + QQ]); });
          });
          sync();
        }
      });
      if (singular_seen && tid == 0) L.flag[0] = 1;
      BIG_TICK(6)                                                             // (profile: LU loop)
      // park the factors: column-major, plus the reciprocals of U's diagonal
      static_for<TP>([&](auto P) LCP_INL { static_for<TP>([&](auto Q) LCP_INL { L.LU[(size_t)(tj + G * Q) * LDU + ti + G * P] = t[P][Q]; }); });
      if (ti == tj) static_for<TP>([&](auto P) LCP_INL { L.dU[ti + G * P] = 1.0 / t[P][P]; });
    }
  };

  // ---- T^-1 hz through the reduced system (wave 0); rows: a_c = lane, u_c = 64 + lane --------------------------------------
  auto tsolve = [&](const M4<double>& hz) -> M4<double> {
    const double r12 = hz.f1 + hz.f2;
    const double w0_ = (rDg * r12 - 2.0 * hz.g) * ridet;
    double ra = hz.n, ru = 0.5 * (hz.f1 - hz.f2) - 0.5 * rSm * w0_;
    // Sweeps in chunks of 8 pivots: the column entries do not depend on the recurrence, so a chunk's 8 or 16 LDS reads
    // are issued together in front of its 8 dependent steps.  Rows / columns at or beyond the contact count are identity
    // (L = 0, 1 / U_ii = 1), so rounding the count up to a multiple of 8 only adds exact no-op steps.
    const int n8 = (ncs + 7) & ~7;
    for (int k0 = 0; k0 < n8; k0 += 8) {                                  // L y = rhs, columns a_k
      double la[8], lu[8];
      static_for<8>([&](auto I) LCP_INL { const double* col = L.LU + (size_t)(k0 + I) * LDU; la[I] = col[lc]; lu[I] = col[NCB + lc]; });
      static_for<8>([&](auto I) LCP_INL {
        const int k = k0 + I;
        const double yk = bcast_lane(ra, k);
        ra = fma(-((lane > k) ? la[I] : 0.0), yk, ra);
        ru = fma(-lu[I], yk, ru);
      });
    }
    for (int k0 = 0; k0 < n8; k0 += 8) {                                  // columns u_k
      double lu[8];
      static_for<8>([&](auto I) LCP_INL { lu[I] = L.LU[(size_t)(NCB + k0 + I) * LDU + NCB + lc]; });
      static_for<8>([&](auto I) LCP_INL {
        const int k = k0 + I;
        const double yk = bcast_lane(ru, k);
        ru = fma(-((lane > k) ? lu[I] : 0.0), yk, ru);
      });
    }
    for (int k0 = n8 - 8; k0 >= 0; k0 -= 8) {                             // U x = y, columns u_k
      double ua_[8], uu_[8], du[8];
      static_for<8>([&](auto I) LCP_INL { const double* col = L.LU + (size_t)(NCB + k0 + I) * LDU; ua_[I] = col[lc]; uu_[I] = col[NCB + lc]; du[I] = L.dU[NCB + k0 + I]; });
      static_for<8>([&](auto IR) LCP_INL {
        constexpr int I = 7 - IR;
        const int k = k0 + I;
        const double xk = bcast_lane(ru, k) * du[I];
        ra = fma(-ua_[I], xk, ra);
        ru = fma(-((lane < k) ? uu_[I] : 0.0), xk, ru);
      });
    }
    for (int k0 = n8 - 8; k0 >= 0; k0 -= 8) {                             // columns a_k
      double ua_[8], du[8];
      static_for<8>([&](auto I) LCP_INL { ua_[I] = L.LU[(size_t)(k0 + I) * LDU + lc]; du[I] = L.dU[k0 + I]; });
      static_for<8>([&](auto IR) LCP_INL {
        constexpr int I = 7 - IR;
        const int k = k0 + I;
        const double xk = bcast_lane(ra, k) * du[I];
        ra = fma(-((lane < k) ? ua_[I] : 0.0), xk, ra);
      });
    }
    const double a = ra * ua, u = ru * uu;
    const double w = w0_ + rwa * a + rwu * u;
    M4<double> dz;
    dz.n = a; dz.f1 = 0.5 * (w + u); dz.f2 = 0.5 * (w - u);
    dz.g = (r12 - rSm * u + rSp * (hz.g - mu_c * a)) * ridet;
    return dz;
  };

#ifdef LCP_BIG_PROFILE
  long long pc_ts = 0;
#endif
  // solve_kkt (pdipm.py:325-354); di = 1 / d
  auto solve_kkt = [&](const M4<double>& di, double rx, const M4<double>& rs, const M4<double>& rz, double ry,
                       double& ox, M4<double>& os, M4<double>& oz, double& oy, bool rxy_zero) {
    // `rxy_zero`: rx = ry = 0 (the corrector solve, pdipm.py:152-158) - the products of the zero vector are skipped
    const double v = qid * rx;                                            // :333 (diagonal Q)
    double gn = 0, gt = 0;
    if (!rxy_zero) Gv(v, gn, gt);
    M4<double> hz = m4<double>(gn + rs.n * di.n - rz.n, gt + rs.f1 * di.f1 - rz.f1, -gt + rs.f2 * di.f2 - rz.f2, rs.g * di.g - rz.g);   // :334-340
    double hy = 0;
    if (e > 0 && !rxy_zero) {
      hy = Av(v) - ry;
      double an, at;
      GAt(S11v(hy), an, at);
      hz.n -= an; hz.f1 -= at; hz.f2 += at;
    }
    if (!vc) hz = m4<double>(0, 0, 0, 0);
#ifdef LCP_BIG_PROFILE
    const long long t0_ = clock64();
#endif
    const M4<double> wz = tsolve(hz);
#ifdef LCP_BIG_PROFILE
    pc_ts += clock64() - t0_;
#endif
    double dy = 0;
    if (e > 0) dy = -S11v(hy - GAtw(vc ? wz.n : 0.0, vc ? wz.f1 - wz.f2 : 0.0));
    oz = m4<double>(-wz.n, -wz.f1, -wz.f2, -wz.g);                        // :342
    if (!vc) oz = m4<double>(0, 0, 0, 0);
    os = m4<double>((-rs.n - oz.n) * di.n, (-rs.f1 - oz.f1) * di.f1, (-rs.f2 - oz.f2) * di.f2, (-rs.g - oz.g) * di.g);   // :347,350
    if (!vc) os = m4<double>(0, 0, 0, 0);
    oy = dy;
    double g1 = -rx - Gtw(oz.n, oz.f1 - oz.f2);                           // :344-346
    if (e > 0) g1 -= Aty(dy);
    ox = qid * g1;                                                        // :349
  };

  // get_step for (z, dz), (s, ds) (pdipm.py:182-186), NaN semantics as in lcp_quad.hip step_pair_q
  auto step_pair = [&](const M4<double>& z, const M4<double>& dz, const M4<double>& s, const M4<double>& ds) -> double {
    const double ninf = -inf_of<double>(), pinf = inf_of<double>();
    const M4<double> az = m4<double>(-z.n / dz.n, -z.f1 / dz.f1, -z.f2 / dz.f2, -z.g / dz.g);
    const M4<double> as = m4<double>(-s.n / ds.n, -s.f1 / ds.f1, -s.f2 / ds.f2, -s.g / ds.g);
    auto key4 = [&](const M4<double>& a) { return umax(umax(nan_key(a.n), nan_key(a.f1)), umax(nan_key(a.f2), nan_key(a.g))); };
    auto max4 = [&](const M4<double>& a) { return __builtin_fmax(__builtin_fmax(a.n, a.f1), __builtin_fmax(a.f2, a.g)); };
    auto min4 = [&](const M4<double>& a) { return __builtin_fmin(__builtin_fmin(a.n, a.f1), __builtin_fmin(a.f2, a.g)); };
    const uint32_t kmz = wave_umax(vc ? key4(az) : 0u), kms = wave_umax(vc ? key4(as) : 0u);
    const double mz = wave_max(vc ? max4(az) : ninf), ms = wave_max(vc ? max4(as) : ninf);
    const double fz = key_is_nan(kmz) ? 1.0 : __builtin_fmax(mz, 1.0), fs = key_is_nan(kms) ? 1.0 : __builtin_fmax(ms, 1.0);
    auto pick = [&](double dv, double a, double fill) { return (dv > 0.0) ? fill : a; };
    const M4<double> pz = m4<double>(pick(dz.n, az.n, fz), pick(dz.f1, az.f1, fz), pick(dz.f2, az.f2, fz), pick(dz.g, az.g, fz));
    const M4<double> ps = m4<double>(pick(ds.n, as.n, fs), pick(ds.f1, as.f1, fs), pick(ds.f2, as.f2, fs), pick(ds.g, as.g, fs));
    const uint32_t kl = wave_umax(vc ? umax(key4(pz), key4(ps)) : 0u);
    const double l = wave_min(vc ? __builtin_fmin(min4(pz), min4(ps)) : pinf);
    return key_is_nan(kl) ? nan_of<double>() : l;
  };

  if (BWD) {
    // ---- backward: d(loss)/d(v_new) -> d(loss)/d(Mdiag, v, f, rest, fric, contact normal / arms) ----------------------------
    double x = 0, dx = 0, dnu = 0;
    M4<double> z = m4<double>(1, 1, 1, 1), s = z, dinv = z, ds, dl;
    if (w0) {
      x = (lane < nz) ? Wit[lane] : 0.0;
      if (vc) {
        z = m4<double>(Wit[72 + lane], Wit[72 + LX + lane], Wit[72 + 2 * LX + lane], Wit[72 + 3 * LX + lane]);
        s = m4<double>(Wit[72 + 4 * LX + lane], Wit[72 + 5 * LX + lane], Wit[72 + 6 * LX + lane], Wit[72 + 7 * LX + lane]);
        dinv = m4<double>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g);                 // 1 / d, d = z / s (lcp.py:44)
      }
      reduce_setup(dinv);
    }
    __syncthreads();
    factor();                                                               // lcp.py:46
    __syncthreads();
    if constexpr (MF) {
      // Round 6: the guard the other three backward families got in round 5.  At an iterate that converged to rounding
      // s / z of the active rows is lost against a W that redundant contact points make singular: a pivot that cancelled to less than
      // 1e-13 of its row's diagonal of W (or an exact zero) is rounding noise, the elimination would return multipliers of 1e15 - the
      // factorisation is repeated with s / z floored at 1e-12 x that diagonal (factor_bwd_q of lcp_quad_kernels.inc, same constants).
      bool tiny = false;
      double wa = 0, wu = 0;
      if (w0 && vc) {
        wa = L.prow[lc]; wu = L.prow[NCB + lc];
        wa = wa < 0 ? -wa : wa; wu = wu < 0 ? -wu : wu;
        double pa = 1.0 / L.dU[lc], pu = 1.0 / L.dU[NCB + lc];
        pa = pa < 0 ? -pa : pa; pu = pu < 0 ? -pu : pu;
        tiny = !(pa >= 1e-13 * wa) || !(pu >= 1e-13 * wu);
      }
      const bool again = __syncthreads_or((tiny || (w0 && L.flag[0] != 0)) ? 1 : 0) != 0;
      if (again) {
        if (w0) {
          if (vc) { dinv.n = __builtin_fmax(dinv.n, 1e-12 * wa); dinv.f1 = __builtin_fmax(dinv.f1, 1e-12 * wu); dinv.f2 = __builtin_fmax(dinv.f2, 1e-12 * wu); }
          reduce_setup(dinv);
        }
        __syncthreads();
        factor();
        __syncthreads();
      }
    }
    if constexpr (DENSE) {
      // ---- LCPFunction.backward (lcp.py:37-64): one KKT solve with rhs (dl_dx, 0, 0, 0), then the outer products ----------------
      double* V = L.LU;                                                     // staging area (the factors are dead after the solve)
      double *X = V, *DX = V + 64, *NU = V + 128, *DNU = V + 136, *LAM = V + 144, *DLAM = V + 144 + 4 * NCB;
      const int m = 4 * ncap;
      if (w0) {
        ua = L.dU[lc]; uu = L.dU[NCB + lc];
        const double g = !tag_ok ? nan_of<double>() : (lane < nz) ? (double)DN.dl_dx[(size_t)scene * nz + lane] : 0.0;
        const M4<double> zero = m4<double>(0, 0, 0, 0);
        solve_kkt(dinv, g, zero, zero, 0.0, dx, ds, dl, dnu, false);         // lcp.py:47-50
        const double nu = (lane < 8) ? Wit[64 + lane] : 0.0;
        wsync();
        X[lane] = x; DX[lane] = dx;
        if (lane < 8) { NU[lane] = nu; DNU[lane] = dnu; }
        if (lane < ncap) {                                                  // dense row order: [normal | friction pairs | cone]
          LAM[lane] = z.n; LAM[ncap + 2 * lane] = z.f1; LAM[ncap + 2 * lane + 1] = z.f2; LAM[3 * ncap + lane] = z.g;
          DLAM[lane] = dl.n; DLAM[ncap + 2 * lane] = dl.f1; DLAM[ncap + 2 * lane + 1] = dl.f2; DLAM[3 * ncap + lane] = dl.g;
        }
      }
      __syncthreads();
      if (DN.dp) for (int j = tid; j < nz; j += NT) DN.dp[(size_t)scene * nz + j] = (float)DX[j];                            // lcp.py:52
      if (DN.dh) for (int i = tid; i < m; i += NT) DN.dh[(size_t)scene * m + i] = (float)(-DLAM[i]);                          // :56
      if (DN.db) for (int a = tid; a < e; a += NT) DN.db[(size_t)scene * e + a] = (float)(-DNU[a]);                           // :58
      if (DN.dQ) for (int i = tid; i < nz * nz; i += NT) { const int j = i / nz, k = i - j * nz;
        DN.dQ[(size_t)scene * nz * nz + i] = (float)(0.5 * (DX[j] * X[k] + X[j] * DX[k])); }                                  // :59-60
      if (DN.dA) for (int i = tid; i < e * nz; i += NT) { const int a = i / nz, k = i - a * nz;
        DN.dA[(size_t)scene * e * nz + i] = (float)(DNU[a] * X[k] + NU[a] * DX[k]); }                                         // :57
      if (DN.dG) for (int i = tid; i < m * nz; i += NT) { const int r = i / nz, k = i - r * nz;
        DN.dG[(size_t)scene * m * nz + i] = (float)(DLAM[r] * X[k] + LAM[r] * DX[k]); }                                       // :53
      if (DN.dF) { float* o = DN.dF + (size_t)scene * m * m;
        for (int i = tid; i < m * m; i += NT) { const int r = i / m, c = i - r * m; o[i] = (float)(-DLAM[r] * LAM[c]); } }    // :54
      return;
    }
    if (!w0) return;
    ua = L.dU[lc]; uu = L.dU[NCB + lc];
    // v_new = -x (engines.py:76-77)  =>  d(loss)/dx = -d(loss)/d(v_new)
    const double g = !tag_ok ? nan_of<double>() : (lane < nz) ? -(double)((const float*)Gd.dl_dv)[(size_t)scene * nz + lane] : 0.0;
    const M4<double> zero = m4<double>(0, 0, 0, 0);
    solve_kkt(dinv, g, zero, zero, 0.0, dx, ds, dl, dnu, false);                   // lcp.py:47-50
    // x-space vectors to LDS so that a contact lane can read the entries of its two bodies
    double* X = L.xv; double* DX = L.wv; double* CR = L.add; double* CF = L.add + LX; int* B12 = (int*)(L.add + 2 * LX);
    X[lane] = x; DX[lane] = dx; wsync();
    double gh_rbar = 0;
    {
      double cr = 0, cf = 0, dnx = 0, dny = 0, d1x = 0, d1y = 0, d2x = 0, d2y = 0;
      int b1 = 0, b2 = 0;
      if (vc) {
        const size_t cb = (size_t)scene * ncap + lane;
        const double nx = ((const float*)SP.c_n)[cb * 2], ny = ((const float*)SP.c_n)[cb * 2 + 1];
        const double p1x = ((const float*)SP.c_p1)[cb * 2], p1y = ((const float*)SP.c_p1)[cb * 2 + 1];
        const double p2x = ((const float*)SP.c_p2)[cb * 2], p2y = ((const float*)SP.c_p2)[cb * 2 + 1];
        b1 = SP.c_i1[cb]; b2 = SP.c_i2[cb];
        const double rbar = 0.5 * ((double)((const float*)SP.rest)[(size_t)scene * nb + b1] + (double)((const float*)SP.rest)[(size_t)scene * nb + b2]);
        const double jn[6] = {p1x * ny - p1y * nx, nx, ny, -(p2x * ny - p2y * nx), -nx, -ny};      // world.py:177-183
        const double gh = -dl.n;                                              // dh = -dlam (lcp.py:56)
        const double af = dl.f1 - dl.f2, lf = z.f1 - z.f2;                    // Jf rows are +jt, -jt (world.py:191-192)
        double gjn[6], gjf[6], jnv = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int col = (q < 3) ? 3 * b1 + q : 3 * b2 + (q - 3);
          const double xq = X[col], dxq = DX[col], vq = (double)vv[col];
          jnv = fma(jn[q], vq, jnv);
          gjn[q] = dl.n * xq + z.n * dxq + gh * rbar * vq;                    // dG row n (lcp.py:53) + h = (Jc v) rbar
          gjf[q] = af * xq + lf * dxq;
        }
        gh_rbar = gh * rbar;
        cr = 0.5 * gh * jnv;                                                  // rbar = (rest_b1 + rest_b2) / 2 (world.py:144-151)
        cf = 0.5 * (-dl.g * z.n);                                             // dF[gamma_c, n_c] = -dlam_g lam_n (lcp.py:54), F = mu there
        dnx = -gjn[0] * p1y + gjn[1] + gjn[3] * p2y - gjn[4] - gjf[0] * p1x - gjf[2] + gjf[3] * p2x + gjf[5];
        dny = gjn[0] * p1x + gjn[2] - gjn[3] * p2x - gjn[5] - gjf[0] * p1y + gjf[1] + gjf[3] * p2y - gjf[4];
        d1x = gjn[0] * ny - gjf[0] * nx; d1y = -gjn[0] * nx - gjf[0] * ny;
        d2x = -gjn[3] * ny + gjf[3] * nx; d2y = gjn[3] * nx + gjf[3] * ny;
      }
      wsync();
      CR[lane] = cr; CF[lane] = cf; B12[lane] = b1; B12[LX + lane] = b2;
      if (lane < ncap) {
        const size_t cb = (size_t)scene * ncap + lane;
        if (Gd.dcn) { ((float*)Gd.dcn)[cb * 2] = (float)dnx; ((float*)Gd.dcn)[cb * 2 + 1] = (float)dny; }
        if (Gd.dcp1) { ((float*)Gd.dcp1)[cb * 2] = (float)d1x; ((float*)Gd.dcp1)[cb * 2 + 1] = (float)d1y; }
        if (Gd.dcp2) { ((float*)Gd.dcp2)[cb * 2] = (float)d2x; ((float*)Gd.dcp2)[cb * 2 + 1] = (float)d2y; }
      }
    }
    const double dv_h = Gtw(gh_rbar, 0.0);                                   // Jc^T (dh rbar)   (uses L.wv: DX is dead by now)
    if (lane < nz) {
      const size_t o = (size_t)scene * nz + lane;
      const double md = (double)Md[lane], v = (double)vv[lane];
      if (Gd.dMdiag) ((float*)Gd.dMdiag)[o] = (float)(dx * x + dx * v);      // Q = diag(M) (dQ, lcp.py:59-60) and p = M v + dt f
      if (Gd.dv) ((float*)Gd.dv)[o] = (float)(dx * md + dv_h);
      if (Gd.df) ((float*)Gd.df)[o] = (float)(dx * (double)SP.dt);
    }
    if (Gd.dJe && e > 0) {                                                    // dA = dnu (x) x + nu (x) dx (lcp.py:57; A = Je)
      float* o = (float*)Gd.dJe + (size_t)scene * e * nz;
#pragma unroll
      for (int a = 0; a < EQB; ++a) {
        if (a < e) {
          const double dn = bcast_lane(dnu, a), nu = Wit[64 + a];
          if (lane < nz) o[a * nz + lane] = (float)(dn * x + nu * dx);
        }
      }
    }
    if (lane < nb) {                                                          // per-body sums over the contacts, fixed order
      double ar = 0, af = 0;
      for (int c = 0; c < ncs; ++c) {
        const double w = ((B12[c] == lane) ? 1.0 : 0.0) + ((B12[LX + c] == lane) ? 1.0 : 0.0);
        if (w != 0.0) { ar += w * CR[c]; af += w * CF[c]; }
      }
      if (Gd.drest) ((float*)Gd.drest)[(size_t)scene * nb + lane] = (float)ar;
      if (Gd.dfric) ((float*)Gd.dfric)[(size_t)scene * nb + lane] = (float)af;
    }
    return;
  }

  // ---- the PDIPM loop (pdipm.py:49-179) -----------------------------------------------------------------------------------
  const int max_iter = SP.max_iter, lim = SP.lim;
  const double eps = SP.eps;
  const double mf = (double)(4 * ncs);
  double x = 0, y = 0, b = DENSE ? b_in : 0.0;
  M4<double> s = m4<double>(1, 1, 1, 1), z = s, dinv = s;
  double bx = 0, by = 0;
  M4<double> bz = s, bs = s;
  double best_resid = inf_of<double>();
  bool have_best = false, done = false;
  int n_not = 0, iters = 0;
  for (int it = -1; it < max_iter; ++it) {
    double rx = 0, ry = 0, mu = 0, resid = 0;
    M4<double> rs = m4<double>(0, 0, 0, 0), rz = rs;
    if (w0) {
      if (it < 0) {                                                         // init: (p, 0, -h, -b), d = 1 (:57-63)
        rx = p; ry = -b; rz = m4<double>(-hn, 0, 0, 0); dinv = m4<double>(1, 1, 1, 1);
      } else {                                                              // residuals (:82-96)
        rx = Gtw(z.n, z.f1 - z.f2) + qd * x + p;
        if (e > 0) rx += Aty(y);
        rs = z;
        double gn, gt;
        Gv(x, gn, gt);
        rz = m4<double>(gn + s.n - hn, gt + s.f1 - z.g, -gt + s.f2 - z.g, s.g - (mu_c * z.n - (z.f1 + z.f2)));
        if (!vc) rz = m4<double>(0, 0, 0, 0);
        ry = (e > 0) ? (Av(x) - b) : 0.0;
        const double n_rx = wave_sum((lane < nz) ? rx * rx : 0.0);
        const double n_rz = wave_sum(rz.n * rz.n + rz.f1 * rz.f1 + rz.f2 * rz.f2 + rz.g * rz.g);
        const double n_ry = wave_sum((lane < e) ? ry * ry : 0.0);
        const double sz = wave_sum(vc ? (s.n * z.n + s.f1 * z.f1) + (s.f2 * z.f2 + s.g * z.g) : 0.0);
        mu = sz / mf; mu = mu < 0 ? -mu : mu;                               // (:91)
        resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + mf * mu;             // (:92-96)
        dinv = vc ? m4<double>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g) : m4<double>(1, 1, 1, 1);   // 1 / d, d = z / s (:98)
      }
      reduce_setup(dinv);
    }
    BIG_TICK(0)                                                             // residuals
    __syncthreads();
    BIG_TICK(4)
    factor();                                                               // (:99-100)
    __syncthreads();
    BIG_TICK(1)                                                             // factorisation
    if (w0) {
      ua = L.dU[lc]; uu = L.dU[NCB + lc];
      const bool singular = L.flag[0] != 0;
      if (it >= 0 && !done) {
        ++iters;
        if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; done = true; }   // except: return best (:99-102)
        else {
          const bool improved = !have_best || (resid < best_resid);             // (:107-132)
          if (improved) { best_resid = resid; n_not = 0; have_best = true; bx = x; by = y; bz = z; bs = s; }
          else ++n_not;
          if (n_not == lim || best_resid < eps || mu > mu_limit<double>()) done = true;   // (:133)
        }
      }
      // (the iterate the last pass would produce is never evaluated - pdipm.py:176-179 - so its solves are skipped)
      if (it >= 0 && it == max_iter - 1) done = true;
      if (!done) {
        double ax = 0, ay = 0;
        M4<double> as_ = m4<double>(0, 0, 0, 0), az = as_;
        const int npass = (it < 0) ? 1 : 2;
        for (int pass = 0; pass < npass; ++pass) {
          double ox, oy;
          M4<double> os, oz;
          BIG_TICK(2)
          solve_kkt(dinv, rx, rs, rz, ry, ox, os, oz, oy, pass == 1);
          BIG_TICK(3)                                                           // solve_kkt
          if (it < 0) {
            x = ox; s = os; z = oz; y = oy;                                     // (:60-63)
            auto min4 = [&](const M4<double>& a) { return pmin(pmin(a.n, a.f1), pmin(a.f2, a.g)); };
            const uint32_t ks = wave_umax(vc ? umax(umax(nan_key(s.n), nan_key(s.f1)), umax(nan_key(s.f2), nan_key(s.g))) : 0u);
            const uint32_t kz = wave_umax(vc ? umax(umax(nan_key(z.n), nan_key(z.f1)), umax(nan_key(z.f2), nan_key(z.g))) : 0u);
            double smin = wave_min(vc ? min4(s) : inf_of<double>()), zmin = wave_min(vc ? min4(z) : inf_of<double>());
            if (key_is_nan(ks)) smin = nan_of<double>();
            if (key_is_nan(kz)) zmin = nan_of<double>();
            if (smin <= 0.0) { const double sh = 1.0 - smin; s = m4<double>(s.n + sh, s.f1 + sh, s.f2 + sh, s.g + sh); }   // (:66-75)
            if (zmin <= 0.0) { const double sh = 1.0 - zmin; z = m4<double>(z.n + sh, z.f1 + sh, z.f2 + sh, z.g + sh); }
            if (!vc) { s = m4<double>(1, 1, 1, 1); z = s; }
            if (ncs == 0) { bx = x; by = y; done = true; }                      // engines.py:36-50: x = P^-1 u, no LCP
          } else if (pass == 0) {
            ax = ox; ay = oy; as_ = os; az = oz;                                // affine direction (:138-139)
            const double alpha = pmin(step_pair(z, az, s, as_), 1.0);          // (:142-144)
            auto sc = [&](double sv, double dsv, double zv, double dzv) { return (sv + alpha * dsv) * (zv + alpha * dzv); };
            const double t3 = wave_sum(vc ? (sc(s.n, as_.n, z.n, az.n) + sc(s.f1, as_.f1, z.f1, az.f1)) + (sc(s.f2, as_.f2, z.f2, az.f2) + sc(s.g, as_.g, z.g, az.g)) : 0.0);
            const double t4 = wave_sum(vc ? (s.n * z.n + s.f1 * z.f1) + (s.f2 * z.f2 + s.g * z.g) : 0.0);
            const double r3 = t3 / t4, sig = r3 * r3 * r3;                      // (:146-150)
            const double ms = -mu * sig;
            rx = 0; ry = 0; rz = m4<double>(0, 0, 0, 0);
            rs = vc ? m4<double>((ms + as_.n * az.n) / s.n, (ms + as_.f1 * az.f1) / s.f1, (ms + as_.f2 * az.f2) / s.f2, (ms + as_.g * az.g) / s.g)
                    : m4<double>(0, 0, 0, 0);                                   // (:153)
          } else {
            const double cx = ox + ax, cy = oy + ay;                            // (:160-163)
            const M4<double> cs = m4<double>(os.n + as_.n, os.f1 + as_.f1, os.f2 + as_.f2, os.g + as_.g);
            const M4<double> cz = m4<double>(oz.n + az.n, oz.f1 + az.f1, oz.f2 + az.f2, oz.g + az.g);
            const double alpha = pmin(0.999 * step_pair(z, cz, s, cs), 1.0);   // (:164-166)
            x += alpha * cx; y += alpha * cy;                                   // (:171-174)
            if (vc) {
              s = m4<double>(s.n + alpha * cs.n, s.f1 + alpha * cs.f1, s.f2 + alpha * cs.f2, s.g + alpha * cs.g);
              z = m4<double>(z.n + alpha * cz.n, z.f1 + alpha * cz.f1, z.f2 + alpha * cz.f2, z.g + alpha * cz.g);
            }
          }
        }
      }
      if (lane == 0) L.flag[1] = done ? 1 : 0;
    }
    BIG_TICK(2)                                                             // step lengths, bookkeeping
    __syncthreads();
    if (L.flag[1]) break;
  }


  // ---- outputs (row layout of a capacity-sized LCP, padded slots 0) ---------------------------------------------------------
  if (!w0) return;
  bool bad = (lane < nz) && (bx != bx);
  if (vc) bad = bad || (bz.n != bz.n) || (bs.n != bs.n) || (bz.f1 != bz.f1) || (bz.f2 != bz.f2) || (bz.g != bz.g) ||
                (bs.f1 != bs.f1) || (bs.f2 != bs.f2) || (bs.g != bs.g);
  if (__any(bad)) status |= LCP_ST_NAN;
  const int m = 4 * ncap;
  if (lane < ncap) {
    const float k = vc ? 1.0f : 0.0f;
    if (SP.z) { float* o = (float*)SP.z + (size_t)scene * m; o[lane] = k * (float)bz.n; o[ncap + 2 * lane] = k * (float)bz.f1; o[ncap + 2 * lane + 1] = k * (float)bz.f2; o[3 * ncap + lane] = k * (float)bz.g; }
    if (SP.s) { float* o = (float*)SP.s + (size_t)scene * m; o[lane] = k * (float)bs.n; o[ncap + 2 * lane] = k * (float)bs.f1; o[ncap + 2 * lane + 1] = k * (float)bs.f2; o[3 * ncap + lane] = k * (float)bs.g; }
  }
  if (lane < e && SP.y) ((float*)SP.y)[(size_t)scene * e + lane] = (float)by;
  if (lane < nz) {
    const double nv = DENSE ? bx : -bx;                                               // engines.py:76-77 (dense boundary: zhats = x itself, lcp.py:35)
    ((float*)SP.v_new)[(size_t)scene * nz + lane] = (float)nv;
    if (SP.p_new) ((float*)SP.p_new)[(size_t)scene * nz + lane] = (float)((double)((const float*)SP.pos)[(size_t)scene * nz + lane] + nv * SP.dt);   // bodies.py:81
  }
  // the iterate the backward starts from (lcp.py:29 keeps nus, lams, slacks on the op)
  Wit[lane] = (lane < nz) ? bx : 0.0;
  if (lane < 8) Wit[64 + lane] = (lane < e) ? by : 0.0;
  Wit[72 + lane] = bz.n; Wit[72 + LX + lane] = bz.f1; Wit[72 + 2 * LX + lane] = bz.f2; Wit[72 + 3 * LX + lane] = bz.g;
  Wit[72 + 4 * LX + lane] = bs.n; Wit[72 + 5 * LX + lane] = bs.f1; Wit[72 + 6 * LX + lane] = bs.f2; Wit[72 + 7 * LX + lane] = bs.g;
  if (lane == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
#ifdef LCP_BIG_PROFILE
  __builtin_amdgcn_s_waitcnt(0);
  if (tid == 0 && SP.s) { float* o = (float*)SP.s + (size_t)scene * 4 * ncap + (4 * ncap - 8); for (int i = 0; i < 4; ++i) o[i] = (float)pc[i]; o[4] = (float)pc_ts; o[5] = (float)pc[5]; o[6] = (float)pc[6];
    if (SP.z) { float* oz = (float*)SP.z + (size_t)scene * 4 * ncap + (4 * ncap - 24); for (int i = 0; i < 20; ++i) oz[i] = (float)L.prow[i]; } }
#endif
}

}  // namespace big

// capacity class of a scene batch: 16, 32 or 64 contacts
// (up to 16 contacts with more than ten bodies - beyond lcp_quad.hip - run in the 32-contact class: its panel steps without live
//  contacts are skipped; the 256-thread 16-contact instantiation of the rank-1 form only exists in LCP_BIG_MFMA = 0 builds)
static inline int big_class(int nc) { return (LCP_BIG_MFMA != 0) ? (nc <= 32 ? 32 : 64) : (nc <= 16 ? 16 : (nc <= 32 ? 32 : 64)); }
// 64 contacts: nz <= 43 (the two 64 x nz Jacobians have to fit next to the 129 KB of factors in the 160 KB of LDS);
// smaller classes: nz <= 48 (16 bodies, the contact kernel's limit)
bool big_supported(int nz, int m, int e) {
  if ((m % 4) != 0 || m / 4 > 64 || e > big::EQB) return false;
  return (m / 4 > 32) ? nz <= 43 : nz <= 48;
}
size_t big_ws_bytes(int m) {
  const int c = big_class(m / 4);
  const int total = c == 16 ? big::WsLayout<16>::TOTAL : (c == 32 ? big::WsLayout<32>::TOTAL : big::WsLayout<64>::TOTAL);
  return sizeof(double) * (size_t)total;
}

template <int NCB, bool BWD, bool DENSE = false>
static int big_launch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream, const DenseIO& DN = DenseIO{}) {
  const int nz = DENSE ? DN.nz : 3 * SP.nb, nzs = nz | 1;
  big::Lds L;
  const size_t lds = big::carve<NCB>(L, nullptr, nzs);
  auto k = big::lcp_big_kernel<NCB, BWD, DENSE>;
  if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return LCP_E_LAUNCH;
  hipLaunchKernelGGL(k, dim3(SP.B), dim3(big::Grid<NCB>::NT), lds, (hipStream_t)stream, SP, Gd, nzs, DN);
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}
template <bool BWD>
static int big_dispatch(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) {
  switch (big_class(SP.nc)) {
#if !LCP_BIG_MFMA
    case 16: return big_launch<16, BWD>(SP, Gd, stream);
#endif
    case 32: return big_launch<32, BWD>(SP, Gd, stream);
    default: return big_launch<64, BWD>(SP, Gd, stream);
  }
}
int big_step(const StepArgs& SP, void* stream) { StepBwdArgs Gd = {}; return big_dispatch<false>(SP, Gd, stream); }
int big_step_backward(const StepArgs& SP, const StepBwdArgs& Gd, void* stream) { return big_dispatch<true>(SP, Gd, stream); }

// ---------------------------------------------------------------- dense boundary (LCPFunction sizes 17 .. 64 contacts)
namespace big {
// One workgroup per scene: is the dense LCP the mixed contact LCP of engines.py:50-74 with a diagonal Q ?
//   G = [Jc; Jf; 0] with Jf rows in (+jt, -jt) pairs, F = [[0, 0, 0], [0, 0, E], [mu, -E^T, 0]], h = [h_n; 0; 0], Q diagonal.
// cls[scene] = 2 if so (lcp_big_kernel<.., DENSE> serves it), else 0 (the generic kernels do).
// ... 3 if, in addition, `primal_ok` (the sizes fit lcp_primal.hip) and no contact's rows touch more than two bodies (columns in
// body triples): the body-space kernel takes those; 4 if, on top of that, `pin_ok` (the sizes of lcp_primal_pin.hip) and the equality
// rows pin the leading coordinates - A = [I 0], b = 0, the TotalConstraint on the floor of the reference's worlds - : its pinned form.
// Round 5: this pass is the ONLY one that reads the scene's dense tensors in full (F: 16-byte loads, eight in flight per lane); what the
// body-space kernels need of them - per contact the six entries of its Jc / Jt rows on its two bodies, the bodies' first columns, mu and
// h_n - it leaves in the scene's workspace block (`DENSE_EXTRACT_OFF`, component-major: 16 x ncap floats), so that the solver neither
// scans the 33 KB of G again nor gathers the diagonal of F from 64 cache lines.
#ifndef LCP_CLS_UNROLL
#define LCP_CLS_UNROLL 8      // 16-byte loads of F in flight per lane
#endif
#ifndef LCP_CLS_NT
#define LCP_CLS_NT 1          // F read with non-temporal loads: it is read once (A/B round 5: 0.668 -> 0.653 ms forward call; 8 or 16 loads in flight: no difference)
#endif
__global__ void __launch_bounds__(256) lcp_classify_big(DenseIO DN, int e, int primal_ok, int pin_ok, unsigned char* ws) {
  const int scene = blockIdx.x, tid = threadIdx.x, nz = DN.nz, m = DN.m, nc = m >> 2;
  const float* G = DN.G + (size_t)scene * m * nz;
  const float* F = DN.F + (size_t)scene * m * m;
  const float* Q = DN.Q + (size_t)scene * nz * nz;
  const float* h = DN.h + (size_t)scene * m;
  int good = 1;                                                            // (accumulated with &: no short-circuit between loads)
  // F, four consecutive columns of one row per load (m is a multiple of four: a float4 never straddles a row)
  {
    const float4* F4 = reinterpret_cast<const float4*>(F);
    const int q4 = m >> 2, total = m * q4;
    for (int i0 = tid; i0 < total; i0 += 256 * LCP_CLS_UNROLL) {
      float4 v[LCP_CLS_UNROLL];
#pragma unroll
      for (int u = 0; u < LCP_CLS_UNROLL; ++u) {
        const int i = i0 + 256 * u;
#if LCP_CLS_NT
        typedef float v4f __attribute__((ext_vector_type(4)));
        if (i < total) { const v4f w = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(F4 + i)); v[u] = make_float4(w.x, w.y, w.z, w.w); }
        else v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
#else
        v[u] = (i < total) ? F4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
      }
#pragma unroll
      for (int u = 0; u < LCP_CLS_UNROLL; ++u) {
        const int i = i0 + 256 * u;
        if (i < total) {
          const int r = i / q4, j0 = 4 * (i - r * q4);
          const float e4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = j0 + t;
            float want = 0.0f;
            if (r >= nc && r < 3 * nc) want = (j == 3 * nc + ((r - nc) >> 1)) ? 1.0f : 0.0f;
            else if (r >= 3 * nc) { const int cg = r - 3 * nc; if (j == cg) want = e4[t]; else if (j == nc + 2 * cg || j == nc + 2 * cg + 1) want = -1.0f; }
            good &= (e4[t] == want) ? 1 : 0;
          }
        }
      }
    }
  }
  for (int i = tid; i < nc * nz; i += 256) { const int c = i / nz, k = i - c * nz; good &= (G[(size_t)(nc + 2 * c + 1) * nz + k] == -G[(size_t)(nc + 2 * c) * nz + k]) ? 1 : 0; }
  for (int i = tid; i < nc * nz; i += 256) good &= (G[(size_t)3 * nc * nz + i] == 0.0f) ? 1 : 0;
  for (int i = nc + tid; i < m; i += 256) good &= (h[i] == 0.0f) ? 1 : 0;
  for (int i = tid; i < nz * nz; i += 256) { const int r = i / nz, c = i - r * nz; good &= (r == c || Q[i] == 0.0f) ? 1 : 0; }
  // the per-contact records (and "at most two bodies"): four lanes per contact scan the bodies of its two rows, the contact's lane
  // writes the record
  __shared__ int sbf[64], sbl[64], sbn[64];
  if (tid < 64) { sbf[tid] = 1 << 30; sbl[tid] = -1; sbn[tid] = 0; }
  __syncthreads();
  int two = 1;
  if (primal_ok) {
    const int c = tid >> 2, part = tid & 3, nbod = nz / 3;
    if (c < nc) {
      const float* gc = G + (size_t)c * nz; const float* gt = G + (size_t)(nc + 2 * c) * nz;
      for (int bq = part; bq < nbod; bq += 4) {
        const bool nzb = (gc[3 * bq] != 0.0f) || (gc[3 * bq + 1] != 0.0f) || (gc[3 * bq + 2] != 0.0f) ||
                         (gt[3 * bq] != 0.0f) || (gt[3 * bq + 1] != 0.0f) || (gt[3 * bq + 2] != 0.0f);
        if (nzb) { atomicMin(&sbf[c], bq); atomicMax(&sbl[c], bq); atomicAdd(&sbn[c], 1); }
      }
    }
    __syncthreads();
    if (tid < nc) {
      const float* gc = G + (size_t)tid * nz; const float* gt = G + (size_t)(nc + 2 * tid) * nz;
      two = sbn[tid] <= 2 ? 1 : 0;
      int bf = sbf[tid], bl = sbl[tid];
      if (bl < 0) { bf = 0; bl = 0; }
      float* rec = reinterpret_cast<float*>(ws + (size_t)scene * DN.ws_scene + DENSE_EXTRACT_OFF);
      const int c0 = 3 * bf, c1 = 3 * bl;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        rec[q * nc + tid] = gc[c0 + q]; rec[(6 + q) * nc + tid] = gt[c0 + q];
        rec[(3 + q) * nc + tid] = (bl != bf) ? gc[c1 + q] : 0.0f; rec[(9 + q) * nc + tid] = (bl != bf) ? gt[c1 + q] : 0.0f;
      }
      rec[12 * nc + tid] = __int_as_float(c0); rec[13 * nc + tid] = __int_as_float(c1);
      rec[14 * nc + tid] = F[(size_t)(3 * nc + tid) * m + tid];             // mu_c (engines.py:71)
      rec[15 * nc + tid] = h[tid];                                          // h_n (:74)
    }
  }
  int pinned = 1;
  if (pin_ok) {
    const float* A = DN.A + (size_t)scene * e * nz;
    for (int i = tid; i < e * nz; i += 256) { const int a = i / nz, k = i - a * nz; pinned &= (A[i] == ((k == a) ? 1.0f : 0.0f)) ? 1 : 0; }
    if (tid < e) pinned &= (DN.b[(size_t)scene * e + tid] == 0.0f) ? 1 : 0;
  }
  const int all = __syncthreads_and(good);
  const int sparse = __syncthreads_and(two);
  const int pin = __syncthreads_and(pinned);
  if (tid == 0) DN.cls[scene] = all ? ((primal_ok && sparse) ? ((pin_ok && pin) ? 4 : 3) : 2) : 0;
}
}  // namespace big

bool big_dense_supported(int nz, int m, int e) {
  if ((m % 4) != 0 || m / 4 <= 16 || !big_supported(nz, m, e)) return false;      // (<= 16 contacts: the wave-per-scene kernels)
  return nz <= 64;
}

static void dense_io(DenseIO& DN, int nz, int m, int32_t* cls, size_t ws_scene) { DN = DenseIO{}; DN.nz = nz; DN.m = m; DN.cls = cls; DN.ws_scene = ws_scene; }

int big_dense_forward(const FwdArgs& P, int32_t* cls, size_t ws_scene, int primal_ok, void* stream) {
  DenseIO DN;
  dense_io(DN, P.nz, P.m, cls, ws_scene);
  DN.Q = (const float*)P.Q; DN.p = (const float*)P.p; DN.G = (const float*)P.G; DN.h = (const float*)P.h;
  DN.A = (const float*)P.A; DN.b = (const float*)P.b; DN.F = (const float*)P.F;
  hipLaunchKernelGGL(big::lcp_classify_big, dim3(P.B), dim3(256), 0, (hipStream_t)stream, DN, P.e, primal_ok & 1, (primal_ok >> 1) & 1, (unsigned char*)P.ws);
  StepArgs SP = {};
  SP.B = P.B; SP.nb = (P.nz + 2) / 3; SP.nc = P.m / 4; SP.e = P.e; SP.ws = P.ws;
  SP.eps = P.eps; SP.max_iter = P.max_iter; SP.lim = P.lim;
  SP.v_new = P.x; SP.z = P.z; SP.s = P.s; SP.y = P.y; SP.iters = P.iters; SP.status = P.status;
  SP.tag = P.tag; SP.tag_value = P.tag_value;
  StepBwdArgs Gd = {};
  return big_class(SP.nc) == 32 ? big_launch<32, false, true>(SP, Gd, stream, DN) : big_launch<64, false, true>(SP, Gd, stream, DN);
}

int big_dense_backward(const BwdArgs& P, int32_t* cls, size_t ws_scene, void* stream) {
  DenseIO DN;
  dense_io(DN, P.nz, P.m, cls, ws_scene);
  DN.G = (const float*)P.G; DN.A = (const float*)P.A; DN.dl_dx = (const float*)P.dl_dx;
  DN.dQ = (float*)P.dQ; DN.dp = (float*)P.dp; DN.dG = (float*)P.dG; DN.dh = (float*)P.dh; DN.dA = (float*)P.dA; DN.db = (float*)P.db; DN.dF = (float*)P.dF;
  StepArgs SP = {};
  SP.B = P.B; SP.nb = (P.nz + 2) / 3; SP.nc = P.m / 4; SP.e = P.e; SP.ws = P.ws;
  SP.tag = (int32_t*)P.tag; SP.tag_value = P.tag_value;
  StepBwdArgs Gd = {};
  return big_class(SP.nc) == 32 ? big_launch<32, true, true>(SP, Gd, stream, DN) : big_launch<64, true, true>(SP, Gd, stream, DN);
}

}  // namespace lcp
