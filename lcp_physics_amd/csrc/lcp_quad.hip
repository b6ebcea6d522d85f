// lcp_quad.hip - contact-structured PDIPM kernels, FOUR SCENES PER WAVEFRONT (16 lanes per scene).
//
// The fastest path: contact-structured LCPs (engines.py:67-73) with diagonal Q, nc <= 16 contacts,
// nz <= 16 (dense inputs) or nz <= 32 (contact-list inputs, template parameter XH = 2), neq <= 4, fp32 I/O.
// Mapping (CDNA4-first):
//   * one 16-lane DPP row = one scene, lane c = contact c.  The lane holds everything that belongs to its
//     contact: the four inequality components (normal, friction +, friction -, gamma) of every m-space
//     vector, its rows of Jc and Jt, and BOTH of its rows of the reduced 2nc x 2nc system (a_c and u_c,
//     see lcp_wave64.hip `Red` for the algebra).  The per-contact 2x2 elimination, F z, the reduction's
//     right-hand side and back-substitution are therefore lane-local - no cross-lane traffic at all.
//   * x-space vectors live one entry per lane of the row (two with XH = 2), e-space (neq <= 4) likewise.
//   * every cross-lane move is a DPP row operation: `row_newbcast:k` broadcasts lane k of each row to its row
//     (pivot rows in the LU, vector entries in the products), quad_perm / row_mirror give row-local
//     reductions.  Micro-benchmark (tools/microbench/pair_cost.hip): 6.7 cycles per fp64 FMA fed by a 64-bit DPP
//     move, 4.7 with the broadcast folded into v_fmac_f64_dpp (the LU), versus 21-29 through v_readlane, and one
//     instruction serves four scenes.
//   * the pivot-free LU runs over the 2 x 32 register-resident row entries of each lane.
// Scenes of a wave advance in lock-step; a scene that terminates (pdipm.py:133) freezes its state and waits.  Scenes may
// have different contact counts (the fused step with `c_count`): lane masks are per row, loops run to the wave maximum
// over identity rows, a scene without contacts leaves after the initialisation solve (engines.py:36-50).
// Kernels: lcp_fwd_quad (dense or fused forward), lcp_bwd_quad (lcp.py:37-64, dense gradients), lcp_bwd_step_quad
// (gradients w.r.t. the physical inputs of the fused step).
// lcp_fwd_quad comes in three factorisations (template parameter ALG): 0 = the reduced 2 nc x 2 nc contact-space system described
// above (the dense boundary, nz > 16, fp32 arithmetic); 1 = the BODY-space system K = [[Q + G^T M^-1 G, A^T], [A, 0]] of nz + neq <= 20
// rows (x-row and equality row per lane; lcp_primal.hip has the algebra); 2 = the same with the pinned leading coordinates of a
// fixed floor (A = [I 0]) taken out - one row per lane, nz - neq pivots - alone in its kernel.  The contact-list entry points
// launch 2 and, behind it, 1 for the waves 2 declined; the backward kernels stay in contact space (factor_bwd_q).
// Same algorithm and the same reference lines as lcp_wave64.hip / lcp_generic.hip.
#include "lcp_quad_prims.h"

// Measured on the MI355X (profiles/r03_ab_occupancy.txt) and NOT the default: with its contact's rows of Jc / Jt read from LDS (LCP_Q_ROWL)
// instead of held in 32 registers the pinned body-space kernel fits 239 registers - two wavefronts per SIMD (LCP_Q_OCC2 = 2) - but the
// forward of 4096 scenes goes from 0.096 to 0.132 ms (one wave per SIMD waits out every LDS round trip), and at 32768 scenes, where
// two waves do share a SIMD, 0.749 against 0.716 ms: a second wave does not buy back what the LDS reads cost.
#ifndef LCP_Q_OCC2
#define LCP_Q_OCC2 1          // wavefronts per SIMD the pinned body-space kernel is allocated for (2: at most 256 unified registers)
#endif
#ifndef LCP_Q_ROWL
#define LCP_Q_ROWL 0          // 1: the pinned body-space kernel reads its contact's rows of Jc / Jt from LDS instead of keeping them in registers
#endif
#ifndef LCP_Q_BEST_LDS
#define LCP_Q_BEST_LDS 1      // 0: the pinned body-space kernel keeps its best iterate in registers (A/B aid)
#endif

namespace lcp {
namespace q16 {

// ---------------------------------------------------------------- per-scene LDS block
template <typename TI, typename TC>
struct LdsQ {
  TI* GL;    // [16][16]  Jc rows: GL[c*16 + j]  (lane j of G^T w reads GL[c*16 + j]: consecutive lanes, consecutive banks.
             //           The transposed layout - 16 contiguous entries per lane - was tried: 8-way bank conflicts, slower.)
  TI* GTL;   // [16][16]  Jt rows
  TI* AtL;   // [EQ][16]  A rows
  TC* GAL;   // [16][2][EQ]  (J Q^-1 A^T) of the n / t row of every contact
  TC* S11;   // [EQ][EQ]     (A Q^-1 A^T)^-1
  TC* WL;    // [528]        W = J P J^T of the reduced system, upper triangle packed by rows (forward kernel only):
             //              the 11 factorisations of a solve rebuild T = W + diag from here instead of re-reading HBM
};
constexpr int NRED = 32;                                   // rows of the reduced system: a_0..a_15, u_0..u_15
__host__ __device__ constexpr int wl_row(int i) { return i * NRED - (i * (i - 1)) / 2 - i; }   // (i, j >= i) lives at wl_row(i) + j
constexpr int WL_ELEMS = NRED * (NRED + 1) / 2;
// `xh` = x-space halves: 1 (nz <= 16, one entry per lane) or 2 (nz <= 32, entries j and 16 + j per lane); it is the row
// length of GL / GTL / AtL in units of 16
template <typename TI, typename TC>
__host__ __device__ inline size_t carve_q(LdsQ<TI, TC>& L, unsigned char* smem, bool with_w, int xh = 1, bool with_gal = true) {
  unsigned char* q = smem;
  auto take = [&](size_t bytes) { unsigned char* r = q; q += (bytes + 15) & ~(size_t)15; return r; };
  L.GAL = with_gal ? (TC*)take(sizeof(TC) * NCQ * 2 * EQ) : nullptr;      // (contact-space pre-factorisation; scratch of the backward kernels)
  L.S11 = with_gal ? (TC*)take(sizeof(TC) * EQ * EQ) : nullptr;
  L.GL = (TI*)take(sizeof(TI) * NCQ * 16 * xh);
  L.GTL = (TI*)take(sizeof(TI) * NCQ * 16 * xh);
  L.AtL = (TI*)take(sizeof(TI) * EQ * 16 * xh);
  L.WL = with_w ? (TC*)take(sizeof(TC) * WL_ELEMS + 16) : nullptr;   // (+16 B: scene blocks do not all start on the same LDS bank)
  return (size_t)(q - smem);
}

// ---------------------------------------------------------------- J v / J^T w products as v_fmac_f64_dpp blocks (fp64, nz <= 16)
// acc += (lane K of the row's `src`) * mult in ONE instruction instead of v_mov_b64_dpp + v_fma_f64 (same FMA, same accumulators in the
// same order: bitwise the builtin form).  Rules as for the LU blocks below: `s_nop 1` first, DPP sources read-only in the statement.
#define LCP_DPP_FULL_EARLY "row_mask:0xf bank_mask:0xf"
#define LCP_GV_ONE(ACC, M, K) "v_fmac_f64_dpp %[" #ACC "], %[v], %[" #M "] row_newbcast:%[" #K "] " LCP_DPP_FULL_EARLY "\n\t"
// columns J0 .. J0+7 of (Jc v, Jt v): n0 / t0 take the even columns, n1 / t1 the odd ones
template <int J0> __device__ __forceinline__ void gv8_dpp(double& n0, double& n1, double& t0, double& t1, double v, const double (&c)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t"
      LCP_GV_ONE(n0, c0, k0) LCP_GV_ONE(t0, t0_, k0) LCP_GV_ONE(n1, c1, k1) LCP_GV_ONE(t1, t1_, k1)
      LCP_GV_ONE(n0, c2, k2) LCP_GV_ONE(t0, t2_, k2) LCP_GV_ONE(n1, c3, k3) LCP_GV_ONE(t1, t3_, k3)
      LCP_GV_ONE(n0, c4, k4) LCP_GV_ONE(t0, t4_, k4) LCP_GV_ONE(n1, c5, k5) LCP_GV_ONE(t1, t5_, k5)
      LCP_GV_ONE(n0, c6, k6) LCP_GV_ONE(t0, t6_, k6) LCP_GV_ONE(n1, c7, k7) LCP_GV_ONE(t1, t7_, k7)
      : [n0] "+v"(n0), [n1] "+v"(n1), [t0] "+v"(t0), [t1] "+v"(t1)
      : [v] "v"(v), [c0] "v"(c[0]), [c1] "v"(c[1]), [c2] "v"(c[2]), [c3] "v"(c[3]), [c4] "v"(c[4]), [c5] "v"(c[5]), [c6] "v"(c[6]), [c7] "v"(c[7]),
        [t0_] "v"(t[0]), [t1_] "v"(t[1]), [t2_] "v"(t[2]), [t3_] "v"(t[3]), [t4_] "v"(t[4]), [t5_] "v"(t[5]), [t6_] "v"(t[6]), [t7_] "v"(t[7]),
        [k0] "n"(J0), [k1] "n"(J0 + 1), [k2] "n"(J0 + 2), [k3] "n"(J0 + 3), [k4] "n"(J0 + 4), [k5] "n"(J0 + 5), [k6] "n"(J0 + 6), [k7] "n"(J0 + 7));
}
// contacts C0 .. C0+7 of J^T w: a0 += wn_C g_C, a1 += wt_C t_C for the even contacts, a2 / a3 for the odd ones
#define LCP_GTW_ONE(ACC, SRC, M, K) "v_fmac_f64_dpp %[" #ACC "], %[" #SRC "], %[" #M "] row_newbcast:%[" #K "] " LCP_DPP_FULL_EARLY "\n\t"
template <int C0> __device__ __forceinline__ void gtw8_dpp(double& a0, double& a1, double& a2, double& a3, double wn, double wt, const double (&g)[8], const double (&t)[8]) {
  asm("s_nop 1\n\t"
      LCP_GTW_ONE(a0, wn, g0, k0) LCP_GTW_ONE(a1, wt, t0_, k0) LCP_GTW_ONE(a2, wn, g1, k1) LCP_GTW_ONE(a3, wt, t1_, k1)
      LCP_GTW_ONE(a0, wn, g2, k2) LCP_GTW_ONE(a1, wt, t2_, k2) LCP_GTW_ONE(a2, wn, g3, k3) LCP_GTW_ONE(a3, wt, t3_, k3)
      LCP_GTW_ONE(a0, wn, g4, k4) LCP_GTW_ONE(a1, wt, t4_, k4) LCP_GTW_ONE(a2, wn, g5, k5) LCP_GTW_ONE(a3, wt, t5_, k5)
      LCP_GTW_ONE(a0, wn, g6, k6) LCP_GTW_ONE(a1, wt, t6_, k6) LCP_GTW_ONE(a2, wn, g7, k7) LCP_GTW_ONE(a3, wt, t7_, k7)
      : [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3)
      : [wn] "v"(wn), [wt] "v"(wt), [g0] "v"(g[0]), [g1] "v"(g[1]), [g2] "v"(g[2]), [g3] "v"(g[3]), [g4] "v"(g[4]), [g5] "v"(g[5]), [g6] "v"(g[6]), [g7] "v"(g[7]),
        [t0_] "v"(t[0]), [t1_] "v"(t[1]), [t2_] "v"(t[2]), [t3_] "v"(t[3]), [t4_] "v"(t[4]), [t5_] "v"(t[5]), [t6_] "v"(t[6]), [t7_] "v"(t[7]),
        [k0] "n"(C0), [k1] "n"(C0 + 1), [k2] "n"(C0 + 2), [k3] "n"(C0 + 3), [k4] "n"(C0 + 4), [k5] "n"(C0 + 5), [k6] "n"(C0 + 6), [k7] "n"(C0 + 7));
}
#ifndef LCP_Q_ASM_PRODUCTS
#define LCP_Q_ASM_PRODUCTS 1
#endif

// ---------------------------------------------------------------- per-lane scene data and products
// An x-space vector: entry 16 h + l16 of the scene's nz-vector for h < XH (XH = 1: nz <= 16, the tuned headline case;
// XH = 2: nz <= 32, six to ten bodies).
template <typename TC, int XH> struct XV { TC v[XH]; };

// ROWL: the lane's own rows of Jc / Jt are not kept in registers (32 of them) but read from their LDS copies at every use (the
// pinned body-space kernel: its register budget is what decides how many wavefronts share a SIMD)
template <typename TI, typename TC, int XH = 1, bool ROWL = false>
struct SceneQ {
  static constexpr int RS = 16 * XH;     // row length of GL / GTL / AtL
  LdsQ<TI, TC> L;
  int nz, nc, e, l16;      // nc: live contacts of THIS scene (row-uniform)
  int ncw, ncap;          // ncw: max nc over the scenes of the wave (loop bound); ncap: contact capacity (array strides)
  TI jc[ROWL ? 1 : RS], jt[ROWL ? 1 : RS];      // rows of Jc and Jt of this lane's contact (Jf rows are +jt, -jt: world.py:191-192)
  template <int J> __device__ __forceinline__ TI jcv(int oz) const { if constexpr (ROWL) return L.GL[l16 * RS + J + oz]; else return launder(jc[J]); }
  template <int J> __device__ __forceinline__ TI jtv(int oz) const { if constexpr (ROWL) return L.GTL[l16 * RS + J + oz]; else return launder(jt[J]); }
  TC gan[EQ], gat[EQ];    // (J Q^-1 A^T) rows of this contact
  TC s11row[EQ];          // row l16 of (A Q^-1 A^T)^-1
  TC qd[XH], qid[XH];     // Q[j][j], 1 / Q[j][j] for j = 16 h + l16
  TC mu;                  // friction coefficient of this contact

  // m-space <- x-space:  (Jc v)_c and (Jt v)_c
  // (the wave is alone on its SIMD: a single accumulator would serialise on the FMA latency, so every product below
  //  runs two to four independent partial sums)
  // ASMP: the v_fmac_f64_dpp form (the body-space forward asks for it: -1.3 % there; the backward kernels got 3 % slower with it)
  template <bool ASMP = false>
  __device__ __forceinline__ void Gv(const XV<TC, XH>& v, TC& gn, TC& gt) const {
    TC n0 = 0, n1 = 0, t0 = 0, t1 = 0;
    if constexpr (ASMP && LCP_Q_ASM_PRODUCTS && std::is_same<TC, double>::value && XH == 1) {
      static_for<2>([&](auto Hh) LCP_INL {
        constexpr int J0 = 8 * Hh;
        double c[8], t[8];
        const int ozr = ROWL ? lds_opaque_zero() : 0;
        static_for<8>([&](auto I) LCP_INL { c[I] = (double)this->template jcv<J0 + I>(ozr); t[I] = (double)this->template jtv<J0 + I>(ozr); });
        gv8_dpp<J0>(n0, n1, t0, t1, v.v[0], c, t);
      });
      gn = n0 + n1; gt = t0 + t1;
      return;
    }
    static_for<8 * XH>([&](auto H) LCP_INL {
      constexpr int J = 2 * H, hx = J >> 4;
      fmac_bc<J & 15>(n0, v.v[hx], (TC)jcv<J>(0)); fmac_bc<J & 15>(t0, v.v[hx], (TC)jtv<J>(0));
      fmac_bc<(J + 1) & 15>(n1, v.v[hx], (TC)jcv<J + 1>(0)); fmac_bc<(J + 1) & 15>(t1, v.v[hx], (TC)jtv<J + 1>(0));
    });
    gn = n0 + n1; gt = t0 + t1;
  }
  // x-space <- m-space:  (G^T w)_j = sum_c Jc[c][j] w_n,c + Jt[c][j] (w_f1,c - w_f2,c)
  template <bool ASMP = false>
  __device__ __forceinline__ XV<TC, XH> Gtw(TC wn, TC wt) const {
    XV<TC, XH> out;
    const int oz = lds_opaque_zero();
    static_for<XH>([&](auto HX) LCP_INL {
      TC a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      const TI* gl = L.GL + 16 * HX + l16 + oz;
      const TI* gtl = L.GTL + 16 * HX + l16 + oz;
      // two batches of 16 LDS loads, each fenced from its FMAs: left alone, the register-starved scheduler issues one
      // ds_read per FMA pair and waits out the LDS latency sixteen times per product
      static_for<2>([&](auto Hh) LCP_INL {
        constexpr int C0 = 8 * Hh;
        TI gv[8], tv[8];
        static_for<8>([&](auto I) LCP_INL { gv[I] = gl[(C0 + I) * RS]; tv[I] = gtl[(C0 + I) * RS]; });
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ASMP && LCP_Q_ASM_PRODUCTS && std::is_same<TC, double>::value) {
          double gd[8], td[8];
          static_for<8>([&](auto I) LCP_INL { gd[I] = (double)gv[I]; td[I] = (double)tv[I]; });
          gtw8_dpp<C0>(a0, a1, a2, a3, wn, wt, gd, td);
        } else
        static_for<4>([&](auto H) LCP_INL {
          constexpr int I = 2 * H, C = C0 + I;
          fmac_bc<C>(a0, wn, (TC)gv[I]);
          fmac_bc<C>(a1, wt, (TC)tv[I]);
          fmac_bc<C + 1>(a2, wn, (TC)gv[I + 1]);
          fmac_bc<C + 1>(a3, wt, (TC)tv[I + 1]);
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      out.v[HX] = (a0 + a1) + (a2 + a3);
    });
    return out;
  }
  __device__ __forceinline__ TC Av(const XV<TC, XH>& v) const {        // e-space <- x-space
    TC a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const TI* ar = L.AtL + (l16 & (EQ - 1)) * RS + lds_opaque_zero();   // row l16 of A (rows >= e are zero; lanes >= EQ unused)
    static_for<4 * XH>([&](auto H) LCP_INL {
      constexpr int K = 4 * H, hx = K >> 4;
      fmac_bc<K & 15>(a0, v.v[hx], (TC)ar[K]); fmac_bc<(K + 1) & 15>(a1, v.v[hx], (TC)ar[K + 1]);
      fmac_bc<(K + 2) & 15>(a2, v.v[hx], (TC)ar[K + 2]); fmac_bc<(K + 3) & 15>(a3, v.v[hx], (TC)ar[K + 3]);
    });
    return (l16 < EQ) ? (a0 + a1) + (a2 + a3) : (TC)0;
  }
  __device__ __forceinline__ XV<TC, XH> Aty(TC y) const {       // x-space <- e-space
    XV<TC, XH> out;
    const int oz = lds_opaque_zero();
    static_for<XH>([&](auto HX) LCP_INL {
      TC a0 = 0, a1 = 0;
      const TI* at = L.AtL + 16 * HX + l16 + oz;
      fmac_bc<0>(a0, y, (TC)at[0]); fmac_bc<1>(a1, y, (TC)at[RS]);
      fmac_bc<2>(a0, y, (TC)at[2 * RS]); fmac_bc<3>(a1, y, (TC)at[3 * RS]);
      out.v[HX] = a0 + a1;
    });
    return out;
  }
  __device__ __forceinline__ void GAt(TC t, TC& gn, TC& gt) const {     // m-space <- e-space
    gn = 0; gt = 0;
    static_for<EQ>([&](auto A) LCP_INL { fmac_bc<A>(gn, t, gan[A]); fmac_bc<A>(gt, t, gat[A]); });
  }
  __device__ __forceinline__ TC GAtw(TC wn, TC wt) const {              // e-space <- m-space
    TC out = 0;
    static_for<EQ>([&](auto A) LCP_INL { const TC sm = row_sum(gan[A] * wn + gat[A] * wt); if (l16 == A) out = sm; });
    return out;
  }
  __device__ __forceinline__ TC S11v(TC v) const {
    TC a0 = 0, a1 = 0;
    fmac_bc<0>(a0, v, s11row[0]); fmac_bc<1>(a1, v, s11row[1]);
    fmac_bc<2>(a0, v, s11row[2]); fmac_bc<3>(a1, v, s11row[3]);
    return a0 + a1;
  }
};

// ---------------------------------------------------------------- reduced system in registers
// Lane c holds row a_c (index c) in ta[] and row u_c (index 16 + c) in tu[]; columns 0..15 <-> a, 16..31 <-> u.
template <typename TC>
struct RedQ {
  TC Sp, Sm, Dg, idet, wa, wu;        // per factorisation (functions of D = s/z), see lcp_wave64.hip `Red`
  TC ua, uu;                          // 1 / U[c][c], 1 / U[16+c][16+c]
};


template <typename TI, typename TC, bool LDSW, int XH>
__device__ __forceinline__ bool factor_q(TC (&ta)[32], TC (&tu)[32], RedQ<TC>& R, const SceneQ<TI, TC, XH>& S, const TC* W2q,
                                         const M4<TC>& D, bool valid LCP_QPROF_ARG) {
  const int l16 = launder(S.l16), nc = __builtin_amdgcn_readfirstlane(S.ncw);     // (keeps the step guards scalar branches)
  R.Dg = D.g;
  R.Sp = (TC)0.5 * (D.f1 + D.f2); R.Sm = (TC)0.5 * (D.f1 - D.f2);
  R.idet = fast_rcp(R.Sp * R.Dg + (TC)2);
  R.wa = (TC)2 * S.mu * R.idet; R.wu = -R.Dg * R.Sm * R.idet;
  const TC addA = valid ? D.n : (TC)1;                                           // on column c of row a_c
  const TC addB = valid ? (TC)0.5 * R.Sm * R.wa : (TC)0;                         // on column c of row u_c
  const TC addU = valid ? (TC)0.5 * (R.Sp + R.Sm * R.wu) : (TC)1;                // on column 16 + c of row u_c
  if (LDSW) {
    // symmetric W from LDS: row r of lane (r = l16 for a, 16 + l16 for u), column q: (r, q) if q >= r else (q, r)
    const TC* wl = S.L.WL + lds_opaque_zero();
    const int ra = wl_row(l16), ru = wl_row(16 + l16);
    static_for<32>([&](auto Qc) LCP_INL {
      constexpr int q = Qc;
      if (q >= 16) ta[q] = wl[ra + q];
      else ta[q] = wl[(q >= l16) ? (ra + q) : (wl_row(q) + l16)];
      if (q < 16) tu[q] = wl[wl_row(q) + 16 + l16];
      else tu[q] = wl[(q >= 16 + l16) ? (ru + q) : (wl_row(q) + 16 + l16)];
    });
  } else {
    static_for<16>([&](auto P) LCP_INL {
      constexpr int q = 2 * P;
      load2(W2q + (((size_t)P * 2 + 0) * 16 + l16) * 2, ta[q], ta[q + 1]);
      load2(W2q + (((size_t)P * 2 + 1) * 16 + l16) * 2, tu[q], tu[q + 1]);
    });
  }
  static_for<16>([&](auto Q) LCP_INL {
    ta[Q] += (l16 == Q) ? addA : (TC)0;
    tu[Q] += (l16 == Q) ? addB : (TC)0;
    tu[16 + Q] += (l16 == Q) ? addU : (TC)0;
  });
  bool singular = false;
  R.ua = 1; R.uu = 1;
  LCP_QTICK(pr, 1)                                                               // W load + diagonal
  // Right-looking LU without pivoting.  The wave is alone on its SIMD (register budget), so nothing hides the
  // latency of the pivot chain (row broadcast -> v_rcp_f64 -> Newton steps): each step therefore updates the NEXT pivot
  // column first and launches that pivot's reciprocal before it sweeps the remaining columns.
  TC pivv = bc<0>(ta[0]);
  TC inv = fast_rcp(pivv);
  static_for<16>([&](auto K) LCP_INL {                                           // pivots a_0 .. a_15
    constexpr int k = K;
    if (k < nc) {
      singular = singular || (pivv == (TC)0);
      const TC la = (l16 > k) ? ta[k] * inv : (TC)0;
      const TC lu = tu[k] * inv;
      ta[k] = (l16 > k) ? la : ta[k];
      tu[k] = lu;
      R.ua = (l16 == k) ? inv : R.ua;
      fnmac_bc<k>(tu[k + 1], ta[k + 1], lu);                     // (reads row k's ta[j] before it is updated below)
      fnmac_bc<k>(ta[k + 1], ta[k + 1], la);
      if constexpr (k + 1 < 16) pivv = bc<(k + 1) & 15>(ta[k + 1]); else pivv = bc<0>(tu[16]);
      inv = fast_rcp(pivv);
      if (LCP_Q_ASM_LU) lu_cols_a<k, k + 2, 30 - k>(ta, tu, la, lu);
      else static_for<30 - k>([&](auto JJ) LCP_INL {
        constexpr int j = k + 2 + JJ;
        fnmac_bc<k>(tu[j], ta[j], lu);
        fnmac_bc<k>(ta[j], ta[j], la);
      });
    }
  });
  if (nc < 16) { pivv = bc<0>(tu[16]); inv = fast_rcp(pivv); }                   // (the a sweep stopped early)
  static_for<16>([&](auto K) LCP_INL {                                           // pivots u_0 .. u_15
    constexpr int kk = K, k = 16 + K;
    if (kk < nc) {
      singular = singular || (pivv == (TC)0);
      const TC lu = (l16 > kk) ? tu[k] * inv : (TC)0;
      tu[k] = (l16 > kk) ? lu : tu[k];
      R.uu = (l16 == kk) ? inv : R.uu;
      if constexpr (kk < 15) {
        fnmac_bc<kk>(tu[k + 1], tu[k + 1], lu);
        pivv = bc<(kk + 1) & 15>(tu[k + 1]);
        inv = fast_rcp(pivv);
        if (LCP_Q_ASM_LU) lu_cols_u<kk, k + 2, 14 - kk>(tu, lu);
        else static_for<14 - kk>([&](auto JJ) LCP_INL {
          constexpr int j = k + 2 + JJ;
          fnmac_bc<kk>(tu[j], tu[j], lu);
        });
      }
    }
  });
  LCP_QTICK(pr, 2)                                                               // LU
  return singular;
}

// T^-1 hz through the reduced system; everything except the two triangular sweeps is lane-local.
template <typename TI, typename TC, int XH>
__device__ __forceinline__ M4<TC> tsolve_q(const TC (&ta)[32], const TC (&tu)[32], const RedQ<TC>& R, const SceneQ<TI, TC, XH>& S,
                                           const M4<TC>& hz) {
  const int l16 = S.l16, nc = __builtin_amdgcn_readfirstlane(S.ncw);
  const TC r12 = hz.f1 + hz.f2;
  const TC w0 = (R.Dg * r12 - (TC)2 * hz.g) * R.idet;
  TC ra = hz.n, ru = (TC)0.5 * (hz.f1 - hz.f2) - (TC)0.5 * R.Sm * w0;
  // The sweeps are guarded per group of LCP_Q_TS_GROUP steps: rows / columns at or beyond the contact count are identity
  // (never touched by the factorisation), so the surplus steps of the last group are exact no-ops, the scalar branches
  // between the steps go, and the compiler schedules across the steps of a group (measured: 0.2105 -> 0.1947 ms).
  constexpr int TG = LCP_Q_TS_GROUP, NG = 16 / TG;
  static_for<NG>([&](auto Gq) LCP_INL {                    // L y = rhs
    if (TG * Gq < nc) static_for<TG>([&](auto Kq) LCP_INL {
      constexpr int k = TG * Gq + Kq;
      fnmac_bc<k>(ru, ra, tu[k]);
      fnmac_bc<k>(ra, ra, keep_if(ta[k], l16 > k));
    });
  });
  static_for<NG>([&](auto Gq) LCP_INL {
    if (TG * Gq < nc) static_for<TG>([&](auto Kq) LCP_INL {
      constexpr int kk = TG * Gq + Kq, k = 16 + kk;
      fnmac_bc<kk>(ru, ru, keep_if(tu[k], l16 > kk));
    });
  });
  static_for<NG>([&](auto GR) LCP_INL {                    // U x = y
    constexpr int Gq = NG - 1 - GR;
    if (TG * Gq < nc) static_for<TG>([&](auto KR) LCP_INL {
      constexpr int kk = TG * Gq + TG - 1 - KR, k = 16 + kk;
      const TC xs = ru * R.uu;                             // x_kk lives in lane kk of xs
      fnmac_bc<kk>(ra, xs, ta[k]);
      fnmac_bc<kk>(ru, xs, keep_if(tu[k], l16 < kk));
    });
  });
  static_for<NG>([&](auto GR) LCP_INL {
    constexpr int Gq = NG - 1 - GR;
    if (TG * Gq < nc) static_for<TG>([&](auto KR) LCP_INL {
      constexpr int k = TG * Gq + TG - 1 - KR;
      const TC xs = ra * R.ua;
      fnmac_bc<k>(ra, xs, keep_if(ta[k], l16 < k));
    });
  });
  const TC a = ra * R.ua, u = ru * R.uu;
  const TC w = w0 + R.wa * a + R.wu * u;
  M4<TC> dz;
  dz.n = a; dz.f1 = (TC)0.5 * (w + u); dz.f2 = (TC)0.5 * (w - u);
  dz.g = (r12 - R.Sm * u + R.Sp * (hz.g - S.mu * a)) * R.idet;
  return dz;
}

// solve_kkt (pdipm.py:325-354).  rs, rz given per contact (M4), rx in x-space, ry in e-space.
template <typename TI, typename TC, int XH>
__device__ __forceinline__ void solve_kkt_q(const SceneQ<TI, TC, XH>& S, const TC (&ta)[32], const TC (&tu)[32], const RedQ<TC>& R,
                                            const M4<TC>& di, bool valid, const XV<TC, XH>& rx, const M4<TC>& rs, const M4<TC>& rz, TC ry,
                                            XV<TC, XH>& ox, M4<TC>& os, M4<TC>& oz, TC& oy, bool rxy_zero LCP_QPROF_ARG) {
  // `rxy_zero` (wave-uniform): rx = 0 and ry = 0, the corrector solve (pdipm.py:152-158).  Then Q^-1 rx, G Q^-1 rx and
  // A Q^-1 rx - ry are exact zeros and the products that would form them are skipped: 10 of the 21 solves of a step.
  TC gn = 0, gt = 0, hy = 0;
  XV<TC, XH> v;
  if (!rxy_zero) {
    static_for<XH>([&](auto HX) LCP_INL { v.v[HX] = S.qid[HX] * rx.v[HX]; });    // :333 (diagonal Q)
    S.Gv(v, gn, gt);
  }
  // `di` = 1 / d (already formed for T = R + diag(1/d)): rs / d is taken as rs * di - one rounding more than the
  // reference's division, 8 fp64 divisions less per solve
  M4<TC> hz = m4<TC>(gn + rs.n * di.n - rz.n, gt + rs.f1 * di.f1 - rz.f1, -gt + rs.f2 * di.f2 - rz.f2, rs.g * di.g - rz.g);   // :334-340
  if (S.e > 0 && !rxy_zero) {
    hy = S.Av(v) - ry;
    TC an, at;
    S.GAt(S.S11v(hy), an, at);
    hz.n -= an; hz.f1 -= at; hz.f2 += at;
  }
  if (!valid) hz = m4<TC>(0, 0, 0, 0);
  LCP_QTICK(pr, 3)                                                         // solve_kkt: products before
  const M4<TC> wz = tsolve_q<TI, TC, XH>(ta, tu, R, S, hz);
  LCP_QTICK(pr, 4)                                                         // triangular sweeps
  TC dy = 0;
  if (S.e > 0) dy = -S.S11v(hy - S.GAtw(valid ? wz.n : (TC)0, valid ? wz.f1 - wz.f2 : (TC)0));    // dy = -wy
  oz = m4<TC>(-wz.n, -wz.f1, -wz.f2, -wz.g);                               // :342
  if (!valid) oz = m4<TC>(0, 0, 0, 0);
  os = m4<TC>((-rs.n - oz.n) * di.n, (-rs.f1 - oz.f1) * di.f1, (-rs.f2 - oz.f2) * di.f2, (-rs.g - oz.g) * di.g);   // :347,350
  if (!valid) os = m4<TC>(0, 0, 0, 0);
  oy = dy;
  const XV<TC, XH> gw = S.Gtw(oz.n, oz.f1 - oz.f2);                        // :344-346
  XV<TC, XH> g1;
  static_for<XH>([&](auto HX) LCP_INL { g1.v[HX] = -rx.v[HX] - gw.v[HX]; });
  if (S.e > 0) { const XV<TC, XH> ay = S.Aty(dy); static_for<XH>([&](auto HX) LCP_INL { g1.v[HX] -= ay.v[HX]; }); }
  static_for<XH>([&](auto HX) LCP_INL { ox.v[HX] = S.qid[HX] * g1.v[HX]; });    // :349
  LCP_QTICK(pr, 5)                                                         // solve_kkt: products after
}

// ---------------------------------------------------------------- body-space variant of factor / solve (ALG = 1, nz <= 16)
// The same Newton step with the inequality block eliminated first (lcp_primal.hip has the algebra): per contact the 4 x 4 block
// M = F_c + diag(s / z) is inverted in closed form in the contact's lane, and what is factored is
//     K = [[Q + G^T M^-1 G, A^T], [A, 0]]      (nz + neq <= 20 rows instead of the 2 nc = 32 of the reduced contact-space system)
// Lane i holds x-row i in xr[] and (lanes < neq) equality row i in er[]; columns 0..15 <-> x, 16..19 <-> y.  G^T M^-1 G is
// accumulated contact by contact, the contact's 2 x 16 block P = B [jc; jt] broadcast from its lane with row_newbcast.
// PINC: the kernel is the pinned variant ALONE (ALG = 2; `pin_rt` then only says whether the wave qualified)
template <typename TC, bool PINC = false>
struct PrimQ {
  TC idn, i1, i2, kap;               // 1 / Dn, 1 / D1, 1 / D2, 1 / (Dg + 1 / D1 + 1 / D2) of this lane's contact
  TC udx, ude;                       // 1 / U[i][i] of the lane's x-row and equality row
  bool pin_rt;                       // (wave-uniform) the equality rows pin the first neq coordinates: A = [I 0], b = 0
  __device__ __forceinline__ bool pin() const { return PINC || pin_rt; }
  TC sp[EQ];                         // pinned variant: the lane's entries in the pinned columns, S[i][a] (taken out of xr[], where
                                     // zeros make the sweep steps of the pinned coordinates exact no-ops)
};
// PINNED COORDINATES.  Every world of the reference's demos fixes its floor with a TotalConstraint on body 0 (constraints.py:175-192,
// A = [I_3 0]).  Then A dx = -ry IS dx_a = -ry_a for a < neq, the rows of K that belong to the other coordinates close over
// themselves (S_ff dx_f = rhs_f - S_fp dx_p) and dy_a = rhs_a - (S dx)_a is read off row a afterwards: the LU runs over
// nz - neq rows of ONE row per lane (66 row updates instead of 360 on the headline config), the rows of the pinned coordinates
// are never touched by it and stay S.  Same equations, fewer of them; detected per wave (all four scenes), any other A takes
// the general path.
template <typename TI, typename TC, typename PQ, typename SQ>
__device__ __forceinline__ M4<TC> minv_pq(const PQ& R, const SQ& S, const M4<TC>& t) {   // M^-1 t
  M4<TC> o;
  o.n = R.idn * t.n;
  o.g = R.kap * ((t.g - S.mu * o.n) + fma(R.i1, t.f1, R.i2 * t.f2));
  o.f1 = R.i1 * (t.f1 - o.g);
  o.f2 = R.i2 * (t.f2 - o.g);
  return o;
}
template <typename TI, typename TC, typename PQ, typename SQ>
__device__ __forceinline__ bool factor_pq(TC (&xr)[20], TC (&er)[20], PQ& R, const SQ& S, const M4<TC>& D,
                                          bool valid LCP_QPROF_ARG) {
  const int l16 = launder(S.l16), ncw = __builtin_amdgcn_readfirstlane(S.ncw);
  const int nz = __builtin_amdgcn_readfirstlane(S.nz), e = __builtin_amdgcn_readfirstlane(S.e);
  R.idn = fast_rcp(D.n); R.i1 = fast_rcp(D.f1); R.i2 = fast_rcp(D.f2);
  R.kap = fast_rcp(D.g + (R.i1 + R.i2));
  const TC b00 = valid ? R.idn : (TC)0;
  const TC b10 = valid ? R.kap * (R.i1 - R.i2) * (S.mu * R.idn) : (TC)0;
  const TC b11 = valid ? R.kap * fma(R.i1 + R.i2, D.g, (TC)4 * (R.i1 * R.i2)) : (TC)0;   // = (i1 + i2) - kap (i1 - i2)^2, no cancellation
  TC p0[16], p1[16];
  const int oz = lds_opaque_zero();
  static_for<16>([&](auto J) LCP_INL { const TC c = (TC)S.template jcv<J>(oz), t = (TC)S.template jtv<J>(oz); p0[J] = b00 * c; p1[J] = fma(b10, c, b11 * t); });
  {
    const TI* at = S.L.AtL + oz;
    const TC dsel = (l16 < nz) ? S.qd[0] : (TC)1;
    static_for<16>([&](auto J) LCP_INL { xr[J] = (l16 == J) ? dsel : (TC)0; });
    if (!R.pin()) {                                                               // (the pinned variant has no equality rows / columns)
      static_for<16>([&](auto J) LCP_INL { er[J] = (l16 < EQ) ? (TC)at[(l16 & (EQ - 1)) * 16 + J] : (TC)0; });
      static_for<EQ>([&](auto A) LCP_INL {
        xr[16 + A] = (TC)at[A * 16 + l16];
        er[16 + A] = (l16 == A && A >= e) ? (TC)1 : (TC)0;
      });
    }
  }
  {
    const TI* gl = S.L.GL + l16 + oz;
    const TI* gtl = S.L.GTL + l16 + oz;
    static_for<2>([&](auto Hh) LCP_INL {
      constexpr int C0 = 8 * Hh;
      if (C0 < ncw) {
        TI gv[8], tv[8];
        static_for<8>([&](auto I) LCP_INL { gv[I] = gl[(C0 + I) * 16]; tv[I] = gtl[(C0 + I) * 16]; });
        __builtin_amdgcn_sched_barrier(0);
        static_for<8>([&](auto I) LCP_INL {
          constexpr int C = C0 + I;
          const TC a = (TC)gv[I], b = (TC)tv[I];
          pq_form8<C, 0>(xr, p0, p1, a, b);
          pq_form8<C, 8>(xr, p0, p1, a, b);
        });
      }
    });
  }
  LCP_QTICK(pr, 1)                                                               // formation
  bool singular = false;
  R.udx = 1; R.ude = 1;
  if (R.pin()) {
    // x pivots e .. nz-1 only, one row per lane, columns k+1 .. 15
    TC pivv = bc<0>(xr[0]);                                                      // (placeholder until the first live step)
    TC inv = (TC)1;
    bool primed = false;
    static_for<16>([&](auto K) LCP_INL {
      constexpr int k = K;
      if (k >= e && k < nz) {
        if (!primed) { pivv = bc<k>(xr[k]); inv = fast_rcp(pivv); primed = true; }
        singular = singular || (pivv == (TC)0);
        // column k scaled by 1 / U[k][k] on every row but k: the multipliers below the diagonal, and above it U[i][k] / U[k][k] - the
        // backward sweep then needs no multiplication on its dependent chain (w_i -= (U_ik / U_kk) w_k, x_k = w_k / U_kk at the end)
        const TC sk = xr[k] * inv;
        const TC lx = keep_if(sk, l16 > k);                                            // (numerically zero on the rows above: one v_cndmask)
        xr[k] = (l16 == k) ? xr[k] : sk;
        R.udx = (l16 == k) ? inv : R.udx;
        if constexpr (k + 1 < 16) {
          fnmac_bc<k>(xr[k + 1], xr[k + 1], lx);
          pivv = bc<(k + 1) & 15>(xr[k + 1]); inv = fast_rcp(pivv);
          lu_cols_x<k, k + 2, 14 - k>(xr, lx);
        }
      }
    });
    static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; R.sp[a] = xr[a]; xr[a] = (a < e) ? (TC)0 : xr[a]; });
    LCP_QTICK(pr, 2)                                                             // LU
    return singular;
  }
  // (as in factor_q: each step updates the NEXT pivot column first and launches that pivot's reciprocal before the block of
  //  the remaining columns - the wave is alone on its SIMD and nothing else hides the broadcast -> v_rcp_f64 -> Newton chain)
  TC pivv = bc<0>(xr[0]);
  TC inv = fast_rcp(pivv);
  static_for<16>([&](auto K) LCP_INL {                                           // x pivots
    constexpr int k = K;
    if (k < nz) {
      singular = singular || (pivv == (TC)0);
      const TC lx = (l16 > k) ? xr[k] * inv : (TC)0;
      const TC le = er[k] * inv;
      xr[k] = (l16 > k) ? lx : xr[k];
      er[k] = le;
      R.udx = (l16 == k) ? inv : R.udx;
      fnmac_bc<k>(er[k + 1], xr[k + 1], le);                                     // (reads row k's xr[j] before it is updated below)
      fnmac_bc<k>(xr[k + 1], xr[k + 1], lx);
      if constexpr (k + 1 < 16) { pivv = bc<(k + 1) & 15>(xr[k + 1]); inv = fast_rcp(pivv); }
      lu_cols_p<k, k + 2, 18 - k>(xr, er, lx, le);
    }
  });
  static_for<EQ>([&](auto A_) LCP_INL {                                          // equality pivots
    constexpr int a = A_, k = 16 + A_;
    if (a < e) {
      const TC pivv = bc<a>(er[k]);
      singular = singular || (pivv == (TC)0);
      const TC inv = fast_rcp(pivv);
      const TC le = (l16 > a) ? er[k] * inv : (TC)0;
      er[k] = (l16 > a) ? le : er[k];
      R.ude = (l16 == a) ? inv : R.ude;
      static_for<EQ - 1 - a>([&](auto JJ) LCP_INL { constexpr int j = k + 1 + JJ; fnmac_bc<a>(er[j], er[j], le); });
    }
  });
  LCP_QTICK(pr, 2)                                                               // LU
  return singular;
}

// solve_kkt (pdipm.py:325-354) in body space:  q = rs / d - rz,  K [dx; dy] = [-rx + G^T M^-1 q; -ry],
// dz = M^-1 (G dx - q),  ds = (-rs - dz) / d
template <typename TI, typename TC, typename PQ, typename SQ>
__device__ __forceinline__ void solve_kkt_pq(const SQ& S, const TC (&xr)[20], const TC (&er)[20], const PQ& R,
                                             const M4<TC>& di, bool valid, const XV<TC, 1>& rx, const M4<TC>& rs, const M4<TC>& rz, TC ry,
                                             XV<TC, 1>& ox, M4<TC>& os, M4<TC>& oz, TC& oy LCP_QPROF_ARG) {
  const int l16 = S.l16, nz = __builtin_amdgcn_readfirstlane(S.nz), e = __builtin_amdgcn_readfirstlane(S.e);
  M4<TC> q = m4<TC>(rs.n * di.n - rz.n, rs.f1 * di.f1 - rz.f1, rs.f2 * di.f2 - rz.f2, rs.g * di.g - rz.g);
  if (!valid) q = m4<TC>(0, 0, 0, 0);
  const M4<TC> u = minv_pq<TI, TC, PQ, SQ>(R, S, q);
  const XV<TC, 1> gu = S.template Gtw<true>(valid ? u.n : (TC)0, valid ? u.f1 - u.f2 : (TC)0);
  TC wx = (l16 < nz) ? gu.v[0] - rx.v[0] : (TC)0;
  TC we = (l16 < e) ? -ry : (TC)0;
  LCP_QTICK(pr, 3)                                                         // solve_kkt: products before
  if (R.pin()) {
    // dx_p = -ry on the pinned lanes; the free rows solve S_ff dx_f = rhs_f - S_fp dx_p; the pinned lanes, whose rows are still
    // S, ride along in the backward sweep and end up with rhs_a - (S dx)_a = dy_a
    if (e > 0) static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; fnmac_bc<a>(wx, we, keep_if(R.sp[a], a < e)); });
    static_for<4>([&](auto Gq) LCP_INL {                                   // L y = rhs (the steps of the pinned columns meet zeros)
      if (4 * Gq < nz) static_for<4>([&](auto Kq) LCP_INL {
        constexpr int k = 4 * Gq + Kq;
        fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 > k));
      });
    });
    static_for<4>([&](auto GR) LCP_INL {                                   // U x = y
      constexpr int Gq = 3 - GR;
      if (4 * Gq < nz) static_for<4>([&](auto KR) LCP_INL {
        constexpr int k = 4 * Gq + 3 - KR;
        fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 < k));                        // (xr[k] = U[i][k] / U[k][k] above the diagonal)
      });
    });
    LCP_QTICK(pr, 4)                                                       // triangular sweeps
    ox.v[0] = (l16 < e) ? we : ((l16 < nz) ? wx * R.udx : (TC)0);
    oy = (l16 < e) ? wx : (TC)0;
    TC gn, gt;
    S.template Gv<true>(ox, gn, gt);
    oz = minv_pq<TI, TC, PQ, SQ>(R, S, m4<TC>(gn - q.n, gt - q.f1, -gt - q.f2, -q.g));
    if (!valid) oz = m4<TC>(0, 0, 0, 0);
    os = m4<TC>((-rs.n - oz.n) * di.n, (-rs.f1 - oz.f1) * di.f1, (-rs.f2 - oz.f2) * di.f2, (-rs.g - oz.g) * di.g);   // :347,350
    if (!valid) os = m4<TC>(0, 0, 0, 0);
    LCP_QTICK(pr, 5)                                                       // solve_kkt: products after
    return;
  }
  static_for<4>([&](auto Gq) LCP_INL {                                     // L y = rhs
    if (4 * Gq < nz) static_for<4>([&](auto Kq) LCP_INL {
      constexpr int k = 4 * Gq + Kq;
      fnmac_bc<k>(we, wx, er[k]);
      fnmac_bc<k>(wx, wx, keep_if(xr[k], l16 > k));
    });
  });
  if (e > 0) {
    static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; fnmac_bc<a>(we, we, keep_if(er[16 + a], l16 > a)); });
    static_for<EQ>([&](auto AR) LCP_INL {                                  // U x = y
      constexpr int a = EQ - 1 - AR;
      const TC xs = we * R.ude;
      fnmac_bc<a>(wx, xs, xr[16 + a]);
      fnmac_bc<a>(we, xs, keep_if(er[16 + a], l16 < a));
    });
  }
  static_for<4>([&](auto GR) LCP_INL {
    constexpr int Gq = 3 - GR;
    if (4 * Gq < nz) static_for<4>([&](auto KR) LCP_INL {
      constexpr int k = 4 * Gq + 3 - KR;
      const TC xs = wx * R.udx;
      fnmac_bc<k>(wx, xs, keep_if(xr[k], l16 < k));
    });
  });
  LCP_QTICK(pr, 4)                                                         // triangular sweeps
  ox.v[0] = (l16 < nz) ? wx * R.udx : (TC)0;
  oy = (l16 < e) ? we * R.ude : (TC)0;
  TC gn, gt;
  S.template Gv<true>(ox, gn, gt);
  oz = minv_pq<TI, TC, PQ, SQ>(R, S, m4<TC>(gn - q.n, gt - q.f1, -gt - q.f2, -q.g));
  if (!valid) oz = m4<TC>(0, 0, 0, 0);
  os = m4<TC>((-rs.n - oz.n) * di.n, (-rs.f1 - oz.f1) * di.f1, (-rs.f2 - oz.f2) * di.f2, (-rs.g - oz.g) * di.g);   // :347,350
  if (!valid) os = m4<TC>(0, 0, 0, 0);
  LCP_QTICK(pr, 5)                                                         // solve_kkt: products after
}

// get_step for (z,dz),(s,ds) of one scene (pdipm.py:182-186): min(step(z,dz), step(s,ds)), NaN semantics kept
template <typename TC>
__device__ __forceinline__ TC step_pair_q(const M4<TC>& z, const M4<TC>& dz, const M4<TC>& s, const M4<TC>& ds, bool valid) {
  const TC ninf = -inf_of<TC>(), pinf = inf_of<TC>();
  const M4<TC> az = m4<TC>(qdiv_x(-z.n, dz.n), qdiv_x(-z.f1, dz.f1), qdiv_x(-z.f2, dz.f2), qdiv_x(-z.g, dz.g));
  const M4<TC> as = m4<TC>(qdiv_x(-s.n, ds.n), qdiv_x(-s.f1, ds.f1), qdiv_x(-s.f2, ds.f2), qdiv_x(-s.g, ds.g));
  auto key4 = [&](const M4<TC>& a) { return umax(umax(nan_key(a.n), nan_key(a.f1)), umax(nan_key(a.f2), nan_key(a.g))); };
  // a.max() (NaN if any entry is NaN), then max(1.0, .) which maps NaN to 1.0
  TC mz = valid ? fmax_(fmax_(az.n, az.f1), fmax_(az.f2, az.g)) : ninf;
  TC ms = valid ? fmax_(fmax_(as.n, as.f1), fmax_(as.f2, as.g)) : ninf;
  const uint32_t kmz = row_umax(valid ? key4(az) : 0u), kms = row_umax(valid ? key4(as) : 0u);
  mz = row_fmax(mz); ms = row_fmax(ms);
  const TC fz = key_is_nan<TC>(kmz) ? (TC)1 : fmax_(mz, (TC)1), fs = key_is_nan<TC>(kms) ? (TC)1 : fmax_(ms, (TC)1);
  // a[dv > 0] = fill ; a.min() (NaN if any remaining entry is NaN) ; the two vectors are merged before the row reduction
  auto pick = [&](TC dv, TC a, TC fill) { return (dv > (TC)0) ? fill : a; };
  const M4<TC> pz = m4<TC>(pick(dz.n, az.n, fz), pick(dz.f1, az.f1, fz), pick(dz.f2, az.f2, fz), pick(dz.g, az.g, fz));
  const M4<TC> ps = m4<TC>(pick(ds.n, as.n, fs), pick(ds.f1, as.f1, fs), pick(ds.f2, as.f2, fs), pick(ds.g, as.g, fs));
  TC l = valid ? fmin_(fmin_(fmin_(pz.n, pz.f1), fmin_(pz.f2, pz.g)), fmin_(fmin_(ps.n, ps.f1), fmin_(ps.f2, ps.g))) : pinf;
  const uint32_t kl = row_umax(valid ? umax(key4(pz), key4(ps)) : 0u);
  l = row_fmin(l);
  return key_is_nan<TC>(kl) ? nan_of<TC>() : l;
}

// workspace view of the quad path: W (8 KB) in the R2 region, the rest in the w64 fields
//   W2q[((p*2 + slot)*16 + lane)*2 + (q&1)], p = q>>1, slot 0 = row a_lane, 1 = row u_lane
template <typename TI, typename TC, int XH>
__device__ __forceinline__ void store_scene_ws(const Ws<TI, TC>& W, const SceneQ<TI, TC, XH>& S) {
  const int l16 = S.l16;
  static_for<EQ>([&](auto A) LCP_INL { W.GAc[(l16 * 2 + 0) * EQ + A] = S.gan[A]; W.GAc[(l16 * 2 + 1) * EQ + A] = S.gat[A]; });
  static_for<EQ>([&](auto C) LCP_INL { if (l16 < EQ) W.S11i[l16 * EQ + C] = S.s11row[C]; });
  static_for<XH>([&](auto HX) LCP_INL { W.Qit[16 * HX + l16] = S.qid[HX]; });
}
// where the x-space part of the best iterate lives in the workspace: the 16-entry x field, or (nz > 16) a free stretch
// of the Q^-1 field, which this path only uses for the diagonal
template <int XH, typename TI, typename TC> __device__ __forceinline__ TC* ws_x(const Ws<TI, TC>& W) { return XH == 1 ? W.x : W.Qit + 64; }

// ---------------------------------------------------------------- inputs
// dense contact-structured LCP -> per-lane rows (+ LDS copies for the transposed products)
template <typename TI, typename TC>
__device__ __forceinline__ void load_dense_q(SceneQ<TI, TC, 1>& S, const FwdArgs& P, const Ws<TI, TC>& W, int scene,
                                             XV<TC, 1>& p, TC& hn, TC& b) {
  const int nz = S.nz, nc = S.ncap, e = S.e, l16 = S.l16, m = 4 * nc;
  const bool vc = l16 < S.nc;
  const TI* Grow_n = (const TI*)P.G + ((size_t)scene * m + (vc ? l16 : 0)) * nz;
  const TI* Grow_t = (const TI*)P.G + ((size_t)scene * m + nc + 2 * (vc ? l16 : 0)) * nz;
  static_for<16>([&](auto J) LCP_INL {
    S.jc[J] = (vc && J < nz) ? Grow_n[J] : (TI)0;
    S.jt[J] = (vc && J < nz) ? Grow_t[J] : (TI)0;
    S.L.GL[l16 * 16 + J] = S.jc[J];
    S.L.GTL[l16 * 16 + J] = S.jt[J];
  });
  const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
  static_for<16>([&](auto K) LCP_INL {
    if (l16 < EQ) S.L.AtL[l16 * 16 + K] = (l16 < e && K < nz) ? A[l16 * nz + K] : (TI)0;
  });
  const TI q = (l16 < nz) ? ((const TI*)P.Q)[(size_t)scene * nz * nz + l16 * nz + l16] : (TI)1;
  S.qd[0] = (l16 < nz) ? (TC)q : (TC)0;
  S.qid[0] = (l16 < nz) ? (TC)1 / (TC)q : (TC)0;
  S.mu = vc ? W.meta[1 + l16] : (TC)0;
  p.v[0] = (l16 < nz) ? (TC)((const TI*)P.p)[(size_t)scene * nz + l16] : (TC)0;
  hn = vc ? (TC)((const TI*)P.h)[(size_t)scene * m + l16] : (TC)0;
  b = (l16 < e) ? (TC)((const TI*)P.b)[(size_t)scene * e + l16] : (TC)0;
}

// contact list -> per-lane rows (physics/engines.py:31-32,50-74; physics/world.py:144-234)
template <typename TI, typename TC, int XH, typename SQ>
__device__ __forceinline__ void assemble_q(SQ& S, const StepArgs& P, int scene, XV<TC, XH>& p, TC& hn, TC& b) {
  constexpr int RS = 16 * XH;
  const int nb = P.nb, nc = S.ncap, nz = S.nz, e = S.e, l16 = S.l16;
  const bool vc = l16 < S.nc;
  const TI* Md = (const TI*)P.Mdiag + (size_t)scene * nz;
  const TI* vv = (const TI*)P.v + (size_t)scene * nz;
  const TI* ff = (const TI*)P.f + (size_t)scene * nz;
  static_for<RS>([&](auto J) LCP_INL { S.L.GL[l16 * RS + J] = (TI)0; S.L.GTL[l16 * RS + J] = (TI)0; });
  TI hrow = (TI)0, mu = (TI)0;
  if (vc) {
    const ContactRows<TI> r = make_contact<TI>((const TI*)P.c_n + (size_t)scene * nc * 2, (const TI*)P.c_p1 + (size_t)scene * nc * 2,
                                               (const TI*)P.c_p2 + (size_t)scene * nc * 2, P.c_i1 + (size_t)scene * nc,
                                               P.c_i2 + (size_t)scene * nc, (const TI*)P.rest + (size_t)scene * nb,
                                               (const TI*)P.fric + (size_t)scene * nb, vv, l16);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int col = (q < 3) ? 3 * r.b1 + q : 3 * r.b2 + (q - 3);
      S.L.GL[l16 * RS + col] = r.jn[q];
      S.L.GTL[l16 * RS + col] = r.jf[q];
    }
    hrow = r.h; mu = r.mu;
  }
  if constexpr (sizeof(S.jc) == sizeof(TI) * RS)                                                        // (ROWL: the rows stay in LDS)
    static_for<RS>([&](auto J) LCP_INL { S.jc[J] = S.L.GL[l16 * RS + J]; S.jt[J] = S.L.GTL[l16 * RS + J]; });   // own row only
  const TI* Je = (const TI*)P.Je + (size_t)scene * e * nz;
  static_for<RS>([&](auto K) LCP_INL {
    if (l16 < EQ) S.L.AtL[l16 * RS + K] = (l16 < e && K < nz) ? Je[l16 * nz + K] : (TI)0;
  });
  static_for<XH>([&](auto HX) LCP_INL {
    const int j = 16 * HX + l16;
    const TI q = (j < nz) ? Md[j] : (TI)1;
    S.qd[HX] = (j < nz) ? (TC)q : (TC)0;
    S.qid[HX] = (j < nz) ? (TC)1 / (TC)q : (TC)0;
    p.v[HX] = (j < nz) ? (TC)momentum_entry<TI>(Md[j], vv[j], (TI)P.dt, ff[j]) : (TC)0;     // engines.py:32
  });
  S.mu = (TC)mu;
  hn = (TC)hrow;
  b = (TC)0;
}

// pre_factor_kkt for diagonal Q: G Q^-1 A^T rows, (A Q^-1 A^T)^-1, W = J P J^T into the workspace
template <typename TI, typename TC, int XH>
__device__ __forceinline__ int prefactor_q(SceneQ<TI, TC, XH>& S, const Ws<TI, TC>& W, bool live) {
  constexpr int RS = 16 * XH;
  const int nc = S.nc, e = S.e, l16 = S.l16, nz = S.nz;
  int status = 0;
  bool qbad = false;
  static_for<XH>([&](auto HX) LCP_INL { qbad = qbad || (16 * HX + l16 < nz && !(S.qd[HX] != (TC)0)); });
  if (row_any(qbad)) status |= LCP_ST_SINGULAR_Q;
  TC gqn[RS], gqt[RS];
  static_for<RS>([&](auto K) LCP_INL { const TC qi = bc<K & 15>(S.qid[K >> 4]); gqn[K] = (TC)S.jc[K] * qi; gqt[K] = (TC)S.jt[K] * qi; });
  TC ccn[EQ], cct[EQ];
  static_for<EQ>([&](auto A) LCP_INL { S.gan[A] = 0; S.gat[A] = 0; S.s11row[A] = 0; ccn[A] = 0; cct[A] = 0; });
  __syncthreads();                                   // LDS rows written by the loaders
  if (e > 0) {
    static_for<EQ>([&](auto A) LCP_INL {
      TC an = 0, at = 0;
      static_for<RS>([&](auto K) LCP_INL { const TC aak = (TC)S.L.AtL[A * RS + K]; an = fma(gqn[K], aak, an); at = fma(gqt[K], aak, at); });
      S.gan[A] = an; S.gat[A] = at;
      S.L.GAL[(l16 * 2 + 0) * EQ + A] = an; S.L.GAL[(l16 * 2 + 1) * EQ + A] = at;
    });
    {                                                 // S11 = A Q^-1 A^T, entry (a, c') in lane a*EQ + c'
      const int a = l16 >> 2, c = l16 & 3;
      TC acc = 0;
      static_for<RS>([&](auto K) LCP_INL { acc = fma((TC)S.L.AtL[a * RS + K] * bc<K & 15>(S.qid[K >> 4]), (TC)S.L.AtL[c * RS + K], acc); });
      S.L.S11[l16] = acc;
    }
    __syncthreads();
    if (l16 == 0) {                                   // tiny e x e Gauss-Jordan, one lane per scene
      TC* a = S.L.S11;
      bool bad = false;
      for (int k = 0; k < e; ++k) {
        const TC piv = a[k * EQ + k];
        bad = bad || !(piv != (TC)0) || (piv != piv);
        const TC pinv = (TC)1 / piv;
        for (int j = 0; j < e; ++j) if (j != k) a[k * EQ + j] *= pinv;
        for (int i = 0; i < e; ++i) if (i != k) { const TC f = a[i * EQ + k]; for (int j = 0; j < e; ++j) if (j != k) a[i * EQ + j] -= f * a[k * EQ + j]; a[i * EQ + k] = -f * pinv; }
        a[k * EQ + k] = pinv;
      }
      for (int i = 0; i < EQ; ++i) for (int j = 0; j < EQ; ++j) if (i >= e || j >= e) a[i * EQ + j] = 0;
      a[0] = bad ? nan_of<TC>() : a[0];
    }
    __syncthreads();
    if (S.L.S11[0] != S.L.S11[0]) status |= LCP_ST_SINGULAR_S11;
    static_for<EQ>([&](auto C) LCP_INL { S.s11row[C] = (l16 < EQ) ? S.L.S11[l16 * EQ + C] : (TC)0; });
    static_for<EQ>([&](auto A) LCP_INL {
      TC an = 0, at = 0;
      static_for<EQ>([&](auto C) LCP_INL { const TC sca = S.L.S11[C * EQ + A]; an = fma(S.gan[C], sca, an); at = fma(S.gat[C], sca, at); });
      ccn[A] = an; cct[A] = at;
    });
  }
  const bool vr = l16 < nc;
#pragma unroll 1
  for (int q0 = 0; q0 < 32; q0 += 2) {
    TC va[2], vu[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int q = q0 + h2, cq = q & 15, kind = q >> 4;
      const TI* jrow = (kind ? S.L.GTL : S.L.GL) + cq * RS;
      const TC* garow = S.L.GAL + (cq * 2 + kind) * EQ;
      TC a = 0, u = 0;
      static_for<RS>([&](auto K) LCP_INL { const TC jk = (TC)jrow[K]; a = fma(gqn[K], jk, a); u = fma(gqt[K], jk, u); });
      if (e > 0) static_for<EQ>([&](auto A) LCP_INL { const TC g = garow[A]; a = fma(-ccn[A], g, a); u = fma(-cct[A], g, u); });
      const bool ok = vr && (cq < nc);
      va[h2] = ok ? a : (TC)0; vu[h2] = ok ? u : (TC)0;
    }
    if (live) {
      store2(W.R2 + (((size_t)(q0 >> 1) * 2 + 0) * 16 + l16) * 2, va[0], va[1]);
      store2(W.R2 + (((size_t)(q0 >> 1) * 2 + 1) * 16 + l16) * 2, vu[0], vu[1]);
    }
    if (S.L.WL) {                                       // upper triangle -> LDS (W is symmetric up to rounding)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int q = q0 + h2;
        if (q >= l16) S.L.WL[wl_row(l16) + q] = va[h2];
        if (q >= 16 + l16) S.L.WL[wl_row(16 + l16) + q] = vu[h2];
      }
    }
  }
  if (live) store_scene_ws<TI, TC, XH>(W, S);
  __threadfence_block();
  __syncthreads();
  return status;
}

// ---------------------------------------------------------------- forward kernel
// `accept`: value of the classification flag (meta[0]) this kernel serves for dense inputs.
// XH: x-space halves (1: nz <= 16; 2: nz <= 32, fused inputs only)
// ALG: 0 = the reduced contact-space system (32 x 32), 1 = the body-space system (nz + neq <= 20 rows; XH = 1 only),
//      2 = the body-space system for waves whose equality rows pin the leading coordinates, ALONE in the kernel (contact-list
//          inputs): without the general path's equality rows the kernel needs 47 instead of 95 accumulation registers and runs
//          6 % faster.  A wave that does not qualify marks its scenes (meta[21]) and leaves; the ALG = 1 kernel, launched right
//          behind with accept = 3, serves exactly the marked scenes - on the usual worlds it finds none and is gone in 2-3 us.
template <typename TI, typename TC, bool FUSED, int XH, int ALG = 0>
__global__ void __launch_bounds__(64, (ALG == 2 ? LCP_Q_OCC2 : 1)) lcp_fwd_quad(FwdArgs P, StepArgs SP, int lds_per_scene, int accept) {
  static_assert(XH == 1 || FUSED, "the dense loader is written for nz <= 16");
  static_assert(ALG == 0 || XH == 1, "the body-space variant holds one x-row per lane");
  static_assert(ALG != 2 || FUSED, "the pinned-only kernel is launched by the contact-list entry points");
  using XVt = XV<TC, XH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4;
  const int Btot = FUSED ? SP.B : P.B;
  const int scene_raw = blockIdx.x * 4 + row;
  const int scene = scene_raw < Btot ? scene_raw : Btot - 1;           // tail rows shadow the last scene (never stored)
  const int nz = FUSED ? 3 * SP.nb : P.nz, nc = FUSED ? SP.nc : (P.m >> 2), e = FUSED ? SP.e : P.e;
  const int m = 4 * nc;                                                  // capacity: strides and output layout
  const int max_iter = FUSED ? SP.max_iter : P.max_iter, lim = FUSED ? SP.lim : P.lim;
  const TC eps = (TC)(FUSED ? SP.eps : P.eps);
  Ws<TI, TC> W(FUSED ? SP.ws : P.ws, scene);
  bool live = scene_raw < Btot;
  if (!FUSED) live = live && ((int)W.meta[0] == accept);
  if (FUSED && ALG == 1 && accept == 3) live = live && ((int)W.meta[21] == 1);   // second pass: the scenes the pinned-only kernel left
  if (!__any(live)) return;
  SceneQ<TI, TC, XH, (ALG == 2) && (LCP_Q_ROWL != 0)> S;
  carve_q(S.L, smem_all + (size_t)row * lds_per_scene, LCP_Q_LDSW != 0, XH, ALG == 0);
  // per-scene contact count (solve_dynamics with detection); a scene without contacts takes the
  // direct KKT solve of engines.py:36-50, which is what the initialisation solve computes
  int ncs = nc;
  int truncated = 0;                                                     // more contacts found than the list holds
  if (FUSED && SP.c_count) { const int c = SP.c_count[scene]; ncs = c < nc ? (c < 0 ? 0 : c) : nc; truncated = c > nc ? LCP_ST_TRUNCATED : 0; }
  int ncw = ncs;
  ncw = max(ncw, __shfl_xor(ncw, 16, 64)); ncw = max(ncw, __shfl_xor(ncw, 32, 64));
  ncw = __builtin_amdgcn_readfirstlane(ncw);
  S.nz = nz; S.nc = ncs; S.ncw = ncw; S.ncap = nc; S.e = e; S.l16 = l16;
  const bool vc = l16 < ncs;                                             // this lane owns a contact
  XVt p;
  TC hn, b;
  if constexpr (FUSED) {
    assemble_q<TI, TC, XH>(S, SP, scene, p, hn, b);
    if (live && vc) W.meta[1 + l16] = S.mu;
    if (live && l16 == 0) { W.meta[0] = (TC)2; W.meta[18] = (TC)1; }
  } else {
    load_dense_q<TI, TC>(S, P, W, scene, p, hn, b);
  }
  if (live && l16 == 0) W.meta[19] = (TC)ncs;
  std::conditional_t<ALG == 0, RedQ<TC>, PrimQ<TC, ALG == 2>> R;
  if constexpr (ALG != 0) {
    // do the equality rows pin the first neq coordinates (A = [I 0], b = 0) in all four scenes of the wave ?  (or are there none)
    __syncthreads();                                                     // (the A rows the loaders put into LDS)
    bool okl = (l16 >= e) || (b == (TC)0);
    static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; if (a < e) okl = okl && ((TC)S.L.AtL[a * 16 + l16] == ((l16 == a) ? (TC)1 : (TC)0)); });
    R.pin_rt = __all(okl) != 0;
    if constexpr (ALG == 2) {
      if (live && l16 == 0) W.meta[21] = R.pin_rt ? (TC)0 : (TC)1;
      if (!R.pin_rt) {                                                   // the general kernel behind this one takes the wave ...
        if (accept == 4 && live) {                                       // ... unless the caller promised there was no such wave (LCP_HINT_PINNED)
          if (l16 < nz) ((TI*)SP.v_new)[(size_t)scene * nz + l16] = nan_of<TI>();
          if (l16 == 0 && SP.status) SP.status[scene] = LCP_ST_NAN;
        }
        return;
      }
    }
  }
  if (blockIdx.x == 0 && lane == 0) { int32_t* tg = FUSED ? SP.tag : P.tag; if (tg) *tg = FUSED ? SP.tag_value : P.tag_value; }
  int status = truncated;
  if constexpr (ALG == 0) status |= prefactor_q<TI, TC, XH>(S, W, live);
  else {
    // body space: nothing of the contact-space pre-factorisation (W = J P J^T, G Q^-1 A^T, (A Q^-1 A^T)^-1) is formed - the
    // backward kernels of this path factor in body space too (bwd_solve_body) and read Q's diagonal, mu and the best iterate only
    if (row_any(l16 < nz && !(S.qd[0] != (TC)0))) status |= LCP_ST_SINGULAR_Q;
    if (live) { W.Qit[l16] = S.qid[0]; W.Qit[128 + l16] = S.qd[0]; }
  }
  TC* const wsx = ws_x<XH>(W);

  TC ta[ALG == 0 ? 32 : 20], tu[ALG == 0 ? 32 : 20];                    // ALG 1, 2: x-rows and equality rows of the body-space system
  XVt x;
  static_for<XH>([&](auto HX) LCP_INL { x.v[HX] = 0; });
  TC y = 0;
  M4<TC> s = m4<TC>(1, 1, 1, 1), z = m4<TC>(1, 1, 1, 1), dinv = m4<TC>(1, 1, 1, 1);
  TC best_resid = inf_of<TC>();
  XVt bx;                                                                 // best iterate (pdipm.py:107-132)
  static_for<XH>([&](auto HX) LCP_INL { bx.v[HX] = 0; });
  TC by = 0;
  M4<TC> bz = m4<TC>(1, 1, 1, 1), bs = bz;
  // BEST_LDS (the pinned body-space kernel): the best iterate is parked in LDS - ten doubles per lane, lane-major, behind the four
  // scene blocks - instead of twenty registers that are written a few times per solve and read once
  constexpr bool BEST_LDS = (ALG == 2) && (LCP_Q_BEST_LDS != 0);
  TC* const bestL = (TC*)(smem_all + (size_t)4 * lds_per_scene) + lane;
  auto park = [&](const XVt& x_, TC y_, const M4<TC>& z_, const M4<TC>& s_) LCP_INL {
    bestL[0] = x_.v[0]; bestL[64] = y_;
    bestL[128] = z_.n; bestL[192] = z_.f1; bestL[256] = z_.f2; bestL[320] = z_.g;
    bestL[384] = s_.n; bestL[448] = s_.f1; bestL[512] = s_.f2; bestL[576] = s_.g;
  };
  if constexpr (BEST_LDS) park(bx, by, bz, bs);
  // ... and so is the affine direction while the corrector solve runs (pdipm.py:138-163): ten more doubles per lane
  TC* const stashL = bestL + 640;
  bool have_best = false, done = !live;
  int n_not = 0, iters = 0;
  const TC mf = (TC)(4 * ncs);                                           // nineq of this scene
#ifdef LCP_Q_PROFILE
  Prof pr; for (int i = 0; i < 10; ++i) pr.t[i] = 0;
  pr.last = clock64();
#endif

#pragma unroll 1
  for (int it = -1; it < max_iter; ++it) {
    if (!__any(!done)) break;
    XVt rx;
    TC ry, mu = 0, resid = 0, szsum = 0;
    M4<TC> rs, rz;
    if (it < 0) {                                                          // init: (p, 0, -h, -b), d = 1 (:57-63)
      rx = p; ry = -b; rs = m4<TC>(0, 0, 0, 0); rz = m4<TC>(-hn, 0, 0, 0); dinv = m4<TC>(1, 1, 1, 1);
    } else {                                                               // residuals (:82-96)
      rx = S.template Gtw<ALG != 0>(z.n, z.f1 - z.f2);
      static_for<XH>([&](auto HX) LCP_INL { rx.v[HX] = rx.v[HX] + S.qd[HX] * x.v[HX] + p.v[HX]; });
      if (e > 0) {
        if constexpr (ALG != 0) {
          if (R.pin()) rx.v[0] += (l16 < e) ? y : (TC)0;                        // A = [I 0]: A^T y is y on the pinned lanes (the product's exact value)
          else { const XVt ay_ = S.Aty(y); rx.v[0] += ay_.v[0]; }
        } else { const XVt ay_ = S.Aty(y); static_for<XH>([&](auto HX) LCP_INL { rx.v[HX] += ay_.v[HX]; }); }
      }
      rs = z;
      TC gn, gt;
      S.template Gv<ALG != 0>(x, gn, gt);
      // F z is lane-local for the contact structure (engines.py:69-73)
      rz = m4<TC>(gn + s.n - hn, gt + s.f1 - z.g, -gt + s.f2 - z.g, s.g - (S.mu * z.n - (z.f1 + z.f2)));
      if (!vc) rz = m4<TC>(0, 0, 0, 0);
      if constexpr (ALG != 0) ry = (e > 0) ? (R.pin() ? ((l16 < e) ? x.v[0] : (TC)0) : (S.Av(x) - b)) : (TC)0;   // (pinned: A x = x_p, b = 0)
      else ry = (e > 0) ? (S.Av(x) - b) : (TC)0;
      TC rx2 = 0;
      static_for<XH>([&](auto HX) LCP_INL { rx2 += (16 * HX + l16 < nz) ? rx.v[HX] * rx.v[HX] : (TC)0; });
      const TC n_rx = row_sum(rx2);
      const TC n_rz = row_sum(rz.n * rz.n + rz.f1 * rz.f1 + rz.f2 * rz.f2 + rz.g * rz.g);
      const TC n_ry = row_sum((l16 < e) ? ry * ry : (TC)0);
      const TC sz = row_sum(vc ? (s.n * z.n + s.f1 * z.f1) + (s.f2 * z.f2 + s.g * z.g) : (TC)0);
      szsum = sz;
      mu = sz / mf; mu = mu < 0 ? -mu : mu;                                // (:91)
      resid = sqrt(n_rz) + sqrt(n_ry) + sqrt(n_rx) + mf * mu;              // (:92-96)
      dinv = vc ? m4<TC>(qdiv(s.n, z.n), qdiv(s.f1, z.f1), qdiv(s.f2, z.f2), qdiv(s.g, z.g)) : m4<TC>(1, 1, 1, 1);   // 1 / d, d = z / s (:98)
    }
    LCP_QTICK(pr, 0)                                                       // residuals, d
    bool sing_;
    if constexpr (ALG == 0) sing_ = factor_q<TI, TC, LCP_Q_LDSW != 0, XH>(ta, tu, R, S, W.R2, dinv, vc LCP_QPROF_PASS);            // (:99-100)
    else sing_ = factor_pq<TI, TC, decltype(R)>(ta, tu, R, S, dinv, vc LCP_QPROF_PASS);
    const bool singular = row_any(sing_);
    if (ALG != 0 && it < 0 && singular && e > 0) status |= LCP_ST_SINGULAR_S11;   // (d = 1: the x block is positive definite, a zero pivot is A's)
    if (it >= 0 && !done) {
      ++iters;
      if (singular && it > 0) { status |= LCP_ST_SINGULAR_T; done = true; }   // except: return best (:99-102)
      else {
        const bool improved = !have_best || (resid < best_resid);             // (:107-132)
        if (improved) {                                                       // best iterate (registers; stored once, below)
          best_resid = resid; n_not = 0; have_best = true;
          if constexpr (BEST_LDS) park(x, y, z, s); else { bx = x; by = y; bz = z; bs = s; }
        } else ++n_not;
        if (n_not == lim || best_resid < eps || mu > mu_limit<TC>()) done = true;   // (:133)
      }
    }
    if (!__any(!done)) break;
    // the iterate the last pass would produce is never evaluated (pdipm.py:176-179 returns `best` straight after the
    // loop): its two solves and the update are dead work in the reference too, and skipped here
    if (it >= 0 && it == max_iter - 1) break;
    LCP_QTICK(pr, 6)                                                       // bookkeeping, best iterate
    XVt ax;
    static_for<XH>([&](auto HX) LCP_INL { ax.v[HX] = 0; });
    TC ay = 0;
    M4<TC> as_ = m4<TC>(0, 0, 0, 0), az = as_;
    const int npass = (it < 0) ? 1 : 2;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
      XVt ox;
      TC oy;
      M4<TC> os, oz;
      if constexpr (ALG == 0) solve_kkt_q<TI, TC, XH>(S, ta, tu, R, dinv, vc, rx, rs, rz, ry, ox, os, oz, oy, pass == 1 LCP_QPROF_PASS);
      else solve_kkt_pq<TI, TC, decltype(R)>(S, ta, tu, R, dinv, vc, rx, rs, rz, ry, ox, os, oz, oy LCP_QPROF_PASS);
      if (it < 0) {
        x = ox; s = os; z = oz; y = oy;                                       // (:60-63)
        const TC smin = row_pmin(vc ? pmin(pmin(s.n, s.f1), pmin(s.f2, s.g)) : inf_of<TC>());       // (once per solve)
        const TC zmin = row_pmin(vc ? pmin(pmin(z.n, z.f1), pmin(z.f2, z.g)) : inf_of<TC>());
        if (smin <= (TC)0) { const TC sh = (TC)1 - smin; s = m4<TC>(s.n + sh, s.f1 + sh, s.f2 + sh, s.g + sh); }   // (:66-75)
        if (zmin <= (TC)0) { const TC sh = (TC)1 - zmin; z = m4<TC>(z.n + sh, z.f1 + sh, z.f2 + sh, z.g + sh); }
        if (!vc) { s = m4<TC>(1, 1, 1, 1); z = s; }
        if (ncs == 0 && !done) {                                              // engines.py:36-50: x = P^-1 u, no LCP
          if constexpr (BEST_LDS) { bestL[0] = x.v[0]; bestL[64] = y; } else { bx = x; by = y; }
          done = true;
        }
      } else if (pass == 0) {
        ax = ox; ay = oy; as_ = os; az = oz;                                  // affine direction (:138-139)
        if constexpr (BEST_LDS) {
          stashL[0] = ax.v[0]; stashL[64] = ay;
          stashL[128] = as_.n; stashL[192] = as_.f1; stashL[256] = as_.f2; stashL[320] = as_.g;
          stashL[384] = az.n; stashL[448] = az.f1; stashL[512] = az.f2; stashL[576] = az.g;
        }
        const TC alpha = pmin(step_pair_q(z, az, s, as_, vc), (TC)1);        // (:142-144)
        auto sc = [&](TC sv, TC dsv, TC zv, TC dzv) { return (sv + alpha * dsv) * (zv + alpha * dzv); };
        const TC t3 = row_sum(vc ? (sc(s.n, as_.n, z.n, az.n) + sc(s.f1, as_.f1, z.f1, az.f1)) + (sc(s.f2, as_.f2, z.f2, az.f2) + sc(s.g, as_.g, z.g, az.g)) : (TC)0);
        const TC t4 = szsum;                                                  // sum(s z) of this iterate: formed with the residuals (:91)
        const TC r3 = t3 / t4, sig = r3 * r3 * r3;                            // (:146-150)
        const TC ms = -mu * sig;
        static_for<XH>([&](auto HX) LCP_INL { rx.v[HX] = 0; });
        ry = 0; rz = m4<TC>(0, 0, 0, 0);
        rs = vc ? m4<TC>(qdiv(ms + as_.n * az.n, s.n), qdiv(ms + as_.f1 * az.f1, s.f1), qdiv(ms + as_.f2 * az.f2, s.f2), qdiv(ms + as_.g * az.g, s.g))
                : m4<TC>(0, 0, 0, 0);                                         // (:153)
      } else {
        if constexpr (BEST_LDS) {
          ax.v[0] = stashL[0]; ay = stashL[64];
          as_ = m4<TC>(stashL[128], stashL[192], stashL[256], stashL[320]); az = m4<TC>(stashL[384], stashL[448], stashL[512], stashL[576]);
        }
        XVt cx;
        static_for<XH>([&](auto HX) LCP_INL { cx.v[HX] = ox.v[HX] + ax.v[HX]; });
        const TC cy = oy + ay;                                                // (:160-163)
        const M4<TC> cs = m4<TC>(os.n + as_.n, os.f1 + as_.f1, os.f2 + as_.f2, os.g + as_.g);
        const M4<TC> cz = m4<TC>(oz.n + az.n, oz.f1 + az.f1, oz.f2 + az.f2, oz.g + az.g);
        const TC alpha = pmin((TC)0.999 * step_pair_q(z, cz, s, cs, vc), (TC)1);   // (:164-166)
        if (!done) {
          static_for<XH>([&](auto HX) LCP_INL { x.v[HX] += alpha * cx.v[HX]; });
          y += alpha * cy;                                                    // (:171-174)
          if (vc) {
            s = m4<TC>(s.n + alpha * cs.n, s.f1 + alpha * cs.f1, s.f2 + alpha * cs.f2, s.g + alpha * cs.g);
            z = m4<TC>(z.n + alpha * cz.n, z.f1 + alpha * cz.f1, z.f2 + alpha * cz.f2, z.g + alpha * cz.g);
          }
        }
      }
      LCP_QTICK(pr, 7)                                                     // step lengths, sigma, update
    }
  }
#ifdef LCP_Q_PROFILE
  if (!FUSED && P.trace && lane == 0) { double* tr = P.trace + (size_t)scene * 4 * max_iter; for (int i = 0; i < 8; ++i) tr[i] = (double)pr.t[i]; tr[8] = (double)iters; }
  long long prof_keep[9];                               // (contact-list entry: the record replaces the tail of the wave's first `s` row, below)
  for (int i = 0; i < 8; ++i) prof_keep[i] = pr.t[i];
  prof_keep[8] = iters;
#endif

  // outputs (natural m-space order: n rows, friction pairs, gamma rows).  The best iterate also goes to the workspace,
  // in fp64, for the backward kernels (lcp.py:29,34: the op keeps its solution)
  if (!live) return;
  if constexpr (BEST_LDS) {
    bx.v[0] = bestL[0]; by = bestL[64];
    bz = m4<TC>(bestL[128], bestL[192], bestL[256], bestL[320]); bs = m4<TC>(bestL[384], bestL[448], bestL[512], bestL[576]);
  }
  static_for<XH>([&](auto HX) LCP_INL { if (16 * HX + l16 < nz) wsx[16 * HX + l16] = bx.v[HX]; else bx.v[HX] = 0; });
  if (l16 < e) W.y[l16] = by; else by = 0;
  if (vc) {
    W.z[l16] = bz.n; W.z[nc + 2 * l16] = bz.f1; W.z[nc + 2 * l16 + 1] = bz.f2; W.z[3 * nc + l16] = bz.g;
    W.s[l16] = bs.n; W.s[nc + 2 * l16] = bs.f1; W.s[nc + 2 * l16 + 1] = bs.f2; W.s[3 * nc + l16] = bs.g;
  } else { bz = m4<TC>(1, 1, 1, 1); bs = bz; }
  bool bad = false;
  static_for<XH>([&](auto HX) LCP_INL { bad = bad || (bx.v[HX] != bx.v[HX]); });
  if (vc) bad = bad || (bz.n != bz.n) || (bs.n != bs.n) || (bz.f1 != bz.f1) || (bz.f2 != bz.f2) || (bz.g != bz.g) ||
                (bs.f1 != bs.f1) || (bs.f2 != bs.f2) || (bs.g != bs.g);
  if (row_any(bad)) status |= LCP_ST_NAN;
  TI* zo = (TI*)(FUSED ? SP.z : P.z);
  TI* so = (TI*)(FUSED ? SP.s : P.s);
  TI* yo = (TI*)(FUSED ? SP.y : P.y);
  const bool vslot = l16 < nc;                                            // slot of the (padded) contact list
  if (vslot && zo) {
    TI* o = zo + (size_t)scene * m;
    const TI k = vc ? (TI)1 : (TI)0;                                      // padded slots report 0
    o[l16] = k * (TI)bz.n; o[nc + 2 * l16] = k * (TI)bz.f1; o[nc + 2 * l16 + 1] = k * (TI)bz.f2; o[3 * nc + l16] = k * (TI)bz.g;
  }
  if (vslot && so) {
    TI* o = so + (size_t)scene * m;
    const TI k = vc ? (TI)1 : (TI)0;
    o[l16] = k * (TI)bs.n; o[nc + 2 * l16] = k * (TI)bs.f1; o[nc + 2 * l16 + 1] = k * (TI)bs.f2; o[3 * nc + l16] = k * (TI)bs.g;
  }
  if (l16 < e && yo) yo[(size_t)scene * e + l16] = (TI)by;
  if (FUSED) {
    static_for<XH>([&](auto HX) LCP_INL {
      const int j = 16 * HX + l16;
      if (j < nz) {
        const TC nv = -bx.v[HX];                                              // engines.py:76-77
        ((TI*)SP.v_new)[(size_t)scene * nz + j] = (TI)nv;
        if (SP.p_new) ((TI*)SP.p_new)[(size_t)scene * nz + j] = (TI)((TC)((const TI*)SP.pos)[(size_t)scene * nz + j] + nv * (TC)SP.dt);   // bodies.py:81
      }
    });
    if (l16 == 0) { if (SP.iters) SP.iters[scene] = iters; if (SP.status) SP.status[scene] = status; }
#ifdef LCP_Q_PROFILE
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0 && so && m >= 16) { TI* o = so + (size_t)scene * m + (m - 9); for (int i = 0; i < 9; ++i) o[i] = (TI)prof_keep[i]; }
#endif
  } else {
    if (l16 < nz) ((TI*)P.x)[(size_t)scene * nz + l16] = (TI)bx.v[0];
    if (l16 == 0) { if (P.iters) P.iters[scene] = iters; if (P.status) P.status[scene] = status; }
  }
}

// Backward at an iterate whose T = W + diag(s / z) has an exact zero (or NaN) pivot.  A forward in contact space never leaves one
// (pdipm.py:99-102: an iterate whose factorisation fails is not recorded), the body-space forward factors another matrix and can
// run a step further into convergence, where s / z underflows against a W that redundant contacts make singular.  Then - and
// only then - the factorisation is repeated with s / z floored at 1e-9 x the row's diagonal of W (the perturbation
// lcp_primal.hip's backward applies always): a gradient of the converged solution instead of NaN.
template <typename TI, typename TC, int XH>
__device__ __forceinline__ void factor_bwd_q(TC (&ta)[32], TC (&tu)[32], RedQ<TC>& R, const SceneQ<TI, TC, XH>& S, const TC* W2q,
                                             M4<TC>& dinv, bool vc LCP_QPROF_ARG) {
  const bool sing = factor_q<TI, TC, false, XH>(ta, tu, R, S, W2q, dinv, vc LCP_QPROF_PASS);
  if (!__any(row_any(sing))) return;
  const int l16 = S.l16;
  const TC waa = W2q[((((size_t)(l16 >> 1)) * 2 + 0) * 16 + l16) * 2 + (l16 & 1)];
  const TC wuu = W2q[((((size_t)((16 + l16) >> 1)) * 2 + 1) * 16 + l16) * 2 + (l16 & 1)];
  const TC fa = (TC)1e-9 * waa, fu = (TC)1e-9 * wuu;
  if (vc && row_any(sing)) { dinv.n = fmax_(dinv.n, fa); dinv.f1 = fmax_(dinv.f1, fu); dinv.f2 = fmax_(dinv.f2, fu); }
  factor_q<TI, TC, false, XH>(ta, tu, R, S, W2q, dinv, vc LCP_QPROF_PASS);
}

#ifndef LCP_Q_BWD_REFINE
#define LCP_Q_BWD_REFINE 2      // refinement steps of the body-space backward solve
#endif
// ---------------------------------------------------------------- backward solve in body space (lcp.py:44-50)
// The backward of a forward that ran in body space (ALG = 1, 2): the same K = [[Q + G^T M^-1 G, A^T], [A, 0]] factorisation at the
// best iterate, nothing read from the workspace but Q's diagonal, mu and the iterate (no W: the forward does not form it).
// At a converged iterate the ratios D = s / z of the active rows underflow against Q (1e-12 and below) and Q + G^T M^-1 G would
// lose Q: as in lcp_primal_step.inc the factorisation uses D floored at 1e-9 x the row's effective inverse mass j Q^-1 j^T, and
// ONE step of iterative refinement on the UNREDUCED equations (residuals formed with M = F_c + diag(D), the true D, not with
// M^-1) takes the perturbation out again - 1e-8 of the natural scale |g| / min Q against the contact-space solve (LCP_Q_BWD_REFINE
// steps; they cost a KKT solve each and the kernels around them are bound by their stores).
// g: entry l16 of d(loss)/dx.  Returns dx (x lanes), dlam per contact, dnu (equality lanes).
template <typename TI, typename TC>
__device__ __forceinline__ void bwd_solve_body(const SceneQ<TI, TC, 1>& S, bool vc, const M4<TC>& z, const M4<TC>& s, TC g,
                                               TC& dx, M4<TC>& dl, TC& dnu LCP_QPROF_ARG) {
  const int l16 = S.l16, nz = S.nz, e = S.e;
  const M4<TC> dinv = vc ? m4<TC>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g) : m4<TC>(1, 1, 1, 1);   // 1 / d, d = z / s (lcp.py:44)
  M4<TC> dfl = dinv;
  {
    TC wn = 0, wt = 0;
    static_for<16>([&](auto J) LCP_INL { const TC qi = bc<J>(S.qid[0]); wn = fma((TC)S.jc[J] * (TC)S.jc[J], qi, wn); wt = fma((TC)S.jt[J] * (TC)S.jt[J], qi, wt); });
    constexpr TC BWD_FLOOR = (TC)1e-9;
    if (vc) { dfl.n = fmax_(dinv.n, BWD_FLOOR * wn); dfl.f1 = fmax_(dinv.f1, BWD_FLOOR * wt); dfl.f2 = fmax_(dinv.f2, BWD_FLOOR * wt); }
  }
  PrimQ<TC, false> R;
  {                                                     // A = [I 0] in all four scenes of the wave ?  (the pinned floor of the demo worlds)
    bool okl = true;
    static_for<EQ>([&](auto A_) LCP_INL { constexpr int a = A_; if (a < e) okl = okl && ((TC)S.L.AtL[a * 16 + l16] == ((l16 == a) ? (TC)1 : (TC)0)); });
    R.pin_rt = __all(okl) != 0;
  }
  TC xr[20], er[20];
  factor_pq<TI, TC, PrimQ<TC, false>>(xr, er, R, S, dfl, vc LCP_QPROF_PASS);                               // lcp.py:46
  const M4<TC> zero = m4<TC>(0, 0, 0, 0);
  XV<TC, 1> rx, ox;
  M4<TC> ds;
  rx.v[0] = g;
  solve_kkt_pq<TI, TC, PrimQ<TC, false>>(S, xr, er, R, dfl, vc, rx, zero, zero, (TC)0, ox, ds, dl, dnu LCP_QPROF_PASS);   // lcp.py:47-50
  dx = ox.v[0];
  // residuals of  Q dx + G^T dl + A^T dnu = -g ,  G dx - M dl = 0 ,  A dx = 0  with the TRUE D, then a correction solve; twice
  // (the first pass leaves ~1e-7 of the natural scale on the worst-conditioned scenes of the stack configs, the second 1e-9)
#pragma unroll 1
  for (int ref = 0; ref < LCP_Q_BWD_REFINE; ++ref) {
    const XV<TC, 1> gl = S.template Gtw<false>(vc ? dl.n : (TC)0, vc ? dl.f1 - dl.f2 : (TC)0);
    TC r1 = -g - (S.qd[0] * dx + gl.v[0]);
    if (e > 0) { const XV<TC, 1> ay = S.Aty(dnu); r1 -= ay.v[0]; }
    if (!(l16 < nz)) r1 = 0;
    TC gn, gt;
    ox.v[0] = dx;
    S.template Gv<false>(ox, gn, gt);
    M4<TC> r3 = m4<TC>(-(gn - dinv.n * dl.n), -(gt - (dinv.f1 * dl.f1 + dl.g)), -(-gt - (dinv.f2 * dl.f2 + dl.g)),
                       (S.mu * dl.n - (dl.f1 + dl.f2)) + dinv.g * dl.g);
    if (!vc) r3 = zero;
    const TC r2 = (e > 0) ? -S.Av(ox) : (TC)0;
    XV<TC, 1> cx;
    M4<TC> cs, cl;
    TC cnu;
    rx.v[0] = -r1;
    solve_kkt_pq<TI, TC, PrimQ<TC, false>>(S, xr, er, R, dfl, vc, rx, zero, m4<TC>(-r3.n, -r3.f1, -r3.f2, -r3.g), -r2, cx, cs, cl, cnu LCP_QPROF_PASS);
    dx += cx.v[0]; dnu += cnu;
    dl = m4<TC>(dl.n + cl.n, dl.f1 + cl.f1, dl.f2 + cl.f2, dl.g + cl.g);
  }
}

// ---------------------------------------------------------------- backward kernel (lcp.py:37-64)
// BODY: the workspace was left by a body-space forward (lcp_fwd_quad ALG = 1, 2 behind the contact-list entry points): no W in it,
// the solve runs in body space (bwd_solve_body); otherwise the contact-space W is re-factored (factor_bwd_q).
template <typename TI, typename TC, bool BODY = false>
__global__ void __launch_bounds__(64) lcp_bwd_quad(BwdArgs P, int lds_per_scene, int accept) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4;
  const int scene_raw = blockIdx.x * 4 + row;
  const int scene = scene_raw < P.B ? scene_raw : P.B - 1;
  const int nz = P.nz, nc = P.m >> 2, e = P.e, m = P.m;
  Ws<TI, TC> W(P.ws, scene);
  const bool live = (scene_raw < P.B) && ((int)W.meta[0] == accept);
  if (!__any(live)) return;
  SceneQ<TI, TC> S;
  carve_q(S.L, smem_all + (size_t)row * lds_per_scene, false);
  int ncs = live ? (int)W.meta[19] : 0;                                  // live contacts of the scene (set by the forward)
  ncs = ncs < 0 ? 0 : (ncs > nc ? nc : ncs);
  int ncw = ncs;
  ncw = max(ncw, __shfl_xor(ncw, 16, 64)); ncw = max(ncw, __shfl_xor(ncw, 32, 64));
  ncw = __builtin_amdgcn_readfirstlane(ncw);
  S.nz = nz; S.nc = ncs; S.ncw = ncw; S.ncap = nc; S.e = e; S.l16 = l16;
  const bool vc = l16 < ncs;
  {
    const TI* Grow_n = (const TI*)P.G + ((size_t)scene * m + (vc ? l16 : 0)) * nz;
    const TI* Grow_t = (const TI*)P.G + ((size_t)scene * m + nc + 2 * (vc ? l16 : 0)) * nz;
    static_for<16>([&](auto J) LCP_INL {
      S.jc[J] = (vc && J < nz) ? Grow_n[J] : (TI)0;
      S.jt[J] = (vc && J < nz) ? Grow_t[J] : (TI)0;
      S.L.GL[l16 * 16 + J] = S.jc[J]; S.L.GTL[l16 * 16 + J] = S.jt[J];
    });
    const TI* A = (const TI*)P.A + (size_t)scene * e * nz;
    static_for<16>([&](auto K) LCP_INL {
      if (l16 < EQ) S.L.AtL[l16 * 16 + K] = (l16 < e && K < nz) ? A[l16 * nz + K] : (TI)0;
    });
    if constexpr (!BODY) static_for<EQ>([&](auto A_) LCP_INL {
      S.gan[A_] = W.GAc[(l16 * 2 + 0) * EQ + A_]; S.gat[A_] = W.GAc[(l16 * 2 + 1) * EQ + A_];
      S.s11row[A_] = (l16 < EQ) ? W.S11i[l16 * EQ + A_] : (TC)0;
    });
    S.qid[0] = W.Qit[l16]; S.qd[0] = BODY ? W.Qit[128 + l16] : (TC)0;
    S.mu = vc ? W.meta[1 + l16] : (TC)0;
  }
  __syncthreads();
  const TC x = (l16 < nz) ? W.x[l16] : (TC)0, y = (l16 < e) ? W.y[l16] : (TC)0;
  const M4<TC> z = vc ? m4<TC>(W.z[l16], W.z[nc + 2 * l16], W.z[nc + 2 * l16 + 1], W.z[3 * nc + l16]) : m4<TC>(1, 1, 1, 1);
  const M4<TC> s = vc ? m4<TC>(W.s[l16], W.s[nc + 2 * l16], W.s[nc + 2 * l16 + 1], W.s[3 * nc + l16]) : m4<TC>(1, 1, 1, 1);
  XV<TC, 1> g;
  g.v[0] = (l16 < nz) ? (TC)((const TI*)P.dl_dx)[(size_t)scene * nz + l16] : (TC)0;
  if (P.tag && *P.tag != P.tag_value) g.v[0] = nan_of<TC>();      // a workspace another kernel family laid out: NaN gradients, not a misread
#ifdef LCP_Q_PROFILE
  Prof pr; pr.last = 0;
#endif
  TC dx, dnu;
  M4<TC> dl;
  if constexpr (BODY) {
    bwd_solve_body<TI, TC>(S, vc, z, s, g.v[0], dx, dl, dnu LCP_QPROF_PASS);
  } else {
    M4<TC> dinv = m4<TC>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g);                  // 1 / d, d = z / s (lcp.py:44)
    TC ta[32], tu[32];
    RedQ<TC> R;
    factor_bwd_q<TI, TC, 1>(ta, tu, R, S, W.R2, dinv, vc LCP_QPROF_PASS);                // lcp.py:46
    XV<TC, 1> dxv;
    M4<TC> ds;
    const M4<TC> zero = m4<TC>(0, 0, 0, 0);
    solve_kkt_q<TI, TC, 1>(S, ta, tu, R, dinv, vc, g, zero, zero, (TC)0, dxv, ds, dl, dnu, false LCP_QPROF_PASS);  // lcp.py:47-50
    dx = dxv.v[0];
  }
  if (!live) return;
  // outer products (lcp.py:52-61), one output row per instruction, lanes over the columns
  if (P.dp && l16 < nz) ((TI*)P.dp)[(size_t)scene * nz + l16] = (TI)dx;
  if (P.db && l16 < e) ((TI*)P.db)[(size_t)scene * e + l16] = (TI)(-dnu);
  const bool vslot = l16 < nc;                                            // padded contact slots get zero gradients
  const M4<TC> zq = vc ? z : m4<TC>(0, 0, 0, 0);
  if (P.dh && vslot) {
    TI* o = (TI*)P.dh + (size_t)scene * m;
    o[l16] = (TI)(-dl.n); o[nc + 2 * l16] = (TI)(-dl.f1); o[nc + 2 * l16 + 1] = (TI)(-dl.f2); o[3 * nc + l16] = (TI)(-dl.g);
  }
  if (P.dQ) {
    TI* o = (TI*)P.dQ + (size_t)scene * nz * nz;
    static_for<16>([&](auto Rr) LCP_INL {
      if (Rr < nz) { const TC dxr = bc<Rr>(dx), xr = bc<Rr>(x); if (l16 < nz) o[Rr * nz + l16] = (TI)((TC)0.5 * (dxr * x + xr * dx)); }
    });
  }
  if (P.dA && e > 0) {
    TI* o = (TI*)P.dA + (size_t)scene * e * nz;
    static_for<EQ>([&](auto Rr) LCP_INL {
      if (Rr < e) { const TC dn = bc<Rr>(dnu), yr = bc<Rr>(y); if (l16 < nz) o[Rr * nz + l16] = (TI)(dn * x + yr * dx); }
    });
  }
  if (P.dG) {
    TI* o = (TI*)P.dG + (size_t)scene * m * nz;
    static_for<16>([&](auto C) LCP_INL {
      if (C < nc) {
        const TC a0 = bc<C>(dl.n), a1 = bc<C>(dl.f1), a2 = bc<C>(dl.f2), a3 = bc<C>(dl.g);
        const TC b0 = bc<C>(zq.n), b1 = bc<C>(zq.f1), b2 = bc<C>(zq.f2), b3 = bc<C>(zq.g);
        if (l16 < nz) {
          o[(size_t)C * nz + l16] = (TI)(a0 * x + b0 * dx);
          o[(size_t)(nc + 2 * C) * nz + l16] = (TI)(a1 * x + b1 * dx);
          o[(size_t)(nc + 2 * C + 1) * nz + l16] = (TI)(a2 * x + b2 * dx);
          o[(size_t)(3 * nc + C) * nz + l16] = (TI)(a3 * x + b3 * dx);
        }
      }
    });
  }
  if (P.dF) {
    TI* o = (TI*)P.dF + (size_t)scene * m * m;
    // dF[i][j] = -dlam_i lam_j (lcp.py:54): 16 KB per scene, the bulk of what this kernel writes.  Each lane takes four
    // CONSECUTIVE columns of every row (lam in the natural row order [n | friction pairs | gamma], fetched once through
    // LDS), so a row leaves as one 16-byte store per lane - 256 contiguous bytes per scene and instruction.
    TC* LAM = S.L.GAL;                                                   // (free in this kernel: 128 TC >= 4 nc)
    __syncthreads();
    if (vslot) { LAM[l16] = zq.n; LAM[nc + 2 * l16] = zq.f1; LAM[nc + 2 * l16 + 1] = zq.f2; LAM[3 * nc + l16] = zq.g; }
    __syncthreads();
    TC lq[4] = {0, 0, 0, 0};
    if (vslot) { lq[0] = LAM[4 * l16]; lq[1] = LAM[4 * l16 + 1]; lq[2] = LAM[4 * l16 + 2]; lq[3] = LAM[4 * l16 + 3]; }
    static_for<16>([&](auto C) LCP_INL {
      if (C < nc) {
        const TC a0 = bc<C>(dl.n), a1 = bc<C>(dl.f1), a2 = bc<C>(dl.f2), a3 = bc<C>(dl.g);
        if (vslot) {
          auto wr = [&](int i, TC dli) {
            TI* r = o + (size_t)i * m + 4 * l16;
            store4(r, (TI)(-dli * lq[0]), (TI)(-dli * lq[1]), (TI)(-dli * lq[2]), (TI)(-dli * lq[3]));
          };
          wr(C, a0); wr(nc + 2 * C, a1); wr(nc + 2 * C + 1, a2); wr(3 * nc + C, a3);
        }
      }
    });
  }
}


// ---------------------------------------------------------------- backward of the fused step w.r.t. the physical inputs
// d(loss)/d(v_new) -> d(loss)/d(Mdiag, v, f, restitution, friction, contact normal / arms): the reference gets these by
// autograd through the engine assembly (engines.py:31-32,50-74; world.py:144-234) after LCPFunction.backward
// (lcp.py:37-64) has materialised dQ, dp, dG, dh, dF.  Here the rank-1 LCP gradients are contracted in registers:
//   dp = dx, dQ_jj = dx_j x_j, dG_row = dlam_row x + lam_row dx, dh = -dlam, dF[gamma_c, n_c] = -dlam_gamma lam_n
// and only ~0.6 KB per scene leaves the chip instead of the 21.6 KB of dense gradients.
template <typename TI, typename TC, int XH, bool BODY = false>
__global__ void __launch_bounds__(64) lcp_bwd_step_quad(StepArgs SP, StepBwdArgs Gd, int lds_per_scene) {
  static_assert(!BODY || XH == 1, "the body-space forward serves nz <= 16");
  using XVt = XV<TC, XH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4;
  const int scene_raw = blockIdx.x * 4 + row;
  const int scene = scene_raw < SP.B ? scene_raw : SP.B - 1;
  const int nb = SP.nb, nz = 3 * SP.nb, nc = SP.nc, e = SP.e;
  Ws<TI, TC> W(SP.ws, scene);
  const bool live = scene_raw < SP.B;
  SceneQ<TI, TC, XH> S;
  carve_q(S.L, smem_all + (size_t)row * lds_per_scene, false, XH);
  int ncs = (int)W.meta[19];                                             // contacts the forward solved with
  ncs = ncs < 0 ? 0 : (ncs > nc ? nc : ncs);
  int ncw = ncs;
  ncw = max(ncw, __shfl_xor(ncw, 16, 64)); ncw = max(ncw, __shfl_xor(ncw, 32, 64));
  ncw = __builtin_amdgcn_readfirstlane(ncw);
  S.nz = nz; S.nc = ncs; S.ncw = ncw; S.ncap = nc; S.e = e; S.l16 = l16;
  const bool vc = l16 < ncs;
  XVt p_;
  TC hn_, b_;
  assemble_q<TI, TC, XH>(S, SP, scene, p_, hn_, b_);                      // the same rows the forward solved with
  if constexpr (!BODY) static_for<EQ>([&](auto A_) LCP_INL {
    S.gan[A_] = W.GAc[(l16 * 2 + 0) * EQ + A_]; S.gat[A_] = W.GAc[(l16 * 2 + 1) * EQ + A_];
    S.s11row[A_] = (l16 < EQ) ? W.S11i[l16 * EQ + A_] : (TC)0;
  });
  __syncthreads();
  XVt x, g;
  static_for<XH>([&](auto HX) LCP_INL {
    const int j = 16 * HX + l16;
    x.v[HX] = (j < nz) ? ws_x<XH>(W)[j] : (TC)0;
    // v_new = -x (engines.py:76-77)  =>  d(loss)/dx = -d(loss)/d(v_new)
    g.v[HX] = (j < nz) ? -(TC)((const TI*)Gd.dl_dv)[(size_t)scene * nz + j] : (TC)0;
  });
  const M4<TC> z = vc ? m4<TC>(W.z[l16], W.z[nc + 2 * l16], W.z[nc + 2 * l16 + 1], W.z[3 * nc + l16]) : m4<TC>(1, 1, 1, 1);
  const M4<TC> s = vc ? m4<TC>(W.s[l16], W.s[nc + 2 * l16], W.s[nc + 2 * l16 + 1], W.s[3 * nc + l16]) : m4<TC>(1, 1, 1, 1);
  if (SP.tag && *SP.tag != SP.tag_value) static_for<XH>([&](auto HX) LCP_INL { g.v[HX] = nan_of<TC>(); });   // (foreign workspace: NaN gradients)
#ifdef LCP_Q_PROFILE
  Prof pr; pr.last = 0;
#endif
  XVt dx;
  TC dnu;
  M4<TC> dl;
  if constexpr (BODY) {
    bwd_solve_body<TI, TC>(S, vc, z, s, g.v[0], dx.v[0], dl, dnu LCP_QPROF_PASS);
  } else {
    M4<TC> dinv = m4<TC>(s.n / z.n, s.f1 / z.f1, s.f2 / z.f2, s.g / z.g);
    TC ta[32], tu[32];
    RedQ<TC> R;
    factor_bwd_q<TI, TC, XH>(ta, tu, R, S, W.R2, dinv, vc LCP_QPROF_PASS);               // lcp.py:46
    M4<TC> ds;
    const M4<TC> zero = m4<TC>(0, 0, 0, 0);
    solve_kkt_q<TI, TC, XH>(S, ta, tu, R, dinv, vc, g, zero, zero, (TC)0, dx, ds, dl, dnu, false LCP_QPROF_PASS);  // lcp.py:47-50
  }
  // x-space vectors to LDS so that a contact lane can read the entries of its two bodies (GAL is free in this kernel:
  // 128 TC = X[32] DX[32] CR[16] CF[16] B12[32 ints])
  TC* X = S.L.GAL; TC* DX = X + 32; TC* CR = X + 64; TC* CF = X + 80;
  int* B12 = (int*)(X + 96);
  static_for<XH>([&](auto HX) LCP_INL { X[16 * HX + l16] = x.v[HX]; DX[16 * HX + l16] = dx.v[HX]; });
  __syncthreads();
  const TI* vv = (const TI*)SP.v + (size_t)scene * nz;
  TC gh_rbar = 0;                                                         // (dh * rbar)_c: feeds d v through h = (Jc v) rbar
  {
    TC cr = 0, cf = 0, dnx = 0, dny = 0, d1x = 0, d1y = 0, d2x = 0, d2y = 0;
    int b1 = 0, b2 = 0;
    if (vc) {
      const size_t cb = (size_t)scene * nc + l16;
      const TC nx = ((const TI*)SP.c_n)[cb * 2], ny = ((const TI*)SP.c_n)[cb * 2 + 1];
      const TC p1x = ((const TI*)SP.c_p1)[cb * 2], p1y = ((const TI*)SP.c_p1)[cb * 2 + 1];
      const TC p2x = ((const TI*)SP.c_p2)[cb * 2], p2y = ((const TI*)SP.c_p2)[cb * 2 + 1];
      b1 = SP.c_i1[cb]; b2 = SP.c_i2[cb];
      const TC rbar = (TC)0.5 * ((TC)((const TI*)SP.rest)[(size_t)scene * nb + b1] + (TC)((const TI*)SP.rest)[(size_t)scene * nb + b2]);
      const TC jn[6] = {p1x * ny - p1y * nx, nx, ny, -(p2x * ny - p2y * nx), -nx, -ny};      // world.py:177-183
      const TC gh = -dl.n;                                                  // dh = -dlam (lcp.py:56)
      const TC af = dl.f1 - dl.f2, lf = z.f1 - z.f2;                        // Jf rows are +jt, -jt (world.py:191-192)
      TC gjn[6], gjf[6], jnv = 0;
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int col = (q < 3) ? 3 * b1 + q : 3 * b2 + (q - 3);
        const TC xq = X[col], dxq = DX[col], vq = (TC)vv[col];
        jnv = fma(jn[q], vq, jnv);
        gjn[q] = dl.n * xq + z.n * dxq + gh * rbar * vq;                    // dG row n (lcp.py:53) + h = (Jc v) rbar
        gjf[q] = af * xq + lf * dxq;
      }
      gh_rbar = gh * rbar;
      cr = (TC)0.5 * gh * jnv;                                              // rbar = (rest_b1 + rest_b2) / 2 (world.py:144-151)
      cf = (TC)0.5 * (-dl.g * z.n);                                         // dF[gamma_c, n_c] = -dlam_g lam_n (lcp.py:54), F = mu there
      // jn = [p1 x n, n | -(p2 x n), -n] ; jf = [p1 x t, t | -(p2 x t), -t], t = (ny, -nx)   (utils.py:93-102)
      dnx = -gjn[0] * p1y + gjn[1] + gjn[3] * p2y - gjn[4] - gjf[0] * p1x - gjf[2] + gjf[3] * p2x + gjf[5];
      dny = gjn[0] * p1x + gjn[2] - gjn[3] * p2x - gjn[5] - gjf[0] * p1y + gjf[1] + gjf[3] * p2y - gjf[4];
      d1x = gjn[0] * ny - gjf[0] * nx; d1y = -gjn[0] * nx - gjf[0] * ny;
      d2x = -gjn[3] * ny + gjf[3] * nx; d2y = gjn[3] * nx + gjf[3] * ny;
    }
    CR[l16] = cr; CF[l16] = cf; B12[l16] = b1; B12[16 + l16] = b2;
    if (live && l16 < nc) {
      const size_t cb = (size_t)scene * nc + l16;
      if (Gd.dcn) { ((TI*)Gd.dcn)[cb * 2] = (TI)dnx; ((TI*)Gd.dcn)[cb * 2 + 1] = (TI)dny; }
      if (Gd.dcp1) { ((TI*)Gd.dcp1)[cb * 2] = (TI)d1x; ((TI*)Gd.dcp1)[cb * 2 + 1] = (TI)d1y; }
      if (Gd.dcp2) { ((TI*)Gd.dcp2)[cb * 2] = (TI)d2x; ((TI*)Gd.dcp2)[cb * 2 + 1] = (TI)d2y; }
    }
  }
  const XVt dv_h = S.Gtw(gh_rbar, (TC)0);                                  // Jc^T (dh rbar)
  __syncthreads();
  if (!live) return;
  static_for<XH>([&](auto HX) LCP_INL {
    const int j = 16 * HX + l16;
    if (j < nz) {
      const size_t o = (size_t)scene * nz + j;
      const TC md = (TC)((const TI*)SP.Mdiag)[o], v = (TC)vv[j], dxj = dx.v[HX];
      if (Gd.dMdiag) ((TI*)Gd.dMdiag)[o] = (TI)(dxj * x.v[HX] + dxj * v);  // Q = diag(M) (dQ, lcp.py:59-60) and p = M v + dt f
      if (Gd.dv) ((TI*)Gd.dv)[o] = (TI)(dxj * md + dv_h.v[HX]);
      if (Gd.df) ((TI*)Gd.df)[o] = (TI)(dxj * (TC)SP.dt);
    }
  });
  if (Gd.dJe && e > 0) {                                                  // dA = dnu (x) x + nu (x) dx (lcp.py:57; A = Je)
    const TC nu = (l16 < e) ? W.y[l16] : (TC)0;
    TI* o = (TI*)Gd.dJe + (size_t)scene * e * nz;
    static_for<EQ>([&](auto A_) LCP_INL {
      constexpr int a = A_;
      if (a < e) {
        const TC dn = bc<a>(dnu), yr = bc<a>(nu);
        static_for<XH>([&](auto HX) LCP_INL { const int j = 16 * HX + l16; if (j < nz) o[a * nz + j] = (TI)(dn * x.v[HX] + yr * dx.v[HX]); });
      }
    });
  }
  if (l16 < nb) {                                                          // per-body sums over the contacts, fixed order
    TC ar = 0, af = 0;
    for (int c = 0; c < ncs; ++c) {
      const bool hit = (B12[c] == l16) || (B12[16 + c] == l16);
      const TC w = ((B12[c] == l16) ? (TC)1 : (TC)0) + ((B12[16 + c] == l16) ? (TC)1 : (TC)0);
      if (hit) { ar += w * CR[c]; af += w * CF[c]; }
    }
    if (Gd.drest) ((TI*)Gd.drest)[(size_t)scene * nb + l16] = (TI)ar;
    if (Gd.dfric) ((TI*)Gd.dfric)[(size_t)scene * nb + l16] = (TI)af;
  }
}

}  // namespace q16

// ---------------------------------------------------------------- host-side launchers
bool quad_supported(int nz, int m, int e) { return (m % 4 == 0) && (m / 4 <= q16::NCQ) && nz <= 16 && e <= q16::EQ; }
// contact-list entry points only (lcp_solve_dynamics_f32 / lcp_step_backward_f32): up to ten bodies
bool quad_step_supported(int nz, int m, int e) { return (m % 4 == 0) && (m / 4 <= q16::NCQ) && nz <= 32 && e <= q16::EQ; }

// LDS bytes per scene.  The four scenes of a wave issue every LDS access together (16 lanes each, same offsets inside
// their blocks): the block stride is padded to 64 B modulo 256 B = 16 banks modulo 64, so that the four 16-lane groups
// land on four disjoint bank groups (a stride that is a multiple of 128 B - the unpadded 3456 B - puts them all on the
// same banks).  Measured: SQ_LDS_BANK_CONFLICT 273 k -> 169 k cycles per launch; no change of the kernel time - the
// single wave waits out the LDS latency either way.
template <typename TC, typename TI = float>
static size_t q16_lds(bool with_w, int xh = 1, bool with_gal = true) {
  q16::LdsQ<TI, TC> L;
  size_t n = q16::carve_q<TI, TC>(L, nullptr, with_w, xh, with_gal);
  while (n % 256 != 64) n += 16;
  return n;
}

// The dense LCPFunction boundary keeps the contact-space factorisation: it is the reference's own formulation, and with it the
// exit tests of pdipm.py:133 fall where the reference's fall (iteration counts equal to the oracle's even where a solve converges
// to rounding - tests/test_hip_parity.py).  The body-space variant takes the same Newton steps to ~1e-12 but can leave a converged
// solve one iteration later; it serves the contact-list entry points (1 builds it here too, for profiling: the phase trace of
// LCP_Q_PROFILE is written by the dense forward).
#ifndef LCP_Q_DENSE_BODY_SPACE
#define LCP_Q_DENSE_BODY_SPACE 0
#endif
int quad_forward(const FwdArgs& P, int compute, int accept, void* stream, int io_f64) {
  StepArgs SP = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P.B + 3) / 4), blk(64);
  if (io_f64) {                                  // the reference's native dtype (physics/utils.py:34): fp64 loads / stores
    const int ls = (int)q16_lds<double, double>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<double, double, false, 1>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  } else if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, false, 1, LCP_Q_DENSE_BODY_SPACE>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  } else {
    const int ls = (int)q16_lds<float>(LCP_Q_LDSW != 0);
    hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, false, 1>), grid, blk, 4 * ls, st, P, SP, ls, accept);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

// does quad_step / quad_step_backward run the body-space kernels for these arguments ?  (fp64 arithmetic, nz <= 16, not forced off)
bool quad_step_is_body_space(int nz, int compute, int body_space) { return body_space && compute == LCP_COMPUTE_F64 && nz <= 16; }

// `body_space`: factor / solve the (nz + neq)-row body-space system instead of the 32-row contact-space one (fp64 arithmetic, nz <= 16)
// Below LCP_SOLO_MAX_B scenes the body-space step runs ONE scene per wavefront (lcp_solo.hip): four scenes per wavefront leave
// most SIMDs of the chip without a wave (B / 4 wavefronts on 1024 SIMDs) and a scene's dependent chain takes the same time
// whatever the batch.  Measured cross-over: see DESIGN.md (latency curve B = 256 .. 4096).
#ifndef LCP_SOLO_MAX_B
#define LCP_SOLO_MAX_B 1024
#endif
int quad_step(const StepArgs& SP, int compute, void* stream, int body_space, int solo, bool pinned) {
  FwdArgs P = {};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((SP.B + 3) / 4), blk(64);
  const bool wide = 3 * SP.nb > 16;
  if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(LCP_Q_LDSW != 0, wide ? 2 : 1);
    const int lb = (int)q16_lds<double>(LCP_Q_LDSW != 0, 1, false);          // body-space kernels: no contact-space tables
    if (wide) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 2>), grid, blk, 4 * ls, st, P, SP, ls, 2);
    else if (body_space && (solo > 0 || (solo < 0 && SP.B <= LCP_SOLO_MAX_B))) {
      int rc = solo_step(SP, stream, pinned);                                                                     // pinned leading coordinates, one scene per wave
      if (rc) return rc;
      if (!pinned) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 1>), grid, blk, 4 * lb, st, P, SP, lb, 3);   // whatever that one left
    } else if (body_space) {
      // (+ the parked best iterate and affine direction: 2 x 10 doubles per lane)
      // (accept 4 = LCP_HINT_PINNED: nothing is launched behind; a wave that does not qualify returns NaN velocities)
      hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 2>), grid, blk, 4 * lb + 20 * 64 * sizeof(double), st, P, SP, lb, pinned ? 4 : 2);   // pinned leading coordinates
      if (!pinned) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1, 1>), grid, blk, 4 * lb, st, P, SP, lb, 3);   // whatever that one left
    }
    else hipLaunchKernelGGL((q16::lcp_fwd_quad<float, double, true, 1>), grid, blk, 4 * ls, st, P, SP, ls, 2);
  } else {
    const int ls = (int)q16_lds<float>(LCP_Q_LDSW != 0, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, true, 2>), grid, blk, 4 * ls, st, P, SP, ls, 2);
    else hipLaunchKernelGGL((q16::lcp_fwd_quad<float, float, true, 1>), grid, blk, 4 * ls, st, P, SP, ls, 2);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int quad_backward(const BwdArgs& P, int compute, int accept, void* stream, int io_f64, int body) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((P.B + 3) / 4), blk(64);
  if (body) {                                    // workspace of a body-space forward (fp32 I/O, fp64 arithmetic)
    if (io_f64 || compute != LCP_COMPUTE_F64) return LCP_E_BADARG;
    const int ls = (int)q16_lds<double>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<float, double, true>), grid, blk, 4 * ls, st, P, ls, accept);
  } else if (io_f64) {
    const int ls = (int)q16_lds<double, double>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<double, double>), grid, blk, 4 * ls, st, P, ls, accept);
  } else if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<float, double>), grid, blk, 4 * ls, st, P, ls, accept);
  } else {
    const int ls = (int)q16_lds<float>(false);
    hipLaunchKernelGGL((q16::lcp_bwd_quad<float, float>), grid, blk, 4 * ls, st, P, ls, accept);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

int quad_step_backward(const StepArgs& SP, const StepBwdArgs& Gd, int compute, void* stream, int body_space) {
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((SP.B + 3) / 4), blk(64);
  const bool wide = 3 * SP.nb > 16;
  if (compute == LCP_COMPUTE_F64) {
    const int ls = (int)q16_lds<double>(false, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 2>), grid, blk, 4 * ls, st, SP, Gd, ls);
    else if (body_space) hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 1, true>), grid, blk, 4 * ls, st, SP, Gd, ls);
    else hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, double, 1>), grid, blk, 4 * ls, st, SP, Gd, ls);
  } else {
    const int ls = (int)q16_lds<float>(false, wide ? 2 : 1);
    if (wide) hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, float, 2>), grid, blk, 4 * ls, st, SP, Gd, ls);
    else hipLaunchKernelGGL((q16::lcp_bwd_step_quad<float, float, 1>), grid, blk, 4 * ls, st, SP, Gd, ls);
  }
  return hipGetLastError() == hipSuccess ? 0 : LCP_E_LAUNCH;
}

}  // namespace lcp
