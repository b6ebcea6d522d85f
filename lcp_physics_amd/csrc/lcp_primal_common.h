// lcp_primal_common.h - what the three translation units of the body-space kernels share (lcp_primal.hip: the step and the
// dense boundary at up to 4 equality rows; lcp_primal_chain.hip: the step with 5 .. 24 equality rows; lcp_primal_poststab.hip:
// post-stabilisation): constants, the workspace layout, small device helpers.
#pragma once
#include "lcp_wave_scene.h"
#include "lcp_quad_prims.h"      // (row_newbcast moves and fused multiply-adds for the lane-grid LU of the pinned form)
#include "lcp_primal_gridlu.h"   // (generated: the trailing update of a pivot of the lane-grid LU as one asm block)
#include "lcp_primal_bsweep.h"   // (generated: the triangular sweeps behind the lane-grid LU, in block layout on three DPP rows)

#ifndef LCP_PRIMAL_OCC40
#define LCP_PRIMAL_OCC40 2     // wavefronts per SIMD the 40-column instantiations are allocated for (A/B: 1 = no scratch, one wave per SIMD)
#endif

#ifndef LCP_PRIMAL_GRIDLU
#define LCP_PRIMAL_GRIDLU 2    // pinned form, <= 32 columns: LU on a 4 x 16 lane grid (pivot rows by row_newbcast) instead of row per lane (v_readlane) - 2: forward and backward kernels (round 4), 1: backward kernels only (round 3), 0: none (A/B)
#endif
#ifndef LCP_PRIMAL_GRIDLU_V2
#define LCP_PRIMAL_GRIDLU_V2 2 // form of the lane-grid LU - 2: fused multiply-adds, multipliers through LDS, constant lane masks, software-pipelined (round 4); 1: the same without the pipelining; 0: round 3's form (v_mov_b64_dpp + v_fma pairs, v_permlane swaps) (A/B)
#endif
#ifndef LCP_PRIMAL_PEEL_INIT
#define LCP_PRIMAL_PEEL_INIT 0     // 1: the initialisation pass (it = -1) as its own copy of the loop body - what gave the four-scenes kernels 4 % (LCP_Q_PEEL_INIT) costs these kernels registers (221 -> 229; the 40-column instantiation spills 37 instead of 25): off
#endif
#ifndef LCP_PRIMAL_RCP_STEP
#define LCP_PRIMAL_RCP_STEP 1      // step lengths, d = z / s and the corrector's rs / s through one reciprocal per z_i, s_i (LCP_Q_RCP_STEP in lcp_quad_kernels.inc; 0: IEEE quotients, A/B)
#endif
#ifndef LCP_PRIMAL_HOIST_DENSE_BWD
#define LCP_PRIMAL_HOIST_DENSE_BWD 1          // dense-boundary backward: the iterate read in the prologue without lane predicates (0: at its uses)
#endif
#ifndef LCP_PRIMAL_EXP_STEP
#define LCP_PRIMAL_EXP_STEP 0     // experiments on the branch of step_pair_rcp (profiles/r06_chain_rootcause.txt): 1 exact form always, 2 fast form always, 3 without __builtin_expect
#endif
#ifndef LCP_PRIMAL_UNROLL_PASS
#define LCP_PRIMAL_UNROLL_PASS 1   // the two KKT solves of an iteration as two copies of the code instead of a two-trip loop (0: the loop - A/B: config 5 forward 0.317 -> 0.313 ms, 239 -> 221 registers; the four-scenes kernels: LCP_Q_UNROLL_PASS)
#endif
#ifndef LCP_PRIMAL_GRIDLU_XROW
#define LCP_PRIMAL_GRIDLU_XROW 0   // lane-grid LU (pipelined form): a pivot's multipliers reach the other DPP rows by lane swaps (1) instead of through LDS (0) (A/B)
#endif
#ifndef LCP_PRIMAL_BSWEEP
#define LCP_PRIMAL_BSWEEP 1    // kernels with the lane-grid LU (second form): triangular sweeps in block layout on three DPP rows - solution entries by row_newbcast inside the multiply-add, blocks across the rows by lane swaps - instead of row per lane with v_readlane broadcasts (0: A/B)
#endif
#ifndef LCP_PRIMAL_CPERM
#define LCP_PRIMAL_CPERM 1     // 1: contact 4 (lane % 16) + lane / 16 on a lane (neighbours in the list -> different 16-lane rows); 0: contact = lane (A/B)
#endif

namespace lcp {
namespace primal {

using namespace w64;
using namespace wsc;

constexpr int LX = 64;           // lanes = stride of the stored iterate
constexpr int EQB = 4;           // padded neq
// workspace per scene (doubles): a 64-entry header (contact count), then the best iterate the backward needs, in the layout
// lcp_big.hip uses, with room for 24 equality multipliers: x[64] y[24] z[4][64] s[4][64] mu[64] diag(Q)[64]
// floats the packed A image needs: e rows of nz <= NCOL - e entries, e <= eqc
constexpr int at_cap(int ncol, int eqc) {
  int m = 0;
  for (int e = 1; e <= eqc; ++e) { const int v = e * (ncol - e); if (v > m) m = v; }
  return m;
}
struct WsLayout { static constexpr int IT = 64, YCAP = 24, ZO = 64 + YCAP, TOTAL = IT + ZO + 10 * LX; };
constexpr int ZO = WsLayout::ZO;     // offset of z in the iterate block

#ifdef LCP_PRIMAL_PROFILE
#define PR_TICK(i) { const long long now_ = clock64(); pc[i] += now_ - tk; tk = now_; }
#else
#define PR_TICK(i)
#endif

// keeps a wave-uniform value in scalar registers at this point (the batches of pivot-row broadcasts stay batches)
__device__ __forceinline__ void sgpr_pin(double& v) { asm volatile("" : "+s"(v)); }
// a double of DPP row R to the same lane of all four rows: copies through v_permlane16_swap / v_permlane32_swap (gfx950)
template <int R> __device__ __forceinline__ double rows_bcast(double x) {
  auto one = [](uint32_t v) -> uint32_t {
    auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);     // -> [v0 v0 v2 v2] and [v1 v1 v3 v3] (by row)
    const uint32_t h = (R & 1) ? s16[1] : s16[0];
    auto s32 = __builtin_amdgcn_permlane32_swap(h, h, false, false);     // -> [h0 h1 h0 h1] and [h2 h3 h2 h3]
    return (R & 2) ? s32[1] : s32[0];
  };
  return __hiloint2double((int)one((uint32_t)__double2hiint(x)), (int)one((uint32_t)__double2loint(x)));
}
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);     // ds_add_f64 (no return)
}

}  // namespace primal
}  // namespace lcp
