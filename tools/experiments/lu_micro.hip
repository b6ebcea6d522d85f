// lu_micro.hip - what bounds the pivot-free LU of lcp_primal_step.inc (one wavefront per scene, row per lane, pivot rows by
// v_readlane) on an MI355X, and what other lane layouts of the same factorisation cost.  Stand-alone (no torch):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lcp_physics_amd/csrc -I include tools/experiments/lu_micro.hip -o tools/experiments/lu_micro
//   tools/experiments/lu_micro            (on the GPU box; prints microseconds per launch and cycles per factorisation)
// Every block factors REPS synthetic diagonally dominant matrices; B blocks of one wavefront, occupancy forced by LDS padding.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "lcp_quad_prims.h"

using namespace lcp;
using namespace lcp::q16;

__device__ __forceinline__ double rl(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ void sgpr_pin(double& v) { asm volatile("" : "+s"(v)); }
__device__ __forceinline__ void vgpr_use(double& v) { asm volatile("" : "+v"(v)); }

// MODE 0: the shipped scheme.  1: pivot-row broadcasts only.  2: FMAs only (scalar operand = a constant).
template <int MODE, int NCOL, int OCC>
__global__ void __launch_bounds__(64, OCC) lu_readlane(double* out, int reps, int n) {
  extern __shared__ double pad[];
  const int lane = threadIdx.x;
  int ln = lane; asm volatile("" : "+v"(ln));
  double t[NCOL];
  double acc = 0, udinv = 1;
  for (int r = 0; r < reps; ++r) {
    static_for<NCOL>([&](auto J) LCP_INL { t[J] = (lane == J ? 8.0 + acc : 0.0) + 0.01 * (double)((lane ^ (J + r)) & 15); });
    static_for<NCOL / 8>([&](auto G8) LCP_INL {
      if (8 * G8 < n) {
        static_for<8>([&](auto KK) LCP_INL {
          constexpr int k = 8 * G8 + KK;
          const double pk = rl(t[k], k);
          const double inv = fast_rcp(pk);
          if (ln == k) udinv = inv;
          const double l = (ln > k) ? t[k] * inv : 0.0;
          if (ln > k) t[k] = l;
          constexpr int NJ = NCOL - 1 - k;
          static_for<(NJ + 7) / 8>([&](auto C8) LCP_INL {
            constexpr int j0 = k + 1 + 8 * C8, nj = (NJ - 8 * C8) < 8 ? (NJ - 8 * C8) : 8;
            double pv[8];
            if constexpr (MODE != 2) {
              static_for<nj>([&](auto I) LCP_INL { pv[I] = rl(t[j0 + I], k); });
              static_for<nj>([&](auto I) LCP_INL { sgpr_pin(pv[I]); });
            } else {
              static_for<nj>([&](auto I) LCP_INL { pv[I] = pk; });
            }
            if constexpr (MODE != 1) static_for<nj>([&](auto I) LCP_INL { t[j0 + I] = fma(-l, pv[I], t[j0 + I]); });
          });
        });
      }
    });
    double sm = udinv;
    static_for<NCOL>([&](auto J) LCP_INL { sm += t[J]; });
    acc = 1e-3 * sm;
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc + (pad ? 0.0 : 1.0);
}


// MODE 4 / 5: the same factorisation with the pivot chain taken off the critical path: column k + 1 is updated first, the next
// pivot is read and its reciprocal started, and only then the rest of the trailing update is issued (5: the Newton steps of the
// reciprocal are spread over the batches by hand).
template <int MODE, int NCOL, int OCC>
__global__ void __launch_bounds__(64, OCC) lu_lookahead(double* out, int reps, int n) {
  extern __shared__ double pad[];
  const int lane = threadIdx.x;
  int ln = lane; asm volatile("" : "+v"(ln));
  double t[NCOL];
  double acc = 0, udinv = 1;
  for (int r = 0; r < reps; ++r) {
    static_for<NCOL>([&](auto J) LCP_INL { t[J] = (lane == J ? 8.0 + acc : 0.0) + 0.01 * (double)((lane ^ (J + r)) & 15); });
    double inv = fast_rcp(rl(t[0], 0));
    static_for<NCOL>([&](auto K) LCP_INL {
      constexpr int k = K;
      if (ln == k) udinv = inv;
      const double l = (ln > k) ? t[k] * inv : 0.0;
      if (ln > k) t[k] = l;
      double pkn = 1.0, rr = 1.0;
      if constexpr (k + 1 < NCOL) {
        const double p1 = rl(t[k + 1], k);
        t[k + 1] = fma(-l, p1, t[k + 1]);
        pkn = rl(t[k + 1], k + 1);
        rr = __builtin_amdgcn_rcp(pkn);
        if constexpr (MODE == 4) { rr = fma(fma(-pkn, rr, 1.0), rr, rr); rr = fma(fma(-pkn, rr, 1.0), rr, rr); }
      }
      constexpr int NJ = NCOL - 2 - k;                                     // columns k + 2 ...
      if constexpr (NJ > 0) {
        static_for<(NJ + 7) / 8>([&](auto C8) LCP_INL {
          constexpr int j0 = k + 2 + 8 * C8, nj = (NJ - 8 * C8) < 8 ? (NJ - 8 * C8) : 8;
          double pv[8];
          static_for<nj>([&](auto I) LCP_INL { pv[I] = rl(t[j0 + I], k); });
          static_for<nj>([&](auto I) LCP_INL { sgpr_pin(pv[I]); });
          static_for<nj>([&](auto I) LCP_INL { t[j0 + I] = fma(-l, pv[I], t[j0 + I]); });
          if constexpr (MODE == 5 && C8 < 2) { rr = fma(fma(-pkn, rr, 1.0), rr, rr); vgpr_use(rr); }
        });
      }
      if constexpr (MODE == 5) {
        constexpr int done = NJ > 0 ? ((NJ + 7) / 8 < 2 ? (NJ + 7) / 8 : 2) : 0;
        static_for<2 - done>([&](auto) LCP_INL { rr = fma(fma(-pkn, rr, 1.0), rr, rr); });
      }
      inv = rr;
    });
    double sm = udinv;
    static_for<NCOL>([&](auto J) LCP_INL { sm += t[J]; });
    acc = 1e-3 * sm;
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc + (pad ? 0.0 : 1.0);
}

// cross-row broadcast of a double from DPP row R to all four rows (same lane of the row): copies + v_permlane16/32_swap
template <int R> __device__ __forceinline__ double rows_bcast(double x) {
  const uint32_t lo = (uint32_t)__double2loint(x), hi = (uint32_t)__double2hiint(x);
  auto one = [&](uint32_t v) -> uint32_t {
    // 16: odd rows of a <-> even rows of b.  a = b = v  ->  a = [v0, v0, v2, v2], b = [v1, v1, v3, v3]
    auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    const uint32_t h = (R & 1) ? s16[1] : s16[0];                       // [vR', vR', vR'', vR''] with R' in {0,1}: R's row of the low pair, ...
    auto s32 = __builtin_amdgcn_permlane32_swap(h, h, false, false);   // rows {2,3} of a <-> rows {0,1} of b
    return (R & 2) ? s32[1] : s32[0];
  };
  return __hiloint2double((int)one(hi), (int)one(lo));
}

// MODE 3: 4 x 16 grid.  lane (r = lane >> 4, l = lane & 15) holds rows l and l + 16 of the columns 4 jj + r (jj < NCOL / 4).
// pivot rows arrive by row_newbcast inside the DPP row, multipliers cross the rows by lane swaps.
template <int NCOL, int OCC, bool ASM>
__global__ void __launch_bounds__(64, OCC) lu_grid(double* out, int reps, int n) {
  extern __shared__ double pad[];
  static_assert(NCOL == 32, "two row slots of sixteen");
  constexpr int NJ = NCOL / 4;
  const int lane = threadIdx.x;
  int l16 = lane & 15, r4 = lane >> 4;
  asm volatile("" : "+v"(l16)); asm volatile("" : "+v"(r4));
  double a0[NJ], a1[NJ];
  double acc = 0, ud0 = 1, ud1 = 1;
  for (int r = 0; r < reps; ++r) {
    static_for<NJ>([&](auto J) LCP_INL {
      const int col = 4 * J + r4;
      a0[J] = (l16 == col ? 8.0 + acc : 0.0) + 0.01 * (double)((l16 ^ (col + r)) & 15);
      a1[J] = (l16 + 16 == col ? 8.0 + acc : 0.0) + 0.01 * (double)(((l16 + 16) ^ (col + r)) & 15);
    });
    static_for<NCOL>([&](auto K) LCP_INL {
      constexpr int k = K, rk = k % 4, jk = k / 4, lk = k % 16, ak = k / 16;
      // pivot (valid in DPP row rk), its reciprocal in every lane of that row
      const double piv = bc<lk>(ak ? a1[jk] : a0[jk]);
      const double inv = fast_rcp(piv);
      // multipliers of the rows below the pivot (valid in DPP row rk), then to all four rows
      double m0 = keep_if(a0[jk] * inv, l16 > k), m1 = keep_if(a1[jk] * inv, l16 + 16 > k);
      if (r4 == rk) { if (l16 == k) ud0 = inv; if (l16 + 16 == k) ud1 = inv; if (l16 > k) a0[jk] = m0; if (l16 + 16 > k) a1[jk] = m1; }
      double b0 = 0, b1;
      if constexpr (k < 15) b0 = rows_bcast<rk>(m0);
      b1 = rows_bcast<rk>(m1);
      // trailing update: my columns to the right of k
      static_for<NJ - jk>([&](auto JJ) LCP_INL {
        constexpr int jj = jk + JJ;
        double u0 = b0, u1 = b1;
        if constexpr (jj == jk) { u0 = keep_if(b0, r4 > rk); u1 = keep_if(b1, r4 > rk); }    // (columns <= k of this slot are finished)
        if constexpr (ak == 0) {
          if constexpr (k < 15) fnmac_bc<lk>(a0[jj], a0[jj], u0);
          fnmac_bc<lk>(a1[jj], a0[jj], u1);
        } else {
          fnmac_bc<lk>(a1[jj], a1[jj], u1);
        }
      });
    });
    double sm = ud0 + ud1;
    static_for<NJ>([&](auto J) LCP_INL { sm += a0[J] + a1[J]; });
    acc = 1e-3 * sm;
  }
  out[(size_t)blockIdx.x * 64 + lane] = acc + (pad ? 0.0 : 1.0);
}

template <typename F> static float time_launch(F&& launch, int iters = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, reps = 10;
  double* out; hipMalloc(&out, sizeof(double) * 64 * (size_t)B);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("B %d, %d factorisations per wavefront, clock %d kHz\n", B, reps, clk);
  auto report = [&](const char* name, float us, int occ) {
    // waves per SIMD slot in sequence: B / (1024 SIMDs * occ)
    const double rounds = (double)B / (1024.0 * occ);
    printf("%-44s occ %d  %8.1f us per launch  = %7.0f cycles per factorisation of one wavefront (its SIMD shared by %d)\n", name, occ, us,
           us * 1e-6 * clk * 1e3 / (rounds * reps), occ);
  };
#define RUN_RL(MODE, NCOL, OCC, N, NAME) { const size_t lds = (OCC == 3) ? 12500 : (OCC == 2) ? 17500 : 38000; \
    report(NAME, time_launch([&] { hipLaunchKernelGGL((lu_readlane<MODE, NCOL, OCC>), dim3(B), dim3(64), lds, 0, out, reps, N); }), OCC); }
#define RUN_LA(MODE, NCOL, OCC, N, NAME) { const size_t lds = (OCC == 3) ? 12500 : (OCC == 2) ? 17500 : 38000; \
    report(NAME, time_launch([&] { hipLaunchKernelGGL((lu_lookahead<MODE, NCOL, OCC>), dim3(B), dim3(64), lds, 0, out, reps, N); }), OCC); }
#define RUN_GR(ASM, OCC, N, NAME) { const size_t lds = (OCC == 2) ? 17500 : 38000; \
    report(NAME, time_launch([&] { hipLaunchKernelGGL((lu_grid<32, OCC, ASM>), dim3(B), dim3(64), lds, 0, out, reps, N); }), OCC); }
  RUN_RL(0, 40, 2, 36, "readlane rows, 40 columns (shipped, n 36)");
  RUN_RL(0, 32, 2, 30, "readlane rows, 32 columns (pinned form, n 30)");
  RUN_RL(0, 24, 2, 24, "readlane rows, 24 columns");
  RUN_RL(1, 32, 2, 30, "  32 columns: broadcasts only");
  RUN_RL(2, 32, 2, 30, "  32 columns: FMAs only");
  RUN_RL(0, 40, 1, 36, "readlane rows, 40 columns");
  RUN_RL(0, 32, 1, 30, "readlane rows, 32 columns");
  RUN_RL(1, 32, 1, 30, "  32 columns: broadcasts only");
  RUN_RL(2, 32, 1, 30, "  32 columns: FMAs only");
  RUN_RL(0, 32, 3, 30, "readlane rows, 32 columns");
  RUN_LA(4, 32, 1, 30, "look-ahead, 32 columns");
  RUN_LA(4, 32, 2, 30, "look-ahead, 32 columns");
  RUN_LA(4, 32, 3, 30, "look-ahead, 32 columns");
  RUN_LA(5, 32, 1, 30, "look-ahead, staged reciprocal, 32 columns");
  RUN_LA(5, 32, 2, 30, "look-ahead, staged reciprocal, 32 columns");
  RUN_LA(5, 32, 3, 30, "look-ahead, staged reciprocal, 32 columns");
  RUN_LA(4, 40, 2, 36, "look-ahead, 40 columns");
  RUN_GR(false, 2, 30, "4 x 16 grid, 32 columns (DPP + lane swaps)");
  RUN_GR(false, 1, 30, "4 x 16 grid, 32 columns (DPP + lane swaps)");
  std::vector<double> h(64);
  hipMemcpy(h.data(), out, sizeof(double) * 64, hipMemcpyDeviceToHost);
  printf("checksum %g\n", h[0]);
  return 0;
}
