// mfma_form.hip - round 6, VERDICT r05 item 3: what the formation S = Q + J^T (B J) of the headline forward would cost on
// v_mfma_f64_4x4x4_4b_f64 (16 lanes = one scene = one 4 x 4 block of the four-blocks MFMA, 12 x 12 outputs = 9 blocks, K = 32 = 8 steps:
// 72 MFMAs), next to the 384 v_fmac_f64 it costs today - and whether the matrix pipe runs BESIDE the vector pipe of the same wavefront.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_form.hip -o tools/microbench/mfma_form && tools/microbench/mfma_form
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64

template <int MODE>
__global__ void __launch_bounds__(64) kern(double* out, long long* cyc, int n) {
  __shared__ double lds[4][160];
  const int lane = threadIdx.x, l16 = lane & 15, row = lane >> 4;
  double acc[9], v[12], a[8], b[8];
  for (int i = 0; i < 9; ++i) acc[i] = 0.0;
  for (int i = 0; i < 12; ++i) v[i] = 1.0 + lane * 1e-3 + i;
  for (int i = 0; i < 8; ++i) { a[i] = 1.0 + 0.01 * (lane + i); b[i] = 0.5 - 0.01 * (lane - i); }
  const double w0 = 1.0000001, w1 = 0.9999999;
  __syncthreads();
  const long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < n; ++r) {
    if constexpr (MODE == 0 || MODE == 2) {
      // 72 MFMAs: 9 accumulators x 8 K-steps (operands recycled: the microbench measures issue, not values)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[(k + i) & 7], b[k], acc[i], 0, 0, 0);
          if constexpr (MODE == 2) {
            // five to six independent vector FMAs per MFMA (384 in all): do they issue while the matrix pipe works ?
#pragma unroll
            for (int q = 0; q < 5; ++q) { const int j = (6 * i + q + k) % 12; v[j] = __builtin_fma(v[j], w0, w1); }
            if ((k * 9 + i) % 3 == 0) { const int j = (i + k) % 12; v[j] = __builtin_fma(v[j], w1, w0); }
          }
        }
      }
    }
    if constexpr (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 384; ++q) { const int j = q % 12; v[j] = __builtin_fma(v[j], (q & 1) ? w0 : w1, (q & 1) ? w1 : w0); }
    }
    if constexpr (MODE == 3) {
      // layout change of the result: the 9 blocks of a lane (element (4 I + i, 4 J + j) at lane 4 i + j, register (I, J)) through LDS into
      // row-per-lane form (lane r holds row r: 12 doubles) - what the LU and the sweeps of lcp_fwd_quad work on
#pragma unroll
      for (int I = 0; I < 3; ++I)
#pragma unroll
        for (int J = 0; J < 3; ++J) lds[row][(4 * I + (l16 >> 2)) * 13 + 4 * J + (l16 & 3)] = acc[3 * I + J] + r;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int c = 0; c < 12; ++c) v[c] += lds[row][(l16 < 12 ? l16 : 0) * 13 + c];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i];
  for (int i = 0; i < 12; ++i) s += v[i];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> static void run(const char* what, int blocks) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 64); hipMalloc(&cyc, sizeof(long long) * blocks);
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, REP);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, REP);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double m = 0; for (auto c : h) m += (double)c; m /= blocks;
  printf("%-78s blocks %5d  %9.1f cycles per formation-equivalent\n", what, blocks, m / REP);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1024, 2048}) {
    run<0>("72 x v_mfma_f64_4x4x4_4b (9 accumulators x 8 K-steps)", blocks);
    run<1>("384 x v_fma_f64 (today's formation: 16 contacts x 12 columns x 2)", blocks);
    run<2>("72 MFMAs with 384 independent v_fma_f64 between them (same wavefront)", blocks);
    run<3>("9 result blocks -> LDS -> row per lane (12 doubles)", blocks);
  }
  return 0;
}
