// Micro-benchmark: v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 issue rate on gfx950 against the number of
// independent accumulators and waves per SIMD, plus a layout check of the operand / result maps the blocked LU of
// lcp_big.hip relies on:
//   A operand : lane l holds A[i = l & 15][k = l >> 4]            (one f64 per lane per K-chunk of 4)
//   B operand : lane l holds B[k = l >> 4][j = l & 15]
//   C / D     : 4 f64 per lane, reg r holds D[row = (l >> 4) + 4 r][col = l & 15]
// consequence used by the LU: register r of a D tile IS chunk r of a B operand (B = D), and as chunk r of an A
// operand it is the transpose (A = D^T).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_f64.hip -o mfma_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ void __launch_bounds__(256) kthr(double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x & 63;
  d4 acc[NACC];
  double acc1[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = (d4){0, 0, 0, 0}; acc1[i] = 0; }
  double a = 1e-3 * lane, b = 2e-3 * lane;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else if (KIND == 1) acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
      else { acc[i][0] = fma(a, b, acc[i][0]); acc[i][1] = fma(a, b, acc[i][1]); acc[i][2] = fma(a, b, acc[i][2]); acc[i][3] = fma(a, b, acc[i][3]); }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND>
static void run(const char* name, int blocks, int threads, int reps, double flops_per_inst) {
  double* out; long long* cyc;
  hipMalloc(&out, (size_t)blocks * threads * sizeof(double));
  hipMalloc(&cyc, blocks * sizeof(long long));
  kthr<NACC, KIND><<<blocks, threads>>>(out, cyc, 8);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kthr<NACC, KIND><<<blocks, threads>>>(out, cyc, reps);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  const double waves = (double)blocks * threads / 64.0;
  const double insts = (double)reps * NACC * (KIND == 2 ? 4 : 1);
  const double tf = waves * insts * flops_per_inst / (ms * 1e-3) / 1e12;
  printf("%-40s acc %2d  blocks %5d x %3d  %8.2f ticks/inst   wall %.3f ms   %.1f TFLOP/s chip\n", name, NACC, blocks, threads,
         avg / insts, ms, tf);
  hipFree(out); hipFree(cyc);
}

// ---- layout check: D = A(16x16) . B(16x16) through 4 chained MFMAs, then E = D-as-B, Ft = D-as-A -------------------
__global__ void __launch_bounds__(64) klayout(const double* A, const double* B, double* D, double* E, double* F) {
  const int l = threadIdx.x, g = l >> 4, c = l & 15;
  d4 acc = (d4){0, 0, 0, 0};
  for (int ch = 0; ch < 4; ++ch) {
    const double a = A[c * 16 + (4 * ch + g)];          // A[i = c][k = 4 ch + g]
    const double b = B[(4 * ch + g) * 16 + c];          // B[k = 4 ch + g][j = c]
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + c] = acc[r];
  // E = A . D   with D used directly as the B operand (chunk r = register r)
  d4 e = (d4){0, 0, 0, 0};
  for (int ch = 0; ch < 4; ++ch) e = __builtin_amdgcn_mfma_f64_16x16x4f64(A[c * 16 + (4 * ch + g)], acc[ch], e, 0, 0, 0);
  for (int r = 0; r < 4; ++r) E[(g + 4 * r) * 16 + c] = e[r];
  // F = D^T . B with D used directly as the A operand
  d4 f = (d4){0, 0, 0, 0};
  for (int ch = 0; ch < 4; ++ch) f = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[ch], B[(4 * ch + g) * 16 + c], f, 0, 0, 0);
  for (int r = 0; r < 4; ++r) F[(g + 4 * r) * 16 + c] = f[r];
}

static void layout_check() {
  std::vector<double> A(256), B(256), D(256), E(256), F(256), Dr(256, 0), Er(256, 0), Fr(256, 0);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { A[i * 16 + j] = 1.0 + 0.37 * i - 0.11 * j + 0.01 * i * j; B[i * 16 + j] = 0.5 - 0.21 * i + 0.13 * j * j; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) Dr[i * 16 + j] += A[i * 16 + k] * B[k * 16 + j];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 16; ++k) { Er[i * 16 + j] += A[i * 16 + k] * Dr[k * 16 + j]; Fr[i * 16 + j] += Dr[k * 16 + i] * B[k * 16 + j]; }
  double *dA, *dB, *dD, *dE, *dF;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 2048); hipMalloc(&dE, 2048); hipMalloc(&dF, 2048);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  klayout<<<1, 64>>>(dA, dB, dD, dE, dF);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(E.data(), dE, 2048, hipMemcpyDeviceToHost); hipMemcpy(F.data(), dF, 2048, hipMemcpyDeviceToHost);
  double ed = 0, ee = 0, ef = 0, sc = 0;
  for (int i = 0; i < 256; ++i) { ed = fmax(ed, fabs(D[i] - Dr[i])); ee = fmax(ee, fabs(E[i] - Er[i])); ef = fmax(ef, fabs(F[i] - Fr[i])); sc = fmax(sc, fabs(Fr[i])); }
  printf("layout check: |D - A.B| %.3e   |E - A.D (D as B operand)| %.3e   |F - D^T.B (D as A operand)| %.3e   (scale %.3e)\n", ed, ee, ef, sc);
}

int main() {
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s  CUs %d  clockRate %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
  layout_check();
  const int reps = 2000;
  // one wave per SIMD (1024 waves), 2, 4
  run<1, 0>("mfma f64 16x16x4 dependent", 1024, 64, reps, 2048);
  run<2, 0>("mfma f64 16x16x4", 1024, 64, reps, 2048);
  run<4, 0>("mfma f64 16x16x4", 1024, 64, reps, 2048);
  run<8, 0>("mfma f64 16x16x4", 1024, 64, reps, 2048);
  run<16, 0>("mfma f64 16x16x4", 1024, 64, reps, 2048);
  run<4, 0>("mfma f64 16x16x4 (2 waves/SIMD)", 2048, 64, reps, 2048);
  run<8, 0>("mfma f64 16x16x4 (2 waves/SIMD)", 2048, 64, reps, 2048);
  run<4, 0>("mfma f64 16x16x4 (4 waves/SIMD)", 4096, 64, reps, 2048);
  run<8, 0>("mfma f64 16x16x4 (256 thr, 1 blk/CU)", 256, 256, reps, 2048);
  run<16, 0>("mfma f64 16x16x4 (256 thr, 1 blk/CU)", 256, 256, reps, 2048);
  run<8, 0>("mfma f64 16x16x4 (256 thr, 2 blk/CU)", 512, 256, reps, 2048);
  run<1, 1>("mfma f64 4x4x4_4b dependent", 1024, 64, reps, 512);
  run<4, 1>("mfma f64 4x4x4_4b", 1024, 64, reps, 512);
  run<8, 1>("mfma f64 4x4x4_4b", 1024, 64, reps, 512);
  run<16, 1>("mfma f64 4x4x4_4b", 1024, 64, reps, 512);
  run<8, 1>("mfma f64 4x4x4_4b (2 waves/SIMD)", 2048, 64, reps, 512);
  run<4, 2>("v_fma_f64 x4 per acc", 1024, 64, reps, 128);
  run<8, 2>("v_fma_f64 x4 per acc", 1024, 64, reps, 128);
  run<8, 2>("v_fma_f64 x4 per acc (2 waves/SIMD)", 2048, 64, reps, 128);
  return 0;
}
