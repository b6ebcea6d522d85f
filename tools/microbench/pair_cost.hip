// Micro-benchmark: cost of the LU inner "pair" (v_readlane x2 + v_fma_f64) and friends on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 pair_cost.hip -o pair_cost ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>
template <int... Is, typename F> __device__ __forceinline__ void sf_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sf_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }
#define INL __attribute__((always_inline))
__device__ __forceinline__ double rdl(double v, int s) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), s), __builtin_amdgcn_readlane(__double2loint(v), s));
}
__device__ __forceinline__ float rdl(float v, int s) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), s)); }

// acc += bcast_K(src) * mult with the broadcast folded into the FMA (64-bit DPP of the DP ALU, row_newbcast only)
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc(double& acc, double src, double mult) {
  if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(K));
}
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc_self(double& acc, double mult) {
  if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mult), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mult), "n"(K));
}
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc(float&, float, float) {}
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc_self(float&, float) {}

// MODE 0: pair f64 (2 readlane + fma)   1: pair f32   2: fma f64 only (VGPR operands)   3: fma f32 only
// MODE 4: 16x16x4 f64 MFMA chain (4 independent accumulators)   5: 16x16x4 f32 MFMA
template <typename T, int MODE, int NP>
__global__ void __launch_bounds__(64) k(T* out, long long* cyc, int reps, int pl) {
  const int lane = threadIdx.x;
  T t[NP];
  sfor<NP>([&](auto J) INL { t[J] = (T)(lane + J) * (T)1e-3; });
  T l = (T)1e-9 * lane, l2 = (T)2e-9 * lane;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    const int p = (pl + r) & 63;
    if (MODE == 0 || MODE == 1) {
      sfor<NP>([&](auto J) INL { t[J] = fma(-l, rdl(t[J], p), t[J]); });
    } else if (MODE == 6) {          // broadcast through ds_bpermute (VGPR result, no SGPR hazard)
      sfor<NP>([&](auto J) INL { t[J] = fma(-l, __shfl(t[J], p, 64), t[J]); });
    } else if (MODE == 10) {         // DPP row_newbcast (lane (r&15) of every 16-lane row -> that row) + 2 FMAs
      sfor<NP / 2>([&](auto J) INL {
        constexpr int ctrl = 0x150 + (J % 16);
        T sj;
        if constexpr (sizeof(T) == 8) {
          const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(t[J]), ctrl, 0xf, 0xf, true);
          const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(t[J]), ctrl, 0xf, 0xf, true);
          sj = __hiloint2double(hi, lo);
        } else {
          sj = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t[J]), ctrl, 0xf, 0xf, true));
        }
        t[J] = fma(-l, sj, t[J]);
        t[J + NP / 2] = fma(-l2, sj, t[J + NP / 2]);
      });
    } else if (MODE == 11) {         // one v_mov_b64_dpp row_newbcast (64-bit DPP of the DP ALU) + 2 FMAs
      if constexpr (sizeof(T) == 8) {
        sfor<NP / 2>([&](auto J) INL {
          constexpr int ctrl = 0x150 + (J % 16);
          const T sj = __builtin_amdgcn_update_dpp((T)0, t[J], ctrl, 0xf, 0xf, true);
          t[J] = fma(-l, sj, t[J]);
          t[J + NP / 2] = fma(-l2, sj, t[J + NP / 2]);
        });
      }
    } else if (MODE == 12) {         // broadcast folded into the FMA: v_fmac_f64_dpp row_newbcast (no mov at all)
      if constexpr (sizeof(T) == 8) {
        const T nl = -l, nl2 = -l2;
        sfor<NP / 2>([&](auto J) INL {
          constexpr int K = J % 16;
          fmac_bc<K, false>(t[J + NP / 2], t[J], nl2);
          fmac_bc_self<K, false>(t[J], nl);
        });
      }
    } else if (MODE == 13) {         // as 12, with the 2 wait states a freshly written DPP source needs (s_nop 1) in front of each
      if constexpr (sizeof(T) == 8) {
        const T nl = -l, nl2 = -l2;
        sfor<NP / 2>([&](auto J) INL {
          constexpr int K = J % 16;
          fmac_bc<K, true>(t[J + NP / 2], t[J], nl2);
          fmac_bc_self<K, true>(t[J], nl);
        });
      }
    } else if (MODE == 14) {         // dependent chain: x += bcast_k(x) * c   (the triangular-solve pattern), builtin mov + fma
      if constexpr (sizeof(T) == 8) {
        sfor<NP>([&](auto J) INL {
          constexpr int ctrl = 0x150 + (J % 16);
          const T sj = __builtin_amdgcn_update_dpp((T)0, t[0], ctrl, 0xf, 0xf, true);
          t[0] = fma(-l, sj, t[0]);
        });
      }
    } else if (MODE == 15) {         // dependent chain with v_fmac_f64_dpp (+ s_nop 1 for the DPP read-after-write hazard)
      if constexpr (sizeof(T) == 8) {
        const T nl = -l;
        sfor<NP>([&](auto J) INL {
          constexpr int K = J % 16;
          fmac_bc_self<K, true>(t[0], nl);
        });
      }
    } else if (MODE == 16) {         // dependent chain with v_fmac_f64_dpp, no s_nop
      if constexpr (sizeof(T) == 8) {
        const T nl = -l;
        sfor<NP>([&](auto J) INL { fmac_bc_self<J % 16, false>(t[0], nl); });
      }
    } else if (MODE == 7 || MODE == 8 || MODE == 9) {   // batched: G readlane pairs first, then G fmas
      constexpr int G = (MODE == 7) ? 4 : (MODE == 8) ? 8 : 16;
      sfor<NP / G>([&](auto B) INL {
        T sv[G];
        sfor<G>([&](auto I) INL { sv[I] = rdl(t[B * G + I], p); });
        __builtin_amdgcn_sched_barrier(0);
        sfor<G>([&](auto I) INL { t[B * G + I] = fma(-l, sv[I], t[B * G + I]); });
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      sfor<NP>([&](auto J) INL { t[J] = fma(-l, t[(J + 1) % NP], t[J]); });
    }
  }
  long long t1 = clock64();
  T s = 0;
  sfor<NP>([&](auto J) INL { s += t[J]; });
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int F64>
__global__ void __launch_bounds__(64) kmfma(double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x;
  d4 acc[4]; f4 accf[4];
  for (int i = 0; i < 4; ++i) { acc[i] = (d4){0, 0, 0, 0}; accf[i] = (f4){0, 0, 0, 0}; }
  double a = 1e-3 * lane, b = 2e-3 * lane;
  float af = a, bf = b;
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (F64) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else accf[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, accf[i], 0, 0, 0);
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + accf[i][0] + accf[i][1] + accf[i][2] + accf[i][3];
  out[blockIdx.x * 64 + lane] = s;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename K, typename... A>
static void run(const char* name, K kern, int blocks, int reps, double per_rep, A... args) {
  long long* cyc; hipMalloc(&cyc, blocks * sizeof(long long));
  kern<<<blocks, 64>>>(args..., cyc, reps, 3);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<<<blocks, 64>>>(args..., cyc, reps, 5);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-34s blocks %5d  %.2f clock64-ticks/item  (wall %.3f ms)\n", name, blocks, avg / (reps * per_rep), ms);
  hipFree(cyc);
}
template <typename K, typename... A>
static void run2(const char* name, K kern, int blocks, int reps, double per_rep, A... args) {
  long long* cyc; hipMalloc(&cyc, blocks * sizeof(long long));
  kern<<<blocks, 64>>>(args..., cyc, reps);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  kern<<<blocks, 64>>>(args..., cyc, reps);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-34s blocks %5d  %.2f clock64-ticks/item  (wall %.3f ms)\n", name, blocks, avg / (reps * per_rep), ms);
  hipFree(cyc);
}

int main() {
  double* o; hipMalloc(&o, 8192 * 64 * 8);
  float* of = (float*)o;
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("device %s clockRate %d kHz  (clock64 may tick at a fixed 100 MHz: compare wall time too)\n", pr.name, pr.clockRate);
  const int reps = 2000;
  for (int blocks : {1024, 2048}) {
    run("pair f64 (2 rdlane+fma) x48", k<double, 0, 48>, blocks, reps, 48, o);
    run("pair f32 (1 rdlane+fma) x48", k<float, 1, 48>, blocks, reps, 48, of);
    run("pair f64 via ds_bpermute x48", k<double, 6, 48>, blocks, reps, 48, o);
    run("pair f32 via ds_bpermute x48", k<float, 6, 48>, blocks, reps, 48, of);
    run("pair f64 batched 4", k<double, 7, 48>, blocks, reps, 48, o);
    run("pair f64 batched 8", k<double, 8, 48>, blocks, reps, 48, o);
    run("pair f64 batched 16", k<double, 9, 48>, blocks, reps, 48, o);
    run("pair f32 batched 16", k<float, 9, 48>, blocks, reps, 48, of);
    run("dpp newbcast f64: 24 bcast + 48 fma", k<double, 10, 48>, blocks, reps, 48, o);
    run("dpp newbcast f32: 24 bcast + 48 fma", k<float, 10, 48>, blocks, reps, 48, of);
    run("dpp b64 mov: 24 bcast + 48 fma", k<double, 11, 48>, blocks, reps, 48, o);
    run("v_fmac_f64_dpp newbcast x48", k<double, 12, 48>, blocks, reps, 48, o);
    run("v_fmac_f64_dpp + s_nop 1 x48", k<double, 13, 48>, blocks, reps, 48, o);
    run("chain: b64 dpp mov + fma x48", k<double, 14, 48>, blocks, reps, 48, o);
    run("chain: s_nop 1 + v_fmac_f64_dpp x48", k<double, 15, 48>, blocks, reps, 48, o);
    run("chain: v_fmac_f64_dpp (no nop) x48", k<double, 16, 48>, blocks, reps, 48, o);
    run("fma f64 only x48", k<double, 2, 48>, blocks, reps, 48, o);
    run("fma f32 only x48", k<float, 3, 48>, blocks, reps, 48, of);
    run2("mfma f64 16x16x4 x4", kmfma<1>, blocks, reps, 4, o);
    run2("mfma f32 16x16x4 x4", kmfma<0>, blocks, reps, 4, o);
  }
  return 0;
}
