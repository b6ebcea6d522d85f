#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
template <int... Is, typename F> __device__ __forceinline__ void sf_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sf_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }
#define INL __attribute__((always_inline))
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc_self(double& acc, double mult) {
  if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mult), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(mult), "n"(K));
}
template <int K, bool NOP> __device__ __forceinline__ void fmac_bc(double& acc, double src, double mult) {
  if (NOP) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(K));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mult), "n"(K));
}
// MODE 0: builtin mov+fma (compiler-managed), 1: asm with nop, 2: asm without nop (back-to-back dependent),
// 3: asm without nop where the DPP source was written by a plain VALU op right before
template <int MODE> __global__ void k(double* out, const double* in) {
  const int lane = threadIdx.x;
  double x = in[lane], c = 1e-3 * (lane % 7 + 1), y = in[64 + lane];
  for (int r = 0; r < 4; ++r) {
    sfor<32>([&](auto J) INL {
      constexpr int K = (J * 5 + 3) % 16;
      if (MODE == 0) { const double b = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xf, 0xf, true); x = fma(b, c, x); }
      else if (MODE == 1) fmac_bc_self<K, true>(x, c);
      else if (MODE == 2) fmac_bc_self<K, false>(x, c);
      else if (MODE == 3) { x = x * 1.0000001 + 1e-7; asm volatile("" : "+v"(x)); fmac_bc<K, false>(y, x, c); }
      else if (MODE == 5) {          // 32-bit VALU write of one half of the source right before the 64-bit DPP read
        int lo = __double2loint(x) + 1; asm volatile("v_add_u32 %0, %0, 2" : "+v"(lo)); x = __hiloint2double(__double2hiint(x), lo);
        fmac_bc<K, false>(y, x, c); }
      else if (MODE == 6) {
        int lo = __double2loint(x) + 1; asm volatile("v_add_u32 %0, %0, 2" : "+v"(lo)); x = __hiloint2double(__double2hiint(x), lo);
        const double b = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xf, 0xf, true); y = fma(b, c, y); }
      else if (MODE == 4) { x = x * 1.0000001 + 1e-7; const double b = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + K, 0xf, 0xf, true); y = fma(b, c, y); }
    });
  }
  out[lane] = x + y;
}
int main() {
  double h[128], o[7][64]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + 0.01 * i;
  double *di, *dout; hipMalloc(&di, sizeof(h)); hipMalloc(&dout, 64 * 8); hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
#define RUN(M) k<M><<<1, 64>>>(dout, di); hipMemcpy(o[M], dout, 64 * 8, hipMemcpyDeviceToHost);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  double d1 = 0, d2 = 0, d3 = 0, d5 = 0;
  for (int i = 0; i < 64; ++i) { d1 = fmax(d1, fabs(o[1][i] - o[0][i])); d2 = fmax(d2, fabs(o[2][i] - o[0][i])); d3 = fmax(d3, fabs(o[3][i] - o[4][i])); d5 = fmax(d5, fabs(o[5][i] - o[6][i])); }
  printf("after 32-bit VALU write diff %.3g (ref6 %.15g)\n", d5, o[6][5]);
  printf("ref %.15g | nop-asm diff %.3g | no-nop chain diff %.3g | no-nop after VALU write diff %.3g (ref4 %.15g)\n", o[0][5], d1, d2, d3, o[4][5]);
  return 0;
}
