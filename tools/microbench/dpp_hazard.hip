// Which producer -> v_fmac_f64_dpp adjacency is a real hazard on gfx950?  Each case runs the producer and the
// DPP FMA back to back inside one asm block, with and without "s_nop 4" between, and compares.
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPPI "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf"
template <int CASE, bool NOP> __global__ void k(double* out, const double* in, const float* inf_) {
  const int lane = threadIdx.x;
  double acc = in[lane], src = in[64 + lane], mult = in[128 + lane], t = in[192 + lane];
  float f = inf_[lane];
  int sel = lane & 1;
  for (int r = 0; r < 8; ++r) {
    if (CASE == 0) {        // DPP source written by v_mul_f64 right before
      if (NOP) asm volatile("v_mul_f64 %1, %3, %3\n\ts_nop 4\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult), "v"(t));
      else asm volatile("v_mul_f64 %1, %3, %3\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult), "v"(t));
    } else if (CASE == 1) { // multiplier (plain src1) written by v_cvt_f64_f32 right before
      if (NOP) asm volatile("v_cvt_f64_f32 %2, %3\n\ts_nop 4\n\t" DPPI : "+v"(acc) : "v"(src), "v"(mult), "v"(f));
      else asm volatile("v_cvt_f64_f32 %2, %3\n\t" DPPI : "+v"(acc) : "v"(src), "v"(mult), "v"(f));
    } else if (CASE == 2) { // accumulator written by v_mov_b64 right before
      if (NOP) asm volatile("v_mov_b64 %0, %3\n\ts_nop 4\n\t" DPPI : "+v"(acc) : "v"(src), "v"(mult), "v"(t));
      else asm volatile("v_mov_b64 %0, %3\n\t" DPPI : "+v"(acc) : "v"(src), "v"(mult), "v"(t));
    } else if (CASE == 3) { // DPP source written by v_fma_f64 right before
      if (NOP) asm volatile("v_fma_f64 %1, %3, %3, %1\n\ts_nop 4\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult), "v"(t));
      else asm volatile("v_fma_f64 %1, %3, %3, %1\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult), "v"(t));
    } else if (CASE == 4) { // DPP source written by the previous v_fmac_f64_dpp (chain)
      if (NOP) asm volatile("v_fmac_f64_dpp %1, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult));
      else asm volatile("v_fmac_f64_dpp %1, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult));
    } else if (CASE == 5) { // DPP source: written by v_add_f64 right before
      if (NOP) asm volatile("v_add_f64 %1, %1, %2\n\ts_nop 4\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult));
      else asm volatile("v_add_f64 %1, %1, %2\n\t" DPPI : "+v"(acc), "+v"(src) : "v"(mult));
    }
    t = t * 1.01 + 0.003; f = f * 1.01f; mult = mult * 0.99; sel ^= 1;
  }
  out[lane] = acc + src;
}
template <int C> void run(const char* name, double* dout, const double* di, const float* df) {
  double a[64], b[64];
  k<C, true><<<1, 64>>>(dout, di, df); hipMemcpy(a, dout, sizeof(a), hipMemcpyDeviceToHost);
  k<C, false><<<1, 64>>>(dout, di, df); hipMemcpy(b, dout, sizeof(b), hipMemcpyDeviceToHost);
  double d = 0; for (int i = 0; i < 64; ++i) d = fmax(d, fabs(a[i] - b[i]));
  printf("%-70s max diff with/without s_nop: %.3g %s\n", name, d, d == 0 ? "(no hazard)" : "HAZARD");
}
int main() {
  double h[256]; float hf[64];
  for (int i = 0; i < 256; ++i) h[i] = 0.5 + 0.01 * ((i * 37) % 101);
  for (int i = 0; i < 64; ++i) hf[i] = 0.25f + 0.01f * i;
  double *di, *dout; float* df;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, 64 * 8); hipMalloc(&df, sizeof(hf));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(df, hf, sizeof(hf), hipMemcpyHostToDevice);
  run<0>("v_mul_f64 writes the DPP source", dout, di, df);
  run<3>("v_fma_f64 writes the DPP source", dout, di, df);
  run<4>("v_fmac_f64_dpp writes the DPP source (dependent chain)", dout, di, df);
  run<5>("v_add_f64 writes the DPP source", dout, di, df);
  run<1>("v_cvt_f64_f32 writes the plain multiplier", dout, di, df);
  run<2>("v_mov_b64 writes the accumulator", dout, di, df);
  return 0;
}
