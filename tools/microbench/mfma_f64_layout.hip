// Layout discovery for v_mfma_f64_4x4x4_4b_f64 by one-hot probing: for every (la, lb) the A operand is 1 in lane la only,
// the B operand 1 in lane lb only; the D lanes that come out non-zero tell which (block, i, k) / (block, k, j) / (block, i, j)
// each lane holds.  Prints one line per A lane: the B lanes that pair with it and the D lane each pair lands in.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) probe(double* out) {
  const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
  const double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[((size_t)la * 64 + lb) * 64 + lane] = d;
}
int main() {
  double* o; hipMalloc(&o, 64 * 64 * 64 * sizeof(double));
  probe<<<dim3(64, 64), 64>>>(o);
  std::vector<double> h(64 * 64 * 64); hipMemcpy(h.data(), o, h.size() * sizeof(double), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf("  B%2d->D%2d", lb, l);
    printf("\n");
  }
  return 0;
}
