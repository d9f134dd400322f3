#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int K> __device__ __forceinline__ double qb(double x) {
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), K * 0x55, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void k(int mode, int n, double* out, long long* cyc) {
  double v = threadIdx.x * 1e-3, a = 1.000001, b = -0.5;
  long long t0 = __builtin_readcyclecounter();
  if (mode == 0) { for (int i = 0; i < n; ++i) { v = v + a; } }                       // dependent add
  else if (mode == 1) { for (int i = 0; i < n; ++i) { v = __builtin_fmax(v + a, b); } } // add + max
  else if (mode == 2) { for (int i = 0; i < n; ++i) { v = qb<1>(v) + a; } }             // dpp + add
  else if (mode == 3) { for (int i = 0; i < n; ++i) { double c = v + a; v = (c > b) ? c : b; } } // add cmp sel
  else if (mode == 4) { for (int i = 0; i < n; ++i) {                                   // full viterbi-like step
      double v0 = qb<0>(v), v1 = qb<1>(v), v2 = qb<2>(v);
      double c0 = (a + v0) + b, c1 = (a + v1) + b * 1.1, c2 = (a + v2) + b * 1.2;
      v = __builtin_fmax(__builtin_fmax(__builtin_fmax(-HUGE_VAL, c0), c1), c2); } }
  else if (mode == 5) { double w = v; for (int i = 0; i < n; ++i) { v = v + a; w = w + b; } v += w; } // 2 independent chains
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[mode] = t1 - t0;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64);
  const int n = 20000;
  for (int blocks : {1, 64, 256, 512, 1024, 2048, 4096}) {
    for (int mode = 4; mode < 5; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, mode, n, out, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, mode, n, out, cyc); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc + mode, 8, hipMemcpyDeviceToHost);
      printf("blocks %4d mode %d: %.1f ns/iter  (s_memtime ticks/iter %.2f)\n", blocks, mode, ms * 1e6 / n, (double)c / n);
    }
  }
  return 0;
}
