// Microbenchmark: cost of building per-sample histograms of the counts with global atomics (3 per cell),
// layout hist[q][v][S], cells [E][S] sample-minor.  hipcc --offload-arch=gfx950 -O3 tools/ubench_hist.hip -o /tmp/ubh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void k_hist(const int* __restrict__ test, const int* __restrict__ ref, long E, long S, int K, unsigned* __restrict__ hist)
{
  const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= E * S) return;
  const long s = cell % S;
  const int y = test[cell], r = ref[cell], n = y + r;
  if (n <= 0) return;
  if (y < K && r < K && n < K) {
    if (MODE == 0) {
      atomicAdd(&hist[((long)0 * K + y) * S + s], 1u);
      atomicAdd(&hist[((long)1 * K + r) * S + s], 1u);
      atomicAdd(&hist[((long)2 * K + n) * S + s], 1u);
    } else {
      __hip_atomic_fetch_add(&hist[((long)0 * K + y) * S + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&hist[((long)1 * K + r) * S + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&hist[((long)2 * K + n) * S + s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

int main()
{
  const long E = 200000, S = 1024; const int K = 4096;
  std::vector<int> t(E * S), r(E * S);
  std::mt19937 g(1);
  for (long e = 0; e < E; ++e) {
    std::lognormal_distribution<double> ld(4.6, 0.8);
    const double lam = ld(g);
    std::poisson_distribution<int> py(lam), pr(8 * lam);
    for (long s = 0; s < S; s += 64) {   // cheap: one draw per 64 samples + jitter
      const int y0 = py(g), r0 = pr(g);
      for (long k = 0; k < 64; ++k) { t[e * S + s + k] = y0 + (int)((g() >> 8) % 21) ; r[e * S + s + k] = r0 + (int)((g() >> 8) % 61); }
    }
  }
  int *dt, *dr; unsigned* dh;
  CK(hipMalloc(&dt, E * S * 4)); CK(hipMalloc(&dr, E * S * 4)); CK(hipMalloc(&dh, 3L * K * S * 4));
  CK(hipMemcpy(dt, t.data(), E * S * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, r.data(), E * S * 4, hipMemcpyHostToDevice));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode) {
    for (int it = 0; it < 3; ++it) {
      CK(hipMemset(dh, 0, 3L * K * S * 4));
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k_hist<0>, dim3((E * S + 255) / 256), dim3(256), 0, 0, dt, dr, E, S, K, dh);
      else hipLaunchKernelGGL(k_hist<1>, dim3((E * S + 255) / 256), dim3(256), 0, 0, dt, dr, E, S, K, dh);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("mode %d (scope %s): %.3f ms\n", mode, mode ? "workgroup" : "agent", ms);
    }
  }
  std::vector<unsigned> h(3L * K * S);
  CK(hipMemcpy(h.data(), dh, h.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long tot = 0; for (auto v : h) tot += v;
  printf("total counted %llu (expect %ld)\n", tot, 3 * E * S);
  return 0;
}
