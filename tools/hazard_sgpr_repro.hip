// hazard_sgpr_repro.hip -- reproducer of the hardware hazard behind round 1's one unexplained emission mismatch.
//
// gfx940 / gfx950: a VALU instruction that WRITES an SGPR (v_readlane_b32, v_readfirstlane_b32, v_cmp ... to an SGPR pair)
// must be followed by 2 wait states before a VALU instruction READS that SGPR as an operand.  LLVM's hazard recogniser
// inserts the s_nop for instructions it generates itself (GCNHazardRecognizer::checkVALUHazards, VALUWriteSGPRVALURead
// = 2) but it cannot see inside an inline-asm statement.  ed_pmath.h's Horner step used to be
//        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k))
// and the register allocator, short of SGPRs, parks such constants in VGPR lanes and brings them back with
// v_readlane_b32 right in front of the asm (tools/isa_hazard_scan.py finds those places in the compiler's output).
// This program issues exactly that sequence by hand:
//        s[40:41] = K_old (SALU);  v_readlane_b32 s41 <- high half of K_new (VALU);  s_mov_b32 s40 <- low half (SALU);
//        v_fma_f64 d, a, b, s[40:41]
// with 0, 1 or 2 wait states of padding before the fma, and counts how often d != fma(a, b, K_new).
//   hipcc --offload-arch=gfx950 -O2 tools/hazard_sgpr_repro.hip -o /tmp/hazard && /tmp/hazard
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

template <int PAD>
__global__ void k_probe(const double* __restrict__ a_in, const double* __restrict__ b_in, int iters, unsigned long long* __restrict__ wrong,
                        unsigned long long* __restrict__ stale)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double a = a_in[i], b = b_in[i];
  unsigned long long nw = 0, ns = 0;
  for (int it = 0; it < iters; ++it) {
    double d;
    int hi_new = 0x3fa55555;      // high half of 1/24 = 0x3fa5555555555555
    asm volatile(
        "s_mov_b32 s40, 0x11111111\n\t"
        "s_mov_b32 s41, 0x3f811111\n\t"          // s[40:41] = 1/120 (the previous Horner coefficient)
        "s_nop 7\n\t"
        "v_readlane_b32 s41, %[hi], 0\n\t"        // VALU writes s41: the spill reload
        "s_mov_b32 s40, 0x55555555\n\t"           // SALU: 1 wait state, as in the compiler's output
        ".if %[pad] == 1\n\ts_nop 0\n\t.endif\n\t"
        ".if %[pad] == 2\n\ts_nop 1\n\t.endif\n\t"
        "v_fma_f64 %[d], %[a], %[b], s[40:41]\n\t"
        : [d] "=v"(d)
        : [a] "v"(a), [b] "v"(b), [hi] "v"(hi_new), [pad] "n"(PAD)
        : "s40", "s41");
    const double want = __builtin_fma(a, b, 0x1.5555555555555p-5);                  // 1/24
    const double with_stale_hi = __builtin_fma(a, b, __longlong_as_double(0x3f81111155555555ll));
    if (__double_as_longlong(d) != __double_as_longlong(want)) {
      ++nw;
      if (__double_as_longlong(d) == __double_as_longlong(with_stale_hi)) ++ns;
    }
    a = a * 1.0000001 + 1e-9;     // new operands every iteration
  }
  if (nw) atomicAdd(wrong, nw);
  if (ns) atomicAdd(stale, ns);
}

template <int PAD>
static void run(int blocks, int threads, int iters, const char* label)
{
  const int n = blocks * threads;
  std::vector<double> a(n), b(n);
  for (int i = 0; i < n; ++i) { a[i] = 1e-3 * (1 + i % 97); b[i] = 0.5 + 1e-4 * (i % 89); }
  double *da, *db;
  unsigned long long *dw, *ds, w = 0, s = 0;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dw, 8); hipMalloc(&ds, 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipMemset(dw, 0, 8); hipMemset(ds, 0, 8);
  hipLaunchKernelGGL(k_probe<PAD>, dim3(blocks), dim3(threads), 0, 0, da, db, iters, dw, ds);
  hipDeviceSynchronize();
  hipMemcpy(&w, dw, 8, hipMemcpyDeviceToHost); hipMemcpy(&s, ds, 8, hipMemcpyDeviceToHost);
  printf("{\"padding_wait_states\": %d, \"launch\": \"%s\", \"fma_evaluations\": %lld, \"wrong\": %llu, \"wrong_equal_to_stale_high_half\": %llu}\n",
         PAD, label, (long long)n * iters, w, s);
  hipFree(da); hipFree(db); hipFree(dw); hipFree(ds);
}

int main()
{
  // one wave per SIMD (back-to-back issue from the same wave is the rule), then a full machine
  run<0>(1024, 64, 20000, "1024 x 64 threads");
  run<1>(1024, 64, 20000, "1024 x 64 threads");
  run<2>(1024, 64, 20000, "1024 x 64 threads");
  run<0>(8192, 256, 5000, "8192 x 256 threads");
  run<1>(8192, 256, 5000, "8192 x 256 threads");
  run<2>(8192, 256, 5000, "8192 x 256 threads");
  return 0;
}
