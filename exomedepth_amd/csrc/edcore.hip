// edcore.hip -- kernels and C-ABI of libedcore.so (gfx950 / MI355X).  See include/exomedepth_amd.h.
//
// Kernels (names follow the reference's domain: exons, samples, chains = (sample, chromosome)):
//   k_sample_consts   per sample: the three (a1, a2, lnbeta(a1,a2)) triples of myprob (src/CNV_estimate.cpp:44-50)
//   k_emit_batch      per (exon, sample) cell: three beta-binomial log-likelihoods (src/CNV_estimate.cpp:71-81)
//   k_emit_rows       the same for per-exon phi/expected (the reference's .Call signature)
//   k_viterbi         per chain: forward max-plus pass with back-pointers, trace-back, call count
//                     (src/hmm.cpp:58-100, :104-126)
//   k_scan_counts     exclusive scan of the per-chain call counts
//   k_calls_fill      per chain: writes the call records (src/hmm.cpp:104-126, R/class_definition.R:371-372,:409-410)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contraction off is part of the contract:
// the arithmetic must match the CPU checker bit for bit).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/exomedepth_amd.h"
#include "ed_sf_dev.hpp"

#define ED_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int ed_fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                                         \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess) return ed_fail(ED_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                         __FILE__, __LINE__);                                                 \
  } while (0)

static int require_device()
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return ed_fail(ED_ERR_NO_DEVICE, "no usable HIP device (hipGetDeviceCount: %s, count=%d); this library has no CPU path",
                   hipGetErrorString(e), n);
  return ED_OK;
}

// ------------------------------------------------------------------------------------------
// device code
// ------------------------------------------------------------------------------------------
namespace {

constexpr int kEmitBlock = 256;
constexpr int kWave = 64;

// myprob's shape parameters for one state (src/CNV_estimate.cpp:45-46)
__device__ __forceinline__ void shape_params(double ep, double sd, double& a1, double& a2)
{
  a1 = ((ep * ep) * (1 - ep)) / (sd * sd) - ep;
  a2 = ((1 - ep) / ep) * a1;
}

// the three per-state expected proportions (src/CNV_estimate.cpp:65-66, :75-77)
__device__ __forceinline__ void state_props(double e, double mixture, double ep[3])
{
  const double odds_del = 1 - 0.5 * mixture;
  const double odds_dup = 1 + 0.5 * mixture;
  ep[0] = (e * odds_del) / ((e * odds_del + 1) - e);
  ep[1] = e;
  ep[2] = (e * odds_dup) / ((e * odds_dup + 1) - e);
}

// consts layout: [9][S] = a1_del,a2_del,C_del, a1_norm,a2_norm,C_norm, a1_dup,a2_dup,C_dup ; flags[3][S]
__global__ void k_sample_consts(const double* __restrict__ phi, const double* __restrict__ expected, double mixture,
                                int64_t S, double* __restrict__ consts, int* __restrict__ cflags)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const double e = expected[s];
  const double sd = __builtin_sqrt((phi[s] * e) * (1. - e));
  double ep[3];
  state_props(e, mixture, ep);
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    double a1, a2;
    shape_params(ep[st], sd, a1, a2);
    int flag = 0;
    const double c = edsf::lnbeta(a1, a2, &flag);
    consts[(st * 3 + 0) * S + s] = a1;
    consts[(st * 3 + 1) * S + s] = a2;
    consts[(st * 3 + 2) * S + s] = c;
    cflags[st * S + s] = flag;
  }
}

// one thread per (exon, sample) cell; cells are numbered exon-major / sample-minor so that a wave
// reads 64 consecutive samples of one exon (coalesced) and writes three coalesced rows.
__global__ void __launch_bounds__(kEmitBlock)
k_emit_batch(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, const double* __restrict__ consts,
             const int* __restrict__ cflags, int64_t E, int64_t S, double* __restrict__ loglik,
             unsigned long long* __restrict__ nerr)
{
  const int64_t cell = (int64_t)blockIdx.x * kEmitBlock + threadIdx.x;
  if (cell >= E * S) return;
  const int64_t e = cell / S;
  const int64_t s = cell - e * S;
  const int32_t obs = test[cell];
  const int32_t tot = obs + ref[cell];   // as.integer(reference + test), R/class_definition.R:187
  int nflag = 0;
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const double a1 = consts[(st * 3 + 0) * S + s];
    const double a2 = consts[(st * 3 + 1) * S + s];
    const double c = consts[(st * 3 + 2) * S + s];
    int flag = 0;
    const double x = a1 + (double)obs;
    const double y = (a2 + (double)tot) - (double)obs;
    const double v = edsf::lnbeta(x, y, &flag) - c;
    loglik[(e * 3 + st) * S + s] = v;
    nflag += flag + cflags[st * S + s];
  }
  if (nflag) atomicAdd(nerr, (unsigned long long)nflag);
}

// the reference's own signature: per-exon phi and expected; out is n x 3 column-major
__global__ void __launch_bounds__(kEmitBlock)
k_emit_rows(const double* __restrict__ phi, const double* __restrict__ expected, const int32_t* __restrict__ total,
            const int32_t* __restrict__ observed, int64_t n, double mixture, double* __restrict__ out,
            unsigned long long* __restrict__ nerr)
{
  const int64_t i = (int64_t)blockIdx.x * kEmitBlock + threadIdx.x;
  if (i >= n) return;
  const double e = expected[i];
  const double sd = __builtin_sqrt((phi[i] * e) * (1. - e));
  double ep[3];
  state_props(e, mixture, ep);
  const int32_t tot = total[i], obs = observed[i];
  int nflag = 0;
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    double a1, a2;
    shape_params(ep[st], sd, a1, a2);
    int f1 = 0, f2 = 0;
    const double v1 = edsf::lnbeta(a1 + (double)obs, (a2 + (double)tot) - (double)obs, &f1);
    const double v0 = edsf::lnbeta(a1, a2, &f2);
    out[i + n * st] = v1 - v0;
    nflag += f1 + f2;
  }
  if (nflag) atomicAdd(nerr, (unsigned long long)nflag);
}

// ---- Viterbi --------------------------------------------------------------------------------

// One forward step (src/hmm.cpp:68-88).  v[] are the previous scores, e[] the three emissions in HMM
// order (normal, deletion, duplication), lt[j*3+k] = log(trans k->j) for this exon gap (host-built).
// Candidate order and the strict '>' reproduce the reference's tie-breaking (first maximum wins).
// Back-pointer of a state no candidate improves on is 0 (the reference leaves -1 there, which it can
// only ever use as an out-of-bounds index: see include/exomedepth_amd.h).
__device__ __forceinline__ unsigned vit_step(double v[3], const double e[3], const double* __restrict__ lt)
{
  double nv[3];
  unsigned bp = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    double best = -HUGE_VAL;
    unsigned fw = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double cand = (e[j] + v[k]) + lt[j * 3 + k];
      if (cand > best) {
        best = cand;
        fw = k;
      }
    }
    if (e[j] == -HUGE_VAL) fw = 0;
    nv[j] = best;
    bp |= fw << (2 * j);
  }
  v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2];
  return bp;
}

// Run-length summary of one chain (src/hmm.cpp:104-126), quirks kept: `start` is only (re)set when
// leaving state 0, `nexons` only resets when a call is pushed.  tb(i) returns the state of padded
// observation i (0..last).  emit(start_i, end_i, type, nexons) receives 0-based padded indices.
template <class TB, class EMIT>
__device__ __forceinline__ int summarise_chain(int64_t last, TB tb, EMIT emit)
{
  int64_t start = -1;
  int nexons = 0, current = 0, ncalls = 0;
  int prev = tb(0);
  for (int64_t i = 1; i <= last; ++i) {
    const int cur = tb(i);
    if (prev != cur) {
      if (current == 0) start = i;
      if (current != 0) {
        emit(start, i - 1, current, nexons, ncalls);
        ++ncalls;
        nexons = 0;
      }
    }
    if (cur != 0) ++nexons;
    current = cur;
    prev = cur;
  }
  return ncalls;
}

// Batched chains.  Block = one wave; lane = sample; blockIdx.y = chromosome.
//   loglik [E][3][S] in S4 column order (deletion, normal, duplication): HMM state j reads column {1,0,2}[j]
//   lt     [(E + C)][9]: gap g = lo + c + (i-1) for padded step i = 1..m+1 of chromosome c (lo = chrom_off[c])
//   path   [E][S]: holds the packed back-pointers during the forward pass, the Viterbi state afterwards
// The two dummy observations of CallCNVs (R/class_definition.R:364) are implicit: the chain starts
// from (0,-inf,-inf) (src/hmm.cpp:48-52; the first dummy row is never read) and ends with one extra
// step whose emissions are (-100, 0, -100).
__global__ void __launch_bounds__(kWave)
k_viterbi(const double* __restrict__ loglik, const double* __restrict__ lt, const int32_t* __restrict__ chrom_off,
          int64_t S, int32_t C, uint8_t* __restrict__ path, int32_t* __restrict__ counts)
{
  const int64_t s = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const int c = blockIdx.y;
  if (s >= S) return;
  const int64_t lo = chrom_off[c], hi = chrom_off[c + 1];
  const int64_t m = hi - lo;
  if (m <= 0) {
    counts[s * C + c] = 0;
    return;
  }
  const double* __restrict__ ltc = lt + (lo + c) * 9;
  double v[3] = {0., -HUGE_VAL, -HUGE_VAL};
  // forward pass with a one-step-ahead prefetch of the emissions
  double en[3];
  {
    const double* p = loglik + (lo * 3) * S + s;
    en[0] = p[S]; en[1] = p[0]; en[2] = p[2 * S];
  }
  for (int64_t i = 0; i < m; ++i) {
    double e[3] = {en[0], en[1], en[2]};
    if (i + 1 < m) {
      const double* p = loglik + ((lo + i + 1) * 3) * S + s;
      en[0] = p[S]; en[1] = p[0]; en[2] = p[2 * S];
    }
    const unsigned bp = vit_step(v, e, ltc + i * 9);
    path[(lo + i) * S + s] = (uint8_t)bp;
  }
  // dummy last observation: only the back-pointer of state 0 is ever used (src/hmm.cpp:96)
  int cur;
  {
    const double e[3] = {-100., 0., -100.};
    const unsigned bp = vit_step(v, e, ltc + m * 9);
    cur = bp & 3;
  }
  // trace back (src/hmm.cpp:95-100), overwriting the back-pointers with the states
  for (int64_t i = m - 1; i >= 0; --i) {
    const int64_t a = (lo + i) * S + s;
    const unsigned bp = path[a];
    path[a] = (uint8_t)cur;
    cur = (bp >> (2 * cur)) & 3;
  }
  const int tb0 = cur;  // state of the first dummy observation
  // count the calls
  auto tb = [&](int64_t i) -> int { return i == 0 ? tb0 : (i == m + 1 ? 0 : (int)path[(lo + i - 1) * S + s]); };
  auto nop = [](int64_t, int64_t, int, int, int) {};
  counts[s * C + c] = summarise_chain(m + 1, tb, nop);
}

// exclusive scan of n int32 counts by one workgroup; total written to *total
__global__ void __launch_bounds__(1024) k_scan_counts(const int32_t* __restrict__ counts, int64_t n,
                                                      int64_t* __restrict__ offsets, int64_t* __restrict__ total)
{
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    int64_t x = (i < n) ? counts[i] : 0;
    int64_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int64_t t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int64_t wpre = 0;
    for (int k = 0; k < w; ++k) wpre += wsum[k];
    const int64_t c0 = carry;
    if (i < n) offsets[i] = c0 + wpre + incl - x;
    __syncthreads();
    if (tid == 1023) carry = c0 + wpre + incl;
    __syncthreads();
  }
  if (tid == 0) *total = carry;
}

__global__ void __launch_bounds__(kWave)
k_calls_fill(const uint8_t* __restrict__ path, const int32_t* __restrict__ chrom_off, int64_t S, int32_t C,
             const int64_t* __restrict__ offsets, ed_call* __restrict__ calls, int64_t cap)
{
  const int64_t s = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const int c = blockIdx.y;
  if (s >= S) return;
  const int64_t lo = chrom_off[c], hi = chrom_off[c + 1];
  const int64_t m = hi - lo;
  if (m <= 0) return;
  const int64_t off = offsets[s * C + c];
  auto tb = [&](int64_t i) -> int { return (i == 0 || i == m + 1) ? 0 : (int)path[(lo + i - 1) * S + s]; };
  auto emit = [&](int64_t st, int64_t en, int type, int nexons, int k) {
    const int64_t r = off + k;
    if (r < cap) {
      ed_call rec;
      rec.sample = (int32_t)s;
      rec.chrom = c;
      rec.start_exon = (int32_t)(lo + st - 1);
      rec.end_exon = (int32_t)(lo + en - 1);
      rec.type = type;
      rec.nexons = nexons;
      calls[r] = rec;
    }
  };
  summarise_chain(m + 1, tb, emit);
}

// Single chain with caller-supplied probabilities: the reference's C_hmm signature.
//   proba nobs x 3 column-major (HMM order), lt [(nobs-1)][9], path_out double[nobs]
//   calls_out column-major cap x 4, ncalls_out
// One lane does the sequential work (a drop-in for fidelity, not a throughput path).
__global__ void k_viterbi_single(const double* __restrict__ proba, const double* __restrict__ lt, int64_t nobs,
                                 uint8_t* __restrict__ bp, double* __restrict__ path_out,
                                 double* __restrict__ calls_out, int64_t cap, int64_t* __restrict__ ncalls_out)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  double v[3] = {0., -HUGE_VAL, -HUGE_VAL};
  for (int64_t i = 1; i < nobs; ++i) {
    const double e[3] = {proba[i], proba[nobs + i], proba[2 * nobs + i]};
    bp[i] = (uint8_t)vit_step(v, e, lt + (i - 1) * 9);
  }
  int cur = 0;  // last observation forced to state 0 (src/hmm.cpp:96)
  for (int64_t i = nobs - 1; i >= 1; --i) {
    const unsigned b = bp[i];
    bp[i] = (uint8_t)cur;
    cur = (b >> (2 * cur)) & 3;
  }
  bp[0] = (uint8_t)cur;
  for (int64_t i = 0; i < nobs; ++i) path_out[i] = (double)bp[i];
  auto tb = [&](int64_t i) -> int { return (int)bp[i]; };
  auto emit = [&](int64_t st, int64_t en, int type, int nexons, int k) {
    if (k < cap) {
      calls_out[0 * cap + k] = (double)(st + 1);
      calls_out[1 * cap + k] = (double)(en + 1);
      calls_out[2 * cap + k] = (double)type;
      calls_out[3 * cap + k] = (double)nexons;
    }
  };
  *ncalls_out = summarise_chain(nobs - 1, tb, emit);
}

// test hook: element-wise device special functions
__global__ void k_eval_sf(int which, int64_t n, const double* __restrict__ x, const double* __restrict__ y,
                          double* __restrict__ out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int flag = 0;
  double r;
  switch (which) {
    case 0: r = edsf::lnbeta(x[i], y[i], &flag); break;
    case 1: r = ed_plog(x[i]); break;
    case 2: r = ed_pexp(x[i]); break;
    case 3: r = __builtin_sqrt(x[i]); break;
    case 4: r = x[i] / y[i]; break;
    case 5: r = ed_psin_0pi(x[i]); break;
    default: r = ed_pm_nan();
  }
  out[i] = r;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ed_plan {
  int device = 0;
  int64_t E = 0;
  int32_t C = 0;
  double tprob = 0, L = 0;
  std::vector<int32_t> chrom_off;
  int32_t* d_chrom_off = nullptr;
  double* d_lt = nullptr;  // [(E + C)][9]
};

struct ed_batch {
  ed_plan* plan = nullptr;
  int64_t S = 0;
  double* d_loglik = nullptr;
  uint8_t* d_path = nullptr;
  double* d_consts = nullptr;
  int* d_cflags = nullptr;
  int32_t* d_counts = nullptr;
  int64_t* d_offsets = nullptr;
  int64_t* d_total = nullptr;
  unsigned long long* d_nerr = nullptr;
  ed_call* d_calls = nullptr;
  int64_t calls_cap = 0;
  hipStream_t stream = nullptr;
  bool ran = false;
  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool have_run_times = false, have_fit_time = false;
};

// log-transition table of one chain: for padded positions pos[0..n-1], gaps i=1..n-1
// (src/hmm.cpp:62-79).  Host libm on purpose: these are the reference's own exp()/log() calls, and
// the table is sample-independent (9 doubles per gap shared by every sample of every batch).
static void fill_log_transitions(const double T[9], double L, const int32_t* pos, int64_t n, double* lt)
{
  for (int64_t i = 1; i < n; ++i) {
    const double dist = double(pos[i]) - double(pos[i - 1]);
    const double d = std::exp(-dist / L);
    double* o = lt + (i - 1) * 9;
    for (int j = 0; j < 3; ++j) {
      const double t0 = T[j * 3];
      const double t1 = d * T[j * 3 + 1] + (1.0 - d) * T[j * 3];
      const double t2 = d * T[j * 3 + 2] + (1.0 - d) * T[j * 3];
      o[j * 3 + 0] = std::log(t0);
      o[j * 3 + 1] = std::log(t1);
      o[j * 3 + 2] = std::log(t2);
    }
  }
}

ED_EXPORT const char* ed_version(void) { return "exomedepth_amd 0.1 (gfx950)"; }
ED_EXPORT const char* ed_last_error(void) { return g_last_error.c_str(); }

ED_EXPORT int ed_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

ED_EXPORT int ed_device_info(int device, char* name, size_t name_len, int* compute_units, size_t* total_mem)
{
  if (int rc = require_device()) return rc;
  hipDeviceProp_t p;
  HIP_TRY(hipGetDeviceProperties(&p, device));
  if (name && name_len) {
    std::strncpy(name, p.gcnArchName, name_len - 1);
    name[name_len - 1] = 0;
  }
  if (compute_units) *compute_units = p.multiProcessorCount;
  if (total_mem) *total_mem = p.totalGlobalMem;
  return ED_OK;
}

ED_EXPORT int ed_malloc(void** dptr, size_t bytes)
{
  if (!dptr) return ed_fail(ED_ERR_INVALID, "ed_malloc: NULL output pointer");
  if (int rc = require_device()) return rc;
  HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return ED_OK;
}
ED_EXPORT int ed_free(void* dptr)
{
  if (dptr) HIP_TRY(hipFree(dptr));
  return ED_OK;
}
ED_EXPORT int ed_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return ED_OK;
}
ED_EXPORT int ed_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return ED_OK;
}
ED_EXPORT int ed_synchronize(void* stream)
{
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return ED_OK;
}

namespace {
// RAII device buffer for the host-buffer entry points
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
  template <class T> T* as() { return (T*)p; }
};
}  // namespace

ED_EXPORT int ed_eval_sf(int which, int64_t n, const double* x, const double* y, double* out)
{
  if (n < 0 || !x || !out) return ed_fail(ED_ERR_INVALID, "ed_eval_sf: bad arguments");
  if (int rc = require_device()) return rc;
  if (n == 0) return ED_OK;
  DevBuf dx, dy, dout;
  HIP_TRY(dx.alloc(n * 8)); HIP_TRY(dy.alloc(n * 8)); HIP_TRY(dout.alloc(n * 8));
  HIP_TRY(hipMemcpy(dx.p, x, n * 8, hipMemcpyHostToDevice));
  if (y) HIP_TRY(hipMemcpy(dy.p, y, n * 8, hipMemcpyHostToDevice));
  else HIP_TRY(hipMemset(dy.p, 0, n * 8));
  hipLaunchKernelGGL(k_eval_sf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, which, n, dx.as<double>(),
                     dy.as<double>(), dout.as<double>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
  return ED_OK;
}

// ---- drop-in 1: get_loglike_matrix ---------------------------------------------------------
ED_EXPORT int ed_get_loglike_matrix(const double* phi, const double* expected, const int32_t* total,
                                    const int32_t* observed, int64_t n, double mixture, double* out,
                                    int64_t* n_gsl_errors)
{
  if (n < 0 || (n > 0 && (!phi || !expected || !total || !observed || !out)))
    return ed_fail(ED_ERR_INVALID, "ed_get_loglike_matrix: NULL buffer or negative n");
  if (int rc = require_device()) return rc;
  if (n_gsl_errors) *n_gsl_errors = 0;
  if (n == 0) return ED_OK;
  DevBuf dphi, dexp, dtot, dobs, dout, dnerr;
  HIP_TRY(dphi.alloc(n * 8)); HIP_TRY(dexp.alloc(n * 8)); HIP_TRY(dtot.alloc(n * 4)); HIP_TRY(dobs.alloc(n * 4));
  HIP_TRY(dout.alloc(n * 24)); HIP_TRY(dnerr.alloc(8));
  HIP_TRY(hipMemcpy(dphi.p, phi, n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dexp.p, expected, n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dtot.p, total, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dobs.p, observed, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(dnerr.p, 0, 8));
  hipLaunchKernelGGL(k_emit_rows, dim3((unsigned)((n + kEmitBlock - 1) / kEmitBlock)), dim3(kEmitBlock), 0, 0,
                     dphi.as<double>(), dexp.as<double>(), dtot.as<int32_t>(), dobs.as<int32_t>(), n, mixture,
                     dout.as<double>(), dnerr.as<unsigned long long>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(out, dout.p, n * 24, hipMemcpyDeviceToHost));
  unsigned long long ne = 0;
  HIP_TRY(hipMemcpy(&ne, dnerr.p, 8, hipMemcpyDeviceToHost));
  if (n_gsl_errors) *n_gsl_errors = (int64_t)ne;
  return ED_OK;
}

// ---- drop-in 2: C_hmm ----------------------------------------------------------------------
ED_EXPORT int ed_hmm(int32_t nstates, int32_t nobs, const double* transitions, const double* probabilities,
                     const int32_t* positions, double expected_length, double* path_out, double* calls_out,
                     int64_t calls_cap, int64_t* n_calls)
{
  if (nstates != 3) return ed_fail(ED_ERR_INVALID, "ERROR: The code must assume 3 states");  // src/hmm.cpp:37-40
  if (nobs < 0 || calls_cap < 0 || !n_calls || (nobs > 0 && (!transitions || !probabilities || !positions || !path_out)))
    return ed_fail(ED_ERR_INVALID, "ed_hmm: bad arguments");
  if (calls_cap > 0 && !calls_out) return ed_fail(ED_ERR_INVALID, "ed_hmm: calls_out is NULL");
  if (int rc = require_device()) return rc;
  *n_calls = 0;
  if (nobs == 0) return ED_OK;
  std::vector<double> lt((size_t)std::max<int64_t>(nobs - 1, 1) * 9);
  fill_log_transitions(transitions, expected_length, positions, nobs, lt.data());
  DevBuf dproba, dlt, dbp, dpath, dcalls, dn;
  HIP_TRY(dproba.alloc((size_t)nobs * 24)); HIP_TRY(dlt.alloc(lt.size() * 8)); HIP_TRY(dbp.alloc(nobs));
  HIP_TRY(dpath.alloc((size_t)nobs * 8)); HIP_TRY(dcalls.alloc((size_t)calls_cap * 32)); HIP_TRY(dn.alloc(8));
  HIP_TRY(hipMemcpy(dproba.p, probabilities, (size_t)nobs * 24, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dlt.p, lt.data(), lt.size() * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_viterbi_single, dim3(1), dim3(64), 0, 0, dproba.as<double>(), dlt.as<double>(), (int64_t)nobs,
                     dbp.as<uint8_t>(), dpath.as<double>(), dcalls.as<double>(), calls_cap, dn.as<int64_t>());
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(path_out, dpath.p, (size_t)nobs * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(n_calls, dn.p, 8, hipMemcpyDeviceToHost));
  if (calls_cap > 0) HIP_TRY(hipMemcpy(calls_out, dcalls.p, (size_t)calls_cap * 32, hipMemcpyDeviceToHost));
  return ED_OK;
}

// ---- plan ----------------------------------------------------------------------------------
ED_EXPORT int ed_plan_create(ed_plan** plan, int device, int64_t n_exons, int32_t n_chrom, const int32_t* chrom_off,
                             const int32_t* start, const int32_t* end, double transition_probability,
                             double expected_cnv_length)
{
  if (!plan || n_exons < 0 || n_chrom < 0 || !chrom_off || (n_exons > 0 && (!start || !end)))
    return ed_fail(ED_ERR_INVALID, "ed_plan_create: bad arguments");
  if (chrom_off[0] != 0 || chrom_off[n_chrom] != n_exons)
    return ed_fail(ED_ERR_INVALID, "ed_plan_create: chrom_off must run from 0 to n_exons");
  for (int c = 0; c < n_chrom; ++c)
    if (chrom_off[c + 1] < chrom_off[c]) return ed_fail(ED_ERR_INVALID, "ed_plan_create: chrom_off not monotone");
  if (int rc = require_device()) return rc;
  HIP_TRY(hipSetDevice(device));
  ed_plan* p = new (std::nothrow) ed_plan;
  if (!p) return ed_fail(ED_ERR_NOMEM, "out of host memory");
  p->device = device; p->E = n_exons; p->C = n_chrom; p->tprob = transition_probability; p->L = expected_cnv_length;
  p->chrom_off.assign(chrom_off, chrom_off + n_chrom + 1);
  // transitions <- matrix(c(1-t, t/2, t/2, .5,.5,0, .5,0,.5), byrow=TRUE)  (R/class_definition.R:343-347),
  // stored column-major as C_hmm reads it: T[j*3+k] = row k, column j
  const double t = transition_probability;
  const double rows[3][3] = {{1. - t, t / 2., t / 2.}, {0.5, 0.5, 0.}, {0.5, 0., 0.5}};
  double T[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[c * 3 + r] = rows[r][c];
  std::vector<double> lt((size_t)(n_exons + n_chrom) * 9 + 9);
  {
    // one task per chromosome, spread over the host threads
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    std::vector<std::thread> pool;
    std::vector<int> order(n_chrom);
    for (int c = 0; c < n_chrom; ++c) order[c] = c;
    auto work = [&](unsigned tid) {
      for (int idx = tid; idx < n_chrom; idx += nt) {
        const int c = order[idx];
        const int64_t lo = chrom_off[c], hi = chrom_off[c + 1], m = hi - lo;
        if (m <= 0) continue;
        std::vector<int32_t> pos((size_t)m + 2);
        // as.integer(c(positions[1] - 2*L, positions, end[last] + 2*L))  (R/class_definition.R:368)
        pos[0] = (int32_t)((double)start[lo] - 2 * expected_cnv_length);
        for (int64_t i = 0; i < m; ++i) pos[1 + i] = start[lo + i];
        pos[m + 1] = (int32_t)((double)end[hi - 1] + 2 * expected_cnv_length);
        fill_log_transitions(T, expected_cnv_length, pos.data(), m + 2, lt.data() + (size_t)(lo + c) * 9);
      }
    };
    for (unsigned tid = 1; tid < nt; ++tid) pool.emplace_back(work, tid);
    work(0);
    for (auto& th : pool) th.join();
  }
  hipError_t e1 = hipMalloc((void**)&p->d_lt, lt.size() * 8);
  hipError_t e2 = hipMalloc((void**)&p->d_chrom_off, (size_t)(n_chrom + 1) * 4);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    ed_plan_destroy(p);
    return ed_fail(ED_ERR_NOMEM, "ed_plan_create: device allocation failed");
  }
  HIP_TRY(hipMemcpy(p->d_lt, lt.data(), lt.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_chrom_off, chrom_off, (size_t)(n_chrom + 1) * 4, hipMemcpyHostToDevice));
  *plan = p;
  return ED_OK;
}

ED_EXPORT void ed_plan_destroy(ed_plan* p)
{
  if (!p) return;
  if (p->d_lt) (void)hipFree(p->d_lt);
  if (p->d_chrom_off) (void)hipFree(p->d_chrom_off);
  delete p;
}

ED_EXPORT int64_t ed_plan_n_exons(const ed_plan* p) { return p ? p->E : 0; }

// ---- batch ---------------------------------------------------------------------------------
ED_EXPORT int ed_batch_create(ed_batch** batch, ed_plan* plan, int64_t n_samples)
{
  if (!batch || !plan || n_samples <= 0) return ed_fail(ED_ERR_INVALID, "ed_batch_create: bad arguments");
  if (int rc = require_device()) return rc;
  HIP_TRY(hipSetDevice(plan->device));
  ed_batch* b = new (std::nothrow) ed_batch;
  if (!b) return ed_fail(ED_ERR_NOMEM, "out of host memory");
  b->plan = plan; b->S = n_samples;
  const int64_t E = plan->E, S = n_samples, C = plan->C;
  // capacity of the call table: generous for real data (a few hundred calls per sample), bounded so a
  // pathological input cannot exhaust HBM; ed_batch_n_calls reports the true total either way.
  b->calls_cap = std::min<int64_t>(std::max<int64_t>(1 << 20, 512 * S), std::max<int64_t>(E * S / 2, 1));
  bool ok = true;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes ? bytes : 1) != hipSuccess) ok = false; };
  A((void**)&b->d_loglik, (size_t)E * 3 * S * 8);
  A((void**)&b->d_path, (size_t)E * S);
  A((void**)&b->d_consts, (size_t)9 * S * 8);
  A((void**)&b->d_cflags, (size_t)3 * S * 4);
  A((void**)&b->d_counts, (size_t)S * std::max<int64_t>(C, 1) * 4);
  A((void**)&b->d_offsets, (size_t)S * std::max<int64_t>(C, 1) * 8);
  A((void**)&b->d_total, 8);
  A((void**)&b->d_nerr, 8);
  A((void**)&b->d_calls, (size_t)b->calls_cap * sizeof(ed_call));
  if (!ok) {
    ed_batch_destroy(b);
    return ed_fail(ED_ERR_NOMEM, "ed_batch_create: device allocation failed (E=%lld S=%lld)", (long long)E, (long long)S);
  }
  for (auto& e : b->ev) HIP_TRY(hipEventCreate(&e));
  *batch = b;
  return ED_OK;
}

ED_EXPORT void ed_batch_destroy(ed_batch* b)
{
  if (!b) return;
  void* ptrs[] = {b->d_loglik, b->d_path, b->d_consts, b->d_cflags, b->d_counts, b->d_offsets, b->d_total, b->d_nerr, b->d_calls};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  delete b;
}

ED_EXPORT int ed_batch_enable_timing(ed_batch* b, int enable)
{
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->timing = enable != 0;
  return ED_OK;
}

ED_EXPORT int ed_batch_run(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                           const double* d_expected, double mixture, void* stream_)
{
  if (!b || !d_test || !d_ref || !d_phi || !d_expected) return ed_fail(ED_ERR_INVALID, "ed_batch_run: NULL argument");
  hipStream_t st = (hipStream_t)stream_;
  const ed_plan* p = b->plan;
  const int64_t E = p->E, S = b->S;
  const int32_t C = p->C;
  b->stream = st;
  HIP_TRY(hipMemsetAsync(b->d_nerr, 0, 8, st));
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[0], st));
  hipLaunchKernelGGL(k_sample_consts, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, d_phi, d_expected, mixture, S,
                     b->d_consts, b->d_cflags);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[1], st));
  const int64_t cells = E * S;
  if (cells > 0)
    hipLaunchKernelGGL(k_emit_batch, dim3((unsigned)((cells + kEmitBlock - 1) / kEmitBlock)), dim3(kEmitBlock), 0, st,
                       d_test, d_ref, b->d_consts, b->d_cflags, E, S, b->d_loglik, b->d_nerr);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[2], st));
  if (C > 0 && cells > 0)
    hipLaunchKernelGGL(k_viterbi, dim3((unsigned)((S + kWave - 1) / kWave), (unsigned)C), dim3(kWave), 0, st,
                       b->d_loglik, p->d_lt, p->d_chrom_off, S, C, b->d_path, b->d_counts);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[3], st));
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, b->d_counts, (C > 0 && cells > 0) ? S * C : 0,
                     b->d_offsets, b->d_total);
  if (C > 0 && cells > 0)
    hipLaunchKernelGGL(k_calls_fill, dim3((unsigned)((S + kWave - 1) / kWave), (unsigned)C), dim3(kWave), 0, st,
                       b->d_path, p->d_chrom_off, S, C, b->d_offsets, b->d_calls, b->calls_cap);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[4], st));
  HIP_TRY(hipGetLastError());
  b->ran = true;
  b->have_run_times = b->timing;
  return ED_OK;
}

ED_EXPORT int ed_batch_fit(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, double* d_phi, double* d_expected,
                           void* stream_)
{
  (void)b; (void)d_test; (void)d_ref; (void)d_phi; (void)d_expected; (void)stream_;
  return ed_fail(ED_ERR_STATE, "ed_batch_fit: dispersion-fit kernel not built yet");
}

ED_EXPORT const double* ed_batch_loglik(const ed_batch* b) { return b ? b->d_loglik : nullptr; }
ED_EXPORT const uint8_t* ed_batch_path(const ed_batch* b) { return b ? b->d_path : nullptr; }
ED_EXPORT const ed_call* ed_batch_calls(const ed_batch* b) { return b ? b->d_calls : nullptr; }

static int batch_ready(ed_batch* b)
{
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (!b->ran) return ed_fail(ED_ERR_STATE, "no ed_batch_run has been issued on this batch");
  HIP_TRY(hipStreamSynchronize(b->stream));
  return ED_OK;
}

ED_EXPORT int ed_batch_n_calls(ed_batch* b, int64_t* n_calls)
{
  if (int rc = batch_ready(b)) return rc;
  if (!n_calls) return ed_fail(ED_ERR_INVALID, "NULL output");
  HIP_TRY(hipMemcpy(n_calls, b->d_total, 8, hipMemcpyDeviceToHost));
  return ED_OK;
}

ED_EXPORT int ed_batch_n_gsl_errors(ed_batch* b, int64_t* n_events)
{
  if (int rc = batch_ready(b)) return rc;
  if (!n_events) return ed_fail(ED_ERR_INVALID, "NULL output");
  unsigned long long v = 0;
  HIP_TRY(hipMemcpy(&v, b->d_nerr, 8, hipMemcpyDeviceToHost));
  *n_events = (int64_t)v;
  return ED_OK;
}

ED_EXPORT int ed_batch_copy_calls(ed_batch* b, ed_call* host_calls, int64_t cap)
{
  int64_t n = 0;
  if (int rc = ed_batch_n_calls(b, &n)) return rc;
  if (n > b->calls_cap)
    return ed_fail(ED_ERR_STATE, "call table overflow: %lld calls, capacity %lld", (long long)n, (long long)b->calls_cap);
  const int64_t k = std::min(n, cap);
  if (k > 0) {
    if (!host_calls) return ed_fail(ED_ERR_INVALID, "NULL output");
    HIP_TRY(hipMemcpy(host_calls, b->d_calls, (size_t)k * sizeof(ed_call), hipMemcpyDeviceToHost));
  }
  return ED_OK;
}

ED_EXPORT int ed_batch_copy_path(ed_batch* b, uint8_t* host_path)
{
  if (int rc = batch_ready(b)) return rc;
  if (!host_path) return ed_fail(ED_ERR_INVALID, "NULL output");
  HIP_TRY(hipMemcpy(host_path, b->d_path, (size_t)b->plan->E * b->S, hipMemcpyDeviceToHost));
  return ED_OK;
}

ED_EXPORT int ed_batch_copy_loglik(ed_batch* b, double* host_loglik)
{
  if (int rc = batch_ready(b)) return rc;
  if (!host_loglik) return ed_fail(ED_ERR_INVALID, "NULL output");
  HIP_TRY(hipMemcpy(host_loglik, b->d_loglik, (size_t)b->plan->E * 3 * b->S * 8, hipMemcpyDeviceToHost));
  return ED_OK;
}

ED_EXPORT int ed_batch_stage_ms(ed_batch* b, float ms[5])
{
  if (!b || !ms) return ed_fail(ED_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 5; ++i) ms[i] = 0.f;
  if (b->have_run_times) {
    HIP_TRY(hipEventSynchronize(b->ev[4]));
    for (int i = 0; i < 4; ++i) HIP_TRY(hipEventElapsedTime(&ms[i], b->ev[i], b->ev[i + 1]));
  }
  return ED_OK;
}
