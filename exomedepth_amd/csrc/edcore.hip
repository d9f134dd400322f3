// edcore.hip -- kernels and C-ABI of libedcore.so (gfx950 / MI355X).  See include/exomedepth_amd.h.
//
// Kernels (names follow the reference's domain: exons, samples, chains = (sample, chromosome)):
//   k_sample_consts   per sample: the three (a1, a2, lnbeta(a1,a2)) triples of myprob (src/CNV_estimate.cpp:44-50)
//   k_emit_tables     per (sample, state, observed): the terms of log B that depend on the test count alone
//   k_emit_batch      per (exon, sample) cell: three beta-binomial log-likelihoods (src/CNV_estimate.cpp:71-81)
//   k_emit_rows       the same for per-exon phi/expected (the reference's .Call signature)
//   k_viterbi         per chain: forward max-plus pass with back-pointers (src/hmm.cpp:58-90)
//   k_tb_maps / k_tb_chain / k_tb_paths   the trace-back (src/hmm.cpp:95-100), data-parallel, + call counts
//   k_scan_counts     exclusive scan of the per-chain call counts
//   k_calls_fill      per chain: writes the call records (src/hmm.cpp:104-126, R/class_definition.R:371-372,:409-410)
//   k_call_info       decoration of the calls (R/class_definition.R:379-405)
//   k_fit_*           per-sample beta-binomial fit (aod::betabin's role, R/class_definition.R:118):
//                     k_fit_moments/start, k_fit_accum/update (per cell)
//   edfit_hist.inc    k_fit_hist + k_fit_hnewton: the histogram form of that fit, in three geometries (hg8 / hg4 / hg2)
//   edtab.inc         k_tab_* + k_emit_tab: the table-driven emission mode (log-gamma difference tables per sample and state)
//   edfused.inc       k_emit_viterbi: emissions + Viterbi in one kernel (optional mode)
//   edrefset.inc      select.reference.set (R/optimize_reference_set.R:53-148)
//   edbins.inc        phi.bins > 1 (R/class_definition.R:120-147)
//   edcov.inc         covariates in the mean model (data + formula, R/class_definition.R:86-118)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (contraction off is part of the contract:
// the arithmetic must match the CPU checker bit for bit).
#include <hip/hip_runtime.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <thread>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/exomedepth_amd.h"
#include "ed_sf_dev.hpp"
#include "ed_fit_dev.hpp"
#include "ed_dtab.h"

#define ED_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

// Experiment knobs (table reach / tile width, waves of the persistent Viterbi grid, shares per sample of the emission launch, the cohort
// pipeline's queue layout) are read from the environment only in builds made with -DED_EXPERIMENT_KNOBS (exomedepth_amd/_build.py variant
// "knobs"): the shipped library's behaviour does not depend on the caller's environment.
static const char* ed_knob(const char* name)
{
#ifdef ED_EXPERIMENT_KNOBS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

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

// hipMemset() on device memory is a NULL-STREAM operation that may return before it has run.  A buffer initialised that way and used next on a
// hipStreamNonBlocking stream (the pipeline's streams with option own_queues = 0, or a caller's) is not ordered behind the fill: found as a fit that
// started from poisoned partial sums (tests/test_gpu_cohort.py, six slabs in flight, own_queues = 0, torch's HIP runtime in the process: results equal
// to 1e-15 instead of bit for bit) -- the workspace's all-ones fill landed after the first pass had written its sums.  Every one-time initialisation by
// hipMemset is therefore followed by this fence before the buffer is handed to a stream.
#ifdef ED_X_NO_NULL_FENCE      // (diagnostic build only)
static hipError_t ed_null_stream_fence() { return hipSuccess; }
#else
static hipError_t ed_null_stream_fence() { return hipStreamSynchronize(nullptr); }
#endif

// No C++ exception leaves the library: the callers are C (R's .Call, ctypes).  Every int-returning entry point is a function-try-block
// closed by ED_CATCH, which turns what was thrown (in practice std::bad_alloc from a host container, std::system_error from a
// thread that could not be started) into an error code + ed_last_error().
static int ed_caught(const char* where) noexcept
{
  try { throw; }
  catch (const std::bad_alloc&) { try { return ed_fail(ED_ERR_NOMEM, "%s: out of host memory", where); } catch (...) { return ED_ERR_NOMEM; } }
  catch (const std::exception& e) { try { return ed_fail(ED_ERR_STATE, "%s: %s", where, e.what()); } catch (...) { return ED_ERR_STATE; } }
  catch (...) { try { return ed_fail(ED_ERR_STATE, "%s: unknown C++ exception", where); } catch (...) { return ED_ERR_STATE; } }
}
#define ED_CATCH(name) catch (...) { return ed_caught(name); }
// host threads that are joined on every way out of their scope (an exception between start and join would otherwise terminate the process)
struct ed_thread_pool {
  std::vector<std::thread> t;
  template <class... A> void start(A&&... a) { t.emplace_back(std::forward<A>(a)...); }
  void join() { for (auto& x : t) if (x.joinable()) x.join(); }
  ~ed_thread_pool() { join(); }
};

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
constexpr int kEmitTab = 1024;   // test counts covered by the per-sample tables (k_emit_tables)
constexpr uint32_t kEmitRun = 256;   // exon blocks an XCD spends on one sample block before taking up the next (k_emit_batch)
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

// fitted(mod) = plogis(beta_0 + sum_k beta_k x_ek) for exon e, sample s (edcov.inc)
__device__ __forceinline__ double cov_expected(const double* __restrict__ X, int K, const double* __restrict__ beta, int64_t e,
                                               int64_t S, int64_t s)
{
  double eta = beta[s];
  for (int k = 0; k < K; ++k) eta += beta[(int64_t)(k + 1) * S + s] * X[e * K + k];
  return 1.0 / (1.0 + ed_pexp(-eta));
}

// consts layout: [9][S] = a1_del,a2_del,C_del, a1_norm,a2_norm,C_norm, a1_dup,a2_dup,C_dup ; flags[3][S]
// Also zeroes, when handed them, the small per-run words the kernels behind it accumulate into -- the error / table counters (64 bytes),
// the table statistics of k_tab_stats [3][S], the list of samples without tables, the strict lists' counters: as hipMemsetAsync each
// of them was a fill kernel of its own on the stream that carries the emissions, queued behind whatever the chip was busy with
// (round 5 trace: 0.6 ms for the first one next to a starting k_viterbi_sm).
__global__ void k_sample_consts(const double* __restrict__ phi, const double* __restrict__ expected, double mixture,
                                int64_t S, double* __restrict__ consts, int* __restrict__ cflags, unsigned long long* __restrict__ z_nerr,
                                unsigned long long* __restrict__ z_tacc, unsigned int* __restrict__ z_notab, unsigned int* __restrict__ z_cold_n,
                                int n_cold_n, double* __restrict__ phi_copy, double* __restrict__ exp_copy)
{
  // phi_copy / exp_copy (cohort pipeline, parameters given by the caller): the slot's own copy of the slab's (phi, expected) -- what
  // the accessors and the call decoration read later -- written here instead of by two device-to-device copies on the emission stream
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (z_nerr && s < 8) z_nerr[s] = 0ull;
  if (z_notab && s == 0) { z_notab[0] = 0u; z_notab[S + 1] = 0u; }      // samples without tables; tail samples (edtab.inc)
  if (z_cold_n) for (int64_t i = s; i < n_cold_n; i += (int64_t)gridDim.x * blockDim.x) z_cold_n[i] = 0u;
  if (s >= S) return;
  if (z_tacc) { z_tacc[s] = 0ull; z_tacc[S + s] = 0ull; z_tacc[2 * S + s] = 0ull; }
  const double e = expected[s];
  if (phi_copy) { phi_copy[s] = phi[s]; exp_copy[s] = e; }
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

// x = a1 + observed is the only argument of three of the terms of log B(x, y): Gamma*(x) and log(x) on the ratio
// route (when x is the smaller argument -- the usual case, the test sample being one of ~10), log Gamma(x) on the
// general route.  For a sample and a state a1 is a constant and `observed` a small integer, so these terms are
// tabulated per (sample, state, observed < kEmitTab) before the emission kernel runs -- by the very functions the
// routes call, hence the same bits -- and the emission kernel gathers them instead of evaluating them: about a
// quarter of its arithmetic.  The gathers stay in L2: XCD x only ever works on sample blocks x, x + 8, ...
//   tab_gl [3][kEmitTab][S] double2 (Gamma*(x), log x)      tab_lg [3][kEmitTab][S] double  log Gamma(x)
__global__ void __launch_bounds__(256)
k_emit_tables(const double* __restrict__ consts, int64_t S, double2* __restrict__ tab_gl, double* __restrict__ tab_lg)
{
  __shared__ double s_logt[ED_PM_LOGT_N * 3];   // the portable log's table, see edsf::plog_pos
  {
    const double T0[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;
    for (int i = threadIdx.x; i < ED_PM_LOGT_N * 3; i += 256) s_logt[i] = (&T0[0][0])[i];
  }
  __syncthreads();
  const int64_t s = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int obs = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int st = blockIdx.z;
  if (s >= S || obs >= kEmitTab) return;
  const double x = consts[(st * 3 + 0) * S + s] + (double)obs;   // src/CNV_estimate.cpp:49
  double2 gl;
  double lg;
  if (x > 0.0 && x < HUGE_VAL) {
    gl.x = edsf::gammastar_pos(x);
    gl.y = edsf::plog_fast(x, s_logt);
    lg = edsf::lngamma_pos(x, false, s_logt);
  } else {
    gl.x = gl.y = lg = ed_pm_nan();   // not tabulated: the task evaluates (or takes the cold path) itself
  }
  const int64_t i = ((int64_t)st * kEmitTab + obs) * S + s;
  tab_gl[i] = gl;
  tab_lg[i] = lg;
}

// Raw buffer access: an SGPR resource (base pointer, size in bytes) + a 32-bit lane offset + a scalar offset.  Reads beyond
// the size return 0, writes beyond it are dropped.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ed_rsrc(const void* base, int32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double ed_buf_f64(__amdgpu_buffer_rsrc_t r, uint32_t voff, int32_t soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void ed_buf_store_f64_nt(__amdgpu_buffer_rsrc_t r, uint32_t voff, int32_t soff, double v)
{
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, 2);
}

// Emissions for the batch.  Each (cell, state) pair is one log-Beta *task*; a task takes one of two
// value-exact routes (ratio / general, see ed_sf_dev.hpp) that differ ~1.5x in cost and that adjacent
// cells pick differently (the selector is min/max < 0.2 of the shape parameters, src/beta.c:64-69).
// Left to the exec mask, nearly every wave would execute both routes.  Instead the workgroup bins its
// kEmitCells*3*kEmitBlock tasks through LDS: ratio-route tasks are packed from the front of the task
// array, all others from the back, so every wave of the evaluation phase but at most one runs a
// single route.  Cells are numbered exon-major / sample-minor: a wave reads 64 consecutive samples of
// one exon (coalesced) and writes three coalesced rows of the [E][3][S] likelihood matrix.
constexpr int kEmitCells = 1;                                // cells per thread
constexpr int kEmitTasks = kEmitBlock * kEmitCells * 3;      // tasks per workgroup
constexpr int kEmitRows = kEmitCells * kEmitBlock / 64;      // exons per workgroup tile (x 64 samples)
constexpr int64_t kEmitHeadBlocks = 2048;                    // workgroups of a group's short leading launch (ed_batch_run)
constexpr int kSideStreams = 3;                              // HIP maps streams onto 4 hardware queues: main + 3

// the portable log's table as a device global: staged into LDS by one load per thread (SGPR base + lane offset)
__device__ const double k_logt_rows[ED_PM_LOGT_N][3] = ED_PM_LOGT_ROWS;

// wave-level helpers: a ballot straight from a condition (no 0/1 materialisation) and the rank of a lane inside a mask
__device__ __forceinline__ unsigned long long ed_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ int ed_rank(unsigned long long m)
{
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

__global__ void __launch_bounds__(kEmitBlock)
k_emit_batch(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, const double* __restrict__ consts,
             const int* __restrict__ cflags, const int64_t* __restrict__ seg, int nseg, int64_t blk_base, int64_t S,
             uint32_t nsb, const double2* __restrict__ tab_gl, const double* __restrict__ tab_lg,
             double* __restrict__ loglik, unsigned long long* __restrict__ nerr, int* __restrict__ cold_flag)
{
  __shared__ double t_a[kEmitTasks];   // min(x, y); overwritten by the result
  __shared__ double t_b[kEmitTasks];   // max(x, y)
  __shared__ double t_r[kEmitTasks];   // min/max
  __shared__ uint32_t t_i[kEmitTasks]; // where the task's tabulated terms are: byte offset / 8 into tab_lg (= / 16 into tab_gl);
                                       // bit 31: x = a1 + obs, the tabulated argument, is the LARGER one; 0xffffffff: not tabulated
  __shared__ double s_logt[ED_PM_LOGT_N * 3];   // the portable log's table (3 KB), see edsf::plog_pos
  __shared__ int n_front, n_back;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // The batch's exons are cut into nseg segments of whole chromosomes (job order); seg[3*i .. 3*i+2] = (first
  // workgroup, first exon, end exon) of segment i.  A workgroup owns a tile of kEmitRows exons x 64 samples (a wave
  // reads 64 consecutive samples of one exon; no per-cell division); inside a segment the workgroups are numbered
  // exon-block major over nsb = ceil(S / 64) sample blocks.  This launch covers workgroups blk_base ..
  // blk_base + gridDim.x - 1 of that numbering; a short uniform search finds the workgroup's segment.
  const int64_t blk = (int64_t)blockIdx.x + blk_base;
  // the segment: the last one whose first workgroup is <= blk (the table is sorted).  Lane i looks at segment i and a
  // ballot counts them -- one load latency instead of a chain of up to nseg dependent scalar loads; the triple of the
  // segment found is then read off its lane.
  int si;
  int64_t seg_first, seg_e0, e_end;
  {
    const bool has = lane < nseg;
    const int64_t f = has ? seg[3 * lane] : 0, a = has ? seg[3 * lane + 1] : 0, z = has ? seg[3 * lane + 2] : 0;
    si = __popcll(ed_ballot(has && f <= blk)) - 1;       // seg[0] = 0 <= blk: at least one
    if (nseg > 64) while (si + 1 < nseg && seg[3 * (si + 1)] <= blk) ++si;
    if (si < 64) {
      auto pick = [&](int64_t v) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, si);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), si);
        return (int64_t)(((uint64_t)hi << 32) | lo);
      };
      seg_first = pick(f); seg_e0 = pick(a); e_end = pick(z);
    } else {
      seg_first = seg[3 * si]; seg_e0 = seg[3 * si + 1]; e_end = seg[3 * si + 2];
    }
  }
  const uint32_t local = (uint32_t)(blk - seg_first);
  uint32_t eb, sb;
  if (nsb >= 8) {
    // XCD-aware numbering (workgroup i runs on XCD i % 8; every segment starts at a multiple of 8): XCD x works on
    // sample block x + 8 r only, and on ONE r at a time -- kEmitRun exon blocks of round r, then of round r + 1,
    // ... -- so that the tables of one sample block (~2.3 MB of hot lines) are what its L2 holds.  Workgroups
    // whose (exon block, sample block) falls outside the segment exit at once.
    const uint32_t nsg = (nsb + 7) / 8, per_super = kEmitRun * 8 * nsg;
    const uint32_t sup = local / per_super, idx = local - sup * per_super;
    const uint32_t r = idx / (kEmitRun * 8), rem = idx % (kEmitRun * 8);
    eb = sup * kEmitRun + rem / 8;
    sb = (rem % 8) + 8 * r;
    if (sb >= nsb || seg_e0 + (int64_t)eb * kEmitRows >= e_end) return;   // uniform: before any barrier
  } else {
    eb = local / nsb;
    sb = local - eb * nsb;
  }
  // the wave's exon is wave-uniform (said explicitly: the compiler cannot know that tid >> 6 is), the sample is block + lane:
  // every address below is a scalar base plus a 32-bit lane offset -- no 64-bit vector address arithmetic
  const int wrow = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t e_first = seg_e0 + (int64_t)eb * kEmitRows + wrow;
  const int64_t s0 = (int64_t)sb * 64;
  int slot[kEmitCells * 3];
  static_assert(kEmitCells == 1, "the slot allocation below handles one cell per thread");
  // Everything this thread reads from memory is requested here, in one go and ahead of the workgroup's first barrier:
  // the counts (HBM), the per-sample shape parameters and flags (L2) and the logarithm's table (L2) then cost the
  // workgroup ONE exposed latency at its start instead of three in a row.
  const int32_t blk_samples = (int32_t)((S - s0 < 64) ? (S - s0) : 64);
  const bool row_in = e_first < e_end;                          // scalar
  const bool live = row_in && (lane < blk_samples);
  // Buffer addressing (SGPR resource + 32-bit lane offset + scalar row offset): no 64-bit vector address arithmetic, and
  // the hardware's range check stands in for predication -- lanes beyond the sample block's end, or a whole wave beyond the
  // segment's last exon, read zeros (and are masked out of the results by `live`).
  const uint32_t l4 = (uint32_t)lane * 4u, l8 = (uint32_t)lane * 8u;
  // streamed once (aux 2 = nt): keep the counts (and the likelihood rows below) from evicting the tables out of L2
  const int32_t obs = __builtin_amdgcn_raw_buffer_load_b32(ed_rsrc(test + (e_first * S + s0), row_in ? blk_samples * 4 : 0), l4, 0, 2);
  const int32_t nref = __builtin_amdgcn_raw_buffer_load_b32(ed_rsrc(ref + (e_first * S + s0), row_in ? blk_samples * 4 : 0), l4, 0, 2);
  // consts [9][S] and flags [3][S]: one resource each from the block's first sample, the row picked by the scalar offset
  // (rows are S * 8 bytes apart: 9 S * 8 < 2^31 is checked by ed_batch_create).  The range check covers lane offset +
  // scalar offset against the resource's size, so lanes beyond the sample block read the NEXT row's first samples on all
  // rows but the last (d_consts / d_cflags carry 64 elements of padding for that one): harmless -- such lanes are not
  // `live`, create no task and store nothing.
  const __amdgpu_buffer_rsrc_t rc = ed_rsrc(consts + s0, (int32_t)((8 * S + 64) * 8));
  const __amdgpu_buffer_rsrc_t rf = ed_rsrc(cflags + s0, (int32_t)((2 * S + 64) * 4));
  const int32_t rowb = (int32_t)(S * 8);
  double pa1[3], pa2[3];
  int pcf[3];
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    pa1[st] = ed_buf_f64(rc, l8, (st * 3 + 0) * rowb);
    pa2[st] = ed_buf_f64(rc, l8, (st * 3 + 1) * rowb);
    pcf[st] = __builtin_amdgcn_raw_buffer_load_b32(rf, l4, st * (rowb / 2), 0);
  }
  if (tid == 0) { n_front = 0; n_back = 0; }
  {
    // 384 doubles: one per thread, and a second one for the first half of the threads
    const double* __restrict__ flat = &k_logt_rows[0][0];
    const double v0 = flat[tid];
    const double v1 = (tid < ED_PM_LOGT_N * 3 - kEmitBlock) ? flat[kEmitBlock + tid] : 0.0;
    s_logt[tid] = v0;
    if (tid < ED_PM_LOGT_N * 3 - kEmitBlock) s_logt[kEmitBlock + tid] = v1;
  }
  static_assert(ED_PM_LOGT_N * 3 > kEmitBlock && ED_PM_LOGT_N * 3 <= 2 * kEmitBlock, "two loads per thread cover the table");
  __syncthreads();
  // ---- phase 1: classify and scatter the tasks ----
  int nflag = 0;
  {
    const int32_t tot = obs + nref;   // as.integer(reference + test), R/class_definition.R:187
    const double dobs = (double)obs, dtot = (double)tot;
    // A cell without reads: a1 + 0 and (a2 + 0) - 0 are a1 and a2 themselves, so the reference's second log-Beta
    // call repeats the per-sample one bit for bit (same value, same GSL error) -- no task, the result is c - c
    // (exactly +0; NaN if c is not finite).  ~14 % of the exons of the bundled exome data have no reads.
    const bool empty = live && (obs | tot) == 0;
    const bool work = live && !empty;
    double tmn[3], tmx[3], tr[3];
    bool xl[3], posv[3];
    bool inr = true;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const double x = pa1[st] + dobs;                              // src/CNV_estimate.cpp:49
      const double y = (pa2[st] + dtot) - dobs;
      const bool pos = (x > 0.0 && y > 0.0);
      // both arguments positive numbers (possibly +inf): IEEE minNum / maxNum are the reference's GSL_MIN / GSL_MAX
      tmn[st] = __builtin_fmin(x, y);
      tmx[st] = __builtin_fmax(x, y);
      xl[st] = x > y;
      posv[st] = pos;
      inr = inr && (!pos || !work || (tmn[st] >= 1e-100 && tmx[st] <= 1e100));
    }
    // min/max: a correctly rounded quotient.  When every task of the wave has 1e-100 <= min <= max <= 1e100 the quotient is a
    // normal number in [1e-200, 1] and the 8-instruction form gives the bits of '/' (edsf::fdiv); anything else takes '/'.
    if (ed_ballot(!inr) == 0ull) {
#pragma unroll
      for (int st = 0; st < 3; ++st) tr[st] = edsf::fdiv(tmn[st], tmx[st]);
    } else {
#pragma unroll
      for (int st = 0; st < 3; ++st) tr[st] = tmn[st] / tmx[st];
    }
    unsigned long long mf[3], mb[3];
    bool cold_any = false;
    bool frontv[3];
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const bool front = work && posv[st] && (tr[st] < 0.2);
      const bool back = work && posv[st] && !front;
      // A non-positive or NaN argument (phi >= 1, expected outside (0, 1), negative counts) takes the reference's general
      // route through gsl_sf_lngamma_sgn_e's reflection / singular branches: a deep, register-hungry call tree.  This kernel
      // does not contain it (its register allocation would be the callee's: 102 instead of 73, two waves per SIMD lost);
      // such a task is left to k_emit_cold, which runs after the group's launches when this flag is up.
      const bool cold = work && !posv[st];
      cold_any |= cold;
      mf[st] = ed_ballot(front);
      mb[st] = ed_ballot(back);
      frontv[st] = front;
      slot[st] = empty ? -2 : (cold ? -3 : ((front || back) ? 0 : -1));
    }
    // GSL error events of the per-sample constants (rare: one branch for the wave)
    if (ed_ballot(live && (pcf[0] | pcf[1] | pcf[2]) != 0) != 0ull)
      nflag = live ? (pcf[0] + pcf[1] + pcf[2]) * (empty ? 2 : 1) : 0;
    if (ed_ballot(cold_any) != 0ull) { if (cold_any) *cold_flag = 1; }
    // wave-aggregated slot allocation: ONE pair of LDS atomics per wave for its (up to) 192 tasks
    const int nf0 = __popcll(mf[0]), nf1 = __popcll(mf[1]), nb0 = __popcll(mb[0]), nb1 = __popcll(mb[1]);
    int basef = 0, baseb = 0;
    if (lane == 0) {
      basef = atomicAdd(&n_front, nf0 + nf1 + __popcll(mf[2]));
      baseb = atomicAdd(&n_back, nb0 + nb1 + __popcll(mb[2]));
    }
    basef = __builtin_amdgcn_readfirstlane(basef);
    baseb = __builtin_amdgcn_readfirstlane(baseb);
    // the task's table entry: (st * kEmitTab + obs) * S + s, in units of one tab_lg element (3 * kEmitTab * S < 2^28, ed_batch_create)
    const uint32_t ti0 = (uint32_t)obs * (uint32_t)S + (uint32_t)(s0 + lane);
    const bool in_tab = (unsigned)obs < (unsigned)kEmitTab;
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      if (slot[st] >= 0) {
        const int offf = basef + (st > 0 ? nf0 : 0) + (st > 1 ? nf1 : 0);
        const int offb = (kEmitTasks - 1) - (baseb + (st > 0 ? nb0 : 0) + (st > 1 ? nb1 : 0));
        const int sl = frontv[st] ? offf + ed_rank(mf[st]) : offb - ed_rank(mb[st]);
        // the gather itself is done by whichever thread evaluates the task: issued at the top of the route, its
        // result is needed ~100 instructions later, so the latency hides behind the task's own arithmetic.
        // Both routes only use symmetric expressions of (x, y): the task is (min, max) + which of them x = a1 + obs is
        uint32_t ti = in_tab ? (ti0 + (uint32_t)(st * kEmitTab) * (uint32_t)S) | (xl[st] ? 0x80000000u : 0u) : 0xffffffffu;
        t_a[sl] = tmn[st]; t_b[sl] = tmx[st]; t_r[sl] = tr[st]; t_i[sl] = ti;
        slot[st] = sl;
      }
    }
  }
  __syncthreads();
  // ---- phase 2: evaluate; slots [0,nf) take the ratio route, slots [kEmitTasks-nb, kEmitTasks) the rest ----
  const int nf = n_front, nb = n_back;
  // the tables through buffer resources: scalar base + 32-bit byte offset (one shift), no 64-bit vector address arithmetic
  const __amdgpu_buffer_rsrc_t rgl = ed_rsrc(tab_gl, (int32_t)(0x7fffffff));
  const __amdgpu_buffer_rsrc_t rlg = ed_rsrc(tab_lg, (int32_t)(0x7fffffff));
#pragma unroll 1
  for (int r = 0; r < kEmitCells * 3; ++r) {
    const int sl = r * kEmitBlock + tid;
    if (sl < nf) {
      const uint32_t ti = t_i[sl];
      double2 gl = make_double2(ed_pm_nan(), ed_pm_nan());
      if (ti != 0xffffffffu) {
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rgl, (ti & 0x0fffffffu) << 4, 0, 0);
        gl.x = __hiloint2double((int)q.y, (int)q.x);
        gl.y = __hiloint2double((int)q.w, (int)q.z);
      }
      if (ti & 0x80000000u) gl.x = -gl.x;      // Gamma*(max) handed in (edsf::lnbeta_ratio_pre)
      t_a[sl] = edsf::lnbeta_ratio_pre(t_a[sl], t_b[sl], t_r[sl], gl.x, gl.y, s_logt);
    } else if (sl >= kEmitTasks - nb) {
      const double mn = t_a[sl], mx = t_b[sl];
      const uint32_t ti = t_i[sl];
      const bool x_is_max = (ti & 0x80000000u) != 0u;
      const double lgx = (ti != 0xffffffffu) ? ed_buf_f64(rlg, (ti & 0x0fffffffu) << 3, 0) : ed_pm_nan();
      // lgamma(x) + lgamma(y) - lgamma(x + y) is symmetric in (x, y) (IEEE addition commutes): x is min or max as the flag says
      t_a[sl] = edsf::lnbeta_general_pre(x_is_max ? mx : mn, x_is_max ? mn : mx, lgx, s_logt);
    }
  }
  // the per-sample constants of phase 3, requested before the barrier: their latency is spent waiting for the other waves
  double pc[3];
#pragma unroll
  for (int st = 0; st < 3; ++st) pc[st] = ed_buf_f64(rc, l8, (st * 3 + 2) * rowb);
  __syncthreads();
  // ---- phase 3: gather, subtract the per-sample constant, store ----
  if (live) {
    // the three rows [e][st][s0 ..] of this exon: one resource, rows S * 8 bytes apart
    const __amdgpu_buffer_rsrc_t rl = ed_rsrc(loglik + (e_first * 3 * S + s0), (int32_t)((2 * S + blk_samples) * 8));
#pragma unroll
    for (int st = 0; st < 3; ++st) {
      const double c = pc[st];
      const int sl = slot[st];
      if (sl == -3) continue;                       // k_emit_cold writes this value
      const double v = t_a[sl < 0 ? 0 : sl];
      ed_buf_store_f64_nt(rl, l8, st * rowb, (sl == -2 ? c : v) - c);
    }
  }
  if (nflag) atomicAdd(nerr, (unsigned long long)nflag);
}

// The tasks k_emit_batch left out: (cell, state) pairs with a non-positive or NaN log-Beta argument.  Launched after the
// emission launches of every overlap group (jobs j0 .. j1-1 of the segment table) with a small fixed grid; returns at
// once unless k_emit_batch raised the flag, else walks the group's cells and evaluates exactly those pairs with the full
// general route of the reference (edsf::lnbeta_cold: reflection, lngamma_sgn_sing, sign rule, error sites).
__global__ void __launch_bounds__(256)
k_emit_cold(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, const double* __restrict__ consts,
            const int64_t* __restrict__ seg, int j0, int j1, int64_t S, double* __restrict__ loglik,
            unsigned long long* __restrict__ nerr, const int* __restrict__ cold_flag)
{
  if (*cold_flag == 0) return;
  int nflag = 0;
  for (int j = j0; j < j1; ++j) {
    const int64_t e0 = seg[3 * j + 1], e1 = seg[3 * j + 2];
    const int64_t ncell = (e1 - e0) * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e = e0 + i / S, s = i % S;
      const int32_t obs = test[e * S + s];
      const int32_t tot = obs + ref[e * S + s];
      if (obs == 0 && tot == 0) continue;      // a cell without reads is c - c in k_emit_batch, whatever c is
      for (int st = 0; st < 3; ++st) {
        const double x = consts[(st * 3 + 0) * S + s] + (double)obs;
        const double y = (consts[(st * 3 + 1) * S + s] + (double)tot) - (double)obs;
        if (x > 0.0 && y > 0.0) continue;
        int flag = 0;
        const double v = edsf::lnbeta_cold(x, y, &flag);
        loglik[(e * 3 + st) * S + s] = v - consts[(st * 3 + 2) * S + s];
        nflag += flag;
      }
    }
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

// Error sites of the six gsl_sf_lnbeta calls of every row, in the reference's call order (edsf::lnbeta_sites):
// codes[6 i + 2 st + 0] for lnbeta(a1 + obs, a2 + tot - obs), [.. + 1] for lnbeta(a1, a2)  (src/CNV_estimate.cpp:49)
__global__ void __launch_bounds__(kEmitBlock)
k_emit_rows_sites(const double* __restrict__ phi, const double* __restrict__ expected, const int32_t* __restrict__ total,
                  const int32_t* __restrict__ observed, int64_t n, double mixture, uint16_t* __restrict__ codes)
{
  const int64_t i = (int64_t)blockIdx.x * kEmitBlock + threadIdx.x;
  if (i >= n) return;
  const double e = expected[i];
  const double sd = __builtin_sqrt((phi[i] * e) * (1. - e));
  double ep[3];
  state_props(e, mixture, ep);
  const int32_t tot = total[i], obs = observed[i];
#pragma unroll 1
  for (int st = 0; st < 3; ++st) {
    double a1, a2;
    shape_params(ep[st], sd, a1, a2);
    codes[i * 6 + st * 2 + 0] = (uint16_t)edsf::lnbeta_sites(a1 + (double)obs, (a2 + (double)tot) - (double)obs);
    codes[i * 6 + st * 2 + 1] = (uint16_t)edsf::lnbeta_sites(a1, a2);
  }
}

// Self-check of the batched emissions (ed_batch_verify_emissions): every cell is evaluated a second time with the
// straight per-cell arithmetic of the reference's loop -- log B(a1 + obs, a2 + tot - obs) - log B(a1, a2) per state as
// src/CNV_estimate.cpp:44-50, :71-81 has it, shape parameters recomputed from (phi, expected), through edsf::lnbeta:
// no tables, no binning, no LDS -- and compared bit for bit with what k_emit_batch left in the likelihood matrix, on the
// device.  A thread walks kVerifyRun consecutive exons of one sample, so log B(a1, a2) -- the same call with the same
// arguments for every exon of the sample -- is evaluated once per thread and state.  NaN matches NaN.
// counters[0] = mismatching values, [1] = values compared, [2] = mismatches recorded in `first` (up to cap).
constexpr int kVerifyRun = 8;
__global__ void __launch_bounds__(kEmitBlock)
k_emit_verify(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, const double* __restrict__ phi,
              const double* __restrict__ expected, double mixture, int64_t E, int64_t S, const double* __restrict__ loglik,
              unsigned long long* __restrict__ counters, ed_emit_mismatch* __restrict__ first, int64_t cap, int64_t ce, int64_t cs)
{
  const int64_t s_raw = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t e0 = (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * (kEmitBlock / 64) + (threadIdx.x >> 6)) * kVerifyRun;
  const bool live_s = s_raw < S;
  const int64_t s = live_s ? s_raw : 0;      // idle lanes shadow sample 0 and count nothing
  const double ex = expected[s];
  const double sd = __builtin_sqrt((phi[s] * ex) * (1. - ex));
  double ep[3];
  state_props(ex, mixture, ep);
  int bad = 0, ncell = 0;
#pragma unroll 1
  for (int st = 0; st < 3; ++st) {
    double a1, a2;
    shape_params(ep[st], sd, a1, a2);
    int f2 = 0;
    const double v0 = edsf::lnbeta(a1, a2, &f2);
#pragma unroll 1
    for (int k = 0; k < kVerifyRun; ++k) {
      const int64_t e = e0 + k;
      const bool live = live_s && e < E;
      const int64_t ec = e < E ? e : E - 1;
      const int32_t obs = test[ec * ce + s * cs];     // (ce, cs) = (S, 1): counts [E][S]; (1, E): [S][E]
      const int32_t tot = obs + ref[ec * ce + s * cs];
      int f1 = 0;
      const double v1 = edsf::lnbeta(a1 + (double)obs, (a2 + (double)tot) - (double)obs, &f1);
      const double want = v1 - v0;
      const double got = loglik[(ec * 3 + st) * S + s];
      const bool same = (__double_as_longlong(want) == __double_as_longlong(got)) || (want != want && got != got);
      if (live && st == 0) ++ncell;
      if (live && !same) {
        ++bad;
        const unsigned long long q = atomicAdd(&counters[2], 1ull);
        if ((int64_t)q < cap) {
          ed_emit_mismatch m;
          m.exon = e; m.sample = s; m.state = st; m.observed = obs; m.total = tot; m.pad_ = 0;
          m.got = got; m.want = want;
          first[q] = m;
        }
      }
    }
  }
  // one pair of atomics per wave
  int wbad = bad, wcell = ncell;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    wbad += __shfl_xor(wbad, d, 64);
    wcell += __shfl_xor(wcell, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (wbad) atomicAdd(&counters[0], (unsigned long long)wbad);
    if (wcell) atomicAdd(&counters[1], 3ull * (unsigned long long)wcell);
  }
}

// ---- Viterbi --------------------------------------------------------------------------------

// ---- batched chains -------------------------------------------------------------------------
// A chain (one sample x one chromosome) is strictly sequential: the max-plus recurrence is evaluated in
// the reference's association order, so there is no scan over exons.  The parallelism inside a chain
// is the 3 target states: a chain is run by a QUAD of lanes (lane j of the quad owns HMM state j; the
// 4th lane shadows state 0), 16 chains per wave.  Per step a lane needs the three previous scores --
// two DPP quad-broadcasts away -- its own emission and its own row of log-transitions:
//       cand_k = (e_j + v_k) + lt[j][k],  k = 0,1,2   (src/hmm.cpp:79; first strict maximum wins, :81-84)
// This cuts the per-step instruction count ~2.5x against one-lane-per-chain and quadruples the waves
// that hide each other's latency; the chain of the longest chromosome is the critical path.
//   loglik [E][3][S]   S4 column order (deletion, normal, duplication): state j reads column {1,0,2}[j];
//                      fetched kVitTile steps ahead (register double buffer)
//   lt4    [(E+C)][4][2]  per exon gap and per quad lane the pair (lt[j][1], lt[j][2]) of CallCNVs' matrix:
//                      lane 0/3: (A, A), lane 1: (B, C), lane 2: (C, B);  lt[j][0] is a per-lane constant
//                      (c0 = log(1-t) for state 0, c1 = log(t/2) otherwise).  gap of padded step i of
//                      chromosome c: lo + c + (i-1).  All quads of a wave read the same 64 bytes.
//   bpq    [words][S][4]  back-pointers: each lane keeps the 2-bit pointers of ITS state, 16 steps per
//                      32-bit word, one coalesced 4-byte store per lane per 16 steps
//   ppath  [words][S]  Viterbi states written by the trace-back, 16 exons x 2 bits per word (one store per
//                      16 steps keeps the trace-back's loads from queueing behind byte stores);
//                      k_path_expand turns it into the byte-per-exon path [E][S] of the interface
// The two dummy observations of CallCNVs (R/class_definition.R:364) are implicit: the chain starts from
// (0,-inf,-inf) (src/hmm.cpp:48-52; the first dummy row is never read) and ends with one extra step
// whose emissions are (-100, 0, -100).
constexpr int kVitTile = 16;      // steps per back-pointer / packed-state word

constexpr int kVitChains = 16;    // chains per wave

template <int K>
__device__ __forceinline__ double quad_bcast(double x)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), K * 0x55, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), K * 0x55, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ int quad_bcast_i(int x)
{
  return __builtin_amdgcn_update_dpp(0, x, K * 0x55, 0xf, 0xf, true);
}

// One forward step of the lane's state; returns the 2-bit back-pointer.
// The reference scans k = 0,1,2 keeping the first strict maximum, starting from -inf (src/hmm.cpp:78-85), and
// forces the pointer to 0 when the emission is -inf (:87).  That is: v' = the largest non-NaN candidate or
// -inf, pointer = the first k attaining it, 0 if nothing exceeds -inf.  IEEE maxNum (v_max_f64: the non-NaN
// operand wins) seeded with -inf computes exactly that value in 3 dependent instructions instead of 3 x
// (compare, select, select); the pointer falls out of two equality tests off the critical path.  Candidates
// are < 0 here (emissions <= 0, log-transitions < 0), so +0/-0 ties cannot arise.  e = -inf makes every
// candidate -inf or NaN, hence best = -inf and the pointer 0 without a separate test.
__device__ __forceinline__ unsigned vit_step_q(double& v, double e, double t0, double t1, double t2)
{
  const double NI = -HUGE_VAL;
  const double v0 = quad_bcast<0>(v), v1 = quad_bcast<1>(v), v2 = quad_bcast<2>(v);
  const double c0 = (e + v0) + t0;
  const double c1 = (e + v1) + t1;
  const double c2 = (e + v2) + t2;
  // maxNum ignores NaN operands and NI is never NaN, so any grouping gives the same value; this one is two deep
  const double best = __builtin_fmax(__builtin_fmax(NI, c0), __builtin_fmax(c1, c2));
  unsigned fw = (c0 == best) ? 0u : ((c1 == best) ? 1u : 2u);
  if (best == NI) fw = 0u;
  v = best;
  return fw;
}

// Register budget: a Viterbi wave shares its SIMD with emission waves (96 registers each; ed_batch_run overlaps
// the two kernels), so its own allocation decides how many of them stay resident beside it: 154 registers
// (no scratch) leave room for three.  Measured on MI355X: a 128-register build (four) is no faster.
#ifndef ED_VIT_OCC
#define ED_VIT_OCC 3
#endif
__global__ void __launch_bounds__(kWave, ED_VIT_OCC)
k_viterbi(const double* __restrict__ loglik, const double* __restrict__ lt4, double c0, double c1,
          const int32_t* __restrict__ chrom_off, const int64_t* __restrict__ word_off, int64_t S, int32_t C,
          uint32_t* __restrict__ bpq, uint8_t* __restrict__ last, const int32_t* __restrict__ job_off,
          const int32_t* __restrict__ job_chrom, int job_base)
{
  const int lane = threadIdx.x;
  const int j = lane & 3;
  const int64_t s_raw = (int64_t)blockIdx.x * kVitChains + (lane >> 2);
  const bool live = s_raw < S;
  const int64_t s = live ? s_raw : S - 1;   // idle quads shadow the last sample (loads stay in bounds, no stores)
  __shared__ double2 lds_lt[2][kVitTile][4];
  // A workgroup runs a JOB: one or more whole chromosomes, one after the other.  The host packs the
  // chromosomes into jobs of about the longest chromosome's length so that, when the batch has fewer
  // waves than the chip has SIMDs, every wave has a SIMD to itself and the makespan is one long chain.
  const int job = job_base + (int)blockIdx.y;
  for (int jc = job_off[job]; jc < job_off[job + 1]; ++jc) {
  const int c = job_chrom[jc];
  const int64_t lo = chrom_off[c], hi = chrom_off[c + 1];
  const int64_t m = hi - lo;
  if (m <= 0) continue;
  __syncthreads();   // the previous chromosome's readers are done with lds_lt
  const int col = (j == 1) ? 0 : ((j == 2) ? 2 : 1);
  const int64_t estride = 3 * S;                                  // doubles between consecutive exons
  const double2* __restrict__ ltp = reinterpret_cast<const double2*>(lt4) + (lo + c) * 4 + j;  // step i: ltp[i * 4]
  const int64_t wstride = S * 4;
  const double t0 = (j == 0 || j == 3) ? c0 : c1;
  double v = (j == 0 || j == 3) ? 0.0 : -HUGE_VAL;

  // ---- forward pass ----
  // Emissions live in a register ring of kRing steps: step i's value sits in er[i % kRing] and, as soon as
  // it is consumed, the slot is re-loaded with step i + kRing (clamped to the chromosome's last exon, a
  // scalar min), so loads are always kRing steps ahead of their use and nothing is copied.  Addresses are a
  // wave-uniform base (scalar registers) plus one 32-bit lane offset.  The log-transition rows of tile t+1
  // are fetched by the wave as one 16-byte load per lane at the start of tile t, parked in LDS at the end of
  // tile t and read back per step (every quad reads the same 64 bytes, its lane its own 16); the table
  // carries padding, so this prefetch may run past the chromosome's last gap.  The loop is unrolled over two
  // tiles so that ring slots and LDS buffers are compile-time; the last (possibly partial) tiles take the
  // guarded instance, whose guards are scalar branches.
  constexpr int kRing = 2 * kVitTile;
  const char* __restrict__ emb = reinterpret_cast<const char*>(loglik + lo * 3 * S);       // wave-uniform
  const uint32_t eoff = (uint32_t)((col * S + s) * 8);                                     // lane part, bytes (< 2^32)
  const int64_t ebytes = estride * 8;                                                      // bytes between exons
  char* __restrict__ bpb = reinterpret_cast<char*>(bpq + word_off[c] * S * 4);            // wave-uniform
  const uint32_t boff = (uint32_t)((s * 4 + j) * 4);
  const int64_t wbytes = wstride * 4;
  const double2* __restrict__ ltw = reinterpret_cast<const double2*>(lt4) + (lo + c) * 4;  // wave-level view
  const int m32 = (int)m;   // chromosome lengths are int32 (chrom_off); 32-bit so that the clamp is one s_min_i32
  // scalar base + 32-bit lane offset: the load needs no vector address arithmetic
  auto em_ld = [&](const char* row) { return *reinterpret_cast<const double*>(row + eoff); };
  auto em_at = [&](int i) { return em_ld(emb + (int64_t)(i < m32 ? i : m32 - 1) * ebytes); };
  double er[kRing];
  double2 stg = ltw[lane];
#pragma unroll
  for (int k = 0; k < kRing; ++k) er[k] = em_at(k);
  (&lds_lt[0][0][0])[lane] = stg;
  __syncthreads();
  const char* nxt = emb + (int64_t)kRing * ebytes;   // row of the next re-load in the unclamped (safe) region
  auto tile = [&](auto par, auto full, int t, int nsteps) {
    constexpr int P = decltype(par)::value;
    constexpr bool kFull = decltype(full)::value;   // full tile whose re-loads (16t+k+kRing) all lie inside the chromosome
    stg = ltw[(t + 1) * 64 + lane];
    uint32_t w = 0;
#pragma unroll
    for (int g4 = 0; g4 < kVitTile / 4; ++g4) {
      double2 lrow[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) lrow[k] = lds_lt[P][g4 * 4 + k][j];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int kk = g4 * 4 + k;
        if (kFull || kk < nsteps) {
          unsigned fw = vit_step_q(v, er[P * kVitTile + kk], t0, lrow[k].x, lrow[k].y);
          // pin the pointer to a register here: otherwise the compiler defers all 16 pointer computations of the
          // tile to the store below (keeping 3 doubles per step alive) and turns the shifts into constant tables
          asm("" : "+v"(fw));
          w |= fw << (2 * kk);
          if (kFull) {
            er[P * kVitTile + kk] = em_ld(nxt);
            nxt += ebytes;
          } else {
            er[P * kVitTile + kk] = em_at(t * kVitTile + kk + kRing);
          }
        }
      }
    }
    // idle quads shadow sample S-1: same value to the same address
    *reinterpret_cast<uint32_t*>(bpb + (int64_t)t * wbytes + boff) = w;
    (&lds_lt[P ^ 1][0][0])[lane] = stg;
    __syncthreads();
  };
  {
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int ntile = (m32 + kVitTile - 1) / kVitTile;
    const int tsafe = (m32 >= kRing ? (m32 - kRing) / kVitTile : 0) & ~1;   // tiles t < tsafe never re-load past exon m-1
    int t = 0;
    for (; t < tsafe; t += 2) {
      tile(I0{}, std::true_type{}, t, kVitTile);
      tile(I1{}, std::true_type{}, t + 1, kVitTile);
    }
    for (; t < ntile; ++t) {   // the last few tiles (at most four): clamped re-loads, guarded steps
      const int nst = (m32 - t * kVitTile < kVitTile) ? (m32 - t * kVitTile) : kVitTile;
      if (t & 1) tile(I1{}, std::false_type{}, t, nst);
      else tile(I0{}, std::false_type{}, t, nst);
    }
  }
  // dummy last observation (R/class_definition.R:364): only state 0's back-pointer is ever used
  int st;
  {
    const double2 l = ltp[m * 4];
    const double e = (j == 1) ? 0.0 : -100.0;
    const unsigned fw = vit_step_q(v, e, t0, l.x, l.y);
    st = quad_bcast_i<0>((int)fw);
  }
  // the trace-back is data-parallel and lives in k_tb_maps / k_tb_chain / k_tb_paths
  if (live && j == 0) last[(int64_t)c * S + s] = (uint8_t)st;
  }   // chromosomes of the job
}


// ---- trace-back (src/hmm.cpp:95-100), data-parallel ------------------------------------------------------
// tb[i-1] = from[i][tb[i]] is a composition of maps {0,1,2} -> {0,1,2}; composition is associative and the
// objects are small integers, so regrouping it is exact.  A chromosome's back-pointers are stored 16 steps per
// word (one word per state); word w therefore defines a map T_w from the state of its last exon to the state
// of the last exon of word w-1:
//   k_tb_maps   every (sample, word) in parallel: the three images of T_w (6 bits)
//   k_tb_chain  one lane per chain: walk the ~m/16 maps from the last word down, recording the state of each
//               word's last exon (a 3-instruction dependent step per 16 exons)
//   k_tb_paths  every (sample, word) in parallel: replay the 16 steps from the now known state, write the
//               packed states, the byte-per-exon path of the interface, and count the calls (a call ends
//               wherever a run of a non-zero state ends, src/hmm.cpp:109-121; integer atomics: exact)
__device__ __forceinline__ uint32_t tb_pick(const uint4& w, int st) { return st == 0 ? w.x : (st == 1 ? w.y : w.z); }

__global__ void __launch_bounds__(256)
k_tb_maps(const uint32_t* __restrict__ bpq, const int32_t* __restrict__ chrom_off, const int64_t* __restrict__ word_off,
          int64_t S, const int32_t* __restrict__ job_off, const int32_t* __restrict__ job_chrom, int job_base,
          uint8_t* __restrict__ maps)
{
  const int64_t s = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t w = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  const int job = job_base + (int)blockIdx.z;
  if (s >= S) return;
  for (int jc = job_off[job]; jc < job_off[job + 1]; ++jc) {
    const int c = job_chrom[jc];
    const int64_t m = chrom_off[c + 1] - chrom_off[c];
    if (w * kVitTile >= m) continue;
    const int n = (int)((m - w * kVitTile < kVitTile) ? (m - w * kVitTile) : kVitTile);
    const uint4 bw = reinterpret_cast<const uint4*>(bpq)[(word_off[c] + w) * S + s];
    int x0 = 0, x1 = 1, x2 = 2;
#pragma unroll
    for (int k = kVitTile - 1; k >= 0; --k) {
      if (k < n) {
        x0 = (int)((tb_pick(bw, x0) >> (2 * k)) & 3u);
        x1 = (int)((tb_pick(bw, x1) >> (2 * k)) & 3u);
        x2 = (int)((tb_pick(bw, x2) >> (2 * k)) & 3u);
      }
    }
    maps[(word_off[c] + w) * S + s] = (uint8_t)(x0 | (x1 << 2) | (x2 << 4));
  }
}

// States are left packed 16 words per 32-bit word (ent, rows (word_off[c] >> 4) + c + (w >> 4)): one store per 16
// steps, issued after the next group's loads, so the dependent walk never waits on a store.
__global__ void __launch_bounds__(kWave)
k_tb_chain(const int32_t* __restrict__ chrom_off, const int64_t* __restrict__ word_off, int64_t S,
           const int32_t* __restrict__ job_off, const int32_t* __restrict__ job_chrom, int job_base,
           const uint8_t* __restrict__ last, const uint8_t* __restrict__ maps, uint32_t* __restrict__ ent)
{
  const int64_t s = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const int job = job_base + (int)blockIdx.y;
  if (s >= S) return;
  for (int jc = job_off[job]; jc < job_off[job + 1]; ++jc) {
    const int c = job_chrom[jc];
    const int64_t m = chrom_off[c + 1] - chrom_off[c];
    if (m <= 0) continue;
    const int nw = (int)((m + kVitTile - 1) / kVitTile);
    const int ng = (nw + 15) / 16;
    const uint8_t* __restrict__ mp = maps + word_off[c] * S + s;              // word w: mp[w * S]
    uint32_t* __restrict__ ep = ent + ((word_off[c] >> 4) + c) * S + s;       // group g: ep[g * S]
    int st = last[(int64_t)c * S + s];
    uint32_t cur[16], nxt[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int w = (ng - 1) * 16 + k;
      cur[k] = (w < nw) ? mp[(int64_t)w * S] : 0u;
    }
    for (int g = ng - 1; g >= 0; --g) {
      if (g > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) nxt[k] = mp[(int64_t)((g - 1) * 16 + k) * S];
      }
      uint32_t pk = 0;
#pragma unroll
      for (int k = 15; k >= 0; --k) {
        if (g * 16 + k < nw) {
          pk |= (uint32_t)st << (2 * k);                  // state of word (16g + k)'s last exon
          st = (int)((cur[k] >> (2 * st)) & 3u);          // ... and of the word before it
        }
      }
      ep[(int64_t)g * S] = pk;
#pragma unroll
      for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
  }
}

__global__ void __launch_bounds__(256)
k_tb_paths(const uint32_t* __restrict__ bpq, const int32_t* __restrict__ chrom_off, const int64_t* __restrict__ word_off,
           int64_t S, int32_t C, const int32_t* __restrict__ job_off, const int32_t* __restrict__ job_chrom, int job_base,
           const uint32_t* __restrict__ ent, uint32_t* __restrict__ ppath, uint8_t* __restrict__ path,
           int32_t* __restrict__ counts)
{
  const int64_t s = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t w = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  const int job = job_base + (int)blockIdx.z;
  if (s >= S) return;
  for (int jc = job_off[job]; jc < job_off[job + 1]; ++jc) {
    const int c = job_chrom[jc];
    const int64_t lo = chrom_off[c], m = chrom_off[c + 1] - lo;
    if (w * kVitTile >= m) continue;
    const int n = (int)((m - w * kVitTile < kVitTile) ? (m - w * kVitTile) : kVitTile);
    const uint4 bw = reinterpret_cast<const uint4*>(bpq)[(word_off[c] + w) * S + s];
    int st = (int)((ent[((word_off[c] >> 4) + c + (w >> 4)) * S + s] >> (2 * (int)(w & 15))) & 3u);   // state of exon 16w + n - 1
    int count = ((w + 1) * kVitTile >= m && st != 0) ? 1 : 0;  // the chromosome's last exon against the dummy end state 0
    uint32_t pw = 0;
    uint8_t* __restrict__ o = path + (lo + w * kVitTile) * S + s;
#pragma unroll
    for (int k = kVitTile - 1; k >= 0; --k) {
      if (k < n) {
        pw |= (uint32_t)st << (2 * k);
        o[(int64_t)k * S] = (uint8_t)st;
        const int prev = (int)((tb_pick(bw, st) >> (2 * k)) & 3u);   // state of exon 16w + k - 1
        if ((k > 0 || w > 0) && prev != st && prev != 0) ++count;
        st = prev;
      }
    }
    ppath[(word_off[c] + w) * S + s] = pw;
    if (count) atomicAdd(&counts[s * C + c], count);
  }
}

}  // namespace

#include "edtab.inc"
#include "edfused.inc"

namespace {

// packed states [words][S] -> path [E][S] (one byte per exon).  thread = (sample, word)
__global__ void __launch_bounds__(256)
k_path_expand(const uint32_t* __restrict__ ppath, const int32_t* __restrict__ chrom_off,
              const int64_t* __restrict__ word_off, int64_t S, uint8_t* __restrict__ path)
{
  const int64_t s = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int c = blockIdx.z;
  const int64_t lo = chrom_off[c], m = chrom_off[c + 1] - lo;
  const int64_t w = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (s >= S || w * kVitTile >= m) return;
  const uint32_t pw = ppath[(word_off[c] + w) * S + s];
  uint8_t* __restrict__ o = path + (lo + w * kVitTile) * S + s;
  const int64_t left = m - w * kVitTile;
#pragma unroll
  for (int k = 0; k < kVitTile; ++k)
    if (k < left) o[k * S] = (uint8_t)((pw >> (2 * k)) & 3);
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

// Call records of every chain (src/hmm.cpp:104-126 + R/class_definition.R:371-372, :409-410).  The
// reference's summary loop, restated on exon indices x = 0..m (x = m is the dummy last observation,
// state 0): when the state changes after a zero, `start` is set; when it changes after a non-zero
// state a call (start, x-1, state, nexons) is pushed.  Quirks kept: `start` is NOT reset when one CNV
// state switches directly to the other (the second call inherits the first one's start), and nexons
// restarts only at a push.  The path is read in its packed form (16 exons per 32-bit word; unused high
// bits of a chain's last word are zero, which reads as the dummy end state).
// A chain is cut into kCallSeg runs of words walked by different waves (one lane per sample walking a whole
// chromosome alone left the GPU to 384 waves: 127 us of latency).  Pass 1 counts the calls each run pushes (bit
// operations only), an LDS prefix turns the counts into the rank of each run's first call, pass 2 writes.  The loop
// state a run inherits: the state of the exon before it (the previous word's top bits); if that is a CNV state,
// `run_begin` and `start` are recovered by walking back to the beginning of that run of non-zero states.
constexpr int kCallSeg = 8;
__global__ void __launch_bounds__(kWave * kCallSeg)
k_calls_fill(const uint32_t* __restrict__ ppath, const int32_t* __restrict__ chrom_off,
             const int64_t* __restrict__ word_off, int64_t S, int32_t C, const int64_t* __restrict__ offsets,
             const int32_t* __restrict__ counts, ed_call* __restrict__ calls, int64_t cap)
{
  __shared__ int seg_n[kCallSeg][kWave];
  const int lane = threadIdx.x, sg = threadIdx.y;
  const int64_t s = (int64_t)blockIdx.x * kWave + lane;
  const int c = blockIdx.y;
  const bool live = s < S;
  const int64_t sc = live ? s : S - 1;
  const int64_t lo = chrom_off[c], hi = chrom_off[c + 1];
  const int64_t m = hi - lo;
  if (m <= 0) return;
  const int32_t todo = live ? counts[sc * C + c] : 0;
  if (!__syncthreads_or(todo > 0)) return;
  const int64_t off = offsets[sc * C + c];
  const uint32_t* __restrict__ pp = ppath + word_off[c] * S + sc;
  const int64_t nw = (m + kVitTile - 1) / kVitTile;
  const int64_t w0 = nw * sg / kCallSeg, w1 = nw * (sg + 1) / kCallSeg;
  auto state_at = [&](int64_t x) -> int { return (int)((pp[(x / kVitTile) * S] >> (2 * (int)(x % kVitTile))) & 3u); };
  const int prev0 = (w0 > 0 && w0 < w1) ? (int)(pp[(w0 - 1) * S] >> (2 * (kVitTile - 1))) : 0;   // exon before the run
  constexpr int kW = 8;
  // ---- pass 1: how many calls does this run push? ----
  int n = 0;
  {
    int prev = prev0;
    for (int64_t wb = w0; wb < w1; wb += kW) {
      uint32_t w[kW];
#pragma unroll
      for (int t = 0; t < kW; ++t) w[t] = (wb + t < w1) ? pp[(wb + t) * S] : 0u;
#pragma unroll
      for (int t = 0; t < kW; ++t) {
        if (wb + t < w1) {
          const uint32_t before = (w[t] << 2) | (uint32_t)prev;
          const uint32_t diff = w[t] ^ before;
          const uint32_t mk = (diff | (diff >> 1)) & 0x55555555u;          // positions whose state differs from the one before
          const uint32_t nz = (before | (before >> 1)) & 0x55555555u;      // ... and the one before is a CNV state: a push
          n += __popc(mk & nz);
          prev = (int)(w[t] >> (2 * (kVitTile - 1)));
        }
      }
    }
    if (w1 == nw && w0 < w1 && prev != 0) ++n;   // a full last word: the dummy end observation closes the run
  }
  seg_n[sg][lane] = live ? n : 0;
  __syncthreads();
  int k = 0;
  for (int q = 0; q < sg; ++q) k += seg_n[q][lane];
  const int kend = k + (live ? n : 0);
  if (!__any(k < kend)) return;
  // ---- pass 2: the summary loop over this run ----
  // Only the positions where the state CHANGES are visited: between two changes the reference's loop does nothing but
  // count, and `nexons` -- reset at every push, incremented on every non-zero exon -- is the length of the run of
  // equal states that ends, x - run_begin.
  int64_t start = -1, run_begin = 0;
  int prev = prev0;
  if (prev0 != 0 && k < kend) {
    int64_t rb = w0 * kVitTile - 1;                       // an exon of the CNV run the boundary cuts
    while (rb > 0 && state_at(rb - 1) == prev0) --rb;
    int64_t y = rb;
    while (y > 0 && state_at(y - 1) != 0) --y;            // a direct switch between CNV states keeps the first `start`
    run_begin = rb;
    start = y;
  }
  auto change = [&](int cur, int64_t x) {   // the state changes from prev to cur at exon x
    if (prev == 0) {
      start = x;
    } else {
      const int64_t r = off + k;
      if (r < cap && k < kend) {
        ed_call rec;
        rec.sample = (int32_t)s;
        rec.chrom = c;
        rec.start_exon = (int32_t)(lo + start);
        rec.end_exon = (int32_t)(lo + x - 1);
        rec.type = prev;
        rec.nexons = (int32_t)(x - run_begin);
        calls[r] = rec;
      }
      ++k;
    }
    run_begin = x;
    prev = cur;
  };
  uint32_t nxt[kW];   // the next kW words are requested before the current ones are walked
#pragma unroll
  for (int t = 0; t < kW; ++t) nxt[t] = (w0 + t < w1) ? pp[(w0 + t) * S] : 0u;
  for (int64_t wb = w0; wb < w1 && __any(k < kend); wb += kW) {
    uint32_t w[kW];
#pragma unroll
    for (int t = 0; t < kW; ++t) w[t] = nxt[t];
#pragma unroll
    for (int t = 0; t < kW; ++t) nxt[t] = (wb + kW + t < w1) ? pp[(wb + kW + t) * S] : 0u;
#pragma unroll
    for (int t = 0; t < kW; ++t) {
      if (wb + t < w1) {
        const uint32_t before = (w[t] << 2) | (uint32_t)prev;        // state of the exon before each position
        const uint32_t diff = w[t] ^ before;
        uint32_t mk = (diff | (diff >> 1)) & 0x55555555u;            // one bit per position whose state differs
        const int64_t x0 = (wb + t) * kVitTile;
        while (mk) {
          const int q = __builtin_ctz(mk) >> 1;
          mk &= mk - 1;
          change((int)((w[t] >> (2 * q)) & 3u), x0 + q);
        }
      }
    }
  }
  if (w1 == nw && prev != 0 && k < kend) change(0, m);   // the dummy last observation closes a run that reaches the end
}

// Decoration of the call table (R/class_definition.R:379-405): per call, BF = sum over its exons of
// (loglik[type] - loglik[normal]), reads.expected = as.integer(sum(total * expected)), reads.observed =
// sum(test), reads.ratio.  R's sum() accumulates in long double; here each sum is carried as a
// double-double (error-free TwoSum), which is at least as accurate, and rounded once.
__device__ __forceinline__ void dd_add(double& hi, double& lo, double x)
{
  const double s = hi + x;
  const double bb = s - hi;
  lo += (hi - (s - bb)) + (x - bb);
  hi = s;
}

// R's signif(x, 3) for finite non-zero x (src/nmath/fprec.c, the e10 in [-max10e, max10e] branch): scale by a
// power of ten so that the value has 3 integer digits, round half to even, scale back
__device__ __forceinline__ double signif3(double x)
{
  if (!(x == x) || x == 0.0 || x - x != 0.0) return x;
  const double ax = fabs(x);
  int e10 = 2 - (int)floor(log10(ax));
  double p10 = 1.0;
  const int n = e10 < 0 ? -e10 : e10;
  for (int i = 0; i < n; ++i) p10 *= 10.0;
  double r = (e10 >= 0) ? rint(ax * p10) / p10 : rint(ax / p10) * p10;
  return x < 0 ? -r : r;
}

// loglik may be NULL (matrix not kept): the two emissions each exon of a call needs are then recomputed
// from the per-sample constants -- the same arithmetic, hence the same bits, as the fused kernel produced.
__global__ void k_call_info(const ed_call* __restrict__ calls, int64_t ncalls, const double* __restrict__ loglik,
                            const double* __restrict__ consts, const int32_t* __restrict__ test,
                            const int32_t* __restrict__ ref, const double* __restrict__ expected, int64_t S,
                            ed_call_info* __restrict__ out, const double* __restrict__ X, int K, const double* __restrict__ beta,
                            int64_t le, int64_t lst, int64_t ls,   // likelihood element (e, st, s) at e * le + st * lst + s * ls
                            int64_t ce, int64_t cs,                // count of (e, s) at e * ce + s * cs
                            int cb)                                // bytes per count: 4, or 2 (ed_batch_set_counts_bits(batch, 16))
{
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ncalls) return;
  const ed_call c = calls[r];
  const int64_t s = c.sample;
  const int col = (c.type == 1) ? 0 : 2;   // likelihood columns: deletion, normal, duplication
  const double p_s = (K >= 0) ? 0.0 : expected[s];
  double bh = 0, bl = 0, eh = 0, el = 0;
  int64_t obs = 0;
  for (int64_t e = c.start_exon; e <= c.end_exon; ++e) {
    const int32_t t = ed_ldc(test, e * ce + s * cs, cb);
    const int32_t tot = t + ed_ldc(ref, e * ce + s * cs, cb);
    double lc, ln;
    if (loglik) {
      lc = loglik[e * le + col * lst + s * ls];
      ln = loglik[e * le + lst + s * ls];
    } else {
      int flag = 0;
      lc = edsf::lnbeta(consts[(col * 3 + 0) * S + s] + (double)t, (consts[(col * 3 + 1) * S + s] + (double)tot) - (double)t, &flag) -
           consts[(col * 3 + 2) * S + s];
      ln = edsf::lnbeta(consts[(1 * 3 + 0) * S + s] + (double)t, (consts[(1 * 3 + 1) * S + s] + (double)tot) - (double)t, &flag) -
           consts[(1 * 3 + 2) * S + s];
    }
    dd_add(bh, bl, lc - ln);
    dd_add(eh, el, (double)tot * ((K >= 0) ? cov_expected(X, K, beta, e, S, s) : p_s));
    obs += t;
  }
  ed_call_info o;
  o.BF_raw = 0.43429448190325182765 * (bh + bl);   // log10(exp(1)) * sum
  o.BF = signif3(o.BF_raw);
  o.reads_expected = (int64_t)(eh + el);             // as.integer(): truncation
  o.reads_observed = obs;
  o.reads_ratio = signif3((double)obs / (double)o.reads_expected);
  out[r] = o;
}

// ---- K5: per-sample beta-binomial fit -------------------------------------------------------
// Exon axis cut into chunks of kFitChunk exons; thread = (sample lane, sub-chunk).  Every pass writes
// per-chunk partial sums [chunk][q][S] that the update kernel adds up in a fixed order, so the fit is
// reproducible run to run (no floating-point atomics).
constexpr int kFitSub = 4;          // sub-chunks (waves) per workgroup
constexpr int kFitChunk = 1024;     // exons per workgroup
constexpr int kFitQ = 6;            // quantities per partial

// moments for the starting point: sum y, sum n, sum y^2/n, #cells with n > 0
__global__ void __launch_bounds__(kWave * kFitSub)
k_fit_moments(const int32_t* __restrict__ test, int64_t trs, int64_t tcs, const int32_t* __restrict__ ref, int64_t rrs,
              int64_t E, int64_t S, int stride, double* __restrict__ partial, int64_t tmod)
{
  const int64_t s = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const int64_t ts = (tmod ? s % tmod : s) * tcs;   // tmod > 0: column s takes test column s mod tmod (edrefcohort.inc)
  const int sub = threadIdx.y;
  const int64_t chunk = (int64_t)blockIdx.y * kFitSub + sub;
  if (s >= S) return;
  const int64_t e0 = (int64_t)blockIdx.y * kFitChunk + sub * (kFitChunk / kFitSub);
  const int64_t e1 = min(e0 + kFitChunk / kFitSub, E);
  double sy = 0, sn = 0, syy = 0, cnt = 0;
  for (int64_t e = e0; e < e1; e += stride) {
    const int y = test[e * trs + ts];   // (trs, tcs) = (S, 1): one column per sample; (1, 0): one shared column
    const int n = y + ref[e * rrs + s];
    if (n > 0) {
      sy += (double)y; sn += (double)n;
      syy += ((double)y * (double)y) / (double)n;
      cnt += 1.0;
    }
  }
  double* o = partial + (chunk * kFitQ) * S + s;
  o[0] = sy; o[S] = sn; o[2 * S] = syy; o[3 * S] = cnt; o[4 * S] = 0; o[5 * S] = 0;
}

// k_fit_moments for sample-major counts ([S][E], `pitch` ints between samples): one workgroup per sample reads every `stride`-th
// 64-exon piece of its row (coalesced) and leaves its sums as chunk 0 of the partials (k_fit_start is then given one chunk).
// Exon e of the fit is element e * by of the row (by > 1: subset.for.speed).
__global__ void __launch_bounds__(256)
k_fit_moments_sm(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, int64_t pitch, int64_t by, int64_t E, int64_t S,
                 int stride, double* __restrict__ partial, double* __restrict__ eta, double* __restrict__ lam, int* __restrict__ done,
                 int* __restrict__ depth_max, int cb)
{
  __shared__ double red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t s = blockIdx.x;
  const int64_t row0 = s * pitch;
  double sy = 0, sn = 0, syy = 0, cnt = 0;
  for (int64_t q = (int64_t)wave * stride; q * 64 < E; q += 4 * (int64_t)stride) {
    const int64_t e = q * 64 + lane;
    if (e < E) {
      const int y = ed_ldc(test, row0 + e * by, cb);
      const int n = y + ed_ldc(ref, row0 + e * by, cb);
      if (n > 0) {
        sy += (double)y; sn += (double)n;
        syy += ((double)y * (double)y) / (double)n;
        cnt += 1.0;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    sy += __shfl_xor(sy, d, 64); sn += __shfl_xor(sn, d, 64); syy += __shfl_xor(syy, d, 64); cnt += __shfl_xor(cnt, d, 64);
  }
  if (lane == 0) { red[wave][0] = sy; red[wave][1] = sn; red[wave][2] = syy; red[wave][3] = cnt; }
  __syncthreads();
  if (tid < 4) {
    const double v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    partial[(int64_t)tid * S + s] = v;
  }
  if (tid == 4 || tid == 5) partial[(int64_t)tid * S + s] = 0.0;
  if (tid == 0) {     // k_fit_start's method-of-moments start for this sample (same arithmetic), saving its launch
    const double tsy = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0], tsn = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
    const double tsyy = ((red[0][2] + red[1][2]) + red[2][2]) + red[3][2], tcnt = ((red[0][3] + red[1][3]) + red[2][3]) + red[3][3];
    double p = (tsn > 0) ? tsy / tsn : 0.5;
    p = fmin(fmax(p, 1e-6), 1.0 - 1e-6);
    const double q = 1.0 - p;
    const double pearson = (tsyy - 2.0 * p * tsy + p * p * tsn) / (p * q);
    double phi = (tsn - tcnt > 0) ? (pearson - tcnt) / (tsn - tcnt) : 0.01;
    phi = fmin(fmax(phi, 1e-4), 0.3);
    eta[s] = ed_plog(p / q);
    lam[s] = ed_plog((1.0 - phi) / phi);
    done[s] = (tcnt < 2.0) ? 1 : 0;
    if (tcnt > 0) atomicMax(depth_max, (int)fmin(tsn / tcnt, 2.0e9));
  }
}

// Sum the per-chunk partials of one quantity set for 64 samples with a 64 x kRedY thread block: thread
// (lane, y) adds chunks y, y + kRedY, ... in order, then the kRedY strands are added in a fixed tree in
// LDS.  The order never depends on timing, so the fit is reproducible run to run.
constexpr int kRedY = 16;
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partial, int64_t nchunk, int64_t S, int64_t s,
                                                bool live, double (&out)[kFitQ], double (*lds)[kRedY][kWave])
{
  const int y = threadIdx.y, lane = threadIdx.x;
  double acc[kFitQ];
#pragma unroll
  for (int q = 0; q < kFitQ; ++q) acc[q] = 0.0;
  if (live) {
    for (int64_t c = y; c < nchunk; c += kRedY) {
      const double* o = partial + (c * kFitQ) * S + s;
#pragma unroll
      for (int q = 0; q < kFitQ; ++q) acc[q] += o[q * S];
    }
  }
#pragma unroll
  for (int q = 0; q < kFitQ; ++q) lds[q][y][lane] = acc[q];
  __syncthreads();
  for (int h = kRedY / 2; h >= 1; h >>= 1) {
    if (y < h) {
#pragma unroll
      for (int q = 0; q < kFitQ; ++q) lds[q][y][lane] += lds[q][y + h][lane];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < kFitQ; ++q) out[q] = lds[q][0][lane];
}

// method-of-moments start: p0 = sum y / sum n; phi0 from the Pearson statistic
//   sum_e (y - n p)^2 / (n p q) ~ cnt + phi * sum_e (n - 1)
__global__ void __launch_bounds__(kWave * kRedY)
k_fit_start(const double* __restrict__ partial, int64_t nchunk, int64_t S, double* __restrict__ eta,
            double* __restrict__ lam, int* __restrict__ done, int* __restrict__ depth_max)
{
  __shared__ double lds[kFitQ][kRedY][kWave];
  const int64_t s = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const bool live = s < S;
  double tot[kFitQ];
  reduce_partials(partial, nchunk, S, s, live, tot, lds);
  if (!live || threadIdx.y != 0) return;
  const double sy = tot[0], sn = tot[1], syy = tot[2], cnt = tot[3];
  double p = (sn > 0) ? sy / sn : 0.5;
  p = fmin(fmax(p, 1e-6), 1.0 - 1e-6);
  const double q = 1.0 - p;
  const double pearson = (syy - 2.0 * p * sy + p * p * sn) / (p * q);
  double phi = (sn - cnt > 0) ? (pearson - cnt) / (sn - cnt) : 0.01;
  phi = fmin(fmax(phi, 1e-4), 0.3);
  eta[s] = ed_plog(p / q);
  lam[s] = ed_plog((1.0 - phi) / phi);
  done[s] = (cnt < 2.0) ? 1 : 0;   // nothing to fit
  // the batch's depth (largest per-sample mean total): picks the histogram geometry, see fit_hist_geometry
  if (cnt > 0) atomicMax(depth_max, (int)fmin(sn / cnt, 2.0e9));
}

// select.reference.set for a cohort (edrefcohort.inc): column i * T + j is test j against its i + 1 best candidates summed, and the
// reference's loop stops at the first i + 1 > 2 whose fitted proportion is below 0.05 (R/optimize_reference_set.R:130) -- the
// prefixes after that are never looked at.  The proportion falls as references are added, and its moment estimate (sum y / sum n,
// the start value in eta) is within a few per cent of the fitted one: the columns after the first prefix whose moment estimate is
// below `p_limit` (0.04: 20 % of margin) are marked done before the Newton passes.  skip_from[j] = the first prefix skipped (K if
// none): the caller falls back to the slow path should the fitted loop ever reach it.
__global__ void k_fit_skip_prefixes(const double* __restrict__ eta, int* __restrict__ done, int K, int64_t T, double p_limit,
                                    int* __restrict__ skip_from)
{
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= T) return;
  int from = K;
  for (int i = 0; i < K; ++i) {
    if (i >= from) { done[(int64_t)i * T + j] = 1; continue; }
    const double p = 1.0 / (1.0 + ed_pexp(-eta[(int64_t)i * T + j]));
    if (i + 1 > 2 && p < p_limit) from = i + 1;
  }
  skip_from[j] = from;
}

#ifndef ED_FIT_PACK_EARLY
#define ED_FIT_PACK_EARLY 1
#endif
#ifndef ED_FIT_PRE
#define ED_FIT_PRE 8      // cells of a column requested ahead of the one being consumed (round 5: 4 -> 8)
#endif
__global__ void __launch_bounds__(kWave * kFitSub)
k_fit_accum(const int32_t* __restrict__ test, int64_t trs, int64_t tcs, const int32_t* __restrict__ ref, int64_t rrs,
            int64_t E, int64_t S, int stride, const double* __restrict__ eta, const double* __restrict__ lam, const int* __restrict__ done,
            double* __restrict__ partial, int64_t tmod, const int32_t* __restrict__ colmap = nullptr, const int* __restrict__ n_map = nullptr)
{
  // colmap (late passes, k_fit_compact): slot j of the launch works on column colmap[j], j < *n_map -- the columns still iterating, packed.
  // A wave is as slow as its slowest lane: after the third full pass of the cohort reference sets' fit 10 % of the columns were still
  // iterating and 96 % of the waves with them; packed, the fourth pass costs a tenth.
  const int64_t j = (int64_t)blockIdx.x * kWave + threadIdx.x;
  if (colmap && j >= *n_map) return;
  const int64_t s = colmap ? (int64_t)colmap[j] : j;
  const int64_t ts = (tmod ? s % tmod : s) * tcs;
  const int sub = threadIdx.y;
  const int64_t chunk = (int64_t)blockIdx.y * kFitSub + sub;
  if (s >= S) return;
  if (done[s]) return;
  const double th = ed_pexp(lam[s]);
  const double p = 1.0 / (1.0 + ed_pexp(-eta[s]));
  const double a = th * p, b = th * (1.0 - p);
  const int64_t e0 = (int64_t)blockIdx.y * kFitChunk + sub * (kFitChunk / kFitSub);
  const int64_t e1 = min(e0 + kFitChunk / kFitSub, E);
  edfit::Acc acc = {0, 0, 0, 0, 0};
  double cnt = 0;
  // (trs, tcs) = (S, 1): one test column per sample; (1, 0): one shared test column.
  // The counts of the next kPre cells are requested before the current ones are consumed: ~150 VALU
  // instructions per cell do not cover an HBM round trip on their own.
  constexpr int kPre = ED_FIT_PRE;
  int yb[kPre], rb[kPre];
#pragma unroll
  for (int k = 0; k < kPre; ++k) {
    const int64_t e = e0 + (int64_t)k * stride;
    yb[k] = (e < e1) ? test[e * trs + ts] : 0;
    rb[k] = (e < e1) ? ref[e * rrs + s] : 0;
  }
  for (int64_t e = e0; e < e1; e += (int64_t)kPre * stride) {
    int yc[kPre], rc[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) { yc[k] = yb[k]; rc[k] = rb[k]; }
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      const int64_t en = e + (int64_t)(kPre + k) * stride;
      yb[k] = (en < e1) ? test[en * trs + ts] : 0;
      rb[k] = (en < e1) ? ref[en * rrs + s] : 0;
    }
    // one logarithm per run of kPre cells and gradient component (ed_fit_dev.hpp: accumulate_cell_run); the short digamma / trigamma series when
    // every lane's arguments of the run are large (the lanes of a wave are 64 columns at the SAME exons: deep or shallow together)
    double small = 1e300;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      if (e + (int64_t)k * stride < e1 && yc[k] + rc[k] > 0) small = fmin(small, fmin(a + (double)yc[k], b + (double)rc[k]));
    }
    const bool big = __all(small >= 32.0) != 0;           // (th + n >= a + y)
    double pa = 1.0, pb = 1.0;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
      if (e + (int64_t)k * stride < e1) {
        const int y = yc[k], n = yc[k] + rc[k];
        if (n > 0) cnt += 1.0;
        edfit::accumulate_cell_run(acc, pa, pb, a, b, th, y, n, big);
      }
    }
    acc.ga += edfit::flog(pa);
    acc.gb += edfit::flog(pb);
  }
  double* o = partial + (chunk * kFitQ) * S + j;       // (the partials of a packed pass sit at the slot, not at the column)
  o[0] = acc.ga; o[S] = acc.gb; o[2 * S] = acc.haa; o[3 * S] = acc.hab; o[4 * S] = acc.hbb; o[5 * S] = cnt;
}

// the columns still iterating, packed (order irrelevant): colmap[0 .. *n_map)
__global__ void k_fit_compact(const int* __restrict__ done, int64_t S, int32_t* __restrict__ colmap, int* __restrict__ n_map)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool act = s < S && !done[s];
  const unsigned long long m = __builtin_amdgcn_ballot_w64(act);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) base = atomicAdd(n_map, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  if (act) colmap[base + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)s;
}

// One Newton step on (eta, lambda) = (logit p, log(a+b)) from the summed gradient/Hessian of the log-likelihood
// with respect to (a, b) (tot = ga, gb, haa, hab, hbb, #cells, without the per-sample constant terms).  Steps are
// capped, and a non-concave local model falls back to a scaled gradient step.  `final_pass` marks passes over
// all exons: only those may declare convergence.
__device__ __forceinline__ void fit_newton_step(const double (&tot)[kFitQ], double& eta_v, double& lam_v, int& done_v, double tol,
                                                int final_pass)
{
  double ga = tot[0], gb = tot[1], haa = tot[2], hab = tot[3], hbb = tot[4], cnt = tot[5];
  const double th = ed_pexp(lam_v);
  const double p = 1.0 / (1.0 + ed_pexp(-eta_v));
  const double q = 1.0 - p;
  const double a = th * p, b = th * q;
  if (cnt != 0.0) {                                // (0: the caller's sums already hold the constant terms -- the histogram forms)
    double pa, qa, pb, qb, pt, qt;
    edfit::digamma_trigamma(a, pa, qa);
    edfit::digamma_trigamma(b, pb, qb);
    edfit::digamma_trigamma(th, pt, qt);
    ga -= cnt * (pa - pt);
    gb -= cnt * (pb - pt);
    haa -= cnt * (qa - qt);
    hab += cnt * qt;
    hbb -= cnt * (qb - qt);
  }
  const double ae = a * q, be = -b * p;          // d a / d eta, d b / d eta
  const double g_e = ga * ae + gb * be;
  const double g_l = ga * a + gb * b;
  const double h_ee = haa * ae * ae + 2.0 * hab * ae * be + hbb * be * be + (ga * a * q - gb * b * p) * (1.0 - 2.0 * p);
  const double h_el = haa * ae * a + hab * (ae * b + a * be) + hbb * be * b + ga * ae + gb * be;
  const double h_ll = haa * a * a + 2.0 * hab * a * b + hbb * b * b + ga * a + gb * b;
  // Dispersion coordinate of the Newton step.  lambda = log(a+b) is the better-conditioned coordinate for
  // ordinary dispersions (one full pass fewer than psi), but for nearly binomial data (phi -> 0) the
  // likelihood is exponentially flat in lambda and regular in psi = 1/(a+b) = phi/(1-phi), and a Newton
  // iteration in lambda walks off into the flat region: below psi = 5e-5 the step is taken in psi.
  // Chain rule:  d/dpsi = -th d/dlambda  =>  g_s = -th g_l,  h_es = -th h_el,  h_ss = th^2 (h_ll + g_l).
  const double psi = 1.0 / th;
  double de, ds;
  bool newton;   // a genuine Newton step (locally concave model): only such a step may declare convergence
  if (psi >= 5e-5) {
    const double det = h_ee * h_ll - h_el * h_el;
    double dl;
    newton = (h_ee < 0.0 && det > 0.0);
    if (newton) {
      de = -(h_ll * g_e - h_el * g_l) / det;
      dl = -(h_ee * g_l - h_el * g_e) / det;
    } else {
      // Not locally concave (typically psi far below its optimum, where the likelihood is convex and nearly flat
      // in lambda): move uphill in lambda by the Newton magnitude but at most 0.5 -- a factor 1.65 in psi -- per
      // iteration.  (A step scaled by the whole Hessian crawls: 1.4 % per iteration on the case kept in
      // tests/golden/fit_slow_start_case.npz.)
      de = g_e / (fabs(h_ee) + 1e-300);
      dl = (g_l > 0.0 ? 1.0 : -1.0) * fmin(0.5, fabs(g_l) / (fabs(h_ll) + 1e-300));
    }
    dl = fmin(fmax(dl, -1.0), 1.0);
    ds = psi * (ed_pexp(-dl) - 1.0);
  } else {
    const double g_s = -th * g_l;
    const double h_es = -th * h_el;
    const double h_ss = th * th * (h_ll + g_l);
    const double det = h_ee * h_ss - h_es * h_es;
    newton = (h_ee < 0.0 && det > 0.0);
    if (newton) {
      de = -(h_ss * g_e - h_es * g_s) / det;
      ds = -(h_ee * g_s - h_es * g_e) / det;
    } else {
      de = g_e / (fabs(h_ee) + 1e-300);
      ds = (g_s > 0.0 ? 1.0 : -1.0) * fmin(0.5 * psi, fabs(g_s) / (fabs(h_ss) + 1e-300));
    }
  }
  // Pinned at the lower bound of psi with the step pointing further down (nearly binomial data): psi stays, and the mean takes its
  // own one-dimensional Newton step -- it is what still has to converge (declared converged on reaching the bound, as rounds 1-3
  // did, a 36-row column kept an expected proportion 2 % off its maximum).
  const bool pinned = psi <= 1.0000001e-6 && ds <= 0.0;
  if (pinned) {
    newton = h_ee < 0.0;
    de = newton ? -g_e / h_ee : g_e / (fabs(h_ee) + 1e-300);
    ds = 0.0;
  }
  de = fmin(fmax(de, -1.0), 1.0);
  double npsi = psi + ds;
  npsi = fmin(fmax(npsi, 0.1 * psi), 10.0 * psi);       // multiplicative trust region
  // phi in [1e-6, 2/3].  Below phi ~ 1e-6 the model is numerically binomial: a + b > 1e6 and the digamma
  // differences that make up the gradient cancel to noise, so the dispersion is not estimable in binary64.
  npsi = fmin(fmax(npsi, 1e-6), 2.0);
  const double ne = fmin(fmax(eta_v + de, -20.0), 20.0);
  eta_v = ne;
  lam_v = -ed_plog(npsi);
  // Newton converges quadratically: once a step over ALL exons is below tol (kFitStepTol = 2e-5 in the per-cell passes, relative for psi), the
  // error left after applying it is of order C tol^2 = C 4e-10 against the 1e-8 the fit is held to -- no confirming pass is needed
  // (tests/test_gpu_fit.py::test_per_cell_fit_on_ill_conditioned_columns pins that margin on tiny-phi / few-exon columns).  (Pinned at the lower bound of psi: npsi = psi, and the test is the mean's.)
  if (final_pass && newton && fabs(de) < tol && fabs(npsi - psi) < tol * psi) done_v = 1;
}

__global__ void __launch_bounds__(kWave * kRedY)
k_fit_update(const double* __restrict__ partial, int64_t nchunk, int64_t S, double* __restrict__ eta,
             double* __restrict__ lam, int* __restrict__ done, double tol, int final_pass, const int32_t* __restrict__ colmap = nullptr,
             const int* __restrict__ n_map = nullptr)
{
  __shared__ double lds[kFitQ][kRedY][kWave];
  const int64_t j = (int64_t)blockIdx.x * kWave + threadIdx.x;
  const bool in_map = !colmap || j < *n_map;
  const int64_t s = (colmap && in_map) ? (int64_t)colmap[j] : j;
  const bool live = in_map && (s < S) && !done[s < S ? s : 0];
  if (!__syncthreads_or(live ? 1 : 0)) return;   // the whole tile has converged
  double tot[kFitQ];
  reduce_partials(partial, nchunk, S, j, live, tot, lds);
  if (!live || threadIdx.y != 0) return;
  double e = eta[s], l = lam[s];
  int d = 0;
  fit_newton_step(tot, e, l, d, tol, final_pass);
  eta[s] = e; lam[s] = l;
  if (d) done[s] = 1;
}

// ---- K5 through count histograms --------------------------------------------------------------------
// The gradient and Hessian of the log-likelihood are sums over cells of psi / psi' at a + y, b + (n - y) and
// a + b + n: functions of ONE count each.  So they are sums over the distinct values of y, n - y and n weighted by
// how often each value occurs in the sample -- three histograms per sample, built in ONE pass over the counts
// (k_fit_hist: LDS-privatised, 8 / 4 / 2 samples per workgroup).  Every Newton iteration then costs one reciprocal per
// bin instead of 3 x n_exons digamma evaluations, and all iterations run inside one launch (k_fit_hnewton).  Cells
// with a count beyond the histogram range go to per-sample overflow lists (second-level bins in LDS, then one by
// one); a sample whose list runs out is summed cell by cell inside the same launch.
// (Sums are grouped by value instead of by exon: the result differs from the per-cell path by rounding only.)
// Which geometry serves a batch: depth = the largest per-sample mean total count n (k_fit_start).  The unit bins plus
// the second level of k_fit_hnewton reach n = 9216 / 13312 / 21504; they should cover ~5 x the mean (the synthetic
// design's log-normal exon depths put 99.5 % of the cells below that).  Never a matter of correctness: cells beyond
// the bins are evaluated one by one.
__host__ __device__ __forceinline__ int fit_hist_geometry(int depth) { return depth <= 1843 ? 8 : (depth <= 2662 ? 4 : 2); }
// Which of the launched geometries runs: fit_columns launches the one the PREVIOUS fit's depth points to (all three
// the first time), `launched` says which (bit 0 / 1 / 2 = 8 / 4 / 2 samples per workgroup).  If the data's own choice
// is among them it runs; if not (the depth changed between calls) the launched geometry with the longest bins does --
// every geometry gives the same answer, only the time differs.
__host__ __device__ __forceinline__ int fit_hist_bit(int ks) { return ks == 8 ? 1 : (ks == 4 ? 2 : 4); }
__device__ __forceinline__ bool fit_hist_runs(int depth, int launched, int mine)
{
  const int sel = fit_hist_geometry(depth);
  if (launched & fit_hist_bit(sel)) return mine == sel;
  return mine == ((launched & 4) ? 2 : ((launched & 2) ? 4 : 8));
}

namespace hg8 {
#define ED_HG_KS 8
#ifdef ED_HG8_WG
#define ED_HG_WG ED_HG8_WG
#endif
#include "edfit_hist.inc"
#undef ED_HG_KS
#undef ED_HG_WG
}
namespace hg4 {
#define ED_HG_KS 4
#include "edfit_hist.inc"
#undef ED_HG_KS
}
namespace hg2 {
#define ED_HG_KS 2
#include "edfit_hist.inc"
#undef ED_HG_KS
}

__global__ void k_fit_finish(const double* __restrict__ eta, const double* __restrict__ lam, int64_t S,
                             double* __restrict__ phi, double* __restrict__ expected)
{
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const double th = ed_pexp(lam[s]);
  phi[s] = 1.0 / (th + 1.0);                        // phi = 1/(a + b + 1)
  expected[s] = 1.0 / (1.0 + ed_pexp(-eta[s]));     // fitted(mod) = plogis(eta)
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
    case 6: { double a, b; edfit::digamma_trigamma(x[i], a, b); r = a; } break;
    case 7: { double a, b; edfit::digamma_trigamma(x[i], a, b); r = b; } break;
    case 8: r = edsf::fdiv(x[i], y[i]); break;
    case 9: r = edsf::pexp_small(x[i]); break;
    case 10: r = edsf::plog_fast(x[i]); break;
    case 11: r = (double)edsf::lnbeta_sites(x[i], y[i]); break;
    case 12: { double sg; unsigned st; r = edsf::lngamma_sgn_any(x[i], &sg, &st); } break;
    case 13: { double sg; unsigned st; (void)edsf::lngamma_sgn_any(x[i], &sg, &st); r = sg + 8.0 * (double)st; } break;
    case 14: r = ed_psin_any(x[i]); break;
    case 15: { double a, b; edfit::digamma_trigamma_nolog_big(x[i], a, b); r = edfit::flog(x[i]) + a; } break;      // the short series of the fit's large arguments (x >= 32)
    case 16: { double a, b; edfit::digamma_trigamma_nolog_big(x[i], a, b); r = b; } break;
    case 17: {                                                                                                       // one cell through accumulate_cell_run, short series: d/da
      edfit::Acc c = {0, 0, 0, 0, 0}; double pa = 1.0, pb = 1.0;
      edfit::accumulate_cell_run(c, pa, pb, x[i], 4.0 * x[i], 5.0 * x[i], (int)y[i], 9 * (int)y[i], fmin(x[i] + y[i], 4.0 * x[i] + 8.0 * y[i]) >= 32.0);
      r = c.ga + edfit::flog(pa);
    } break;
    case 18: {                                                                                                       // ... the full series
      edfit::Acc c = {0, 0, 0, 0, 0}; double pa = 1.0, pb = 1.0;
      edfit::accumulate_cell_run(c, pa, pb, x[i], 4.0 * x[i], 5.0 * x[i], (int)y[i], 9 * (int)y[i], false);
      r = c.ga + edfit::flog(pa);
    } break;
    default: r = ed_pm_nan();
  }
  out[i] = r;
}

}  // namespace

#include "edfit_col.inc"

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
  double* d_lt3 = nullptr;       // [(E + C)][4][2]  per exon gap and quad lane: (lt[j][1], lt[j][2])
  int64_t* d_tile_off = nullptr;  // [C + 1] back-pointer word offsets (16 steps per word) per chromosome
  int64_t n_words = 0;
  int64_t max_words = 0;
  double c0 = 0, c1 = 0;         // log(1 - t), log(t / 2)
};

struct ed_batch {
  ed_plan* plan = nullptr;
  int64_t S = 0;
  double* d_loglik = nullptr;
  uint8_t* d_path = nullptr;
  uint32_t* d_bp = nullptr;      // [n_words][S][4] packed back-pointers (16 steps x 2 bits per quad lane)
  uint32_t* d_ppath = nullptr;   // [n_words][S] packed Viterbi states (16 exons x 2 bits)
  uint8_t* d_maps = nullptr;     // [n_words][S] trace-back maps of the 16-step words
  uint32_t* d_ent = nullptr;     // [n_words / 16 + C + 1][S] state of each word's last exon, 16 words per u32
  uint8_t* d_last = nullptr;     // [C][S] state of each chain's last exon (end of the forward pass)
  int32_t* d_job_off = nullptr;  // [n_jobs + 1]
  int32_t* d_job_chrom = nullptr;  // chromosomes in job order
  int64_t* d_seg = nullptr;        // emission segments in job order: (first workgroup, first exon, end exon) x n_jobs
  std::vector<int64_t> seg;        // host copy (+ one closing entry holding the total workgroup count)
  int32_t n_jobs = 0;
  std::vector<std::vector<int>> jobs;   // host copy: chromosomes of each Viterbi job
  std::vector<int32_t> group_off;       // job ranges of the overlap groups (the set in use)
  std::vector<int32_t> group_off_model; // ... as the cost model cut them (Viterbi of a group under the emissions of the next)
  bool overlap_groups = true;           // false: ONE group -- all emissions, then all chains (ed_batch_set_viterbi_overlap)
  std::vector<hipStream_t> sides;       // side streams (groups round-robin): Viterbi overlaps the emissions of later groups
  std::vector<hipEvent_t> job_ev;       // emissions of group g are complete
  std::vector<hipEvent_t> join_ev;      // Viterbi (+ trace-back) of group g is complete
  hipEvent_t zero_ev = nullptr;         // the call counts have been zeroed (first side stream)
  double* d_consts = nullptr;
  double2* d_tab_gl = nullptr;   // [3][kEmitTab][S] (Gamma*(a1 + obs), log(a1 + obs))   (k_emit_tables)
  double* d_tab_lg = nullptr;    // [3][kEmitTab][S] log Gamma(a1 + obs)
  int* d_cflags = nullptr;
  int32_t* d_counts = nullptr;
  int64_t* d_offsets = nullptr;
  int64_t* d_total = nullptr;
  unsigned long long* d_nerr = nullptr;
  ed_call* d_calls = nullptr;
  uint8_t* d_left_out = nullptr;     // one byte per workgroup of k_emit_bins_tab: a cell was left to the per-cell kernel
  double* d_ctab = nullptr;          // [3][kBinsRtab][S] lbeta(a1, a2) of the depth-binned model per reference count (edbins.inc)
  ed_call_info* d_info = nullptr;    // decoration of the call table (grown on demand: hipFree would synchronise the device
  int64_t info_cap = 0;              // and with it every other batch of a pipeline)
  struct FitWork* fitw = nullptr;    // workspace of ed_batch_fit (allocated on first use)
  void* binsw = nullptr;             // workspace of ed_batch_fit_bins' histogram form (BinsWork, edbins_hist.inc)
  int bins_form = 0;                 // form the last ed_batch_fit_bins took (ed_batch_fit_bins_form)
  int64_t bins_unconverged = 0;      // samples the last depth-binned fit left short of its tolerance (ed_batch_fit_bins_n_unconverged)
  int bins_pieces = 1;               // launches the depth-binned emission kernel is cut into (the cohort pipeline sets it)
  int64_t calls_cap = 0;
  hipStream_t stream = nullptr;         // where the results of the last run become available: the caller's stream, or `fin`
  hipStream_t fit_stream = nullptr;     // stream of the last ed_batch_fit (its own timing events only)
  // Asynchronous tail (ed_batch_set_async_tail): the caller's stream carries the emissions only; the Viterbi groups run on
  // the side streams as before, and what follows them (call counts -> scan -> call records) runs on `fin` instead of
  // being joined back into the caller's stream.  `done_ev` marks the end of the run; the next ed_batch_run on this batch
  // waits for it (the buffers are reused), the accessors synchronise on it.  With two batches used alternately on ONE
  // stream, batch N's Viterbi tail and call table then run underneath batch N+1's emissions.
  // Emission launch cut in two (single-group mode only): the first `split_frac` of the workgroups, an event, the rest.  The
  // cohort pipeline (edcohort.inc) makes the NEXT slab's dispersion fit wait for that event, so that the fit is issued --
  // in stream order, whatever the host's timing -- while this slab's emissions are under way.
  const double* src_phi = nullptr;      // cohort pipeline, given parameters: k_sample_consts of the next run / prepare reads (phi, expected) from
  const double* src_exp = nullptr;      // here and writes the copies the run is handed (the slot's arrays); cleared when consumed
  bool prepared_zeroed = false;         // ... batch_prepare also zeroed the run's counters (the cohort says when that is safe)
  bool prepared = false;                // the per-sample constants and tables of the NEXT run are already made (batch_prepare)
  const double* prepared_phi = nullptr; // ... from these parameters
  const double* prepared_exp = nullptr;
  double prepared_mix = 0.0;
  double split_frac = 0.0;
  hipEvent_t split_ev = nullptr;
  bool split_recorded = false;          // the last run recorded split_ev
  bool own_queues = false;              // streams of this batch are created with a (full) CU mask: a hardware queue each
  bool async_tail = false;
  bool last_run_async = false;          // the last run left its tail on `fin` (done_ev marks its end)
  hipStream_t fin = nullptr;
  hipEvent_t done_ev = nullptr, fork_ev = nullptr;
  int last_layout = 0;                  // counts_layout of the last run's inputs
  int counts_bits = 32;                 // 32: int32 counts; 16: uint16 (sample-major table mode only: ed_batch_set_counts_bits)
  int last_cb = 4;                      // bytes per count of the last run's inputs
  int cb() const { return counts_bits == 16 ? 2 : 4; }
  const int32_t* last_test = nullptr;   // inputs of the last ed_batch_run (for ed_batch_copy_call_info)
  const int32_t* last_ref = nullptr;
  const double* last_expected = nullptr;
  const double* last_cov_X = nullptr;   // covariate model of the last run (edcov.inc): expected is per exon
  const double* last_cov_beta = nullptr;
  int last_cov_K = -1;
  bool ran = false;
  bool fused = false;        // run emissions + Viterbi as ONE kernel (edfused.inc) instead of two overlapped ones
  bool keep_loglik = true;   // fused mode only: also write the [E][3][S] likelihood matrix (the S4 `likelihood` slot)
  int fit_hist = 1;          // ed_batch_fit: 1 = iterate on count histograms (one pass over the counts), geometry picked from
                             // the data; 8 / 4 / 2 = that geometry (samples per workgroup of k_fit_hist); 0 = per cell
  // table-driven emission mode (edtab.inc; ed_batch_set_emit_mode): buffers allocated on the first run in that mode
  int counts_layout = 0;         // 0: count matrices [E][S] (sample-minor); 1: [S][E] (sample-major, R's column-major E x S matrix) --
                                 // ed_batch_set_counts_layout; layout 1 serves the histogram fit and emit mode 2
  int emit_mode = 0;             // 0: strict (GSL's arithmetic operation for operation), 1: log-gamma difference tables
  int tab_tw = 16;               // samples per tile of k_emit_tab (16 / 32 / 64)
  int tab_capY = 4096, tab_capR = 32768;   // longest obs / ref table of a sample (entries); the tot table has their sum
  double tab_reach = 8.0;        // a table covers this multiple of the sample's mean count (+ 64)
  int tab_tails = 1;             // emit mode 2: samples whose counts outgrow the LDS windows are served by Stirling's series beyond them (ed_batch_set_emit_tails)
  int64_t tab_stride = 0;        // entries between the tables of consecutive samples = 2 (capY + capR)
  double* d_tabs = nullptr;      // [S][tab_stride][3]
  int4* d_tdims = nullptr;       // [S + 64] (Ly, Lr, Tm1, reason) per sample (edtab.inc: tab_dims_of)
  unsigned int* d_notab = nullptr;   // [2][1 + S] samples without tables: their number, then the samples; the same for the tail samples (k_tab_build)
  int4* d_twins = nullptr;           // [S] (n1, n2, n3, tail): the LDS windows of the sample-major form, and whether the sample is a tail sample (edtab.inc)
  double* d_tlg0 = nullptr;          // [S][3 tables][3 states][2] log Gamma of the tables' shape parameters (double-doubles): what a tail sample's series subtract
  unsigned long long* d_tacc = nullptr;   // [3][S] subsampled count sums (k_tab_stats)
  uint2* d_cold_list = nullptr;  // cells outside their sample's tables
  unsigned int* d_cold_n = nullptr;
  unsigned int cold_cap = 0;
  std::vector<int64_t> seg_t;    // emission segments for k_emit_tab's tile shape (as `seg`)
  int64_t* d_seg_t = nullptr;
  // emit mode 2 (sample-major form): the counts as [S][E], the likelihood matrix as [S][3][Epad]; the documented [E][3][S] form
  // (d_loglik) is made from it when an accessor asks (rows_valid)
  int32_t* d_test_sm = nullptr;
  int32_t* d_ref_sm = nullptr;
  double* d_loglik_sm = nullptr;
  int64_t Epad = 0;
  std::vector<int64_t> seg_sm;   // segments in blocks of 64 exons (first block, first exon, end exon), job order
  int2* d_blk_sm = nullptr;      // per block of that numbering: (its first exon, the end of its chromosome)
  unsigned int* d_vit_queue = nullptr;   // [8][2] work counters of k_viterbi_sm's persistent grid (one pair per launch group in flight; self-resetting)
  int vit_waves = 1024;          // waves of that grid: the SIMDs of the device (tab_setup_sm)
  int64_t nblk_sm = 0;
  bool rows_valid = true;
  int fit_mode = 0;          // ed_batch_fit: 0 = maximum likelihood (Newton); 1 = aod::betabin's procedure (Nelder-Mead from the
                             // glm start, optim()'s defaults) on the same histograms -- ed_batch_set_fit_mode
  bool timing = false;
  hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t epoch = nullptr;            // (cohort pipeline, option timing) a completed event of the owner: the emission launches' start / end times are kept relative to it
  std::vector<float>* emit_iv = nullptr; // ... appended here as (start_ms, end_ms) when a run's times are folded (ed_cohort_emission_intervals)
  bool have_run_times = false, have_fit_time = false;
  double stage_total[5] = {0, 0, 0, 0, 0};   // sums of the stage times of all timed runs / fits since timing was enabled
  float last_ms[5] = {0, 0, 0, 0, 0};        // stage times of the most recent folded run / fit (ed_batch_stage_ms)
  int64_t n_runs_timed = 0, n_fits_timed = 0;
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
try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
ED_CATCH("ed_device_count")

ED_EXPORT int ed_device_info(int device, char* name, size_t name_len, int* compute_units, size_t* total_mem)
try {
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
ED_CATCH("ed_device_info")

ED_EXPORT int ed_malloc(void** dptr, size_t bytes)
try {
  if (!dptr) return ed_fail(ED_ERR_INVALID, "ed_malloc: NULL output pointer");
  if (int rc = require_device()) return rc;
  HIP_TRY(hipMalloc(dptr, bytes ? bytes : 1));
  return ED_OK;
}
ED_CATCH("ed_malloc")
ED_EXPORT int ed_free(void* dptr)
try {
  if (dptr) HIP_TRY(hipFree(dptr));
  return ED_OK;
}
ED_CATCH("ed_free")
ED_EXPORT int ed_memcpy_h2d(void* dst, const void* src, size_t bytes)
try {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return ED_OK;
}
ED_CATCH("ed_memcpy_h2d")
ED_EXPORT int ed_memcpy_d2h(void* dst, const void* src, size_t bytes)
try {
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return ED_OK;
}
ED_CATCH("ed_memcpy_d2h")
ED_EXPORT int ed_synchronize(void* stream)
try {
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return ED_OK;
}
ED_CATCH("ed_synchronize")

// Device-to-host copy ordered on `st` and waited for there.  The accessors use this instead of hipMemcpy, which runs on the
// legacy null stream and would wait for every blocking stream of the process -- and the streams of the cohort pipeline
// (edcohort.inc) are blocking ones when they are created with a CU mask to get a hardware queue of their own.
static int ed_d2h(void* dst, const void* src, size_t bytes, hipStream_t st)
{
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
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
try {
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
ED_CATCH("ed_eval_sf")

#include "eddropin.inc"      // the two reference-shaped entries: ed_get_loglike_matrix, ed_hmm

// The text the reference prints while it computes the same matrix: one gsl_error() call = two Rprintf lines
// (src/error.c:45-48), in the order the reference makes them (rows in order; deletion, normal, duplication;
// within myprob the left operand of the subtraction first).  The sites are classified on the device
// (k_emit_rows_sites); only the formatting is host code.
ED_EXPORT int ed_get_loglike_matrix_messages(const double* phi, const double* expected, const int32_t* total,
                                             const int32_t* observed, int64_t n, double mixture, char* buf, size_t cap,
                                             size_t* needed)
try {
  if (n < 0 || !needed || (n > 0 && (!phi || !expected || !total || !observed)) || (cap > 0 && !buf))
    return ed_fail(ED_ERR_INVALID, "ed_get_loglike_matrix_messages: bad arguments");
  if (int rc = require_device()) return rc;
  *needed = 0;
  if (cap > 0) buf[0] = 0;
  if (n == 0) return ED_OK;
  DevBuf dphi, dexp, dtot, dobs, dcodes;
  HIP_TRY(dphi.alloc(n * 8)); HIP_TRY(dexp.alloc(n * 8)); HIP_TRY(dtot.alloc(n * 4)); HIP_TRY(dobs.alloc(n * 4));
  HIP_TRY(dcodes.alloc((size_t)n * 12));
  HIP_TRY(hipMemcpy(dphi.p, phi, n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dexp.p, expected, n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dtot.p, total, n * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(dobs.p, observed, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_emit_rows_sites, dim3((unsigned)((n + kEmitBlock - 1) / kEmitBlock)), dim3(kEmitBlock), 0, 0,
                     dphi.as<double>(), dexp.as<double>(), dtot.as<int32_t>(), dobs.as<int32_t>(), n, mixture,
                     dcodes.as<uint16_t>());
  HIP_TRY(hipGetLastError());
  std::vector<uint16_t> codes((size_t)n * 6);
  HIP_TRY(hipMemcpy(codes.data(), dcodes.p, (size_t)n * 12, hipMemcpyDeviceToHost));
  size_t len = 0;
  auto emit = [&](const char* file, int line, const char* reason) {
    char tmp[160];
    const int k = snprintf(tmp, sizeof tmp, "ERROR %s %i %s\nDefault GSL error handler invoked.\n", file, line, reason);
    if (k <= 0) return;
    for (int j = 0; j < k; ++j)
      if (len + (size_t)j + 1 < cap) buf[len + j] = tmp[j];
    len += (size_t)k;
  };
  static const int kGammaLine[6] = {0, 1283, 1239, 1253, 803, 1261};          // src/VP_gamma.c
  static const char* const kGammaWhat[6] = {"", "error", "domain error", "domain error", "error", "error"};
  for (size_t c = 0; c < codes.size(); ++c) {
    const unsigned code = codes[c];
    if (!code) continue;
    if (code & (1u << 9)) emit("beta.c", 56, "domain error");
    else if (code & (1u << 10)) emit("beta.c", 59, "domain error");
    else {
      for (int t = 0; t < 3; ++t) {
        const unsigned g = (code >> (3 * t)) & 7u;
        if (g >= 1 && g <= 5) emit("VP_gamma.c", kGammaLine[g], kGammaWhat[g]);
      }
      if (code & (1u << 11)) emit("beta.c", 44, "domain error");
    }
    emit("beta.c", 163, "gsl_sf_lnbeta_e(x, y, &result)");                     // src/eval.h:3-9 via src/beta.c:161-164
  }
  if (cap > 0) buf[std::min(len, cap - 1)] = 0;
  *needed = len;
  return ED_OK;
}
ED_CATCH("ed_get_loglike_matrix_messages")

// ---- plan ----------------------------------------------------------------------------------
ED_EXPORT int ed_plan_create(ed_plan** plan, int device, int64_t n_exons, int32_t n_chrom, const int32_t* chrom_off,
                             const int32_t* start, const int32_t* end, double transition_probability,
                             double expected_cnv_length)
try {
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
  struct Guard { ed_plan* p; ~Guard() { if (p) ed_plan_destroy(p); } } guard{p};   // released on success only
  p->device = device; p->E = n_exons; p->C = n_chrom; p->tprob = transition_probability; p->L = expected_cnv_length;
  p->chrom_off.assign(chrom_off, chrom_off + n_chrom + 1);
  // transitions <- matrix(c(1-t, t/2, t/2, .5,.5,0, .5,0,.5), byrow=TRUE)  (R/class_definition.R:343-347),
  // stored column-major as C_hmm reads it: T[j*3+k] = row k, column j
  const double t = transition_probability;
  const double rows[3][3] = {{1. - t, t / 2., t / 2.}, {0.5, 0.5, 0.}, {0.5, 0., 0.5}};
  double T[9];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[c * 3 + r] = rows[r][c];
  // per gap: the three distance-dependent entries A = lt[0][1] (= lt[0][2]), B = lt[1][1] (= lt[2][2]),
  // C = lt[1][2] (= lt[2][1]); the symmetric pairs are bitwise equal because rows 1 and 2 of T mirror
  // each other -- checked below.
  std::vector<double> lt3((size_t)(n_exons + n_chrom + 128) * 8, 0.0);   // +128 gaps: the kernel's prefetch may over-read
  std::vector<int64_t> tile_off((size_t)n_chrom + 1, 0);
  for (int c = 0; c < n_chrom; ++c) tile_off[c + 1] = tile_off[c] + ((int64_t)(chrom_off[c + 1] - chrom_off[c]) + 15) / 16;
  p->n_words = tile_off[n_chrom];
  for (int c = 0; c < n_chrom; ++c) p->max_words = std::max(p->max_words, tile_off[c + 1] - tile_off[c]);
  p->c0 = std::log(T[0]);   // log(1 - t): into normal from normal
  p->c1 = std::log(T[3]);   // log(t / 2): into a CNV state from normal (T[3] == T[6])
  bool symmetric = (T[3] == T[6]), out_of_range = false;
  {
    // one task per chromosome, spread over the host threads
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
    ed_thread_pool pool;
    std::vector<char> bad(nt, 0);
    auto work = [&](unsigned tid) {
      try {
      std::vector<int32_t> pos;
      std::vector<double> lt9;
      for (int c = tid; c < n_chrom; c += nt) {
        const int64_t lo = chrom_off[c], hi = chrom_off[c + 1], m = hi - lo;
        if (m <= 0) continue;
        pos.resize((size_t)m + 2);
        lt9.resize((size_t)(m + 1) * 9);
        // as.integer(c(positions[1] - 2*L, positions, end[last] + 2*L))  (R/class_definition.R:368)
        const double p_first = (double)start[lo] - 2 * expected_cnv_length, p_last = (double)end[hi - 1] + 2 * expected_cnv_length;
        if (!(p_first > -2147483649.0 && p_first < 2147483648.0 && p_last > -2147483649.0 && p_last < 2147483648.0)) {
          bad[tid] |= 2;      // (int32_t) of such a double is undefined behaviour; reported after the join
          continue;
        }
        pos[0] = (int32_t)p_first;
        for (int64_t i = 0; i < m; ++i) pos[1 + i] = start[lo + i];
        pos[m + 1] = (int32_t)p_last;
        fill_log_transitions(T, expected_cnv_length, pos.data(), m + 2, lt9.data());
        double* o = lt3.data() + (size_t)(lo + c) * 8;
        for (int64_t g = 0; g <= m; ++g) {
          const double* q = lt9.data() + g * 9;
          // quad lane 0 and 3: state 0 (A, A); lane 1: state 1 (B, C); lane 2: state 2 (C, B)
          o[g * 8 + 0] = q[1]; o[g * 8 + 1] = q[2]; o[g * 8 + 2] = q[4]; o[g * 8 + 3] = q[5];
          o[g * 8 + 4] = q[7]; o[g * 8 + 5] = q[8]; o[g * 8 + 6] = q[1]; o[g * 8 + 7] = q[2];
          if (std::memcmp(&q[1], &q[2], 8) || std::memcmp(&q[4], &q[8], 8) || std::memcmp(&q[5], &q[7], 8) ||
              std::memcmp(&q[0], &p->c0, 8) || std::memcmp(&q[3], &p->c1, 8) || std::memcmp(&q[6], &p->c1, 8))
            bad[tid] |= 1;
        }
      }
      } catch (...) { bad[tid] |= 4; }      // (host memory: nothing may leave a thread's function)
    };
    for (unsigned tid = 1; tid < nt; ++tid) pool.start(work, tid);
    work(0);
    pool.join();
    for (char b : bad) { if (b & 1) symmetric = false; if (b & 2) out_of_range = true; if (b & 4) return ed_fail(ED_ERR_NOMEM, "ed_plan_create: out of host memory"); }
  }
  if (!symmetric)
    return ed_fail(ED_ERR_STATE, "ed_plan_create: internal error, log-transition table is not symmetric");
  if (out_of_range)
    return ed_fail(ED_ERR_INVALID, "ed_plan_create: a padded position (first start - 2 L or last end + 2 L of a chromosome) does "
                                   "not fit a 32-bit integer (R's as.integer() would give NA, R/class_definition.R:368)");
  hipError_t e1 = hipMalloc((void**)&p->d_lt3, lt3.size() * 8);
  hipError_t e2 = hipMalloc((void**)&p->d_chrom_off, (size_t)(n_chrom + 1) * 4);
  hipError_t e3 = hipMalloc((void**)&p->d_tile_off, (size_t)(n_chrom + 1) * 8);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
    return ed_fail(ED_ERR_NOMEM, "ed_plan_create: device allocation failed");
  HIP_TRY(hipMemcpy(p->d_lt3, lt3.data(), lt3.size() * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_chrom_off, chrom_off, (size_t)(n_chrom + 1) * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_tile_off, tile_off.data(), (size_t)(n_chrom + 1) * 8, hipMemcpyHostToDevice));
  guard.p = nullptr;
  *plan = p;
  return ED_OK;
}
ED_CATCH("ed_plan_create")

ED_EXPORT void ed_plan_destroy(ed_plan* p)
{
  if (!p) return;
  if (p->d_lt3) (void)hipFree(p->d_lt3);
  if (p->d_tile_off) (void)hipFree(p->d_tile_off);
  if (p->d_chrom_off) (void)hipFree(p->d_chrom_off);
  delete p;
}

ED_EXPORT int64_t ed_plan_n_exons(const ed_plan* p) { return p ? p->E : 0; }

// A stream for the library's own use.  own_queue: created with a CU mask that names every CU -- the runtime gives such a
// stream a hardware queue of its own instead of multiplexing it with others onto the GPU_MAX_HW_QUEUES shared ones, so which
// of the pipeline's streams can run side by side no longer depends on how many streams the process created before (DESIGN.md
// 4.10).  Such a stream is a blocking stream (it synchronises with the legacy null stream); the library itself never uses the
// null stream on these paths.  Falls back to an ordinary non-blocking stream if the runtime refuses the mask.
static hipError_t ed_stream_create(hipStream_t* st, bool own_queue, int device)
{
  if (own_queue) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0xffffffffu);
      if (ncu % 32) mask.back() = (1u << (ncu % 32)) - 1u;
      if (hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()) == hipSuccess) return hipSuccess;
      (void)hipGetLastError();
    }
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

// ---- batch ---------------------------------------------------------------------------------
ED_EXPORT int ed_batch_create(ed_batch** batch, ed_plan* plan, int64_t n_samples)
try {
  if (!batch || !plan || n_samples <= 0) return ed_fail(ED_ERR_INVALID, "ed_batch_create: bad arguments");
  // k_emit_batch reads the per-sample tables through a buffer resource whose size field is 32 bits: 3 * kEmitTab * n_samples entries
  // of 16 bytes must stay below 2^31 (n_samples <= 43 690), or the entries beyond read as zero
  if (n_samples > 32768)
    return ed_fail(ED_ERR_INVALID, "ed_batch_create: %lld samples in one batch; at most 32768 (split the cohort into slabs: ed_cohort_*)",
                   (long long)n_samples);
  if (int rc = require_device()) return rc;
  HIP_TRY(hipSetDevice(plan->device));
  ed_batch* b = new (std::nothrow) ed_batch;
  if (!b) return ed_fail(ED_ERR_NOMEM, "out of host memory");
  struct Guard { ed_batch* b; ~Guard() { if (b) ed_batch_destroy(b); } } guard{b};   // released on success only
  b->plan = plan; b->S = n_samples;
  const int64_t E = plan->E, S = n_samples, C = plan->C;
  // capacity of the call table: generous for real data (a few hundred calls per sample); a run that needs more
  // gets a larger table from ed_batch_n_calls (the fill kernel is run again).
  b->calls_cap = std::min<int64_t>(std::max<int64_t>(1 << 20, 512 * S), std::max<int64_t>(E * S / 2, 1));
  bool ok = true;
  auto A = [&](void** p, size_t bytes) { if (ok && hipMalloc(p, bytes ? bytes : 1) != hipSuccess) ok = false; };
  A((void**)&b->d_path, (size_t)E * S);
  A((void**)&b->d_bp, (size_t)std::max<int64_t>(plan->n_words, 1) * S * 4 * 4);
  A((void**)&b->d_ppath, (size_t)std::max<int64_t>(plan->n_words, 1) * S * 4);
  A((void**)&b->d_maps, (size_t)std::max<int64_t>(plan->n_words, 1) * S);
  A((void**)&b->d_ent, (size_t)(plan->n_words / 16 + C + 1) * S * 4);
  A((void**)&b->d_last, (size_t)std::max<int64_t>(C, 1) * S);
  A((void**)&b->d_consts, (size_t)(9 * S + 64) * 8);    // + 64: k_emit_batch's row resources reach one sample block past the last row
  A((void**)&b->d_tab_gl, (size_t)3 * kEmitTab * S * 16);
  A((void**)&b->d_tab_lg, (size_t)3 * kEmitTab * S * 8);
  A((void**)&b->d_cflags, (size_t)(3 * S + 64) * 4);
  A((void**)&b->d_counts, (size_t)S * std::max<int64_t>(C, 1) * 4);
  A((void**)&b->d_offsets, (size_t)S * std::max<int64_t>(C, 1) * 8);
  A((void**)&b->d_total, 8);
  A((void**)&b->d_nerr, 64);   // [0..7] GSL error events, [8..11] "cold tasks were left out" flag of k_emit_batch, [16..39] table-mode counters (k_tab_cold)
  A((void**)&b->d_calls, (size_t)b->calls_cap * sizeof(ed_call));
  if (!ok)
    return ed_fail(ED_ERR_NOMEM, "ed_batch_create: device allocation failed (E=%lld S=%lld)", (long long)E, (long long)S);
  {
    // Viterbi jobs: one chromosome each, longest first, cut into a few GROUPS.  ed_batch_run issues the
    // emissions group by group; the Viterbi chains of a group start on a side stream as soon as that
    // group's emissions are done and run underneath the (VALU-bound) emissions of the following groups.
    // Groups shrink geometrically so that the last, exposed Viterbi launch only holds short chains.
    std::vector<int> order;
    int64_t total = 0;
    for (int c = 0; c < (int)C; ++c) {
      const int64_t mc = plan->chrom_off[c + 1] - plan->chrom_off[c];
      if (mc > 0) { order.push_back(c); total += mc; }
    }
    std::sort(order.begin(), order.end(), [&](int a, int bb) {
      const int64_t la = plan->chrom_off[a + 1] - plan->chrom_off[a], lb = plan->chrom_off[bb + 1] - plan->chrom_off[bb];
      return la != lb ? la > lb : a < bb; });
    std::vector<std::vector<int>> jobs;
    for (int c : order) jobs.push_back(std::vector<int>(1, c));
    // How to cut the jobs (longest chromosome first) into groups?  A group's Viterbi runs on a side stream as soon
    // as the group's emissions are done.  HIP multiplexes streams onto 4 hardware queues, and streams that share
    // a queue serialise (seen in the kernel trace: with 7 side streams the emissions themselves waited behind
    // Viterbi kernels), so there are kSideStreams = 3 side streams, used round-robin; a group's chains finish at
    // max(its emissions done, its stream free) + (its longest chain) -- slower when the resident Viterbi waves
    // outnumber the SIMDs.
    // Small batches want the long chromosomes in groups of their own, issued first, so that their chains start
    // early; large batches want few groups (every group boundary costs a short extra launch and the tail of
    // an emission launch).  A two-constant cost model picks among a handful of cut sets (measured on MI355X:
    // emissions 5.4e-11 s per cell; 6.5e-8 s per chain step, forward + trace-back).
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, plan->device));
    const double simds = 4.0 * prop.multiProcessorCount;
    const double c_emit = 5.4e-11, c_step = 6.5e-8, c_launch = 1.5e-5;
    const int32_t J = (int32_t)jobs.size();
    std::vector<std::vector<int32_t>> candidates;
    auto by_fraction = [&](const std::vector<double>& cuts) {
      std::vector<int32_t> goff(1, 0);
      int64_t run = 0;
      size_t gi = 0;
      for (int32_t k = 0; k < J; ++k) {
        run += plan->chrom_off[jobs[k][0] + 1] - plan->chrom_off[jobs[k][0]];
        while (gi + 1 < cuts.size() && (double)run >= cuts[gi] * (double)total && k + 1 < J) {
          if (k + 1 > goff.back()) goff.push_back(k + 1);
          ++gi;
        }
      }
      if (goff.back() != J) goff.push_back(J);
      return goff;
    };
    auto by_index = [&](std::vector<int32_t> idx) {
      std::vector<int32_t> goff(1, 0);
      for (int32_t v : idx) if (v > goff.back() && v < J) goff.push_back(v);
      if (goff.back() != J) goff.push_back(J);
      return goff;
    };
    candidates.push_back(by_fraction({1.0}));
    candidates.push_back(by_fraction({0.55, 1.0}));
    candidates.push_back(by_fraction({0.40, 0.72, 0.90, 1.0}));
    candidates.push_back(by_index({1}));
    candidates.push_back(by_index({1, 3}));
    candidates.push_back(by_index({1, 3, 8}));
    candidates.push_back(by_index({1, 2, 4, 10}));
    double best_cost = 1e300;
    for (const auto& goff : candidates) {
      double t_main = 0.0, finish = 0.0, waves = 0.0;
      double busy[kSideStreams] = {0.0, 0.0, 0.0};   // when each side stream becomes free
      for (size_t g = 0; g + 1 < goff.size(); ++g) {
        int64_t exons = 0, longest = 0;
        for (int k = goff[g]; k < goff[g + 1]; ++k) {
          const int64_t mc = plan->chrom_off[jobs[k][0] + 1] - plan->chrom_off[jobs[k][0]];
          exons += mc; longest = std::max(longest, mc);
        }
        t_main += c_emit * (double)exons * (double)S + (g > 0 ? c_launch : 0.0);
        waves += (double)(goff[g + 1] - goff[g]) * std::ceil((double)S / kVitChains);   // (earlier groups still running)
        const double vit = c_step * (double)longest * std::max(1.0, waves / (2.0 * simds));
        double& q = busy[g % kSideStreams];
        q = std::max(q, t_main) + vit;
        finish = std::max(finish, q);
      }
      const double cost = std::max(t_main, finish);
      if (cost < best_cost) { best_cost = cost; b->group_off = goff; }
    }
    b->group_off_model = b->group_off;
    std::vector<int32_t> joff(1, 0), jchr;
    for (auto& jb : jobs) { for (int c : jb) jchr.push_back(c); joff.push_back((int32_t)jchr.size()); }
    b->n_jobs = (int32_t)jobs.size();
    b->jobs = jobs;
    const size_t n_groups = std::max<size_t>(b->group_off.size() - 1, 1);
    b->sides.resize(std::min<size_t>(n_groups, kSideStreams));
    for (auto& sd : b->sides) HIP_TRY(ed_stream_create(&sd, false, plan->device));
    b->job_ev.resize(n_groups);
    for (auto& e : b->job_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    b->join_ev.resize(n_groups);
    for (auto& e : b->join_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&b->zero_ev, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void**)&b->d_job_off, joff.size() * 4));
    HIP_TRY(hipMalloc((void**)&b->d_job_chrom, std::max<size_t>(jchr.size(), 1) * 4));
    HIP_TRY(hipMemcpy(b->d_job_off, joff.data(), joff.size() * 4, hipMemcpyHostToDevice));
    if (!jchr.empty()) HIP_TRY(hipMemcpy(b->d_job_chrom, jchr.data(), jchr.size() * 4, hipMemcpyHostToDevice));
    // emission segments: one per job (= chromosome), in job order, so that a whole group is ONE launch
    int64_t blk = 0;
    for (auto& jb : jobs) {
      const int c = jb[0];
      const int64_t eb = plan->chrom_off[c], ee = plan->chrom_off[c + 1];
      b->seg.push_back(blk); b->seg.push_back(eb); b->seg.push_back(ee);
      {
        const int64_t neb = (ee - eb + kEmitRows - 1) / kEmitRows, nsb = (S + 63) / 64;
        blk += (nsb >= 8) ? ((neb + kEmitRun - 1) / kEmitRun) * (int64_t)kEmitRun * 8 * ((nsb + 7) / 8) : neb * nsb;
      }
    }
    b->seg.push_back(blk); b->seg.push_back(0); b->seg.push_back(0);
    HIP_TRY(hipMalloc((void**)&b->d_seg, b->seg.size() * 8));
    HIP_TRY(hipMemcpy(b->d_seg, b->seg.data(), b->seg.size() * 8, hipMemcpyHostToDevice));
  }
  for (auto& e : b->ev) HIP_TRY(hipEventCreate(&e));
  guard.b = nullptr;
  *batch = b;
  return ED_OK;
}
ED_CATCH("ed_batch_create")

static void fitwork_free(struct FitWork* w);
static void binswork_free(void* w);

ED_EXPORT void ed_batch_destroy(ed_batch* b)
{
  if (!b) return;
  fitwork_free(b->fitw);
  binswork_free(b->binsw);
  void* ptrs[] = {b->d_job_off, b->d_job_chrom, b->d_seg, b->d_maps, b->d_ent, b->d_last, b->d_ppath, b->d_loglik, b->d_path, b->d_bp, b->d_consts, b->d_tab_gl, b->d_tab_lg, b->d_cflags, b->d_counts, b->d_offsets, b->d_total, b->d_nerr, b->d_calls, b->d_info, b->d_ctab, b->d_left_out, b->d_tabs, b->d_tdims, b->d_twins, b->d_tlg0, b->d_notab, b->d_tacc, b->d_cold_list, b->d_cold_n, b->d_seg_t, b->d_test_sm, b->d_ref_sm, b->d_loglik_sm, b->d_blk_sm, b->d_vit_queue};
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : b->job_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : b->join_ev) if (e) (void)hipEventDestroy(e);
  if (b->zero_ev) (void)hipEventDestroy(b->zero_ev);
  for (auto& sd : b->sides) if (sd) (void)hipStreamDestroy(sd);
  if (b->fin) (void)hipStreamDestroy(b->fin);
  if (b->split_ev) (void)hipEventDestroy(b->split_ev);
  if (b->done_ev) (void)hipEventDestroy(b->done_ev);
  if (b->fork_ev) (void)hipEventDestroy(b->fork_ev);
  delete b;
}

// Add the stage times of the last timed RUN to the running totals (called before its events are recorded again, and by
// the readers).  The events belong to the previous run of this batch object; with batches used in rotation that work is a
// whole step old, so the synchronisation does not stall the pipeline.  The fit's pair of events is folded separately --
// before the NEXT fit records them again, or by a reader -- so that a run issued right after a fit on the same batch never
// waits for that fit on the host.  last_ms keeps the most recent values for ed_batch_stage_ms.
static int fold_run_times(ed_batch* b)
{
  if (b->have_run_times) {
    HIP_TRY(hipEventSynchronize(b->ev[4]));
    for (int i = 0; i < 4; ++i) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, b->ev[i], b->ev[i + 1]));
      b->stage_total[i] += ms;
      b->last_ms[i] = ms;
    }
    if (b->epoch && b->emit_iv) {
      float t0 = 0.f, t1 = 0.f;
      HIP_TRY(hipEventElapsedTime(&t0, b->epoch, b->ev[1]));
      HIP_TRY(hipEventElapsedTime(&t1, b->epoch, b->ev[2]));
      b->emit_iv->push_back(t0); b->emit_iv->push_back(t1);
    }
    ++b->n_runs_timed;
    b->have_run_times = false;
  }
  return ED_OK;
}
static int fold_fit_time(ed_batch* b)
{
  if (b->have_fit_time) {
    HIP_TRY(hipEventSynchronize(b->ev[6]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, b->ev[5], b->ev[6]));
    b->stage_total[4] += ms;
    b->last_ms[4] = ms;
    ++b->n_fits_timed;
    b->have_fit_time = false;
  }
  return ED_OK;
}

ED_EXPORT int ed_batch_enable_timing(ed_batch* b, int enable)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->timing = enable != 0;
  b->have_run_times = b->have_fit_time = false;
  for (double& t : b->stage_total) t = 0.0;
  for (float& t : b->last_ms) t = 0.f;
  b->n_runs_timed = b->n_fits_timed = 0;
  return ED_OK;
}
ED_CATCH("ed_batch_enable_timing")

ED_EXPORT int ed_batch_set_viterbi_overlap(ed_batch* b, int on)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->overlap_groups = on != 0;
  if (b->overlap_groups) b->group_off = b->group_off_model;
  else {
    b->group_off.assign(1, 0);
    if (b->n_jobs > 0) b->group_off.push_back(b->n_jobs);
  }
  return ED_OK;
}
ED_CATCH("ed_batch_set_viterbi_overlap")

ED_EXPORT int ed_batch_set_async_tail(ed_batch* b, int on)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  HIP_TRY(hipSetDevice(b->plan->device));
  if (on && !b->fin) {
    HIP_TRY(ed_stream_create(&b->fin, b->own_queues, b->plan->device));
    HIP_TRY(hipEventCreateWithFlags(&b->done_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&b->fork_ev, hipEventDisableTiming));
  }
  b->async_tail = on != 0;
  return ED_OK;
}
ED_CATCH("ed_batch_set_async_tail")

ED_EXPORT int ed_batch_wait(ed_batch* b, void* stream_)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (b->ran && b->last_run_async && b->done_ev) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream_, b->done_ev, 0));
  return ED_OK;
}
ED_CATCH("ed_batch_wait")

// How the emissions get their per-cell (phi, expected): the default model has one pair per sample; edbins.inc
// interpolates phi from the reference depth (phi.bins > 1); edcov.inc computes expected = plogis(X beta) per exon.
struct EmitModel {
  int bins = 0;                   // > 0: d_phi = phi.estimates [bins][S], edges = complete.bins [(bins + 1)][S]
  const double* edges = nullptr;
  const int* skip = nullptr;      // bins: a device word; non-zero = the parameters are not valid (the fit's histogram form declined, the cohort
                                  // pipeline will redo the slab): the emission kernels return at once
  bool cov = false;               // true: X [E][K] covariates, beta [K + 1][S] coefficients, d_phi [S]
  int K = 0;
  const double* X = nullptr;
  const double* beta = nullptr;
};

namespace {
// edbins.inc: emissions with per-cell shape parameters (six log-Betas per cell)
__global__ void k_emit_bins(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, int B, const double* __restrict__ edges,
                            const double* __restrict__ phib, const double* __restrict__ expected, const double* __restrict__ X, int K,
                            const double* __restrict__ beta, double mixture, int64_t E, int64_t S, double* __restrict__ loglik,
                            unsigned long long* __restrict__ nerr, const double* __restrict__ ctab, int rtab, const uint8_t* __restrict__ left_out,
                            const int* __restrict__ skip, int tiles_per_wg, int64_t n_blk);
__global__ void k_bins_ctab(int B, const double* __restrict__ edges, const double* __restrict__ phib, const double* __restrict__ expected,
                            double mixture, int64_t S, int rtab, double* __restrict__ ctab, const int* __restrict__ skip);
__global__ void k_emit_bins_tab(const int32_t* __restrict__ test, const int32_t* __restrict__ ref, int B, const double* __restrict__ edges,
                                const double* __restrict__ phib, const double* __restrict__ expected, double mixture, int64_t E, int64_t S,
                                const double* __restrict__ ctab, int rtab, double* __restrict__ loglik, uint8_t* __restrict__ left_out,
                                const int* __restrict__ skip, int64_t blk0, int64_t nblk);
constexpr int kBinsRtab = 8192;    // reference counts covered by the table of the depth-binned model's constants (edbins.inc)
}

// ---- table-driven emission mode (edtab.inc): buffers, segment table, the per-run table build ----
static int64_t tab_rows_per_wg(int tw) { return 4 * (64 / tw); }

static void tab_release(ed_batch* b)
{
  void** ptrs[] = {(void**)&b->d_tabs, (void**)&b->d_tdims, (void**)&b->d_notab, (void**)&b->d_tacc, (void**)&b->d_cold_list, (void**)&b->d_cold_n, (void**)&b->d_seg_t,
                   (void**)&b->d_twins, (void**)&b->d_tlg0};
  for (void** q : ptrs) { if (*q) (void)hipFree(*q); *q = nullptr; }
  (void)hipGetLastError();
}

// all of the mode's buffers or none (a half-made set must not pass for a made one on the next call: ADVICE r4)
static int tab_setup(ed_batch* b)
{
  if (b->d_tabs) return ED_OK;
  const ed_plan* p = b->plan;
  const int64_t S = b->S, E = p->E;
  b->tab_stride = 2 * ((int64_t)b->tab_capY + b->tab_capR);
  if ((int64_t)b->tab_tw * b->tab_stride * 24 >= ((int64_t)1 << 31))
    return ed_fail(ED_ERR_INVALID, "emit mode 1: %d samples x %lld table entries x 24 bytes per tile exceed 2^31 (smaller table caps or tile width)",
                   b->tab_tw, (long long)b->tab_stride);
  // (a sixteenth of the cells: the lists of round 5 -- a thirty-second -- ran out at 1 600 reads per exon, where 2 % of the cells lie beyond the conditioning limit,
  //  and the strict pass then walked all 2 x 10^8 cells again: 7.6 ms)
  b->cold_cap = (unsigned int)std::min<int64_t>(std::max<int64_t>(E * S / 16, 1 << 16), (int64_t)1 << 28) / kColdLists * kColdLists;
  // segments in job order, workgroups numbered for k_emit_tab's tile (rows x tab_tw samples), as ed_batch_create does for k_emit_batch
  const int64_t rows = tab_rows_per_wg(b->tab_tw), nsb = (S + b->tab_tw - 1) / b->tab_tw;
  int64_t blk = 0;
  b->seg_t.clear();
  for (auto& jb : b->jobs) {
    const int c = jb[0];
    const int64_t eb = p->chrom_off[c], ee = p->chrom_off[c + 1];
    b->seg_t.push_back(blk); b->seg_t.push_back(eb); b->seg_t.push_back(ee);
    const int64_t neb = (ee - eb + rows - 1) / rows;
    blk += (nsb >= 8) ? ((neb + kTabRun - 1) / kTabRun) * (int64_t)kTabRun * 8 * ((nsb + 7) / 8) : neb * nsb;
  }
  b->seg_t.push_back(blk); b->seg_t.push_back(0); b->seg_t.push_back(0);
  bool ok = true;
  auto A = [&](void** q, size_t bytes) { if (ok && hipMalloc(q, bytes ? bytes : 1) != hipSuccess) { ok = false; *q = nullptr; } };
  A((void**)&b->d_tabs, (size_t)S * b->tab_stride * 24);
  A((void**)&b->d_tdims, (size_t)(S + 64) * 16);
  A((void**)&b->d_notab, (size_t)2 * (S + 1) * 4);
  A((void**)&b->d_twins, (size_t)(S + 64) * 16);
  A((void**)&b->d_tlg0, (size_t)S * 18 * 8);
  A((void**)&b->d_tacc, (size_t)3 * S * 8);
  A((void**)&b->d_cold_list, (size_t)b->cold_cap * 8);
  A((void**)&b->d_cold_n, (size_t)(kColdLists + 1) * 4);
  A((void**)&b->d_seg_t, b->seg_t.size() * 8);
  if (ok && hipMemset(b->d_tdims, 0, (size_t)(S + 64) * 16) != hipSuccess) ok = false;
  if (ok && hipMemset(b->d_notab, 0, (size_t)2 * (S + 1) * 4) != hipSuccess) ok = false;
  if (ok && hipMemset(b->d_twins, 0, (size_t)(S + 64) * 16) != hipSuccess) ok = false;
  if (ok && hipMemcpy(b->d_seg_t, b->seg_t.data(), b->seg_t.size() * 8, hipMemcpyHostToDevice) != hipSuccess) ok = false;
  if (ok && ed_null_stream_fence() != hipSuccess) ok = false;
  if (!ok) {
    tab_release(b);
    return ed_fail(ED_ERR_NOMEM, "emit mode 1: cannot allocate the tables (%lld bytes for %lld samples)",
                   (long long)(S * b->tab_stride * 24), (long long)S);
  }
  return ED_OK;
}

static int tab_setup_sm(ed_batch* b)
{
  if (b->d_loglik_sm) return ED_OK;
  const ed_plan* p = b->plan;
  const int64_t S = b->S, E = p->E;
  b->Epad = ((E + 15) / 16) * 16 + 96;    // (k_viterbi_sm loads whole tiles up to four tiles past a chromosome's end)
  // Blocks of 64 exons on the ABSOLUTE exon grid, clipped to their chromosome: every block but the first and last of a chromosome
  // starts at a multiple of 64 exons, so that a wave's three 512-byte stores are whole aligned 128-byte lines of the [S][3][Epad]
  // matrix (chromosome-relative blocks made nearly every store begin and end with a partial line).
  int64_t blk = 0;
  b->seg_sm.clear();
  std::vector<int2> bm;
  for (auto& jb : b->jobs) {
    const int c = jb[0];
    const int64_t eb = p->chrom_off[c], ee = p->chrom_off[c + 1];
    b->seg_sm.push_back(blk); b->seg_sm.push_back(eb); b->seg_sm.push_back(ee);
    for (int64_t q = (eb / 64) * 64; q < ee; q += 64) {
      bm.push_back(make_int2((int)std::max(q, eb), (int)std::min(q + 64, ee)));
      ++blk;
    }
  }
  b->seg_sm.push_back(blk); b->seg_sm.push_back(0); b->seg_sm.push_back(0);
  if (bm.empty()) bm.push_back(make_int2(0, 0));
  b->nblk_sm = (int64_t)bm.size();
  bool ok = true;
  auto A = [&](void** q, size_t bytes) { if (ok && hipMalloc(q, bytes ? bytes : 1) != hipSuccess) { ok = false; *q = nullptr; } };
  A((void**)&b->d_test_sm, (size_t)std::max<int64_t>(E, 1) * S * 4);
  A((void**)&b->d_ref_sm, (size_t)std::max<int64_t>(E, 1) * S * 4);
  A((void**)&b->d_blk_sm, bm.size() * 8);
  A((void**)&b->d_vit_queue, 64);
  A((void**)&b->d_loglik_sm, ((size_t)S * 3 * b->Epad + 512) * 8);     // (last: it is the "already set up" sentinel)
  if (ok && hipMemset(b->d_vit_queue, 0, 64) != hipSuccess) ok = false;
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) b->vit_waves = prop.multiProcessorCount * 4;
    if (const char* e = ed_knob("ED_VIT_WAVES")) { if (atoi(e) > 0) b->vit_waves = atoi(e); }      // (experiments: profiles/r05_viterbi_waves.txt)
  }
  if (ok && hipMemset(b->d_loglik_sm, 0, ((size_t)S * 3 * b->Epad + 512) * 8) != hipSuccess) ok = false;
  if (ok && hipMemcpy(b->d_blk_sm, bm.data(), bm.size() * 8, hipMemcpyHostToDevice) != hipSuccess) ok = false;
  if (ok && ed_null_stream_fence() != hipSuccess) ok = false;
  if (!ok) {     // all of them or none
    void** ptrs[] = {(void**)&b->d_test_sm, (void**)&b->d_ref_sm, (void**)&b->d_blk_sm, (void**)&b->d_vit_queue, (void**)&b->d_loglik_sm};
    for (void** q : ptrs) { if (*q) (void)hipFree(*q); *q = nullptr; }
    (void)hipGetLastError();
    return ed_fail(ED_ERR_NOMEM, "emit mode 2: cannot allocate the sample-major matrices (E=%lld S=%lld)", (long long)E, (long long)S);
  }
  return ED_OK;
}

// emit mode 2: the [E][3][S] form of the likelihood matrix, made from the sample-major one when somebody asks for it
static int ensure_loglik_rows(ed_batch* b)
{
  if (b->rows_valid) return ED_OK;
  HIP_TRY(hipSetDevice(b->plan->device));      // (an accessor may be called from a thread whose current device is another one)
  const int64_t E = b->plan->E, S = b->S;
  if (!b->d_loglik) {
    if (hipMalloc((void**)&b->d_loglik, (size_t)std::max<int64_t>(E, 1) * 3 * S * 8) != hipSuccess)
      return ed_fail(ED_ERR_NOMEM, "cannot allocate the [E][3][S] form of the likelihood matrix");
  }
  if (E > 0)
    hipLaunchKernelGGL(k_ll_sm_to_rows, dim3((unsigned)((E + 63) / 64), (unsigned)((S + 63) / 64), 3), dim3(256), 0, b->stream, b->d_loglik_sm, E, b->Epad, S,
                       b->d_loglik);
  HIP_TRY(hipGetLastError());
  b->rows_valid = true;
  return ED_OK;
}

// after k_sample_consts: table lengths from a subsampled pass over the counts, then the entries
static int tab_build(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, hipStream_t st)
{
  if (int rc = tab_setup(b)) return rc;
  const int64_t E = b->plan->E, S = b->S;
  const int step = E >= 4096 ? 16 : 1;     // (d_tacc, d_notab[0], d_cold_n were zeroed by k_sample_consts, launched right before this on the same stream)
  const int tails = (b->emit_mode == 2 && b->tab_tails) ? 1 : 0;      // tail samples exist in the sample-major form only (edtab.inc: tab_windows)
  if (E > 0 && b->counts_layout == 1)
    hipLaunchKernelGGL(k_tab_stats_sm, dim3((unsigned)S, 4), dim3(64), 0, st, d_test, d_ref, E, E, S, step, b->d_tacc, b->cb());
  else if (E > 0)
    hipLaunchKernelGGL(k_tab_stats, dim3((unsigned)((S + 63) / 64), (unsigned)((E + 64 * step - 1) / (64 * step))), dim3(256), 0, st, d_test, d_ref, E, S,
                       step, b->d_tacc);
  if (S < 256 && ED_TAB_BUILD_THREADS == 64)
    hipLaunchKernelGGL(k_tab_build<256>, dim3((unsigned)S, 3), dim3(256), 0, st, b->d_consts, b->d_cflags, b->d_tacc, b->tab_reach, b->tab_capY, b->tab_capR,
                       b->d_tdims, S, b->d_tabs, b->tab_stride, b->d_notab, tails, b->d_twins, b->d_tlg0);
  else
    hipLaunchKernelGGL(k_tab_build<ED_TAB_BUILD_THREADS>, dim3((unsigned)S, 3), dim3(ED_TAB_BUILD_THREADS), 0, st, b->d_consts, b->d_cflags, b->d_tacc, b->tab_reach,
                       b->tab_capY, b->tab_capR, b->d_tdims, S, b->d_tabs, b->tab_stride, b->d_notab, tails, b->d_twins, b->d_tlg0);
  HIP_TRY(hipGetLastError());
  return ED_OK;
}

// The per-sample constants (k_sample_consts) and tables (k_emit_tables) of the default model, made AHEAD of ed_batch_run on
// another stream: they only need (phi, expected), and in the cohort pipeline those come from a fit that finishes in the middle
// of the previous slab's emission launch -- so the two latency-bound little kernels (0.3 ms) run there instead of standing
// between two emission launches.  The caller orders `stream_` after the batch's previous emission kernels (they read the
// tables) and the next ed_batch_run after this work (an event); ed_batch_run then skips the two kernels if it is handed the
// same parameters.
static int batch_prepare(ed_batch* b, const double* d_phi, const double* d_expected, double mixture, hipStream_t st,
                         const int32_t* d_test = nullptr, const int32_t* d_ref = nullptr, bool zero_counters = false)
{
  if (b->fused) return ED_OK;
  if (b->emit_mode >= 1 && (!d_test || !d_ref)) return ED_OK;   // the tables need the counts: made by the run itself
  const int64_t S = b->S;
  if (b->emit_mode >= 1) { if (int rc = tab_setup(b)) return rc; }
  // (the error / table counters of the batch's previous run may not have been read yet: they are zeroed by the run itself)
  const bool from_src = b->src_phi && b->src_exp;
  hipLaunchKernelGGL(k_sample_consts, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, from_src ? b->src_phi : d_phi, from_src ? b->src_exp : d_expected, mixture, S,
                     b->d_consts, b->d_cflags,
                     zero_counters ? b->d_nerr : (unsigned long long*)nullptr, b->emit_mode >= 1 ? b->d_tacc : (unsigned long long*)nullptr,
                     b->emit_mode >= 1 ? b->d_notab : (unsigned int*)nullptr, b->emit_mode >= 1 ? b->d_cold_n : (unsigned int*)nullptr, kColdLists + 1,
                     from_src ? const_cast<double*>(d_phi) : (double*)nullptr, from_src ? const_cast<double*>(d_expected) : (double*)nullptr);
  b->src_phi = b->src_exp = nullptr;
  b->prepared_zeroed = zero_counters;
  if (b->emit_mode >= 1) { if (int rc = tab_build(b, d_test, d_ref, st)) return rc; }
  else
  hipLaunchKernelGGL(k_emit_tables, dim3((unsigned)((S + 63) / 64), (unsigned)(kEmitTab / 4), 3), dim3(256), 0, st, b->d_consts, S, b->d_tab_gl,
                     b->d_tab_lg);
  HIP_TRY(hipGetLastError());
  b->prepared = true; b->prepared_phi = d_phi; b->prepared_exp = d_expected; b->prepared_mix = mixture;
  return ED_OK;
}

// bins > 0: depth-binned dispersion (d_phi = phi.estimates [bins][S], d_edges = complete.bins [(bins + 1)][S])
static int batch_run_impl(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                          const double* d_expected, double mixture, void* stream_, const EmitModel& em)
{
  const int bins = em.bins;
  const double* d_edges = em.edges;
  const bool plain = (em.bins == 0 && !em.cov);   // per-sample (phi, expected): k_emit_batch with hoisted constants
  if (!b || !d_test || !d_ref || !d_phi || !d_expected) return ed_fail(ED_ERR_INVALID, "ed_batch_run: NULL argument");
  if (!plain && b->fused) return ed_fail(ED_ERR_STATE, "ed_batch_run_bins / _cov: not available in fused mode");
  const bool tabm = plain && b->emit_mode >= 1 && !b->fused;   // emissions from log-gamma difference tables (edtab.inc)
  const bool tabsm = tabm && b->emit_mode == 2;                // ... sample-major form: tables in LDS, [S][3][Epad] likelihood matrix
  b->rows_valid = !tabsm;
  const bool cl1 = b->counts_layout == 1;                       // counts handed over sample-major [S][E]
  if (cl1 && !tabsm) return ed_fail(ED_ERR_STATE, "ed_batch_run: sample-major counts (ed_batch_set_counts_layout(batch, 1)) are served by emit mode 2 only");
  if (b->counts_bits == 16 && !(cl1 && tabsm)) return ed_fail(ED_ERR_STATE, "ed_batch_run: 16-bit counts (ed_batch_set_counts_bits(batch, 16)) are served by counts_layout 1 + emit mode 2 only");
  const int cb = b->cb();
  HIP_TRY(hipSetDevice(b->plan->device));   // the caller's thread may have another device current (one process, many GPUs)
  hipStream_t st = (hipStream_t)stream_;
  const ed_plan* p = b->plan;
  const int64_t E = p->E, S = b->S;
  const int32_t C = p->C;
  const bool async = b->async_tail && !b->fused;
  hipStream_t tail = async ? b->fin : st;   // where the call table is built and the results become available
  if (b->timing) { if (int rc = fold_run_times(b)) return rc; }
  // the previous run's asynchronous tail still reads the buffers -- whatever the mode of THIS run
  if (b->ran && b->last_run_async && b->done_ev) HIP_TRY(hipStreamWaitEvent(st, b->done_ev, 0));
  b->stream = tail;
  b->split_recorded = false;
  b->last_test = d_test; b->last_ref = d_ref; b->last_expected = d_expected; b->last_layout = b->counts_layout; b->last_cb = cb;
  struct SrcClear { ed_batch* b; ~SrcClear() { b->src_phi = b->src_exp = nullptr; } } src_clear{b};   // (whatever path the run takes)
  b->last_cov_X = em.cov ? em.X : nullptr; b->last_cov_K = em.cov ? em.K : -1; b->last_cov_beta = em.cov ? em.beta : nullptr;
  const bool ready = plain && b->prepared && b->prepared_phi == d_phi && b->prepared_exp == d_expected && b->prepared_mix == mixture && !b->fused;
  b->prepared = false;
  if (!plain || (ready && !b->prepared_zeroed)) HIP_TRY(hipMemsetAsync(b->d_nerr, 0, 64, st));     // (otherwise: k_sample_consts, below or in batch_prepare)
  b->prepared_zeroed = false;
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[0], st));
  if (plain && !ready) {
    if (tabm) { if (int rc = tab_setup(b)) return rc; }
    const bool from_src = b->src_phi && b->src_exp;
    hipLaunchKernelGGL(k_sample_consts, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, from_src ? b->src_phi : d_phi, from_src ? b->src_exp : d_expected, mixture, S,
                       b->d_consts, b->d_cflags, b->d_nerr, tabm ? b->d_tacc : (unsigned long long*)nullptr, tabm ? b->d_notab : (unsigned int*)nullptr,
                       tabm ? b->d_cold_n : (unsigned int*)nullptr, kColdLists + 1,
                       from_src ? const_cast<double*>(d_phi) : (double*)nullptr, from_src ? const_cast<double*>(d_expected) : (double*)nullptr);
    if (tabm) { if (int rc = tab_build(b, d_test, d_ref, st)) return rc; }
    else if (!b->fused)
      hipLaunchKernelGGL(k_emit_tables, dim3((unsigned)((S + 63) / 64), (unsigned)(kEmitTab / 4), 3), dim3(256), 0, st, b->d_consts, S,
                         b->d_tab_gl, b->d_tab_lg);
  }
  // one emission launch over workgroups [base, base + n) of the mode's numbering
  if (tabsm) { if (int rc = tab_setup_sm(b)) return rc; }
  if (tabsm && E > 0 && !cl1) {   // the counts sample-major (R's own layout; with ed_batch_set_counts_layout(batch, 1) they arrive that way)
    const dim3 tg((unsigned)((E + 63) / 64), (unsigned)((S + 63) / 64));
    hipLaunchKernelGGL(k_tab_rows_to_cols, tg, dim3(256), 0, st, d_test, E, S, b->d_test_sm);
    hipLaunchKernelGGL(k_tab_rows_to_cols, tg, dim3(256), 0, st, d_ref, E, S, b->d_ref_sm);
  }
  const std::vector<int64_t>& segv = tabsm ? b->seg_sm : (tabm ? b->seg_t : b->seg);
  int* const cold_flag = reinterpret_cast<int*>(b->d_nerr + 1);
  auto emit_launch = [&](int64_t n, int64_t base) {
    if (n <= 0) return;
    if (!tabm) {
      hipLaunchKernelGGL(k_emit_batch, dim3((unsigned)n), dim3(kEmitBlock), 0, st, d_test, d_ref, b->d_consts, b->d_cflags, b->d_seg, b->n_jobs, base,
                         S, (uint32_t)((S + 63) / 64), b->d_tab_gl, b->d_tab_lg, b->d_loglik, b->d_nerr, cold_flag);
      return;
    }
    if (tabsm) {    // n, base: blocks of 64 exons; every sample gets nsplit workgroups that share them
      int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(16, (512 + S - 1) / S));
      if (const char* e = ed_knob("ED_SM_NSPLIT")) { if (atoi(e) > 0) nsplit = atoi(e); }
      nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(nsplit, n / 32));
      const int64_t nwg = ((S + 7) / 8) * 8 * nsplit;
      if (cl1 && cb == 2)
        hipLaunchKernelGGL(k_emit_tab_sm<2>, dim3((unsigned)nwg), dim3(kSmBlock), 0, st, d_test, d_ref, b->d_tdims, b->d_tabs, b->tab_stride,
                           b->d_blk_sm, b->nblk_sm, base, n, nsplit, S, E, b->Epad, b->d_loglik_sm, b->d_cold_list, b->d_cold_n, b->cold_cap, b->d_twins, b->d_tlg0,
                           b->d_consts);
      else
        hipLaunchKernelGGL(k_emit_tab_sm<4>, dim3((unsigned)nwg), dim3(kSmBlock), 0, st, cl1 ? d_test : b->d_test_sm, cl1 ? d_ref : b->d_ref_sm, b->d_tdims, b->d_tabs, b->tab_stride,
                           b->d_blk_sm, b->nblk_sm, base, n, nsplit, S, E, b->Epad, b->d_loglik_sm, b->d_cold_list, b->d_cold_n, b->cold_cap, b->d_twins, b->d_tlg0,
                           b->d_consts);
      return;
    }
    const uint32_t nsb = (uint32_t)((S + b->tab_tw - 1) / b->tab_tw);
#define ED_TAB_LAUNCH(TW)                                                                                                                \
    hipLaunchKernelGGL(k_emit_tab<TW>, dim3((unsigned)n), dim3(kTabBlock), 0, st, d_test, d_ref, b->d_tdims, b->d_tabs, b->tab_stride, b->d_seg_t, \
                       b->n_jobs, base, S, nsb, b->d_loglik, b->d_cold_list, b->d_cold_n, b->cold_cap)
    if (b->tab_tw == 64) ED_TAB_LAUNCH(64); else if (b->tab_tw == 32) ED_TAB_LAUNCH(32); else if (b->tab_tw == 8) ED_TAB_LAUNCH(8); else if (b->tab_tw == 4) ED_TAB_LAUNCH(4); else ED_TAB_LAUNCH(16);
#undef ED_TAB_LAUNCH
  };
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[1], st));
  const int64_t cells = E * S;
  if (b->fused) {
    // One fused launch: every workgroup owns 16 samples x one chromosome (longest chromosomes first) and runs
    // emissions and Viterbi together (edfused.inc); the likelihood matrix is written only if it is kept.
    HIP_TRY(hipMemsetAsync(b->d_counts, 0, (size_t)S * std::max<int64_t>(C, 1) * 4, st));   // empty chromosomes: no calls
    if (b->keep_loglik && !b->d_loglik) {
      if (hipMalloc((void**)&b->d_loglik, (size_t)std::max<int64_t>(E, 1) * 3 * S * 8) != hipSuccess)
        return ed_fail(ED_ERR_NOMEM, "ed_batch_run: cannot allocate the likelihood matrix (%lld bytes)", (long long)(E * 3 * S * 8));
    }
    if (cells > 0 && b->n_jobs > 0)
      hipLaunchKernelGGL(k_emit_viterbi, dim3((unsigned)((S + kFuS - 1) / kFuS), (unsigned)b->n_jobs), dim3(kFuBlock), 0, st,
                         d_test, d_ref, b->d_consts, b->d_cflags, p->d_lt3, p->c0, p->c1, p->d_chrom_off, p->d_tile_off, S, C,
                         b->keep_loglik ? b->d_loglik : (double*)nullptr, b->d_bp, b->d_ppath, b->d_counts, b->d_job_off,
                         b->d_job_chrom, b->d_nerr);
    if (b->timing) HIP_TRY(hipEventRecord(b->ev[2], st));
  } else {
    if (!b->d_loglik && !tabsm) {
      if (hipMalloc((void**)&b->d_loglik, (size_t)std::max<int64_t>(E, 1) * 3 * S * 8) != hipSuccess)
        return ed_fail(ED_ERR_NOMEM, "ed_batch_run: cannot allocate the likelihood matrix");
    }
    // Emissions are issued group by group (groups of whole chromosomes, longest chromosomes first); when a
    // group's emissions are done its Viterbi chains start on the side stream and run underneath the
    // VALU-bound emissions of the following groups.  Only the last group's (short) chains are exposed.
    // the call counts (atomics of k_tb_paths; empty chromosomes: no calls) are zeroed on the first side stream, after the
    // first group's emissions: off the caller's stream, where it would stand between two emission launches
    bool counts_zeroed = false;
    if (!(b->group_off.size() > 1 && cells > 0)) {
      HIP_TRY(hipMemsetAsync(b->d_counts, 0, (size_t)S * std::max<int64_t>(C, 1) * 4, st));
      counts_zeroed = true;
    }
    for (size_t g = 0; g + 1 < b->group_off.size() && cells > 0; ++g) {
      const int j0 = b->group_off[g], j1 = b->group_off[g + 1];
      // One launch per group, except that a group following another starts with a short separate launch: the
      // previous group's Viterbi workgroups (side stream) are dispatched into the slots freed at that launch
      // boundary instead of queueing behind this group's thousands of pending workgroups.
      const int64_t blk0 = segv[3 * j0], nblk = plain ? segv[3 * j1] - blk0 : 0;
      if (tabm && g > 0) HIP_TRY(hipMemsetAsync(b->d_cold_n, 0, (size_t)(kColdLists + 1) * 4, st));   // (group 0: zeroed by k_sample_consts)
      const int64_t head = tabsm ? ((g > 0 && nblk > 512) ? 128 : 0) : ((g > 0 && nblk > 2 * kEmitHeadBlocks) ? kEmitHeadBlocks : 0);
      if (!plain && g == 0)   // one launch over every cell; the Viterbi groups follow it
      {
        const int64_t eblk = (E + kEmitBlock / 64 - 1) / (kEmitBlock / 64);   // 4 exons x 64 samples per workgroup
        const dim3 egrid((unsigned)((S + 63) / 64), (unsigned)std::min<int64_t>(eblk, 65535), (unsigned)((eblk + 65534) / 65535));
        const double* ctab = nullptr;
        if (bins > 0 && !em.cov) {
          // depth-binned dispersion: the constants lbeta(a1, a2) per (sample, state, reference count) first, then three tasks per cell
          if (!b->d_ctab || !b->d_left_out) {   // both or neither: a half-made pair must not pass for a made one on the next call
            if (b->d_ctab) { (void)hipFree(b->d_ctab); b->d_ctab = nullptr; }
            if (b->d_left_out) { (void)hipFree(b->d_left_out); b->d_left_out = nullptr; }
            if (hipMalloc((void**)&b->d_ctab, (size_t)3 * kBinsRtab * S * 8) != hipSuccess ||
                hipMalloc((void**)&b->d_left_out, (size_t)egrid.x * egrid.y * egrid.z) != hipSuccess) {
              if (b->d_ctab) { (void)hipFree(b->d_ctab); b->d_ctab = nullptr; }
              b->d_left_out = nullptr;
              (void)hipGetLastError();
              return ed_fail(ED_ERR_NOMEM, "ed_batch_run_bins: cannot allocate the table of constants");
            }
          }
          hipLaunchKernelGGL(k_bins_ctab, dim3((unsigned)((S + 63) / 64), (unsigned)(kBinsRtab / 4)), dim3(256), 0, st, bins, d_edges, d_phi, d_expected,
                             mixture, S, kBinsRtab, b->d_ctab, em.skip);
          // In the cohort pipeline the launch is cut into pieces: the NEXT slab's fit consists of several kernels whose workgroups need a
          // (nearly) whole CU each, and such a workgroup only gets one at a launch boundary, where the CUs drain -- under one uncut
          // launch the 3.5-ms fit took the launch's whole 13.8 ms and then stood between two emission launches.
          const int pieces = (int)std::max<int64_t>(1, std::min<int64_t>(b->bins_pieces, eblk));
          for (int pc = 0; pc < pieces; ++pc) {
            const int64_t pb0 = eblk * pc / pieces, pb1 = eblk * (pc + 1) / pieces, pn = pb1 - pb0;
            if (pn <= 0) continue;
            const dim3 pgrid((unsigned)((S + 63) / 64), (unsigned)std::min<int64_t>(pn, 65535), (unsigned)((pn + 65534) / 65535));
            hipLaunchKernelGGL(k_emit_bins_tab, pgrid, dim3(kEmitBlock), 0, st, d_test, d_ref, bins, d_edges, d_phi, d_expected, mixture, E, S,
                               b->d_ctab, kBinsRtab, b->d_loglik, b->d_left_out, em.skip, pb0, pn);
          }
          ctab = b->d_ctab;
        }
        {
          const int tpw = ctab ? 32 : 1;   // behind the tabulated kernel nearly every tile returns on one byte (1 / 8 / 32 on one box: 17.0 / 16.2 / 16.1 ms per slab)
          const int64_t nwg = (eblk + tpw - 1) / tpw;
          const dim3 fgrid((unsigned)((S + 63) / 64), (unsigned)std::min<int64_t>(nwg, 65535), (unsigned)((nwg + 65534) / 65535));
          hipLaunchKernelGGL(k_emit_bins, fgrid,
                             dim3(kEmitBlock), 0, st, d_test, d_ref, bins, d_edges, d_phi, em.cov ? (const double*)nullptr : d_expected, em.X,
                             em.cov ? em.K : -1, em.beta, mixture, E, S, b->d_loglik,
                             b->d_nerr, ctab, kBinsRtab, b->d_left_out, em.skip, tpw, eblk);
        }
      }
      emit_launch(head, blk0);
      // single-group mode with a split: [first part][split_ev][rest]; the cut is a multiple of 8 workgroups (XCD numbering)
      int64_t cut = 0;
      if (plain && b->group_off.size() == 2 && b->split_frac > 0.0 && b->split_frac < 1.0 && b->split_ev) {
        cut = ((int64_t)((double)nblk * b->split_frac) / 8) * 8;     // (mode 2: blocks of 64 exons -- any cut will do)
        if (cut <= 0 || cut >= nblk) cut = 0;
      }
      if (cut > 0) {
        emit_launch(cut, blk0);
        HIP_TRY(hipEventRecord(b->split_ev, st));
        b->split_recorded = true;
      }
      emit_launch(nblk - head - cut, blk0 + head + cut);
      if (tabm && nblk > 0) {  // the cells outside their sample's tables and the samples without tables (returns at once when there are none)
        const bool csm = cl1 || tabsm;       // counts walked sample-major ([S][E]: the caller's, or this mode's transposed copy)
        hipLaunchKernelGGL(k_tab_cold, dim3(1024), dim3(256), 0, st, csm ? (cl1 ? d_test : b->d_test_sm) : d_test, csm ? (cl1 ? d_ref : b->d_ref_sm) : d_ref,
                           b->d_consts, b->d_cflags, b->d_tdims, b->d_cold_list, b->d_cold_n,
                           b->cold_cap, b->d_seg_t, j0, j1, S, tabsm ? b->d_loglik_sm : b->d_loglik, b->d_nerr, tabsm ? (int64_t)1 : 3 * S,
                           tabsm ? b->Epad : S, tabsm ? 3 * b->Epad : (int64_t)1, csm ? (int64_t)1 : S, csm ? E : (int64_t)1, b->d_notab, b->d_nerr + 2, cl1 ? cb : 4);
      }
      else if (plain && nblk > 0)   // the out-of-domain tasks of this group, if k_emit_batch met any (returns at once otherwise)
        hipLaunchKernelGGL(k_emit_cold, dim3(512), dim3(256), 0, st, d_test, d_ref, b->d_consts, b->d_seg, j0, j1, S, b->d_loglik,
                           b->d_nerr, cold_flag);
      HIP_TRY(hipEventRecord(b->job_ev[g], st));
      hipStream_t side = b->sides[g % b->sides.size()];
      HIP_TRY(hipStreamWaitEvent(side, b->job_ev[g], 0));
      if (!counts_zeroed) {   // (g == 0; later groups' trace-backs are ordered after it through their own job events on st)
        HIP_TRY(hipMemsetAsync(b->d_counts, 0, (size_t)S * std::max<int64_t>(C, 1) * 4, side));
        HIP_TRY(hipEventRecord(b->zero_ev, side));
        counts_zeroed = true;
      } else if (b->sides.size() > 1) {
        HIP_TRY(hipStreamWaitEvent(side, b->zero_ev, 0));
      }
      const dim3 gw((unsigned)((S + 63) / 64), (unsigned)((p->max_words + 3) / 4), (unsigned)(j1 - j0));
      if (tabsm) {
        // a persistent grid, one wave per SIMD, pulling (chromosome, sample group) items longest chromosome first (edtab.inc)
        const unsigned n_groups = (unsigned)((S + kVitChains - 1) / kVitChains), n_items = n_groups * (unsigned)(j1 - j0);
        const unsigned n_waves = std::min<unsigned>(n_items, (unsigned)std::max(1, b->vit_waves));
        hipLaunchKernelGGL(k_viterbi_sm, dim3(n_waves), dim3(kWave), 0,
                           side, b->d_loglik_sm, b->Epad, p->d_lt3, p->c0, p->c1, p->d_chrom_off, p->d_tile_off, S, C, b->d_bp,
                           b->d_last, b->d_job_off, b->d_job_chrom, j0, b->d_vit_queue + 2 * (g % 8), n_groups, n_items);
      }
      else
      hipLaunchKernelGGL(k_viterbi, dim3((unsigned)((S + kVitChains - 1) / kVitChains), (unsigned)(j1 - j0)), dim3(kWave), 0,
                         side, b->d_loglik, p->d_lt3, p->c0, p->c1, p->d_chrom_off, p->d_tile_off, S, C, b->d_bp,
                         b->d_last, b->d_job_off, b->d_job_chrom, j0);
      hipLaunchKernelGGL(k_tb_maps, gw, dim3(256), 0, side, b->d_bp, p->d_chrom_off, p->d_tile_off, S, b->d_job_off,
                         b->d_job_chrom, j0, b->d_maps);
      hipLaunchKernelGGL(k_tb_chain, dim3((unsigned)((S + kWave - 1) / kWave), (unsigned)(j1 - j0)), dim3(kWave), 0, side,
                         p->d_chrom_off, p->d_tile_off, S, b->d_job_off, b->d_job_chrom, j0, b->d_last, b->d_maps, b->d_ent);
      hipLaunchKernelGGL(k_tb_paths, gw, dim3(256), 0, side, b->d_bp, p->d_chrom_off, p->d_tile_off, S, C, b->d_job_off,
                         b->d_job_chrom, j0, b->d_ent, b->d_ppath, b->d_path, b->d_counts);
      HIP_TRY(hipEventRecord(b->join_ev[g], side));
    }
    if (b->timing) HIP_TRY(hipEventRecord(b->ev[2], st));   // all emissions issued and done on the main stream
    if (async) {   // orders `tail` after everything this run put on the caller's stream, groups or not
      HIP_TRY(hipEventRecord(b->fork_ev, st));
      HIP_TRY(hipStreamWaitEvent(tail, b->fork_ev, 0));
    }
    for (size_t g = 0; g + 1 < b->group_off.size() && cells > 0; ++g) HIP_TRY(hipStreamWaitEvent(tail, b->join_ev[g], 0));
  }
  if (b->fused && C > 0 && cells > 0 && p->max_words > 0)   // (the two-kernel path writes the byte path in k_tb_paths)
    hipLaunchKernelGGL(k_path_expand, dim3((unsigned)((S + 63) / 64), (unsigned)((p->max_words + 3) / 4), (unsigned)C), dim3(256),
                       0, st, b->d_ppath, p->d_chrom_off, p->d_tile_off, S, b->d_path);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[3], tail));
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, tail, b->d_counts, (C > 0 && cells > 0) ? S * C : 0,
                     b->d_offsets, b->d_total);
  if (C > 0 && cells > 0)
    hipLaunchKernelGGL(k_calls_fill, dim3((unsigned)((S + kWave - 1) / kWave), (unsigned)C), dim3(kWave, kCallSeg), 0, tail,
                       b->d_ppath, p->d_chrom_off, p->d_tile_off, S, C, b->d_offsets, b->d_counts, b->d_calls, b->calls_cap);
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[4], tail));
  if (async) HIP_TRY(hipEventRecord(b->done_ev, tail));
  HIP_TRY(hipGetLastError());
  b->ran = true;
  b->last_run_async = async;
  b->have_run_times = b->timing;
  return ED_OK;
}

ED_EXPORT int ed_batch_run(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                           const double* d_expected, double mixture, void* stream_)
try {
  return batch_run_impl(b, d_test, d_ref, d_phi, d_expected, mixture, stream_, EmitModel());
}
ED_CATCH("ed_batch_run")

// workspace of the column-wise beta-binomial fit (shared by ed_batch_fit and ed_select_reference_set)
struct FitWork {
  double* partial = nullptr;   // [nchunk][kFitQ][S]
  double* eta = nullptr;
  double* lam = nullptr;
  int* done = nullptr;
  int32_t* colmap = nullptr;   // [S] the columns still iterating after the third full pass, packed (k_fit_compact)
  int* n_map = nullptr;        // ... their number
  int* fevals = nullptr;       // fit mode 1: objective evaluations of each sample's Nelder-Mead search
  uint32_t* hist = nullptr;    // count histograms (k_fit_hist; layout and size depend on the geometry, sized for the largest)
  int32_t* ov_y = nullptr;     // [lists][cap][S] cells beyond the histogram range, per row group of k_fit_hist
  int32_t* ov_r = nullptr;
  int32_t* ovn = nullptr;      // [lists][S] overflow cells of each (row group, sample)
  int* depth = nullptr;        // largest per-sample mean total of the columns being fitted (device; k_fit_start)
  int* h_depth = nullptr;      // the same, copied back after every fit (pinned host memory): next call's hint, -1 = none yet
  int64_t cap8 = 0, cap4 = 0, cap2 = 0;   // cells per list, by geometry
  int64_t nchunk = 0, S = 0, E_max = 0;
  int alloc_hist()
  {
    if (hist) return ED_OK;
    const int64_t E = E_max;   // sized for the largest fit this workspace serves
    // cells per (row group, sample) list: ~1/3 of the group's rows, and lists x cap < 65536 (the 16-bit second-level bins
    // of k_fit_hnewton count the cells of all lists of a sample)
    auto cap_of = [&](int64_t lists) { return std::min<int64_t>(65535 / lists, std::max<int64_t>(8, (E / lists) / 3 + 1)); };
    cap8 = cap_of(hg8::kHistGroups); cap4 = cap_of(hg4::kHistGroups); cap2 = cap_of(hg2::kHistGroups);
    const int64_t slots = std::max({cap8 * hg8::kHistGroups, cap4 * hg4::kHistGroups, cap2 * hg2::kHistGroups});
    // all four or none (`hist` is what says "made"): a failed allocation must not leave the sentinel set over missing lists
    const bool ok = hipMalloc((void**)&hist, (size_t)hg2::kHistHalves * hg2::kHistK * hg2::hist_padded(S) * 4) == hipSuccess &&
                    hipMalloc((void**)&ov_y, (size_t)slots * S * 4) == hipSuccess && hipMalloc((void**)&ov_r, (size_t)slots * S * 4) == hipSuccess &&
                    hipMalloc((void**)&ovn, (size_t)hg2::kHistGroups * S * 4) == hipSuccess;
    if (!ok) {
      (void)hipGetLastError();
      void** four[] = {(void**)&hist, (void**)&ov_y, (void**)&ov_r, (void**)&ovn};
      for (void** q : four) { if (*q) (void)hipFree(*q); *q = nullptr; }
      return ed_fail(ED_ERR_NOMEM, "beta-binomial fit: cannot allocate the count histograms");
    }
    return ED_OK;
  }
  int alloc(int64_t E, int64_t S_)
  {
    const int rc = alloc_impl(E, S_);
    if (rc != ED_OK) release();     // nothing half-allocated stays behind
    return rc;
  }
  int alloc_impl(int64_t E, int64_t S_)
  {
    release();
    S = S_;
    E_max = E;
    nchunk = ((E + kFitChunk - 1) / kFitChunk) * kFitSub;
    HIP_TRY(hipMalloc((void**)&partial, (size_t)std::max<int64_t>(nchunk, 1) * kFitQ * S * 8));
    // all-ones = NaN: a chunk read without having been written by the current fit shows up instead of passing as zero
    HIP_TRY(hipMemset(partial, 0xff, (size_t)std::max<int64_t>(nchunk, 1) * kFitQ * S * 8));
    HIP_TRY(hipMalloc((void**)&eta, (size_t)S * 8));
    HIP_TRY(hipMalloc((void**)&lam, (size_t)S * 8));
    HIP_TRY(hipMalloc((void**)&done, (size_t)S * 4));
    HIP_TRY(hipMalloc((void**)&colmap, (size_t)S * 4));
    HIP_TRY(hipMalloc((void**)&n_map, 4));
    HIP_TRY(hipMalloc((void**)&fevals, (size_t)S * 4));
    HIP_TRY(hipMemset(fevals, 0, (size_t)S * 4));
    HIP_TRY(ed_null_stream_fence());
    HIP_TRY(hipMalloc((void**)&depth, 4));
    HIP_TRY(hipHostMalloc((void**)&h_depth, 4, hipHostMallocDefault));
    *h_depth = -1;
    return ED_OK;
  }
  void release()
  {
    void* ptrs[] = {partial, eta, lam, done, colmap, n_map, fevals, hist, ov_y, ov_r, ovn, depth};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h_depth) (void)hipHostFree(h_depth);
    h_depth = nullptr;
    partial = eta = lam = nullptr; done = nullptr; colmap = nullptr; n_map = nullptr; fevals = nullptr; hist = nullptr; ov_y = ov_r = ovn = nullptr; depth = nullptr;
  }
};

static void fitwork_free(FitWork* w)
{
  if (w) { w->release(); delete w; }
}

// Fit S columns: column s has test counts test[e*trs + s*tcs] and reference counts ref[e*rrs + s], e < E.
// Step below which a full pass of the per-cell form declares a column converged.  Newton converges quadratically (fit_newton_step): a
// step s leaves an error ~C s^2, and the fit is held to 1e-8 -- 2e-5 leaves ~4e-10 C.  Rounds 1-4 asked for 1e-6, which the third full
// pass met only barely (its step IS ~1e-6 from the stride-16 start): a fourth pass over most of the cohort reference sets' 32 768 columns,
// 1.5 of the leg's 9 ms, confirmed what was already there.
constexpr double kFitStepTol = 2e-5;
#ifndef ED_FIT_COARSE
#define ED_FIT_COARSE 4
#endif
constexpr int kFitCoarsePasses = ED_FIT_COARSE;   // Newton steps on every 16th exon before the full passes (1 / 16 of a full pass each)
// use_hist: build count histograms once and iterate on them in one launch (needs one test column per sample laid
// out like the reference counts: tcs == 1, trs == rrs); otherwise per-cell passes, one launch pair per pass.
static int fit_columns(FitWork& w, const int32_t* d_test, int64_t trs, int64_t tcs, const int32_t* d_ref, int64_t rrs,
                       int64_t E, int64_t S, double* d_phi, double* d_expected, hipStream_t st, int use_hist = 0, int fit_mode = 0,
                       int64_t tmod = 0, int skip_K = 0, int* d_skip_from = nullptr, int64_t sm_pitch = 0, int cb = 4)
{
  // sm_pitch > 0: SAMPLE-MAJOR counts -- column s is the row test[s * sm_pitch + e * trs] (trs = the exon step, 1 or subset.for.speed's);
  // histogram form only
  const bool sm = sm_pitch > 0;
  if (cb != 4 && !sm) return ed_fail(ED_ERR_STATE, "fit: 16-bit counts (ed_batch_set_counts_bits(batch, 16)) are fitted from sample-major matrices only (counts_layout 1)");
  if (sm && (!use_hist || tmod)) return ed_fail(ED_ERR_STATE, "fit: sample-major counts are fitted on the count histograms only (ed_batch_set_fit_histograms(batch, 0) excludes them)");
  const int64_t nblk = (E + kFitChunk - 1) / kFitChunk;
  const int64_t nch = sm ? 1 : nblk * kFitSub;   // chunks THIS fit writes (the workspace may have been sized for more exons)
  const dim3 grid((unsigned)((S + kWave - 1) / kWave), (unsigned)nblk), block(kWave, kFitSub);
  const dim3 g1((unsigned)((S + 255) / 256)), b1(256);
  const dim3 gr((unsigned)((S + kWave - 1) / kWave)), br(kWave, kRedY);
  // every pass rewrites all partials, so chunks a strided pass barely touches cannot leave stale sums
  // (fit mode 1 starts from aod's glm-binomial intercept logit(sum y / sum n) over ALL exons; the Newton fit only needs a rough start)
  HIP_TRY(hipMemsetAsync(w.depth, 0, 4, st));
  if (sm) hipLaunchKernelGGL(k_fit_moments_sm, dim3((unsigned)S), dim3(256), 0, st, d_test, d_ref, sm_pitch, trs, E, S, fit_mode == 1 ? 1 : (E >= 65536 ? 16 : 4), w.partial,
                             w.eta, w.lam, w.done, w.depth, cb);
  else {
    hipLaunchKernelGGL(k_fit_moments, grid, block, 0, st, d_test, trs, tcs, d_ref, rrs, E, S, fit_mode == 1 ? 1 : (E >= 65536 ? 16 : 4), w.partial, tmod);
    hipLaunchKernelGGL(k_fit_start, gr, br, 0, st, w.partial, nch, S, w.eta, w.lam, w.done, w.depth);
  }
  if (fit_mode == 1 && !use_hist) return ed_fail(ED_ERR_STATE, "fit mode 1 (aod-nm) works on the count histograms: ed_batch_set_fit_histograms(batch, 0) excludes it");
  if (use_hist) {
    if (!sm && (tcs != 1 || trs != rrs)) return ed_fail(ED_ERR_INVALID, "fit_columns: histogram path needs per-sample test columns");
    const int64_t cell_rs = sm ? trs : rrs, cell_cs = sm ? sm_pitch : 1;   // element (e, s) of the counts at e * cell_rs + s * cell_cs
    if (int rc = w.alloc_hist()) return rc;
    if (const char* poison = getenv("ED_FIT_POISON")) {     // (diagnostic) the histogram workspace starts every fit from a byte pattern: a result that depends on what
      const int byte = atoi(poison) & 0xff;                // the allocation happened to hold differs with the pattern -- ED_FIT_POISON=165, =0, =255 ...
      const int64_t slots = std::max({w.cap8 * hg8::kHistGroups, w.cap4 * hg4::kHistGroups, w.cap2 * hg2::kHistGroups});
      HIP_TRY(hipMemsetAsync(w.hist, byte, (size_t)hg2::kHistHalves * hg2::kHistK * hg2::hist_padded(w.S) * 4, st));
      HIP_TRY(hipMemsetAsync(w.ov_y, byte, (size_t)slots * w.S * 4, st));
      HIP_TRY(hipMemsetAsync(w.ov_r, byte, (size_t)slots * w.S * 4, st));
      HIP_TRY(hipMemsetAsync(w.ovn, byte, (size_t)hg2::kHistGroups * w.S * 4, st));
    }
    // Which geometries to launch: the one asked for (tests), else the one the previous fit's depth points to, else
    // (first fit of this workspace) all three.  The kernels themselves decide which of the launched ones runs, from
    // THIS fit's depth (fit_hist_runs) -- no host round trip, and a stale hint costs time, never correctness.
    const int hint = *(volatile int*)w.h_depth;
    const int launched = use_hist > 1 ? fit_hist_bit(use_hist) : (hint < 0 ? 7 : fit_hist_bit(fit_hist_geometry(hint)));
    HIP_TRY(hipMemcpyAsync(w.h_depth, w.depth, 4, hipMemcpyDeviceToHost, st));
#define ED_FIT_HIST(NS, CAP)                                                                                                       \
    if ((launched & fit_hist_bit(NS::kHistId)) && sm)                                                                          \
      hipLaunchKernelGGL(NS::k_fit_hist_sm, dim3((unsigned)S), dim3(NS::kHsBlock), 0, st, d_test, d_ref, sm_pitch, trs, E, S, w.hist, w.ov_y, w.ov_r, \
                         w.ovn, CAP, w.depth, launched, cb);                                                                      \
    else if (launched & fit_hist_bit(NS::kHistId))                                                                                 \
      hipLaunchKernelGGL(NS::k_fit_hist, dim3((unsigned)((((S + NS::kHistSamples - 1) / NS::kHistSamples * NS::kHistHalves + 7) / 8) * 8)), \
                         dim3(NS::kHistBlock), 0, st, d_test, d_ref, rrs, E, S, w.hist, w.ov_y, w.ov_r, w.ovn, CAP, w.depth, launched);
    ED_FIT_HIST(hg8, w.cap8) ED_FIT_HIST(hg4, w.cap4) ED_FIT_HIST(hg2, w.cap2)
#undef ED_FIT_HIST
#define ED_FIT_NM(NS, CAP)                                                                                                         \
    if (launched & fit_hist_bit(NS::kHistId))                                                                                 \
      hipLaunchKernelGGL(NS::k_fit_hnm, dim3((unsigned)((S + NS::kHnS - 1) / NS::kHnS)), dim3(NS::kHnS, NS::kHnY), 0, st, w.hist, w.ov_y,     \
                         w.ov_r, w.ovn, CAP, S, w.eta, w.lam, w.done, 2000 /* optim(control = list(maxit = 2000)) */, d_test, d_ref, cell_rs,  \
                         E, w.depth, launched, w.fevals, cell_cs, d_phi, d_expected, sm ? 1 : 0, cb);
    if (fit_mode == 1) {
      ED_FIT_NM(hg8, w.cap8) ED_FIT_NM(hg4, w.cap4) ED_FIT_NM(hg2, w.cap2)
      HIP_TRY(hipGetLastError());
      return ED_OK;
    }
#undef ED_FIT_NM
#define ED_FIT_NEWTON(NS, CAP)                                                                                                     \
    if (launched & fit_hist_bit(NS::kHistId))                                                                                 \
      hipLaunchKernelGGL(NS::k_fit_hnewton, dim3((unsigned)((S + NS::kHnS - 1) / NS::kHnS)), dim3(NS::kHnS, NS::kHnY), 0, st, w.hist, w.ov_y, \
                         w.ov_r, w.ovn, CAP, S, w.eta, w.lam, w.done, 100, 1e-9 /* iterations are cheap here: converge tightly */, \
                         d_test, d_ref, cell_rs, E, w.depth, launched, cell_cs, d_phi, d_expected, sm ? 1 : 0, cb);
    ED_FIT_NEWTON(hg8, w.cap8) ED_FIT_NEWTON(hg4, w.cap4) ED_FIT_NEWTON(hg2, w.cap2)
#undef ED_FIT_NEWTON
    HIP_TRY(hipGetLastError());
    return ED_OK;
  }
  if (skip_K > 0 && tmod > 0 && d_skip_from)
    hipLaunchKernelGGL(k_fit_skip_prefixes, dim3((unsigned)((tmod + 255) / 256)), dim3(256), 0, st, w.eta, w.done, skip_K, tmod, 0.04, d_skip_from);
  // coarse Newton steps on every 16th exon, then full passes until the step is below tolerance
  const int coarse = (E >= 8192) ? kFitCoarsePasses : 0;   // a stride-16 subset below ~500 exons is too noisy to help
  const int cstride = 16;                   // (8 / 4 / 2 measured on the cohort reference sets' 10 000-row fit: 9.9 / 11.1 / 12.3 ms against 9.6)
  // Prefixes closed before the first pass (k_fit_skip_prefixes) are lanes that idle through every pass: the columns that iterate are packed from the
  // start then (round 6), and again before each of the first full passes as columns converge.  Which slot a column takes does not show in its sums.
  const bool can_pack = S >= 4096 && w.colmap;
  const bool pack_early = can_pack && skip_K > 0 && tmod > 0 && d_skip_from && ED_FIT_PACK_EARLY;
  auto compact = [&]() {
    (void)hipMemsetAsync(w.n_map, 0, 4, st);
    hipLaunchKernelGGL(k_fit_compact, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, st, w.done, S, w.colmap, w.n_map);
  };
  if (pack_early) compact();
  for (int it = 0; it < coarse; ++it) {
    hipLaunchKernelGGL(k_fit_accum, grid, block, 0, st, d_test, trs, tcs, d_ref, rrs, E, S, cstride, w.eta, w.lam, w.done, w.partial, tmod,
                       pack_early ? w.colmap : (const int32_t*)nullptr, pack_early ? w.n_map : (const int*)nullptr);
    hipLaunchKernelGGL(k_fit_update, gr, br, 0, st, w.partial, nch, S, w.eta, w.lam, w.done, 1e-6, 0,
                       pack_early ? w.colmap : (const int32_t*)nullptr, pack_early ? w.n_map : (const int*)nullptr);
  }
  // (round 5: two passes on every 4th exon between the coarse and the full ones were tried -- 9.2 -> 10.0 ms for the cohort reference sets'
  //  32 768 columns: the full passes needed are the same three, the medium ones came on top)
  // The looser stopping rule is the reference-set searches' (one test column shared by the columns: tcs == 0, or taken modulo tmod), whose fits feed a
  // choice among candidates; a fit handed back as the sample's (phi, expected) keeps 1e-6 -- on a column with phi ~ 8e-6 the constant C of the header above is
  // ~500 and 2e-5 left 1.9e-7 (tests/test_gpu_fit.py::test_per_cell_fit_on_ill_conditioned_columns).
  const double step_tol = (tcs == 0 || tmod > 0) ? kFitStepTol : 1e-6;
  for (int it = 0; it < 10; ++it) {   // converged columns skip their work; typically 3 passes do something
    // from the fourth pass on: the columns still iterating packed into the first waves (k_fit_compact), many columns only
    const bool packed = can_pack && (it >= 3 || pack_early);
    if (packed && (it == 3 || (pack_early && it >= 1 && it < 3))) compact();
    hipLaunchKernelGGL(k_fit_accum, grid, block, 0, st, d_test, trs, tcs, d_ref, rrs, E, S, 1, w.eta, w.lam, w.done, w.partial, tmod,
                       packed ? w.colmap : (const int32_t*)nullptr, packed ? w.n_map : (const int*)nullptr);
    hipLaunchKernelGGL(k_fit_update, gr, br, 0, st, w.partial, nch, S, w.eta, w.lam, w.done, step_tol, 1,
                       packed ? w.colmap : (const int32_t*)nullptr, packed ? w.n_map : (const int*)nullptr);
  }
  hipLaunchKernelGGL(k_fit_finish, g1, b1, 0, st, w.eta, w.lam, S, d_phi, d_expected);
  HIP_TRY(hipGetLastError());
  return ED_OK;
}

// R/class_definition.R:107-113: a scalar subset.for.speed = n fits on rows seq(1, nrow, by = floor(nrow / n)),
// i.e. 0-based exons 0, by, 2*by, ... -- a strided view of the count matrices.
ED_EXPORT int ed_batch_fit_subset(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, int64_t by, double* d_phi,
                                  double* d_expected, void* stream_)
try {
  if (!b || !d_test || !d_ref || !d_phi || !d_expected) return ed_fail(ED_ERR_INVALID, "ed_batch_fit: NULL argument");
  hipStream_t st = (hipStream_t)stream_;
  const int64_t E = b->plan->E, S = b->S;
  if (E <= 0) return ed_fail(ED_ERR_INVALID, "ed_batch_fit: no exons");
  if (by < 1) return ed_fail(ED_ERR_INVALID, "ed_batch_fit_subset: row step %lld < 1 (subset.for.speed larger than the number of exons?)", (long long)by);
  HIP_TRY(hipSetDevice(b->plan->device));
  if (!b->fitw) {
    b->fitw = new (std::nothrow) FitWork;
    if (!b->fitw) return ed_fail(ED_ERR_NOMEM, "out of host memory");
    if (int rc = b->fitw->alloc(E, S)) return rc;
  }
  b->fit_stream = st;
  if (b->timing) { if (int rc = fold_fit_time(b)) return rc; }
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[5], st));
  const int64_t rows = (E - 1) / by + 1;
  if (b->counts_layout == 1) {
    if (int rc = fit_columns(*b->fitw, d_test, by, 1, d_ref, by, rows, S, d_phi, d_expected, st, b->fit_hist, b->fit_mode, 0, 0, nullptr, E, b->cb())) return rc;
  } else {
    if (b->counts_bits == 16) return ed_fail(ED_ERR_STATE, "ed_batch_fit: 16-bit counts (ed_batch_set_counts_bits(batch, 16)) come sample-major (ed_batch_set_counts_layout(batch, 1))");
    if (int rc = fit_columns(*b->fitw, d_test, S * by, 1, d_ref, S * by, rows, S, d_phi, d_expected, st, b->fit_hist, b->fit_mode)) return rc;
  }
  if (b->timing) HIP_TRY(hipEventRecord(b->ev[6], st));
  b->have_fit_time = b->timing;
  return ED_OK;
}
ED_CATCH("ed_batch_fit_subset")

ED_EXPORT int ed_batch_fit(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, double* d_phi, double* d_expected,
                           void* stream_)
try {
  return ed_batch_fit_subset(b, d_test, d_ref, 1, d_phi, d_expected, stream_);
}
ED_CATCH("ed_batch_fit")

// How the last dispersion fit of the batch ended: a sample whose Newton iteration used up its iteration budget without a
// step below tolerance is reported, not silently returned (aod::betabin exposes optim()'s convergence code likewise).
ED_EXPORT int ed_batch_fit_n_unconverged(ed_batch* b, int64_t* n_unconverged, int32_t* first_sample)
try {
  if (!b || !n_unconverged) return ed_fail(ED_ERR_INVALID, "NULL argument");
  if (!b->fitw || !b->fitw->done) return ed_fail(ED_ERR_STATE, "no ed_batch_fit has been issued on this batch");
  HIP_TRY(hipSetDevice(b->plan->device));
  HIP_TRY(hipStreamSynchronize(b->fit_stream));
  std::vector<int> done((size_t)b->S);
  if (int rc = ed_d2h(done.data(), b->fitw->done, (size_t)b->S * 4, b->fit_stream)) return rc;
  int64_t n = 0;
  int32_t first = -1;
  for (int64_t s = 0; s < b->S; ++s)
    if (!done[s]) { if (first < 0) first = (int32_t)s; ++n; }
  *n_unconverged = n;
  if (first_sample) *first_sample = first;
  return ED_OK;
}
ED_CATCH("ed_batch_fit_n_unconverged")

ED_EXPORT int64_t ed_batch_n_samples(const ed_batch* b) { return b ? b->S : 0; }
ED_EXPORT const double* ed_batch_loglik(const ed_batch* b)
{
  if (!b || !(b->keep_loglik || !b->fused)) return nullptr;
  if (!b->rows_valid && ensure_loglik_rows(const_cast<ed_batch*>(b)) != ED_OK) return nullptr;   // (emit mode 2: made on the run's stream)
  return b->d_loglik;
}

ED_EXPORT int ed_batch_set_fused(ed_batch* b, int fused)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->fused = fused != 0;
  return ED_OK;
}
ED_CATCH("ed_batch_set_fused")

ED_EXPORT int ed_batch_set_fit_histograms(ed_batch* b, int on)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (on != 0 && on != 1 && on != 2 && on != 4 && on != 8) return ed_fail(ED_ERR_INVALID, "ed_batch_set_fit_histograms: 0, 1 (automatic), or a geometry 8 / 4 / 2");
  b->fit_hist = on;
  return ED_OK;
}
ED_CATCH("ed_batch_set_fit_histograms")

ED_EXPORT int ed_batch_set_fit_mode(ed_batch* b, int mode)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (mode != 0 && mode != 1) return ed_fail(ED_ERR_INVALID, "ed_batch_set_fit_mode: 0 (maximum likelihood) or 1 (aod-nm)");
  b->fit_mode = mode;
  return ED_OK;
}
ED_CATCH("ed_batch_set_fit_mode")

ED_EXPORT int ed_batch_set_emit_mode(ed_batch* b, int mode)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (mode != 0 && mode != 1 && mode != 2) return ed_fail(ED_ERR_INVALID, "ed_batch_set_emit_mode: 0 (strict), 1 (tables, exon-major tiles) or 2 (tables, sample-major)");
  if (mode >= 1) {
    HIP_TRY(hipSetDevice(b->plan->device));
    if (const char* e = ed_knob("ED_TAB_REACH")) { if (!b->d_tabs && atof(e) >= 1.0) b->tab_reach = atof(e); }   // (experiments)
    if (const char* e = ed_knob("ED_TAB_TW")) { const int tw = atoi(e); if (!b->d_tabs && (tw == 4 || tw == 8 || tw == 16 || tw == 32 || tw == 64)) b->tab_tw = tw; }
    if (int rc = tab_setup(b)) return rc;
    if (mode == 2) { if (int rc = tab_setup_sm(b)) return rc; }
  }
  b->emit_mode = mode;
  b->prepared = false;
  return ED_OK;
}
ED_CATCH("ed_batch_set_emit_mode")

ED_EXPORT int ed_batch_set_counts_layout(ed_batch* b, int layout)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (layout != 0 && layout != 1) return ed_fail(ED_ERR_INVALID, "ed_batch_set_counts_layout: 0 ([n_exons][n_samples]) or 1 ([n_samples][n_exons])");
  b->counts_layout = layout;
  b->prepared = false;
  return ED_OK;
}
ED_CATCH("ed_batch_set_counts_layout")

// 16: the count matrices handed to ed_batch_fit / ed_batch_run are uint16 [n_samples][n_exons] (counts below 65 536) -- half the bytes of every pass over them.
// Served by the sample-major table mode only (counts_layout 1, emit mode 2, per-sample dispersion); anything else says so when it is run.
ED_EXPORT int ed_batch_set_counts_bits(ed_batch* b, int bits)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (bits != 32 && bits != 16) return ed_fail(ED_ERR_INVALID, "ed_batch_set_counts_bits: 32 (int32) or 16 (uint16)");
  b->counts_bits = bits;
  b->prepared = false;
  return ED_OK;
}
ED_CATCH("ed_batch_set_counts_bits")

ED_EXPORT int ed_batch_set_emit_tables(ed_batch* b, int32_t cap_obs, int32_t cap_ref, double reach)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (b->d_tabs) return ed_fail(ED_ERR_STATE, "ed_batch_set_emit_tables: the tables are already allocated (call before ed_batch_set_emit_mode(batch, 1))");
  if (cap_obs < 64 || cap_ref < 64 || cap_obs > (1 << 22) || cap_ref > (1 << 22) || (cap_obs & 7) || (cap_ref & 7) || !(reach >= 1.0 && reach <= 1e6))
    return ed_fail(ED_ERR_INVALID, "ed_batch_set_emit_tables: caps are multiples of 8 in [64, 2^22], reach in [1, 1e6]");
  b->tab_capY = cap_obs; b->tab_capR = cap_ref; b->tab_reach = reach;
  return ED_OK;
}
ED_CATCH("ed_batch_set_emit_tables")

ED_EXPORT int ed_batch_n_emit_launches(const ed_batch* b)
try {
  if (!b) return 0;
  if (b->fused) return 1;
  int n = 0;
  for (size_t g = 0; g + 1 < b->group_off.size(); ++g) {
    const std::vector<int64_t>& segv = (b->emit_mode == 2 && !b->seg_sm.empty()) ? b->seg_sm : ((b->emit_mode == 1 && !b->seg_t.empty()) ? b->seg_t : b->seg);
    const int64_t nblk = segv[3 * b->group_off[g + 1]] - segv[3 * b->group_off[g]];
    const int64_t head = b->emit_mode == 2 ? ((g > 0 && nblk > 512) ? 128 : 0) : ((g > 0 && nblk > 2 * kEmitHeadBlocks) ? kEmitHeadBlocks : 0);
    n += (head > 0) + (nblk - head > 0);
  }
  if (b->group_off.size() == 2 && b->split_frac > 0.0 && b->split_frac < 1.0 && b->split_ev) n += 1;
  return n;
}
ED_CATCH("ed_batch_n_emit_launches")

ED_EXPORT int ed_batch_keep_loglik(ed_batch* b, int keep)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->keep_loglik = keep != 0;
  if (!b->keep_loglik && b->fused && b->d_loglik) { HIP_TRY(hipFree(b->d_loglik)); b->d_loglik = nullptr; }
  return ED_OK;
}
ED_CATCH("ed_batch_keep_loglik")
ED_EXPORT const uint8_t* ed_batch_path(const ed_batch* b) { return b ? b->d_path : nullptr; }
ED_EXPORT const ed_call* ed_batch_calls(const ed_batch* b) { return b ? b->d_calls : nullptr; }

static int batch_ready(ed_batch* b)
{
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  if (!b->ran) return ed_fail(ED_ERR_STATE, "no ed_batch_run has been issued on this batch");
  HIP_TRY(hipSetDevice(b->plan->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return ED_OK;
}

// The call table is sized by a heuristic when the batch is created; a run that produced more calls than that (up
// to one per exon is possible) gets a table of the right size and the fill kernel is run again -- the records
// are a pure function of the packed path.
ED_EXPORT int ed_batch_n_calls(ed_batch* b, int64_t* n_calls)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!n_calls) return ed_fail(ED_ERR_INVALID, "NULL output");
  if (int rc = ed_d2h(n_calls, b->d_total, 8, b->stream)) return rc;
  if (*n_calls > b->calls_cap) {
    const ed_plan* p = b->plan;
    const int64_t cap = *n_calls + *n_calls / 8 + 1024;
    ed_call* fresh = nullptr;
    if (hipMalloc((void**)&fresh, (size_t)cap * sizeof(ed_call)) != hipSuccess)
      return ed_fail(ED_ERR_NOMEM, "call table: cannot allocate %lld records", (long long)cap);
    (void)hipFree(b->d_calls);
    b->d_calls = fresh;
    b->calls_cap = cap;
    hipLaunchKernelGGL(k_calls_fill, dim3((unsigned)((b->S + kWave - 1) / kWave), (unsigned)p->C), dim3(kWave, kCallSeg), 0, b->stream,
                       b->d_ppath, p->d_chrom_off, p->d_tile_off, b->S, p->C, b->d_offsets, b->d_counts, b->d_calls, b->calls_cap);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(b->stream));
  }
  return ED_OK;
}
ED_CATCH("ed_batch_n_calls")

ED_EXPORT int ed_batch_n_gsl_errors(ed_batch* b, int64_t* n_events)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!n_events) return ed_fail(ED_ERR_INVALID, "NULL output");
  unsigned long long v = 0;
  if (int rc = ed_d2h(&v, b->d_nerr, 8, b->stream)) return rc;
  *n_events = (int64_t)v;
  return ED_OK;
}
ED_CATCH("ed_batch_n_gsl_errors")

ED_EXPORT int ed_batch_copy_calls(ed_batch* b, ed_call* host_calls, int64_t cap)
try {
  int64_t n = 0;
  if (int rc = ed_batch_n_calls(b, &n)) return rc;   // (grows the table if the run needed more records)
  const int64_t k = std::min(n, cap);
  if (k > 0) {
    if (!host_calls) return ed_fail(ED_ERR_INVALID, "NULL output");
    if (int rc = ed_d2h(host_calls, b->d_calls, (size_t)k * sizeof(ed_call), b->stream)) return rc;
  }
  return ED_OK;
}
ED_CATCH("ed_batch_copy_calls")

ED_EXPORT int ed_batch_copy_call_info(ed_batch* b, ed_call_info* host_info, int64_t cap)
try {
  int64_t n = 0;
  if (int rc = ed_batch_n_calls(b, &n)) return rc;   // (grows the table if the run needed more records)
  const int64_t k = std::min(n, cap);
  if (k <= 0) return ED_OK;
  if (!host_info) return ed_fail(ED_ERR_INVALID, "NULL output");
  if (k > b->info_cap) {
    if (b->d_info) { HIP_TRY(hipFree(b->d_info)); b->d_info = nullptr; b->info_cap = 0; }
    const int64_t cap = std::max<int64_t>(k + k / 4, 4096);
    if (hipMalloc((void**)&b->d_info, (size_t)cap * sizeof(ed_call_info)) != hipSuccess)
      return ed_fail(ED_ERR_NOMEM, "call decoration: cannot allocate %lld records", (long long)cap);
    b->info_cap = cap;
  }
  hipLaunchKernelGGL(k_call_info, dim3((unsigned)((k + 127) / 128)), dim3(128), 0, b->stream, b->d_calls, k,
                     (b->keep_loglik || !b->fused) ? (b->rows_valid ? b->d_loglik : b->d_loglik_sm) : (double*)nullptr, b->d_consts, b->last_test, b->last_ref,
                     b->last_expected, b->S, b->d_info, b->last_cov_X, b->last_cov_K, b->last_cov_beta, b->rows_valid ? 3 * b->S : (int64_t)1,
                     b->rows_valid ? b->S : b->Epad, b->rows_valid ? (int64_t)1 : 3 * b->Epad, b->last_layout ? (int64_t)1 : b->S,
                     b->last_layout ? b->plan->E : (int64_t)1, b->last_cb);
  HIP_TRY(hipGetLastError());
  if (int rc = ed_d2h(host_info, b->d_info, (size_t)k * sizeof(ed_call_info), b->stream)) return rc;
  return ED_OK;
}
ED_CATCH("ed_batch_copy_call_info")

ED_EXPORT int ed_batch_copy_path(ed_batch* b, uint8_t* host_path)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!host_path) return ed_fail(ED_ERR_INVALID, "NULL output");
  if (int rc = ed_d2h(host_path, b->d_path, (size_t)b->plan->E * b->S, b->stream)) return rc;
  return ED_OK;
}
ED_CATCH("ed_batch_copy_path")

ED_EXPORT int ed_batch_copy_loglik(ed_batch* b, double* host_loglik)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!host_loglik) return ed_fail(ED_ERR_INVALID, "NULL output");
  if (int rc = ensure_loglik_rows(b)) return rc;
  if ((b->fused && !b->keep_loglik) || !b->d_loglik)
    return ed_fail(ED_ERR_STATE, "the likelihood matrix is not kept (ed_batch_keep_loglik)");
  if (int rc = ed_d2h(host_loglik, b->d_loglik, (size_t)b->plan->E * 3 * b->S * 8, b->stream)) return rc;
  return ED_OK;
}
ED_CATCH("ed_batch_copy_loglik")

ED_EXPORT int ed_batch_verify_emissions(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                                        const double* d_expected, double mixture, int64_t* n_compared, int64_t* n_mismatch,
                                        ed_emit_mismatch* first, int64_t cap)
try {
  if (b && b->counts_bits == 16) return ed_fail(ED_ERR_STATE, "ed_batch_verify_emissions: 16-bit counts are checked by ed_batch_verify_emissions_tol");
  if (int rc = batch_ready(b)) return rc;
  if (!d_test || !d_ref || !d_phi || !d_expected || !n_compared || !n_mismatch || cap < 0 || (cap > 0 && !first))
    return ed_fail(ED_ERR_INVALID, "ed_batch_verify_emissions: bad arguments");
  if (int rc = ensure_loglik_rows(b)) return rc;
  if ((b->fused && !b->keep_loglik) || !b->d_loglik)
    return ed_fail(ED_ERR_STATE, "ed_batch_verify_emissions: the likelihood matrix is not kept (ed_batch_keep_loglik)");
  const int64_t E = b->plan->E, S = b->S;
  *n_compared = 0; *n_mismatch = 0;
  if (E == 0) return ED_OK;
  DevBuf dcnt, dfirst;
  HIP_TRY(dcnt.alloc(24)); HIP_TRY(dfirst.alloc((size_t)cap * sizeof(ed_emit_mismatch)));
  HIP_TRY(hipMemsetAsync(dcnt.p, 0, 24, b->stream));
  const int64_t rows_per_block = (int64_t)(kEmitBlock / 64) * kVerifyRun;
  const int64_t eblk = (E + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(k_emit_verify, dim3((unsigned)((S + 63) / 64), (unsigned)std::min<int64_t>(eblk, 65535), (unsigned)((eblk + 65534) / 65535)),
                     dim3(kEmitBlock), 0, b->stream, d_test, d_ref, d_phi, d_expected, mixture, E, S, b->d_loglik,
                     dcnt.as<unsigned long long>(), dfirst.as<ed_emit_mismatch>(), cap, b->counts_layout ? (int64_t)1 : S, b->counts_layout ? E : (int64_t)1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(b->stream));
  unsigned long long c[3] = {0, 0, 0};
  if (int rc = ed_d2h(c, dcnt.p, 24, b->stream)) return rc;
  *n_mismatch = (int64_t)c[0];
  *n_compared = (int64_t)c[1];
  const int64_t k = std::min<int64_t>((int64_t)c[2], cap);
  if (k > 0) { if (int rc = ed_d2h(first, dfirst.p, (size_t)k * sizeof(ed_emit_mismatch), b->stream)) return rc; }
  return ED_OK;
}
ED_CATCH("ed_batch_verify_emissions")

ED_EXPORT int ed_batch_verify_emissions_tol(ed_batch* b, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                                            const double* d_expected, double mixture, double rel_tol, double abs_tol, int64_t* n_compared,
                                            int64_t* n_beyond, double* max_rel, double* max_abs, ed_emit_mismatch* first, int64_t cap)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!d_test || !d_ref || !d_phi || !d_expected || !n_compared || !n_beyond || cap < 0 || (cap > 0 && !first) || !(rel_tol >= 0) || !(abs_tol >= 0))
    return ed_fail(ED_ERR_INVALID, "ed_batch_verify_emissions_tol: bad arguments");
  if (int rc = ensure_loglik_rows(b)) return rc;
  if ((b->fused && !b->keep_loglik) || !b->d_loglik)
    return ed_fail(ED_ERR_STATE, "ed_batch_verify_emissions_tol: the likelihood matrix is not kept (ed_batch_keep_loglik)");
  const int64_t E = b->plan->E, S = b->S;
  *n_compared = 0; *n_beyond = 0;
  if (max_rel) *max_rel = 0.0;
  if (max_abs) *max_abs = 0.0;
  if (E == 0) return ED_OK;
  DevBuf dcnt, dfirst;
  HIP_TRY(dcnt.alloc(40)); HIP_TRY(dfirst.alloc((size_t)cap * sizeof(ed_emit_mismatch)));
  HIP_TRY(hipMemsetAsync(dcnt.p, 0, 40, b->stream));
  const int64_t rows_per_block = (int64_t)(kEmitBlock / 64) * kVerifyRun;
  const int64_t eblk = (E + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(k_emit_verify_tol, dim3((unsigned)((S + 63) / 64), (unsigned)std::min<int64_t>(eblk, 65535), (unsigned)((eblk + 65534) / 65535)),
                     dim3(kEmitBlock), 0, b->stream, d_test, d_ref, d_phi, d_expected, mixture, E, S, b->d_loglik, rel_tol, abs_tol,
                     dcnt.as<unsigned long long>(), dfirst.as<ed_emit_mismatch>(), cap, b->counts_layout ? (int64_t)1 : S, b->counts_layout ? E : (int64_t)1, b->counts_layout ? b->cb() : 4);
  HIP_TRY(hipGetLastError());
  unsigned long long c[5] = {0, 0, 0, 0, 0};
  if (int rc = ed_d2h(c, dcnt.p, 40, b->stream)) return rc;
  *n_beyond = (int64_t)c[0];
  *n_compared = (int64_t)c[1];
  double d;
  if (max_rel) { std::memcpy(&d, &c[3], 8); *max_rel = d; }
  if (max_abs) { std::memcpy(&d, &c[4], 8); *max_abs = d; }
  const int64_t k = std::min<int64_t>((int64_t)c[2], cap);
  if (k > 0) { if (int rc = ed_d2h(first, dfirst.p, (size_t)k * sizeof(ed_emit_mismatch), b->stream)) return rc; }
  return ED_OK;
}
ED_CATCH("ed_batch_verify_emissions_tol")

// emit mode 1: one sample's tables as the last run built them (test / diagnostic accessor)
ED_EXPORT int ed_batch_copy_emit_tables(ed_batch* b, int64_t sample, int32_t dims[2], double* entries, int64_t cap_entries)
try {
  if (int rc = batch_ready(b)) return rc;
  if (!dims || sample < 0 || sample >= b->S) return ed_fail(ED_ERR_INVALID, "ed_batch_copy_emit_tables: bad arguments");
  if (!b->d_tabs || b->emit_mode < 1) return ed_fail(ED_ERR_STATE, "ed_batch_copy_emit_tables: the batch does not run a table-driven emit mode");
  int4 d, w;
  if (int rc = ed_d2h(&d, b->d_tdims + sample, 16, b->stream)) return rc;
  if (int rc = ed_d2h(&w, b->d_twins + sample, 16, b->stream)) return rc;
  // (a tail sample's tables are its LDS windows, back to back: the lengths reported are the BUILT ones -- obs n1, ref n2 -- and of the n1 + n2
  //  entries of the tot table's slice only the first n3 are made: ed_batch_copy_table_windows)
  if (w.w) { d.x = w.x; d.y = w.y; }
  dims[0] = d.x; dims[1] = d.y;
  const int64_t n = std::min<int64_t>(2 * ((int64_t)d.x + d.y), cap_entries);
  if (n > 0) {
    if (!entries) return ed_fail(ED_ERR_INVALID, "NULL output");
    if (int rc = ed_d2h(entries, b->d_tabs + sample * b->tab_stride * 3, (size_t)n * 24, b->stream)) return rc;
  }
  return ED_OK;
}
ED_CATCH("ed_batch_copy_emit_tables")

// sample-major table mode: (n1, n2, n3, tail) of one sample in the last run -- the entries of its obs / ref / tot table a workgroup keeps in LDS, and
// whether it is a tail sample (its counts outgrow the windows: tables built for the windows only, Stirling's series beyond, served up to the
// conditioning limit that ed_batch_copy_table_dims then reports as Ly / Lr)
ED_EXPORT int ed_batch_copy_table_windows(ed_batch* b, int64_t sample, int32_t out[4])
try {
  if (int rc = batch_ready(b)) return rc;
  if (!out || sample < 0 || sample >= b->S) return ed_fail(ED_ERR_INVALID, "ed_batch_copy_table_windows: bad arguments");
  if (!b->d_tabs || b->emit_mode < 1) return ed_fail(ED_ERR_STATE, "ed_batch_copy_table_windows: the batch does not run a table-driven emit mode");
  int4 w;
  if (int rc = ed_d2h(&w, b->d_twins + sample, 16, b->stream)) return rc;
  out[0] = w.x; out[1] = w.y; out[2] = w.z; out[3] = w.w;
  return ED_OK;
}
ED_CATCH("ed_batch_copy_table_windows")

// emit mode 2: 1 (default) = samples whose counts outgrow the LDS windows are tail samples (edtab.inc); 0 = every sample on full-length tables
// (round 5's behaviour: look-ups beyond the windows in global memory, cells beyond the length caps on the strict lists)
ED_EXPORT int ed_batch_set_emit_tails(ed_batch* b, int on)
try {
  if (!b) return ed_fail(ED_ERR_INVALID, "NULL batch");
  b->tab_tails = on ? 1 : 0;
  return ED_OK;
}
ED_CATCH("ed_batch_set_emit_tails")

// table-driven modes: what the last run left to the strict arithmetic
//   out[0] cells on the strict lists (outside their sample's tables, or under its few-reads rule), summed over the run's launch groups
//   out[1] samples without tables (walked whole by the strict pass)      out[2] launch groups whose lists ran out (full re-scan)
//   out[3] cells of the samples without tables
ED_EXPORT int ed_batch_table_stats(ed_batch* b, int64_t out[4])
try {
  if (int rc = batch_ready(b)) return rc;
  if (!out) return ed_fail(ED_ERR_INVALID, "NULL output");
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!b->d_cold_n || b->emit_mode < 1) return ED_OK;
  unsigned long long v[4] = {0, 0, 0, 0};
  if (int rc = ed_d2h(v, b->d_nerr + 2, 32, b->stream)) return rc;
  out[0] = (int64_t)v[0]; out[1] = (int64_t)v[3]; out[2] = (int64_t)v[1]; out[3] = (int64_t)v[2];
  return ED_OK;
}
ED_CATCH("ed_batch_table_stats")

ED_EXPORT int ed_batch_n_cold_cells(ed_batch* b, int64_t* n_cells)
try {
  if (!n_cells) return ed_fail(ED_ERR_INVALID, "NULL output");
  int64_t v[4];
  if (int rc = ed_batch_table_stats(b, v)) return rc;
  *n_cells = v[0];
  return ED_OK;
}
ED_CATCH("ed_batch_n_cold_cells")

// one sample's (Ly, Lr, Tm1, reason) of the last run (edtab.inc: tab_dims_of)
ED_EXPORT int ed_batch_copy_table_dims(ed_batch* b, int64_t sample, int32_t dims[4])
try {
  if (int rc = batch_ready(b)) return rc;
  if (!dims || sample < 0 || sample >= b->S) return ed_fail(ED_ERR_INVALID, "ed_batch_copy_table_dims: bad arguments");
  if (!b->d_tabs || b->emit_mode < 1) return ed_fail(ED_ERR_STATE, "ed_batch_copy_table_dims: the batch does not run a table-driven emit mode");
  int4 d;
  if (int rc = ed_d2h(&d, b->d_tdims + sample, 16, b->stream)) return rc;
  dims[0] = d.x; dims[1] = d.y; dims[2] = d.z; dims[3] = d.w;
  return ED_OK;
}
ED_CATCH("ed_batch_copy_table_dims")

ED_EXPORT int ed_batch_stage_ms_total(ed_batch* b, double ms_total[5], int64_t* n_runs, int64_t* n_fits)
try {
  if (!b || !ms_total) return ed_fail(ED_ERR_INVALID, "NULL argument");
  if (int rc = fold_run_times(b)) return rc;
  if (int rc = fold_fit_time(b)) return rc;
  for (int i = 0; i < 5; ++i) ms_total[i] = b->stage_total[i];
  if (n_runs) *n_runs = b->n_runs_timed;
  if (n_fits) *n_fits = b->n_fits_timed;
  return ED_OK;
}
ED_CATCH("ed_batch_stage_ms_total")

ED_EXPORT int ed_batch_stage_ms(ed_batch* b, float ms[5])
try {
  if (!b || !ms) return ed_fail(ED_ERR_INVALID, "NULL argument");
  if (int rc = fold_run_times(b)) return rc;     // (a pending pair of events becomes the "last" value)
  if (int rc = fold_fit_time(b)) return rc;
  for (int i = 0; i < 5; ++i) ms[i] = b->last_ms[i];
  return ED_OK;
}
ED_CATCH("ed_batch_stage_ms")

#include "edrefset.inc"
#include "edbins.inc"
#include "edcov.inc"
#include "edcohort.inc"
#include "edmulti.inc"
#include "edrefcohort.inc"
