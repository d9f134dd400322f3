/* exomedepth_amd.h -- C-ABI of libedcore.so, the MI355X (gfx950) CNV-calling core.
 *
 * Drop-in boundary.  The reference crosses from R into native code at exactly two .Call entries,
 * registered in reference src/ExomeDepth_init.c:14-18:
 *     get_loglike_matrix/5   (reference src/CNV_estimate.cpp:52-85, called at R/class_definition.R:184-189)
 *     C_hmm/6                (reference src/hmm.cpp:18-167,         called at R/tools.R:97)
 * ed_get_loglike_matrix() and ed_hmm() below are those two entries with the SEXP wrappers peeled off
 * (plain pointers and sizes; the R shim that re-wraps them is shown in INTEGRATION.md).  They take
 * HOST buffers, copy in, launch on the GPU, synchronise and copy out inside the call, as a .Call must.
 *
 * The reference's granularity -- one sample, one chromosome per call -- cannot fill a GPU, so the
 * library adds a batched interface (ed_plan_* / ed_batch_*): one plan per exon design, one batch per
 * slab of samples, device-resident inputs and outputs.  It computes, per (exon, sample) cell, the
 * same three log-likelihoods, and per (sample, chromosome) chain the same Viterbi path and call
 * table, as running the two entries above in the loop of reference R/class_definition.R:354-414.
 *
 * Conventions: every function returns 0 on success or a negative ed_status; ed_last_error() gives a
 * message for the calling thread.  No torch / HIP types appear in signatures: device pointers are
 * plain pointers and a stream is passed as void* (a hipStream_t; NULL = the default stream).
 * There is no CPU fallback: without a usable gfx950 device every compute entry fails with
 * ED_ERR_NO_DEVICE.
 */
#ifndef EXOMEDEPTH_AMD_H
#define EXOMEDEPTH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ED_OK = 0,
  ED_ERR_INVALID = -1,   /* bad argument (NULL pointer, negative size, nstates != 3 ...) */
  ED_ERR_NO_DEVICE = -2, /* no usable HIP device */
  ED_ERR_HIP = -3,       /* a HIP runtime call failed; see ed_last_error() */
  ED_ERR_NOMEM = -4,
  ED_ERR_STATE = -5      /* call sequence error (e.g. results requested before ed_batch_run) */
} ed_status;

/* ---- library / device ---- */
const char* ed_version(void);
const char* ed_last_error(void);
int ed_device_count(void);
/* name[<=256] receives the gcnArchName ("gfx950:sramecc+:xnack-") */
int ed_device_info(int device, char* name, size_t name_len, int* compute_units, size_t* total_mem);

/* =====================================================================================
 * 1. Per-sample drop-ins (host buffers) -- the reference's two .Call entries
 * ===================================================================================== */

/* get_loglike_matrix: reference src/CNV_estimate.cpp:52-85.
 *   phi[n], expected[n], total[n], observed[n], mixture  -- as the five SEXP arguments (:16)
 *   out[3*n]  column-major n x 3: column 0 deletion, 1 normal, 2 duplication (:69, :75-77)
 *   n_gsl_errors (optional) receives the number of events for which the reference would have printed
 *   a GSL error (src/error.c:45-48): NaN/zero shape parameters.  Values on those rows match the
 *   reference's (NaN, or 0.0 for NaN inputs). */
int ed_get_loglike_matrix(const double* phi, const double* expected, const int32_t* total, const int32_t* observed,
                          int64_t n, double mixture, double* out, int64_t* n_gsl_errors);

/* What the reference PRINTS while it computes that matrix for rows outside the model's domain: every gsl_error() call
 * is two Rprintf lines, "ERROR <file> <line> <reason>\n" and "Default GSL error handler invoked.\n" (src/error.c:45-48;
 * the sites are src/beta.c:44, :56, :59, :163 and src/VP_gamma.c:803, :1239, :1253, :1261, :1283), in the reference's
 * order.  The events happen on the device, so the library hands the text back and the R shim prints it.
 * buf[cap] receives at most cap - 1 characters + NUL; *needed = length of the whole text (call with cap = 0 to size it).
 * Same arguments as ed_get_loglike_matrix. */
int ed_get_loglike_matrix_messages(const double* phi, const double* expected, const int32_t* total, const int32_t* observed,
                                   int64_t n, double mixture, char* buf, size_t cap, size_t* needed);

/* C_hmm: reference src/hmm.cpp:18-167.
 *   nstates must be 3 (else ED_ERR_INVALID; the reference prints and returns a C NULL, :37-40)
 *   transitions[9]        3x3 column-major (:25)
 *   probabilities[3*nobs] nobs x 3 column-major in HMM order normal, deletion, duplication (:26)
 *   positions[nobs], expected_length (:29-30)
 *   path_out[nobs]        Viterbi state per observation, as doubles like the reference's REALSXP (:139-141)
 *   calls_out[4*calls_cap] column-major calls_cap x 4 (start.p, end.p, type, nexons), 1-based (:111-121, :145-149);
 *                         row r of column c is calls_out[c*calls_cap + r]
 *   n_calls               number of calls found (if > calls_cap only calls_cap rows were written) */
int ed_hmm(int32_t nstates, int32_t nobs, const double* transitions, const double* probabilities,
           const int32_t* positions, double expected_length, double* path_out, double* calls_out, int64_t calls_cap,
           int64_t* n_calls);

/* The two entries above keep one process-wide scratch between calls -- a device block, a pinned host block, a stream, and the host-computed
 * log-transitions (src/hmm.cpp:62-79) of the chains seen last: CallCNVs calls C_hmm with the same positions for every sample
 * (R/class_definition.R:354-374).  Calls are serialised (R's API is single-threaded).  This releases all of it; the next call starts afresh. */
void ed_dropin_release(void);

/* =====================================================================================
 * 2. Batched interface (device-resident)
 * ===================================================================================== */

/* One CNV call.  exon indices are 0-based, inclusive, into the plan's exon order; they equal the
 * reference's start.p-1 / end.p-1 after its dummy-exon and shift corrections
 * (R/class_definition.R:371-372, :409-410).  type: 1 deletion, 2 duplication (src/hmm.cpp:115). */
typedef struct {
  int32_t sample;     /* column of the batch */
  int32_t chrom;      /* chromosome index of the plan */
  int32_t start_exon; /* first exon of the call */
  int32_t end_exon;   /* last exon of the call */
  int32_t type;
  int32_t nexons;     /* the reference's nexons counter, quirks included (src/hmm.cpp:104-126) */
} ed_call;

/* Decoration of one call, reference R/class_definition.R:379-405 (same order as the call table). */
typedef struct {
  double BF_raw;           /* log10(e) * sum(loglik[type] - loglik[normal]) over the call's exons (:390-394, :404) */
  double BF;               /* signif(BF_raw, 3) (:404) */
  int64_t reads_expected;  /* as.integer(sum(total * expected)) (:396, :402) */
  int64_t reads_observed;  /* sum(test) (:397) */
  double reads_ratio;      /* signif(reads.observed / reads.expected, 3) (:403) */
} ed_call_info;

typedef struct ed_plan ed_plan;
typedef struct ed_batch ed_batch;

/* A plan fixes the exon design and the HMM parameters of CallCNVs (reference R/class_definition.R:311,
 * defaults transition.probability=1e-4, expected.CNV.length=5e4 at :261):
 *   exons must already be ordered by (chromosome, position) as :323-336 orders them;
 *   chrom_off[n_chrom+1] delimits the chromosomes (chains, :354); start/end are the exon coordinates.
 * Creating the plan builds, on the host with libm exactly as src/hmm.cpp:62-79 does, the
 * distance-dependent log-transition table of every exon gap (8 doubles per gap, shared by all
 * samples: for each of the 4 quad lanes the two distance-dependent entries of its into-state;
 * the from-normal entry does not depend on the distance) and uploads it. */
int ed_plan_create(ed_plan** plan, int device, int64_t n_exons, int32_t n_chrom, const int32_t* chrom_off,
                   const int32_t* start, const int32_t* end, double transition_probability,
                   double expected_cnv_length);
void ed_plan_destroy(ed_plan* plan);
int64_t ed_plan_n_exons(const ed_plan* plan);

/* A batch owns the device working set for n_samples samples of one plan:
 *   log-likelihoods  double [n_exons][3][n_samples]  (deletion, normal, duplication; sample-minor)
 *   Viterbi path     uint8  [n_exons][n_samples]     (0 normal, 1 deletion, 2 duplication)
 *   call table       ed_call[n_calls], ordered by (sample, chromosome, position) */
int ed_batch_create(ed_batch** batch, ed_plan* plan, int64_t n_samples);
/* the n_samples the batch was made for (0 for NULL) -- e.g. of a cohort ticket's batch (ed_cohort_batch), which is the slab's width */
int64_t ed_batch_n_samples(const ed_batch* batch);
void ed_batch_destroy(ed_batch* batch);

/* Fit the per-sample beta-binomial model  cbind(test, reference) ~ 1  (what aod::betabin does at
 * reference R/class_definition.R:118): writes phi[n_samples] and expected[n_samples] (DEVICE pointers).
 * d_test/d_ref: int32 [n_exons][n_samples] sample-minor DEVICE matrices. */
int ed_batch_fit(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, double* d_phi, double* d_expected,
                 void* stream);
/* Convergence of the last ed_batch_fit / ed_batch_fit_subset: number of samples whose Newton iteration ended on its
 * iteration budget instead of a step below tolerance (their phi / expected are the last iterate), and the first such
 * sample (-1 if none).  Synchronises the fit's stream. */
int ed_batch_fit_n_unconverged(ed_batch* batch, int64_t* n_unconverged, int32_t* first_sample);
/* How ed_batch_fit iterates: 1 (default) on per-sample count histograms built in one pass over the counts
 * (every Newton iteration then costs ~9 000 digamma evaluations per sample instead of 3 x n_exons); 0 per cell on
 * every pass.  Same maximum; the two differ by summation order only.  The histograms come in three geometries (unit
 * bins up to 4096 / 8192 / 16384 for the reference and total counts), picked on the device from the batch's depth;
 * on = 8, 4 or 2 asks for one of them (what the tests do), 1 leaves the choice to the data. */
int ed_batch_set_fit_histograms(ed_batch* batch, int on);
/* Which estimate ed_batch_fit returns.
 *   0 (default)  the maximum-likelihood estimate, by Newton's method on the count histograms (converged to 1e-9);
 *   1 "aod-nm"   the procedure the reference runs: aod::betabin's objective (binomial coefficients included) minimised by
 *                optim()'s Nelder-Mead (R's nmmin: alpha 1, beta 0.5, gamma 2, reltol sqrt(eps) on the objective, maxit 2000)
 *                from aod's start (glm-binomial intercept logit(sum y / sum n), phi = 0.1), evaluated on the same histograms.
 *                It stops where Nelder-Mead stops -- within ~1e-3 of the maximum in phi -- so its (phi, expected) are
 *                reference-LIKE parameters; aod is not in the reference tree, so neither mode is pinned against it. */
int ed_batch_set_fit_mode(ed_batch* batch, int mode);
/* The same fit on every `by`-th exon only (exons 0, by, 2*by, ...): the scalar form of subset.for.speed,
 * reference R/class_definition.R:107-113, where by = floor(n_exons / subset.for.speed).  by = 1 is ed_batch_fit. */
int ed_batch_fit_subset(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, int64_t by, double* d_phi,
                        double* d_expected, void* stream);

/* Depth-binned dispersion, `phi.bins > 1` of the reference's initialiser (R/class_definition.R:120-147), per sample:
 *   edges     double [(phi_bins + 1)][n_samples]  complete.bins: seq(0, q85, by = q85/(phi_bins-1)), max + 1  (:124-126)
 *   phi_bins  double [phi_bins][n_samples]        one dispersion per level of depth.quant, common intercept     (:135-139)
 *   expected  double [n_samples]                  fitted(mod)
 * 2 <= phi_bins <= 8.  Fails with "Binning did not happen properly" (:130-133) if a level of some sample is empty.
 * Synchronises the stream before returning (the level check is done on the host). */
int ed_batch_fit_bins(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, int phi_bins, double* d_phi_bins,
                      double* d_edges, double* d_expected, void* stream);
/* Which form the last ed_batch_fit_bins took: 1 = count histograms (three passes over the counts: unit bins of the reference
 * counts for the quantile, per-level bins of the test count and of the total for the Newton sums), 0 = per cell on every pass
 * (what ed_batch_set_fit_histograms(batch, 0) asks for, and what data beyond the bins -- a 0.85 quantile of the reference counts
 * >= 8192, > 32768 cells of a sample outside its level's bins -- fall back to).  Same estimate to the fit's tolerance. */
int ed_batch_fit_bins_form(const ed_batch* batch);
/* Samples the last depth-binned fit (ed_batch_fit_bins, or the cohort pipeline's fit of this batch's slab once its ticket has been
 * waited for) left short of its tolerance after the 40-pass budget -- the depth-binned Newton's own flags, not those of the
 * single-dispersion fit it starts from (ed_batch_fit_n_unconverged).  ed_cohort_run_status reports their sum in phi_bins mode. */
int ed_batch_fit_bins_n_unconverged(const ed_batch* batch, int64_t* n);
/* ed_batch_run with the per-exon dispersion phi.linear = approxfun(bin mid-points, phi.estimates)(reference)
 * (:141-147) evaluated on the fly; everything downstream of the emissions is ed_batch_run's.  Not available in
 * fused mode. */
int ed_batch_run_bins(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, int phi_bins, const double* d_phi_bins,
                      const double* d_edges, const double* d_expected, double mixture, void* stream);
/* The S4 `phi` slot of that model: d_phi_out double [n_exons][n_samples] = phi.linear. */
int ed_batch_phi_linear(ed_batch* batch, const int32_t* d_ref, int phi_bins, const double* d_phi_bins, const double* d_edges,
                        double* d_phi_out, void* stream);

/* Covariates in the mean model: `data` + `formula = cbind(test, reference) ~ x1 + ... + xK` of the reference's
 * initialiser (R/class_definition.R:86-118, :168).  aod::betabin fits one coefficient per column of the model matrix
 * (logit link) and one dispersion; `expected` = fitted(mod) = plogis(X beta) is per exon.
 *   d_X     double [n_exons][n_cov]  covariates, exon-major, shared by the samples of the batch (0 <= n_cov <= 3)
 *   d_beta  double [(n_cov + 1)][n_samples]  intercept + slopes        d_phi  double [n_samples]
 * ed_batch_fit_cov synchronises the stream before returning.  ed_batch_run_cov is ed_batch_run with the per-exon
 * expected evaluated on the fly (not available in fused mode); ed_batch_expected_cov writes the S4 `expected`
 * slot, double [n_exons][n_samples]. */
int ed_batch_fit_cov(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, const double* d_X, int n_cov, double* d_beta,
                     double* d_phi, void* stream);
int ed_batch_run_cov(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, const double* d_X, int n_cov,
                     const double* d_beta, const double* d_phi, double mixture, void* stream);
int ed_batch_expected_cov(ed_batch* batch, const double* d_X, int n_cov, const double* d_beta, double* d_expected_out, void* stream);

/* Emissions + Viterbi + call segmentation for the whole batch.  All pointers are DEVICE pointers.
 * d_phi/d_expected: per-sample dispersion and expected proportion (from ed_batch_fit or given).
 * Asynchronous on `stream`; results are valid after the stream is synchronised (the accessors that
 * return host data synchronise themselves). */
int ed_batch_run(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                 const double* d_expected, double mixture, void* stream);

/* Execution mode of ed_batch_run.
 *   fused = 0 (default)  two kernels: emissions into the [n_exons][3][n_samples] likelihood matrix, then Viterbi,
 *                        overlapped by chromosome groups on two streams -- the fastest path today;
 *   fused = 1            ONE kernel: producer waves hand each tile of emissions to a Viterbi wave through LDS
 *                        (csrc/edfused.inc).  HBM traffic drops from ~32 to ~10 bytes per cell and the likelihood
 *                        matrix becomes an optional by-product (ed_batch_keep_loglik).  Same results bit for bit.
 * ed_batch_keep_loglik(batch, 0) (fused mode only): do not materialise the matrix -- ed_batch_loglik() returns NULL,
 * ed_batch_copy_loglik() fails with ED_ERR_STATE, 3*8*n_exons*n_samples bytes of HBM are not allocated. */
int ed_batch_set_fused(ed_batch* batch, int fused);
int ed_batch_keep_loglik(ed_batch* batch, int keep);
/* Number of emission-kernel launches one ed_batch_run issues (= overlap groups; 1 in fused mode): the
 * denominator of per-launch figures in bench.py. */
int ed_batch_n_emit_launches(const ed_batch* batch);

/* Pipelining consecutive batches.  By default ed_batch_run joins everything it started back into `stream`: the stream is
 * then blocked until the last Viterbi chains and the call table of this batch are done (~0.5 ms at 200 000 x 1024 during
 * which the chip is nearly idle).  With ed_batch_set_async_tail(batch, 1) the caller's stream carries the emission
 * kernels only; the Viterbi groups, the trace-back and the call table finish on streams of the batch, and
 *   - the accessors below (ed_batch_n_calls, ed_batch_copy_*, ...) wait for them, as before;
 *   - ed_batch_wait(batch, stream) makes another stream wait for them (device-side, no host synchronisation);
 *   - the next ed_batch_run on the SAME batch waits for them by itself (its buffers are reused).
 * Two batches used alternately on one stream, the dispersion fit of the next batch issued on a second stream while the
 * current one runs, give a two-deep pipeline: fit(N+1) and the tail of N execute underneath the VALU-bound emissions
 * of N / N+1 (bench.py does exactly this).  Results are unchanged bit for bit.  Not available in fused mode (ignored).
 * The caller owns the hazards on ITS buffers: d_phi / d_expected handed to ed_batch_run are read by the first kernels of
 * that run, so a later fit that overwrites them must be ordered after those kernels (an event recorded on the run's
 * stream after ed_batch_run returns is enough). */
int ed_batch_set_async_tail(ed_batch* batch, int on);
int ed_batch_wait(ed_batch* batch, void* stream);
/* Where the Viterbi chains of a batch run relative to ITS OWN emissions.  1 (default): the chromosomes are cut into a few
 * groups and the chains of a group run underneath the emissions of the following groups (what a lone batch wants: only the
 * last, short chains are exposed).  0: one group -- all emissions as one launch, then all chains.  Co-running chains and
 * emissions costs the (VALU-bound) emissions about as much as it hides; in a pipeline of batches the chains of batch N are
 * better run next to the dispersion fit of batch N+1, which leaves the VALUs idle: asynchronous tail + 0 here. */
int ed_batch_set_viterbi_overlap(ed_batch* batch, int on);

/* device-resident results of the last ed_batch_run */
const double* ed_batch_loglik(const ed_batch* batch);  /* [n_exons][3][n_samples]; emit mode 2 keeps the matrix as [n_samples][3][n_exons + pad]:
                                                         * the first call after a run then ENQUEUES the conversion on the run's stream (and allocates
                                                         * the [n_exons][3][n_samples] form once) -- not for two threads at a time; NULL + ed_last_error()
                                                         * on failure */
const uint8_t* ed_batch_path(const ed_batch* batch);   /* [n_exons][n_samples]    */
const ed_call* ed_batch_calls(const ed_batch* batch);  /* device array, length ed_batch_n_calls(); call that first: it
                                                         * also re-sizes the table if the run produced more calls than
                                                         * the batch had provisioned */
/* synchronises the stream of the last run */
int ed_batch_n_calls(ed_batch* batch, int64_t* n_calls);
int ed_batch_n_gsl_errors(ed_batch* batch, int64_t* n_events);
/* copy results to host buffers (each synchronises) */
int ed_batch_copy_calls(ed_batch* batch, ed_call* host_calls, int64_t cap);
/* decoration of the calls (computed on the device from the inputs of the last ed_batch_run, which must still
 * be valid); host_info[i] belongs to call i of ed_batch_copy_calls */
int ed_batch_copy_call_info(ed_batch* batch, ed_call_info* host_info, int64_t cap);
int ed_batch_copy_path(ed_batch* batch, uint8_t* host_path /* [n_exons][n_samples] */);
int ed_batch_copy_loglik(ed_batch* batch, double* host_loglik /* [n_exons][3][n_samples] */);

/* Self-check of the emissions of the last ed_batch_run (default model, two-kernel mode or fused mode with the
 * matrix kept).  ed_batch_run evaluates a cell's three log-likelihoods through per-sample hoisted constants, per-sample
 * tables and route binning; this entry evaluates every cell again ON THE DEVICE with the reference's own loop
 * (src/CNV_estimate.cpp:71-81: shape parameters from (phi, expected), six log-Betas, nothing hoisted or tabulated)
 * and compares the bits with the likelihood matrix -- no host round trip, so whole batches can be soaked.
 * Inputs are the DEVICE arrays the run was given.  n_compared / n_mismatch count values (3 per cell); the first
 * `cap` mismatches are returned in `first` (may be NULL with cap = 0).  Synchronous. */
typedef struct {
  int64_t exon, sample;
  int32_t state;            /* 0 deletion, 1 normal, 2 duplication */
  int32_t observed, total;
  int32_t pad_;
  double got, want;         /* the likelihood matrix's value / the per-cell evaluation */
} ed_emit_mismatch;
int ed_batch_verify_emissions(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                              const double* d_expected, double mixture, int64_t* n_compared, int64_t* n_mismatch,
                              ed_emit_mismatch* first, int64_t cap);

/* Emission mode of ed_batch_run (default model: per-sample phi and expected).
 *   0 (default) "strict"  every log-Beta through GSL's evaluation routes operation for operation (reference src/beta.c:49-114,
 *               src/VP_gamma.c): bit-identical to the CPU checker's portable flavour; ~1 140 binary64 instructions per cell.
 *   2 "tables, sample-major"  mode 1's arithmetic (same tables, same values) organised by SAMPLE: a workgroup keeps the hot part of
 *               one sample's tables in LDS (147 KB = 6 144 entries) and walks that sample's exons -- counts read as [n_samples][n_exons],
 *               likelihood matrix written as [n_samples][3][n_exons + padding], which the Viterbi kernel of this mode reads through an
 *               LDS transposition; the documented [n_exons][3][n_samples] form is made when an accessor asks for it.  Mode 1 walks
 *               [n_exons][n_samples] tiles and gathers the tables through L1/L2, where every gather misses L1: 3x slower.
 *   1 "tables"  log B(x, y) = lgamma(x) + lgamma(y) - lgamma(x + y) (the identity at reference src/beta.c:101-108) turns a cell's
 *               emission log B(a1 + obs, a2 + tot - obs) - log B(a1, a2) (src/CNV_estimate.cpp:44-50) into
 *                   D(a1, obs) + D(a2, tot - obs) - D(a1 + a2, tot),   D(x, k) = lgamma(x + k) - lgamma(x) = sum_{i<k} log(x + i),
 *               three per-(sample, state) tables over the sample's count ranges, built before every run as double-double prefix
 *               sums of logarithms (each entry within 0.5001 ulp of the exact sum); a cell is three gathers and a compensated
 *               sum.  Log-likelihoods agree with mode 0 / the reference to ~1e-14 relative (1e-10 is the bar); cells whose counts
 *               lie beyond their sample's tables, and samples whose parameters the tables do not serve (GSL error events,
 *               non-positive shape parameters, expected < ~1e-4), go through mode 0's arithmetic -- same bits, same error
 *               counts.  Not available with phi_bins > 1, covariates or fused mode (those runs stay strict).
 * ed_batch_set_emit_tables (before the mode is first set): longest obs / ref table of a sample in entries (multiples of 8;
 * defaults 4096 / 32768; the tot table has their sum; 48 (cap_obs + cap_ref) bytes of HBM per sample) and `reach`: a sample's
 * tables cover reach x its mean count + 64 (default 8). */
int ed_batch_set_emit_mode(ed_batch* batch, int mode);
/* Layout of the DEVICE count matrices handed to ed_batch_fit* and ed_batch_run: 0 (default) int32 [n_exons][n_samples]
 * (sample-minor); 1 int32 [n_samples][n_exons] (sample-major) -- the memory image of R's column-major n_exons x n_samples integer
 * matrix, i.e. what the reference's user holds (R/class_definition.R:82: test / reference vectors are its columns).  Layout 1 is
 * served by the histogram fit and by emit mode 2 (which works sample-major throughout: with layout 0 it first transposes the
 * counts); the depth-binned and covariate models take layout 0. */
int ed_batch_set_counts_layout(ed_batch* batch, int layout);
/* Width of the DEVICE counts: 32 (default) int32 -- R's integers; 16: uint16 [n_samples][n_exons] (counts below 65 536; the pointers are passed as
 * const int32_t* all the same) -- half the bytes of every pass over the counts (moments, histograms, table statistics, emissions, strict pass,
 * decoration).  Served by counts_layout 1 + emit mode 2 + the per-sample dispersion model only; anything else returns ED_ERR_STATE when it is run.
 * Same results as the int32 form (the fit to its tolerance: a sample's overflow cells are met in another order). */
int ed_batch_set_counts_bits(ed_batch* batch, int bits);
int ed_batch_set_emit_tables(ed_batch* batch, int32_t cap_obs, int32_t cap_ref, double reach);
/* Tolerance form of ed_batch_verify_emissions: |matrix - per-cell evaluation| <= max(abs_tol, rel_tol |per-cell value|) (NaN matches
 * NaN).  n_beyond counts values outside it; max_rel / max_abs (optional) the largest differences seen among finite values. */
int ed_batch_verify_emissions_tol(ed_batch* batch, const int32_t* d_test, const int32_t* d_ref, const double* d_phi,
                                  const double* d_expected, double mixture, double rel_tol, double abs_tol, int64_t* n_compared,
                                  int64_t* n_beyond, double* max_rel, double* max_abs, ed_emit_mismatch* first, int64_t cap);
/* Table-mode diagnostics: one sample's tables as the last run built them -- dims = (Ly, Lr); entries [2 (Ly + Lr)][3]: the obs
 * table (Ly entries), the ref table (Lr), the tot table (Ly + Lr), each entry (deletion, normal, duplication).
 * ed_batch_copy_table_dims: (Ly, Lr, Tm1, w) of a sample -- a cell is served by the tables when obs < Ly, ref < Lr, not
 * 0 < tot <= Tm1 (few reads under a nearly binomial model) and not (ref = 0 and obs >= w) (the reference rounds its second argument
 * (a2 + total) - observed at the size of the total: with a2 << 1 its own value is off by more than the bar) -- in both cases the
 * REFERENCE's value is too noisy for a 1e-10 relative comparison, so the cell takes the reference's arithmetic.  Ly = 0: the sample
 * has no tables and w is the reason (1 GSL error in a per-sample constant, 2 shape parameters not positive normal numbers,
 * 3 ill-conditioned at any useful table length, 4 too many few-read cells); it is evaluated whole by the strict arithmetic.
 * ed_batch_table_stats: what the last run left to the strict arithmetic -- out[0] cells on the strict lists (all launch groups),
 * out[1] samples without tables, out[2] launch groups whose lists ran out (every cell looked at again), out[3] cells of the
 * samples without tables.  ed_batch_n_cold_cells = out[0]. */
int ed_batch_copy_emit_tables(ed_batch* batch, int64_t sample, int32_t dims[2], double* entries, int64_t cap_entries);
int ed_batch_copy_table_dims(ed_batch* batch, int64_t sample, int32_t dims[4]);
/* Sample-major table mode (emit mode 2), one sample of the last run: out = (n1, n2, n3, tail).  n1 / n2 / n3: the entries of its obs / ref / tot table
 * a workgroup keeps in LDS.  tail = 1: a TAIL sample -- its counts outgrow those windows (mean total >= 1.25 x the tot window: ~250 reads per exon and
 * test sample upwards with 8 x deeper references); its tables are built for the windows only, an index beyond a window is served by Stirling's series
 * (csrc/ed_dtab.h: ed_dtab_tail) and its cells are served up to the conditioning limit, which ed_batch_copy_table_dims then reports as Ly / Lr.
 * ed_batch_copy_emit_tables reports a tail sample's BUILT lengths (n1, n2); of its tot slice the first n3 entries are made. */
int ed_batch_copy_table_windows(ed_batch* batch, int64_t sample, int32_t out[4]);
/* 1 (default): tail samples as above; 0: every sample on full-length tables (look-ups beyond the windows in global memory, cells beyond the length
 * caps of ed_batch_set_emit_tables on the strict lists -- round 5's behaviour; the time per cell then grows with the depth: 2.2 x at 400 reads per
 * exon, 7 x at 1 600).  Values within the same 1e-10 of the reference's arithmetic either way.  Cohort option "emit_tails". */
int ed_batch_set_emit_tails(ed_batch* batch, int on);
int ed_batch_table_stats(ed_batch* batch, int64_t out[4]);
int ed_batch_n_cold_cells(ed_batch* batch, int64_t* n_cells);

/* Per-stage device times of the last ed_batch_run / ed_batch_fit, measured with HIP events on the
 * stream the kernels were launched on (enable before the run; costs nothing when disabled).
 * ms[]: 0 sample constants, 1 emissions, 2 Viterbi (forward + trace-back + call count),
 *       3 call table (scan + fill), 4 dispersion fit.  Synchronises. */
int ed_batch_enable_timing(ed_batch* batch, int enable);
int ed_batch_stage_ms(ed_batch* batch, float ms[5]);
/* The same stage times summed over every timed ed_batch_run (n_runs) and fit (n_fits) since timing was enabled: what a
 * pipelined caller reads once at the end instead of synchronising after every batch (the library folds the events of a
 * batch's previous run into the sums when the next run on that batch is issued). */
int ed_batch_stage_ms_total(ed_batch* batch, double ms_total[5], int64_t* n_runs, int64_t* n_fits);

/* =====================================================================================
 * 3. Reference-set optimisation (the "next" row of the path: select.reference.set)
 * ===================================================================================== */

/* One row of the reference's summary.stats data frame (R/optimize_reference_set.R:104-111), in order of
 * decreasing correlation.  Fields the R loop never reaches after its early exit (:130) are NaN. */
typedef struct {
  int32_t ref_index;     /* column of the reference matrix (0-based) */
  int32_t selected;      /* 1 on the row where expected.BF is maximal (:143-144) */
  double correlation;    /* :100 */
  double expected_BF;    /* :135-139, get.power.betabinom (R/tools.R:128-166) */
  double phi;            /* :125 */
  double ratio_sd;       /* :128 */
  double mean_p;         /* :126 */
  double median_depth;   /* :127 */
} ed_refset_row;

/* select.reference.set(test.counts, reference.counts, bin.length, n.bins.reduced), reference
 * R/optimize_reference_set.R:53-148, with formula ~ 1 and phi.bins = 1 (its defaults).
 *   d_test   DEVICE int32 [n_bins]
 *   d_refs   DEVICE int32 [n_bins][n_refs]  (reference-minor; R's matrix is the transpose in memory)
 *   bin_length  HOST double [n_bins] or NULL (= rep(1, n))        n_bins_reduced  0 = use all selected bins
 *   rows     HOST out, n_refs entries sorted by decreasing correlation
 *   n_chosen = which.max(expected.BF): reference.choice is rows[0 .. n_chosen-1].ref_index
 *   n_selected_bins (optional) = "Number of selected bins" (:97)
 * The R loop fits one beta-binomial model per prefix of the sorted references, sequentially; here all
 * prefixes are fitted as one batch.  Synchronous (the result is host data). */
int ed_select_reference_set(const int32_t* d_test, const int32_t* d_refs, int64_t n_bins, int64_t n_refs,
                            const double* bin_length, int64_t n_bins_reduced, ed_refset_row* rows, int32_t* n_chosen,
                            int64_t* n_selected_bins, void* stream);

/* The same for the cumulative references prefix_begin <= i < prefix_end of the correlation-sorted axis only (raw
 * statistics; no early exit, no choice): the prefixes are independent given the order, so N ranks that each hold
 * the count matrix (4 GB at 500 000 x 2048: replicated, not sharded, in 288 GB of HBM) take contiguous shares and
 * exchange nothing but these rows.  Every row gets ref_index and correlation; *n_chosen = 0 (1 with
 * *n_selected_bins = 0 when the coverage is too low, :57-61). */
int ed_select_reference_set_part(const int32_t* d_test, const int32_t* d_refs, int64_t n_bins, int64_t n_refs,
                                 const double* bin_length, int64_t n_bins_reduced, int64_t prefix_begin, int64_t prefix_end,
                                 ed_refset_row* rows, int32_t* n_chosen, int64_t* n_selected_bins, void* stream);
/* The loop's early exit (:130: statistics up to and including the breaking iteration, expected.BF before it) and
 * reference.choice (:143-145) on a complete table of raw rows.  Host code only (no device needed). */
int ed_refset_finalize(ed_refset_row* rows, int64_t n_refs, int32_t* n_chosen);

/* The bins select.reference.set keeps when n.bins.reduced > 0: 0-based positions of  x[seq(1, len, len / n_reduced)]
 * (R/optimize_reference_set.R:86) with R's seq() and subscript-truncation semantics.  positions[cap] receives the first
 * cap of them, *n_positions their number (at most n_reduced + 1).  Host code only (no device needed). */
int ed_refset_thin_positions(int64_t len, int64_t n_reduced, int64_t* positions, int64_t cap, int64_t* n_positions);

/* select.reference.set for EVERY sample of a cohort at once -- each sample in turn as the test, all the others as candidates,
 * as the loop of reference vignette/vignette.Rnw:390-402 does -- and the aggregate reference of every sample.
 *   d_counts   DEVICE int32 [n_bins][n_samples] (sample-minor)       bin_length  HOST double[n_bins] or NULL
 *   max_refs   K: cumulative references formed per test (0 = 32).  The R loop stops at the first i > 2 with mean.p < 0.05
 *              (R/optimize_reference_set.R:130), in practice well before 32; a test whose loop would run on is handed to
 *              ed_select_reference_set (all prefixes), so the result never depends on K -- only a choice LONGER than K is an error
 *   n_chosen   HOST int32 [n_samples]: which.max(expected.BF) per test (:143-144)
 *   choice     HOST int32 [n_samples][K]: the chosen references (columns of the cohort) in order of decreasing correlation, -1 padded
 *   rows       HOST (optional) [n_samples][K]: summary.stats of the first K cumulative references of every test (:104-111)
 *   correlations HOST (optional) double [n_samples][n_samples]: the correlation matrix of :100 (diagonal 1)
 *   d_ref_out  DEVICE (optional) int32 [n_bins][n_samples]: aggregate reference = sum of the chosen columns (vignette.Rnw:398-402),
 *              what ed_batch_run / ed_cohort_submit take as d_ref next to d_test = d_counts
 * The S x S correlations are one binary64 Gram matrix on the matrix cores (v_mfma_f64_16x16x4_f64); the K x n_samples cumulative
 * references are fitted as one batch.  Synchronous.
 *   stream     the HIP stream (hipStream_t) the entry issues its kernels and transfers on; NULL = the null stream, which is ordered against
 *              every blocking stream of the process -- name a stream of your own to let a copy on another stream (the next cohort's counts)
 *              run beside the call.  The entry's ~25 small host <-> device transfers do not use the DMA queues (a pinned block and a copy
 *              kernel), so they do not wait behind such a copy either.  d_counts must be complete on `stream` when the call is made. */
int ed_cohort_select_reference_sets(const int32_t* d_counts, int64_t n_bins, int64_t n_samples, const double* bin_length,
                                    int64_t n_bins_reduced, int32_t max_refs, int32_t* n_chosen, int32_t* choice, ed_refset_row* rows,
                                    double* correlations, int32_t* d_ref_out, int64_t* n_selected_bins, void* stream);

/* The same for the tests test_begin <= t < test_end only, every sample of the cohort still a candidate: what ONE RANK of a sample-sharded
 * cohort runs once it holds all the count columns (reference vignette/vignette.Rnw:390-402 run for its own samples).  Only the rows
 * [test_begin, test_end) of the correlation matrix are formed (a (test_end - test_begin) x n_samples block of the Gram matrix + its
 * diagonal), and only the owned tests' prefixes, fits and aggregate references.  The per-test outputs are indexed by t - test_begin:
 * n_chosen [n_tests], choice [n_tests][K] (columns of the WHOLE cohort), rows [n_tests][K], correlations [n_tests][n_samples],
 * d_ref_out DEVICE int32 [n_bins][n_tests].  The bin selection uses the row sums over all samples, as the whole-cohort call does:
 * the shards' results are exactly the corresponding rows / columns of the whole-cohort call's. */
int ed_cohort_select_reference_sets_range(const int32_t* d_counts, int64_t n_bins, int64_t n_samples, const double* bin_length,
                                          int64_t n_bins_reduced, int32_t max_refs, int64_t test_begin, int64_t test_end, int32_t* n_chosen,
                                          int32_t* choice, ed_refset_row* rows, double* correlations, int32_t* d_ref_out,
                                          int64_t* n_selected_bins, void* stream);

/* The same with the aggregate references written SAMPLE-MAJOR -- d_ref_sm_out DEVICE int32 [n_tests][n_bins], the memory image of R's
 * n_bins x n_tests matrix -- and, optionally, the count matrix transposed alongside, d_counts_sm_out DEVICE int32 [n_samples][n_bins]
 * (NULL = not wanted): what a cohort with options emit_mode = 2, counts_layout = 1 takes as they are (ed_cohort_submit), so the calls that
 * follow the reference sets (vignette/vignette.Rnw:403-431) neither transpose anything nor leave the sample-major fit.  (One pass over the
 * counts while a tile of all candidates and the tests' lists fit LDS -- 1024 samples x 32 candidates with room to spare; other geometries through
 * the [n_bins][n_tests] form and a transposition.) */
int ed_cohort_select_reference_sets_sm(const int32_t* d_counts, int64_t n_bins, int64_t n_samples, const double* bin_length,
                                       int64_t n_bins_reduced, int32_t max_refs, int64_t test_begin, int64_t test_end, int32_t* n_chosen,
                                       int32_t* choice, ed_refset_row* rows, double* correlations, int32_t* d_ref_sm_out,
                                       int32_t* d_counts_sm_out, int64_t* n_selected_bins, void* stream);

/* ed_cohort_select_reference_sets keeps its device scratch (about 1.5 GB at 10 000 selected bins x 1024 samples) between calls;
 * this returns it to the device. */
int ed_release_scratch(void);
/* How the last ed_cohort_select_reference_sets* call of this process formed its cumulative references' statistics (tests, diagnostics): out = {chunks
 * served by the column-major kernel (csrc/edrefcohort.inc: k_rc_column), chunks served by the row-major kernels (more than 65 535 selected bins, or counts
 * beyond the reach of the column kernel's large bins), columns that kept counts beyond their bins as values, the most Newton iterations a column took,
 * columns served by the large geometry (34 816 bins; the others: 10 240)}.  The two forms follow R/optimize_reference_set.R:114-128 alike; they differ in
 * the rounding of the fits. */
int ed_refcohort_last_path(int64_t out[5]);
/* ... on host data in R's layout: counts the n_bins x n_samples integer matrix, column-major; reference_out_colmajor (optional)
 * receives the aggregate reference in the same layout.  The other arguments as above. */
int ed_cohort_select_reference_sets_host(const int32_t* counts_colmajor, int64_t n_bins, int64_t n_samples, const double* bin_length,
                                         int64_t n_bins_reduced, int32_t max_refs, int32_t* n_chosen, int32_t* choice, ed_refset_row* rows,
                                         double* correlations, int32_t* reference_out_colmajor, int64_t* n_selected_bins);

/* get.power.betabinom(size, my.phi, my.p, my.alt.p) (reference R/tools.R:128-166), default mode (theory = FALSE,
 * frequentist = FALSE, limit = FALSE): the expected log10 Bayes factor sum_{x=0}^{size} dbetabinom(x; alt) log10 BF(x),
 * for n parameter sets at once.  HOST arrays; synchronous. */
int ed_get_power_betabinom(int64_t n, const double* size, const double* phi, const double* p, const double* alt_p, double* out);
/* The same with the reference's switches.  mode 1 = `theory = TRUE`, its binomial case (R/tools.R:137-142): the sum of
 * dbinom(x; size, alt_p) * log10 of the binomial likelihood ratio (phi is not used; 0 < p, alt_p < 1).  mode 2 = `limit = TRUE`
 * (:145-153): the reference averages log10[dbeta(X/size; alt) / dbeta(X/size; null)] over 2000 draws X ~ rbetabinom.ab(alt) of R's
 * generator; this entry returns the EXPECTATION that average estimates -- sum_{0 < x < size} dbetabinom.ab(x; alt) * that log ratio --
 * i.e. the reference's value without its Monte-Carlo noise (standard error ~ sd / sqrt(2000)). */
int ed_get_power_betabinom_mode(int64_t n, const double* size, const double* phi, const double* p, const double* alt_p,
                                int theory, double* out);

/* =====================================================================================
 * 4. Cohort pipeline: slabs of a cohort through batch objects in rotation, on the library's own streams
 * ===================================================================================== */

/* The reference's user loops over the samples of a cohort (vignette/vignette.Rnw:390-431): per sample one
 * new('ExomeDepth') -- aod::betabin (R/class_definition.R:118), then .Call get_loglike_matrix (:184-189) -- and one
 * CallCNVs() -- one .Call C_hmm per chromosome (:354-374, R/tools.R:97).  A cohort object does that work for SLABS of
 * samples: it owns its streams and `slabs_in_flight` batch objects used in rotation, and orders the stages of consecutive
 * slabs with events so that the dispersion fit of slab t+1 and the Viterbi tail of slab t execute underneath the VALU-bound
 * emissions of slabs t / t+1.  The schedule is built from stream order alone: it does not depend on the
 * host's timing, on GPU_MAX_HW_QUEUES, or on the streams the process used before.  Results are those of ed_batch_fit +
 * ed_batch_run on each slab, bit for bit.
 *   slab_samples     samples per slab (the batch objects are sized for it; a last, smaller slab is accepted)
 *   slabs_in_flight  1: every slab runs to completion before the next (no overlap); 2 or more: pipelined; 4 with option "lanes" = 2 is the
 *                    fastest form for full-width slabs fitted on the device (a slot holds a slab's likelihood matrix, 24 B per cell).
 * Options (ed_cohort_set_option, before the first submission unless noted):
 *   "split"            fraction of a slab's emission launch after which the NEXT slab's fit is issued (default 0.30; 0 = at once)
 *   "own_queues"       1 (default): every stream of the pipeline gets a hardware queue of its own; 0: ordinary streams
 *   "fit_mode"         0 (default) maximum likelihood; 1 aod::betabin's Nelder-Mead procedure (ed_batch_set_fit_mode)
 *   "phi_bins"         1 (default): one dispersion per sample; 2..8: the depth-binned dispersion model of the reference's `phi.bins`
 *                      argument (R/class_definition.R:120-147) for every slab -- ed_batch_fit_bins + ed_batch_run_bins, with the fit on
 *                      count histograms issued without a host look at its outcome: that is settled when the ticket is first waited
 *                      for (an empty depth level is that ticket's error, "Binning did not happen properly"; data the histogram form
 *                      declines are done again through the per-cell form then -- the slab's count arrays must stay in place until
 *                      the ticket has been waited for).  Parameters cannot be given in this mode; read them with
 *                      ed_cohort_copy_bins_params / ed_cohort_copy_bins.
 *   "bins_pieces"      phi_bins > 1, pipelined: launches the emission kernel of a slab is cut into (default 10): the next slab's fit is
 *                      a chain of kernels whose workgroups need a whole CU each and only get one where a launch ends
 *   "emit_mode"        0 (default) strict, 1 tables, 2 tables sample-major: ed_batch_set_emit_mode for every slab (phi_bins must be 1)
 *   "counts_layout"    0 (default): device counts [n_exons][n_samples]; 1: [n_samples][n_exons] (needs emit_mode 2): ed_cohort_submit takes
 *                      them that way, and host-fed slabs in layout 1 (R's column-major matrix) are uploaded without a transposition
 *   "counts_bits"      32 (default) / 16: ed_batch_set_counts_bits for every slab of ed_cohort_submit -- device-resident uint16 counts (needs
 *                      counts_layout 1 and emit_mode 2).  Host-fed slabs choose their device format themselves, see "host_narrow"
 *   "emit_tails"       1 (default) / 0: ed_batch_set_emit_tails for every slab (emit_mode 2)
 *   "host_narrow"      1 (default) / 0: host-fed slabs of a cohort with emit_mode 2 + counts_layout 1 stay 16 bits wide on the device whenever
 *                      their counts fit -- uint16 host blocks go up as they lie, int32 blocks in pageable memory (R's integer matrices) are
 *                      narrowed by the host threads that stage them: 2 bytes per count on the link, no widening pass.  A slab holding a count
 *                      outside 0 .. 65 535 goes up as int32 (ed_cohort_n_wide_slabs counts them).  Same bits of every result either way
 *   "viterbi_overlap"  0 (default): one emission launch per slab, its chains afterwards; 1: ed_batch_set_viterbi_overlap(1)
 *   "tables_early"     1: a slab's per-sample constants and tables are made right behind its fit, on the fit stream; 0 (default):
 *                      between two emission launches (the same work either way: measured equal)
 *   "lanes"            independent pipelines inside the object: slot s belongs to lane s % lanes, a lane has its own emission and fit streams,
 *                      nothing orders the slabs of different lanes against each other (one lane's emission launch fills the CUs that another's
 *                      table build and chains leave idle: 3.70 / 3.63 ms per 200 000 x 1024 slab with 2 / 3 lanes of two slots against 4.04 with
 *                      one).  1 (default): one pipeline; 2..4: that many (a divisor of slabs_in_flight); 0: slabs_in_flight / 2 when that is 4, 6
 *                      or 8.  Opt-in: it pays for slabs that fill the chip and are fitted on the device, and was measured worse for 64-sample
 *                      slabs with given parameters and for host-fed (link-bound) slabs.  Results do not depend on it.
 *   "timing"           1: stage times are accumulated (ed_cohort_stage_ms_total); may be switched at any time (resets the sums) */
typedef struct ed_cohort ed_cohort;
int ed_cohort_create(ed_cohort** cohort, ed_plan* plan, int64_t slab_samples, int slabs_in_flight);
void ed_cohort_destroy(ed_cohort* cohort);
int ed_cohort_set_option(ed_cohort* cohort, const char* name, double value);

/* Submit one slab whose counts are on the device: d_test / d_ref int32 [n_exons][n_samples] sample-minor, n_samples <=
 * slab_samples.  d_phi / d_expected: DEVICE double[n_samples], or both NULL = fit them (ed_batch_fit).  ready_stream: a stream
 * whose already enqueued work produces the counts (the pipeline waits for it on the device), NULL = they are complete.
 * Asynchronous.  *ticket numbers the slabs from 0.  The results of a ticket stay available until `slabs_in_flight` further
 * slabs have been submitted; its counts must stay valid until then too (the call decoration reads them). */
int ed_cohort_submit(ed_cohort* cohort, const int32_t* d_test, const int32_t* d_ref, int64_t n_samples, const double* d_phi,
                     const double* d_expected, double mixture, void* ready_stream, int64_t* ticket);
/* The batch object holding a ticket's results -- read them with the ed_batch_* accessors (ed_batch_n_calls, ed_batch_copy_*,
 * ed_batch_path, ...), which wait for that slab only -- and the DEVICE arrays of its (phi, expected).  ED_ERR_STATE once the
 * ticket's slot has been reused. */
int ed_cohort_batch(ed_cohort* cohort, int64_t ticket, ed_batch** batch, const double** d_phi, const double** d_expected);
/* that slab's (phi, expected) to HOST arrays [n_samples of the slab] (either may be NULL); waits for that slab only */
int ed_cohort_copy_params(ed_cohort* cohort, int64_t ticket, double* phi_out, double* expected_out);
/* option phi_bins > 1: that slab's phi.estimates [phi_bins][n_samples of the slab], complete.bins [(phi_bins + 1)][n_samples] and
 * fitted(mod) [n_samples] to HOST arrays (any may be NULL) */
int ed_cohort_copy_bins_params(ed_cohort* cohort, int64_t ticket, double* phi_bins_out, double* edges_out, double* expected_out);
int ed_cohort_wait(ed_cohort* cohort, int64_t ticket);   /* host waits for that slab's call table */
int ed_cohort_drain(ed_cohort* cohort);                  /* ... for everything submitted so far */
void* ed_cohort_stream(ed_cohort* cohort);               /* the pipeline's main stream (a hipStream_t) */
/* ed_batch_stage_ms_total summed over the cohort's batch objects; launches of the emission kernel per slab */
int ed_cohort_stage_ms_total(ed_cohort* cohort, double ms_total[5], int64_t* n_runs, int64_t* n_fits);
/* Option "timing": (start_ms, end_ms) of the emission stage of every run timed since the option was set, relative to one reference event of the
 * cohort.  With several lanes the emission launches of consecutive slabs run side by side: the union of these intervals is the chip time during
 * which an emission launch was active (bench.py's roofline.kernel_ms), their mean length a launch's own duration (what a kernel trace's average
 * shows).  out[2 * cap]; *n = pairs available, the first min(*n, cap) are written. */
int ed_cohort_emission_intervals(ed_cohort* cohort, float* out, int64_t cap, int64_t* n);
int ed_cohort_n_emit_launches(ed_cohort* cohort);

/* ed_cohort_submit_host with only the TEST counts in host memory and the references on the device (d_ref: int32, complete when this
 * is called, in the COHORT's device layout: [n_exons][n_samples] with option counts_layout = 0 -- what ed_cohort_select_reference_sets
 * writes -- and [n_samples][n_exons] with counts_layout = 1; the host matrix's own layout is the `layout` argument, as for
 * ed_cohort_submit_host).  In the reference's workflow a sample's reference is the sum of
 * other samples of the same cohort (vignette/vignette.Rnw:390-402): ed_cohort_select_reference_sets makes it on the device from counts
 * uploaded once, so only one matrix per slab crosses the link. */
int ed_cohort_submit_host_test(ed_cohort* cohort, const void* test, const int32_t* d_ref, int64_t n_samples, int layout, int wire,
                               int64_t row_stride, const double* phi, const double* expected, double mixture, int64_t* ticket);
/* ---- slabs from host memory (ingest) ----
 * layout 0: the host matrix is [n_exons][row_stride] sample-minor (the EDCOUNT1 container of exomedepth_amd/io.py); the slab
 *           is its first n_samples columns from the given pointer (row_stride = the cohort's width for a window of columns)
 * layout 1: the host matrix is R's n_exons x n_samples integer matrix, column-major (row_stride ignored); transposed on the
 *           device to the pipeline's sample-minor layout
 * wire 4: int32 elements; wire 2: uint16 elements (counts below 65536: half the PCIe bytes), widened on the device -- except in a cohort with
 *         emit_mode 2 + counts_layout 1, where slabs whose counts fit 16 bits STAY uint16 on the device and pageable int32 blocks are narrowed
 *         on the host while they are staged (option "host_narrow")
 * Pinned host memory (ed_host_alloc) is read by the DMA engine in place; pageable memory goes through the cohort's pinned
 * double buffer (a few host threads copy chunk k+1 while chunk k is on the link).  Uploads run on a copy stream of the
 * cohort and overlap the compute of earlier slabs.  phi / expected: DEVICE arrays or NULL (fit), as ed_cohort_submit.
 * The slot's device buffers are reused: collect the results of ticket - slabs_in_flight first. */
int ed_cohort_submit_host(ed_cohort* cohort, const void* test, const void* ref, int64_t n_samples, int layout, int wire,
                          int64_t row_stride, const double* d_phi, const double* d_expected, double mixture, int64_t* ticket);
int ed_cohort_ingest_stats(ed_cohort* cohort, double* bytes, double* host_seconds);
int ed_cohort_n_wide_slabs(ed_cohort* cohort, int64_t* n);   /* host-fed int32 slabs that held a count outside 0 .. 65 535 and went up 32 bits wide */
int ed_host_alloc(void** hptr, size_t bytes);   /* pinned host memory */
int ed_host_free(void* hptr);

/* CallCNVs for a whole cohort held in host memory -- what the R-level wrapper ed_call_cnvs_batch (shim/edcore_shim.c) calls:
 * n_total samples cut into slabs, uploaded, fitted (phi == NULL) or given (HOST phi[n_total], expected[n_total]), run, and
 * collected while later slabs compute.  Host outputs (each may be NULL): phi_out / expected_out [n_total]; path_out uint8 in
 * the layout of the input (layout 1: [n_total][n_exons], R's n_exons x n_total matrix; layout 0: [n_exons][n_total]).
 * The call table (sample = column of the cohort) and its decoration are kept by the cohort: *n_calls rows, copied out with
 * ed_cohort_copy_calls; ed_cohort_run_status gives the number of samples whose fit did not converge and of GSL error events. */
int ed_cohort_run_host(ed_cohort* cohort, const void* test, const void* ref, int64_t n_total, int layout, int wire,
                       const double* phi, const double* expected, double mixture, double* phi_out, double* expected_out,
                       uint8_t* path_out, int64_t* n_calls);
int ed_cohort_copy_calls(ed_cohort* cohort, ed_call* calls, ed_call_info* info, int64_t cap);
int ed_cohort_run_status(ed_cohort* cohort, int64_t* n_unconverged, int64_t* n_gsl_errors);
/* table-driven emit modes: ed_batch_table_stats summed over the slabs of the last ed_cohort_run_host */
int ed_cohort_table_status(ed_cohort* cohort, int64_t out[4]);
/* option phi_bins > 1, after ed_cohort_run_host (whose phi_out is not written in that mode): phi.estimates [phi_bins][n_total] and
 * complete.bins [(phi_bins + 1)][n_total] of the whole cohort */
int ed_cohort_copy_bins(ed_cohort* cohort, double* phi_bins_out, double* edges_out);

/* ---- one process, several devices --------------------------------------------------------------------------------------
 * ed_cohort_run_host over every device of the node.  The reference's user loops over the samples of a cohort in one R process
 * (vignette/vignette.Rnw:390-431) and the samples are independent (R/class_definition.R:82-191, :311-419 see one test vector each), so
 * nothing is exchanged on the path: the plan is replicated on every device, the cohort's slabs sit in ONE queue, one host thread per
 * device drives an ordinary cohort pipeline over the slabs it takes from it (reading the caller's host matrices in place, writing its
 * windows of the outputs) -- a slower or busier device takes fewer slabs instead of setting the wall time with a fixed share -- and the
 * compact call tables are put together in slab order = column order.  Every slab is fitted and called on its own, so the results are those
 * of ed_cohort_run_host with the same slab_samples on one device, bit for bit, whatever the number of devices and whichever device served
 * a slab.  A device's thread is kept on the CPUs of the device's NUMA node when sysfs names one (option "numa_bind", default 1; never the
 * caller's own thread).
 *   devices / n_devices   HIP device ordinals; NULL / 0 = every visible device once.  A device may be named more than once
 *                         (two pipelines on one GPU: how a single-GPU box exercises the threads and the merge)
 *   the plan arguments as ed_plan_create, slab_samples / slabs_in_flight as ed_cohort_create, options as ed_cohort_set_option
 * ed_multi_run_host / _copy_calls / _run_status / _table_status / _copy_bins: as their ed_cohort_* namesakes, over all devices.
 * ed_multi_device_stats: slabs and columns device i took from the queue in the last run, the wall seconds of its thread and the NUMA node
 * it was kept on (-1: unknown); arrays of ed_multi_n_devices entries, any may be NULL. */
typedef struct ed_multi ed_multi;
int ed_multi_create(ed_multi** multi, const int* devices, int n_devices, int64_t n_exons, int32_t n_chrom, const int32_t* chrom_off,
                    const int32_t* start, const int32_t* end, double transition_probability, double expected_cnv_length,
                    int64_t slab_samples, int slabs_in_flight);
void ed_multi_destroy(ed_multi* multi);
int ed_multi_n_devices(const ed_multi* multi);
int ed_multi_set_option(ed_multi* multi, const char* name, double value);
int ed_multi_run_host(ed_multi* multi, const void* test, const void* ref, int64_t n_total, int layout, int wire, const double* phi,
                      const double* expected, double mixture, double* phi_out, double* expected_out, uint8_t* path_out, int64_t* n_calls);
int ed_multi_copy_calls(ed_multi* multi, ed_call* calls, ed_call_info* info, int64_t cap);
int ed_multi_run_status(ed_multi* multi, int64_t* n_unconverged, int64_t* n_gsl_errors);
int ed_multi_table_status(ed_multi* multi, int64_t out[4]);
int ed_multi_copy_bins(ed_multi* multi, double* phi_bins_out, double* edges_out);
int ed_multi_device_stats(ed_multi* multi, int* devices, int64_t* n_slabs, int64_t* n_columns, double* seconds, int* numa_nodes);

/* The model fit of new('ExomeDepth') alone, for every column of a host-resident cohort: what stands where the reference calls
 * aod::betabin(cbind(test, reference) ~ 1, random = ~ 1) and fitted(mod) (R/class_definition.R:118-119, :168).  layout / wire as
 * ed_cohort_submit_host (a dense matrix: layout 0 [n_exons][n_samples], layout 1 R's column-major n_exons x n_samples).
 * fit_mode as ed_batch_set_fit_mode.  converged_out (optional) int32[n_samples]: 1 = converged. */
int ed_fit_betabin_host(const void* test, const void* ref, int64_t n_exons, int64_t n_samples, int layout, int wire, int fit_mode,
                        double* phi_out, double* expected_out, int32_t* converged_out);
/* ed_select_reference_set on host data in R's layout: test.counts integer[n_bins], reference.counts the n_bins x n_refs integer
 * matrix column-major (uploaded and transposed on the device). */
int ed_select_reference_set_host(const int32_t* test, const int32_t* refs_colmajor, int64_t n_bins, int64_t n_refs,
                                 const double* bin_length, int64_t n_bins_reduced, ed_refset_row* rows, int32_t* n_chosen,
                                 int64_t* n_selected_bins);

/* ---- utilities ---- */
/* device memory through the library, for callers without a HIP binding (tests, R shim) */
int ed_malloc(void** dptr, size_t bytes);
int ed_free(void* dptr);
int ed_memcpy_h2d(void* dst, const void* src, size_t bytes);
int ed_memcpy_d2h(void* dst, const void* src, size_t bytes);
int ed_synchronize(void* stream);

/* Element-wise evaluation of the device special functions on host arrays (used by the parity tests
 * to compare the device arithmetic with the checker bit for bit).
 * which: 0 lnbeta(x,y)  1 portable log(x)  2 portable exp(x)  3 sqrt(x)  4 x/y  5 portable sin(x) on [0,pi]
 *        6 digamma(x)  7 trigamma(x)  8 in-range exact division x/y  9 exp for |x| < ln2/2  10 log, fast path
 *        11 error sites of lnbeta(x,y) (see ed_get_loglike_matrix_messages)  12 log|Gamma(x)| for any x
 *        13 sign of Gamma(x) + 8 * its error site  14 portable sin(x), |x| < 2^52
 *        15 digamma(x), 16 trigamma(x) by the fit's short series (x >= 32)  17 / 18 d/da of one cell's log-likelihood term at (a, b, a + b) =
 *        (x, 4x, 5x), (y, n) = (y, 9y) through the batched fit's cell routine, short series where its arguments allow / long series */
int ed_eval_sf(int which, int64_t n, const double* x, const double* y, double* out);

#ifdef __cplusplus
}
#endif
#endif /* EXOMEDEPTH_AMD_H */
