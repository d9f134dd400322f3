/* edcore_shim.c -- the R side of the drop-in: the reference's two .Call entries on top of libedcore.so.
 *
 * Replaces, in the R package's src/, the files CNV_estimate.cpp, hmm.cpp, ExomeDepth_init.c and the vendored GSL
 * sources; R/class_definition.R:184-189 and R/tools.R:97 call it unchanged:
 *     .Call("get_loglike_matrix", phi, expected, total, observed, mixture)        reference src/CNV_estimate.cpp:16, :52-85
 *     .Call("C_hmm", nstates, nobs, transitions, probabilities, positions, L)     reference src/hmm.cpp:13, :18-167
 * registered exactly as reference src/ExomeDepth_init.c:14-24 registers them ({"C_hmm", 6}, {"get_loglike_matrix", 5},
 * dynamic symbols off) -- plus, next to them, the cohort-level entries at the end of this file (ed_call_cnvs_batch,
 * ed_fit_betabin_batch, ed_select_reference_set, ed_cohort_reference_sets: whole count matrices in, plain lists out; INTEGRATION.md has the R functions).  Inputs stay R-owned and read-only; outputs are R allocations; the library copies in, launches,
 * synchronises and copies out inside the call (R's API is single-threaded).
 *
 * What it prints is what the reference prints: the mixture notice (src/CNV_estimate.cpp:61) and, for shape parameters
 * outside the model's domain, the lines of the GSL error handler (src/error.c:45-48), which the library reproduces as
 * text (ed_get_loglike_matrix_messages) since the events happen on the device.
 * Deviation: nstates != 3 raises an R error; the reference prints "ERROR: The code must assume 3 states" and returns a
 * C NULL (src/hmm.cpp:37-40), which crashes R.
 *
 * Build: shim/Makevars (R CMD SHLIB / R CMD INSTALL).  This image has no R: tests/test_shim.py compiles this file
 * against declarations-only headers (tests/rapi/) and drives it through a miniature runtime kept under tests/.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "exomedepth_amd.h"

SEXP get_loglike_matrix(SEXP phi, SEXP expected, SEXP total, SEXP observed, SEXP mixture)
{
  const R_xlen_t n = XLENGTH(total);                       /* src/CNV_estimate.cpp:57: n comes from `total` */
  const double mix = REAL(mixture)[0];
  if (mix != 1) Rprintf("As a warning (this could be normal), the mixture coefficient is %f\n", mix);   /* :61 */
  SEXP ans = PROTECT(allocMatrix(REALSXP, (int)n, 3));     /* :69 */
  int64_t nerr = 0;
  const int rc = ed_get_loglike_matrix(REAL(phi), REAL(expected), INTEGER(total), INTEGER(observed), (int64_t)n, mix,
                                       REAL(ans), &nerr);
  if (rc != ED_OK) {
    UNPROTECT(1);
    Rf_error("exomedepth_amd: %s", ed_last_error());
  }
  if (nerr) {
    /* the reference's gsl_error() lines, in its order (src/error.c:45-48) */
    size_t need = 0;
    if (ed_get_loglike_matrix_messages(REAL(phi), REAL(expected), INTEGER(total), INTEGER(observed), (int64_t)n, mix,
                                       NULL, 0, &need) == ED_OK && need > 0) {
      char *buf = (char *) R_alloc(need + 1, 1);
      if (ed_get_loglike_matrix_messages(REAL(phi), REAL(expected), INTEGER(total), INTEGER(observed), (int64_t)n, mix,
                                         buf, need + 1, &need) == ED_OK)
        Rprintf("%s", buf);
    }
  }
  UNPROTECT(1);
  return ans;
}

SEXP C_hmm(SEXP nstates, SEXP nobs, SEXP transitions, SEXP probabilities, SEXP positions, SEXP expectedLength)
{
  const int ns = INTEGER(nstates)[0], no = INTEGER(nobs)[0];
  if (ns != 3) Rf_error("ERROR: The code must assume 3 states");              /* src/hmm.cpp:37-40 */
  SEXP path = PROTECT(allocVector(REALSXP, no));                              /* :134 */
  const int64_t cap = no > 0 ? no : 1;
  double *tmp = (double *) R_alloc((size_t)cap * 4, sizeof(double));
  int64_t ncalls = 0;
  const int rc = ed_hmm(ns, no, REAL(transitions), REAL(probabilities), INTEGER(positions), REAL(expectedLength)[0],
                        REAL(path), tmp, cap, &ncalls);
  if (rc != ED_OK) {
    UNPROTECT(1);
    Rf_error("exomedepth_amd: %s", ed_last_error());
  }
  SEXP calls = PROTECT(allocMatrix(REALSXP, (int)ncalls, 4));                 /* :135, column-major ncalls x 4 */
  for (int j = 0; j < 4; j++)
    for (int64_t i = 0; i < ncalls; i++) REAL(calls)[j * ncalls + i] = tmp[j * cap + i];
  SEXP out = PROTECT(allocVector(VECSXP, 2));                                 /* :133 */
  SET_VECTOR_ELT(out, 0, path);
  SET_VECTOR_ELT(out, 1, calls);
  UNPROTECT(3);
  return out;
}

/* =====================================================================================================================
 * Cohort-level entries (not in the reference): the per-sample granularity of the two entries above moves ~200 k cells per
 * call and cannot fill a GPU.  These take the WHOLE count matrices, as R holds them (integer, n_exons x n_samples,
 * column-major), and return plain lists that the R functions of INTEGRATION.md turn into the reference's objects.  The C
 * functions carry the prefix edr_ (the C-ABI of libedcore.so owns ed_*); the .Call names are the ones registered below.
 * ===================================================================================================================== */

static void set_names(SEXP list, const char *const *names, int n)
{
  SEXP nm = PROTECT(allocVector(STRSXP, n));
  for (int i = 0; i < n; i++) SET_STRING_ELT(nm, i, mkChar(names[i]));
  setAttrib(list, R_NamesSymbol, nm);
  UNPROTECT(1);
}

/* device objects of one .Call, owned by an external pointer (released by the call itself on every return path, by its finalizer if
 * an R error unwound the call) */
typedef struct { ed_multi *multi; } edr_guard;
static void edr_guard_release(SEXP p)
{
  edr_guard *g = (edr_guard *) R_ExternalPtrAddr(p);
  if (!g) return;
  if (g->multi) ed_multi_destroy(g->multi);
  free(g);
  R_ClearExternalPtr(p);
}

/* CallCNVs for a cohort: per sample what new('ExomeDepth') (R/class_definition.R:82-191: aod::betabin, get_loglike_matrix) and
 * CallCNVs() (:311-419: C_hmm per chromosome, call decoration :379-405) produce.
 *   test, reference   integer matrices n_exons x n_samples; exons ordered as CallCNVs orders them (:323-336)
 *   chrom_off         integer[n_chrom + 1], 0-based first exon of every chromosome (+ n_exons)
 *   start, end        integer[n_exons]
 *   tprob, ecl        transition.probability, expected.CNV.length (:261)
 *   phi, expected     double[n_samples] or NULL (fitted on the device: fit_mode 0 = maximum likelihood, 1 = aod-nm)
 *   prop_tumor        the mixture (:86, :189)            slab     samples per slab of the pipeline (integer)
 *   want_path         integer 0/1: also return the Viterbi state of every exon (raw n_exons x n_samples)
 *   phi_bins          the reference's phi.bins (R/class_definition.R:86, :120-147): 1 = one dispersion per sample; 2..8 = one per depth
 *                     level of the reference counts, phi.linear interpolated per exon (phi / expected cannot be given then)
 *   devices           integer vector of HIP device ordinals, or NULL = every visible device: the samples are independent
 *                     (vignette/vignette.Rnw:390-431 loops over them), so the cohort's slabs are dealt to the devices from one queue,
 *                     one host thread per device (include/exomedepth_amd.h: ed_multi_*); R objects are touched by the calling thread
 *                     only -- the worker threads see plain C arrays.  The result does not depend on the number of devices
 * Value: list(sample, start.p, end.p, type, nexons, BF, reads.expected, reads.observed, reads.ratio  -- one element per call,
 *             ordered by (sample, chromosome, position); sample / start.p / end.p 1-based, type 1 = deletion 2 = duplication --
 *             phi, expected (double[n_samples]), path (raw matrix or NULL), n.unconverged, n.gsl.errors,
 *             phi.bins (phi_bins x n_samples: phi.estimates per level; NULL for phi_bins = 1; `phi` is NA then),
 *             complete.bins ((phi_bins + 1) x n_samples: the level edges, :125-126; NULL for phi_bins = 1)) */
SEXP edr_call_cnvs_batch(SEXP test, SEXP reference, SEXP chrom_off, SEXP start, SEXP end, SEXP tprob, SEXP ecl, SEXP phi,
                        SEXP expected, SEXP prop_tumor, SEXP slab, SEXP want_path, SEXP fit_mode, SEXP phi_bins, SEXP emit_mode, SEXP devices)
{
  const int B = INTEGER(phi_bins)[0];
  const int em = INTEGER(emit_mode)[0];      /* 0 strict, 1 tables (tiles), 2 tables sample-major (include/exomedepth_amd.h: ed_batch_set_emit_mode) */
  if (em < 0 || em > 2) Rf_error("emit.mode must be 0 (strict), 1 or 2 (table-driven emissions)");
  const int E = nrows(test), S = ncols(test);
  if (nrows(reference) != E || ncols(reference) != S) Rf_error("test and reference must be integer matrices of the same shape");
  if (XLENGTH(start) != E || XLENGTH(end) != E) Rf_error("start and end must have one element per exon");
  const int C = (int)XLENGTH(chrom_off) - 1;
  const int given = (phi != R_NilValue);
  if (given && (expected == R_NilValue || XLENGTH(phi) != S || XLENGTH(expected) != S))
    Rf_error("phi and expected must both be given, one value per sample");
  if (B < 1 || B > 8) Rf_error("phi.bins must be in 1..8");
  if (B > 1 && given) Rf_error("with phi.bins > 1 the dispersions are fitted per depth level: phi and expected cannot be given");
  const double mix = REAL(prop_tumor)[0];
  if (mix != 1) Rprintf("As a warning (this could be normal), the mixture coefficient is %f\n", mix);   /* src/CNV_estimate.cpp:61 */
  /* the plan and the cohort live in an external pointer with a finalizer from before the first R allocation on: an R error raised
   * inside an allocation below (a longjmp out of this function) leaves them to the garbage collector instead of leaking device memory */
  int nprot = 0;
  SEXP guard = PROTECT(R_MakeExternalPtr(NULL, R_NilValue, R_NilValue)); nprot++;
  edr_guard *gd = (edr_guard *) calloc(1, sizeof(edr_guard));
  if (!gd) Rf_error("exomedepth_amd: out of memory");
  R_SetExternalPtrAddr(guard, gd);
  R_RegisterCFinalizerEx(guard, edr_guard_release, TRUE);
  const int n_dev = devices == R_NilValue ? 0 : (int)XLENGTH(devices);
  /* The slab width is the caller's, whatever the number of devices: the devices take whole slabs from one queue, every slab is fitted and called on its
   * own, and so the result does not depend on how many GPUs the node happens to show (the width used to shrink to ceil(S / devices), and with
   * it the slab composition the fit's histogram geometry is chosen from).  A cohort of fewer slabs than devices leaves devices idle. */
  int sl = INTEGER(slab)[0];
  if (sl <= 0 || sl > S) sl = S;
  int rc = ed_multi_create(&gd->multi, n_dev > 0 ? INTEGER(devices) : NULL, n_dev, E, C, INTEGER(chrom_off), INTEGER(start), INTEGER(end),
                           REAL(tprob)[0], REAL(ecl)[0], sl, 2);
  if (rc != ED_OK) {
    edr_guard_release(guard);
    Rf_error("exomedepth_amd: %s", ed_last_error());
  }
  ed_multi *co = gd->multi;
  if (rc == ED_OK) rc = ed_multi_set_option(co, "fit_mode", (double)INTEGER(fit_mode)[0]);
  if (rc == ED_OK && B > 1) rc = ed_multi_set_option(co, "phi_bins", (double)B);
  if (rc == ED_OK && B == 1 && em > 0) rc = ed_multi_set_option(co, "emit_mode", (double)em);
  /* R's column-major exons x samples matrix IS the sample-major layout emit mode 2 works in: uploaded as it lies, no transposition */
  if (rc == ED_OK && B == 1 && em == 2) rc = ed_multi_set_option(co, "counts_layout", 1.0);
  SEXP out = R_NilValue;
  int64_t n = 0;
  if (rc == ED_OK) {
    SEXP rphi = PROTECT(allocVector(REALSXP, S)); nprot++;
    SEXP rexp = PROTECT(allocVector(REALSXP, S)); nprot++;
    SEXP rpath = R_NilValue;
    if (INTEGER(want_path)[0]) { rpath = PROTECT(allocMatrix(RAWSXP, E, S)); nprot++; }
    rc = ed_multi_run_host(co, INTEGER(test), INTEGER(reference), S, 1 /* R's column-major */, 4, given ? REAL(phi) : NULL,
                            given ? REAL(expected) : NULL, mix, REAL(rphi), REAL(rexp), rpath != R_NilValue ? RAW(rpath) : NULL, &n);
    if (rc == ED_OK) {
      ed_call *calls = (ed_call *) R_alloc((size_t)(n > 0 ? n : 1), sizeof(ed_call));
      ed_call_info *info = (ed_call_info *) R_alloc((size_t)(n > 0 ? n : 1), sizeof(ed_call_info));
      rc = ed_multi_copy_calls(co, calls, info, n);
      int64_t nu = 0, ne = 0;
      if (rc == ED_OK) rc = ed_multi_run_status(co, &nu, &ne);
      if (rc == ED_OK) {
        static const char *const names[] = {"sample", "start.p", "end.p", "type", "nexons", "BF", "reads.expected", "reads.observed",
                                            "reads.ratio", "phi", "expected", "path", "n.unconverged", "n.gsl.errors", "phi.bins",
                                            "complete.bins"};
        out = PROTECT(allocVector(VECSXP, 16)); nprot++;
        SEXP col[9];
        for (int j = 0; j < 9; j++) {
          col[j] = allocVector((j == 5 || j == 7 || j == 8) ? REALSXP : INTSXP, (R_xlen_t)n);
          SET_VECTOR_ELT(out, j, col[j]);                                   /* (protected through `out`) */
        }
        for (int64_t i = 0; i < n; i++) {
          INTEGER(col[0])[i] = calls[i].sample + 1;
          INTEGER(col[1])[i] = calls[i].start_exon + 1;                    /* start.p after the dummy-exon and shift corrections */
          INTEGER(col[2])[i] = calls[i].end_exon + 1;                      /* (R/class_definition.R:371-372, :409-410) */
          INTEGER(col[3])[i] = calls[i].type;
          INTEGER(col[4])[i] = calls[i].nexons;
          REAL(col[5])[i] = info[i].BF;                                    /* :404 */
          INTEGER(col[6])[i] = (info[i].reads_expected > 2147483647LL || info[i].reads_expected < -2147483647LL)
                                   ? NA_INTEGER : (int)info[i].reads_expected;   /* as.integer(): NA beyond the integer range (:402) */
          REAL(col[7])[i] = (double)info[i].reads_observed;                /* :397 */
          REAL(col[8])[i] = info[i].reads_ratio;                           /* :403 */
        }
        SET_VECTOR_ELT(out, 9, rphi);
        SET_VECTOR_ELT(out, 10, rexp);
        SET_VECTOR_ELT(out, 11, rpath);
        SEXP rnu = allocVector(INTSXP, 1); SET_VECTOR_ELT(out, 12, rnu); INTEGER(rnu)[0] = (int)nu;
        SEXP rne = allocVector(INTSXP, 1); SET_VECTOR_ELT(out, 13, rne); INTEGER(rne)[0] = (int)ne;
        SET_VECTOR_ELT(out, 14, R_NilValue);
        SET_VECTOR_ELT(out, 15, R_NilValue);
        if (B > 1) {
          /* [level][sample] row-major on the library's side = the n_samples x levels matrix column-major: transposed here */
          double *pb = (double *) R_alloc((size_t)B * S, sizeof(double)), *eb = (double *) R_alloc((size_t)(B + 1) * S, sizeof(double));
          rc = ed_multi_copy_bins(co, pb, eb);
          if (rc == ED_OK) {
            SEXP rpb = allocMatrix(REALSXP, B, S); SET_VECTOR_ELT(out, 14, rpb);
            SEXP reb = allocMatrix(REALSXP, B + 1, S); SET_VECTOR_ELT(out, 15, reb);
            for (int s = 0; s < S; s++) {
              for (int g = 0; g < B; g++) REAL(rpb)[(size_t)s * B + g] = pb[(size_t)g * S + s];
              for (int g = 0; g <= B; g++) REAL(reb)[(size_t)s * (B + 1) + g] = eb[(size_t)g * S + s];
              REAL(rphi)[s] = NA_REAL;
            }
          }
        }
        set_names(out, names, 16);
      }
    }
  }
  edr_guard_release(guard);
  UNPROTECT(nprot);
  if (rc != ED_OK) Rf_error("exomedepth_amd: %s", ed_last_error());
  return out;
}

/* The model fit of new('ExomeDepth') alone, for every sample: stands where R/class_definition.R:118 calls
 * aod::betabin(cbind(test, reference) ~ 1, random = ~ 1) and :168 calls fitted(mod).
 * Value: list(phi, expected, converged) with one element per sample. */
SEXP edr_fit_betabin_batch(SEXP test, SEXP reference, SEXP fit_mode)
{
  const int E = nrows(test), S = ncols(test);
  if (nrows(reference) != E || ncols(reference) != S) Rf_error("test and reference must be integer matrices of the same shape");
  SEXP out = PROTECT(allocVector(VECSXP, 3));
  SEXP rphi = allocVector(REALSXP, S); SET_VECTOR_ELT(out, 0, rphi);
  SEXP rexp = allocVector(REALSXP, S); SET_VECTOR_ELT(out, 1, rexp);
  SEXP rcv = allocVector(INTSXP, S); SET_VECTOR_ELT(out, 2, rcv);
  const int rc = ed_fit_betabin_host(INTEGER(test), INTEGER(reference), E, S, 1, 4, INTEGER(fit_mode)[0], REAL(rphi), REAL(rexp),
                                     INTEGER(rcv));
  if (rc != ED_OK) {
    UNPROTECT(1);
    Rf_error("exomedepth_amd: %s", ed_last_error());
  }
  static const char *const names[] = {"phi", "expected", "converged"};
  set_names(out, names, 3);
  UNPROTECT(1);
  return out;
}

/* select.reference.set(test.counts, reference.counts, bin.length, n.bins.reduced) (R/optimize_reference_set.R:53-148), formula ~ 1,
 * phi.bins = 1.  reference_counts: integer matrix n_bins x n_refs; bin_length: double[n_bins] or NULL.
 * Value: list(reference.choice = 1-based columns of reference.counts in order of decreasing correlation (:143-145),
 *             ref.samples (1-based column), correlations, expected.BF, phi, RatioSd, mean.p, median.depth, selected (:104-111),
 *             n.bins) -- the summary.stats columns one element per reference, in order of decreasing correlation. */
SEXP edr_select_reference_set(SEXP test_counts, SEXP reference_counts, SEXP bin_length, SEXP n_bins_reduced)
{
  const int E = nrows(reference_counts), R = ncols(reference_counts);
  if (XLENGTH(test_counts) != E)
    Rf_error("The number of rows of the reference matrix must match the length of the test count data\n");   /* :64 */
  if (bin_length != R_NilValue && XLENGTH(bin_length) != E) Rf_error("bin.length must have one element per bin");
  ed_refset_row *rows = (ed_refset_row *) R_alloc((size_t)R, sizeof(ed_refset_row));
  int32_t n_chosen = 0;
  int64_t n_sel = 0;
  const int rc = ed_select_reference_set_host(INTEGER(test_counts), INTEGER(reference_counts), E, R,
                                              bin_length != R_NilValue ? REAL(bin_length) : NULL, (int64_t)INTEGER(n_bins_reduced)[0],
                                              rows, &n_chosen, &n_sel);
  if (rc != ED_OK) Rf_error("exomedepth_amd: %s", ed_last_error());
  static const char *const names[] = {"reference.choice", "ref.samples", "correlations", "expected.BF", "phi", "RatioSd", "mean.p",
                                      "median.depth", "selected", "n.bins"};
  SEXP out = PROTECT(allocVector(VECSXP, 10));
  SEXP choice = allocVector(INTSXP, n_chosen); SET_VECTOR_ELT(out, 0, choice);
  for (int i = 0; i < n_chosen; i++) INTEGER(choice)[i] = rows[i].ref_index + 1;
  SEXP idx = allocVector(INTSXP, R); SET_VECTOR_ELT(out, 1, idx);
  SEXP col[6];
  for (int j = 0; j < 6; j++) { col[j] = allocVector(REALSXP, R); SET_VECTOR_ELT(out, 2 + j, col[j]); }
  SEXP sel = allocVector(INTSXP, R); SET_VECTOR_ELT(out, 8, sel);
  for (int i = 0; i < R; i++) {
    INTEGER(idx)[i] = rows[i].ref_index + 1;
    REAL(col[0])[i] = rows[i].correlation;
    REAL(col[1])[i] = rows[i].expected_BF;
    REAL(col[2])[i] = rows[i].phi;
    REAL(col[3])[i] = rows[i].ratio_sd;
    REAL(col[4])[i] = rows[i].mean_p;
    REAL(col[5])[i] = rows[i].median_depth;
    INTEGER(sel)[i] = rows[i].selected;
  }
  SEXP nb = allocVector(REALSXP, 1); SET_VECTOR_ELT(out, 9, nb); REAL(nb)[0] = (double)n_sel;
  set_names(out, names, 10);
  UNPROTECT(1);
  return out;
}

/* select.reference.set for every sample of a cohort against all the others, and the aggregate reference of every sample: the loop
 * of vignette/vignette.Rnw:390-402 in one call.  counts: integer matrix n_bins x n_samples; bin_length double[n_bins] or NULL;
 * n_bins_reduced, max_refs integers.
 * Value: list(n.chosen integer[n_samples], choice integer matrix max_refs x n_samples (column t: the chosen references of sample t,
 *             1-based columns of `counts`, in order of decreasing correlation, NA padded), reference integer matrix n_bins x n_samples
 *             (column t = rowSums of the chosen columns: what new('ExomeDepth', test = counts[, t], reference = reference[, t]) takes),
 *             correlations double matrix n_samples x n_samples, n.bins) */
SEXP edr_cohort_reference_sets(SEXP counts, SEXP bin_length, SEXP n_bins_reduced, SEXP max_refs)
{
  const int E = nrows(counts), S = ncols(counts);
  if (S < 2) Rf_error("The reference sequence count data must be provided as a matrix");   /* R/optimize_reference_set.R:63 */
  if (bin_length != R_NilValue && XLENGTH(bin_length) != E) Rf_error("bin.length must have one element per bin");
  int K = INTEGER(max_refs)[0];
  if (K <= 0) K = 32;
  if (K > S - 1) K = S - 1;
  SEXP out = PROTECT(allocVector(VECSXP, 5));
  SEXP nch = allocVector(INTSXP, S); SET_VECTOR_ELT(out, 0, nch);
  SEXP cho = allocMatrix(INTSXP, K, S); SET_VECTOR_ELT(out, 1, cho);
  SEXP ref = allocMatrix(INTSXP, E, S); SET_VECTOR_ELT(out, 2, ref);
  SEXP cor = allocMatrix(REALSXP, S, S); SET_VECTOR_ELT(out, 3, cor);
  SEXP nb = allocVector(REALSXP, 1); SET_VECTOR_ELT(out, 4, nb);
  int64_t nsel = 0;
  /* choice comes back [n_samples][K] row-major = the K x n_samples matrix column-major; correlations are symmetric */
  int rc;
  for (;;) {
    rc = ed_cohort_select_reference_sets_host(INTEGER(counts), E, S, bin_length != R_NilValue ? REAL(bin_length) : NULL,
                                              (int64_t)INTEGER(n_bins_reduced)[0], K, INTEGER(nch), INTEGER(cho), NULL, REAL(cor),
                                              INTEGER(ref), &nsel);
    /* a choice longer than max.refs: the library asks for a larger one (its outputs have max.refs columns; the result does not depend
       on it otherwise) -- done here, so that the R caller never sees a limit the reference does not have */
    if (rc == ED_OK || K >= S - 1 || strstr(ed_last_error(), "larger max_refs") == NULL) break;
    K = (2 * K < S - 1) ? 2 * K : S - 1;
    cho = allocMatrix(INTSXP, K, S); SET_VECTOR_ELT(out, 1, cho);
  }
  if (rc != ED_OK) {
    UNPROTECT(1);
    Rf_error("exomedepth_amd: %s", ed_last_error());
  }
  for (R_xlen_t i = 0; i < (R_xlen_t)K * S; i++) INTEGER(cho)[i] = INTEGER(cho)[i] < 0 ? NA_INTEGER : INTEGER(cho)[i] + 1;
  REAL(nb)[0] = (double)nsel;
  static const char *const names[] = {"n.chosen", "choice", "reference", "correlations", "n.bins"};
  set_names(out, names, 5);
  UNPROTECT(1);
  return out;
}

static const R_CallMethodDef CallEntries[] = {                                /* src/ExomeDepth_init.c:14-18 */
  {"C_hmm",              (DL_FUNC) &C_hmm,              6},
  {"get_loglike_matrix", (DL_FUNC) &get_loglike_matrix, 5},
  /* cohort-level entries of this library (not in the reference) */
  {"ed_call_cnvs_batch",      (DL_FUNC) &edr_call_cnvs_batch,      16},
  {"ed_fit_betabin_batch",    (DL_FUNC) &edr_fit_betabin_batch,    3},
  {"ed_select_reference_set", (DL_FUNC) &edr_select_reference_set, 4},
  {"ed_cohort_reference_sets", (DL_FUNC) &edr_cohort_reference_sets, 4},
  {NULL, NULL, 0}
};

void R_init_ExomeDepth(DllInfo *dll)                                          /* src/ExomeDepth_init.c:20-24 */
{
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
