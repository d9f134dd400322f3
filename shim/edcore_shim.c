/* edcore_shim.c -- the R side of the drop-in: the reference's two .Call entries on top of libedcore.so.
 *
 * Replaces, in the R package's src/, the files CNV_estimate.cpp, hmm.cpp, ExomeDepth_init.c and the vendored GSL
 * sources; R/class_definition.R:184-189 and R/tools.R:97 call it unchanged:
 *     .Call("get_loglike_matrix", phi, expected, total, observed, mixture)        reference src/CNV_estimate.cpp:16, :52-85
 *     .Call("C_hmm", nstates, nobs, transitions, probabilities, positions, L)     reference src/hmm.cpp:13, :18-167
 * registered exactly as reference src/ExomeDepth_init.c:14-24 registers them ({"C_hmm", 6}, {"get_loglike_matrix", 5},
 * dynamic symbols off).  Inputs stay R-owned and read-only; outputs are R allocations; the library copies in, launches,
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

static const R_CallMethodDef CallEntries[] = {                                /* src/ExomeDepth_init.c:14-18 */
  {"C_hmm",              (DL_FUNC) &C_hmm,              6},
  {"get_loglike_matrix", (DL_FUNC) &get_loglike_matrix, 5},
  {NULL, NULL, 0}
};

void R_init_ExomeDepth(DllInfo *dll)                                          /* src/ExomeDepth_init.c:20-24 */
{
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
