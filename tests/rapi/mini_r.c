/* TEST INFRASTRUCTURE -- a miniature runtime behind the declarations of Rinternals.h / R_ext/Rdynload.h, just enough
 * to load shim/edcore_shim.c without R and drive its .Call entries from tests/test_shim.py:
 * vectors are malloc'ed blocks, Rprintf appends to a capture buffer, Rf_error longjmps back to minir_call5/6, and
 * R_registerRoutines / R_useDynamicSymbols record what the shim registers.  Nothing here is R's implementation. */
#include <setjmp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Rinternals.h"
#include "R_ext/Rdynload.h"

struct SEXPREC { SEXPTYPE type; R_xlen_t n; int nrow, ncol; void *data; SEXP names; };
static struct SEXPREC nil_rec = {0, 0, 0, 0, NULL, NULL};
static struct SEXPREC names_sym = {1, 0, 0, 0, NULL, NULL};
SEXP R_NilValue = &nil_rec;
SEXP R_NamesSymbol = &names_sym;
int R_NaInt = (-2147483647 - 1);
double R_NaReal;   /* R's NA_real_ is the NaN with payload 1954 (arithmetic.c): set when the library is loaded */
static void __attribute__((constructor)) minir_set_na(void)
{
  union { double d; unsigned long long u; } v;
  v.u = 0x7FF00000000007A2ULL;
  R_NaReal = v.d;
}

static char g_out[1 << 20];
static size_t g_out_len = 0;
static char g_err[4096];
static int g_protect = 0, g_protect_max = 0;
static jmp_buf g_jmp;
static int g_jmp_armed = 0;
static char g_reg_names[8][64];
static int g_reg_nargs[8], g_reg_n = -1, g_dyn = -1;
static DL_FUNC g_reg_fun[8];

double *REAL(SEXP x) { return (double *)x->data; }
int *INTEGER(SEXP x) { return (int *)x->data; }
R_xlen_t XLENGTH(SEXP x) { return x->n; }
int LENGTH(SEXP x) { return (int)x->n; }
static int g_alloc_countdown = -1;   /* minir_fail_alloc_after(k): the k-th allocation from now raises an R error (allocation failure) */
static SEXP g_fin_obj[64];
static R_CFinalizer_t g_fin_fun[64];
static int g_fin_n = 0;
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t length)
{
  if (g_alloc_countdown > 0 && --g_alloc_countdown == 0) { g_alloc_countdown = -1; Rf_error("cannot allocate vector of size %ld", (long)length); }
  SEXP s = (SEXP)calloc(1, sizeof *s);
  const size_t el = type == REALSXP ? 8 : (type == INTSXP ? 4 : ((type == RAWSXP || type == CHARSXP) ? 1 : sizeof(SEXP)));
  s->type = type; s->n = length; s->nrow = (int)length; s->ncol = 1;
  s->data = calloc((size_t)(length > 0 ? length : 1), el);
  return s;
}
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol)
{
  SEXP s = Rf_allocVector(type, (R_xlen_t)nrow * ncol);
  s->nrow = nrow; s->ncol = ncol;
  return s;
}
Rbyte *RAW(SEXP x) { return (Rbyte *)x->data; }
int Rf_nrows(SEXP x) { return x->nrow; }
int Rf_ncols(SEXP x) { return x->ncol; }
SEXP Rf_mkChar(const char *str)
{
  SEXP s = Rf_allocVector(CHARSXP, (R_xlen_t)strlen(str) + 1);
  memcpy(s->data, str, strlen(str) + 1);
  return s;
}
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)x->data)[i]; }
const char *R_CHAR(SEXP x) { return (const char *)x->data; }
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val) { if (name == R_NamesSymbol) x->names = val; return val; }
SEXP Rf_getAttrib(SEXP x, SEXP name) { return (name == R_NamesSymbol && x->names) ? x->names : R_NilValue; }
SEXP Rf_protect(SEXP x) { if (++g_protect > g_protect_max) g_protect_max = g_protect; return x; }
void Rf_unprotect(int n) { g_protect -= n; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP *)x->data)[i] = v; return v; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP *)x->data)[i]; }
void Rprintf(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  const int k = vsnprintf(g_out + g_out_len, sizeof g_out - g_out_len, fmt, ap);
  va_end(ap);
  if (k > 0) g_out_len += (size_t)k < sizeof g_out - g_out_len ? (size_t)k : sizeof g_out - g_out_len - 1;
}
void Rf_error(const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  if (g_jmp_armed) longjmp(g_jmp, 1);
  fprintf(stderr, "mini_r: Rf_error outside a call: %s\n", g_err);
  abort();
}
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot)
{
  (void)tag; (void)prot;
  SEXP s = Rf_allocVector(EXTPTRSXP, 0);
  free(s->data);
  s->data = p;
  return s;
}
void *R_ExternalPtrAddr(SEXP s) { return s->data; }
void R_SetExternalPtrAddr(SEXP s, void *p) { s->data = p; }
void R_ClearExternalPtr(SEXP s) { s->data = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit)
{
  (void)onexit;
  if (g_fin_n == 64) {   /* forget the pointers that have been cleared already */
    int k = 0;
    for (int i = 0; i < g_fin_n; i++) if (g_fin_obj[i]->data) { g_fin_obj[k] = g_fin_obj[i]; g_fin_fun[k] = g_fin_fun[i]; k++; }
    g_fin_n = k;
  }
  if (g_fin_n < 64) { g_fin_obj[g_fin_n] = s; g_fin_fun[g_fin_n] = fun; g_fin_n++; }
}
char *R_alloc(size_t n, int size) { return (char *)calloc(n ? n : 1, (size_t)size); }   /* (leaks: a test process) */

int R_registerRoutines(DllInfo *info, const R_CMethodDef *const c, const R_CallMethodDef *const call,
                       const R_FortranMethodDef *const f, const R_ExternalMethodDef *const e)
{
  (void)info;
  g_reg_n = 0;
  if (c || f || e) g_reg_n = -100;                       /* the reference registers .Call routines only */
  for (int i = 0; call && call[i].name && i < 8; i++) {
    strncpy(g_reg_names[i], call[i].name, 63);
    g_reg_nargs[i] = call[i].numArgs;
    g_reg_fun[i] = call[i].fun;
    g_reg_n = i + 1;
  }
  return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value) { (void)info; g_dyn = (int)value; return TRUE; }

/* ---- what the test drives ---- */
void minir_fail_alloc_after(int k) { g_alloc_countdown = k; }
/* what a garbage collection does to unreachable external pointers: every registered finalizer runs once; returns how many found
 * their pointer still set (objects a longjmp left behind) */
int minir_run_finalizers(void)
{
  int live = 0;
  for (int i = 0; i < g_fin_n; i++) {
    if (g_fin_obj[i]->data) live++;
    g_fin_fun[i](g_fin_obj[i]);
  }
  g_fin_n = 0;
  return live;
}
int minir_n_registered(void) { return g_reg_n; }
const char *minir_registered_name(int i) { return g_reg_names[i]; }
int minir_registered_nargs(int i) { return g_reg_nargs[i]; }
void *minir_registered_fun(int i) { return (void *)g_reg_fun[i]; }
int minir_dynamic_symbols(void) { return g_dyn; }
const char *minir_output(void) { g_out[g_out_len] = 0; return g_out; }
void minir_reset_output(void) { g_out_len = 0; g_out[0] = 0; g_err[0] = 0; }
const char *minir_error(void) { return g_err; }
int minir_protect_balance(void) { return g_protect; }
int minir_protect_max(void) { return g_protect_max; }
SEXP minir_real(const double *v, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); if (n) memcpy(s->data, v, (size_t)n * 8); return s; }
SEXP minir_int(const int *v, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); if (n) memcpy(s->data, v, (size_t)n * 4); return s; }
SEXP minir_int_matrix(const int *v, int nrow, int ncol) { SEXP s = minir_int(v, (R_xlen_t)nrow * ncol); s->nrow = nrow; s->ncol = ncol; return s; }
SEXP minir_nil(void) { return R_NilValue; }
const char *minir_name(SEXP s, int i) { return (s->names && i < (int)s->names->n) ? R_CHAR(STRING_ELT(s->names, i)) : ""; }
const unsigned char *minir_raw(SEXP s) { return (const unsigned char *)s->data; }
const int *minir_int_data(SEXP s) { return (const int *)s->data; }
int minir_type(SEXP s) { return (int)s->type; }
int minir_nrow(SEXP s) { return s->nrow; }
int minir_ncol(SEXP s) { return s->ncol; }
/* .Call through the registered pointer; returns NULL (and keeps the message) if the routine raised an R error */
SEXP minir_call5(void *fn, SEXP a, SEXP b, SEXP c, SEXP d, SEXP e)
{
  SEXP r = NULL;
  g_protect = 0;
  g_jmp_armed = 1;
  if (setjmp(g_jmp) == 0) r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a, b, c, d, e);
  g_jmp_armed = 0;
  return r;
}
/* any arity up to 13 (the arities the shim registers) */
SEXP minir_callv(void *fn, int nargs, SEXP *a)
{
  SEXP r = NULL;
  g_protect = 0;
  g_jmp_armed = 1;
  if (setjmp(g_jmp) == 0) {
    switch (nargs) {
      case 3: r = ((SEXP (*)(SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2]); break;
      case 4: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3]); break;
      case 5: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4]); break;
      case 6: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
      case 13: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(
                   a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12]); break;
      case 14: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(
                   a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13]); break;
      case 15: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(
                   a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14]); break;
      case 16: r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(
                   a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]); break;
      default: snprintf(g_err, sizeof g_err, "mini_r: no caller for %d arguments", nargs);
    }
  }
  g_jmp_armed = 0;
  return r;
}
SEXP minir_call6(void *fn, SEXP a, SEXP b, SEXP c, SEXP d, SEXP e, SEXP f)
{
  SEXP r = NULL;
  g_protect = 0;
  g_jmp_armed = 1;
  if (setjmp(g_jmp) == 0) r = ((SEXP (*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a, b, c, d, e, f);
  g_jmp_armed = 0;
  return r;
}
