/* TEST INFRASTRUCTURE -- declarations only.
 *
 * The subset of R's C API that shim/edcore_shim.c uses, declared (not implemented) so that the shim can be compiled
 * and linked in an image without R.  Names, argument types and constants are those of R's public API (Rinternals.h,
 * R_ext/Rdynload.h, R_ext/Print.h, R_ext/Error.h, R_ext/Memory.h).  Never used to build the product (libedcore.so), the
 * checker (oracle/) or anything of the reference: only tests/test_shim.py compiles against it, and runs the shim on
 * top of tests/rapi/mini_r.c. */
#ifndef TESTS_RAPI_RINTERNALS_H
#define TESTS_RAPI_RINTERNALS_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
typedef enum { FALSE = 0, TRUE } Rboolean;

#define CHARSXP 9
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
#define EXTPTRSXP 22
#define RAWSXP 24
#define NA_INTEGER R_NaInt
#define NA_REAL R_NaReal

extern SEXP R_NilValue;
extern SEXP R_NamesSymbol;
extern int R_NaInt;
extern double R_NaReal;
typedef unsigned char Rbyte;

double *REAL(SEXP x);
int *INTEGER(SEXP x);
R_xlen_t XLENGTH(SEXP x);
int LENGTH(SEXP x);
SEXP Rf_allocVector(SEXPTYPE type, R_xlen_t length);
SEXP Rf_allocMatrix(SEXPTYPE type, int nrow, int ncol);
SEXP Rf_protect(SEXP x);
void Rf_unprotect(int n);
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP VECTOR_ELT(SEXP x, R_xlen_t i);
Rbyte *RAW(SEXP x);
int Rf_nrows(SEXP x);
int Rf_ncols(SEXP x);
SEXP Rf_mkChar(const char *s);
void SET_STRING_ELT(SEXP x, R_xlen_t i, SEXP v);
SEXP STRING_ELT(SEXP x, R_xlen_t i);
const char *R_CHAR(SEXP x);
SEXP Rf_setAttrib(SEXP x, SEXP name, SEXP val);
SEXP Rf_getAttrib(SEXP x, SEXP name);
void Rprintf(const char *fmt, ...);
void Rf_error(const char *fmt, ...) __attribute__((noreturn));
char *R_alloc(size_t n, int size);
typedef void (*R_CFinalizer_t)(SEXP);
SEXP R_MakeExternalPtr(void *p, SEXP tag, SEXP prot);
void *R_ExternalPtrAddr(SEXP s);
void R_SetExternalPtrAddr(SEXP s, void *p);
void R_ClearExternalPtr(SEXP s);
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit);

#define allocVector Rf_allocVector
#define allocMatrix Rf_allocMatrix
#define nrows Rf_nrows
#define ncols Rf_ncols
#define mkChar Rf_mkChar
#define setAttrib Rf_setAttrib
#define getAttrib Rf_getAttrib
#define CHAR(x) R_CHAR(x)
#define PROTECT(x) Rf_protect(x)
#define UNPROTECT(n) Rf_unprotect(n)

#ifdef __cplusplus
}
#endif
#endif
