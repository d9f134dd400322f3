/* TEST INFRASTRUCTURE -- declarations only (see ../Rinternals.h): native-routine registration of R's API. */
#ifndef TESTS_RAPI_RDYNLOAD_H
#define TESTS_RAPI_RDYNLOAD_H
#include "../Rinternals.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef void *(*DL_FUNC)(void);
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct { const char *name; DL_FUNC fun; int numArgs; void *types; } R_CMethodDef;
typedef R_CMethodDef R_FortranMethodDef;
typedef R_CallMethodDef R_ExternalMethodDef;
typedef struct _DllInfo DllInfo;
int R_registerRoutines(DllInfo *info, const R_CMethodDef *const croutines, const R_CallMethodDef *const callRoutines,
                       const R_FortranMethodDef *const fortranRoutines, const R_ExternalMethodDef *const externalRoutines);
Rboolean R_useDynamicSymbols(DllInfo *info, Rboolean value);
#ifdef __cplusplus
}
#endif
#endif
