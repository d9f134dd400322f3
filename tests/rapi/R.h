/* TEST INFRASTRUCTURE -- see Rinternals.h in this directory. */
#ifndef TESTS_RAPI_R_H
#define TESTS_RAPI_R_H
#include <stdlib.h>
#endif
