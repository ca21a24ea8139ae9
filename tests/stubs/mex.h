/* Declarations-only stand-in for MATLAB's mex.h -- TEST INFRASTRUCTURE for `cc -fsyntax-only matlab/nd_dwt_hip_mex.c`
 * (tests/test_abi.py).  MATLAB is not present in the build container; this file only lets the compiler check the gateway's
 * syntax and the types it passes.  It is never linked, never shipped and pins no behaviour. */
#ifndef NDWT_TEST_STUB_MEX_H
#define NDWT_TEST_STUB_MEX_H
#include "matrix.h"
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#endif
