/* Declarations-only stand-in for MATLAB's matrix.h (see mex.h next to it): syntax check only. */
#ifndef NDWT_TEST_STUB_MATRIX_H
#define NDWT_TEST_STUB_MATRIX_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxSINGLE_CLASS = 7, mxDOUBLE_CLASS = 6, mxUINT64_CLASS = 15 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
#ifndef MX_HAS_INTERLEAVED_COMPLEX
#define MX_HAS_INTERLEAVED_COMPLEX 0
#endif
int mxIsDouble(const mxArray*);
int mxIsSingle(const mxArray*);
int mxIsComplex(const mxArray*);
int mxIsCell(const mxArray*);
int mxIsChar(const mxArray*);
int mxIsUint64(const mxArray*);
double mxGetScalar(const mxArray*);
int mxGetString(const mxArray*, char* buf, mwSize buflen);
mwSize mxGetNumberOfDimensions(const mxArray*);
const mwSize* mxGetDimensions(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mxArray* mxGetCell(const mxArray*, mwSize index);
void* mxGetData(const mxArray*);
void* mxGetImagData(const mxArray*);
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag);
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag);
mxArray* mxCreateUninitNumericArray(mwSize ndim, mwSize* dims, mxClassID cls, mxComplexity flag);   /* R2015a+ */
#endif
