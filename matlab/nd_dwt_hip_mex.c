/*
 * nd_dwt_hip_mex.c -- thin MATLAB gateway onto the C ABI of libndwt_hip.so (include/ndwt.h).
 *
 * Replaces reference mex/nd_dwt_mex.c:8-153 for compute = 'hip'.  Same call shape as the reference gateway
 *     y = nd_dwt_mex(x_f, f_dec, dir, level, pres_l2_norm)            (nd_dwt_3D.m:161,225)
 * but the arrays are in the SIGNAL domain and the filters are named, not materialised:
 *     y = nd_dwt_hip_mex(x, wnames, dir, level, pres_l2_norm [, dilation])
 *   x        real or complex, single or double; forward: size = sizes; inverse: [sizes, bands]
 *   wnames   cell array of 'dbK', one per axis (1-D: a single string)
 *   dir      0 forward, nonzero inverse                                  (nd_dwt_mex.c:33,106)
 *   dilation optional 'reference' (default) | 'atrous'
 * Complex data: with the interleaved-complex mex API (mex -R2018a, MX_HAS_INTERLEAVED_COMPLEX) the array goes through
 * an NDWT_COMPLEX_INTERLEAVED plan; with the split API of the reference's gateway (mxGetPr / mxGetPi,
 * nd_dwt_mex.c:55-58) the real and imaginary parts go through ndwt_{dec,rec}_split_host on a real plan.
 * Build:  matlab/ndwt_hip_compile.m.
 * NOT compiled in the build container (no MATLAB / mex.h there); all behaviour is tested through the C ABI.
 */
#include <string.h>

#include "mex.h"
#include "matrix.h"
#include "ndwt.h"

static void fail(const char* what) {
    /* the reference raises this identifier for every gateway error (nd_dwt_mex.c:20,24,28,37,42,49,125) */
    mexErrMsgIdAndTxt("MATLAB:FFT2mx:invalidNumInputs", "%s: %s", what, ndwt_last_error());
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    if (nrhs < 5) mexErrMsgIdAndTxt("MATLAB:FFT2mx:invalidNumInputs", "Five Inputs Required");
    const mxArray* x = prhs[0];
    if (!mxIsDouble(x) && !mxIsSingle(x)) mexErrMsgIdAndTxt("MATLAB:FFT2mx:invalidNumInputs", "Arrays must be double or single");
    const int inverse = mxGetScalar(prhs[2]) != 0;
    const int level = (int)mxGetScalar(prhs[3]);
    const int l2 = mxGetScalar(prhs[4]) != 0;
    int dilation = NDWT_DILATION_REFERENCE;
    if (nrhs > 5) {
        char buf[16];
        mxGetString(prhs[5], buf, sizeof buf);
        if (!strcmp(buf, "atrous")) dilation = NDWT_DILATION_ATROUS;
    }

    /* dims: column vectors are 1-D like nd_dwt_mex.c:68-70; the inverse input carries the band axis last (:115) */
    mwSize nd = mxGetNumberOfDimensions(x);
    const mwSize* d = mxGetDimensions(x);
    int ndim = (int)nd - (inverse ? 1 : 0);
    if (!inverse && nd == 2 && d[1] == 1) ndim = 1;
    if (inverse && nd == 2) ndim = 1;
    if (ndim < 1 || ndim > NDWT_MAX_DIMS) mexErrMsgIdAndTxt("MATLAB:FFT2mx:invalidNumInputs", "1 to 4 dimensions supported");
    int64_t dims[NDWT_MAX_DIMS];
    for (int a = 0; a < ndim; ++a) dims[a] = (int64_t)d[a];

    /* wavelet names */
    char names[NDWT_MAX_DIMS][16];
    const char* wn[NDWT_MAX_DIMS];
    for (int a = 0; a < ndim; ++a) {
        if (mxIsCell(prhs[1])) mxGetString(mxGetCell(prhs[1], a < (int)mxGetNumberOfElements(prhs[1]) ? a : 0), names[a], 16);
        else mxGetString(prhs[1], names[a], 16);
        wn[a] = names[a];
    }

    const int dtype = mxIsSingle(x) ? NDWT_F32 : NDWT_F64;
#if MX_HAS_INTERLEAVED_COMPLEX
    const int cplx = mxIsComplex(x) ? NDWT_COMPLEX_INTERLEAVED : NDWT_REAL;
#else
    const int cplx = NDWT_REAL;                          /* split storage: one real transform per part */
#endif
    ndwt_plan* plan = NULL;
    if (ndwt_plan_create(&plan, ndim, dims, wn, dtype, cplx, l2, dilation, level, 0) != NDWT_OK) fail("plan");

    /* output: MATLAB-owned, like mxCreateNumericArray at nd_dwt_mex.c:86,136 */
    mwSize od[NDWT_MAX_DIMS + 1];
    for (int a = 0; a < ndim; ++a) od[a] = (mwSize)dims[a];
    mwSize ond = (mwSize)ndim;
    if (!inverse) od[ond++] = (mwSize)ndwt_num_bands(ndim, level);
    if (ond == 1) od[ond++] = 1;
    plhs[0] = mxCreateNumericArray(ond, od, mxIsSingle(x) ? mxSINGLE_CLASS : mxDOUBLE_CLASS, mxIsComplex(x) ? mxCOMPLEX : mxREAL);

#if MX_HAS_INTERLEAVED_COMPLEX
    int rc = inverse ? ndwt_rec_host(plan, mxGetData(x), mxGetData(plhs[0]), level)
                     : ndwt_dec_host(plan, mxGetData(x), mxGetData(plhs[0]), level);
#else
    const void* x_im = mxIsComplex(x) ? mxGetImagData(x) : NULL;
    void* y_im = mxIsComplex(x) ? mxGetImagData(plhs[0]) : NULL;
    int rc = inverse ? ndwt_rec_split_host(plan, mxGetData(x), x_im, mxGetData(plhs[0]), y_im, level)
                     : ndwt_dec_split_host(plan, mxGetData(x), x_im, mxGetData(plhs[0]), y_im, level);
#endif
    ndwt_plan_destroy(plan);
    if (rc != NDWT_OK) fail(inverse ? "rec" : "dec");
    (void)nlhs;
}
