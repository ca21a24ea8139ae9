/* mock_mx.c -- a minimal in-memory stand-in for the parts of MATLAB's mx / mex runtime that matlab/nd_dwt_hip_mex.c calls.
 * TEST INFRASTRUCTURE (tests/test_gpu_mex_gateway.py): MATLAB is not on the image, so the gateway -- argument parsing, plan cache, the
 * band-count guard, the handle registry and every command -- would otherwise only ever be syntax-checked.  Here it is compiled
 * against tests/stubs/{mex,matrix}.h, linked with this file and libndwt_hip.so, and driven from Python through the `mock_*` functions
 * below.  Nothing in this file is shipped or says anything about MATLAB's own behaviour beyond the documented contracts the gateway
 * relies on: column-major numeric arrays with separate (split) or interleaved complex storage, mexErrMsgIdAndTxt does not return,
 * mexAtExit functions run when the mex file is cleared. */
#define _POSIX_C_SOURCE 200809L
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"

struct mxArray_tag {
    int cls;               /* mxClassID, or -1 char, -2 cell */
    int is_complex;
    mwSize ndim;
    mwSize dims[8];
    void* re;              /* real part, or the interleaved data */
    void* im;              /* split storage only */
    char* str;
    struct mxArray_tag** cells;
    size_t ncells;
};

static jmp_buf g_jmp;
static int g_armed = 0;
static char g_err[1024];
static void (*g_at_exit[8])(void);
static int g_n_at_exit = 0;

static size_t elsize(int cls) { return cls == mxSINGLE_CLASS ? 4 : 8; }
static size_t numel(const mxArray* a) {
    size_t n = 1;
    for (mwSize i = 0; i < a->ndim; ++i) n *= a->dims[i];
    return n;
}

int mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
int mxIsSingle(const mxArray* a) { return a->cls == mxSINGLE_CLASS; }
int mxIsComplex(const mxArray* a) { return a->is_complex; }
int mxIsCell(const mxArray* a) { return a->cls == -2; }
int mxIsChar(const mxArray* a) { return a->cls == -1; }
int mxIsUint64(const mxArray* a) { return a->cls == mxUINT64_CLASS; }
double mxGetScalar(const mxArray* a) {
    if (a->cls == mxSINGLE_CLASS) return *(const float*)a->re;
    if (a->cls == mxUINT64_CLASS) return (double)*(const uint64_t*)a->re;
    return *(const double*)a->re;
}
int mxGetString(const mxArray* a, char* buf, mwSize buflen) {
    if (a->cls != -1 || strlen(a->str) + 1 > buflen) return 1;
    strcpy(buf, a->str);
    return 0;
}
mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
size_t mxGetNumberOfElements(const mxArray* a) { return a->cls == -2 ? a->ncells : (a->cls == -1 ? strlen(a->str) : numel(a)); }
mxArray* mxGetCell(const mxArray* a, mwSize i) { return a->cells[i]; }
void* mxGetData(const mxArray* a) { return a->re; }
void* mxGetImagData(const mxArray* a) { return a->im; }

static mxArray* new_numeric(mwSize ndim, const mwSize* dims, int cls, int cplx, int zero) {
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    a->cls = cls;
    a->is_complex = cplx;
    a->ndim = ndim < 2 ? 2 : ndim;
    a->dims[0] = a->dims[1] = 1;
    for (mwSize i = 0; i < ndim; ++i) a->dims[i] = dims[i];
    const size_t bytes = numel(a) * elsize(cls);
#if MX_HAS_INTERLEAVED_COMPLEX
    a->re = zero ? calloc(1, bytes * (cplx ? 2 : 1) + 16) : malloc(bytes * (cplx ? 2 : 1) + 16);
#else
    a->re = zero ? calloc(1, bytes + 16) : malloc(bytes + 16);
    if (cplx) a->im = zero ? calloc(1, bytes + 16) : malloc(bytes + 16);
#endif
    return a;
}
mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID cls, mxComplexity flag) {
    mwSize d[2] = {m, n};
    return new_numeric(2, d, cls, flag == mxCOMPLEX, 1);
}
mxArray* mxCreateNumericArray(mwSize ndim, const mwSize* dims, mxClassID cls, mxComplexity flag) { return new_numeric(ndim, dims, cls, flag == mxCOMPLEX, 1); }
mxArray* mxCreateUninitNumericArray(mwSize ndim, mwSize* dims, mxClassID cls, mxComplexity flag) { return new_numeric(ndim, dims, cls, flag == mxCOMPLEX, 0); }

void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...) {
    va_list ap;
    int n = snprintf(g_err, sizeof g_err, "%s: ", id);
    va_start(ap, fmt);
    vsnprintf(g_err + n, sizeof g_err - (size_t)n, fmt, ap);
    va_end(ap);
    if (g_armed) longjmp(g_jmp, 1);
    fprintf(stderr, "mexErrMsgIdAndTxt outside a call: %s\n", g_err);
    abort();
}
int mexAtExit(void (*fn)(void)) {
    if (g_n_at_exit < 8) g_at_exit[g_n_at_exit++] = fn;
    return 0;
}

/* ---- what the Python test drives ---- */
int mock_interleaved(void) { return MX_HAS_INTERLEAVED_COMPLEX; }
mxArray* mock_numeric(int ndim, const int64_t* dims, int is_single, int is_complex, const void* re, const void* im) {
    mwSize d[8];
    for (int i = 0; i < ndim; ++i) d[i] = (mwSize)dims[i];
    mxArray* a = new_numeric((mwSize)ndim, d, is_single ? mxSINGLE_CLASS : mxDOUBLE_CLASS, is_complex, 0);
    const size_t bytes = numel(a) * elsize(a->cls);
#if MX_HAS_INTERLEAVED_COMPLEX
    (void)im;
    memcpy(a->re, re, bytes * (is_complex ? 2 : 1));   /* interleaved (re, im) pairs */
#else
    memcpy(a->re, re, bytes);
    if (is_complex) memcpy(a->im, im, bytes);
#endif
    return a;
}
mxArray* mock_uint64(uint64_t v) {
    mxArray* a = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
    *(uint64_t*)a->re = v;
    return a;
}
mxArray* mock_string(const char* s) {
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    a->cls = -1;
    a->ndim = 2;
    a->dims[0] = 1;
    a->dims[1] = strlen(s);
    a->str = strdup(s);
    return a;
}
mxArray* mock_cell(int n, mxArray* const* items) {
    mxArray* a = (mxArray*)calloc(1, sizeof *a);
    a->cls = -2;
    a->ndim = 2;
    a->dims[0] = 1;
    a->dims[1] = (mwSize)n;
    a->ncells = (size_t)n;
    a->cells = (mxArray**)malloc(sizeof(mxArray*) * (size_t)n);
    for (int i = 0; i < n; ++i) a->cells[i] = items[i];
    return a;
}
void mock_free(mxArray* a) {
    if (!a) return;
    free(a->re);
    free(a->im);
    free(a->str);
    free(a->cells);                                    /* (the items are freed by whoever made them) */
    free(a);
}
int mock_ndim(const mxArray* a) { return (int)a->ndim; }
int64_t mock_dim(const mxArray* a, int i) { return (int64_t)a->dims[i]; }
int mock_is_single(const mxArray* a) { return a->cls == mxSINGLE_CLASS; }
int mock_is_complex(const mxArray* a) { return a->is_complex; }
void* mock_real(const mxArray* a) { return a->re; }
void* mock_imag(const mxArray* a) { return a->im; }
const char* mock_last_error(void) { return g_err; }

/* y = nd_dwt_hip_mex(args...) with nlhs outputs: 0 on success (out[0] = the result, or NULL), 1 if the gateway raised an error */
int mock_call(int nlhs, mxArray** out, int nrhs, mxArray* const* args) {
    mxArray* plhs[2] = {NULL, NULL};
    g_err[0] = 0;
    g_armed = 1;
    if (setjmp(g_jmp)) {
        g_armed = 0;
        return 1;
    }
    mexFunction(nlhs, plhs, nrhs, (const mxArray**)args);
    g_armed = 0;
    if (out) out[0] = plhs[0];
    return 0;
}
/* `clear mex`: the registered exit functions */
void mock_clear(void) {
    for (int i = g_n_at_exit - 1; i >= 0; --i) g_at_exit[i]();
    g_n_at_exit = 0;
}
