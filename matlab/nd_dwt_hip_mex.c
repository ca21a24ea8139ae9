/*
 * nd_dwt_hip_mex.c -- thin MATLAB gateway onto the C ABI of libndwt_hip.so (include/ndwt.h).
 *
 * Replaces reference mex/nd_dwt_mex.c:8-153 for compute = 'hip'.  Same call shape as the reference gateway
 *     y = nd_dwt_mex(x_f, f_dec, dir, level, pres_l2_norm)            (nd_dwt_3D.m:161,225)
 * but the arrays are in the SIGNAL domain and the filters are named, not materialised:
 *     y = nd_dwt_hip_mex(x, wnames, dir, level, pres_l2_norm [, dilation [, devices [, exchange]]])
 *   x        real or complex, single or double; forward: size = sizes; inverse: [sizes, bands]
 *   wnames   cell array of 'dbK', one per axis (1-D: a single string)
 *   dir      0 forward, nonzero inverse                                  (nd_dwt_mex.c:33,106)
 *   dilation optional 'reference' (default) | 'atrous'
 *   devices  optional vector of HIP device ordinals: the volume is sharded on its outermost axis over them by the
 *            single-process multi-device plan (ndwt_mplan_*, one slab per entry; 2-D .. 4-D)
 *   exchange optional, with several devices: 'scatter' (default: the synthesis exchanges one band of partial sums per level; equal to one
 *            device to rounding, the order of summation differs) | 'gather' (the halo planes of every band: bit-identical to one device)
 * Complex data: with the interleaved-complex mex API (mex -R2018a, MX_HAS_INTERLEAVED_COMPLEX) the array goes through
 * an NDWT_COMPLEX_INTERLEAVED plan; with the split API of the reference's gateway (mxGetPr / mxGetPi,
 * nd_dwt_mex.c:55-58) the real and imaginary parts go through ndwt_{dec,rec}_split_host on a real plan.
 *
 * Commands (first argument a string) keep the coefficients on the device between calls -- the reference's call sites nd_dwt_3D.m:161,225
 * move 11.8 GB per direction over PCIe for a 512^3 3-level transform, 0.2 s against 3 ms of compute; with a handle only the signal moves:
 *     h  = nd_dwt_hip_mex('dec_keep', x, wnames, level, pres_l2_norm [, dilation [, device]])   uint64 handle of device-resident coefficients
 *     x  = nd_dwt_hip_mex('rec_handle', h)                 reconstruct from them
 *          nd_dwt_hip_mex('shrink', h, threshold [, 'soft' | 'hard'])      threshold the detail bands in place
 *     y  = nd_dwt_hip_mex('fetch', h)                      the coefficient array [sizes, bands] (for callers that do need it)
 *          nd_dwt_hip_mex('release', h)                    free them ('release_all': every live handle)
 *     xd = nd_dwt_hip_mex('denoise', x, wnames, level, pres_l2_norm, threshold [, 'soft' | 'hard' [, dilation [, device]]])
 *                                                          rec(shrink(dec(x))) in one call, coefficients never leave the device
 * Handles take real data, or complex data with the interleaved-complex API (mex -R2018a).
 *
 * Plans are cached across calls (iterative solvers call dec / rec with one configuration thousands of times; a plan owns two
 * scratch volumes on the device) and released by mexAtExit -- the reference re-plans FFTW on every call (nddwt.c:110-111).
 * Build: matlab/ndwt_hip_compile.m.  No MATLAB in the build container: tests/test_abi.py checks this file's syntax against
 * declarations-only stand-ins of mex.h / matrix.h; behaviour is tested through the C ABI it calls.
 */
#include <stdint.h>
#include <string.h>

#include "mex.h"
#include "matrix.h"
#include "ndwt.h"

#define ERR_ID "MATLAB:FFT2mx:invalidNumInputs" /* the identifier of every reference gateway error (nd_dwt_mex.c:20,...,125) */
#define NCACHE 8

typedef struct {
    ndwt_plan* plan;
    int pins;            /* live coefficient handles made by this plan: it is not evicted while they exist */
    int ndim, dtype, cplx, l2, dilation, max_level, device;
    int64_t dims[NDWT_MAX_DIMS];
    char names[NDWT_MAX_DIMS][16];
    unsigned long stamp; /* last use, for eviction */
} cached_plan;

static cached_plan g_cache[NCACHE];
static int g_last_slot = 0;  /* g_cache entry the last get_plan call returned */
static unsigned long g_clock = 0;
static int g_at_exit = 0;

/* live device-resident coefficient sets: the uint64 a caller holds is an index + generation, never a raw pointer */
#define NHANDLE 64
typedef struct {
    ndwt_coef* coef;
    int slot;            /* g_cache entry of the plan it belongs to */
    unsigned gen;
    int ndim, is_single, is_complex;
    int64_t dims[NDWT_MAX_DIMS];
} live_coef;
static live_coef g_coef[NHANDLE];
static unsigned g_gen = 0;

static void release_coefs(void) {
    int i;
    for (i = 0; i < NHANDLE; ++i) {
        if (g_coef[i].coef) {
            ndwt_coef_release(g_coef[i].coef);
            g_coef[i].coef = NULL;
            g_cache[g_coef[i].slot].pins--;
        }
    }
}

static void release_plans(void) {
    int i;
    release_coefs();     /* handles first: they are bound to their plans */
    for (i = 0; i < NCACHE; ++i) {
        if (g_cache[i].plan) ndwt_plan_destroy(g_cache[i].plan);
        g_cache[i].plan = NULL;
    }
}

static void fail(const char* what) { mexErrMsgIdAndTxt(ERR_ID, "%s: %s", what, ndwt_last_error()); }

/* the cached plan of this configuration (created, or re-created with more levels, when needed) */
static ndwt_plan* get_plan(int ndim, const int64_t* dims, char names[][16], int dtype, int cplx, int l2, int dilation, int level, int device) {
    int i, a, victim = 0;
    const char* wn[NDWT_MAX_DIMS];
    if (!g_at_exit) {
        mexAtExit(release_plans);
        g_at_exit = 1;
    }
    for (i = 0; i < NCACHE; ++i) {
        cached_plan* c = &g_cache[i];
        int same = c->plan && c->ndim == ndim && c->dtype == dtype && c->cplx == cplx && c->l2 == l2 && c->dilation == dilation && c->device == device;
        for (a = 0; same && a < ndim; ++a) same = c->dims[a] == dims[a] && !strcmp(c->names[a], names[a]);
        if (same && c->max_level >= level) {
            c->stamp = ++g_clock;
            g_last_slot = i;
            return c->plan;
        }
        if (same) { /* same transform, deeper than planned for: rebuild in place */
            victim = i;
            goto build;
        }
    }
    victim = -1;
    for (i = 0; i < NCACHE && victim < 0; ++i) /* an empty slot ... */
        if (!g_cache[i].plan) victim = i;
    if (victim < 0) {                          /* ... else the least recently used one that no coefficient handle is bound to */
        for (i = 0; i < NCACHE; ++i)
            if (!g_cache[i].pins && (victim < 0 || g_cache[i].stamp < g_cache[victim].stamp)) victim = i;
        if (victim < 0) mexErrMsgIdAndTxt(ERR_ID, "every cached plan has live coefficient handles: release some ('release')");
    }
build:
    if (g_cache[victim].pins) mexErrMsgIdAndTxt(ERR_ID, "this configuration has live coefficient handles of fewer levels: release them first");
    if (g_cache[victim].plan) ndwt_plan_destroy(g_cache[victim].plan);
    g_cache[victim].plan = NULL;
    for (a = 0; a < ndim; ++a) wn[a] = names[a];
    if (ndwt_plan_create(&g_cache[victim].plan, ndim, dims, wn, dtype, cplx, l2, dilation, level < 3 ? 3 : level, device) != NDWT_OK) {
        g_cache[victim].plan = NULL;
        fail("plan");
    }
    g_cache[victim].ndim = ndim;
    g_cache[victim].dtype = dtype;
    g_cache[victim].cplx = cplx;
    g_cache[victim].l2 = l2;
    g_cache[victim].dilation = dilation;
    g_cache[victim].device = device;
    g_cache[victim].max_level = level < 3 ? 3 : level;
    for (a = 0; a < ndim; ++a) {
        g_cache[victim].dims[a] = dims[a];
        strcpy(g_cache[victim].names[a], names[a]);
    }
    g_cache[victim].stamp = ++g_clock;
    g_cache[victim].pins = 0;
    g_last_slot = victim;
    return g_cache[victim].plan;
}

/* one cached multi-device plan (the last configuration used) */
static struct {
    ndwt_mplan* plan;
    int ndim, dtype, cplx, l2, dilation, max_level, ndev;
    int devices[64];
    int64_t dims[NDWT_MAX_DIMS];
    char names[NDWT_MAX_DIMS][16];
} g_multi;

static void release_multi(void) {
    if (g_multi.plan) ndwt_mplan_destroy(g_multi.plan);
    g_multi.plan = NULL;
}

static ndwt_mplan* get_mplan(int ndim, const int64_t* dims, char names[][16], int dtype, int cplx, int l2, int dilation, int level,
                             const int* devices, int ndev) {
    int a, same = g_multi.plan && g_multi.ndim == ndim && g_multi.dtype == dtype && g_multi.cplx == cplx && g_multi.l2 == l2 &&
                  g_multi.dilation == dilation && g_multi.ndev == ndev && g_multi.max_level >= level;
    const char* wn[NDWT_MAX_DIMS];
    static int at_exit = 0;
    for (a = 0; same && a < ndim; ++a) same = g_multi.dims[a] == dims[a] && !strcmp(g_multi.names[a], names[a]);
    for (a = 0; same && a < ndev; ++a) same = g_multi.devices[a] == devices[a];
    if (same) return g_multi.plan;
    if (!at_exit) {
        mexAtExit(release_multi);
        at_exit = 1;
    }
    release_multi();
    for (a = 0; a < ndim; ++a) wn[a] = names[a];
    if (ndwt_mplan_create(&g_multi.plan, ndim, dims, wn, dtype, cplx, l2, dilation, level < 3 ? 3 : level, devices, ndev) != NDWT_OK) {
        g_multi.plan = NULL;
        mexErrMsgIdAndTxt(ERR_ID, "multi-device plan: %s", ndwt_mplan_last_error());
    }
    g_multi.ndim = ndim; g_multi.dtype = dtype; g_multi.cplx = cplx; g_multi.l2 = l2; g_multi.dilation = dilation;
    g_multi.max_level = level < 3 ? 3 : level; g_multi.ndev = ndev;
    for (a = 0; a < ndim; ++a) {
        g_multi.dims[a] = dims[a];
        strcpy(g_multi.names[a], names[a]);
    }
    for (a = 0; a < ndev; ++a) g_multi.devices[a] = devices[a];
    return g_multi.plan;
}

/* sizes / wavelet names / precision of a signal array -> the cached plan (shared by the transform call and the commands) */
static ndwt_plan* plan_for_signal(const mxArray* x, const mxArray* wn, int level, int l2, int dilation, int device, int* ndim_out, int64_t* dims) {
    const mwSize nd = mxGetNumberOfDimensions(x);
    const mwSize* d = mxGetDimensions(x);
    char names[NDWT_MAX_DIMS][16];
    int ndim = (int)nd, a, cplx;
    if (!mxIsDouble(x) && !mxIsSingle(x)) mexErrMsgIdAndTxt(ERR_ID, "Arrays must be double or single");
    if (nd == 2 && d[1] == 1) ndim = 1;
    if (ndim < 1 || ndim > NDWT_MAX_DIMS) mexErrMsgIdAndTxt(ERR_ID, "1 to 4 dimensions supported");
    for (a = 0; a < ndim; ++a) dims[a] = (int64_t)d[a];
    for (a = 0; a < ndim; ++a) {
        if (mxIsCell(wn)) {
            const int ncell = (int)mxGetNumberOfElements(wn);
            if (ncell != ndim && ncell != 1) mexErrMsgIdAndTxt(ERR_ID, "one wavelet name per dimension, or a single name");
            if (mxGetString(mxGetCell(wn, ncell == 1 ? 0 : a), names[a], 16)) mexErrMsgIdAndTxt(ERR_ID, "bad wavelet name");
        } else if (mxGetString(wn, names[a], 16)) {
            mexErrMsgIdAndTxt(ERR_ID, "bad wavelet name");
        }
    }
#if MX_HAS_INTERLEAVED_COMPLEX
    cplx = mxIsComplex(x) ? NDWT_COMPLEX_INTERLEAVED : NDWT_REAL;
#else
    if (mxIsComplex(x)) mexErrMsgIdAndTxt(ERR_ID, "device-resident handles take complex data with the interleaved-complex API only (mex -R2018a)");
    cplx = NDWT_REAL;
#endif
    *ndim_out = ndim;
    return get_plan(ndim, dims, names, mxIsSingle(x) ? NDWT_F32 : NDWT_F64, cplx, l2, dilation, level, device);
}

static int read_dilation(int nrhs, const mxArray* prhs[], int at) {
    char buf[16];
    if (nrhs <= at) return NDWT_DILATION_REFERENCE;
    mxGetString(prhs[at], buf, sizeof buf);
    return !strcmp(buf, "atrous") ? NDWT_DILATION_ATROUS : NDWT_DILATION_REFERENCE;
}
static int read_mode(int nrhs, const mxArray* prhs[], int at) {
    char buf[16];
    if (nrhs <= at) return NDWT_SHRINK_SOFT;
    mxGetString(prhs[at], buf, sizeof buf);
    return !strcmp(buf, "hard") ? NDWT_SHRINK_HARD : NDWT_SHRINK_SOFT;
}
static live_coef* find_handle(const mxArray* h) {
    uint64_t v;
    int idx;
    if (!mxIsUint64(h) || mxGetNumberOfElements(h) != 1) mexErrMsgIdAndTxt(ERR_ID, "a coefficient handle is a uint64 scalar returned by 'dec_keep'");
    v = *(const uint64_t*)mxGetData(h);
    idx = (int)(v & 0xFFFF);
    if (idx >= NHANDLE || !g_coef[idx].coef || g_coef[idx].gen != (unsigned)(v >> 16)) mexErrMsgIdAndTxt(ERR_ID, "stale or unknown coefficient handle");
    return &g_coef[idx];
}

/* the string commands; returns 0 if prhs[0] is not a command */
static int command(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    char cmd[24];
    int64_t dims[NDWT_MAX_DIMS];
    mwSize od[NDWT_MAX_DIMS + 1], ond;
    int ndim = 0, a;
    if (nrhs < 1 || !mxIsChar(prhs[0])) return 0;
    mxGetString(prhs[0], cmd, sizeof cmd);
    if (!strcmp(cmd, "dec_keep")) {          /* h = ('dec_keep', x, wnames, level, l2 [, dilation [, device]]) */
        int i, slot = -1, level, l2;
        ndwt_plan* plan;
        ndwt_coef* c = NULL;
        if (nrhs < 5) mexErrMsgIdAndTxt(ERR_ID, "dec_keep: x, wnames, level, pres_l2_norm");
        if (nlhs > 1) mexErrMsgIdAndTxt(ERR_ID, "Too many output arguments.");
        level = (int)mxGetScalar(prhs[3]);
        l2 = mxGetScalar(prhs[4]) != 0;
        if (level < 1) mexErrMsgIdAndTxt(ERR_ID, "level must be at least 1");
        for (i = 0; i < NHANDLE && slot < 0; ++i)
            if (!g_coef[i].coef) slot = i;
        if (slot < 0) mexErrMsgIdAndTxt(ERR_ID, "too many live coefficient handles (%d): release some", NHANDLE);
        plan = plan_for_signal(prhs[1], prhs[2], level, l2, read_dilation(nrhs, prhs, 5), nrhs > 6 ? (int)mxGetScalar(prhs[6]) : 0, &ndim, dims);
        if (ndwt_coef_dec_host(plan, mxGetData(prhs[1]), level, &c) != NDWT_OK) fail("dec_keep");
        g_coef[slot].coef = c;
        g_coef[slot].slot = g_last_slot;
        g_coef[slot].gen = ++g_gen;
        g_coef[slot].ndim = ndim;
        g_coef[slot].is_single = mxIsSingle(prhs[1]);
        g_coef[slot].is_complex = mxIsComplex(prhs[1]);
        for (a = 0; a < ndim; ++a) g_coef[slot].dims[a] = dims[a];
        g_cache[g_last_slot].pins++;
        plhs[0] = mxCreateNumericMatrix(1, 1, mxUINT64_CLASS, mxREAL);
        *(uint64_t*)mxGetData(plhs[0]) = ((uint64_t)g_coef[slot].gen << 16) | (uint64_t)slot;
        return 1;
    }
    if (!strcmp(cmd, "rec_handle") || !strcmp(cmd, "fetch")) {
        live_coef* lc;
        int level = 0;
        int64_t bands = 0;
        if (nrhs < 2) mexErrMsgIdAndTxt(ERR_ID, "%s: a handle", cmd);
        lc = find_handle(prhs[1]);
        ndwt_coef_info(lc->coef, &level, &bands, NULL, NULL);
        for (a = 0; a < lc->ndim; ++a) od[a] = (mwSize)lc->dims[a];
        ond = (mwSize)lc->ndim;
        if (cmd[0] == 'f') od[ond++] = (mwSize)bands;
        if (ond == 1) od[ond++] = 1;
        plhs[0] = mxCreateUninitNumericArray(ond, od, lc->is_single ? mxSINGLE_CLASS : mxDOUBLE_CLASS, lc->is_complex ? mxCOMPLEX : mxREAL);
        if ((cmd[0] == 'f' ? ndwt_coef_get_host(g_cache[lc->slot].plan, lc->coef, mxGetData(plhs[0]))
                           : ndwt_coef_rec_host(g_cache[lc->slot].plan, lc->coef, mxGetData(plhs[0]))) != NDWT_OK)
            fail(cmd);
        return 1;
    }
    if (!strcmp(cmd, "shrink")) {            /* ('shrink', h, threshold [, 'soft' | 'hard']) */
        live_coef* lc;
        if (nrhs < 3) mexErrMsgIdAndTxt(ERR_ID, "shrink: a handle and a threshold");
        lc = find_handle(prhs[1]);
        if (ndwt_coef_shrink(g_cache[lc->slot].plan, lc->coef, mxGetScalar(prhs[2]), read_mode(nrhs, prhs, 3)) != NDWT_OK) fail("shrink");
        return 1;
    }
    if (!strcmp(cmd, "release")) {
        live_coef* lc;
        if (nrhs < 2) mexErrMsgIdAndTxt(ERR_ID, "release: a handle");
        lc = find_handle(prhs[1]);
        ndwt_coef_release(lc->coef);
        lc->coef = NULL;
        g_cache[lc->slot].pins--;
        return 1;
    }
    if (!strcmp(cmd, "release_all")) {
        release_coefs();
        return 1;
    }
    if (!strcmp(cmd, "denoise")) {           /* xd = ('denoise', x, wnames, level, l2, threshold [, mode [, dilation [, device]]]) */
        ndwt_plan* plan;
        int level, l2;
        if (nrhs < 6) mexErrMsgIdAndTxt(ERR_ID, "denoise: x, wnames, level, pres_l2_norm, threshold");
        level = (int)mxGetScalar(prhs[3]);
        l2 = mxGetScalar(prhs[4]) != 0;
        if (level < 1) mexErrMsgIdAndTxt(ERR_ID, "level must be at least 1");
        plan = plan_for_signal(prhs[1], prhs[2], level, l2, read_dilation(nrhs, prhs, 7), nrhs > 8 ? (int)mxGetScalar(prhs[8]) : 0, &ndim, dims);
        for (a = 0; a < ndim; ++a) od[a] = (mwSize)dims[a];
        ond = (mwSize)ndim;
        if (ond == 1) od[ond++] = 1;
        plhs[0] = mxCreateUninitNumericArray(ond, od, mxIsSingle(prhs[1]) ? mxSINGLE_CLASS : mxDOUBLE_CLASS, mxIsComplex(prhs[1]) ? mxCOMPLEX : mxREAL);
        if (ndwt_denoise_host(plan, mxGetData(prhs[1]), mxGetData(plhs[0]), level, mxGetScalar(prhs[5]), read_mode(nrhs, prhs, 6)) != NDWT_OK) fail("denoise");
        return 1;
    }
    mexErrMsgIdAndTxt(ERR_ID, "unknown command '%s'", cmd);
    return 1;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    const mxArray* x;
    const mwSize* d;
    mwSize nd, od[NDWT_MAX_DIMS + 1], ond;
    int inverse, level, l2, dilation = NDWT_DILATION_REFERENCE, ndim, a, dtype, cplx, rc, ndev = 0, devices[64];
    int64_t dims[NDWT_MAX_DIMS];
    char names[NDWT_MAX_DIMS][16];
    ndwt_plan* plan;

    if (command(nlhs, plhs, nrhs, prhs)) return;
    if (nrhs < 5) mexErrMsgIdAndTxt(ERR_ID, "Five Inputs Required");             /* nd_dwt_mex.c:19-21 */
    if (nlhs > 1) mexErrMsgIdAndTxt(ERR_ID, "Too many output arguments.");        /* :23-25 */
    x = prhs[0];
    if (!mxIsDouble(x) && !mxIsSingle(x)) mexErrMsgIdAndTxt(ERR_ID, "Arrays must be double or single");
    inverse = mxGetScalar(prhs[2]) != 0;
    level = (int)mxGetScalar(prhs[3]);
    l2 = mxGetScalar(prhs[4]) != 0;
    if (level < 1) mexErrMsgIdAndTxt(ERR_ID, "level must be at least 1");
    if (nrhs > 5) {
        char buf[16];
        mxGetString(prhs[5], buf, sizeof buf);
        if (!strcmp(buf, "atrous")) dilation = NDWT_DILATION_ATROUS;
    }

    if (nrhs > 6 && mxGetNumberOfElements(prhs[6]) > 0) {   /* devices: doubles holding device ordinals */
        const double* dv = (const double*)mxGetData(prhs[6]);
        ndev = (int)mxGetNumberOfElements(prhs[6]);
        if (!mxIsDouble(prhs[6]) || ndev > 64) mexErrMsgIdAndTxt(ERR_ID, "devices: a double vector of at most 64 device ordinals");
        for (a = 0; a < ndev; ++a) devices[a] = (int)dv[a];
    }

    /* dims: column vectors are 1-D like nd_dwt_mex.c:68-70; the inverse input carries the band axis last (:115) */
    nd = mxGetNumberOfDimensions(x);
    d = mxGetDimensions(x);
    ndim = (int)nd - (inverse ? 1 : 0);
    if (!inverse && nd == 2 && d[1] == 1) ndim = 1;
    if (inverse && nd == 2) ndim = 1;
    if (ndim < 1 || ndim > NDWT_MAX_DIMS) mexErrMsgIdAndTxt(ERR_ID, "1 to 4 dimensions supported");
    for (a = 0; a < ndim; ++a) dims[a] = (int64_t)d[a];
    /* the size-consistency guard of nd_dwt_mex.c:124-127: the band axis of an inverse input must hold exactly the bands of
     * `level` levels -- ndwt_rec_host reads prod(dims) * bands elements from it */
    if (inverse) {
        const int64_t have = (int)nd > ndim ? (int64_t)d[ndim] : 1;
        if (have != ndwt_num_bands(ndim, level)) mexErrMsgIdAndTxt(ERR_ID, "FIlter size and image size not consistant");
    }

    /* wavelet names: one per axis, or one for all */
    for (a = 0; a < ndim; ++a) {
        if (mxIsCell(prhs[1])) {
            const int ncell = (int)mxGetNumberOfElements(prhs[1]);
            if (ncell != ndim && ncell != 1) mexErrMsgIdAndTxt(ERR_ID, "one wavelet name per dimension, or a single name");
            if (mxGetString(mxGetCell(prhs[1], ncell == 1 ? 0 : a), names[a], 16)) mexErrMsgIdAndTxt(ERR_ID, "bad wavelet name");
        } else if (mxGetString(prhs[1], names[a], 16)) {
            mexErrMsgIdAndTxt(ERR_ID, "bad wavelet name");
        }
    }

    dtype = mxIsSingle(x) ? NDWT_F32 : NDWT_F64;
#if MX_HAS_INTERLEAVED_COMPLEX
    cplx = mxIsComplex(x) ? NDWT_COMPLEX_INTERLEAVED : NDWT_REAL;
#else
    cplx = NDWT_REAL;                                    /* split storage: one real transform per part */
#endif
    /* no `devices`: device 0; ONE ordinal: the single-device plan on that device; several: the multi-device plan */
    plan = ndev > 1 ? NULL : get_plan(ndim, dims, names, dtype, cplx, l2, dilation, level, ndev == 1 ? devices[0] : 0);

    /* output: MATLAB-owned like the reference's (nd_dwt_mex.c:86,136), but NOT zero-filled: every element is written by the
     * transform (the reference needs the zeros, nddwt.c:168-173 accumulates into them; 11.8 GB of memset for a 512^3 3-level dec) */
    for (a = 0; a < ndim; ++a) od[a] = (mwSize)dims[a];
    ond = (mwSize)ndim;
    if (!inverse) od[ond++] = (mwSize)ndwt_num_bands(ndim, level);
    if (ond == 1) od[ond++] = 1;
    plhs[0] = mxCreateUninitNumericArray(ond, od, mxIsSingle(x) ? mxSINGLE_CLASS : mxDOUBLE_CLASS, mxIsComplex(x) ? mxCOMPLEX : mxREAL);

    if (ndev > 1) {   /* sharded over the listed devices by the single-process multi-device plan */
        ndwt_mplan* mp = get_mplan(ndim, dims, names, dtype, cplx, l2, dilation, level, devices, ndev);
        if (nrhs > 7) {
            char buf[16];
            mxGetString(prhs[7], buf, sizeof buf);
            if (ndwt_mplan_set_exchange(mp, !strcmp(buf, "gather") ? NDWT_EXCHANGE_GATHER : NDWT_EXCHANGE_SCATTER) != NDWT_OK)
                mexErrMsgIdAndTxt(ERR_ID, "exchange: %s", ndwt_mplan_last_error());
        }
        rc = inverse ? ndwt_mrec_host(mp, mxGetData(x), mxGetData(plhs[0]), level) : ndwt_mdec_host(mp, mxGetData(x), mxGetData(plhs[0]), level);
#if !MX_HAS_INTERLEAVED_COMPLEX
        if (rc == NDWT_OK && mxIsComplex(x))          /* split storage: the imaginary part is a second real transform */
            rc = inverse ? ndwt_mrec_host(mp, mxGetImagData(x), mxGetImagData(plhs[0]), level)
                         : ndwt_mdec_host(mp, mxGetImagData(x), mxGetImagData(plhs[0]), level);
#endif
        if (rc != NDWT_OK) mexErrMsgIdAndTxt(ERR_ID, "%s: %s", inverse ? "rec" : "dec", ndwt_mplan_last_error());
        return;
    }
#if MX_HAS_INTERLEAVED_COMPLEX
    rc = inverse ? ndwt_rec_host(plan, mxGetData(x), mxGetData(plhs[0]), level)
                 : ndwt_dec_host(plan, mxGetData(x), mxGetData(plhs[0]), level);
#else
    {
        const void* x_im = mxIsComplex(x) ? mxGetImagData(x) : NULL;
        void* y_im = mxIsComplex(x) ? mxGetImagData(plhs[0]) : NULL;
        rc = inverse ? ndwt_rec_split_host(plan, mxGetData(x), x_im, mxGetData(plhs[0]), y_im, level)
                     : ndwt_dec_split_host(plan, mxGetData(x), x_im, mxGetData(plhs[0]), y_im, level);
    }
#endif
    if (rc != NDWT_OK) fail(inverse ? "rec" : "dec");
}
