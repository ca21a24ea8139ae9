/*
 * ndwt.h -- C ABI of the MI355X-native non-decimated wavelet transform engine (libndwt_hip.so).
 *
 * This is the drop-in boundary for the reference's native hot path:
 *   reference mex/nddwt.h:13-32   nd_dwt_dec / nd_dwt_rec / nd_dwt_{dec,rec}_1level / pointByPoint /
 *                                 init_fftw_plan      (the FFT-domain core, complex fp64 only)
 *   reference mex/nd_dwt_mex.c:8  mexFunction         (the MATLAB gateway the classes call:
 *                                 nd_dwt_1D.m:158,220  nd_dwt_2D.m:160,222  nd_dwt_3D.m:161,225
 *                                 nd_dwt_4D.m:158,219)
 *
 * The reference hands FFT-domain data (x_f, f_dec) across that boundary; this engine takes the
 * SIGNAL-domain array and a plan that carries the per-axis Daubechies taps, and computes the same
 * coefficients with direct periodic filter banks in hand-written HIP kernels (gfx950).  Plain C types
 * only: pointers, sizes, int status codes.  No torch, no C++ types.
 *
 * Layout (identical to the reference, nd_dwt_mex.c:72-83): column-major, dims[0] fastest; coefficient
 * arrays are band-planar with the band axis last, bands = 2^d + (2^d-1)(level-1); band 0 is the coarsest
 * approximation, the last 2^d-1 bands are the level-1 details; inside a level band b uses the high-pass
 * on axis a iff bit a of b is set (Functions/nd_dwt_3D.m:45-52,334-341).
 *
 * Threading: a plan may be used by one host thread at a time; different plans are independent.
 * Every compute entry point is asynchronous on the given HIP stream unless it takes host pointers.
 */
#ifndef NDWT_H
#define NDWT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDWT_MAX_DIMS 4
#define NDWT_MAX_TAPS 20 /* db10 */

typedef struct ndwt_plan ndwt_plan; /* opaque */

/* status codes (the reference core returns void and cannot fail cleanly; nd_dwt_mex.c raises
 * "MATLAB:FFT2mx:invalidNumInputs" for every error -- the mex shim maps these codes onto that) */
enum {
    NDWT_OK = 0,
    NDWT_ERR_INVALID_ARG = 1,  /* bad ndim / dims / level / null pointer */
    NDWT_ERR_UNKNOWN_WAVELET = 2, /* wave_filters.m:159 "Unknown Wavelet Name" */
    NDWT_ERR_FILTER_TOO_LONG = 3, /* nd_dwt_3D.m:277-286 "... Dimension of Data is shorter than the wavelet filter being used" */
    NDWT_ERR_NO_DEVICE = 4,    /* no HIP device / HIP runtime failure at plan creation */
    NDWT_ERR_HIP = 5,          /* a HIP call failed; see ndwt_last_error() */
    NDWT_ERR_ALLOC = 6,
    NDWT_ERR_UNSUPPORTED = 7
};

enum { NDWT_F32 = 0, NDWT_F64 = 1 };                 /* 'precision' = single / double (nd_dwt_3D.m:130-132) */
enum { NDWT_REAL = 0, NDWT_COMPLEX_INTERLEAVED = 1 }; /* split complex: call twice (re, im) -- the filters are real */
enum { NDWT_DILATION_REFERENCE = 0, /* same un-dilated taps at every level: what the reference computes (nd_dwt_3D.m:178-186, nddwt.c:214-234) */
       NDWT_DILATION_ATROUS = 1 };   /* tap stride 2^(level-1): textbook stationary wavelet transform */
enum { NDWT_PATH_AUTO = 0,    /* fused kernels where they apply, per-axis kernels otherwise */
       NDWT_PATH_GENERIC = 1  /* force the per-axis kernels (used by the parity tests to cross-check the fused ones) */ };

/* ---- filters: Functions/wave_filters.m:1-174 ------------------------------------------------------
 * wname = "db1" .. "db10" (case-insensitive).  Writes LO_D and HI_D (length *len = 2K) exactly as the
 * reference returns them: LO_D[m] = h[L-1-m], HI_D[m] = (-1)^(m+1) h[m].  Buffers hold NDWT_MAX_TAPS. */
int ndwt_wave_filters(const char* wname, double* lo_d, double* hi_d, int* len);

/* ---- band bookkeeping -------------------------------------------------------------------------------
 * nd_dwt_mex.c:83 and the level inference of rec() (nd_dwt_1D.m:213, 2D:215, 3D:217, 4D:213). */
int64_t ndwt_num_bands(int ndim, int level);
int ndwt_level_from_bands(int ndim, int64_t bands); /* returns level >= 1, or -1 if `bands` is not a valid count */

/* ---- plan: replaces the class constructor's get_filters() (nd_dwt_3D.m:263-342) and the per-call
 *      FFTW planning of nddwt.c:110-111 ----------------------------------------------------------------
 * ndim 1..4; dims[0] fastest; wnames[a] per axis; dtype NDWT_F32/F64; complexity NDWT_REAL or
 * NDWT_COMPLEX_INTERLEAVED; pres_l2_norm as in the classes; dilation NDWT_DILATION_*; max_level = the
 * largest level dec/rec will be asked for (sizes the scratch); device = HIP device ordinal. */
int ndwt_plan_create(ndwt_plan** plan, int ndim, const int64_t* dims, const char* const* wnames, int dtype,
                     int complexity, int pres_l2_norm, int dilation, int max_level, int device);
/* Plan for ONE SLAB of a volume sharded on its outermost axis (the *_slab entry points below): dims describe the local slab,
 * global_outer is the length of the sharded axis of the whole volume.  The reference's length check (nd_dwt_3D.m:277-286:
 * filter <= axis) applies to the WHOLE axis: a slab may be thinner than the filter (cfg5: 32 frames over 8 ranks = 4, db4 = 8
 * taps) -- the slab kernels read halo planes and never wrap.  ndwt_dec / ndwt_rec (periodic on every axis) refuse such a plan
 * when the local length is shorter than the filter. */
int ndwt_plan_create_slab(ndwt_plan** plan, int ndim, const int64_t* dims_local, int64_t global_outer, const char* const* wnames,
                          int dtype, int complexity, int pres_l2_norm, int dilation, int max_level, int device);
/* A plan owns scratch (the approximation ping-pong between levels, temporaries): use it from ONE host thread and ONE stream
 * at a time.  Calls on different streams must be ordered by the caller (an event), or use one plan per stream. */
int ndwt_plan_destroy(ndwt_plan* plan);
int ndwt_plan_set_path(ndwt_plan* plan, int path); /* NDWT_PATH_* */
/* which kernels a level of this plan runs: writes a short static string such as "fused3d", "fused2d",
 * "axis" into buf (for tests and logs) */
int ndwt_plan_describe(const ndwt_plan* plan, char* buf, int buflen);
/* tuning hook (tests, benchmarks): workgroup count the fused kernels aim for and/or a forced number of
 * outer-axis planes per workgroup; 0 restores the default.  Results never depend on it. */
int ndwt_plan_set_tuning(ndwt_plan* plan, int target_blocks, int force_zchunk);
/* test / tuning hook of tools/ (interleaved A/B runs): kernel variant per direction, marched chunk per direction, fp64 on the
 * fused (1) or per-axis (0) kernels.  Negative = leave unchanged.  Every variant computes the same transform (bit-identical where
 * only the schedule differs; to rounding where the order of the FMAs does: packed / scalar forms, kernel families); the library never
 * reads the environment. */
int ndwt_plan_set_variant(ndwt_plan* plan, int variant_fwd, int variant_inv, int zchunk_fwd, int zchunk_inv, int fp64_fused);
/* test / tuning hook: 0 makes ndwt_denoise keep the level-1 detail bands in memory (dec, thresholding fused into the synthesis
 * loads, rec); 1 (default) = level 1 in one launch that recomputes them where that is faster (float, real, 3-D, one tap length
 * <= 6 on every axis); 2 = also with 8 taps (compute-bound there: 3 % slower than the materialising path at 512^3). */
int ndwt_plan_set_fused_level1(ndwt_plan* plan, int enable);
/* per-kernel timing with HIP events recorded on the launch stream around every kernel this plan launches
 * (what bench.py's roofline line is computed from).  ndwt_plan_get_profile() synchronises the device, sums and
 * clears the records of one kernel kind. */
enum { NDWT_KERNEL_FUSED_ANALYSIS = 0, NDWT_KERNEL_FUSED_SYNTHESIS = 1, NDWT_KERNEL_AXIS_ANALYSIS = 2, NDWT_KERNEL_AXIS_SYNTHESIS = 3 };
int ndwt_plan_set_profiling(ndwt_plan* plan, int enable);
int ndwt_plan_get_profile(ndwt_plan* plan, int kind, double* total_ms, int64_t* launches);

/* ---- the transform: replaces nd_dwt_dec / nd_dwt_rec (nddwt.c:189-292) and the 1-level forms
 *      (nddwt.c:98-186) -----------------------------------------------------------------------------------
 * x: prod(dims) elements (x2 scalars if complex); y: prod(dims)*ndwt_num_bands(ndim, level) elements.
 * Device pointers, caller-allocated, inputs are never modified (the reference's rec overwrites its
 * input, nddwt.c:163).  stream is a hipStream_t passed as void* (NULL = the null stream). */
int ndwt_dec(ndwt_plan* plan, const void* x_dev, void* y_dev, int level, void* stream);
int ndwt_rec(ndwt_plan* plan, const void* y_dev, void* x_dev, int level, void* stream);
/* The same with a PITCHED coefficient array: band b starts at y_dev + b * band_pitch elements (band_pitch >= prod(dims);
 * 0 = packed, i.e. ndwt_dec / ndwt_rec).  The packed layout is the reference's (a MATLAB array [dims, bands]); callers that
 * own the coefficient buffer (iterative solvers, the Python classes with 'band_pitch', ndwt_denoise's scratch) gain 4 - 12 %
 * on the synthesis of power-of-two volumes by keeping the 2^d band streams a few hundred bytes off a power-of-two distance:
 * with every band at the same address modulo 1 KiB the fabric traffic of the 512^3 synthesis is 1.23x its algorithmic bytes
 * (1.44x before the library skewed the one band it owns, the approximation scratch), with band_pitch = ndwt_band_pitch(plan)
 * (prod(dims) + 256 bytes) 1.08x.  Results are identical. */
int ndwt_dec_pitched(ndwt_plan* plan, const void* x_dev, void* y_dev, int64_t band_pitch, int level, void* stream);
int ndwt_rec_pitched(ndwt_plan* plan, const void* y_dev, int64_t band_pitch, void* x_dev, int level, void* stream);
int64_t ndwt_band_pitch(const ndwt_plan* plan);   /* the recommended pitch in elements */

/* Host-pointer forms for the mex shim (stage through device memory, synchronous). */
int ndwt_dec_host(ndwt_plan* plan, const void* x_host, void* y_host, int level);
int ndwt_rec_host(ndwt_plan* plan, const void* y_host, void* x_host, int level);

/* Staging of the host-pointer forms is owned by the plan, grown on demand and kept across calls (a 512^3 3-level transform stages
 * 12.3 GB: allocating and freeing it per call costs milliseconds, and the gateway is called thousands of times by an iterative
 * solver, README.md:2).  ndwt_plan_release_staging gives the memory back (the next host-pointer call allocates again). */
int ndwt_plan_release_staging(ndwt_plan* plan);

/* ---- device-resident coefficients ------------------------------------------------------------------------
 * The reference's gateway moves the whole coefficient array through host memory on every call (y = nd_dwt_mex(...) at
 * nd_dwt_3D.m:161, x = nd_dwt_mex(y, ...) at :225): 11.8 GB per direction for a 512^3 3-level transform, 0.2 s over PCIe against
 * 3 ms of compute.  A solver that only needs rec(shrink(dec(x))) -- or that keeps y between iterations -- can leave the coefficients
 * on the device behind an opaque handle: only the signal crosses the link.  The handle is bound to the plan that made it (same
 * dims / wavelets / precision); it owns a pitched coefficient array (ndwt_band_pitch) and must be released before its plan is
 * destroyed (ndwt_plan_destroy refuses while handles of the plan are alive).  All calls are synchronous on the null stream, like the other host-pointer forms. */
typedef struct ndwt_coef ndwt_coef;
int ndwt_coef_create(ndwt_plan* plan, int level, ndwt_coef** coef);                       /* uninitialised coefficients of `level` levels */
int ndwt_coef_release(ndwt_coef* coef);
int ndwt_coef_info(const ndwt_coef* coef, int* level, int64_t* bands, int64_t* band_pitch, void** dev_ptr);
/* dec: x_host -> coefficients on the device.  *coef == NULL: a new handle is returned; else that handle (same plan and level) is refilled */
int ndwt_coef_dec_host(ndwt_plan* plan, const void* x_host, int level, ndwt_coef** coef);
int ndwt_coef_rec_host(ndwt_plan* plan, const ndwt_coef* coef, void* x_host);             /* rec: coefficients on the device -> x_host */
int ndwt_coef_shrink(ndwt_plan* plan, ndwt_coef* coef, double threshold, int mode);       /* NDWT_SHRINK_* of the detail bands, in place */
/* the whole array in the reference's packed layout [dims, bands] to / from host memory (for callers that do need y) */
int ndwt_coef_get_host(ndwt_plan* plan, const ndwt_coef* coef, void* y_host);
int ndwt_coef_put_host(ndwt_plan* plan, int level, const void* y_host, ndwt_coef** coef);

/* ---- consumers for iterative solvers (extension; the reference's users threshold the bands in MATLAB, README.md:2) ----
 * ndwt_shrink: in-place soft / hard thresholding of every detail band of a level-`level` coefficient array (band 0, the
 *   coarsest approximation, is kept); complex data: the magnitude is shrunk, the phase kept.
 * ndwt_denoise: dec -> shrink -> rec in one call; the coefficients live in a scratch array owned by the plan, so only
 *   the signal crosses the boundary (x and out may be the same buffer).  The host form moves 2 x prod(dims) elements
 *   over PCIe instead of 2 x prod(dims) x bands.  Float, real, 3-D plans with one tap length <= 6 on every axis and
 *   16-byte aligned, non-overlapping x / out run the FINEST LEVEL WITHOUT ITS DETAIL BANDS IN MEMORY: one launch recomputes
 *   them from x, thresholds them in registers and reconstructs (5 volumes moved at level 1 instead of 18); the values are
 *   those of dec -> shrink -> rec up to fp32 rounding (the recomputed coefficients are not rounded through memory).
 *   x == out keeps the materialising path. */
enum { NDWT_SHRINK_SOFT = 0, NDWT_SHRINK_HARD = 1 };
int ndwt_shrink(ndwt_plan* plan, void* y_dev, int level, double threshold, int mode, void* stream);
int ndwt_shrink_pitched(ndwt_plan* plan, void* y_dev, int64_t band_pitch, int level, double threshold, int mode, void* stream);
int ndwt_denoise(ndwt_plan* plan, const void* x_dev, void* out_dev, int level, double threshold, int mode, void* stream);
int ndwt_denoise_host(ndwt_plan* plan, const void* x_host, void* out_host, int level, double threshold, int mode);

/* Split complex: separate real / imaginary arrays, the layout the reference's gateway receives from MATLAB
 * (mxGetPr / mxGetPi, nd_dwt_mex.c:55-58,90-93) and hands to nd_dwt_dec / nd_dwt_rec (outR/outI, imageR/imageI,
 * nddwt.h:13-20).  The plan must be NDWT_REAL (real filters: each part is transformed on its own); the imaginary
 * pointers may both be NULL (real data).  Device-pointer and host-pointer forms. */
int ndwt_dec_split(ndwt_plan* plan, const void* x_re, const void* x_im, void* y_re, void* y_im, int level, void* stream);
int ndwt_rec_split(ndwt_plan* plan, const void* y_re, const void* y_im, void* x_re, void* x_im, int level, void* stream);
int ndwt_dec_split_host(ndwt_plan* plan, const void* x_re, const void* x_im, void* y_re, void* y_im, int level);
int ndwt_rec_split_host(ndwt_plan* plan, const void* y_re, const void* y_im, void* x_re, void* x_im, int level);

/* ---- one level on a slab of the outermost axis (multi-GPU building block; no reference counterpart:
 *      the reference is single-process) ------------------------------------------------
 * The plan's dims describe the LOCAL slab (dims[ndim-1] = local planes).  `stride` is the tap stride of
 * this level (1 in reference mode).  Analysis: `in` holds halo_before + local + halo_after planes of the
 * approximation band, halo_before = (L/2-1)*stride, halo_after = (L/2)*stride for the outer axis' filter
 * length L; the 2^d outputs are local-sized.  Synthesis: every one of the 2^d inputs holds
 * halo_before = (L/2)*stride and halo_after = (L/2-1)*stride planes around the local slab.
 * ndwt_slab_halo() reports those four numbers. */
/* 1 when the plan offers the slab forms that read a slab in place (split-halo analysis, runs of the zero-extended synthesis:
 * a 3-D plan on the fused kernels whose outer-axis filter is its longest), else 0 */
int ndwt_plan_slab_fast(const ndwt_plan* plan);
int ndwt_slab_halo(const ndwt_plan* plan, int stride, int64_t* ana_before, int64_t* ana_after,
                   int64_t* syn_before, int64_t* syn_after);
int ndwt_analysis_level_slab(ndwt_plan* plan, const void* in_with_halo, void* const* out_bands, int stride, void* stream);
int ndwt_synthesis_level_slab(ndwt_plan* plan, const void* const* in_bands_with_halo, void* out, int stride, void* stream);
/* Copy-free forms for fused 3-D plans (NDWT_ERR_UNSUPPORTED otherwise; the outer-axis filter must be the longest):
 * analysis reads the local slab and the two halo buffers (as received from the neighbours) from separate pointers;
 * synthesis treats the local coefficient slab as zero outside and writes syn_after + local + syn_before planes:
 * the local result plus the partial sums owed to the neighbouring slabs (they are sent there and added -- 1 band of
 * exchange instead of the halo of all 2^d bands).  The zero-extended synthesis also takes 4-D plans sharded on t (16 bands:
 * the 3-D part runs per frame, the t-axis pass over zero-padded frames). */
int ndwt_analysis_level_slab_split(ndwt_plan* plan, const void* in_local, const void* halo_before, const void* halo_after,
                                   void* const* out_bands, int stride, void* stream);
int ndwt_synthesis_level_slab_ext(ndwt_plan* plan, const void* const* in_bands_local, void* out_ext, int stride, void* stream);

/* Runs of planes of the two calls above, so that the exchange can overlap with the planes that do not wait for it.
 * analysis_part: n_planes output planes; in_local points at the first of them, halo_before at the (L/2-1) planes
 *   before it and halo_after at the (L/2) planes after the run -- each may point into the slab itself or into a
 *   received buffer; out[b] points at the first output plane of band b.
 * synthesis_part: output planes [e0, e0 + n_out) of the zero-extended synthesis of n_in coefficient planes
 *   (in_local[b] = first plane of band b; result plane e sums coefficient planes e-(L-1)..e, so planes
 *   [0, L/2-1) and [n_in + L/2-1, n_in + L-1) are the partial sums owed to the neighbours). */
int ndwt_analysis_level_slab_part(ndwt_plan* plan, const void* in_local, const void* halo_before, const void* halo_after,
                                  void* const* out_bands, int stride, int64_t n_planes, void* stream);
int ndwt_synthesis_level_slab_part(ndwt_plan* plan, const void* const* in_local, int64_t n_in, int64_t e0, int64_t n_out,
                                   void* out, int stride, void* stream);
/* Several equal runs in one launch (the two ends of a slab):
 * analysis_runs: the input carries its halo contiguously (as for ndwt_analysis_level_slab); run r reads from
 *   in_with_halo + r*run_stride planes and writes n_planes planes at out[b] + r*run_stride planes.
 * synthesis_runs: run r = planes [e0 + r*e_stride, +n_out) of the zero-extended result, written to out + r*n_out planes. */
int ndwt_analysis_level_slab_runs(ndwt_plan* plan, const void* in_with_halo, void* const* out_bands, int stride, int64_t n_planes,
                                  int64_t n_runs, int64_t run_stride, void* stream);
int ndwt_synthesis_level_slab_runs(ndwt_plan* plan, const void* const* in_local, int64_t n_in, int64_t e0, int64_t e_stride,
                                   int64_t n_runs, int64_t n_out, void* out, int stride, void* stream);

/* Up to NDWT_MAX_SEGMENTS runs of planes copied (op NDWT_SEG_COPY) or added (NDWT_SEG_ADD: dst[i] += src[i]) in ONE launch on
 * `stream`, on the plan's device: the halo planes a slab takes from itself, and the partial sums received from the two neighbours
 * added to the slab's edge planes -- a launch costs about 6 us on MI355X whatever it moves, and a level has two such runs.
 * count[i] = elements (of the plan's scalar type; complex data: 2 per element) of run i; runs must not overlap each other. */
#define NDWT_MAX_SEGMENTS 8
#define NDWT_SEG_COPY 0
#define NDWT_SEG_ADD 1
int ndwt_slab_segments(ndwt_plan* plan, int op, int nseg, void* const* dst, const void* const* src, const int64_t* count, void* stream);

/* ---- single-process multi-device plan (hard part 5; section 8b `devices[]`) ---------------------------
 * The reference's host is ONE process calling one gateway (nd_dwt_3D.m:161,225): this is the multi-GPU path that fits behind
 * that call.  The volume is sharded in slabs on its outermost axis over devices[0..ndev) (a device may appear more than once:
 * independent slabs and streams on one GPU); one host thread drives all of them, planes move between slabs with asynchronous
 * device-to-device (peer, xGMI) copies ordered by events.  ndim 2..4.  Errors: ndwt_mplan_last_error().
 *   ndwt_mdec / ndwt_mrec: DEVICE-RESIDENT data, one pointer per slab (ndwt_mplan_slab gives its device and planes): x_slabs[i] =
 *     the n_i planes of the signal on slab i's device, y_slabs[i] = its coefficient slab, band b at b * n_i planes.  The kernels
 *     read and write these buffers in place.  The data must be complete when the call is made; the call returns when the result is.
 *   ndwt_mdec_host / ndwt_mrec_host: whole-volume host arrays (the layout of ndwt_dec_host / ndwt_rec_host), staged through
 *     slab buffers of the plan.
 *   Exchange per level: analysis -- halo planes of the approximation band (bit-identical to one device).  Synthesis --
 *     NDWT_EXCHANGE_SCATTER (default where every level runs the fused 3-D kernels): each slab reconstructs its own coefficients
 *     zero-extended and sends ONE band of partial sums for the L-1 planes around it, which their owners add in a fixed order
 *     (equal to one device to rounding); NDWT_EXCHANGE_GATHER: halo planes of all 2^d bands assembled with a copy of the slab
 *     (bit-identical to one device; what plans on the per-axis kernels, dilated levels, 2-D and 4-D plans use anyway). */
typedef struct ndwt_mplan ndwt_mplan;
enum { NDWT_EXCHANGE_SCATTER = 0, NDWT_EXCHANGE_GATHER = 1 };
int ndwt_mplan_create(ndwt_mplan** plan, int ndim, const int64_t* dims, const char* const* wnames, int dtype, int complexity,
                      int pres_l2_norm, int dilation, int max_level, const int* devices, int ndev);
int ndwt_mplan_destroy(ndwt_mplan* plan);
int ndwt_mplan_num_slabs(const ndwt_mplan* plan);
int ndwt_mplan_slab(const ndwt_mplan* plan, int idx, int* device, int64_t* first_plane, int64_t* planes);
int ndwt_mplan_set_exchange(ndwt_mplan* plan, int exchange);
/* 1 (default): the planes moved between slabs travel on per-slab copy streams while the launches that do not need them run (fused 3-D
 * plans at tap stride 1 whose slabs are thicker than twice the halo; the interior planes first, then the ends / the partial sums first,
 * then the slab's own planes); 0: exchange, then one launch per slab and level.  Same results either way.  (2: like 1, with the partial
 * sums staged through the receive buffers even between slabs that share a device -- how the tests run that path on one GPU.) */
int ndwt_mplan_set_overlap(ndwt_mplan* plan, int overlap);
int ndwt_mplan_describe(const ndwt_mplan* plan, char* buf, int buflen);   /* slabs, exchange schemes, peer-access findings */
/* 1 (default with more than one slab): one host thread per slab queues that slab's work (persistent workers of the plan: asleep between
 * calls; the calling thread drives slab 0 and returns when every device has finished, as before); 0: the calling thread queues everything.
 * Same work on the same streams in the same per-stream order: identical results.  Queueing a 512^3 db4 dec + rec for 8 slabs from one
 * thread takes 1.7 ms of host time -- more than one of 8 devices needs to compute its share. */
int ndwt_mplan_set_threads(ndwt_mplan* plan, int threads);
/* diagnostic: host microseconds the last ndwt_mdec / ndwt_mrec spent QUEUEING work for all slabs (one host thread drives every device: with
 * many devices this, not the devices, can bound a call); -1 without a plan */
double ndwt_mplan_last_enqueue_us(const ndwt_mplan* plan);
int ndwt_mdec(ndwt_mplan* plan, const void* const* x_slabs, void* const* y_slabs, int level);
int ndwt_mrec(ndwt_mplan* plan, const void* const* y_slabs, void* const* x_slabs, int level);
int ndwt_mdec_host(ndwt_mplan* plan, const void* x_host, void* y_host, int level);
int ndwt_mrec_host(ndwt_mplan* plan, const void* y_host, void* x_host, int level);
const char* ndwt_mplan_last_error(void);

/* ---- exchange of the one-process-per-GPU driver: RCCL point-to-point calls on the caller's stream ------------------------------------
 * (ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd between ring neighbours.)  torch.distributed's
 * batch_isend_irecv runs its RCCL work on a stream of its own: every exchange then pays two cross-stream dependencies (40 + 14 us around a
 * 15-us RCCL kernel on one MI355X, six times per dec + rec step of 0.83 ms); enqueued on the stream that runs the transform, the same grouped
 * send / receive is one more kernel in stream order.  librccl is opened at run time (no link-time dependency).
 * Rank 0 obtains a 128-byte id (ndwt_comm_unique_id) and hands it to every rank by whatever means the host has (the Python driver
 * broadcasts it through torch.distributed); every rank then calls ndwt_comm_create (collective).  ndwt_comm_exchange enqueues one group of
 * sends / receives: op i moves bytes[i] bytes at ptrs[i] to (is_send[i] != 0) or from rank peers[i]; between two ranks, sends and receives are
 * matched in the order they are listed.  Stream-ordered: the buffers may be reused by later work on `stream` without further waiting. */
typedef struct ndwt_comm ndwt_comm;
int ndwt_comm_unique_id(void* id128);
int ndwt_comm_create(ndwt_comm** comm, const void* id128, int nranks, int rank, int device);
int ndwt_comm_destroy(ndwt_comm* comm);
int ndwt_comm_exchange(ndwt_comm* comm, int nops, const int* is_send, void* const* ptrs, const int64_t* bytes, const int* peers, void* stream);
const char* ndwt_comm_last_error(void);

/* ---- errors ------------------------------------------------------------------------------------------ */
const char* ndwt_last_error(void); /* thread-local message of the last failing call */
const char* ndwt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NDWT_H */
