// ndwt_api.hip -- plan, level loop, axis kernels and the C ABI of libndwt_hip.so (include/ndwt.h).
//
// Replaces, for the hot path, reference mex/nddwt.c (nd_dwt_dec :189-239, nd_dwt_rec :242-292 and
// their 1-level forms :98-186) and the gateway mex/nd_dwt_mex.c:8-153.  No CPU fallback exists in
// this library: every entry point runs HIP kernels or returns an error code.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ndwt.h"
#include "ndwt_device.h"
#include "ndwt_filters.h"
#include "ndwt_fused.h"
#include "ndwt_geom.h"

using namespace ndwt;

// ------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(NDWT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------ axis kernels
template <typename T>
__global__ __launch_bounds__(256) void axis_analysis_kernel(const T* __restrict__ in, T* __restrict__ lo, T* __restrict__ hi,
                                                            const AxisTaps<T> tp, const AxisArgs<T> a) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; idx < a.total; idx += step) axis_analysis_elem(idx, in, lo, hi, tp, a);
}

template <typename T>
__global__ __launch_bounds__(256) void axis_synthesis_kernel(const T* __restrict__ ain, const T* __restrict__ din,
                                                             T* __restrict__ out, const AxisTaps<T> tp, const AxisArgs<T> a) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; idx < a.total; idx += step) axis_synthesis_elem(idx, ain, din, out, tp, a);
}

// -------------------------------------------------------------------------------------------- plan
struct ProfRec {
    int kind;                          // NDWT_KERNEL_*
    hipEvent_t start, stop;
};

#ifdef NDWT_NO_APPROX_SKEW
static constexpr size_t kApproxSkew = 0;
#else
static constexpr size_t kApproxSkew = 256;
#endif
struct ndwt_plan {
    int ndim;
    long long dims[NDWT_MAX_DIMS];
    int order[NDWT_MAX_DIMS];          // K of dbK per axis
    AxisFilter filt[NDWT_MAX_DIMS];
    int dtype, complexity, l2, dilation, max_level, device, path;
    size_t esize;                      // bytes per scalar
    long long comp;                    // scalars per element (2 for interleaved complex)
    long long vol;                     // scalars per band
    void* approx[2];                   // approximation ping-pong between levels: approx_base[i] + kApproxSkew bytes
    void* approx_base[2];
    void* tmp;                         // temporaries of the per-axis path / 4-D split
    size_t tmp_bytes;
    int target_blocks;                 // fused-kernel grid sizing: 0 = one round of resident workgroups (per kernel), else as given
    int force_zchunk;
    int zchunk_dir[2];                 // per-direction override of the marched chunk: [0] analysis, [1] synthesis (0 = auto)
    int variant_fwd, variant_inv;      // fused-kernel variants (tuning experiments; same results)
    int num_cus;
    int fp64_fused;                    // fp64: fused 3-D kernels (1) or the per-axis march kernels (0)
    void* taps_dev[2];                 // device tap tables of the fused kernels: [0] analysis, [1] synthesis (Taps3<T, Lp>)
    int shrink_mode;                   // ndwt_denoise, during its rec: 0 none, 1 soft, 2 hard -- fused into the synthesis kernels' loads
    double shrink_thr;
    void* coef;                        // coefficient scratch of ndwt_denoise (all bands of the last level used), lazily allocated
    size_t coef_bytes;
    void* taps_den;                    // device tap table of the fused level-1 denoising kernel (TapsDen<float, L>), built at plan creation where it applies
    void* den_a1;                      // ndwt_denoise, fused level 1: the level-1 approximation / its reconstruction -- a scratch of its own (lazily
                                       // allocated), not `tmp`: dec_impl / rec_impl run in between and may re-allocate that one
    int fused_level1;                  // ndwt_denoise: 1 (default) level 1 in one launch where that is faster (tap lengths <= 6), 2 wherever the
                                       // kernel exists (8 taps too: compute-bound there, +3 %), 0 never (the level-1 detail bands stay in memory)
    // optional per-kernel timing with HIP events on the launch stream (bench.py's roofline figures)
    int profiling;
    std::vector<ProfRec>* prof;
    long long* stamps;                 // diagnostic builds (-DNDWT_STAMPS): device buffer for the per-wave phase cycle sums
    void* stage[2];                    // device staging of the host-pointer forms: [0] one band (x / the result), [1] all bands; lazily grown,
    size_t stage_bytes[2];             // kept across calls (ndwt_plan_release_staging frees them)
    int live_coefs;                    // ndwt_coef handles bound to this plan
    int thin_slab;                     // slab plan whose outer axis is shorter than its filter: slab entry points only
    std::vector<hipEvent_t>* ev_pool;  // profiling events, reused
};

static int ensure_tmp(ndwt_plan* p, size_t bytes) {
    if (bytes <= p->tmp_bytes) return NDWT_OK;
    if (p->tmp) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(p->tmp));
        p->tmp = nullptr;
        p->tmp_bytes = 0;
    }
    hipError_t e = hipMalloc(&p->tmp, bytes);
    if (e != hipSuccess) return fail(NDWT_ERR_ALLOC, "hipMalloc(%zu bytes) for temporaries failed: %s", bytes, hipGetErrorString(e));
    p->tmp_bytes = bytes;
    return NDWT_OK;
}

// Events come from a pool owned by the plan (get_profile returns them to it), and a launch that did not happen
// (no instantiation: rc < 0, the caller falls through to another kernel) leaves no record.
static bool prof_event(const ndwt_plan* p, hipEvent_t* e) {
    if (!p->ev_pool->empty()) { *e = p->ev_pool->back(); p->ev_pool->pop_back(); return true; }
    return hipEventCreate(e) == hipSuccess;
}
static void prof_begin(const ndwt_plan* p, int kind, hipStream_t s) {
    if (!p->profiling) return;
    ProfRec r;
    r.kind = kind;
    if (!prof_event(p, &r.start)) return;
    if (!prof_event(p, &r.stop)) { p->ev_pool->push_back(r.start); return; }
    (void)hipEventRecord(r.start, s);
    p->prof->push_back(r);
}
static void prof_end(const ndwt_plan* p, hipStream_t s, int rc = 0) {
    if (!p->profiling || p->prof->empty()) return;
    if (rc != 0) {                                        // nothing was launched: drop the record
        p->ev_pool->push_back(p->prof->back().start);
        p->ev_pool->push_back(p->prof->back().stop);
        p->prof->pop_back();
        return;
    }
    (void)hipEventRecord(p->prof->back().stop, s);
}

static long long level_stride(const ndwt_plan* p, int lev) { return p->dilation == NDWT_DILATION_ATROUS ? (1LL << (lev - 1)) : 1LL; }

template <typename T> static bool aligned_vec4(const void* ptr) { return ((uintptr_t)ptr % (4 * sizeof(T))) == 0; }
template <typename T> static int launch_march(bool syn, int L, const MarchArgs<T>& a, const double* lo, const double* hi, hipStream_t s);
template <> int launch_march<float>(bool syn, int L, const MarchArgs<float>& a, const double* lo, const double* hi, hipStream_t s) {
    return launch_march_f32(syn, L, a, lo, hi, s);
}
template <> int launch_march<double>(bool syn, int L, const MarchArgs<double>& a, const double* lo, const double* hi, hipStream_t s) {
    return launch_march_f64(syn, L, a, lo, hi, s);
}

template <typename T> static int launch_axisx(bool syn, int L, int ew, const AxisXArgs<T>& a, bool vec4, const double* lo, const double* hi, hipStream_t s);
template <> int launch_axisx<float>(bool syn, int L, int ew, const AxisXArgs<float>& a, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    return launch_axisx_f32(syn, L, ew, a, vec4, lo, hi, s);
}
template <> int launch_axisx<double>(bool syn, int L, int ew, const AxisXArgs<double>& a, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    return launch_axisx_f64(syn, L, ew, a, vec4, lo, hi, s);
}

// ------------------------------------------------------------------------------ one axis, one pass
template <typename T>
static int axis_pass(const ndwt_plan* p, bool synthesis, int axis, const long long* dims_cur, long long stride, bool wrap,
                     const T* in0, const T* in1, T* out0, T* out1, hipStream_t s) {
    const AxisFilter& f = p->filt[axis];
    AxisTaps<T> tp;
    tp.len = f.len;
    for (int j = 0; j < kMaxTaps; ++j) { tp.lo[j] = 0; tp.hi[j] = 0; }
    for (int j = 0; j < f.len; ++j) {
        tp.lo[j] = (T)(synthesis ? f.syn_lo[j] : f.ana_lo[j]);
        tp.hi[j] = (T)(synthesis ? f.syn_hi[j] : f.ana_hi[j]);
    }
    AxisArgs<T> a;
    a.inner = p->comp;
    for (int k = 0; k < axis; ++k) a.inner *= dims_cur[k];
    a.outer = 1;
    for (int k = axis + 1; k < p->ndim; ++k) a.outer *= dims_cur[k];
    a.n = dims_cur[axis];
    a.stride = stride;
    a.left = (synthesis ? (long long)(f.len / 2) : (long long)(f.len / 2 - 1)) * stride;
    a.wrap = wrap ? 1 : 0;
    a.n_in = wrap ? a.n : a.n + (long long)(f.len - 1) * stride;
    a.total = a.outer * a.n * a.inner;
    if (a.total == 0) return NDWT_OK;
    // Register-window march kernel: needs 4-element aligned contiguous runs below the filtered index.  A dilated
    // (a-trous) periodic axis whose length is a multiple of the tap stride s is s interleaved unit-stride problems:
    // view [outer][n][inner] as [outer][n/s][s*inner] and march n/s -- this also covers the contiguous axis for s >= 4.
    long long m_inner = a.inner, m_n = a.n, m_n_in = a.n_in;
    bool march_ok = p->path == NDWT_PATH_AUTO;
    if (stride > 1) {
        if (wrap && a.n % stride == 0 && a.n / stride >= f.len) { m_inner = a.inner * stride; m_n = a.n / stride; m_n_in = m_n; }
        else march_ok = false;
    }
    if (march_ok && m_inner % 4 == 0 && m_inner >= 4 && aligned_vec4<T>(in0) &&
        (!synthesis || aligned_vec4<T>(in1)) && aligned_vec4<T>(out0) && (synthesis || aligned_vec4<T>(out1))) {
        MarchArgs<T> m;
        m.in0 = in0; m.in1 = in1; m.out0 = out0; m.out1 = out1;
        m.inner = m_inner; m.n = m_n; m.n_in = m_n_in; m.outer = a.outer; m.wrap = a.wrap;
        m.ngroups = m.inner / 4;
        const long long iblocks = (m.ngroups * m.outer + 255) / 256;
        long long want = (4096 + iblocks - 1) / iblocks;
        if (want < 1) want = 1;
        long long chunk = (m.n + want - 1) / want;
        const long long min_chunk = 4LL * (f.len - 1) > 8 ? 4LL * (f.len - 1) : 8;
        if (chunk < min_chunk) chunk = min_chunk;
        if (chunk > m.n) chunk = m.n;
        m.chunk = (int)chunk;
        m.nchunks = (int)((m.n + chunk - 1) / chunk);
        prof_begin(p, synthesis ? NDWT_KERNEL_AXIS_SYNTHESIS : NDWT_KERNEL_AXIS_ANALYSIS, s);
        int rc = launch_march<T>(synthesis, f.len, m, synthesis ? f.syn_lo : f.ana_lo, synthesis ? f.syn_hi : f.ana_hi, s);
        prof_end(p, s, rc);
        if (rc == 0) return NDWT_OK;
        if (rc > 0) return fail(NDWT_ERR_HIP, "march kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        // rc < 0: no instantiation / grid too large -> fall through to the element-wise kernel
    }
    // the contiguous axis (1-D signals, interleaved complex): one wave per row segment, neighbours by lane shifts
    if (p->path == NDWT_PATH_AUTO && axis == 0 && stride == 1 && wrap && f.len <= 12 && (p->comp == 1 || p->comp == 2) &&
        a.n * p->comp >= 8 * f.len) {
        AxisXArgs<T> x;
        x.in0 = in0; x.in1 = in1; x.out0 = out0; x.out1 = out1;
        x.row = a.n * p->comp;
        x.outer = a.outer;
        x.nseg = 0;
        const bool v4ok = x.row % 4 == 0 && aligned_vec4<T>(in0) && (!synthesis || aligned_vec4<T>(in1)) && aligned_vec4<T>(out0) &&
                          (synthesis || aligned_vec4<T>(out1));
        prof_begin(p, synthesis ? NDWT_KERNEL_AXIS_SYNTHESIS : NDWT_KERNEL_AXIS_ANALYSIS, s);
        int rc = launch_axisx<T>(synthesis, f.len, (int)p->comp, x, v4ok, synthesis ? f.syn_lo : f.ana_lo, synthesis ? f.syn_hi : f.ana_hi, s);
        prof_end(p, s, rc);
        if (rc == 0) return NDWT_OK;
        if (rc > 0) return fail(NDWT_ERR_HIP, "contiguous-axis kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    long long nb = (a.total + 255) / 256;
    const long long cap = 256LL * 64;   // grid-stride beyond 64 blocks per CU
    if (nb > cap) nb = cap;
    prof_begin(p, synthesis ? NDWT_KERNEL_AXIS_SYNTHESIS : NDWT_KERNEL_AXIS_ANALYSIS, s);
    if (synthesis)
        hipLaunchKernelGGL(axis_synthesis_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, in0, in1, out0, tp, a);
    else
        hipLaunchKernelGGL(axis_analysis_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, in0, out0, out1, tp, a);
    prof_end(p, s);
    HIP_TRY(hipGetLastError());
    return NDWT_OK;
}

// ---------------------------------------------------------------------- per-axis (general) levels
// depth-first over the band tree, outermost axis first; temporaries: 2 volumes per tree depth
template <typename T> struct GenericCtx {
    const ndwt_plan* p;
    long long stride;
    bool slab;
    long long dims_cur[NDWT_MAX_DIMS];
    long long vol_cur;       // scalars per volume with the current dims
    T* tmp;                  // 2*(ndim-1) volumes of vol_tmp scalars
    long long vol_tmp;
    hipStream_t s;
};

template <typename T> static int generic_analysis(GenericCtx<T>& c, int axis, const T* src, int prefix, T* const* out) {
    const bool top = axis == c.p->ndim - 1;
    const bool wrap = !(c.slab && top);
    if (axis == 0) return axis_pass<T>(c.p, false, 0, c.dims_cur, c.stride, wrap, src, nullptr, out[prefix], out[prefix | 1], c.s);
    T* lo = c.tmp + (long long)(2 * (axis - 1)) * c.vol_tmp;
    T* hi = lo + c.vol_tmp;
    int rc = axis_pass<T>(c.p, false, axis, c.dims_cur, c.stride, wrap, src, nullptr, lo, hi, c.s);
    if (rc) return rc;
    rc = generic_analysis(c, axis - 1, lo, prefix, out);
    if (rc) return rc;
    return generic_analysis(c, axis - 1, hi, prefix | (1 << axis), out);
}

template <typename T> static int generic_synthesis(GenericCtx<T>& c, int axis, int prefix, const T* const* in, T* dst) {
    const bool top = axis == c.p->ndim - 1;
    const bool wrap = !(c.slab && top);
    if (axis == 0) return axis_pass<T>(c.p, true, 0, c.dims_cur, c.stride, wrap, in[prefix], in[prefix | 1], dst, nullptr, c.s);
    T* a = c.tmp + (long long)(2 * (axis - 1)) * c.vol_tmp;
    T* d = a + c.vol_tmp;
    int rc = generic_synthesis(c, axis - 1, prefix, in, a);
    if (rc) return rc;
    rc = generic_synthesis(c, axis - 1, prefix | (1 << axis), in, d);
    if (rc) return rc;
    return axis_pass<T>(c.p, true, axis, c.dims_cur, c.stride, wrap, a, d, dst, nullptr, c.s);
}

// ------------------------------------------------------------------------------------ fused levels
// dir: 0 analysis, 1 synthesis, -1 both.  Instantiated tap lengths: 2..12 (db1..db6) for every data kind the checks below let
// through; float real data also 14 .. 20 (db7 .. db10; 18- and 20-tap synthesis with the pair-packed kernel only), double real data 14 and 16.
static bool inv3y_plan_ok(const ndwt_plan* p, int Lp);
static bool fused3_eligible(const ndwt_plan* p, long long stride, int* Lp_out, int dir = -1) {
    if (p->path != NDWT_PATH_AUTO || stride != 1 || p->ndim < 3) return false;
    if (p->dtype == NDWT_F64 && !p->fp64_fused) return false;
    int Lp = 2;
    for (int a = 0; a < 3; ++a) Lp = p->filt[a].len > Lp ? p->filt[a].len : Lp;
    const int lmax = p->complexity != NDWT_REAL ? (p->dtype == NDWT_F32 && dir == 0 ? 16 : 12) : (p->dtype == NDWT_F32 ? (dir == 0 ? 20 : 16) : 16);
    // 18- and 20-tap synthesis exist as the pair-packed kernel only (uniform wavelets, or mixed ones with even padding on every axis)
    if (Lp > lmax && !(dir == 1 && Lp <= 20 && inv3y_plan_ok(p, Lp))) return false;
    if (p->dtype == NDWT_F64 && Lp > 16) return false;   // double: up to db8 (64x8 tiles with 512 threads keep 10 .. 16 taps in 256 registers)
    // interleaved complex: the fused kernels with the x taps stepping over (re, im) pairs, tap lengths <= 8 (float: <= 12); rows of an
    // odd number of elements run the VEC4 = false instances (one access per lane wherever its 4 scalars are contiguous)
    // (complex128: 10 taps both ways, 12 taps analysis only -- its synthesis spills 500+ registers on every tile)
    // (complex64: 14 / 16 taps in the analysis, and in the synthesis through the pair-packed kernel -- the Lp > lmax clause above)
    if (p->complexity != NDWT_REAL && Lp > (p->dtype == NDWT_F32 ? 16 : (dir == 0 ? 12 : 10))) return false;
    long long nbatch = p->ndim == 4 ? p->dims[3] + 64 : 1;
    if (!fused3_fits(p->dims[0] * p->comp, p->dims[1], p->dims[2] + 64, nbatch)) return false;
    *Lp_out = Lp;
    return true;
}

// A dilated (a-trous) 3-D level whose axes all divide by the tap stride s is s^3 independent stride-1 problems on the
// sub-lattices: the fused kernels take x with the taps stepping over s interleaved scalars (the EW parameter, as for
// interleaved complex data) and the s^2 (y, z) sub-lattices as batch items with row / plane strides s*n1, s*n1*n2.
static bool fused3_dilated_eligible(const ndwt_plan* p, long long stride, int* Lp_out) {
    if (p->path != NDWT_PATH_AUTO || p->ndim != 3 || p->complexity != NDWT_REAL) return false;
    if (stride != 2 && !(stride == 4 && p->dtype == NDWT_F32)) return false;   // instantiated: EW = 2 (float, double), EW = 4 (float)
    if (p->dtype == NDWT_F64 && !p->fp64_fused) return false;
    int Lp = 2;
    for (int a = 0; a < 3; ++a) Lp = p->filt[a].len > Lp ? p->filt[a].len : Lp;
    if (Lp > 8) return false;
    for (int a = 0; a < 3; ++a)
        if (p->dims[a] % stride != 0) return false;
    if (p->dims[0] % 4 != 0) return false;
    if (!fused3_fits(p->dims[0], p->dims[1], p->dims[2] + 64, stride * stride)) return false;
    *Lp_out = Lp;
    return true;
}

// the 2-D analogue: x through EW = stride, the `stride` row sub-lattices as batch items
static bool fused2_dilated_eligible(const ndwt_plan* p, long long stride, int* Lp_out) {
    if (p->path != NDWT_PATH_AUTO || p->ndim != 2 || p->complexity != NDWT_REAL) return false;
    if (stride != 2 && !(stride == 4 && p->dtype == NDWT_F32)) return false;
    int Lp = p->filt[0].len > p->filt[1].len ? p->filt[0].len : p->filt[1].len;
    if (Lp > 8 || p->dims[0] % stride != 0 || p->dims[1] % stride != 0 || p->dims[0] % 4 != 0) return false;
    if (p->dims[0] >= (1LL << 30) || p->dims[1] >= (1LL << 30)) return false;
    *Lp_out = Lp;
    return true;
}

static bool fused2_eligible(const ndwt_plan* p, long long stride, int* Lp_out) {
    if (p->path != NDWT_PATH_AUTO || stride != 1 || p->ndim != 2) return false;
    int Lp = p->filt[0].len > p->filt[1].len ? p->filt[0].len : p->filt[1].len;
    // float real: up to db10, complex64 and double real: up to db8 (256-register budget), complex128: up to db4
    if (Lp > (p->dtype == NDWT_F32 ? (p->complexity == NDWT_REAL ? 20 : 16) : (p->complexity == NDWT_REAL ? 16 : 8))) return false;
    if (p->dims[0] >= (1LL << 30) || p->dims[1] >= (1LL << 30)) return false;
    *Lp_out = Lp;
    return true;
}

static FusedTapsD fused_taps(const ndwt_plan* p, int Lp, bool synthesis) {
    FusedTapsD t;
    t.Lp = Lp;
    for (int a = 0; a < 3; ++a) {
        if (a >= p->ndim) {
            for (int j = 0; j < kMaxTaps; ++j) t.lo[a][j] = t.hi[a][j] = 0.0;
            continue;
        }
        const AxisFilter& f = p->filt[a];
        pad_taps(synthesis ? f.syn_lo : f.ana_lo, f.len, Lp, t.lo[a]);
        pad_taps(synthesis ? f.syn_hi : f.ana_hi, f.len, Lp, t.hi[a]);
    }
    return t;
}

template <typename T> static int launch3(bool inverse, const Fused3Args<T>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* td, hipStream_t s);
template <> int launch3<float>(bool inverse, const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* td, hipStream_t s) {
    return inverse ? launch_inv3_f32(a, t, vec4, variant, ew, td, s) : launch_fwd3_f32(a, t, vec4, variant, ew, td, s);
}
template <> int launch3<double>(bool inverse, const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* td, hipStream_t s) {
    return inverse ? launch_inv3_f64(a, t, vec4, variant, ew, td, s) : launch_fwd3_f64(a, t, vec4, variant, ew, td, s);
}

namespace ndwt {
// the tile the launcher of ndwt_fused_kernels.h will pick (ew = scalars per x element: 1 real, 2 interleaved complex or a level
// dilated by 2, 4 a level dilated by 4).  The A/B variants exist for real, undilated data only.
void fused3_tile_shape(bool f64, bool inverse, int variant, int Lp, int* TX, int* TY, int ew) {
    if (ew != 1) variant = 0;
    *TX = 64;
    if (!inverse) {
        *TY = f64 ? (((Lp == 10 && ew == 1) || (variant == 1 && Lp >= 6 && Lp <= 8 && ew == 1) || (ew == 2 && Lp == 8)) ? 16 : 8) : 16;   // double: db5, complex db4 (db3/db4: variant 1) 64x16 with 512 threads; db6 64x8 with 512
        if (!f64 && ew == 1 && (((variant == 2 || variant == 6) && Lp <= 8) || (variant != 1 && Lp >= 10 && Lp <= 16))) *TY = 32;   // float, tall tile: 10 .. 16 taps (<= 8: A/B)
    } else if (f64) {
        *TY = ((variant == 3 && Lp == 8) || Lp >= 10) ? 8 : 16;                   // lane-shift kernel 64x16 (10 / 12 taps: 64x8); variant 3 = LDS kernel
    } else {
        *TY = (variant == 3 && Lp == 8) ? 16 : (ew == 4 ? 16 : 32);              // tall tile; x taps over 4 scalars: 64x16 / 512 threads
    }
}
}  // namespace ndwt

// Float synthesis default: the pair-packed kernel.  It derives the high-pass taps from the low-pass ones (mirror + alternating
// signs), which holds for the zero-padded taps of an axis when its padding (Lp - len) / 2 is even; it keeps plane offsets in
// 32-bit BYTE counts.  variant_inv 4 forces the older lane-shift kernel (Inv3S) for A/B runs.
static bool inv3y_plan_ok(const ndwt_plan* p, int Lp) {
    if (p->dtype != NDWT_F32 || Lp > (p->comp == 1 ? 20 : 16) || p->variant_inv == 3 || p->variant_inv == 4) return false;
    for (int ax = 0; ax < 3; ++ax)
        if (((Lp - p->filt[ax].len) / 2) % 2 != 0) return false;
    return p->dims[0] * p->comp * p->dims[1] < (1LL << 30);
}

// the y and z axes carry the same synthesis taps (the same wavelet): the pair-packed kernel then keeps one set of tap pairs for both
static bool uniform_yz(const ndwt_plan* p) {
    if (p->ndim < 3 || p->filt[1].len != p->filt[2].len) return false;
    for (int j = 0; j < p->filt[1].len; ++j)
        if (p->filt[1].syn_lo[j] != p->filt[2].syn_lo[j]) return false;
    return true;
}

// Nontemporal output stores: float data whose output rows are whole 128-byte lines (a nontemporal store of a partly covered line
// is a read-modify-write in memory; plain stores of neighbouring tiles merge in L2).  Double never (ndwt_device.h: stream_store).
template <typename T> static int nt_store_ok(long long rs, long long plane, long long bstride, T* const* out, int nout) {
    if (sizeof(T) != 4) return 0;
    const long long line = 128 / (long long)sizeof(T);
    if (rs % line != 0 || plane % line != 0 || bstride % line != 0) return 0;
    for (int b = 0; b < nout; ++b)
        if (((uintptr_t)out[b]) % 128 != 0) return 0;
    return 1;
}

// one fused 3-D launch over `nbatch` volumes. n3 = output planes; z_wrap=false: inputs carry the z halo
template <typename T>
static int fused3_run(const ndwt_plan* p, bool inverse, int Lp, const T* const* in, T* const* out, long long n3, long long nbatch,
                      long long in_bstride, long long out_bstride, int z_mode, hipStream_t s, long long zlo = 0, long long zhi = LLONG_MIN, long long zbs = 0, int shrink_mask = 0, int dil = 1,
                      const double* ttaps = nullptr) {
    Fused3Args<T> a;
    memset(&a, 0, sizeof a);
    if (ttaps) {                                          // 4-D analysis, t axis folded in: taps of this launch's t-band, frames = batch items
        for (int j = 0; j < Lp; ++j) a.tt[j] = (T)ttaps[j];
    }
    a.zlo = (int)zlo;                                     // mode 3: input planes outside [zlo, zhi) read as zero
    a.zhi = (int)(zhi != LLONG_MIN ? zhi : n3 - (Lp - 1));
    a.zbs = (int)zbs;
    if (inverse && p->shrink_mode && shrink_mask) {        // only reached with a lane-shift synthesis kernel (fused_shrink_capable)
        a.shrink_thr = (T)p->shrink_thr;
        a.shrink_mask = shrink_mask;
        a.shrink_hard = p->shrink_mode == 2;
    }
    a.n1 = (int)(p->dims[0] * p->comp);                   // scalars along x (interleaved complex: 2 per element)
    a.n2 = (int)(p->dims[1] / dil);                       // dil > 1: one (y, z) sub-lattice per batch item
    a.n3 = (int)(n3 / dil);
    a.nbatch = (int)(dil > 1 ? dil * dil : nbatch);
    a.in_bstride = in_bstride;
    a.out_bstride = out_bstride;
    a.z_wrap = z_mode;
    a.stamps = p->stamps;
    bool vec4 = (a.n1 % 4 == 0) && (in_bstride % 4 == 0) && (out_bstride % 4 == 0);
    const int nin = inverse ? 8 : (z_mode == 2 ? 3 : 1), nout = inverse ? 1 : 8;
    for (int b = 0; b < nin; ++b) { a.in[b] = in[b]; vec4 = vec4 && aligned_vec4<T>(in[b]); }
    for (int b = 0; b < nout; ++b) { a.out[b] = out[b]; vec4 = vec4 && aligned_vec4<T>(out[b]); }
#ifdef NDWT_EXP_BANDPAD   // diagnostic build (timing only, results are garbage): skew the band streams against each other
    if (const char* v = getenv("NDWT_EXP_BANDPAD")) {
        const long long pad = atoll(v);
        const int div = getenv("NDWT_EXP_BANDDIV") ? atoi(getenv("NDWT_EXP_BANDDIV")) : 1;   // bands b, b+1, .. b+div-1 keep their distance
        const int mask = getenv("NDWT_EXP_BANDMASK") ? (int)strtol(getenv("NDWT_EXP_BANDMASK"), nullptr, 0) : 0;   // these bands get +pad
        if (inverse && mask) for (int b = 0; b < nin; ++b) a.in[b] = in[b] + ((mask >> b) & 1) * pad;
        else if (inverse) for (int b = 0; b < nin; ++b) a.in[b] = in[b] + (b / div) * pad;
        else for (int b = 0; b < nout; ++b) a.out[b] = out[b] + (b / div) * pad;
    }
#endif
    int TX = 0, TY = 0;
    int variant = inverse ? p->variant_inv : p->variant_fwd;
    const int ew = dil > 1 ? dil : (int)p->comp;
    // float analysis, 6 and 8 taps: the tall 64x32 tile with 1024 threads (variant 2) where the volume has the tiles to fill the chip
    // with it (512^3 db4: 0.95 -> 0.85 ms per launch, db3 -4 %; db1 / db2 +3 %, 256^3 even); NDWT_VARIANT_FWD=3 keeps the 64x16 tile
    // ... and the same for interleaved complex64 (384^3 db4: 1.19 -> 1.03 ms per launch); NDWT_VARIANT_FWD=3 keeps the 64x16 tile
    const bool cplx_tall = !inverse && sizeof(T) == 4 && ew == 2 && dil == 1 && variant == 0 && Lp >= 6 && Lp <= 8 &&
                           (long long)((a.n1 + 63) / 64) * ((a.n2 + 31) / 32) * a.nbatch >= 32;
    if (!inverse && sizeof(T) == 4 && ew == 1 && variant == 0 && Lp >= 6 && Lp <= 8 &&
        (long long)((a.n1 + 63) / 64) * ((a.n2 + 31) / 32) * a.nbatch >= 32)
        variant = 2;
    if (ttaps) variant = 6;                               // the folded t axis runs on the tall tile with y items of 2 rows
    // float real analysis with 10 / 12 / 14 taps: the tall-tile kernel with its taps pinned in SGPRs and the high-pass ones derived from
    // the low-pass ones (Fwd3 PIN: no scalar loads in the plane loop; 512^3 db5 0.945 -> 0.889, db6 1.027 -> 0.967, db7 1.118 -> 1.103 ms
    // per launch, bit-identical).  Needs vec4 data and an even zero padding of every axis' taps; NDWT_VARIANT_FWD=8 keeps the plain
    // form (A/B).  (6 / 8 taps: +1 %, not used; 16 taps spill in this form.)
    bool pin_fwd = false;
    if (!inverse && sizeof(T) == 4 && ew == 1 && !ttaps && (variant == 0 || variant == 8)) {
        pin_fwd = variant == 0 && Lp >= 10 && Lp <= 14 && vec4;
        for (int ax = 0; ax < 3; ++ax) pin_fwd = pin_fwd && (ax >= p->ndim || ((Lp - p->filt[ax].len) / 2) % 2 == 0);
        if (variant == 8) variant = (Lp >= 6 && Lp <= 8 && (long long)((a.n1 + 63) / 64) * ((a.n2 + 31) / 32) * a.nbatch >= 32) ? 2 : 0;
    }
    // double analysis, 6 and 8 taps: 64x16 tile with 512 threads, one column per thread (384^3 db4: 1.29 -> 0.97 ms per launch,
    // 320^3 -15 %, 512^3 -2 %, 256^3 +2 %); NDWT_VARIANT_FWD=3 keeps the 64x8 tile with 256 threads
    if (!inverse && sizeof(T) == 8 && ew == 1 && variant == 0 && Lp >= 6 && Lp <= 8) variant = 1;
    fused3_tile_shape(sizeof(T) == 8, inverse, ew != 1 ? 0 : variant, Lp, &TX, &TY, ew);
    if (cplx_tall) TY = 32;
    bool use_y = false;                                   // float synthesis default: the pair-packed kernel and its tile
    if constexpr (sizeof(T) == 4) {
        // (plane offsets stay below 2^32 bytes: checked there).  A level dilated by 2 on real data is the interleaved-pair form of the
        // kernel (the two x sub-lattices are its (re, im) halves): 512^3 db4 synthesis at tap stride 2 1.52 -> see DESIGN 4.6
        use_y = inverse && (dil == 1 || ((dil == 2 || (dil == 4 && vec4)) && p->comp == 1 && p->variant_inv != 2)) && inv3y_plan_ok(p, Lp);
        // (the whole-lane-shift form of tap stride 4 exists for 16-byte-aligned data only: anything else keeps Inv3S<.., EW = 4>)
        if (use_y) { TX = ndwt::inv3y_tx(Lp, ew); TY = ndwt::inv3y_ty(Lp, ew); }
    }
    const int zc_force = p->zchunk_dir[inverse ? 1 : 0] > 0 ? p->zchunk_dir[inverse ? 1 : 0] : p->force_zchunk;
    // One round of workgroups that all fit on the chip at once beats several partial rounds (measured, 512^3 float
    // analysis: 512 workgroups 0.88 ms, 1024: 1.09 ms, 2048: 0.99 ms; 256^3 double synthesis: 256 workgroups 0.36 ms,
    // 640: 0.48 ms).  Workgroups per CU: synthesis 1 (1024 threads / 94 KB of LDS), analysis 2 (3 fit, 2 run faster).
    const bool small_inv = inverse && sizeof(T) == 4 && ew == 1 && variant == 3 && Lp == 8;   // 256-thread A/B variant
    const int per_cu = inverse ? (small_inv ? 3 : 1) : ((dil == 4 || (sizeof(T) == 4 && TY == 32)) ? 1 : 2);   // 1024-thread tiles: one per CU
    const int target = p->target_blocks > 0 ? p->target_blocks : p->num_cus * per_cu;
    // more tiles than resident slots: fused3_geometry picks the chunk count with the fewest plane steps over all rounds
    fused3_geometry(a, TX, TY, Lp, target, zc_force);
    if (dil > 1) {
        a.rs = (int)(dil * p->dims[0]);
        a.plane = (long long)dil * p->dims[0] * p->dims[1];
        a.bsplit = dil;
        a.in_bstride = a.out_bstride = p->dims[0];                         // sub-lattice offset along y ...
        a.in_bstride2 = a.out_bstride2 = p->dims[0] * p->dims[1];          // ... and along z
    }
    // (a tile narrower than whole lines -- the 48-wide pair-packed tiles of 20 taps / complex 12 taps -- would put tile edges inside
    // a line: two workgroups' partial nontemporal stores of one line, the read-modify-write case again)
    a.nt = ((long long)TX * (long long)sizeof(T)) % 128 == 0 ? nt_store_ok<T>(a.rs, a.plane, out_bstride, out, nout) : 0;
    FusedTapsD t = fused_taps(p, Lp, inverse);
    const void* td = p->taps_dev[inverse ? 1 : 0];
    if (!td) return fail(NDWT_ERR_UNSUPPORTED, "plan has no device tap table");
    prof_begin(p, inverse ? NDWT_KERNEL_FUSED_SYNTHESIS : NDWT_KERNEL_FUSED_ANALYSIS, s);
    int rc = -1;
    if constexpr (sizeof(T) == 4) {
        if (use_y) {
            // (tap stride 4, 8 taps: the x stage in scatter form -- 512^3 db4 1.78 -> 1.52 ms per level; 6 taps 1.117 / 1.110: the gather form
            //  stays; variant_inv 10 = scatter form for 4 / 6 taps too, 11 = gather form)
            rc = ew == 4 ? (vec4 ? launch_inv3y4_f32(a, Lp, variant == 5 ? 1 : 2, td, s, p->variant_inv == 10 || (Lp == 8 && p->variant_inv != 11)) : -1)
                 // (interleaved pairs: the scatter form from 10 taps on -- complex64 384^3 rec of 3 levels db5 3.78 -> 3.66 ms, db6 5.00 -> 4.66, db8 6.71 -> 5.83;
                 //  8 taps: 3.07 either way, and real data at tap stride 2 4 % SLOWER (0.97 -> 1.02 ms per level): the gather form stays there)
                 : ew == 2 ? launch_inv3yc_f32(a, Lp, vec4, variant == 5 ? 1 : 2, td, s, p->variant_inv == 10 || (Lp >= 10 && p->variant_inv != 11))
                 // real data on rows of whole groups of 4, 10 .. 20 taps: the x stage in scatter form (512^3 per launch db5 1.18 -> 1.09 ms,
                 // db6 1.34 -> 1.23, db9 2.66 -> 2.41, db10 3.08 -> 2.87; 8 taps: 1.04 either way, the gather form stays.  A/B: variant_inv
                 // 10 = scatter form for 8 taps too, 11 = gather form for every tap length)
                 : (vec4 && variant != 11 && (Lp >= 10 || variant == 10)) ? launch_inv3ys_f32(a, Lp, variant == 5 ? 1 : 2, td, s, uniform_yz(p)) : -1;
            if (rc == -1 && ew == 1) rc = launch_inv3y_f32(a, Lp, vec4, variant == 5 ? 1 : 2, td, s, uniform_yz(p) && variant != 9);
            if (rc == -1) {                               // the geometry above is this kernel's: never fall through to another one with it
                prof_end(p, s, rc);
                return fail(NDWT_ERR_UNSUPPORTED, "pair-packed synthesis kernel not instantiated for tap length %d", Lp);
            }
        }
    }
    if constexpr (sizeof(T) == 4) {
        if (ttaps) {
            rc = vec4 ? launch_fwd3_tpre_f32(a, Lp, td, s) : -1;
            if (rc == -1) {
                prof_end(p, s, rc);
                return fail(NDWT_ERR_UNSUPPORTED, "internal: no folded-t analysis kernel for tap length %d / this alignment", Lp);
            }
        }
        if (rc == -1 && pin_fwd) rc = launch_fwd3_pin_f32(a, Lp, td, s);
        if (rc == -1 && Lp > 12 && ew == 1) rc = launch_long3_f32(inverse, a, t, vec4, variant, td, s);
    }
    if constexpr (sizeof(T) == 8) {
        if (Lp > 12 && ew == 1) rc = launch_long3_f64(inverse, a, t, vec4, td, s);
    }
    if (rc == -1) rc = launch3<T>(inverse, a, t, vec4, ew != 1 ? (cplx_tall ? 2 : 0) : variant, ew, td, s);
    prof_end(p, s, rc);
    if (rc == -1) return fail(NDWT_ERR_UNSUPPORTED, "no fused kernel instantiated for tap length %d", Lp);
    if (rc == -2) return fail(NDWT_ERR_UNSUPPORTED, "internal: launch geometry (%d x %d tiles) does not match the kernel's tile shape", TX, TY);
    if (rc != 0) return fail(NDWT_ERR_HIP, "fused kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return NDWT_OK;
}

namespace ndwt {
int fused2_tile_width(bool inverse, int Lp, int ew) {
    const int LH = inverse ? Lp / 2 : Lp / 2 - 1, RH = inverse ? Lp / 2 - 1 : Lp / 2;
    return 4 * (64 - (LH * ew + 3) / 4 - (RH * ew + 3) / 4);
}
}  // namespace ndwt

template <typename T> static int launch2(bool inverse, const Fused2Args<T>& a, int Lp, bool vec4, int ew, const void* td, hipStream_t s);
template <> int launch2<float>(bool inverse, const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* td, hipStream_t s) {
    if (ew == 2 && Lp > 8) return inverse ? launch_inv2_c64_10to16(a, Lp, vec4, td, s) : launch_fwd2_c64_10to16(a, Lp, vec4, td, s);
    if (Lp > 12) return ew != 1 ? -1 : (inverse ? launch_inv2_f32_14to20(a, Lp, vec4, td, s) : launch_fwd2_f32_14to20(a, Lp, vec4, td, s));
    return inverse ? launch_inv2_f32(a, Lp, vec4, ew, td, s) : launch_fwd2_f32(a, Lp, vec4, ew, td, s);
}
template <> int launch2<double>(bool inverse, const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* td, hipStream_t s) {
    if (Lp > 12) return ew != 1 ? -1 : launch_long2_f64(inverse, a, Lp, vec4, td, s);
    return inverse ? launch_inv2_f64(a, Lp, vec4, ew, td, s) : launch_fwd2_f64(a, Lp, vec4, ew, td, s);
}

// one fused 2-D launch; n2 = output rows; y_wrap=false: inputs carry the y halo (slab mode)
template <typename T>
static int fused2_run(const ndwt_plan* p, bool inverse, int Lp, const T* const* in, T* const* out, long long n2, long long in_bstride,
                      long long out_bstride, bool y_wrap, hipStream_t s, int dil = 1) {
    Fused2Args<T> a;
    memset(&a, 0, sizeof a);
    a.n1 = (int)(p->dims[0] * p->comp);
    a.n2 = (int)(n2 / dil);                               // dil > 1: the dil row sub-lattices are the batch items
    a.nbatch = dil;
    a.in_bstride = dil > 1 ? p->dims[0] : in_bstride;
    a.out_bstride = dil > 1 ? p->dims[0] : out_bstride;
    a.y_wrap = y_wrap ? 1 : 0;
    if (inverse && p->shrink_mode) {                     // ndwt_denoise: threshold the 3 detail bands as they are loaded
        a.shrink_thr = (T)p->shrink_thr;
        a.shrink_mask = 0xE;
        a.shrink_hard = p->shrink_mode == 2;
    }
    bool vec4 = (a.n1 % 4 == 0);
    const int nin = inverse ? 4 : 1, nout = inverse ? 1 : 4;
    for (int b = 0; b < nin; ++b) { a.in[b] = in[b]; vec4 = vec4 && aligned_vec4<T>(in[b]); }
    for (int b = 0; b < nout; ++b) { a.out[b] = out[b]; vec4 = vec4 && aligned_vec4<T>(out[b]); }
    const int ew2 = dil > 1 ? dil : (int)p->comp;
    // float synthesis of real data in rows of whole groups of 4 scalars, images whose 70-row chunks fit one round of 1024 waves (up to
    // 4096^2): Inv2P -- 4 rows of band loads in flight per wave, the row loop unrolled in groups of L so that the compiler waits for
    // exactly the load a row needs, half as many waves on chunks twice as long (the L-1 prologue rows of a chunk are re-read).  Interleaved
    // A/B, 3 levels of rec: 1024^2 52 -> 43 us, 2048^2 81 -> 80, 4096^2 289 -> 265; 8192^2 927 -> 961 (not taken there: the run needs
    // more than one round of waves either way).  variant_inv: 1 keeps Inv2S, 2 / 4 = depth on Inv2S's geometry, 6 = depth 2 on 1024 waves.
    // Depth 4 runs in the packed form (pairs of adjacent x outputs per v_pk_fma_f32, tap pairs pinned in SGPRs; variant_inv 7 = scalar FMAs):
    // 4096^2 db4 74 us either way (the kernel sits on its memory floor), db6 116 -> 98 us, 2048^2 db4 27.1 -> 25.1 us per level.
    const int tiles2 = (a.n1 + fused2_tile_width(inverse, Lp, ew2) - 1) / fused2_tile_width(inverse, Lp, ew2);
    const bool deep = inverse && Lp <= 12 && (sizeof(T) == 4 || Lp <= 8) && ew2 == 1 && dil == 1 && vec4 && p->variant_inv != 1 && n2 >= 64 &&
                      ((long long)tiles2 * ((n2 + 69) / 70) <= 1280 || p->variant_inv >= 2);
    const int pdepth = (p->variant_inv == 2 || p->variant_inv == 6) ? 2 : 4;
    const int waves = (deep && p->variant_inv != 2 && p->variant_inv != 4) ? 1024 : 2048;
    fused2_geometry(a, fused2_tile_width(inverse, Lp, ew2), Lp, p->target_blocks > 0 ? p->target_blocks * 2 : waves, p->force_zchunk);
    if (dil > 1) a.rs = (int)(dil * p->dims[0]);   // 8 waves per CU: one round (measured optimum 1024^2 .. 4096^2)
    a.nt = nt_store_ok<T>(a.rs, a.rs, out_bstride, out, nout);
    const void* td = p->taps_dev[inverse ? 1 : 0];
    if (!td) return fail(NDWT_ERR_UNSUPPORTED, "plan has no device tap table");
    prof_begin(p, inverse ? NDWT_KERNEL_FUSED_SYNTHESIS : NDWT_KERNEL_FUSED_ANALYSIS, s);
    int rc = -1;
    if constexpr (sizeof(T) == 4) {
        if (deep) rc = launch_inv2p_f32(a, Lp, pdepth, td, s, p->variant_inv != 7);
    } else {
        if (deep) rc = launch_inv2p_f64(a, Lp, td, s);
    }
    if (rc == -1) rc = launch2<T>(inverse, a, Lp, vec4, ew2, td, s);
    prof_end(p, s, rc);
    if (rc == -1) return fail(NDWT_ERR_UNSUPPORTED, "no fused 2-D kernel instantiated for tap length %d", Lp);
    if (rc != 0) return fail(NDWT_ERR_HIP, "fused 2-D kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return NDWT_OK;
}

// ------------------------------------------------------------------------------------------ levels
// analysis of one level: in (vol scalars, + halo planes on the outer axis in slab mode) -> 2^d bands
template <typename T>
static int analysis_level(ndwt_plan* p, const T* in, T* const* out, long long stride, bool slab, hipStream_t s) {
    const int d = p->ndim;
    const AxisFilter& ftop = p->filt[d - 1];
    const long long n_top = p->dims[d - 1];
    const long long n_top_in = slab ? n_top + (long long)(ftop.len - 1) * stride : n_top;
    const long long vol_in = p->vol / n_top * n_top_in;
    int Lp = 0;
    if (!slab && fused3_dilated_eligible(p, stride, &Lp)) {
        const T* ins[8] = {in};
        return fused3_run<T>(p, false, Lp, ins, out, p->dims[2], 1, 0, 0, 1, s, 0, LLONG_MIN, 0, 0, (int)stride);
    }
    // slab mode hands over exactly (L_top-1) halo planes: the fused kernel marches with the padded length
    if (fused3_eligible(p, stride, &Lp, 0) && !(slab && d == 3 && ftop.len != Lp)) {
        const long long vol3 = p->comp * p->dims[0] * p->dims[1] * p->dims[2];
        if (d == 3) {
            const T* ins[8] = {in};
            return fused3_run<T>(p, false, Lp, ins, out, p->dims[2], 1, vol_in, p->vol, slab ? 0 : 1, s);
        }
        // A/B variant 7 (Plan.set_variant(fwd=7)) -- MEASURED AND NOT THE DEFAULT: the t axis folded into the fused launches, each raw plane a
        // workgroup takes being the t-filtered combination of the same plane of L frames (read where the neighbouring frames' workgroups
        // read them: L2), one launch per t-band: 17 volume transfers per level instead of 21 (nd_dwt_4D.m:394-467).  The bytes go down,
        // the time goes up: cfg5 (256^3 x 32, db4) 6.55 ms per launch against 3.41 ms + half of the 1.36 ms t pass -- the 8 loads per lane
        // and plane (against 1) go through the same per-CU vector-memory pipe as the 4 stores, and that pipe is what bounds the kernel
        // (68.6 ms per dec+rec step against 53.5 ms).
        if constexpr (sizeof(T) == 4) {
            bool ok = p->variant_fwd == 7 && !slab && stride == 1 && p->complexity == NDWT_REAL && Lp <= 8 && ftop.len <= Lp && p->dims[0] % 4 == 0 &&
                      vol3 % 4 == 0 && p->dims[3] >= 2 && aligned_vec4<T>(in);
            for (int b = 0; ok && b < 16; ++b) ok = aligned_vec4<T>(out[b]);
            if (ok) {
                double tlo[kMaxTaps], thi[kMaxTaps];
                pad_taps(ftop.ana_lo, ftop.len, Lp, tlo);
                pad_taps(ftop.ana_hi, ftop.len, Lp, thi);
                const T* ins[8] = {in};
                int rc = fused3_run<T>(p, false, Lp, ins, out, p->dims[2], p->dims[3], vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0, 1, tlo);
                if (rc) return rc;
                return fused3_run<T>(p, false, Lp, ins, out + 8, p->dims[2], p->dims[3], vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0, 1, thi);
            }
        }
        // otherwise: outer axis per-axis (1 -> 2), then the fused 3-D kernel on both halves, batched over n4
        const long long skew = 256 / (long long)sizeof(T);   // the two halves 256 B off a power-of-two distance (see ndwt_band_pitch)
        int rc = ensure_tmp(p, (size_t)(2 * p->vol + skew) * sizeof(T));
        if (rc) return rc;
        T* lo = (T*)p->tmp;
        T* hi = lo + p->vol + skew;
        rc = axis_pass<T>(p, false, 3, p->dims, stride, !slab, in, nullptr, lo, hi, s);
        if (rc) return rc;
        const T* ins_lo[8] = {lo};
        const T* ins_hi[8] = {hi};
        rc = fused3_run<T>(p, false, Lp, ins_lo, out, p->dims[2], p->dims[3], vol3, vol3, 1, s);
        if (rc) return rc;
        return fused3_run<T>(p, false, Lp, ins_hi, out + 8, p->dims[2], p->dims[3], vol3, vol3, 1, s);
    }
    if (!slab && fused2_dilated_eligible(p, stride, &Lp)) {
        const T* ins[4] = {in};
        return fused2_run<T>(p, false, Lp, ins, out, p->dims[1], 0, 0, true, s, (int)stride);
    }
    if (fused2_eligible(p, stride, &Lp) && !(slab && ftop.len != Lp)) {
        const T* ins[4] = {in};
        return fused2_run<T>(p, false, Lp, ins, out, p->dims[1], vol_in, p->vol, !slab, s);
    }
    GenericCtx<T> c;
    c.p = p; c.stride = stride; c.slab = slab; c.s = s;
    for (int k = 0; k < d; ++k) c.dims_cur[k] = p->dims[k];
    c.vol_cur = p->vol;
    c.vol_tmp = p->vol;
    if (d > 1) {
        int rc = ensure_tmp(p, (size_t)(2 * (d - 1)) * (size_t)c.vol_tmp * sizeof(T));
        if (rc) return rc;
    }
    c.tmp = (T*)p->tmp;
    return generic_analysis<T>(c, d - 1, in, 0, out);
}

template <typename T>
static int synthesis_level(ndwt_plan* p, const T* const* in, T* out, long long stride, bool slab, hipStream_t s) {
    const int d = p->ndim;
    const AxisFilter& ftop = p->filt[d - 1];
    const long long n_top = p->dims[d - 1];
    const long long n_top_in = slab ? n_top + (long long)(ftop.len - 1) * stride : n_top;
    const long long vol_in = p->vol / n_top * n_top_in;
    int Lp = 0;
    if (!slab && fused3_dilated_eligible(p, stride, &Lp)) {
        T* outs[8] = {out};
        return fused3_run<T>(p, true, Lp, in, outs, p->dims[2], 1, 0, 0, 1, s, 0, LLONG_MIN, 0, 0, (int)stride);
    }
    if (fused3_eligible(p, stride, &Lp, 1) && !(slab && d == 3 && ftop.len != Lp)) {
        const long long vol3 = p->comp * p->dims[0] * p->dims[1] * p->dims[2];
        if (d == 3) {
            T* outs[8] = {out};
            return fused3_run<T>(p, true, Lp, in, outs, p->dims[2], 1, vol_in, p->vol, slab ? 0 : 1, s, 0, LLONG_MIN, 0, 0xFE);
        }
        const long long skew = 256 / (long long)sizeof(T);
        int rc = ensure_tmp(p, (size_t)(2 * vol_in + skew) * sizeof(T));
        if (rc) return rc;
        T* a = (T*)p->tmp;
        T* dd = a + vol_in + skew;
        T* outs_a[8] = {a};
        T* outs_d[8] = {dd};
        rc = fused3_run<T>(p, true, Lp, in, outs_a, p->dims[2], n_top_in, vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0xFE);   // in[0] = approximation
        if (rc) return rc;
        rc = fused3_run<T>(p, true, Lp, in + 8, outs_d, p->dims[2], n_top_in, vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0xFF);   // t-high half: all details
        if (rc) return rc;
        return axis_pass<T>(p, true, 3, p->dims, stride, !slab, a, dd, out, nullptr, s);
    }
    if (!slab && fused2_dilated_eligible(p, stride, &Lp)) {
        T* outs[4] = {out};
        return fused2_run<T>(p, true, Lp, in, outs, p->dims[1], 0, 0, true, s, (int)stride);
    }
    if (fused2_eligible(p, stride, &Lp) && !(slab && ftop.len != Lp)) {
        T* outs[4] = {out};
        return fused2_run<T>(p, true, Lp, in, outs, p->dims[1], vol_in, p->vol, !slab, s);
    }
    GenericCtx<T> c;
    c.p = p; c.stride = stride; c.slab = slab; c.s = s;
    for (int k = 0; k < d; ++k) c.dims_cur[k] = p->dims[k];
    c.dims_cur[d - 1] = n_top_in;          // inner axes run on the haloed slab; the outer pass trims it
    c.vol_cur = vol_in;
    c.vol_tmp = vol_in;
    if (d > 1) {
        int rc = ensure_tmp(p, (size_t)(2 * (d - 1)) * (size_t)c.vol_tmp * sizeof(T));
        if (rc) return rc;
    }
    c.tmp = (T*)p->tmp;
    // the outermost pass must see dims_cur[d-1] == local length
    if (d == 1) {
        c.dims_cur[0] = n_top;
        return generic_synthesis<T>(c, 0, 0, in, out);
    }
    T* a = c.tmp + (long long)(2 * (d - 2)) * c.vol_tmp;
    T* dd = a + c.vol_tmp;
    int rc = generic_synthesis<T>(c, d - 2, 0, in, a);
    if (rc) return rc;
    rc = generic_synthesis<T>(c, d - 2, 1 << (d - 1), in, dd);
    if (rc) return rc;
    c.dims_cur[d - 1] = n_top;
    return axis_pass<T>(p, true, d - 1, c.dims_cur, stride, !slab, a, dd, out, nullptr, s);
}

// --------------------------------------------------------------------------------- multi-level
// band bookkeeping of nddwt.c:210,225-234 / nd_dwt_3D.m:178-186: level `lev` (1 = finest) stores its
// 2^d-1 detail bands at [1 + (2^d-1)(level-lev), ...); the coarsest approximation is band 0.  Unlike
// the reference there is no cat() copy (nd_dwt_3D.m:184) and no in-place overwrite of the input.
// bs: distance between consecutive bands of y in scalars (p->vol = the packed reference layout; larger = pitched)
// ---- two or three analysis levels of an image in one launch (Fwd2C, ndwt_device.h): float real data at tap stride 1, rows of whole
// groups of 4 scalars, up to 8 taps or 12.  variant_fwd 9 keeps one launch per level (A/B); force_zchunk = rows per wave.
static bool cascade2_eligible(const ndwt_plan* p, int* Lp_out) {
    int Lp = 0;
    if (p->dtype != NDWT_F32 || p->complexity != NDWT_REAL || p->dilation != NDWT_DILATION_REFERENCE || p->variant_fwd == 9) return false;
    if (!fused2_eligible(p, 1, &Lp) || (Lp > 8 && Lp != 12) || p->dims[0] % 4 != 0) return false;
    if (p->dims[0] * p->dims[1] >= (1LL << 31)) return false;   // the kernel's row * row-stride products are formed in 64 bits, offsets in int
    *Lp_out = Lp;
    return true;
}

static int cascade2_run(ndwt_plan* p, int Lp, int nlev, const float* in, float* const* out, hipStream_t s) {
    Fused2CArgs<float> a;
    memset(&a, 0, sizeof a);
    a.in = in;
    a.n1 = (int)p->dims[0];
    a.n2 = (int)p->dims[1];
    a.rs = a.n1;
    bool aligned = aligned_vec4<float>(in);
    for (int b = 0; b < 1 + 3 * nlev; ++b) { a.out[b] = out[b]; aligned = aligned && aligned_vec4<float>(out[b]); }
    if (!aligned) return -1;
    const int WX = fwd2c_tile_width(Lp, nlev);
    a.ntx = (a.n1 + WX - 1) / WX;
    // rows per wave: a wave reads nlev (Lp - 1) rows before its chunk produces anything, so chunks are longer than those of the
    // one-level kernel; one round of `waves` waves (2 per SIMD at the 256-register budget)
    // (db4, 3 levels, us per dec, one launch per level -> cascaded with 1024 / 2048 / 3584 waves: 4096^2 267 -> 206 / 179 / 190,
    // 8192^2 1062 -> 692 / 586 / 636; 2048^2 52 -> 52, 1024^2 25 -> 45: the cascade serves images beyond 2048^2, tools/bench2d_cascade.py)
    const int waves = p->target_blocks > 0 ? p->target_blocks : p->num_cus * 8;
    int chunks = waves / a.ntx;
    if (chunks < 1) chunks = 1;
    int yc = (a.n2 + chunks - 1) / chunks;
    const int min_chunk = nlev * (Lp - 1);                // the march-in is at most half of a wave's steps
    if (yc < min_chunk) yc = min_chunk;
    if (p->force_zchunk > 0) yc = p->force_zchunk;
    if (yc > a.n2) yc = a.n2;
    a.nyc = (a.n2 + yc - 1) / yc;
    a.ychunk = (a.n2 + a.nyc - 1) / a.nyc;
    a.nyc = (a.n2 + a.ychunk - 1) / a.ychunk;
    a.nt = nt_store_ok<float>(a.rs, a.rs, 0, out, 1 + 3 * nlev);
    a.mode = p->variant_fwd == 10 ? 1 : 0;
    const void* td = p->taps_dev[0];
    if (!td) return fail(NDWT_ERR_UNSUPPORTED, "plan has no device tap table");
    prof_begin(p, NDWT_KERNEL_FUSED_ANALYSIS, s);
    const int rc = launch_fwd2c_f32(a, Lp, nlev, td, s);
    prof_end(p, s, rc);
    if (rc == -2) return fail(NDWT_ERR_UNSUPPORTED, "internal: launch geometry does not match the cascaded 2-D kernel's tile");
    if (rc > 0) return fail(NDWT_ERR_HIP, "cascaded 2-D analysis launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;                                            // 0, or -1: no instance / unaligned (the caller takes one launch per level)
}

template <typename T> static int dec_impl(ndwt_plan* p, const T* x, T* y, long long bs, int level, hipStream_t s) {
    const int nb = 1 << p->ndim;
    const T* cur = x;
    if constexpr (sizeof(T) == 4) {
        int Lp = 0;
        // images beyond 2048^2 (below, the rows a wave reads before its chunk produces anything outweigh the volumes saved), or on request
        // (variant_fwd 11: tests and A/B runs on small images)
        if (level >= 2 && p->ndim == 2 && cascade2_eligible(p, &Lp) && p->dims[1] >= 3 * (Lp - 1) &&
            (p->vol > (6LL << 20) || p->variant_fwd == 11 || p->variant_fwd == 10)) {
            int lev = 1, pp = 0;                          // pp: the scratch volume the next launch writes (they alternate launch by launch: a
            while (level - lev + 1 >= 2) {                // launch never writes the approximation it reads); levels lev .. lev + n - 1 in one launch
                const int n = (level - lev + 1 >= 3 && Lp <= 8) ? 3 : 2;   // (12 taps: two levels fit the 256 registers, three do not)
                const int last = lev + n - 1;
                float* out[10];
                out[0] = (last == level) ? y : (float*)p->approx[pp];
                for (int l = 0; l < n; ++l)               // cascade level l (0 = first) is transform level lev + l
                    for (int b = 1; b < nb; ++b) out[1 + 3 * (n - 1 - l) + (b - 1)] = y + (long long)(1 + (nb - 1) * (level - (lev + l)) + (b - 1)) * bs;
                const int rc = cascade2_run(p, Lp, n, cur, out, s);
                if (rc == -1) break;                      // not this data (alignment): one launch per level from here on
                if (rc) return rc;
                cur = out[0];
                pp ^= 1;
                lev = last + 1;
            }
            for (; lev <= level; ++lev, pp ^= 1) {
                T* out[16];
                out[0] = (lev == level) ? y : (T*)p->approx[pp];
                for (int b = 1; b < nb; ++b) out[b] = y + (long long)(1 + (nb - 1) * (level - lev) + (b - 1)) * bs;
                int rc = analysis_level<T>(p, cur, out, level_stride(p, lev), false, s);
                if (rc) return rc;
                cur = out[0];
            }
            return NDWT_OK;
        }
    }
    for (int lev = 1; lev <= level; ++lev) {
        T* out[16];
        out[0] = (lev == level) ? y : (T*)p->approx[(lev - 1) & 1];
        for (int b = 1; b < nb; ++b) out[b] = y + (long long)(1 + (nb - 1) * (level - lev) + (b - 1)) * bs;
        int rc = analysis_level<T>(p, cur, out, level_stride(p, lev), false, s);
        if (rc) return rc;
        cur = out[0];
    }
    return NDWT_OK;
}

// the synthesis side of the cascade (Inv2C): in[0] = approximation of the coarsest level, then the detail bands coarsest level first
static int cascade2_rec_run(ndwt_plan* p, int Lp, int nlev, const float* const* in, float* out, hipStream_t s) {
    Fused2CIArgs<float> a;
    memset(&a, 0, sizeof a);
    a.out = out;
    a.n1 = (int)p->dims[0];
    a.n2 = (int)p->dims[1];
    a.rs = a.n1;
    bool aligned = aligned_vec4<float>(out);
    for (int b = 0; b < 1 + 3 * nlev; ++b) { a.in[b] = in[b]; aligned = aligned && aligned_vec4<float>(in[b]); }
    if (!aligned) return -1;
    const int WX = inv2c_tile_width(Lp, nlev);
    a.ntx = (a.n1 + WX - 1) / WX;
    // one wave per SIMD (db4, 3 levels, us per rec with 512 / 768 / 1024 / 1280 / 2048 waves: 4096^2 278 / 206 / 180 / 193 / 223 against 231
    // for one launch per level, 8192^2 1003 / 708 / 607 / 637 / 707 against 936; two rows of band loads in flight per level: 177 / 608)
    const int waves = p->target_blocks > 0 ? p->target_blocks : p->num_cus * 4;
    int chunks = waves / a.ntx;
    if (chunks < 1) chunks = 1;
    int yc = (a.n2 + chunks - 1) / chunks;
    const int min_chunk = nlev * (Lp - 1);
    if (yc < min_chunk) yc = min_chunk;
    if (p->force_zchunk > 0) yc = p->force_zchunk;
    if (yc > a.n2) yc = a.n2;
    a.nyc = (a.n2 + yc - 1) / yc;
    a.ychunk = (a.n2 + a.nyc - 1) / a.nyc;
    a.nyc = (a.n2 + a.ychunk - 1) / a.ychunk;
    float* outs[1] = {out};
    a.nt = nt_store_ok<float>(a.rs, a.rs, 0, outs, 1);
    if (p->shrink_mode) {                                 // ndwt_denoise: threshold the detail bands as their rows are loaded
        a.shrink_on = 1;
        a.shrink_thr = (float)p->shrink_thr;
        a.shrink_hard = p->shrink_mode == 2;
    }
    const void* td = p->taps_dev[1];
    if (!td) return fail(NDWT_ERR_UNSUPPORTED, "plan has no device tap table");
    prof_begin(p, NDWT_KERNEL_FUSED_SYNTHESIS, s);
    const int rc = launch_inv2c_f32(a, Lp, nlev, p->variant_inv == 12 ? 2 : 1, td, s);
    prof_end(p, s, rc);
    if (rc == -2) return fail(NDWT_ERR_UNSUPPORTED, "internal: launch geometry does not match the cascaded 2-D kernel's tile");
    if (rc > 0) return fail(NDWT_ERR_HIP, "cascaded 2-D synthesis launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

template <typename T> static int rec_impl(ndwt_plan* p, const T* y, long long bs, T* x, int level, hipStream_t s) {
    const int nb = 1 << p->ndim;
    const T* prev = y;   // band 0
    if constexpr (sizeof(T) == 4) {
        int Lp = 0;
        // (variant_inv 9: one launch per level; 11 / 12: cascade whatever the image size, one / two rows of band loads in flight)
        if (level >= 2 && p->ndim == 2 && p->variant_inv != 9 && cascade2_eligible(p, &Lp) && Lp <= 8 &&
            p->dims[1] >= 3 * (Lp - 1) && (p->vol > (6LL << 20) || p->variant_inv == 11 || p->variant_inv == 12)) {
            int lev = level, pp = 0;                      // coarsest level still to be synthesised; pp: the scratch volume the next launch writes
            while (lev >= 2) {
                const int n = lev >= 3 ? 3 : 2;           // levels lev, lev - 1, .. lev - n + 1 in one launch
                const float* in[10];
                in[0] = prev;
                for (int c = 0; c < n; ++c)               // cascade level c (0 = coarsest) is transform level lev - c
                    for (int b = 1; b < nb; ++b) in[1 + 3 * c + (b - 1)] = y + (long long)(1 + (nb - 1) * (level - (lev - c)) + (b - 1)) * bs;
                const int low = lev - n + 1;
                float* dst = (low == 1) ? x : (float*)p->approx[pp];
                const int rc = cascade2_rec_run(p, Lp, n, in, dst, s);
                if (rc == -1) break;
                if (rc) return rc;
                prev = dst;
                pp ^= 1;
                lev = low - 1;
            }
            for (; lev >= 1; --lev, pp ^= 1) {
                const T* in[16];
                in[0] = prev;
                for (int b = 1; b < nb; ++b) in[b] = y + (long long)(1 + (nb - 1) * (level - lev) + (b - 1)) * bs;
                T* dst = (lev == 1) ? x : (T*)p->approx[pp];
                int rc = synthesis_level<T>(p, in, dst, level_stride(p, lev), false, s);
                if (rc) return rc;
                prev = dst;
            }
            return NDWT_OK;
        }
    }
    for (int ind = 1; ind <= level; ++ind) {
        const int lev = level - ind + 1;
        const T* in[16];
        in[0] = prev;
        for (int b = 1; b < nb; ++b) in[b] = y + (long long)(1 + (nb - 1) * (level - lev) + (b - 1)) * bs;
        T* dst = (lev == 1) ? x : (T*)p->approx[(ind - 1) & 1];
        int rc = synthesis_level<T>(p, in, dst, level_stride(p, lev), false, s);
        if (rc) return rc;
        prev = dst;
    }
    return NDWT_OK;
}

static int check_level(const ndwt_plan* p, int level) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    if (level < 1 || level > p->max_level)
        return fail(NDWT_ERR_INVALID_ARG, "level %d outside 1..max_level=%d of this plan", level, p->max_level);
    if (p->thin_slab)
        return fail(NDWT_ERR_FILTER_TOO_LONG, "this slab plan is thinner than its outer-axis filter: only the *_slab entry points apply");
    if (p->dilation == NDWT_DILATION_ATROUS) {
        for (int a = 0; a < p->ndim; ++a) {
            long long span = (long long)(p->filt[a].len - 1) * (1LL << (level - 1)) + 1;
            (void)span;   // spans longer than the axis wrap several times; the kernels handle it
        }
    }
    return NDWT_OK;
}

// fused 3-D slab forms that avoid haloed copies (multi-GPU fast path)
static int slab_fast_ok(const ndwt_plan* p, int stride, int* Lp) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    if (p->ndim != 3 || !fused3_eligible(p, stride, Lp) || p->filt[2].len != *Lp)
        return fail(NDWT_ERR_UNSUPPORTED, "split/extended slab entry points need a fused 3-D plan whose outer-axis filter is the longest");
    return NDWT_OK;
}

template <typename T>
static int slab_split_impl(ndwt_plan* p, int Lp, const void* in, const void* hb, const void* ha, void* const* out, hipStream_t s) {
    const T* ins[8] = {(const T*)in, (const T*)hb, (const T*)ha};
    return fused3_run<T>(p, false, Lp, ins, (T* const*)out, p->dims[2], 1, p->vol, p->vol, 2, s);
}

template <typename T> static int slab_ext_impl(ndwt_plan* p, int Lp, const void* const* in, void* out, hipStream_t s) {
    T* outs[8] = {(T*)out};
    const long long n_out = p->dims[2] + (Lp - 1);
    return fused3_run<T>(p, true, Lp, (const T* const*)in, outs, n_out, 1, p->vol, p->vol / p->dims[2] * n_out, 3, s);
}


// --------------------------------------------------------------------------- shrinkage of detail bands
// Element-wise, in place, 1 read + 1 write per coefficient (HBM-bound).  COMP = 2: interleaved complex, the magnitude
// is shrunk and the phase kept.  mode 0 soft: c * max(|c| - t, 0) / |c|; mode 1 hard: c if |c| > t else 0.
template <typename T, int COMP> __device__ __forceinline__ void shrink_group(T* c, T thr, int hard) {
    if (COMP == 1) {
        const T m = c[0] < T(0) ? -c[0] : c[0];
        if (hard) c[0] = m > thr ? c[0] : T(0);
        else c[0] = m > thr ? (c[0] < T(0) ? c[0] + thr : c[0] - thr) : T(0);
    } else {
        const T m = sqrt(c[0] * c[0] + c[1] * c[1]);
        const T g = m > thr ? (hard ? T(1) : (m - thr) / m) : T(0);
        c[0] *= g;
        c[1] *= g;
    }
}

// VEC: 4 scalars per thread and step (16-byte / 32-byte accesses; pointer aligned, n a multiple of 4); else one group
template <typename T, int COMP, bool VEC>
__global__ void __launch_bounds__(256) shrink_kernel(T* __restrict__ y, long long n_scalars, T thr, int hard) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (VEC) {
        typedef typename VecT<T>::v4 v4;
        v4* yv = reinterpret_cast<v4*>(y);
        for (long long i = t0; i < n_scalars / 4; i += stride) {
            v4 c = yv[i];
            T e[4] = {c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int k = 0; k < 4; k += COMP) shrink_group<T, COMP>(e + k, thr, hard);
            yv[i] = v4{e[0], e[1], e[2], e[3]};
        }
    } else {
        for (long long i = t0; i < n_scalars / COMP; i += stride) shrink_group<T, COMP>(y + i * COMP, thr, hard);
    }
}

template <typename T> static int shrink_run(ndwt_plan* p, T* d, long long n, double thr, int mode, hipStream_t s);
template <typename T> static int shrink_impl(ndwt_plan* p, T* y, long long bs, int level, double thr, int mode, hipStream_t s) {
    const long long nb = ndwt_num_bands(p->ndim, level);
    // band 0 (coarsest approximation) is left as it is; packed bands are one run, pitched ones a launch per band
    if (bs == p->vol) return shrink_run<T>(p, y + p->vol, (nb - 1) * p->vol, thr, mode, s);
    for (long long b = 1; b < nb; ++b) {
        int rc = shrink_run<T>(p, y + b * bs, p->vol, thr, mode, s);
        if (rc) return rc;
    }
    return NDWT_OK;
}
template <typename T> static int shrink_run(ndwt_plan* p, T* d, long long n, double thr, int mode, hipStream_t s) {
    const bool vec = aligned_vec4<T>(d) && n % 4 == 0;
    long long blocks = ((vec ? n / 4 : n / p->comp) + 255) / 256;
    const long long cap = (long long)p->num_cus * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(256);
    if (p->comp == 2) {
        if (vec) hipLaunchKernelGGL((shrink_kernel<T, 2, true>), g, b, 0, s, d, n, (T)thr, mode);
        else hipLaunchKernelGGL((shrink_kernel<T, 2, false>), g, b, 0, s, d, n, (T)thr, mode);
    } else {
        if (vec) hipLaunchKernelGGL((shrink_kernel<T, 1, true>), g, b, 0, s, d, n, (T)thr, mode);
        else hipLaunchKernelGGL((shrink_kernel<T, 1, false>), g, b, 0, s, d, n, (T)thr, mode);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NDWT_ERR_HIP, "shrink kernel launch failed: %s", hipGetErrorString(e));
    return NDWT_OK;
}

// 4-D slab, zero-extended synthesis (scatter scheme of the t-sharded driver): the 3-D part is local to every frame, so
// the 16 bands of the slab are synthesised to the two t-bands (a, d) for the local frames only, in buffers that carry
// L-1 zero frames on each side; the t-axis pass over them yields the n_local + L-1 frames of the zero-extended result.
template <typename T> static int slab_ext4_impl(ndwt_plan* p, int Lp, const void* const* in, void* out, hipStream_t s) {
    const long long vol3 = p->comp * p->dims[0] * p->dims[1] * p->dims[2];
    const long long n = p->dims[3], h = p->filt[3].len - 1;
    const long long nin = n + 2 * h, nout = n + h;
    int rc = ensure_tmp(p, (size_t)(2 * nin * vol3) * sizeof(T));
    if (rc) return rc;
    T* a = (T*)p->tmp;
    T* dd = a + nin * vol3;
    for (T* b : {a, dd}) {
        HIP_TRY(hipMemsetAsync(b, 0, (size_t)(h * vol3) * sizeof(T), s));
        HIP_TRY(hipMemsetAsync(b + (h + n) * vol3, 0, (size_t)(h * vol3) * sizeof(T), s));
    }
    const T* const* inT = (const T* const*)in;
    T* outs_a[8] = {a + h * vol3};
    T* outs_d[8] = {dd + h * vol3};
    rc = fused3_run<T>(p, true, Lp, inT, outs_a, p->dims[2], n, vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0xFE);
    if (rc) return rc;
    rc = fused3_run<T>(p, true, Lp, inT + 8, outs_d, p->dims[2], n, vol3, vol3, 1, s, 0, LLONG_MIN, 0, 0xFF);
    if (rc) return rc;
    long long dims_ext[NDWT_MAX_DIMS];
    for (int k = 0; k < 4; ++k) dims_ext[k] = p->dims[k];
    dims_ext[3] = nout;                                  // outputs; the pass reads nout + L-1 = nin frames (slab mode)
    return axis_pass<T>(p, true, 3, dims_ext, 1, false, a, dd, (T*)out, nullptr, s);
}

// a run of output planes of the slab transform (the caller offsets the pointers): what lets the halo exchange
// overlap with the planes that do not depend on it
template <typename T>
static int slab_analysis_part_impl(ndwt_plan* p, int Lp, const void* in, const void* hb, const void* ha, void* const* out,
                                   long long n_planes, hipStream_t s) {
    const T* ins[8] = {(const T*)in, (const T*)hb, (const T*)ha};
    return fused3_run<T>(p, false, Lp, ins, (T* const*)out, n_planes, 1, p->vol, p->vol, 2, s);
}

template <typename T>
static int slab_analysis_runs_impl(ndwt_plan* p, int Lp, const void* in, void* const* out, long long n_planes, long long n_runs,
                                   long long run_stride, hipStream_t s) {
    const long long plane = p->vol / p->dims[2];
    const T* ins[8] = {(const T*)in};
    return fused3_run<T>(p, false, Lp, ins, (T* const*)out, n_planes, n_runs, run_stride * plane, run_stride * plane, 0, s);
}

// run r: planes [e0 + r*e_stride, +n_out) of the zero-extended synthesis of n_in coefficient planes -> out + r*n_out planes
template <typename T>
static int slab_synthesis_runs_impl(ndwt_plan* p, int Lp, const void* const* in, long long n_in, long long e0, long long e_stride,
                                    long long n_runs, long long n_out, void* out, hipStream_t s) {
    const long long plane = p->vol / p->dims[2];
    const T* ins[8];
    for (int b = 0; b < 8; ++b) ins[b] = (const T*)in[b] + e0 * plane;      // never dereferenced outside [0, n_in)
    T* outs[8] = {(T*)out};
    return fused3_run<T>(p, true, Lp, ins, outs, n_out, n_runs, e_stride * plane, n_out * plane, 3, s, -e0, n_in - e0, e_stride);
}

// ------------------------------------------------------------------------------------------ C ABI
// ---- several runs of planes copied / added in one launch (include/ndwt.h: ndwt_slab_segments) ----
namespace {
struct SegArgs {
    void* dst[NDWT_MAX_SEGMENTS];
    const void* src[NDWT_MAX_SEGMENTS];
    long long count[NDWT_MAX_SEGMENTS];      // in units of V
};
// blockIdx.y = run; 16-byte accesses (V = 4 floats / 2 doubles) where every run allows them, scalars otherwise
template <typename V, bool ADD>
__global__ __launch_bounds__(256) void segments_kernel(const SegArgs a) {
    V* __restrict__ d = (V*)a.dst[blockIdx.y];
    const V* __restrict__ s = (const V*)a.src[blockIdx.y];
    const long long n = a.count[blockIdx.y], step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
        if constexpr (ADD) d[i] = d[i] + s[i];
        else d[i] = s[i];
    }
}
template <typename T> int segments_launch(int op, int nseg, void* const* dst, const void* const* src, const int64_t* count, hipStream_t st) {
    typedef typename VecT<T>::v4 V4;
    constexpr int per = 16 / (int)sizeof(T);
    bool vec = true;
    long long most = 0;
    for (int i = 0; i < nseg; ++i) {
        vec = vec && count[i] % per == 0 && (uintptr_t)dst[i] % 16 == 0 && (uintptr_t)src[i] % 16 == 0;
        if (count[i] > most) most = count[i];
    }
    SegArgs a;
    memset(&a, 0, sizeof a);
    for (int i = 0; i < nseg; ++i) { a.dst[i] = dst[i]; a.src[i] = src[i]; a.count[i] = vec ? count[i] / per : count[i]; }
    long long bx = ((vec ? most / per : most) + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 2048) bx = 2048;
    const dim3 grid((unsigned)bx, (unsigned)nseg);
    if (sizeof(T) == 4) {
        if (vec) { if (op) hipLaunchKernelGGL((segments_kernel<V4, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((segments_kernel<V4, false>), grid, dim3(256), 0, st, a); }
        else { if (op) hipLaunchKernelGGL((segments_kernel<T, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((segments_kernel<T, false>), grid, dim3(256), 0, st, a); }
    } else {
        typedef typename VecT<T>::v2 V2;
        if (vec) { if (op) hipLaunchKernelGGL((segments_kernel<V2, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((segments_kernel<V2, false>), grid, dim3(256), 0, st, a); }
        else { if (op) hipLaunchKernelGGL((segments_kernel<T, true>), grid, dim3(256), 0, st, a); else hipLaunchKernelGGL((segments_kernel<T, false>), grid, dim3(256), 0, st, a); }
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" {

int ndwt_wave_filters(const char* wname, double* lo_d, double* hi_d, int* len) {
    const int K = parse_wavelet(wname);
    if (!K) return fail(NDWT_ERR_UNKNOWN_WAVELET, "Unknown Wavelet Name");
    if (!lo_d || !hi_d || !len) return fail(NDWT_ERR_INVALID_ARG, "null output pointer");
    wave_filters(K, lo_d, hi_d);
    *len = 2 * K;
    return NDWT_OK;
}

int64_t ndwt_num_bands(int ndim, int level) {
    if (ndim < 1 || ndim > NDWT_MAX_DIMS || level < 1) return -1;
    return (int64_t)(1 << ndim) + (int64_t)((1 << ndim) - 1) * (level - 1);
}

int ndwt_level_from_bands(int ndim, int64_t bands) {
    if (ndim < 1 || ndim > NDWT_MAX_DIMS) return -1;
    const int64_t nb = 1 << ndim;
    if (bands < nb || (bands - nb) % (nb - 1) != 0) return -1;
    return (int)(1 + (bands - nb) / (nb - 1));
}

static bool den3_eligible(const ndwt_plan* p, int* Lp_out);
static int den3_taps(ndwt_plan* p, int Lp);
static int plan_create_impl(ndwt_plan** plan, int ndim, const int64_t* dims, long long global_outer, const char* const* wnames, int dtype,
                            int complexity, int pres_l2_norm, int dilation, int max_level, int device) {
    if (!plan) return fail(NDWT_ERR_INVALID_ARG, "null plan pointer");
    *plan = nullptr;
    if (ndim < 1 || ndim > NDWT_MAX_DIMS) return fail(NDWT_ERR_INVALID_ARG, "ndim must be 1..4");
    if (!dims || !wnames) return fail(NDWT_ERR_INVALID_ARG, "null dims/wnames");
    if (dtype != NDWT_F32 && dtype != NDWT_F64) return fail(NDWT_ERR_INVALID_ARG, "dtype must be NDWT_F32 or NDWT_F64");
    if (complexity != NDWT_REAL && complexity != NDWT_COMPLEX_INTERLEAVED) return fail(NDWT_ERR_INVALID_ARG, "bad complexity");
    if (dilation != NDWT_DILATION_REFERENCE && dilation != NDWT_DILATION_ATROUS) return fail(NDWT_ERR_INVALID_ARG, "bad dilation mode");
    if (max_level < 1 || max_level > 30) return fail(NDWT_ERR_INVALID_ARG, "max_level must be 1..30");
    ndwt_plan* p = new ndwt_plan();
    memset(p, 0, sizeof *p);
    p->ndim = ndim;
    p->dtype = dtype;
    p->complexity = complexity;
    p->l2 = pres_l2_norm ? 1 : 0;
    p->dilation = dilation;
    p->max_level = max_level;
    p->device = device;
    p->path = NDWT_PATH_AUTO;
    p->esize = dtype == NDWT_F32 ? 4 : 8;
    p->comp = complexity == NDWT_COMPLEX_INTERLEAVED ? 2 : 1;
    p->target_blocks = 0;
    p->prof = new std::vector<ProfRec>();
    p->ev_pool = new std::vector<hipEvent_t>();
    p->fused_level1 = 1;
    p->fp64_fused = 1;   // measured: 256^3 fp64 db4 L3 2.5 ms fused (LDS analysis + lane-shift synthesis) vs 4.1 ms per-axis
    static const char* ordn[4] = {"First", "Second", "Third", "Fourth"};
    p->vol = p->comp;
    for (int a = 0; a < ndim; ++a) {
        if (dims[a] < 1) { delete p->prof; delete p->ev_pool; delete p; return fail(NDWT_ERR_INVALID_ARG, "dims[%d] must be >= 1", a); }
        const int K = parse_wavelet(wnames[a]);
        if (!K) { delete p->prof; delete p->ev_pool; delete p; return fail(NDWT_ERR_UNKNOWN_WAVELET, "Unknown Wavelet Name"); }
        p->dims[a] = dims[a];
        p->order[a] = K;
        p->filt[a] = make_axis_filter(K, p->l2 != 0);
        // nd_dwt_3D.m:277-286; for a slab plan the check is on the whole sharded axis, not on the local planes
        const long long axis_len = (a == ndim - 1 && global_outer > 0) ? global_outer : dims[a];
        if (a == ndim - 1 && global_outer > 0 && p->filt[a].len > dims[a]) p->thin_slab = 1;
        if (p->filt[a].len > axis_len) {
            delete p->prof;
            delete p->ev_pool;
            delete p;
            return fail(NDWT_ERR_FILTER_TOO_LONG, "%s Dimension of Data is shorter than the wavelet filter being used", ordn[a]);
        }
        p->vol *= dims[a];
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        delete p->prof;
        delete p->ev_pool;
        delete p;
        return fail(NDWT_ERR_NO_DEVICE, "no usable HIP device (requested %d of %d): this engine has no CPU path", device, ndev);
    }
    if (hipSetDevice(device) != hipSuccess) { delete p->prof; delete p->ev_pool; delete p; return fail(NDWT_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device); }
    {
        hipDeviceProp_t prop;
        p->num_cus = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const int napprox = max_level >= 3 ? 2 : (max_level == 2 ? 1 : 0);
    for (int i = 0; i < napprox; ++i) {
        // 256 B off the allocation's (power-of-two) alignment: the approximation is then the one band of a level that does not
        // share the address bits below 1 KiB with the packed detail bands (about a quarter of the pitched gain)
        hipError_t e = hipMalloc(&p->approx_base[i], (size_t)p->vol * p->esize + kApproxSkew);
        if (e == hipSuccess) p->approx[i] = (char*)p->approx_base[i] + kApproxSkew;
        if (e != hipSuccess) {
            ndwt_plan_destroy(p);
            return fail(NDWT_ERR_ALLOC, "hipMalloc of the approximation scratch failed: %s", hipGetErrorString(e));
        }
    }
    int Lp = 0;
    if (fused3_eligible(p, 1, &Lp, 0) || fused2_eligible(p, 1, &Lp)) {   // (the analysis side admits the most tap lengths)
        for (int inv = 0; inv < 2; ++inv) {
            FusedTapsD t = fused_taps(p, Lp, inv != 0);
            // synthesis table: Taps3Y = Taps3 followed by the x tap pairs (lo[0][k], lo[0][k-1]), k = 0..Lp, of the pair-packed kernel
            std::vector<char> host((size_t)(6 * Lp + (inv ? 4 * (Lp + 1) : 0)) * p->esize);
            for (int ax = 0; ax < 3; ++ax)
                for (int j = 0; j < Lp; ++j) {
                    if (dtype == NDWT_F32) {
                        ((float*)host.data())[ax * Lp + j] = (float)t.lo[ax][j];
                        ((float*)host.data())[3 * Lp + ax * Lp + j] = (float)t.hi[ax][j];
                    } else {
                        ((double*)host.data())[ax * Lp + j] = t.lo[ax][j];
                        ((double*)host.data())[3 * Lp + ax * Lp + j] = t.hi[ax][j];
                    }
                }
            if (inv) {
                for (int k = 0; k <= Lp; ++k)
                    for (int h = 0; h < 2; ++h) {
                        const int j = k - h;                                  // (t[k], t[k-1])
                        const double vlo = (j >= 0 && j < Lp) ? t.lo[0][j] : 0.0, vhi = (j >= 0 && j < Lp) ? t.hi[0][j] : 0.0;
                        const size_t ilo = (size_t)6 * Lp + 2 * k + h, ihi = ilo + 2 * (Lp + 1);
                        if (dtype == NDWT_F32) { ((float*)host.data())[ilo] = (float)vlo; ((float*)host.data())[ihi] = (float)vhi; }
                        else { ((double*)host.data())[ilo] = vlo; ((double*)host.data())[ihi] = vhi; }
                    }
            }
            hipError_t e = hipMalloc(&p->taps_dev[inv], host.size());
            if (e == hipSuccess) e = hipMemcpy(p->taps_dev[inv], host.data(), host.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                ndwt_plan_destroy(p);
                return fail(NDWT_ERR_ALLOC, "uploading the tap table failed: %s", hipGetErrorString(e));
            }
        }
    }
    {   // the tap table of the fused level-1 denoising kernel, where that kernel can serve this plan: ndwt_denoise then only enqueues
        // (den3_eligible is defined further down; the default fused_level1 = 1 admits up to 6 taps, 2 admits 8: build for 8)
        int Lden = 0;
        const int keep = p->fused_level1;
        p->fused_level1 = 2;
        const bool den = den3_eligible(p, &Lden);
        p->fused_level1 = keep;
        if (den && den3_taps(p, Lden) != NDWT_OK) {
            ndwt_plan_destroy(p);
            return NDWT_ERR_ALLOC;
        }
    }
    *plan = p;
    return NDWT_OK;
}

int ndwt_plan_create(ndwt_plan** plan, int ndim, const int64_t* dims, const char* const* wnames, int dtype, int complexity,
                     int pres_l2_norm, int dilation, int max_level, int device) {
    return plan_create_impl(plan, ndim, dims, -1, wnames, dtype, complexity, pres_l2_norm, dilation, max_level, device);
}

int ndwt_plan_create_slab(ndwt_plan** plan, int ndim, const int64_t* dims_local, int64_t global_outer, const char* const* wnames,
                          int dtype, int complexity, int pres_l2_norm, int dilation, int max_level, int device) {
    if (ndim >= 1 && ndim <= NDWT_MAX_DIMS && dims_local && global_outer < dims_local[ndim - 1])
        return fail(NDWT_ERR_INVALID_ARG, "global_outer (%lld) is shorter than the local slab (%lld)", (long long)global_outer,
                    (long long)dims_local[ndim - 1]);
    return plan_create_impl(plan, ndim, dims_local, global_outer, wnames, dtype, complexity, pres_l2_norm, dilation, max_level, device);
}

int ndwt_plan_destroy(ndwt_plan* p) {
    if (!p) return NDWT_OK;
    if (p->live_coefs > 0)                                // (a handle points back at its plan: release the handles first)
        return fail(NDWT_ERR_INVALID_ARG, "%d coefficient handle(s) of this plan are still alive: ndwt_coef_release them first", p->live_coefs);
    (void)hipSetDevice(p->device);
    for (int i = 0; i < 2; ++i)
        if (p->approx_base[i]) (void)hipFree(p->approx_base[i]);
    if (p->tmp) (void)hipFree(p->tmp);
    if (p->coef) (void)hipFree(p->coef);
    if (p->taps_den) (void)hipFree(p->taps_den);
    if (p->den_a1) (void)hipFree(p->den_a1);
    for (int i = 0; i < 2; ++i)
        if (p->stage[i]) (void)hipFree(p->stage[i]);
    for (int i = 0; i < 2; ++i)
        if (p->taps_dev[i]) (void)hipFree(p->taps_dev[i]);
    if (p->prof) {
        for (auto& r : *p->prof) { (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
        delete p->prof;
    }
    if (p->ev_pool) {
        for (auto e : *p->ev_pool) (void)hipEventDestroy(e);
        delete p->ev_pool;
    }
    delete p;
    return NDWT_OK;
}

int ndwt_plan_set_profiling(ndwt_plan* p, int enable) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    p->profiling = enable ? 1 : 0;
    return NDWT_OK;
}

// sums (and clears) the event records of one kernel kind; synchronises the device
int ndwt_plan_get_profile(ndwt_plan* p, int kind, double* total_ms, int64_t* launches) {
    if (!p || !total_ms || !launches) return fail(NDWT_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    double tot = 0;
    int64_t n = 0;
    std::vector<ProfRec> keep;
    for (auto& r : *p->prof) {
        if (r.kind != kind) { keep.push_back(r); continue; }
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) { tot += ms; ++n; }
        p->ev_pool->push_back(r.start);
        p->ev_pool->push_back(r.stop);
    }
    p->prof->swap(keep);
    *total_ms = tot;
    *launches = n;
    return NDWT_OK;
}

int ndwt_plan_set_path(ndwt_plan* p, int path) {
    if (!p || (path != NDWT_PATH_AUTO && path != NDWT_PATH_GENERIC)) return fail(NDWT_ERR_INVALID_ARG, "bad plan/path");
    p->path = path;
    return NDWT_OK;
}

// test/tuning hook: grid sizing of the fused kernels (0 = default)
int ndwt_plan_set_tuning(ndwt_plan* p, int target_blocks, int force_zchunk) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    p->target_blocks = target_blocks > 0 ? target_blocks : 0;
    p->force_zchunk = force_zchunk > 0 ? force_zchunk : 0;
    return NDWT_OK;
}

// test/tuning hook (tools/: interleaved A/B runs): kernel variants and per-direction march chunks of the fused kernels; every
// variant computes the same values.  A negative argument leaves that setting as it is.  Nothing in the library reads the
// environment: a plan behaves the same whatever the caller's process has exported.
int ndwt_plan_set_variant(ndwt_plan* p, int variant_fwd, int variant_inv, int zchunk_fwd, int zchunk_inv, int fp64_fused) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    if (variant_fwd >= 0) p->variant_fwd = variant_fwd;
    if (variant_inv >= 0) p->variant_inv = variant_inv;
    if (zchunk_fwd >= 0) p->zchunk_dir[0] = zchunk_fwd;
    if (zchunk_inv >= 0) p->zchunk_dir[1] = zchunk_inv;
    if (fp64_fused >= 0) p->fp64_fused = fp64_fused ? 1 : 0;
    return NDWT_OK;
}

// test / tuning hook: 0 = ndwt_denoise materialises the level-1 detail bands (dec, thresholding in the synthesis loads, rec);
// 1 (default) = the fused level-1 kernel where it is the faster path (tap lengths <= 6); 2 = wherever it exists (8 taps too)
int ndwt_plan_set_fused_level1(ndwt_plan* p, int enable) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    p->fused_level1 = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return NDWT_OK;
}

int ndwt_plan_describe(const ndwt_plan* p, char* buf, int buflen) {
    if (!p || !buf || buflen < 1) return fail(NDWT_ERR_INVALID_ARG, "bad arguments");
    int Lp = 0;
    const char* s = "axis";
    const bool f3a = fused3_eligible(p, 1, &Lp, 0), f3s = fused3_eligible(p, 1, &Lp, 1);
    if (f3a && f3s) s = p->ndim == 3 ? "fused3d" : "axis+fused3d";
    else if (f3a) s = p->ndim == 3 ? "fused3d analysis, axis synthesis" : "axis+fused3d analysis, axis synthesis";
    else if (fused2_eligible(p, 1, &Lp)) s = "fused2d";
    snprintf(buf, (size_t)buflen, "%s", s);
    return NDWT_OK;
}

// band pitch in elements -> scalars; 0 = packed
static int pitch_scalars(const ndwt_plan* p, int64_t band_pitch, long long* bs) {
    *bs = band_pitch == 0 ? p->vol : (long long)band_pitch * p->comp;
    if (*bs < p->vol) return fail(NDWT_ERR_INVALID_ARG, "band pitch %lld is smaller than a band (%lld elements)", (long long)band_pitch, p->vol / p->comp);
    return NDWT_OK;
}

int ndwt_dec_pitched(ndwt_plan* p, const void* x, void* y, int64_t band_pitch, int level, void* stream) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!x || !y) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    long long bs = 0;
    rc = pitch_scalars(p, band_pitch, &bs);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    return p->dtype == NDWT_F32 ? dec_impl<float>(p, (const float*)x, (float*)y, bs, level, s)
                                : dec_impl<double>(p, (const double*)x, (double*)y, bs, level, s);
}

int ndwt_rec_pitched(ndwt_plan* p, const void* y, int64_t band_pitch, void* x, int level, void* stream) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!x || !y) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    long long bs = 0;
    rc = pitch_scalars(p, band_pitch, &bs);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    return p->dtype == NDWT_F32 ? rec_impl<float>(p, (const float*)y, bs, (float*)x, level, s)
                                : rec_impl<double>(p, (const double*)y, bs, (double*)x, level, s);
}

int ndwt_dec(ndwt_plan* p, const void* x, void* y, int level, void* stream) { return ndwt_dec_pitched(p, x, y, 0, level, stream); }
int ndwt_rec(ndwt_plan* p, const void* y, void* x, int level, void* stream) { return ndwt_rec_pitched(p, y, 0, x, level, stream); }

int64_t ndwt_band_pitch(const ndwt_plan* p) {
    if (!p) return 0;
    const long long skew = 256 / (long long)(p->esize * p->comp);   // 256 bytes, in elements
    return p->vol / p->comp + (skew > 0 ? skew : 1);
}

// Device staging of the host-pointer forms, owned by the plan and grown on demand: the gateway calls dec / rec with one configuration
// thousands of times (README.md:2), and a hipMalloc + hipFree of the 12 GB a 512^3 3-level transform stages costs milliseconds per call.
static int ensure_stage(ndwt_plan* p, int which, size_t bytes) {
    if (bytes <= p->stage_bytes[which]) return NDWT_OK;
    if (p->stage[which]) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(p->stage[which]));
        p->stage[which] = nullptr;
        p->stage_bytes[which] = 0;
    }
    hipError_t e = hipMalloc(&p->stage[which], bytes);
    if (e != hipSuccess) return fail(NDWT_ERR_ALLOC, "hipMalloc(%zu bytes) of the staging buffer failed: %s", bytes, hipGetErrorString(e));
    p->stage_bytes[which] = bytes;
    return NDWT_OK;
}

static int host_roundtrip(ndwt_plan* p, bool inverse, const void* src, void* dst, int level) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!src || !dst) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t bx = (size_t)p->vol * p->esize;
    const size_t by = bx * (size_t)ndwt_num_bands(p->ndim, level);
    rc = ensure_stage(p, 0, bx);
    if (rc == NDWT_OK) rc = ensure_stage(p, 1, by);
    if (rc) return rc;
    void *dx = p->stage[0], *dy = p->stage[1];
    hipError_t e = hipMemcpy(inverse ? dy : dx, src, inverse ? by : bx, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = inverse ? ndwt_rec(p, dy, dx, level, nullptr) : ndwt_dec(p, dx, dy, level, nullptr);
        if (rc == NDWT_OK) e = hipMemcpy(dst, inverse ? dx : dy, inverse ? bx : by, hipMemcpyDeviceToHost);
    }
    if (rc) return rc;
    if (e != hipSuccess) return fail(NDWT_ERR_HIP, "staging copy failed: %s", hipGetErrorString(e));
    return NDWT_OK;
}

int ndwt_plan_release_staging(ndwt_plan* p) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 2; ++i) {
        if (p->stage[i]) (void)hipFree(p->stage[i]);
        p->stage[i] = nullptr;
        p->stage_bytes[i] = 0;
    }
    return NDWT_OK;
}

// ---- device-resident coefficients (include/ndwt.h: ndwt_coef_*) ----
struct ndwt_coef {
    ndwt_plan* plan;
    int level;
    long long bands;
    long long pitch;                   // elements between bands (ndwt_band_pitch)
    void* dev;
    size_t bytes;
};

static int coef_check(const ndwt_plan* p, const ndwt_coef* c) {
    if (!p || !c) return fail(NDWT_ERR_INVALID_ARG, "null plan / coefficient handle");
    if (c->plan != p) return fail(NDWT_ERR_INVALID_ARG, "this coefficient handle belongs to another plan");
    return NDWT_OK;
}

int ndwt_coef_create(ndwt_plan* p, int level, ndwt_coef** out) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!out) return fail(NDWT_ERR_INVALID_ARG, "null output pointer");
    HIP_TRY(hipSetDevice(p->device));
    ndwt_coef* c = new ndwt_coef();
    c->plan = p;
    c->level = level;
    c->bands = ndwt_num_bands(p->ndim, level);
    c->pitch = ndwt_band_pitch(p);
    c->bytes = (size_t)c->bands * (size_t)c->pitch * (size_t)p->comp * p->esize;
    hipError_t e = hipMalloc(&c->dev, c->bytes);
    if (e != hipSuccess) {
        delete c;
        return fail(NDWT_ERR_ALLOC, "hipMalloc(%zu bytes) of a coefficient set failed: %s", (size_t)0 + c->bytes, hipGetErrorString(e));
    }
    p->live_coefs++;
    *out = c;
    return NDWT_OK;
}

int ndwt_coef_release(ndwt_coef* c) {
    if (!c) return NDWT_OK;
    (void)hipSetDevice(c->plan->device);
    (void)hipDeviceSynchronize();
    if (c->dev) (void)hipFree(c->dev);
    c->plan->live_coefs--;
    delete c;
    return NDWT_OK;
}

int ndwt_coef_info(const ndwt_coef* c, int* level, int64_t* bands, int64_t* band_pitch, void** dev_ptr) {
    if (!c) return fail(NDWT_ERR_INVALID_ARG, "null coefficient handle");
    if (level) *level = c->level;
    if (bands) *bands = c->bands;
    if (band_pitch) *band_pitch = c->pitch;
    if (dev_ptr) *dev_ptr = c->dev;
    return NDWT_OK;
}

// x (host) -> coefficients that STAY on the device: only the signal crosses PCIe (0.5 GB instead of 12.3 GB at 512^3, 3 levels)
int ndwt_coef_dec_host(ndwt_plan* p, const void* x_host, int level, ndwt_coef** coef) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!x_host || !coef) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    ndwt_coef* c = *coef;
    if (c && (c->plan != p || c->level != level)) return fail(NDWT_ERR_INVALID_ARG, "the handle passed for reuse holds another plan's / level's coefficients");
    const bool fresh = c == nullptr;
    if (fresh) {
        rc = ndwt_coef_create(p, level, &c);
        if (rc) return rc;
    }
    const size_t bx = (size_t)p->vol * p->esize;
    rc = ensure_stage(p, 0, bx);
    hipError_t e = hipSuccess;
    if (rc == NDWT_OK) e = hipMemcpy(p->stage[0], x_host, bx, hipMemcpyHostToDevice);
    if (rc == NDWT_OK && e == hipSuccess) rc = ndwt_dec_pitched(p, p->stage[0], c->dev, c->pitch, level, nullptr);
    if (rc == NDWT_OK && e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (rc != NDWT_OK || e != hipSuccess) {
        if (fresh) ndwt_coef_release(c);
        return rc ? rc : fail(NDWT_ERR_HIP, "staging copy failed: %s", hipGetErrorString(e));
    }
    *coef = c;
    return NDWT_OK;
}

int ndwt_coef_rec_host(ndwt_plan* p, const ndwt_coef* c, void* x_host) {
    int rc = coef_check(p, c);
    if (rc) return rc;
    if (!x_host) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t bx = (size_t)p->vol * p->esize;
    rc = ensure_stage(p, 0, bx);
    if (rc) return rc;
    rc = ndwt_rec_pitched(p, c->dev, c->pitch, p->stage[0], c->level, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(x_host, p->stage[0], bx, hipMemcpyDeviceToHost));
    return NDWT_OK;
}

int ndwt_coef_shrink(ndwt_plan* p, ndwt_coef* c, double threshold, int mode) {
    int rc = coef_check(p, c);
    if (rc) return rc;
    rc = ndwt_shrink_pitched(p, c->dev, c->pitch, c->level, threshold, mode, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return NDWT_OK;
}

// the coefficients in the reference's packed layout (band axis last, nd_dwt_mex.c:83) to / from host memory
int ndwt_coef_get_host(ndwt_plan* p, const ndwt_coef* c, void* y_host) {
    int rc = coef_check(p, c);
    if (rc) return rc;
    if (!y_host) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t band = (size_t)p->vol * p->esize, pitch = (size_t)c->pitch * (size_t)p->comp * p->esize;
    HIP_TRY(hipMemcpy2D(y_host, band, c->dev, pitch, band, (size_t)c->bands, hipMemcpyDeviceToHost));
    return NDWT_OK;
}

int ndwt_coef_put_host(ndwt_plan* p, int level, const void* y_host, ndwt_coef** coef) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!y_host || !coef) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    ndwt_coef* c = *coef;
    if (c && (c->plan != p || c->level != level)) return fail(NDWT_ERR_INVALID_ARG, "the handle passed for reuse holds another plan's / level's coefficients");
    const bool fresh = c == nullptr;
    if (fresh) {
        rc = ndwt_coef_create(p, level, &c);
        if (rc) return rc;
    }
    const size_t band = (size_t)p->vol * p->esize, pitch = (size_t)c->pitch * (size_t)p->comp * p->esize;
    hipError_t e = hipMemcpy2D(c->dev, pitch, y_host, band, band, (size_t)c->bands, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (fresh) ndwt_coef_release(c);
        return fail(NDWT_ERR_HIP, "upload of the coefficients failed: %s", hipGetErrorString(e));
    }
    *coef = c;
    return NDWT_OK;
}

int ndwt_dec_host(ndwt_plan* p, const void* x, void* y, int level) { return host_roundtrip(p, false, x, y, level); }
int ndwt_rec_host(ndwt_plan* p, const void* y, void* x, int level) { return host_roundtrip(p, true, y, x, level); }

// true when every synthesis level of this plan runs a kernel that can shrink its inputs on load (Inv3S / Inv2S)
static bool fused_shrink_capable(const ndwt_plan* p) {
    if (p->dilation != NDWT_DILATION_REFERENCE) return false;         // dilated levels take the per-axis kernels
    int Lp = 0;
    if (fused2_eligible(p, 1, &Lp)) return true;
    if (!fused3_eligible(p, 1, &Lp)) return false;
    return !(p->variant_inv == 3 && Lp == 8);                        // A/B variant 3 = the LDS synthesis kernel
}

// ---- consumers for iterative solvers (not in the reference: its users threshold in MATLAB) ----
static int shrink_check(const ndwt_plan* p, int level, double thr, int mode) {
    int rc = check_level(p, level);
    if (rc) return rc;
    if (!(thr >= 0.0)) return fail(NDWT_ERR_INVALID_ARG, "threshold must be >= 0");
    if (mode != NDWT_SHRINK_SOFT && mode != NDWT_SHRINK_HARD) return fail(NDWT_ERR_INVALID_ARG, "mode must be NDWT_SHRINK_SOFT or NDWT_SHRINK_HARD");
    return NDWT_OK;
}

int ndwt_shrink_pitched(ndwt_plan* p, void* y, int64_t band_pitch, int level, double threshold, int mode, void* stream) {
    int rc = shrink_check(p, level, threshold, mode);
    if (rc) return rc;
    if (!y) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    long long bs = 0;
    rc = pitch_scalars(p, band_pitch, &bs);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(p->device));
    return p->dtype == NDWT_F32 ? shrink_impl<float>(p, (float*)y, bs, level, threshold, mode, (hipStream_t)stream)
                                : shrink_impl<double>(p, (double*)y, bs, level, threshold, mode, (hipStream_t)stream);
}
int ndwt_shrink(ndwt_plan* p, void* y, int level, double threshold, int mode, void* stream) {
    return ndwt_shrink_pitched(p, y, 0, level, threshold, mode, stream);
}

// ---- level 1 of a denoising step without its detail bands in memory (Den3, ndwt_device.h) ----
// Float, real, 3-D, reference dilation, the same tap length L <= 8 on every axis, rows of whole 16-byte groups.
static bool den3_eligible(const ndwt_plan* p, int* Lp_out) {
    if (!p->fused_level1 || p->dtype != NDWT_F32 || p->complexity != NDWT_REAL || p->ndim != 3 || p->dilation != NDWT_DILATION_REFERENCE)
        return false;
    int Lp = 0;
    // measured, 512^3, 3 levels, ndwt_denoise with / without the fused level 1: db1 4.49 / 5.27 ms, db2 5.10 / 5.67, db3 5.72 / 6.09,
    // db4 6.30 / 6.13 -- with 8 taps the recomputation (2.3x the arithmetic of the synthesis kernel, 67 % VALU-busy) costs more than the
    // 13 volume transfers it removes, so 8 taps take the kernel only when asked to (ndwt_plan_set_fused_level1(plan, 2))
    if (!fused3_eligible(p, 1, &Lp) || Lp > (p->fused_level1 >= 2 ? 8 : 6) || !inv3y_plan_ok(p, Lp)) return false;
    for (int ax = 0; ax < 3; ++ax)
        if (p->filt[ax].len != Lp) return false;
    if (p->dims[0] % 4 != 0) return false;
    *Lp_out = Lp;
    return true;
}

static int den3_taps(ndwt_plan* p, int Lp) {
    if (p->taps_den) return NDWT_OK;
    const FusedTapsD ts = fused_taps(p, Lp, true), ta = fused_taps(p, Lp, false);
    std::vector<float> h;                                 // TapsDen<float, Lp>: Taps3Y (lo[3][L], hi[3][L], xplo[L+1][2], xphi[L+1][2]), alo[3][L], azp[L][2], axp[L+1][2]
    for (int ax = 0; ax < 3; ++ax) for (int j = 0; j < Lp; ++j) h.push_back((float)ts.lo[ax][j]);
    for (int ax = 0; ax < 3; ++ax) for (int j = 0; j < Lp; ++j) h.push_back((float)ts.hi[ax][j]);
    for (int hi = 0; hi < 2; ++hi)
        for (int k = 0; k <= Lp; ++k)
            for (int hh = 0; hh < 2; ++hh) {
                const int j = k - hh;
                h.push_back((j >= 0 && j < Lp) ? (float)(hi ? ts.hi[0][j] : ts.lo[0][j]) : 0.0f);
            }
    for (int ax = 0; ax < 3; ++ax) for (int j = 0; j < Lp; ++j) h.push_back((float)ta.lo[ax][j]);
    for (int j = 0; j < Lp; ++j) { h.push_back((float)ta.lo[2][j]); h.push_back((float)ta.hi[2][j]); }
    for (int k = 0; k <= Lp; ++k)                         // axp[L+1][2]: (alo_x[k], alo_x[k-1])
        for (int hh = 0; hh < 2; ++hh) {
            const int j = k - hh;
            h.push_back((j >= 0 && j < Lp) ? (float)ta.lo[0][j] : 0.0f);
        }
    // the kernel derives the analysis high-pass taps of x and y from the low-pass ones: ahi[j] = (-1)^j alo[L-1-j]
    for (int ax = 0; ax < 2; ++ax)
        for (int j = 0; j < Lp; ++j)
            if ((float)ta.hi[ax][j] != ((j % 2) ? -1.0f : 1.0f) * (float)ta.lo[ax][Lp - 1 - j])
                return fail(NDWT_ERR_UNSUPPORTED, "internal: analysis taps are not a mirrored pair");
    hipError_t e = hipMalloc(&p->taps_den, h.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->taps_den, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->taps_den) { (void)hipFree(p->taps_den); p->taps_den = nullptr; }
        return fail(NDWT_ERR_ALLOC, "uploading the tap table of the fused level-1 kernel failed: %s", hipGetErrorString(e));
    }
    return NDWT_OK;
}

// kind 0: approximation band of one analysis level (x -> out);  kind 1: Den3 (x, approximation -> out)
static int den3_launch(ndwt_plan* p, int kind, int Lp, const float* x, const float* apx, float* out, hipStream_t s) {
    Fused3Args<float> a;
    memset(&a, 0, sizeof a);
    a.n1 = (int)p->dims[0]; a.n2 = (int)p->dims[1]; a.n3 = (int)p->dims[2];
    a.nbatch = 1;
    a.z_wrap = 1;
    a.in[0] = x; a.in[1] = apx;
    a.out[0] = out;
    a.in_bstride = a.out_bstride = p->vol;
    if (kind == 1) {
        a.shrink_thr = (float)p->shrink_thr;
        a.shrink_mask = 0xFE;
        a.shrink_hard = p->shrink_mode == 2;
    }
    // one workgroup per CU (1024 threads); Den3's march starts 2 (L - 1) planes before its first output plane
    const bool small_tile = kind == 0 && p->variant_fwd != 2 && a.n2 > 16;   // the band-0 analysis on 64 x 16 tiles, 3 workgroups per CU (0.36 vs 0.42 ms; variant 2: A/B)
    if (small_tile) fused3_geometry(a, 64, 16, Lp, p->target_blocks > 0 ? p->target_blocks : p->num_cus * 3, p->force_zchunk);
    else fused3_geometry(a, 64, 32, kind == 1 ? 2 * Lp - 1 : Lp, p->target_blocks > 0 ? p->target_blocks : p->num_cus, p->force_zchunk);
    float* outs[1] = {out};
    a.nt = nt_store_ok<float>(a.rs, a.plane, p->vol, outs, 1);
    prof_begin(p, kind == 1 ? NDWT_KERNEL_FUSED_SYNTHESIS : NDWT_KERNEL_FUSED_ANALYSIS, s);
    const bool vec4 = aligned_vec4<float>(x) && aligned_vec4<float>(out);
    int rc = kind == 1 ? launch_den3_f32(a, Lp, p->taps_den, s) : launch_fwd3_low_f32(a, Lp, vec4, p->taps_dev[0], s);
    prof_end(p, s, rc);
    if (rc == -1) return fail(NDWT_ERR_UNSUPPORTED, "fused level-1 kernel not instantiated for tap length %d", Lp);
    if (rc == -2) return fail(NDWT_ERR_UNSUPPORTED, "internal: launch geometry does not match the 64 x 32 tile");
    if (rc != 0) return fail(NDWT_ERR_HIP, "fused level-1 kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return NDWT_OK;
}

static int ensure_coef(ndwt_plan* p, size_t need) {
    if (need <= p->coef_bytes) return NDWT_OK;
    if (p->coef) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipFree(p->coef));
        p->coef = nullptr;
        p->coef_bytes = 0;
    }
    hipError_t e = hipMalloc(&p->coef, need);
    if (e != hipSuccess) return fail(NDWT_ERR_ALLOC, "hipMalloc(%zu bytes) for the coefficient scratch failed: %s", need, hipGetErrorString(e));
    p->coef_bytes = need;
    return NDWT_OK;
}

// dec -> shrink -> rec with level 1 fused: x -> approximation of level 1 (band 0 only) -> levels 2 .. `level` as a (level - 1)-level
// transform of that band (the reference applies the same filters at every level, nd_dwt_3D.m:178-186) with the thresholding in the
// synthesis kernels' loads -> Den3(x, reconstructed approximation).  Volumes moved at level 1: 5 instead of 18.
static int denoise_fused_level1(ndwt_plan* p, int Lp, const float* x, float* out, int level, double threshold, int mode, hipStream_t s) {
    int rc = den3_taps(p, Lp);
    if (rc) return rc;
    if (!p->den_a1) {
        hipError_t e = hipMalloc(&p->den_a1, (size_t)p->vol * sizeof(float) + kApproxSkew);
        if (e != hipSuccess) return fail(NDWT_ERR_ALLOC, "hipMalloc of the level-1 approximation scratch failed: %s", hipGetErrorString(e));
    }
    float* a1 = (float*)((char*)p->den_a1 + kApproxSkew);   // level-1 approximation, then its reconstruction (256 B off the alignment, like the ping-pong scratch)
    p->shrink_mode = mode == NDWT_SHRINK_HARD ? 2 : 1;
    p->shrink_thr = threshold;
    rc = den3_launch(p, 0, Lp, x, nullptr, a1, s);
    if (rc == NDWT_OK && level > 1) {
        const int64_t pitch = ndwt_band_pitch(p);
        rc = ensure_coef(p, (size_t)pitch * p->esize * (size_t)ndwt_num_bands(3, level - 1));
        if (rc == NDWT_OK) rc = dec_impl<float>(p, a1, (float*)p->coef, pitch, level - 1, s);
        if (rc == NDWT_OK) rc = rec_impl<float>(p, (const float*)p->coef, pitch, a1, level - 1, s);   // (thresholding in the band loads)
    }
    if (rc == NDWT_OK) rc = den3_launch(p, 1, Lp, x, a1, out, s);
    p->shrink_mode = 0;
    return rc;
}

int ndwt_denoise(ndwt_plan* p, const void* x, void* out, int level, double threshold, int mode, void* stream) {
    int rc = shrink_check(p, level, threshold, mode);
    if (rc) return rc;
    if (!x || !out) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    HIP_TRY(hipSetDevice(p->device));
    {
        // the finest level without its detail bands in memory, where the fused level-1 kernel applies (it reads x around every
        // output voxel while other workgroups write `out`: not in place)
        int Lp1 = 0;
        const char *xb = (const char*)x, *ob = (const char*)out;
        const size_t nbytes = (size_t)p->vol * p->esize;
        const bool disjoint = xb + nbytes <= ob || ob + nbytes <= xb;
        if (den3_eligible(p, &Lp1) && disjoint && aligned_vec4<float>(x) && aligned_vec4<float>(out))
            return denoise_fused_level1(p, Lp1, (const float*)x, (float*)out, level, threshold, mode, (hipStream_t)stream);
    }
    // the scratch coefficients are pitched (ndwt_band_pitch): nobody else reads them
    const int64_t pitch = ndwt_band_pitch(p);
    rc = ensure_coef(p, (size_t)pitch * p->comp * p->esize * (size_t)ndwt_num_bands(p->ndim, level));
    if (rc) return rc;
    rc = ndwt_dec_pitched(p, x, p->coef, pitch, level, stream);
    if (rc) return rc;
    if (fused_shrink_capable(p)) {
        // every level is reconstructed by a lane-shift kernel: the detail bands are thresholded in registers as that
        // kernel loads them, and the separate pass (a read and a write of every detail band) disappears
        p->shrink_mode = mode == NDWT_SHRINK_HARD ? 2 : 1;
        p->shrink_thr = threshold;
        rc = ndwt_rec_pitched(p, p->coef, pitch, out, level, stream);
        p->shrink_mode = 0;
        return rc;
    }
    rc = ndwt_shrink_pitched(p, p->coef, pitch, level, threshold, mode, stream);
    if (rc == NDWT_OK) rc = ndwt_rec_pitched(p, p->coef, pitch, out, level, stream);
    return rc;
}

int ndwt_denoise_host(ndwt_plan* p, const void* x, void* out, int level, double threshold, int mode) {
    int rc = shrink_check(p, level, threshold, mode);
    if (rc) return rc;
    if (!x || !out) return fail(NDWT_ERR_INVALID_ARG, "null data pointer");
    HIP_TRY(hipSetDevice(p->device));
    const size_t bx = (size_t)p->vol * p->esize;
    rc = ensure_stage(p, 0, 2 * bx);                      // the signal and the result, side by side (kept across calls)
    if (rc) return rc;
    void *dx = p->stage[0], *dout = (char*)p->stage[0] + bx;
    hipError_t e = hipMemcpy(dx, x, bx, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = ndwt_denoise(p, dx, dout, level, threshold, mode, nullptr);
        if (rc == NDWT_OK) e = hipMemcpy(out, dout, bx, hipMemcpyDeviceToHost);
    }
    if (rc) return rc;
    if (e != hipSuccess) return fail(NDWT_ERR_HIP, "staging copy failed: %s", hipGetErrorString(e));
    return NDWT_OK;
}

// Split complex (separate real / imaginary arrays: mxGetPr / mxGetPi of nd_dwt_mex.c:55-58).  The filters are
// real, so the complex transform is the real transform of each part: a REAL plan is run once per part.
static int split_check(const ndwt_plan* p) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    if (p->comp != 1) return fail(NDWT_ERR_INVALID_ARG, "split-complex entry points take a plan created with NDWT_REAL");
    return NDWT_OK;
}

int ndwt_dec_split(ndwt_plan* p, const void* x_re, const void* x_im, void* y_re, void* y_im, int level, void* stream) {
    int rc = split_check(p);
    if (rc) return rc;
    if ((x_im == nullptr) != (y_im == nullptr)) return fail(NDWT_ERR_INVALID_ARG, "imaginary input and output must both be given or both be NULL");
    rc = ndwt_dec(p, x_re, y_re, level, stream);
    if (rc == NDWT_OK && x_im) rc = ndwt_dec(p, x_im, y_im, level, stream);
    return rc;
}

int ndwt_rec_split(ndwt_plan* p, const void* y_re, const void* y_im, void* x_re, void* x_im, int level, void* stream) {
    int rc = split_check(p);
    if (rc) return rc;
    if ((x_im == nullptr) != (y_im == nullptr)) return fail(NDWT_ERR_INVALID_ARG, "imaginary input and output must both be given or both be NULL");
    rc = ndwt_rec(p, y_re, x_re, level, stream);
    if (rc == NDWT_OK && y_im) rc = ndwt_rec(p, y_im, x_im, level, stream);
    return rc;
}

int ndwt_dec_split_host(ndwt_plan* p, const void* x_re, const void* x_im, void* y_re, void* y_im, int level) {
    int rc = split_check(p);
    if (rc) return rc;
    if ((x_im == nullptr) != (y_im == nullptr)) return fail(NDWT_ERR_INVALID_ARG, "imaginary input and output must both be given or both be NULL");
    rc = host_roundtrip(p, false, x_re, y_re, level);
    if (rc == NDWT_OK && x_im) rc = host_roundtrip(p, false, x_im, y_im, level);
    return rc;
}

int ndwt_rec_split_host(ndwt_plan* p, const void* y_re, const void* y_im, void* x_re, void* x_im, int level) {
    int rc = split_check(p);
    if (rc) return rc;
    if ((x_im == nullptr) != (y_im == nullptr)) return fail(NDWT_ERR_INVALID_ARG, "imaginary input and output must both be given or both be NULL");
    rc = host_roundtrip(p, true, y_re, x_re, level);
    if (rc == NDWT_OK && y_im) rc = host_roundtrip(p, true, y_im, x_im, level);
    return rc;
}

int ndwt_plan_slab_fast(const ndwt_plan* p) {
    int Lp = 0;
    return p && p->ndim == 3 && fused3_eligible(p, 1, &Lp) && p->filt[2].len == Lp ? 1 : 0;
}

int ndwt_slab_halo(const ndwt_plan* p, int stride, int64_t* ab, int64_t* aa, int64_t* sb, int64_t* sa) {
    if (!p || stride < 1) return fail(NDWT_ERR_INVALID_ARG, "bad plan/stride");
    const int L = p->filt[p->ndim - 1].len;
    if (ab) *ab = (int64_t)(L / 2 - 1) * stride;
    if (aa) *aa = (int64_t)(L / 2) * stride;
    if (sb) *sb = (int64_t)(L / 2) * stride;
    if (sa) *sa = (int64_t)(L / 2 - 1) * stride;
    return NDWT_OK;
}

int ndwt_analysis_level_slab(ndwt_plan* p, const void* in, void* const* out, int stride, void* stream) {
    if (!p || !in || !out || stride < 1) return fail(NDWT_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    return p->dtype == NDWT_F32 ? analysis_level<float>(p, (const float*)in, (float* const*)out, stride, true, s)
                                : analysis_level<double>(p, (const double*)in, (double* const*)out, stride, true, s);
}

int ndwt_synthesis_level_slab(ndwt_plan* p, const void* const* in, void* out, int stride, void* stream) {
    if (!p || !in || !out || stride < 1) return fail(NDWT_ERR_INVALID_ARG, "bad arguments");
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t s = (hipStream_t)stream;
    return p->dtype == NDWT_F32 ? synthesis_level<float>(p, (const float* const*)in, (float*)out, stride, true, s)
                                : synthesis_level<double>(p, (const double* const*)in, (double*)out, stride, true, s);
}

int ndwt_analysis_level_slab_split(ndwt_plan* p, const void* in_local, const void* halo_before, const void* halo_after,
                                   void* const* out, int stride, void* stream) {
    int Lp = 0;
    int rc = slab_fast_ok(p, stride, &Lp);
    if (rc) return rc;
    if (!in_local || !out || (Lp > 2 && !halo_before) || !halo_after) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    if (!halo_before) halo_before = halo_after;   // db1: no plane before the slab is needed
    return p->dtype == NDWT_F32 ? slab_split_impl<float>(p, Lp, in_local, halo_before, halo_after, out, (hipStream_t)stream)
                                : slab_split_impl<double>(p, Lp, in_local, halo_before, halo_after, out, (hipStream_t)stream);
}

int ndwt_synthesis_level_slab_ext(ndwt_plan* p, const void* const* in_local, void* out_ext, int stride, void* stream) {
    int Lp = 0;
    if (p && p->ndim == 4) {                             // t-sharded 4-D: 3-D part per frame, zero-extended t-axis pass
        if (stride != 1 || !fused3_eligible(p, 1, &Lp))
            return fail(NDWT_ERR_UNSUPPORTED, "the zero-extended 4-D slab synthesis needs a plan on the fused 3-D kernels (stride 1)");
        if (!in_local || !out_ext) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
        HIP_TRY(hipSetDevice(p->device));
        return p->dtype == NDWT_F32 ? slab_ext4_impl<float>(p, Lp, in_local, out_ext, (hipStream_t)stream)
                                    : slab_ext4_impl<double>(p, Lp, in_local, out_ext, (hipStream_t)stream);
    }
    int rc = slab_fast_ok(p, stride, &Lp);
    if (rc) return rc;
    if (!in_local || !out_ext) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    HIP_TRY(hipSetDevice(p->device));
    return p->dtype == NDWT_F32 ? slab_ext_impl<float>(p, Lp, in_local, out_ext, (hipStream_t)stream)
                                : slab_ext_impl<double>(p, Lp, in_local, out_ext, (hipStream_t)stream);
}

int ndwt_analysis_level_slab_part(ndwt_plan* p, const void* in_local, const void* halo_before, const void* halo_after,
                                  void* const* out, int stride, int64_t n_planes, void* stream) {
    int Lp = 0;
    int rc = slab_fast_ok(p, stride, &Lp);
    if (rc) return rc;
    if (!in_local || !out || (Lp > 2 && !halo_before) || !halo_after) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    if (n_planes < 1 || n_planes > INT32_MAX) return fail(NDWT_ERR_INVALID_ARG, "n_planes must be >= 1");
    HIP_TRY(hipSetDevice(p->device));
    if (!halo_before) halo_before = halo_after;
    return p->dtype == NDWT_F32
               ? slab_analysis_part_impl<float>(p, Lp, in_local, halo_before, halo_after, out, n_planes, (hipStream_t)stream)
               : slab_analysis_part_impl<double>(p, Lp, in_local, halo_before, halo_after, out, n_planes, (hipStream_t)stream);
}

int ndwt_synthesis_level_slab_runs(ndwt_plan* p, const void* const* in_local, int64_t n_in, int64_t e0, int64_t e_stride,
                                   int64_t n_runs, int64_t n_out, void* out, int stride, void* stream) {
    int Lp = 0;
    int rc = slab_fast_ok(p, stride, &Lp);
    if (rc) return rc;
    if (!in_local || !out) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    if (n_in < 1 || e0 < 0 || n_out < 1 || n_runs < 1 || e_stride < 0 || n_in > INT32_MAX ||
        e0 + (n_runs - 1) * e_stride + n_out > n_in + Lp - 1)
        return fail(NDWT_ERR_INVALID_ARG, "every run of output planes must lie inside the %lld planes of the zero-extended result",
                    (long long)(n_in + Lp - 1));
    HIP_TRY(hipSetDevice(p->device));
    return p->dtype == NDWT_F32
               ? slab_synthesis_runs_impl<float>(p, Lp, in_local, n_in, e0, e_stride, n_runs, n_out, out, (hipStream_t)stream)
               : slab_synthesis_runs_impl<double>(p, Lp, in_local, n_in, e0, e_stride, n_runs, n_out, out, (hipStream_t)stream);
}

int ndwt_synthesis_level_slab_part(ndwt_plan* p, const void* const* in_local, int64_t n_in, int64_t e0, int64_t n_out,
                                   void* out, int stride, void* stream) {
    return ndwt_synthesis_level_slab_runs(p, in_local, n_in, e0, 0, 1, n_out, out, stride, stream);
}

int ndwt_analysis_level_slab_runs(ndwt_plan* p, const void* in_with_halo, void* const* out, int stride, int64_t n_planes,
                                  int64_t n_runs, int64_t run_stride, void* stream) {
    int Lp = 0;
    int rc = slab_fast_ok(p, stride, &Lp);
    if (rc) return rc;
    if (!in_with_halo || !out) return fail(NDWT_ERR_INVALID_ARG, "null pointer");
    if (n_planes < 1 || n_planes > INT32_MAX || n_runs < 1 || run_stride < 0) return fail(NDWT_ERR_INVALID_ARG, "bad run geometry");
    HIP_TRY(hipSetDevice(p->device));
    return p->dtype == NDWT_F32
               ? slab_analysis_runs_impl<float>(p, Lp, in_with_halo, out, n_planes, n_runs, run_stride, (hipStream_t)stream)
               : slab_analysis_runs_impl<double>(p, Lp, in_with_halo, out, n_planes, n_runs, run_stride, (hipStream_t)stream);
}

#ifdef NDWT_STAMPS
// diagnostic builds only (tools/stamps_inv.py): 4 cycle sums per wave of every workgroup of the next synthesis launches
int ndwt_plan_set_stamps(ndwt_plan* p, void* dev_buffer) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    p->stamps = (long long*)dev_buffer;
    return NDWT_OK;
}
#endif

const char* ndwt_last_error(void) { return g_last_error.c_str(); }
int ndwt_slab_segments(ndwt_plan* p, int op, int nseg, void* const* dst, const void* const* src, const int64_t* count, void* stream) {
    if (!p) return fail(NDWT_ERR_INVALID_ARG, "null plan");
    if (op != NDWT_SEG_COPY && op != NDWT_SEG_ADD) return fail(NDWT_ERR_INVALID_ARG, "op must be NDWT_SEG_COPY or NDWT_SEG_ADD");
    if (nseg < 0 || nseg > NDWT_MAX_SEGMENTS) return fail(NDWT_ERR_INVALID_ARG, "at most %d runs per call", NDWT_MAX_SEGMENTS);
    if (nseg == 0) return NDWT_OK;
    if (!dst || !src || !count) return fail(NDWT_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < nseg; ++i)
        if (!dst[i] || !src[i] || count[i] < 0) return fail(NDWT_ERR_INVALID_ARG, "run %d: null pointer or negative count", i);
    HIP_TRY(hipSetDevice(p->device));
    const int rc = p->dtype == NDWT_F32 ? segments_launch<float>(op, nseg, dst, src, count, (hipStream_t)stream)
                                        : segments_launch<double>(op, nseg, dst, src, count, (hipStream_t)stream);
    if (rc != 0) return fail(NDWT_ERR_HIP, "segment kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return NDWT_OK;
}

const char* ndwt_version(void) { return "ndwt-hip 0.1 (gfx950)"; }

}  // extern "C"
