// ndwt_multi.hip -- single-process, multi-device plan: ONE host thread drives the GPUs of a node.
//
// The reference's host is one MATLAB process calling one gateway (Functions/nd_dwt_3D.m:161,225 -> mex/nd_dwt_mex.c:8): a
// drop-in multi-GPU path has to live behind that single call.  This file shards the volume in slabs on its outermost axis
// over `devices[]` (a device may be listed several times: independent slabs and streams on one GPU -- how the path is tested on
// a one-GPU machine), runs the level loop with the slab entry points of include/ndwt.h and moves the halo planes between
// slabs with asynchronous device-to-device copies (peer copies over xGMI when the slabs live on different GPUs).  Streams
// are ordered with events only; the host thread blocks once, at the end of a call.
//
// Exchange scheme: gather for both directions (analysis: halo planes of the approximation band; synthesis: halo planes of all
// 2^d bands), which reproduces the single-device result bit for bit with every kernel path of the library.  The one-process-
// per-GPU driver (sharded.py) additionally has the scatter-add synthesis (1 band of exchange) and the overlap of exchange and
// interior planes; here the halo copies of a level run on the consumer's stream right before its launch.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ndwt.h"

namespace {

struct Slab {
    int device;
    long long z0, n;               // planes [z0, z0 + n) of the outer axis
    ndwt_plan* plan;
    hipStream_t stream;
    hipEvent_t ready[2];           // "approximation buffer k of this slab is complete" (k = level parity)
    char* approx[2];               // [halo_max | n | halo_max] planes each: approximation band between levels / x / result
    char* coef;                    // all bands of the slab: nbt_max * n planes
    char* gather;                  // synthesis input: 2^d bands * (n + L - 1) planes
};

}  // namespace

struct ndwt_mplan {
    int ndim, dtype, complexity, max_level, nb;
    long long dims[NDWT_MAX_DIMS];
    size_t plane_bytes;            // bytes of one plane of the outer axis (one band)
    long long halo_max;            // planes of margin in the approximation buffers
    int L_outer;
    int dilation;
    std::vector<Slab> slabs;
    std::string err;
};

static thread_local std::string g_merr;

static int mfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_merr = buf;
    return code;
}
#define MHIP(expr)                                                                                         \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return mfail(NDWT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)
#define MTRY(expr)                                                     \
    do {                                                               \
        int rc_ = (expr);                                              \
        if (rc_ != NDWT_OK) return mfail(rc_, "%s", ndwt_last_error()); \
    } while (0)

static long long stride_of(const ndwt_mplan* mp, int lev) { return mp->dilation == NDWT_DILATION_ATROUS ? (1LL << (lev - 1)) : 1LL; }

// copy `count` planes starting at GLOBAL plane g (periodic) of a per-slab plane array into dst on slab `to`, on `to`'s stream.
// plane_ptr(slab, local plane) gives the source address; wait_parity >= 0: wait for the source slab's ready[wait_parity] first.
template <class SrcFn>
static int copy_planes(ndwt_mplan* mp, Slab& to, char* dst, long long g, long long count, SrcFn plane_ptr, int wait_parity) {
    const long long N = mp->dims[mp->ndim - 1];
    long long done = 0;
    while (done < count) {
        long long gp = ((g + done) % N + N) % N;
        Slab* src = nullptr;
        for (auto& s : mp->slabs)
            if (gp >= s.z0 && gp < s.z0 + s.n) src = &s;
        if (!src) return mfail(NDWT_ERR_INVALID_ARG, "plane %lld has no owner", gp);
        long long run = src->z0 + src->n - gp;
        if (run > count - done) run = count - done;
        if (wait_parity >= 0 && src != &to) MHIP(hipStreamWaitEvent(to.stream, src->ready[wait_parity], 0));
        const char* sp = plane_ptr(*src, gp - src->z0);
        char* dp = dst + (size_t)done * mp->plane_bytes;
        if (src->device == to.device) MHIP(hipMemcpyAsync(dp, sp, (size_t)run * mp->plane_bytes, hipMemcpyDeviceToDevice, to.stream));
        else MHIP(hipMemcpyPeerAsync(dp, to.device, sp, src->device, (size_t)run * mp->plane_bytes, to.stream));
        done += run;
    }
    return NDWT_OK;
}

extern "C" {

int ndwt_mplan_create(ndwt_mplan** out, int ndim, const int64_t* dims, const char* const* wnames, int dtype, int complexity,
                      int pres_l2_norm, int dilation, int max_level, const int* devices, int ndev) {
    if (!out) return mfail(NDWT_ERR_INVALID_ARG, "null plan pointer");
    *out = nullptr;
    if (ndim < 2 || ndim > NDWT_MAX_DIMS || !dims || !wnames || !devices || ndev < 1)
        return mfail(NDWT_ERR_INVALID_ARG, "multi-device plans shard the outermost of 2..4 axes over ndev >= 1 devices");
    const long long N = dims[ndim - 1];
    if (ndev > N) return mfail(NDWT_ERR_INVALID_ARG, "more slabs (%d) than planes (%lld)", ndev, N);
    if (max_level < 1) return mfail(NDWT_ERR_INVALID_ARG, "max_level must be >= 1");
    ndwt_mplan* mp = new ndwt_mplan();
    mp->ndim = ndim; mp->dtype = dtype; mp->complexity = complexity; mp->max_level = max_level; mp->nb = 1 << ndim;
    mp->dilation = dilation;
    for (int a = 0; a < ndim; ++a) mp->dims[a] = dims[a];
    size_t pb = (dtype == NDWT_F32 ? 4 : 8) * (complexity == NDWT_COMPLEX_INTERLEAVED ? 2 : 1);
    for (int a = 0; a + 1 < ndim; ++a) pb *= (size_t)dims[a];
    mp->plane_bytes = pb;
    double lo[NDWT_MAX_TAPS], hi[NDWT_MAX_TAPS];
    int L = 0;
    if (ndwt_wave_filters(wnames[ndim - 1], lo, hi, &L) != NDWT_OK) { delete mp; return mfail(NDWT_ERR_UNKNOWN_WAVELET, "Unknown Wavelet Name"); }
    mp->L_outer = L;
    const long long smax = dilation == NDWT_DILATION_ATROUS ? (1LL << (max_level - 1)) : 1LL;
    mp->halo_max = (long long)(L / 2) * smax;
    const long long nbt = (long long)ndwt_num_bands(ndim, max_level);
    std::vector<int64_t> ld(dims, dims + ndim);
    for (int i = 0; i < ndev; ++i) {
        Slab s;
        memset(&s, 0, sizeof s);
        s.device = devices[i];
        s.z0 = (long long)i * N / ndev;
        s.n = (long long)(i + 1) * N / ndev - s.z0;
        mp->slabs.push_back(s);
    }
    int rc = NDWT_OK;
    for (auto& s : mp->slabs) {
        ld[ndim - 1] = s.n;
        if ((rc = ndwt_plan_create_slab(&s.plan, ndim, ld.data(), N, wnames, dtype, complexity, pres_l2_norm, dilation, 1, s.device)) != NDWT_OK) break;
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&s.ready[k], hipEventDisableTiming);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipMalloc((void**)&s.approx[k], (size_t)(s.n + 2 * mp->halo_max) * pb);
        if (e == hipSuccess) e = hipMalloc((void**)&s.coef, (size_t)(nbt * s.n) * pb);
        if (e == hipSuccess) e = hipMalloc((void**)&s.gather, (size_t)((long long)mp->nb * (s.n + (long long)(L - 1) * smax)) * pb);
        if (e != hipSuccess) { rc = mfail(NDWT_ERR_ALLOC, "device %d: %s", s.device, hipGetErrorString(e)); break; }
        for (auto& o : mp->slabs)                        // peer access where the runtime offers it (same-device pairs need none)
            if (o.device != s.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, s.device, o.device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(o.device, 0);
                (void)hipGetLastError();                 // "already enabled" is not an error
            }
    }
    if (rc != NDWT_OK) {
        std::string keep = rc == NDWT_ERR_ALLOC ? g_merr : std::string(ndwt_last_error());
        ndwt_mplan_destroy(mp);
        return mfail(rc, "%s", keep.c_str());
    }
    *out = mp;
    return NDWT_OK;
}

int ndwt_mplan_destroy(ndwt_mplan* mp) {
    if (!mp) return NDWT_OK;
    for (auto& s : mp->slabs) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.plan) ndwt_plan_destroy(s.plan);
        for (int k = 0; k < 2; ++k) {
            if (s.approx[k]) (void)hipFree(s.approx[k]);
            if (s.ready[k]) (void)hipEventDestroy(s.ready[k]);
        }
        if (s.coef) (void)hipFree(s.coef);
        if (s.gather) (void)hipFree(s.gather);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete mp;
    return NDWT_OK;
}

int ndwt_mplan_num_slabs(const ndwt_mplan* mp) { return mp ? (int)mp->slabs.size() : -1; }

int ndwt_mplan_slab(const ndwt_mplan* mp, int idx, int* device, int64_t* first_plane, int64_t* planes) {
    if (!mp || idx < 0 || idx >= (int)mp->slabs.size()) return mfail(NDWT_ERR_INVALID_ARG, "bad slab index");
    if (device) *device = mp->slabs[idx].device;
    if (first_plane) *first_plane = mp->slabs[idx].z0;
    if (planes) *planes = mp->slabs[idx].n;
    return NDWT_OK;
}

static int mcheck(const ndwt_mplan* mp, int level) {
    if (!mp) return mfail(NDWT_ERR_INVALID_ARG, "null plan");
    if (level < 1 || level > mp->max_level) return mfail(NDWT_ERR_INVALID_ARG, "level %d outside 1..max_level=%d of this plan", level, mp->max_level);
    return NDWT_OK;
}

// whole-volume host arrays in, whole-volume host arrays out (the layout the MATLAB gateway holds): x is prod(dims) elements,
// y prod(dims) * ndwt_num_bands(ndim, level), band-planar.  Blocks until the result is in y.
int ndwt_mdec_host(ndwt_mplan* mp, const void* x_host, void* y_host, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_host || !y_host) return mfail(NDWT_ERR_INVALID_ARG, "null data pointer");
    const size_t pb = mp->plane_bytes;
    const long long N = mp->dims[mp->ndim - 1], H = mp->halo_max;
    const int nb = mp->nb;
    const long long nbt = (long long)ndwt_num_bands(mp->ndim, level);
    // the signal into the middle of approx[1] (the input of level 1: parity of level 0 is 0 -> buffer index (lev-1)&1 ^ 1 ...)
    // buffer that level `lev` READS: in_buf(lev) = approx[(lev - 1) & 1 ^ 1]; it WRITES its approximation into approx[(lev - 1) & 1]
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        MHIP(hipMemcpyAsync(s.approx[1] + (size_t)H * pb, (const char*)x_host + (size_t)s.z0 * pb, (size_t)s.n * pb, hipMemcpyHostToDevice, s.stream));
        MHIP(hipEventRecord(s.ready[1], s.stream));
    }
    for (int lev = 1; lev <= level; ++lev) {
        const long long st = stride_of(mp, lev);
        const long long ab = (long long)(mp->L_outer / 2 - 1) * st, aa = (long long)(mp->L_outer / 2) * st;
        const int rd = ((lev - 1) & 1) ^ 1, wr = (lev - 1) & 1;
        for (auto& s : mp->slabs) {
            MHIP(hipSetDevice(s.device));
            char* mid = s.approx[rd] + (size_t)H * pb;
            auto src = [&](Slab& o, long long lp) -> const char* { return o.approx[rd] + (size_t)(H + lp) * pb; };
            MTRY(copy_planes(mp, s, mid - (size_t)ab * pb, s.z0 - ab, ab, src, rd));
            MTRY(copy_planes(mp, s, mid + (size_t)s.n * pb, s.z0 + s.n, aa, src, rd));
            // WAR: this level overwrites approx[wr], which the neighbours may still be copying from (their level lev-1 halos):
            // they recorded ready[...]? no -- they READ it on THEIR streams; wait until they have issued level lev's copies below
            void* outs[16];
            outs[0] = lev == level ? (void*)s.coef : (void*)(s.approx[wr] + (size_t)H * pb);
            for (int b = 1; b < nb; ++b) outs[b] = s.coef + (size_t)((1 + (nb - 1) * (level - lev) + (b - 1)) * s.n) * pb;
            MTRY(ndwt_analysis_level_slab(s.plan, mid - (size_t)ab * pb, outs, (int)st, s.stream));
        }
        // a slab's approx[wr] is complete once its launch is; its approx[rd] may be overwritten (level lev + 1 writes it) only
        // after every slab's halo copies of THIS level are done: order both with one event per slab recorded after the launch
        for (auto& s : mp->slabs) {
            MHIP(hipSetDevice(s.device));
            MHIP(hipEventRecord(s.ready[wr], s.stream));
        }
        for (auto& s : mp->slabs)                        // every stream waits for every slab's level-lev work before level lev + 1
            for (auto& o : mp->slabs)
                if (&o != &s) MHIP(hipStreamWaitEvent(s.stream, o.ready[wr], 0));
    }
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        for (long long b = 0; b < nbt; ++b)
            MHIP(hipMemcpyAsync((char*)y_host + (size_t)(b * N + s.z0) * pb, s.coef + (size_t)(b * s.n) * pb, (size_t)s.n * pb, hipMemcpyDeviceToHost, s.stream));
    }
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        MHIP(hipStreamSynchronize(s.stream));
    }
    return NDWT_OK;
}

int ndwt_mrec_host(ndwt_mplan* mp, const void* y_host, void* x_host, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_host || !y_host) return mfail(NDWT_ERR_INVALID_ARG, "null data pointer");
    const size_t pb = mp->plane_bytes;
    const long long N = mp->dims[mp->ndim - 1], H = mp->halo_max;
    const int nb = mp->nb;
    const long long nbt = (long long)ndwt_num_bands(mp->ndim, level);
    for (auto& s : mp->slabs) {                          // coefficients to the slabs; band 0 also into approx[1] (the running approximation)
        MHIP(hipSetDevice(s.device));
        for (long long b = 0; b < nbt; ++b)
            MHIP(hipMemcpyAsync(s.coef + (size_t)(b * s.n) * pb, (const char*)y_host + (size_t)(b * N + s.z0) * pb, (size_t)s.n * pb, hipMemcpyHostToDevice, s.stream));
        MHIP(hipMemcpyAsync(s.approx[1] + (size_t)H * pb, s.coef, (size_t)s.n * pb, hipMemcpyDeviceToDevice, s.stream));
        MHIP(hipEventRecord(s.ready[1], s.stream));
    }
    for (auto& s : mp->slabs)
        for (auto& o : mp->slabs)
            if (&o != &s) MHIP(hipStreamWaitEvent(s.stream, o.ready[1], 0));
    for (int ind = 1; ind <= level; ++ind) {
        const int lev = level - ind + 1;
        const long long st = stride_of(mp, lev);
        const long long sb = (long long)(mp->L_outer / 2) * st, sa = (long long)(mp->L_outer / 2 - 1) * st;
        const int rd = (ind & 1), wr = rd ^ 1;           // ind = 1 reads approx[1]
        for (auto& s : mp->slabs) {
            MHIP(hipSetDevice(s.device));
            const long long nh = s.n + sb + sa;
            const void* ins[16];
            for (int b = 0; b < nb; ++b) {
                char* dst = s.gather + (size_t)((long long)b * nh) * pb;
                ins[b] = dst;
                const long long slot = b == 0 ? -1 : 1 + (long long)(nb - 1) * (level - lev) + (b - 1);
                auto src = [&, slot](Slab& o, long long lp) -> const char* {
                    return slot < 0 ? o.approx[rd] + (size_t)(H + lp) * pb : o.coef + (size_t)(slot * o.n + lp) * pb;
                };
                MTRY(copy_planes(mp, s, dst, s.z0 - sb, nh, src, -1));   // all producers were waited for at the end of the previous level
            }
            void* out = s.approx[wr] + (size_t)H * pb;
            MTRY(ndwt_synthesis_level_slab(s.plan, ins, out, (int)st, s.stream));
        }
        for (auto& s : mp->slabs) {
            MHIP(hipSetDevice(s.device));
            MHIP(hipEventRecord(s.ready[wr], s.stream));
        }
        for (auto& s : mp->slabs)
            for (auto& o : mp->slabs)
                if (&o != &s) MHIP(hipStreamWaitEvent(s.stream, o.ready[wr], 0));
    }
    const int fin = (level & 1) ^ 1;                     // buffer the last level wrote
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        MHIP(hipMemcpyAsync((char*)x_host + (size_t)s.z0 * pb, s.approx[fin] + (size_t)H * pb, (size_t)s.n * pb, hipMemcpyDeviceToHost, s.stream));
    }
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        MHIP(hipStreamSynchronize(s.stream));
    }
    return NDWT_OK;
}

const char* ndwt_mplan_last_error(void) { return g_merr.c_str(); }

}  // extern "C"
