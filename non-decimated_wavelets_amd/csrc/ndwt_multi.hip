// ndwt_multi.hip -- single-process, multi-device plan: ONE host thread drives the GPUs of a node.
//
// The reference's host is one MATLAB process calling one gateway (Functions/nd_dwt_3D.m:161,225 -> mex/nd_dwt_mex.c:8): a
// drop-in multi-GPU path has to live behind that single call.  This file shards the volume in slabs on its outermost axis
// over `devices[]` (a device may be listed several times: independent slabs and streams on one GPU -- how the path is tested on
// a one-GPU machine), runs the level loop with the slab entry points of include/ndwt.h and moves planes between slabs with
// asynchronous device-to-device copies (peer copies over xGMI when the slabs live on different GPUs).  Streams are ordered
// with events only; the host thread blocks once, at the end of a call.
//
// Data stays where it is: ndwt_mdec / ndwt_mrec take one device pointer per slab (the slab of x on its device, the band-planar
// coefficient slab on its device) and the kernels read and write them in place -- no plan-owned copy of a slab exists.  The host
// forms stage whole-volume host arrays through plan-owned slab buffers and call the same code.
//
// Overlap (round 4; ndwt_mplan_set_overlap, on by default): every slab has a second stream for its copies.  Analysis: the halo planes travel
// on the copy stream while the planes that need none of them are computed (one launch), then the two ends (one launch).  Scatter synthesis:
// the partial sums a slab owes its neighbours are computed FIRST (one launch for both margins), travel on the destinations' copy streams
// while every slab synthesises its own planes, and are added behind that launch.  Slabs are still joined by one event barrier per level.
//
// Exchange per level:
//   analysis   the halo planes of the approximation band only ((L/2-1) s before, (L/2) s after the slab), copied from their
//              owners into small buffers / the margins of the approximation scratch: bit-identical to one device.
//   synthesis  NDWT_EXCHANGE_SCATTER (default where the plan's levels run the fused kernels): every slab synthesises its own
//              coefficients zero-extended -- its n planes in place, plus the partial sums it owes the (L-1) s planes around it --
//              and those ONE-band planes are copied to their owners and added (a plane gets its addends in a fixed order, so
//              results are deterministic; they equal the single-device ones to rounding, the order of summation differs).
//              NDWT_EXCHANGE_GATHER: the halo planes of all 2^d bands are assembled with the slab in a scratch array (a copy of
//              every band per level) -- bit-identical to one device, and the path of plans whose levels run per-axis kernels.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "../../include/ndwt.h"

namespace {

struct Slab {
    int device;
    long long z0, n;               // planes [z0, z0 + n) of the outer axis
    ndwt_plan* plan;
    hipStream_t stream;
    hipStream_t cstream;           // this slab's copies (halo planes in, partial sums in) when the exchange overlaps with compute
    hipEvent_t copied;             // "the copies queued on cstream for this level have arrived"
    hipEvent_t ready[2];           // "approximation buffer k of this slab is complete" (k = level parity)
    hipEvent_t margins;            // "the partial sums this slab owes its neighbours are computed" (scatter synthesis)
    char* approx[2];               // [halo_max | n | halo_max] planes each: the approximation band between levels
    char* hb;                      // level-1 analysis halos of a device-resident x: halo_max planes before ...
    char* ha;                      // ... and after the slab
    char* mb;                      // scatter synthesis: partial sums for the planes before the slab (halo_max planes) ...
    char* ma;                      // ... and after it
    char* recv;                    // scatter synthesis: planes received from the neighbours before they are added (2 * halo_max planes)
    long long recv_used;           // planes of recv handed out in the current level (overlapped schedule)
    char* gather;                  // gather synthesis: 2^d bands * (n + L - 1) planes, allocated on first use
    char* xbuf;                    // host forms: the slab of x / of the result, allocated on first use
    char* coef;                    // host forms: all bands of the slab, allocated on first use
};

template <typename T>
__global__ __launch_bounds__(256) void add_planes_kernel(T* __restrict__ dst, const T* __restrict__ src, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += step) dst[i] += src[i];
}

}  // namespace

// The host side of a call with several slabs: one thread per slab queues that slab's work (ONE thread queueing for 8 devices takes longer
// than a device needs to compute its share: 1.7 ms against 1.0 ms for cfg3 -- tools/mplan_host_time.py).  Slab 0 is the caller; the others
// are persistent workers that sleep between calls and spin between the phases of one (a phase = the same function for every slab, all
// slabs done before the next phase starts: what one slab's stream waits for must have been RECORDED by the host before the wait is queued).
struct Team {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    bool stop = false;
    int call_gen = 0;                     // (under m) a call has begun: workers wake up and spin on phase_gen
    std::atomic<int> active{0};           // a call is in progress
    std::atomic<int> phase_gen{0};
    std::atomic<int> remaining{0};
    std::function<int(size_t)> job;
    std::vector<int> rc;
    std::vector<std::string> err;
    pid_t pid = 0;                        // the process that started the workers (a forked child has the plan but not the threads)
};

struct ndwt_mplan {
    int ndim, dtype, complexity, max_level, nb;
    long long dims[NDWT_MAX_DIMS];
    size_t plane_bytes;            // bytes of one plane of the outer axis (one band)
    long long halo_max;            // planes of margin in the approximation buffers
    int L_outer;
    int dilation;
    int exchange;                  // NDWT_EXCHANGE_*
    int fast;                      // the slab plans offer the split-halo analysis and the zero-extended synthesis at tap stride 1
    int overlap;                   // copies on the slabs' copy streams, overlapped with the launches that do not wait for them
    std::vector<Slab> slabs;
    std::string notes;             // peer-access findings of plan creation (ndwt_mplan_describe)
    double last_enqueue_us;        // host time the last ndwt_mdec / ndwt_mrec spent queueing work (before it waited for the devices)
    int threads;                   // 1: one host thread per slab queues its work (default with more than one slab); 0: the caller queues everything
    Team* team;
    std::vector<std::vector<int>> nbr;   // nbr[i]: the slabs (i itself included) that own a plane within halo_max of slab i: the only slabs whose
                                         // buffers slab i ever reads or whose copies ever read slab i's (the relation is symmetric)
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
#define MRET(expr)                          \
    do {                                    \
        int rc_ = (expr);                   \
        if (rc_ != NDWT_OK) return rc_;     \
    } while (0)

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

static void team_worker(ndwt_mplan* mp, size_t i) {
    Team& t = *mp->team;
    int seen_call = 0, seen_phase = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(t.m);
            t.cv.wait(lk, [&] { return t.stop || t.call_gen != seen_call; });
            if (t.stop) return;
            seen_call = t.call_gen;
        }
        while (t.active.load(std::memory_order_acquire)) {
            const int g = t.phase_gen.load(std::memory_order_acquire);
            if (g == seen_phase) { cpu_relax(); continue; }
            seen_phase = g;
            const int rc = t.job(i);
            t.rc[i] = rc;
            if (rc) t.err[i] = g_merr;
            t.remaining.fetch_sub(1, std::memory_order_release);
        }
    }
}

// a call begins: the workers wake up (created on the first call) and spin until team_end
static void team_begin(ndwt_mplan* mp) {
    const size_t G = mp->slabs.size();
    if (!mp->threads || G < 2) return;
    if (mp->team && mp->team->pid != getpid()) mp->team = nullptr;   // forked: the parent's workers do not exist here (its Team object is left alone)
    if (!mp->team) {
        mp->team = new Team();
        mp->team->pid = getpid();
        mp->team->rc.assign(G, 0);
        mp->team->err.assign(G, std::string());
        for (size_t i = 1; i < G; ++i) mp->team->th.emplace_back(team_worker, mp, i);
    }
    Team& t = *mp->team;
    t.active.store(1, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(t.m);
        ++t.call_gen;
    }
    t.cv.notify_all();
}
static void team_end(ndwt_mplan* mp) {
    if (mp->team) mp->team->active.store(0, std::memory_order_release);
}
static void team_destroy(ndwt_mplan* mp) {
    if (!mp->team) return;
    if (mp->team->pid != getpid()) { mp->team = nullptr; return; }
    {
        std::lock_guard<std::mutex> lk(mp->team->m);
        mp->team->stop = true;
    }
    mp->team->cv.notify_all();
    for (auto& th : mp->team->th) th.join();
    delete mp->team;
    mp->team = nullptr;
}

// One phase of a call: fn(i) for every slab i -- concurrently, one host thread per slab, when the team is active; in slab order on the
// caller's thread otherwise.  Returns when every slab's fn has returned, with the first error (its message in this thread's g_merr).
// fn(i) touches slab i's streams only, and starts with hipSetDevice (the current device is per thread).
template <class F> static int phase(ndwt_mplan* mp, F&& fn) {
    const size_t G = mp->slabs.size();
    Team* t = mp->team;
    if (!t || !t->active.load(std::memory_order_relaxed)) {
        for (size_t i = 0; i < G; ++i) MRET(fn(i));
        return NDWT_OK;
    }
    t->job = [&fn](size_t i) -> int { return fn(i); };
    t->remaining.store((int)G - 1, std::memory_order_relaxed);
    t->phase_gen.fetch_add(1, std::memory_order_release);
    const int rc0 = fn(0);
    while (t->remaining.load(std::memory_order_acquire) != 0) cpu_relax();
    if (rc0) return rc0;
    for (size_t i = 1; i < G; ++i)
        if (t->rc[i]) { g_merr = t->err[i]; return t->rc[i]; }
    return NDWT_OK;
}

static long long stride_of(const ndwt_mplan* mp, int lev) { return mp->dilation == NDWT_DILATION_ATROUS ? (1LL << (lev - 1)) : 1LL; }

static Slab* owner_of(ndwt_mplan* mp, long long gp) {
    for (auto& s : mp->slabs)
        if (gp >= s.z0 && gp < s.z0 + s.n) return &s;
    return nullptr;
}

static int copy_run(ndwt_mplan* mp, Slab& to, char* dst, const Slab& from, const char* src, long long planes, hipStream_t st = nullptr) {
    const size_t bytes = (size_t)planes * mp->plane_bytes;
    if (!st) st = to.stream;
    if (from.device == to.device) MHIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    else MHIP(hipMemcpyPeerAsync(dst, to.device, src, from.device, bytes, st));
    return NDWT_OK;
}

// copy `count` planes starting at GLOBAL plane g (periodic) of a per-slab plane array into dst on slab `to`, on `to`'s stream.
// plane_ptr(slab, local plane) gives the source address; wait_parity >= 0: wait for the source slab's ready[wait_parity] first.
template <class SrcFn>
static int copy_planes(ndwt_mplan* mp, Slab& to, char* dst, long long g, long long count, SrcFn plane_ptr, int wait_parity, hipStream_t st = nullptr) {
    const long long N = mp->dims[mp->ndim - 1];
    long long done = 0;
    while (done < count) {
        const long long gp = ((g + done) % N + N) % N;
        Slab* src = owner_of(mp, gp);
        if (!src) return mfail(NDWT_ERR_INVALID_ARG, "plane %lld has no owner", gp);
        long long run = src->z0 + src->n - gp;
        if (run > count - done) run = count - done;
        if (wait_parity >= 0 && src != &to) MHIP(hipStreamWaitEvent(st ? st : to.stream, src->ready[wait_parity], 0));
        MRET(copy_run(mp, to, dst + (size_t)done * mp->plane_bytes, *src, plane_ptr(*src, gp - src->z0), run, st));
        done += run;
    }
    return NDWT_OK;
}

static int add_planes(ndwt_mplan* mp, Slab& s, char* dst, const char* src, long long planes) {
    const size_t es = mp->dtype == NDWT_F32 ? 4 : 8;
    const long long n = (long long)((size_t)planes * mp->plane_bytes / es);
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) return NDWT_OK;
    if (mp->dtype == NDWT_F32) hipLaunchKernelGGL(add_planes_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s.stream, (float*)dst, (const float*)src, n);
    else hipLaunchKernelGGL(add_planes_kernel<double>, dim3((unsigned)blocks), dim3(256), 0, s.stream, (double*)dst, (const double*)src, n);
    MHIP(hipGetLastError());
    return NDWT_OK;
}

// The level barrier between slabs (events only, the host does not block): every stream waits for ready[k] of the slabs within halo reach of
// its own -- the slabs whose approximation planes its next level copies (read after write) and whose copies read the buffer its level after
// next overwrites (write after read).  Slabs further away are ordered through those (a slab records ready[k] behind its waits of the level
// before).  With all G slabs waiting for all G (2 G^2 calls per level) this function was two thirds of the calls of a transform on 8 devices.
// Two phases: every slab's ready[k] is recorded (by the host) before any stream is told to wait for it.
static int barrier_record(ndwt_mplan* mp, size_t i, int k) {
    Slab& s = mp->slabs[i];
    MHIP(hipSetDevice(s.device));
    MHIP(hipEventRecord(s.ready[k], s.stream));
    return NDWT_OK;
}
static int barrier_wait(ndwt_mplan* mp, size_t i, int k) {
    Slab& s = mp->slabs[i];
    MHIP(hipSetDevice(s.device));
    for (int j : mp->nbr[i]) {
        Slab& o = mp->slabs[(size_t)j];
        if (&o != &s) MHIP(hipStreamWaitEvent(s.stream, o.ready[k], 0));
        MHIP(hipStreamWaitEvent(s.cstream, o.ready[k], 0));   // (its own slab's too: the copy stream is ordered by events only)
    }
    return NDWT_OK;
}
static int level_barrier(ndwt_mplan* mp, int k) {
    MRET(phase(mp, [&](size_t i) { return barrier_record(mp, i, k); }));
    return phase(mp, [&](size_t i) { return barrier_wait(mp, i, k); });
}

static int sync_all(ndwt_mplan* mp) {
    for (auto& s : mp->slabs) {
        MHIP(hipSetDevice(s.device));
        MHIP(hipStreamSynchronize(s.stream));
        MHIP(hipStreamSynchronize(s.cstream));
    }
    return NDWT_OK;
}

// the overlapped schedules cut a slab into the planes that need no neighbour and the rest: every slab must be thick enough for that
static bool can_overlap(const ndwt_mplan* mp, long long before, long long after) {
    if (!mp->overlap || !mp->fast) return false;
    const long long m = before > after ? before : after;
    for (auto& s : mp->slabs)
        if (s.n < 2 * m + 1 || s.n <= before + after) return false;
    return true;
}

static int lazy_alloc(Slab& s, char** p, size_t bytes) {
    if (*p) return NDWT_OK;
    hipError_t e = hipSetDevice(s.device);
    if (e == hipSuccess) e = hipMalloc((void**)p, bytes);
    if (e != hipSuccess) return mfail(NDWT_ERR_ALLOC, "device %d: hipMalloc(%zu bytes) failed: %s", s.device, bytes, hipGetErrorString(e));
    return NDWT_OK;
}

static int mcheck(const ndwt_mplan* mp, int level) {
    if (!mp) return mfail(NDWT_ERR_INVALID_ARG, "null plan");
    if (level < 1 || level > mp->max_level) return mfail(NDWT_ERR_INVALID_ARG, "level %d outside 1..max_level=%d of this plan", level, mp->max_level);
    return NDWT_OK;
}

// ------------------------------------------------------------------------------------------------ analysis
// x[i]: the slab of the signal on slab i's device (n_i planes); y[i]: its coefficient slab, band b at b * n_i planes.  Queues the
// whole transform on the slabs' streams; the caller synchronises.
static int mdec_core(ndwt_mplan* mp, const void* const* x, void* const* y, int level) {
    const size_t pb = mp->plane_bytes;
    const long long H = mp->halo_max;
    const int nb = mp->nb;
    for (int lev = 1; lev <= level; ++lev) {
        const long long st = stride_of(mp, lev);
        const long long ab = (long long)(mp->L_outer / 2 - 1) * st, aa = (long long)(mp->L_outer / 2) * st;
        // level lev reads approx[rd] (level 1: the caller's x) and writes its approximation into approx[wr] (the last level: band 0 of y)
        const int rd = ((lev - 1) & 1) ^ 1, wr = (lev - 1) & 1;
        // one slab's work of this level (its copies out of the neighbours' buffers, its launches) and the record of its ready[wr]
        auto body = [&](size_t i) -> int {
            Slab& s = mp->slabs[i];
            MHIP(hipSetDevice(s.device));
            void* outs[16];
            outs[0] = lev == level ? y[i] : (void*)(s.approx[wr] + (size_t)H * pb);
            for (int b = 1; b < nb; ++b) outs[b] = (char*)y[i] + (size_t)((1 + (nb - 1) * (level - lev) + (b - 1)) * s.n) * pb;
            const bool ov = st == 1 && can_overlap(mp, ab, aa);
            const long long m = ab > aa ? ab : aa;
            if (ov) {
                // halo planes on the copy stream; the planes [ab, n - aa) that need none of them meanwhile; then the two ends
                void* oi[16];
                for (int b = 0; b < nb; ++b) oi[b] = (char*)outs[b] + (size_t)ab * pb;
                if (lev == 1) {
                    auto src = [&](Slab& o, long long lp) -> const char* { return (const char*)x[&o - &mp->slabs[0]] + (size_t)lp * pb; };
                    const char* xs = (const char*)x[i];
                    MRET(copy_planes(mp, s, s.hb, s.z0 - ab, ab, src, -1, s.cstream));
                    MRET(copy_planes(mp, s, s.ha, s.z0 + s.n, aa, src, -1, s.cstream));
                    MHIP(hipEventRecord(s.copied, s.cstream));
                    MTRY(ndwt_analysis_level_slab_part(s.plan, xs + (size_t)ab * pb, ab ? xs : nullptr, xs + (size_t)(s.n - aa) * pb, oi, 1, s.n - ab - aa, s.stream));
                    MHIP(hipStreamWaitEvent(s.stream, s.copied, 0));
                    // the slab of x is the caller's (no margins around it): the ends are two launches, halo planes from hb / ha
                    MTRY(ndwt_analysis_level_slab_part(s.plan, xs, ab ? s.hb : nullptr, xs + (size_t)m * pb, outs, 1, m, s.stream));
                    void* oe[16];
                    for (int b = 0; b < nb; ++b) oe[b] = (char*)outs[b] + (size_t)(s.n - m) * pb;
                    MTRY(ndwt_analysis_level_slab_part(s.plan, xs + (size_t)(s.n - m) * pb, ab ? xs + (size_t)(s.n - m - ab) * pb : nullptr, s.ha, oe, 1, m, s.stream));
                } else {
                    char* mid = s.approx[rd] + (size_t)H * pb;
                    auto src = [&](Slab& o, long long lp) -> const char* { return o.approx[rd] + (size_t)(H + lp) * pb; };
                    MRET(copy_planes(mp, s, mid - (size_t)ab * pb, s.z0 - ab, ab, src, -1, s.cstream));
                    MRET(copy_planes(mp, s, mid + (size_t)s.n * pb, s.z0 + s.n, aa, src, -1, s.cstream));
                    MHIP(hipEventRecord(s.copied, s.cstream));
                    MTRY(ndwt_analysis_level_slab_part(s.plan, mid + (size_t)ab * pb, ab ? mid : nullptr, mid + (size_t)(s.n - aa) * pb, oi, 1, s.n - ab - aa, s.stream));
                    MHIP(hipStreamWaitEvent(s.stream, s.copied, 0));
                    MTRY(ndwt_analysis_level_slab_runs(s.plan, mid - (size_t)ab * pb, outs, 1, m, 2, s.n - m, s.stream));   // both ends, one launch
                }
            } else if (lev == 1) {
                // the slab of x is read where it lies; its halo planes come from the neighbours' slabs of x
                auto src = [&](Slab& o, long long lp) -> const char* { return (const char*)x[&o - &mp->slabs[0]] + (size_t)lp * pb; };
                if (mp->fast && st == 1) {
                    MRET(copy_planes(mp, s, s.hb, s.z0 - ab, ab, src, -1));
                    MRET(copy_planes(mp, s, s.ha, s.z0 + s.n, aa, src, -1));
                    MTRY(ndwt_analysis_level_slab_split(s.plan, x[i], ab ? s.hb : nullptr, s.ha, outs, 1, s.stream));
                } else {                                  // kernels that want the halo planes in line with the slab: one copy of the slab
                    char* mid = s.approx[rd] + (size_t)H * pb;
                    MRET(copy_planes(mp, s, mid - (size_t)ab * pb, s.z0 - ab, ab + s.n + aa, src, -1));
                    MTRY(ndwt_analysis_level_slab(s.plan, mid - (size_t)ab * pb, outs, (int)st, s.stream));
                }
            } else {
                char* mid = s.approx[rd] + (size_t)H * pb;   // produced in place by level lev - 1; the margins take the neighbours' planes
                auto src = [&](Slab& o, long long lp) -> const char* { return o.approx[rd] + (size_t)(H + lp) * pb; };
                MRET(copy_planes(mp, s, mid - (size_t)ab * pb, s.z0 - ab, ab, src, -1));   // (producers: waited for at the end of level lev - 1)
                MRET(copy_planes(mp, s, mid + (size_t)s.n * pb, s.z0 + s.n, aa, src, -1));
                MTRY(ndwt_analysis_level_slab(s.plan, mid - (size_t)ab * pb, outs, (int)st, s.stream));
            }
            return barrier_record(mp, i, wr);
        };
        MRET(phase(mp, body));
        // approx[wr] of every slab is complete, and approx[rd] free to be overwritten by level lev + 1, once every slab's work of this
        // level (its launch and its halo copies out of the neighbours' buffers) is done: the level barrier's waits
        MRET(phase(mp, [&](size_t i) { return barrier_wait(mp, i, wr); }));
    }
    return NDWT_OK;
}

// ----------------------------------------------------------------------------------------------- synthesis
// partial sums of `count` planes starting at global plane g (periodic), held in `buf` on slab `from`: to their owners, added there.
// dst_of(slab) = where that slab's n planes of this level's result live.
template <class DstFn>
static int scatter_margin(ndwt_mplan* mp, Slab& from, const char* buf, long long g, long long count, DstFn dst_of) {
    const long long N = mp->dims[mp->ndim - 1];
    long long done = 0;
    while (done < count) {
        const long long gp = ((g + done) % N + N) % N;
        Slab* to = owner_of(mp, gp);
        if (!to) return mfail(NDWT_ERR_INVALID_ARG, "plane %lld has no owner", gp);
        long long run = to->z0 + to->n - gp;
        if (run > count - done) run = count - done;
        MHIP(hipSetDevice(to->device));
        if (to != &from) MHIP(hipStreamWaitEvent(to->stream, from.margins, 0));
        const char* src = buf + (size_t)done * mp->plane_bytes;
        char* dst = dst_of(*to) + (size_t)(gp - to->z0) * mp->plane_bytes;
        if (to->device == from.device) {                  // same memory: add straight from the producer's buffer
            MRET(add_planes(mp, *to, dst, src, run));
        } else {
            MRET(copy_run(mp, *to, to->recv, from, src, run));
            MRET(add_planes(mp, *to, dst, to->recv, run));
        }
        done += run;
    }
    return NDWT_OK;
}

static int mrec_core(ndwt_mplan* mp, const void* const* y, void* const* x, int level) {
    const size_t pb = mp->plane_bytes;
    const long long H = mp->halo_max;
    const int nb = mp->nb;
    for (int ind = 1; ind <= level; ++ind) {
        const int lev = level - ind + 1;
        const long long st = stride_of(mp, lev);
        const long long sb = (long long)(mp->L_outer / 2) * st, sa = (long long)(mp->L_outer / 2 - 1) * st;
        const int rd = ind & 1, wr = rd ^ 1;             // level index ind > 1 reads the approximation level ind - 1 wrote into approx[rd]
        const bool scatter = mp->exchange == NDWT_EXCHANGE_SCATTER && mp->fast && st == 1;
        auto band_ptr = [&](size_t i, Slab& o, int b) -> const char* {
            if (b == 0) return ind == 1 ? (const char*)y[i] : o.approx[rd] + (size_t)H * pb;
            return (const char*)y[i] + (size_t)((1 + (long long)(nb - 1) * (level - lev) + (b - 1)) * o.n) * pb;
        };
        auto dst_of = [&](Slab& o) -> char* { return lev == 1 ? (char*)x[&o - &mp->slabs[0]] : o.approx[wr] + (size_t)H * pb; };
        if (scatter && can_overlap(mp, sa, sb)) {
            // margins first (one launch: two runs of m planes at the ends of the zero-extended result), so that they travel -- on the
            // destinations' copy streams -- while every slab synthesises its own planes; the adds follow that launch in stream order
            const long long m = sa > sb ? sa : sb;
            MRET(phase(mp, [&](size_t i) -> int {
                Slab& s = mp->slabs[i];
                MHIP(hipSetDevice(s.device));
                const void* ins[16];
                for (int b = 0; b < nb; ++b) ins[b] = band_ptr(i, s, b);
                MTRY(ndwt_synthesis_level_slab_runs(s.plan, ins, s.n, 0, s.n + sa + sb - m, 2, m, s.mb, 1, s.stream));   // mb: [run 0 | run 1]
                MHIP(hipEventRecord(s.margins, s.stream));
                MTRY(ndwt_synthesis_level_slab_part(s.plan, ins, s.n, sa, s.n, dst_of(s), 1, s.stream));
                return NDWT_OK;
            }));
            // the partial sums: run 0 holds planes [0, m) of the zero-extended result (the first sa: owed to the planes before the slab),
            // run 1 planes [n + sa + sb - m, n + sa + sb) (the last sb: the planes after it).  Copies to other devices go to the
            // destination's copy stream and a region of its receive buffer of their own; the adds come in slab order of the producers,
            // "before" margins first, behind the destination's own launch: the same fixed order of summation as without overlap.
            // Every DESTINATION slab queues what arrives at it (its own streams only): the producers' margins events were recorded in the
            // phase above.  The record of its ready[wr] (level barrier) closes the phase.
            const long long N = mp->dims[mp->ndim - 1];
            MRET(phase(mp, [&](size_t ti) -> int {
                Slab& t = mp->slabs[ti];
                MHIP(hipSetDevice(t.device));
                struct Pending { char* dst; const char* src; long long run; };
                std::vector<Pending> adds;
                t.recv_used = 0;
                for (auto& from : mp->slabs) {
                    for (int side = 0; side < 2; ++side) {
                        const long long count = side == 0 ? sa : sb;
                        const long long g = side == 0 ? from.z0 - sa : from.z0 + from.n;
                        const char* buf = side == 0 ? from.mb : from.mb + (size_t)(m + (m - sb)) * pb;
                        long long done = 0;
                        while (done < count) {
                            const long long gp = ((g + done) % N + N) % N;
                            Slab* to = owner_of(mp, gp);
                            if (!to) return mfail(NDWT_ERR_INVALID_ARG, "plane %lld has no owner", gp);
                            long long run = to->z0 + to->n - gp;
                            if (run > count - done) run = count - done;
                            if (to == &t) {
                                const char* src = buf + (size_t)done * pb;
                                char* dst = dst_of(t) + (size_t)(gp - t.z0) * pb;
                                if (t.device != from.device || mp->overlap == 2) {   // (2: test hook -- the staged path between slabs of one device)
                                    if (t.recv_used + run > 2 * H) return mfail(NDWT_ERR_UNSUPPORTED, "internal: receive buffer of the overlapped synthesis exhausted");
                                    char* rb = t.recv + (size_t)t.recv_used * pb;
                                    t.recv_used += run;
                                    MHIP(hipStreamWaitEvent(t.cstream, from.margins, 0));
                                    MRET(copy_run(mp, t, rb, from, src, run, t.cstream));
                                    src = rb;
                                } else if (&t != &from) {
                                    MHIP(hipStreamWaitEvent(t.stream, from.margins, 0));   // same memory: added straight from the producer's buffer
                                }
                                adds.push_back({dst, src, run});
                            }
                            done += run;
                        }
                    }
                }
                MHIP(hipEventRecord(t.copied, t.cstream));
                MHIP(hipStreamWaitEvent(t.stream, t.copied, 0));
                for (auto& a : adds) MRET(add_planes(mp, t, a.dst, a.src, a.run));
                return barrier_record(mp, ti, wr);
            }));
            MRET(phase(mp, [&](size_t i) { return barrier_wait(mp, i, wr); }));
            continue;
        } else if (scatter) {
            // zero-extended synthesis: plane k of the extended result = global plane z0 - sa + k.  The slab's own n planes go where
            // the result lives; the sa planes before and the sb planes after it are partial sums owed to their owners.
            size_t i = 0;
            for (auto& s : mp->slabs) {
                MHIP(hipSetDevice(s.device));
                const void* ins[16];
                for (int b = 0; b < nb; ++b) ins[b] = band_ptr(i, s, b);
                MTRY(ndwt_synthesis_level_slab_part(s.plan, ins, s.n, sa, s.n, dst_of(s), 1, s.stream));
                if (sa) MTRY(ndwt_synthesis_level_slab_part(s.plan, ins, s.n, 0, sa, s.mb, 1, s.stream));
                MTRY(ndwt_synthesis_level_slab_part(s.plan, ins, s.n, sa + s.n, sb, s.ma, 1, s.stream));
                MHIP(hipEventRecord(s.margins, s.stream));
                ++i;
            }
            // a destination's own planes are written by its own stream before anything is added to them (stream order); the addends
            // arrive in slab order, "before" margins first: a fixed order of summation
            for (auto& s : mp->slabs) {
                if (sa) MRET(scatter_margin(mp, s, s.mb, s.z0 - sa, sa, dst_of));
                MRET(scatter_margin(mp, s, s.ma, s.z0 + s.n, sb, dst_of));
            }
        } else {
            for (auto& s : mp->slabs) {
                MHIP(hipSetDevice(s.device));
                const long long nh = s.n + sb + sa;
                const long long smax = mp->dilation == NDWT_DILATION_ATROUS ? (1LL << (mp->max_level - 1)) : 1LL;
                MRET(lazy_alloc(s, &s.gather, (size_t)((long long)nb * (s.n + (long long)(mp->L_outer - 1) * smax)) * pb));
                const void* ins[16];
                for (int b = 0; b < nb; ++b) {
                    char* dst = s.gather + (size_t)((long long)b * nh) * pb;
                    ins[b] = dst;
                    auto src = [&, b](Slab& o, long long lp) -> const char* { return band_ptr(&o - &mp->slabs[0], o, b) + (size_t)lp * pb; };
                    MRET(copy_planes(mp, s, dst, s.z0 - sb, nh, src, -1));   // all producers were waited for at the end of the previous level
                }
                MTRY(ndwt_synthesis_level_slab(s.plan, ins, dst_of(s), (int)st, s.stream));
            }
        }
        MRET(level_barrier(mp, wr));
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
    mp->exchange = NDWT_EXCHANGE_SCATTER;
    mp->overlap = 1;
    mp->threads = 1;
    mp->team = nullptr;
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
    std::vector<int64_t> ld(dims, dims + ndim);
    for (int i = 0; i < ndev; ++i) {
        Slab s;
        memset(&s, 0, sizeof s);
        s.device = devices[i];
        s.z0 = (long long)i * N / ndev;
        s.n = (long long)(i + 1) * N / ndev - s.z0;
        mp->slabs.push_back(s);
    }
    mp->nbr.resize(mp->slabs.size());
    for (size_t i = 0; i < mp->slabs.size(); ++i) {
        const Slab& si = mp->slabs[i];
        for (long long d = -mp->halo_max; d < si.n + mp->halo_max; ++d) {
            const Slab* o = owner_of(mp, ((si.z0 + d) % N + N) % N);
            const int j = o ? (int)(o - &mp->slabs[0]) : (int)i;
            bool have = false;
            for (int q : mp->nbr[i]) have = have || q == j;
            if (!have) mp->nbr[i].push_back(j);
        }
    }
    int rc = NDWT_OK;
    mp->fast = 1;
    for (auto& s : mp->slabs) {
        ld[ndim - 1] = s.n;
        if ((rc = ndwt_plan_create_slab(&s.plan, ndim, ld.data(), N, wnames, dtype, complexity, pres_l2_norm, dilation, 1, s.device)) != NDWT_OK) break;
        if (!ndwt_plan_slab_fast(s.plan)) mp->fast = 0;
        hipError_t e = hipSetDevice(s.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.cstream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&s.ready[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.margins, hipEventDisableTiming);
        for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipMalloc((void**)&s.approx[k], (size_t)(s.n + 2 * mp->halo_max) * pb);
        for (char** p : {&s.hb, &s.ha, &s.ma})
            if (e == hipSuccess) e = hipMalloc((void**)p, (size_t)mp->halo_max * pb);
        for (char** p : {&s.mb, &s.recv})               // mb: both margins of the overlapped synthesis (two runs); recv: a region per sender
            if (e == hipSuccess) e = hipMalloc((void**)p, (size_t)(2 * mp->halo_max) * pb);
        if (e != hipSuccess) { rc = mfail(NDWT_ERR_ALLOC, "device %d: %s", s.device, hipGetErrorString(e)); break; }
        for (auto& o : mp->slabs)                        // peer access where the runtime offers it (same-device pairs need none)
            if (o.device != s.device) {
                int can = 0;
                hipError_t pe = hipDeviceCanAccessPeer(&can, s.device, o.device);
                if (pe == hipSuccess && can) {
                    pe = hipDeviceEnablePeerAccess(o.device, 0);
                    if (pe == hipErrorPeerAccessAlreadyEnabled) pe = hipSuccess;
                }
                (void)hipGetLastError();
                if (pe != hipSuccess || !can) {           // the copies still work (staged by the runtime), slower: say so
                    char note[160];
                    snprintf(note, sizeof note, "no peer access %d -> %d (%s); ", s.device, o.device, pe != hipSuccess ? hipGetErrorString(pe) : "not offered");
                    if (mp->notes.find(note) == std::string::npos) mp->notes += note;
                }
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
    team_destroy(mp);
    for (auto& s : mp->slabs) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.plan) ndwt_plan_destroy(s.plan);
        for (int k = 0; k < 2; ++k) {
            if (s.approx[k]) (void)hipFree(s.approx[k]);
            if (s.ready[k]) (void)hipEventDestroy(s.ready[k]);
        }
        if (s.margins) (void)hipEventDestroy(s.margins);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.cstream) { (void)hipStreamSynchronize(s.cstream); (void)hipStreamDestroy(s.cstream); }
        for (char* p : {s.hb, s.ha, s.mb, s.ma, s.recv, s.gather, s.xbuf, s.coef})
            if (p) (void)hipFree(p);
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

int ndwt_mplan_set_exchange(ndwt_mplan* mp, int exchange) {
    if (!mp || (exchange != NDWT_EXCHANGE_SCATTER && exchange != NDWT_EXCHANGE_GATHER)) return mfail(NDWT_ERR_INVALID_ARG, "bad plan / exchange scheme");
    mp->exchange = exchange;
    return NDWT_OK;
}

int ndwt_mplan_set_overlap(ndwt_mplan* mp, int overlap) {
    if (!mp) return mfail(NDWT_ERR_INVALID_ARG, "null plan");
    mp->overlap = overlap == 2 ? 2 : (overlap ? 1 : 0);
    return NDWT_OK;
}

int ndwt_mplan_set_threads(ndwt_mplan* mp, int threads) {
    if (!mp) return mfail(NDWT_ERR_INVALID_ARG, "null plan");
    mp->threads = threads ? 1 : 0;
    if (!mp->threads) team_destroy(mp);
    return NDWT_OK;
}

double ndwt_mplan_last_enqueue_us(const ndwt_mplan* mp) { return mp ? mp->last_enqueue_us : -1.0; }

int ndwt_mplan_describe(const ndwt_mplan* mp, char* buf, int buflen) {
    if (!mp || !buf || buflen < 1) return mfail(NDWT_ERR_INVALID_ARG, "bad arguments");
    const bool scatter = mp->exchange == NDWT_EXCHANGE_SCATTER && mp->fast && mp->dilation == NDWT_DILATION_REFERENCE;
    snprintf(buf, (size_t)buflen, "%d slabs; analysis: halo planes of the approximation band%s; synthesis: %s; %s; %s", (int)mp->slabs.size(),
             mp->fast ? ", slabs read in place" : "", scatter ? "scatter-add of one band of partial sums" : "gather of the halo planes of all bands",
             (mp->overlap && mp->fast) ? "exchange overlapped with the planes that do not wait for it (where every slab is thick enough)" : "exchange, then compute",
             mp->notes.empty() ? "peer access between all devices" : mp->notes.c_str());
    return NDWT_OK;
}

// device-resident form: x_slabs[i] / y_slabs[i] live on slab i's device (ndwt_mplan_slab: n_i planes of x; band b of the coefficient
// slab at b * n_i planes).  The data must be complete when the call is made (no stream of the caller is waited for); the call returns
// when the result is.
int ndwt_mdec(ndwt_mplan* mp, const void* const* x_slabs, void* const* y_slabs, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_slabs || !y_slabs) return mfail(NDWT_ERR_INVALID_ARG, "null pointer array");
    for (size_t i = 0; i < mp->slabs.size(); ++i)
        if (!x_slabs[i] || !y_slabs[i]) return mfail(NDWT_ERR_INVALID_ARG, "null slab pointer %zu", i);
    const auto t0 = std::chrono::steady_clock::now();
    team_begin(mp);
    rc = mdec_core(mp, x_slabs, y_slabs, level);
    team_end(mp);
    mp->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const int rs = sync_all(mp);                          // (also after an error: nothing stays queued on buffers the caller owns)
    return rc ? rc : rs;
}

int ndwt_mrec(ndwt_mplan* mp, const void* const* y_slabs, void* const* x_slabs, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_slabs || !y_slabs) return mfail(NDWT_ERR_INVALID_ARG, "null pointer array");
    for (size_t i = 0; i < mp->slabs.size(); ++i)
        if (!x_slabs[i] || !y_slabs[i]) return mfail(NDWT_ERR_INVALID_ARG, "null slab pointer %zu", i);
    const auto t0 = std::chrono::steady_clock::now();
    team_begin(mp);
    rc = mrec_core(mp, y_slabs, x_slabs, level);
    team_end(mp);
    mp->last_enqueue_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    const int rs = sync_all(mp);
    return rc ? rc : rs;
}

// whole-volume host arrays in, whole-volume host arrays out (the layout the MATLAB gateway holds): x is prod(dims) elements,
// y prod(dims) * ndwt_num_bands(ndim, level), band-planar.  Blocks until the result is in y.
static int host_buffers(ndwt_mplan* mp, std::vector<void*>& xs, std::vector<void*>& ys) {
    const long long nbt = (long long)ndwt_num_bands(mp->ndim, mp->max_level);
    for (auto& s : mp->slabs) {
        MRET(lazy_alloc(s, &s.xbuf, (size_t)s.n * mp->plane_bytes));
        MRET(lazy_alloc(s, &s.coef, (size_t)(nbt * s.n) * mp->plane_bytes));
        xs.push_back(s.xbuf);
        ys.push_back(s.coef);
    }
    return NDWT_OK;
}

int ndwt_mdec_host(ndwt_mplan* mp, const void* x_host, void* y_host, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_host || !y_host) return mfail(NDWT_ERR_INVALID_ARG, "null data pointer");
    const size_t pb = mp->plane_bytes;
    const long long N = mp->dims[mp->ndim - 1];
    const long long nbt = (long long)ndwt_num_bands(mp->ndim, level);
    std::vector<void*> xs, ys;
    MRET(host_buffers(mp, xs, ys));
    // the copies between the host arrays and the slabs are queued by the slabs' own threads too: from pageable memory (what MATLAB holds) a
    // copy occupies its calling thread until it is staged, and one thread would serve the devices' links one after the other
    team_begin(mp);
    rc = phase(mp, [&](size_t i) -> int {
        Slab& s = mp->slabs[i];
        MHIP(hipSetDevice(s.device));
        MHIP(hipMemcpyAsync(s.xbuf, (const char*)x_host + (size_t)s.z0 * pb, (size_t)s.n * pb, hipMemcpyHostToDevice, s.stream));
        return NDWT_OK;
    });
    if (rc == NDWT_OK) rc = level_barrier(mp, 1);         // every slab of x is in place before a neighbour reads its halo planes
    if (rc == NDWT_OK) rc = mdec_core(mp, xs.data(), ys.data(), level);
    if (rc == NDWT_OK)
        rc = phase(mp, [&](size_t i) -> int {
            Slab& s = mp->slabs[i];
            MHIP(hipSetDevice(s.device));
            for (long long b = 0; b < nbt; ++b)
                MHIP(hipMemcpyAsync((char*)y_host + (size_t)(b * N + s.z0) * pb, s.coef + (size_t)(b * s.n) * pb, (size_t)s.n * pb, hipMemcpyDeviceToHost, s.stream));
            return NDWT_OK;
        });
    team_end(mp);
    const int rs = sync_all(mp);
    return rc ? rc : rs;
}

int ndwt_mrec_host(ndwt_mplan* mp, const void* y_host, void* x_host, int level) {
    int rc = mcheck(mp, level);
    if (rc) return rc;
    if (!x_host || !y_host) return mfail(NDWT_ERR_INVALID_ARG, "null data pointer");
    const size_t pb = mp->plane_bytes;
    const long long N = mp->dims[mp->ndim - 1];
    const long long nbt = (long long)ndwt_num_bands(mp->ndim, level);
    std::vector<void*> xs, ys;
    MRET(host_buffers(mp, xs, ys));
    team_begin(mp);
    rc = phase(mp, [&](size_t i) -> int {
        Slab& s = mp->slabs[i];
        MHIP(hipSetDevice(s.device));
        for (long long b = 0; b < nbt; ++b)
            MHIP(hipMemcpyAsync(s.coef + (size_t)(b * s.n) * pb, (const char*)y_host + (size_t)(b * N + s.z0) * pb, (size_t)s.n * pb, hipMemcpyHostToDevice, s.stream));
        return NDWT_OK;
    });
    if (rc == NDWT_OK) rc = level_barrier(mp, 1);
    if (rc == NDWT_OK) rc = mrec_core(mp, ys.data(), xs.data(), level);
    if (rc == NDWT_OK)
        rc = phase(mp, [&](size_t i) -> int {
            Slab& s = mp->slabs[i];
            MHIP(hipSetDevice(s.device));
            MHIP(hipMemcpyAsync((char*)x_host + (size_t)s.z0 * pb, s.xbuf, (size_t)s.n * pb, hipMemcpyDeviceToHost, s.stream));
            return NDWT_OK;
        });
    team_end(mp);
    const int rs = sync_all(mp);
    return rc ? rc : rs;
}

const char* ndwt_mplan_last_error(void) { return g_merr.c_str(); }

}  // extern "C"
