// ndwt_comm.hip -- the halo / partial-sum exchange of the one-process-per-GPU driver as RCCL point-to-point calls ON THE CALLER'S STREAM.
//
// No reference counterpart (the reference is single-process, SURVEY.md section 5, which prescribes exactly this: ncclSend / ncclRecv inside
// ncclGroupStart / ncclGroupEnd between ring neighbours).  Why not torch.distributed's batch_isend_irecv: its NCCL work runs on a stream of its
// own, so every exchange pays two cross-stream dependencies -- measured on one MI355X with a 1-rank group (profiles/r04_timeline_rccl.txt): 40 us
// from the last compute kernel to the RCCL kernel, 15 us of RCCL kernel, 14 us back to the next compute kernel, six times per dec + rec step of
// 0.83 ms.  Enqueued on the stream that runs the transform, the same grouped send / receive is one more kernel in stream order.
//
// RCCL is opened at run time (dlopen): libndwt_hip.so has no link-time dependency on it (the MATLAB gateway and single-GPU users never need
// it), and a process that has imported torch gets the librccl torch has already loaded (same soname), not a second copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/ndwt.h"

namespace {

typedef struct { char internal[128]; } RcclUniqueId;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128, rccl.h:40)
typedef void* RcclComm;
typedef int (*fn_GetUniqueId)(RcclUniqueId*);
typedef int (*fn_CommInitRank)(RcclComm*, int, RcclUniqueId, int);
typedef int (*fn_CommDestroy)(RcclComm);
typedef int (*fn_Group)(void);
typedef int (*fn_Send)(const void*, size_t, int, int, RcclComm, hipStream_t);
typedef int (*fn_Recv)(void*, size_t, int, int, RcclComm, hipStream_t);
typedef const char* (*fn_ErrorString)(int);

struct Rccl {
    void* handle = nullptr;
    fn_GetUniqueId GetUniqueId = nullptr;
    fn_CommInitRank CommInitRank = nullptr;
    fn_CommDestroy CommDestroy = nullptr;
    fn_Group GroupStart = nullptr, GroupEnd = nullptr;
    fn_Send Send = nullptr;
    fn_Recv Recv = nullptr;
    fn_ErrorString ErrorString = nullptr;
};
Rccl g_rccl;
thread_local std::string g_cerr;

int cfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_cerr = buf;
    return code;
}

int load_rccl() {
    if (g_rccl.handle) return NDWT_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return cfail(NDWT_ERR_UNSUPPORTED, "librccl could not be opened: %s", dlerror());
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (fn_GetUniqueId)dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (fn_CommInitRank)dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (fn_CommDestroy)dlsym(h, "ncclCommDestroy");
    r.GroupStart = (fn_Group)dlsym(h, "ncclGroupStart");
    r.GroupEnd = (fn_Group)dlsym(h, "ncclGroupEnd");
    r.Send = (fn_Send)dlsym(h, "ncclSend");
    r.Recv = (fn_Recv)dlsym(h, "ncclRecv");
    r.ErrorString = (fn_ErrorString)dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.ErrorString)
        return cfail(NDWT_ERR_UNSUPPORTED, "librccl lacks a point-to-point entry point");
    g_rccl = r;
    return NDWT_OK;
}

}  // namespace

struct ndwt_comm {
    RcclComm comm;
    int nranks, rank, device;
};

#define RCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        const int r_ = (expr);                                                                                  \
        if (r_ != 0) return cfail(NDWT_ERR_HIP, "%s failed: %s", #expr, g_rccl.ErrorString(r_));                \
    } while (0)

extern "C" {

int ndwt_comm_unique_id(void* id128) {
    if (!id128) return cfail(NDWT_ERR_INVALID_ARG, "null pointer");
    int rc = load_rccl();
    if (rc) return rc;
    RcclUniqueId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return NDWT_OK;
}

int ndwt_comm_create(ndwt_comm** out, const void* id128, int nranks, int rank, int device) {
    if (!out || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return cfail(NDWT_ERR_INVALID_ARG, "bad arguments");
    *out = nullptr;
    int rc = load_rccl();
    if (rc) return rc;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return cfail(NDWT_ERR_HIP, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    RcclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ndwt_comm* c = new ndwt_comm();
    c->nranks = nranks; c->rank = rank; c->device = device; c->comm = nullptr;
    const int r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
        delete c;
        return cfail(NDWT_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.ErrorString(r));
    }
    *out = c;
    return NDWT_OK;
}

int ndwt_comm_destroy(ndwt_comm* c) {
    if (!c) return NDWT_OK;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return NDWT_OK;
}

// One grouped exchange in stream order on `stream`: op i sends (is_send[i] != 0) or receives bytes[i] bytes at ptrs[i] to / from rank peers[i].
// Sends and receives between a pair of ranks are matched in the order they appear (every rank lists the segments in the same order).
int ndwt_comm_exchange(ndwt_comm* c, int nops, const int* is_send, void* const* ptrs, const int64_t* bytes, const int* peers, void* stream) {
    if (!c || nops < 0 || (nops > 0 && (!is_send || !ptrs || !bytes || !peers))) return cfail(NDWT_ERR_INVALID_ARG, "bad arguments");
    if (nops == 0) return NDWT_OK;
    for (int i = 0; i < nops; ++i)
        if (!ptrs[i] || bytes[i] < 0 || peers[i] < 0 || peers[i] >= c->nranks) return cfail(NDWT_ERR_INVALID_ARG, "op %d: bad pointer / size / peer", i);
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return cfail(NDWT_ERR_HIP, "hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
    RCCL_TRY(g_rccl.GroupStart());
    for (int i = 0; i < nops; ++i) {
        const int r = is_send[i] ? g_rccl.Send(ptrs[i], (size_t)bytes[i], 0 /* ncclInt8 */, peers[i], c->comm, (hipStream_t)stream)
                                 : g_rccl.Recv(ptrs[i], (size_t)bytes[i], 0, peers[i], c->comm, (hipStream_t)stream);
        if (r != 0) {
            (void)g_rccl.GroupEnd();
            return cfail(NDWT_ERR_HIP, "%s of op %d failed: %s", is_send[i] ? "ncclSend" : "ncclRecv", i, g_rccl.ErrorString(r));
        }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return NDWT_OK;
}

const char* ndwt_comm_last_error(void) { return g_cerr.c_str(); }

}  // extern "C"
