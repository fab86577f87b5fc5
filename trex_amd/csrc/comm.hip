// comm.hip -- the one exchange of the frame-sharded path, owned by the library (BASELINE.json north_star): every rank
// hands its fixed-size per-blob table (trexhip_export_id_table[_ex]_device) to rank 0, whose sequential matcher consumes the frames in
// order (Tracker::add, tracking/Tracker.cpp:586-587; Tracker::predicted, :237-247).  A gather to rank 0 -- grouped ncclSend / ncclRecv
// over RCCL (xGMI inside a node) on the context's stream -- not an all-gather: only rank 0 reads the tables.
// RCCL is resolved at run time (dlopen): a process that already carries an RCCL (e.g. PyTorch's) shares that instance, and a
// single-GPU user of libtrexhip needs no RCCL at all.
#include "internal.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

typedef struct { char internal[128]; } nccl_unique_id;       // ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128 (rccl.h:40-43)
typedef void* nccl_comm;
enum { NCCL_SUCCESS = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_SUM = 0 };   // ncclSuccess, ncclUint8, ncclInt32, ncclSum (rccl.h ncclResult_t / ncclDataType_t / ncclRedOp_t)

struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // an instance already in the process
        for (const char* n : names) if (!r.so) r.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.so) r.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.so) return;
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, name))
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
        SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
        SYM(GetErrorString, "ncclGetErrorString"); SYM(AllReduce, "ncclAllReduce");
#undef SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv;
    });
    return r;
}

int nccl_fail(const char* what, int rc) {
    Rccl& r = rccl();
    trexhip::set_error(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")");
    return TREXHIP_E_DEVICE;
}

}  // namespace

struct trexhip_comm {
    trexhip_ctx* ctx = nullptr;
    nccl_comm comm = nullptr;         // null when world == 1: nothing to exchange
    int rank = 0, world = 1;
};

extern "C" {

int trexhip_comm_unique_id(void* id128) {
    if (!id128) { trexhip::set_error("trexhip_comm_unique_id: null argument"); return TREXHIP_E_INVALID; }
    Rccl& r = rccl();
    if (!r.ok) { trexhip::set_error("trexhip_comm_unique_id: librccl.so could not be loaded"); return TREXHIP_E_UNSUPPORTED; }
    nccl_unique_id id;
    const int rc = r.GetUniqueId(&id);
    if (rc != NCCL_SUCCESS) return nccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id128, &id, sizeof(id));
    return TREXHIP_OK;
}

int trexhip_comm_create(trexhip_ctx* ctx, const void* id128, int32_t rank, int32_t world, trexhip_comm** out) {
    if (!ctx || !out) { trexhip::set_error("trexhip_comm_create: null argument"); return TREXHIP_E_INVALID; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { trexhip::set_error("trexhip_comm_create: rank must be in 0..world-1"); return TREXHIP_E_INVALID; }
    trexhip_comm* c = new (std::nothrow) trexhip_comm();
    if (!c) { trexhip::set_error("out of host memory"); return TREXHIP_E_NOMEM; }
    c->ctx = ctx; c->rank = rank; c->world = world;
    if (world > 1) {
        if (!id128) { delete c; trexhip::set_error("trexhip_comm_create: the unique id of rank 0 is needed when world > 1"); return TREXHIP_E_INVALID; }
        Rccl& r = rccl();
        if (!r.ok) { delete c; trexhip::set_error("trexhip_comm_create: librccl.so could not be loaded"); return TREXHIP_E_UNSUPPORTED; }
        if (hipSetDevice(ctx->p.device) != hipSuccess) { delete c; trexhip::set_error("trexhip_comm_create: hipSetDevice failed"); return TREXHIP_E_DEVICE; }
        nccl_unique_id id;
        std::memcpy(&id, id128, sizeof(id));
        const int rc = r.CommInitRank(&c->comm, world, id, rank);
        if (rc != NCCL_SUCCESS) { delete c; return nccl_fail("ncclCommInitRank", rc); }
    }
    *out = c;
    return TREXHIP_OK;
}

void trexhip_comm_destroy(trexhip_comm* c) {
    if (!c) return;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
}

int trexhip_comm_rank(trexhip_comm* c) { return c ? c->rank : -1; }
int trexhip_comm_world(trexhip_comm* c) { return c ? c->world : 0; }

int trexhip_comm_gather_device(trexhip_comm* c, const void* d_send, size_t bytes, void* d_recv_rank0) {
    return trexhip_comm_gather_device_on(c, c ? c->ctx : nullptr, d_send, bytes, d_recv_rank0);
}

int trexhip_comm_gather_device_on(trexhip_comm* c, trexhip_ctx* ctx, const void* d_send, size_t bytes, void* d_recv_rank0) {
    if (!c || !ctx || !d_send) { trexhip::set_error("trexhip_comm_gather_device: null argument"); return TREXHIP_E_INVALID; }
    if (c->rank == 0 && !d_recv_rank0) { trexhip::set_error("trexhip_comm_gather_device: rank 0 needs the receive buffer (world x bytes)"); return TREXHIP_E_INVALID; }
    if (ctx->p.device != c->ctx->p.device) { trexhip::set_error("trexhip_comm_gather_device_on: the stream's context is on another device than the communicator"); return TREXHIP_E_INVALID; }
    TH_CHECK_HIP(hipSetDevice(ctx->p.device));
    if (bytes == 0) return TREXHIP_OK;
    if (c->rank == 0 && d_recv_rank0 != d_send)
        TH_CHECK_HIP(hipMemcpyAsync(d_recv_rank0, d_send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (c->world == 1) return TREXHIP_OK;
    Rccl& r = rccl();
    int rc = r.GroupStart();
    if (rc != NCCL_SUCCESS) return nccl_fail("ncclGroupStart", rc);
    if (c->rank == 0) {
        for (int src = 1; src < c->world && rc == NCCL_SUCCESS; ++src)
            rc = r.Recv(static_cast<uint8_t*>(d_recv_rank0) + (size_t)src * bytes, bytes, NCCL_UINT8, src, c->comm, ctx->stream);
    } else {
        rc = r.Send(d_send, bytes, NCCL_UINT8, 0, c->comm, ctx->stream);
    }
    const int rc2 = r.GroupEnd();
    if (rc != NCCL_SUCCESS) return nccl_fail("ncclSend / ncclRecv", rc);
    if (rc2 != NCCL_SUCCESS) return nccl_fail("ncclGroupEnd", rc2);
    return TREXHIP_OK;
}

int trexhip_comm_count_ranks(trexhip_comm* c, int32_t* ranks_seen) {
    if (!c || !ranks_seen) { trexhip::set_error("trexhip_comm_count_ranks: null argument"); return TREXHIP_E_INVALID; }
    *ranks_seen = 0;
    if (c->world == 1) { *ranks_seen = 1; return TREXHIP_OK; }
    Rccl& r = rccl();
    if (!r.AllReduce) { trexhip::set_error("trexhip_comm_count_ranks: ncclAllReduce not found in librccl.so"); return TREXHIP_E_UNSUPPORTED; }
    TH_CHECK_HIP(hipSetDevice(c->ctx->p.device));
    int32_t* d = nullptr;
    TH_CHECK_HIP(hipMalloc(&d, sizeof(int32_t)));
    const int32_t one = 1;
    hipError_t e = hipMemcpyAsync(d, &one, sizeof(one), hipMemcpyHostToDevice, c->ctx->stream);
    int rc = NCCL_SUCCESS;
    if (e == hipSuccess) rc = r.AllReduce(d, d, 1, NCCL_INT32, NCCL_SUM, c->comm, c->ctx->stream);
    int32_t got = 0;
    if (e == hipSuccess && rc == NCCL_SUCCESS) e = hipMemcpyAsync(&got, d, sizeof(got), hipMemcpyDeviceToHost, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    (void)hipFree(d);
    if (rc != NCCL_SUCCESS) return nccl_fail("ncclAllReduce", rc);
    TH_CHECK_HIP(e);
    *ranks_seen = got;
    return TREXHIP_OK;
}

}  // extern "C"
