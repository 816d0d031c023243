// Native RCCL communicator: flat-buffer collectives over xGMI on the caller's HIP stream.
//
// The reference moves everything between processes as pickles through Ray's object store
// (SURVEY 2.4/2.5).  Here each rank (one process per MI355X) owns one ncclComm_t; every collective
// works in place on a flat device buffer and is enqueued on the stream it is given -- the trainer's
// compute stream, or a side stream for the overlapped periodic average -- so it is ordered with the
// kernels around it without host synchronisation and can be captured into a hipGraph.
// Bootstrap: rank 0 creates the 128-byte unique id, the Python side ships it through the
// torch.distributed store (or a file), every rank calls ddl_comm_init.
//
// Messages here are small (1-4 MB flat gradient buffers, [10000,10] vote matrices, scalars): they are
// latency-bound on the 7 point-to-point xGMI links, so the design is "one collective per network per
// step", never per tensor.  Plain C ABI for ctypes.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <string>

#define COMM_API extern "C" __attribute__((visibility("default")))

namespace {
thread_local std::string g_err;

struct Comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    float* scratch = nullptr;       // 1 float for barrier()
};

bool ok(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return true;
    g_err = std::string(what) + ": " + ncclGetErrorString(r);
    return false;
}

bool dtype_of(int code, ncclDataType_t* dt) {
    switch (code) {
        case 0: *dt = ncclFloat32; return true;
        case 1: *dt = ncclBfloat16; return true;
        case 2: *dt = ncclInt32; return true;
        case 3: *dt = ncclInt64; return true;
        case 4: *dt = ncclUint8; return true;
        default: g_err = "unsupported dtype code"; return false;
    }
}

bool op_of(int code, ncclRedOp_t* op) {
    switch (code) {
        case 0: *op = ncclSum; return true;
        case 1: *op = ncclMax; return true;
        case 2: *op = ncclMin; return true;
        case 3: *op = ncclAvg; return true;
        default: g_err = "unsupported reduce op"; return false;
    }
}
}  // namespace

COMM_API const char* ddl_comm_last_error() { return g_err.c_str(); }

COMM_API int ddl_comm_unique_id_bytes() { return (int)sizeof(ncclUniqueId); }

COMM_API int ddl_comm_unique_id(void* out) {
    ncclUniqueId id;
    if (!ok(ncclGetUniqueId(&id), "ncclGetUniqueId")) return -1;
    memcpy(out, &id, sizeof(id));
    return 0;
}

COMM_API void* ddl_comm_init(const void* unique_id, int world, int rank, int device) {
    if (hipSetDevice(device) != hipSuccess) { g_err = "hipSetDevice failed"; return nullptr; }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    auto* c = new Comm();
    c->world = world; c->rank = rank; c->device = device;
    if (!ok(ncclCommInitRank(&c->comm, world, id, rank), "ncclCommInitRank")) { delete c; return nullptr; }
    if (hipMalloc((void**)&c->scratch, sizeof(float)) != hipSuccess) { g_err = "hipMalloc failed"; ncclCommDestroy(c->comm); delete c; return nullptr; }
    hipMemset(c->scratch, 0, sizeof(float));
    return c;
}

COMM_API int ddl_comm_world(void* h) { return ((Comm*)h)->world; }
COMM_API int ddl_comm_rank(void* h) { return ((Comm*)h)->rank; }

COMM_API int ddl_comm_all_reduce(void* h, void* buf, uint64_t count, int dtype, int op, hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt; ncclRedOp_t ro;
    if (!dtype_of(dtype, &dt) || !op_of(op, &ro)) return -1;
    return ok(ncclAllReduce(buf, buf, count, dt, ro, c->comm, s), "ncclAllReduce") ? 0 : -1;
}

COMM_API int ddl_comm_broadcast(void* h, void* buf, uint64_t count, int dtype, int root, hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt;
    if (!dtype_of(dtype, &dt)) return -1;
    return ok(ncclBroadcast(buf, buf, count, dt, root, c->comm, s), "ncclBroadcast") ? 0 : -1;
}

// send: world * recv_count elements; recv: this rank's reduced shard (may alias send + rank * recv_count)
COMM_API int ddl_comm_reduce_scatter(void* h, const void* send, void* recv, uint64_t recv_count, int dtype, int op,
                                     hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt; ncclRedOp_t ro;
    if (!dtype_of(dtype, &dt) || !op_of(op, &ro)) return -1;
    return ok(ncclReduceScatter(send, recv, recv_count, dt, ro, c->comm, s), "ncclReduceScatter") ? 0 : -1;
}

COMM_API int ddl_comm_all_gather(void* h, const void* send, void* recv, uint64_t send_count, int dtype,
                                 hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt;
    if (!dtype_of(dtype, &dt)) return -1;
    return ok(ncclAllGather(send, recv, send_count, dt, c->comm, s), "ncclAllGather") ? 0 : -1;
}

COMM_API int ddl_comm_send(void* h, const void* buf, uint64_t count, int dtype, int peer, hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt;
    if (!dtype_of(dtype, &dt)) return -1;
    return ok(ncclSend(buf, count, dt, peer, c->comm, s), "ncclSend") ? 0 : -1;
}

COMM_API int ddl_comm_recv(void* h, void* buf, uint64_t count, int dtype, int peer, hipStream_t s) {
    auto* c = (Comm*)h;
    ncclDataType_t dt;
    if (!dtype_of(dtype, &dt)) return -1;
    return ok(ncclRecv(buf, count, dt, peer, c->comm, s), "ncclRecv") ? 0 : -1;
}

// Device-side barrier: a 1-element all-reduce on the stream (no host sync; the caller synchronises the
// stream if it needs the host to wait).
COMM_API int ddl_comm_barrier(void* h, hipStream_t s) {
    auto* c = (Comm*)h;
    return ok(ncclAllReduce(c->scratch, c->scratch, 1, ncclFloat32, ncclSum, c->comm, s), "barrier") ? 0 : -1;
}

// An async error on the communicator (a peer died, a link failed): 0 = healthy.
COMM_API int ddl_comm_async_error(void* h) {
    auto* c = (Comm*)h;
    ncclResult_t r = ncclSuccess;
    if (ncclCommGetAsyncError(c->comm, &r) != ncclSuccess) return -1;
    if (r != ncclSuccess) { g_err = ncclGetErrorString(r); return 1; }
    return 0;
}

COMM_API void ddl_comm_abort(void* h) {
    auto* c = (Comm*)h;
    if (c->comm) ncclCommAbort(c->comm);
    c->comm = nullptr;
}

COMM_API void ddl_comm_destroy(void* h) {
    auto* c = (Comm*)h;
    if (c->comm) ncclCommDestroy(c->comm);
    if (c->scratch) hipFree(c->scratch);
    delete c;
}
