// Multi-GPU plumbing of the engine without PyTorch: one process per GPU, cells block-partitioned over the ranks
// (skdownscale_amd/shard.py), no exchange during fit / predict (cells are independent: core.py:87); the only
// communication is the gather of predicted [T, C_local] fields to a root GPU, plus a barrier / max-reduction for
// timing.  RCCL over xGMI: every peer has a direct link to the root, so the gather is grouped ncclSend / ncclRecv
// (7 links in parallel into the root), not a ring collective.  The root receives every shard into its own contiguous
// [T, C_r] block (layout [rank][T][C_r]): no concatenation copy, no padding of ragged shards.
//
// librccl is opened with dlopen on first use, so the library has no link-time dependency on it (single-GPU users never
// load it) and a process that already carries an RCCL (e.g. inside torch) shares that copy.
#include <dlfcn.h>

#include <cstring>

#include "sd_internal.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[SD_COMM_ID_BYTES]; } ncclUniqueId;  // NCCL_UNIQUE_ID_BYTES == 128 (rccl.h:40-43)
enum { kNcclSuccess = 0, kNcclMax = 2, kNcclFloat64 = 8, kNcclInt8 = 0 };  // rccl.h:450, 467

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;             // optional: what the line of an N > 1 run reports about its transport
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

Rccl g_rccl;

int load_rccl() {
    if (g_rccl.handle) return SD_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return sd_set_error(SD_ERR_UNSUPPORTED, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define SD_SYM(field, name)                                                                          \
    do {                                                                                             \
        *reinterpret_cast<void**>(&g_rccl.field) = dlsym(h, name);                                   \
        if (!g_rccl.field) return sd_set_error(SD_ERR_UNSUPPORTED, "RCCL symbol %s not found", name); \
    } while (0)
    SD_SYM(GetUniqueId, "ncclGetUniqueId");
    SD_SYM(CommInitRank, "ncclCommInitRank");
    SD_SYM(CommDestroy, "ncclCommDestroy");
    SD_SYM(Send, "ncclSend");
    SD_SYM(Recv, "ncclRecv");
    SD_SYM(AllReduce, "ncclAllReduce");
    SD_SYM(GroupStart, "ncclGroupStart");
    SD_SYM(GroupEnd, "ncclGroupEnd");
    SD_SYM(GetErrorString, "ncclGetErrorString");
#undef SD_SYM
    *reinterpret_cast<void**>(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");
    *reinterpret_cast<void**>(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
    *reinterpret_cast<void**>(&g_rccl.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
    g_rccl.handle = h;
    return SD_OK;
}

#define SD_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        const int _r = (expr);                                                                                     \
        if (_r != kNcclSuccess)                                                                                    \
            return sd_set_error(SD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
    } while (0)

}  // namespace

struct sd_comm {
    sd_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;  // communication stream (separate from the context's compute stream)
    hipEvent_t ready = nullptr;    // compute -> communication ordering
    double* scalar = nullptr;      // device scratch for reductions
};

extern "C" {

int sd_comm_unique_id(char* id /* [SD_COMM_ID_BYTES] */) {
    SD_CHECK_ARG(id, "sd_comm_unique_id: NULL argument");
    SD_TRY(load_rccl());
    ncclUniqueId u;
    SD_NCCL(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, SD_COMM_ID_BYTES);
    return SD_OK;
}

int sd_comm_create(sd_ctx* ctx, const char* id, int rank, int world, sd_comm** out) {
    SD_CHECK_ARG(ctx && id && out, "sd_comm_create: NULL argument");
    SD_CHECK_ARG(world >= 1 && rank >= 0 && rank < world, "sd_comm_create: rank %d of %d", rank, world);
    *out = nullptr;
    SD_TRY(load_rccl());
    SD_HIP(hipSetDevice(ctx->device));
    sd_comm* c = new sd_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    auto body = [&]() -> int {
        ncclUniqueId u;
        memcpy(u.internal, id, SD_COMM_ID_BYTES);
        SD_NCCL(g_rccl.CommInitRank(&c->comm, world, u, rank));
        SD_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SD_HIP(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        SD_HIP(hipMalloc((void**)&c->scalar, 2 * sizeof(double)));
        return SD_OK;
    };
    const int rc = body();
    if (rc != SD_OK) {
        sd_comm_destroy(c);
        return rc;
    }
    *out = c;
    return SD_OK;
}

int sd_comm_destroy(sd_comm* c) {
    if (!c) return SD_OK;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    if (c->scalar) (void)hipFree(c->scalar);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return SD_OK;
}

int sd_comm_info(const sd_comm* c, int* rank, int* world) {
    SD_CHECK_ARG(c, "sd_comm_info: NULL argument");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return SD_OK;
}

// What RCCL itself says about the communicator: its version code, the number of ranks IT counts (ncclCommCount -- not the
// world the caller passed in) and the device it is bound to; -1 where the loaded library lacks the query.
int sd_comm_rccl_info(const sd_comm* c, int* version, int* ranks, int* device) {
    SD_CHECK_ARG(c && c->comm, "sd_comm_rccl_info: NULL argument");
    int v = -1, n = -1, d = -1;
    if (g_rccl.GetVersion) SD_NCCL(g_rccl.GetVersion(&v));
    if (g_rccl.CommCount) SD_NCCL(g_rccl.CommCount(c->comm, &n));
    if (g_rccl.CommCuDevice) SD_NCCL(g_rccl.CommCuDevice(c->comm, &d));
    if (version) *version = v;
    if (ranks) *ranks = n;
    if (device) *device = d;
    return SD_OK;
}

// max over the ranks of a host scalar (timing: the slowest rank defines a step); also a barrier
int sd_comm_allreduce_max(sd_comm* c, double value, double* result) {
    SD_CHECK_ARG(c && result, "sd_comm_allreduce_max: NULL argument");
    SD_HIP(hipSetDevice(c->ctx->device));
    SD_HIP(hipStreamSynchronize(c->ctx->stream));  // everything this rank has queued is done before it reports
    SD_HIP(hipMemcpyAsync(c->scalar, &value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    SD_NCCL(g_rccl.AllReduce(c->scalar, c->scalar + 1, 1, kNcclFloat64, kNcclMax, c->comm, c->stream));
    SD_HIP(hipMemcpyAsync(result, c->scalar + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    SD_HIP(hipStreamSynchronize(c->stream));
    return SD_OK;
}

int sd_comm_barrier(sd_comm* c) {
    double r = 0.0;
    return sd_comm_allreduce_max(c, 0.0, &r);
}

// Gather of one contiguous [T, C_r] field per rank to `root`.  cells[world] = C_r of every rank.  On the root,
// root_dev receives the blocks back to back: block r starts at sum_{q<r} T * cells[q] doubles and is the [T, cells[r]]
// field of rank r (its own block is a device-to-device copy).  The transfer is queued on the communicator's stream
// behind everything already queued on the context's compute stream; with wait != 0 the call returns when it is done,
// otherwise sd_comm_wait() does (the compute stream is free for the next chunk of cells in between).
int sd_comm_gather_field(sd_comm* c, const double* local_dev, int64_t T, const int64_t* cells, double* root_dev, int root, int wait) {
    SD_CHECK_ARG(c && local_dev && cells, "sd_comm_gather_field: NULL argument");
    SD_CHECK_ARG(root >= 0 && root < c->world && T > 0, "sd_comm_gather_field: bad root / sizes");
    SD_CHECK_ARG(c->rank != root || root_dev, "sd_comm_gather_field: the root needs a receive buffer");
    SD_HIP(hipSetDevice(c->ctx->device));
    SD_HIP(hipEventRecord(c->ready, c->ctx->stream));
    SD_HIP(hipStreamWaitEvent(c->stream, c->ready, 0));
    const size_t mine = (size_t)T * (size_t)cells[c->rank];
    if (c->rank == root) {
        size_t off = 0;
        SD_NCCL(g_rccl.GroupStart());
        int first_error = 0;  // a failing Recv must not leave the group open: close it, then report
        for (int r = 0; r < c->world; ++r) {
            const size_t n = (size_t)T * (size_t)cells[r];
            if (r != root && n > 0 && first_error == 0)
                first_error = (int)g_rccl.Recv(root_dev + off, n, kNcclFloat64, r, c->comm, c->stream);
            off += n;
        }
        const int end_error = (int)g_rccl.GroupEnd();
        if (first_error != 0 || end_error != 0)
            return sd_set_error(SD_ERR_HIP, "RCCL error %d in the gather's receive group", first_error != 0 ? first_error : end_error);
        off = 0;
        for (int r = 0; r < root; ++r) off += (size_t)T * (size_t)cells[r];
        if (mine > 0) SD_HIP(hipMemcpyAsync(root_dev + off, local_dev, mine * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    } else if (mine > 0) {
        SD_NCCL(g_rccl.Send(local_dev, mine, kNcclFloat64, root, c->comm, c->stream));
    }
    if (wait) SD_HIP(hipStreamSynchronize(c->stream));
    return SD_OK;
}

int sd_comm_wait(sd_comm* c) {
    SD_CHECK_ARG(c, "sd_comm_wait: NULL argument");
    SD_HIP(hipSetDevice(c->ctx->device));
    SD_HIP(hipStreamSynchronize(c->stream));
    return SD_OK;
}

}  // extern "C"
