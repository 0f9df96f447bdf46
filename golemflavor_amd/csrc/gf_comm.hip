// gf_comm.hip -- the multi-GPU plumbing of the C ABI: one process per GPU, RCCL over xGMI.
// Chains (grid points) are independent, so the data path needs no collective; what is exchanged is
// the packed model descriptors at start (broadcast, ~1 KB each) and the chain blocks at the end
// (all-gather) -- the role HTCondor + a shared filesystem play in the reference
// (submitter/mc_texture_dag.py:57-71, submitter/sens_dag.py:75-95).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/golemflavor_hip.h"
#include "gf_devcache.h"                // large device allocations are cached, not handed back to the driver (hipMalloc / hipFree are macros from here on)

static_assert(GF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RCCL unique id size changed");

extern "C" void gf_internal_set_error(const char* msg);   // gf_capi.hip

struct gf_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, nranks = 1, device = 0;
};

namespace {
thread_local char g_cerr[256] = "";
int comm_fail(const char* what, const char* msg)
{
    std::snprintf(g_cerr, sizeof(g_cerr), "%s: %s", what, msg);
    gf_internal_set_error(g_cerr);          // one thread-local error text for the whole library: gf_last_hip_error()
    return GF_ERR_COMM;
}
#define GF_NCCL(call)                                                        \
    do {                                                                     \
        ncclResult_t r_ = (call);                                            \
        if (r_ != ncclSuccess) return comm_fail(#call, ncclGetErrorString(r_)); \
    } while (0)
#define GF_CHIP(call)                                                        \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) return comm_fail(#call, hipGetErrorString(e_)); \
    } while (0)
}  // namespace

extern "C" {

const char* gf_comm_last_error(void) { return g_cerr; }

// "<version code> <path of the loaded librccl>": which RCCL this process actually runs on (a process that imported
// PyTorch first may have mapped torch's bundled copy under the same soname)
int gf_comm_library_info(char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return GF_ERR_INVALID_ARG;
    int ver = 0;
    (void)ncclGetVersion(&ver);
    char path[512] = "?";
    if (FILE* f = std::fopen("/proc/self/maps", "r")) {
        char line[1024];
        while (std::fgets(line, sizeof(line), f)) {
            const char* p = std::strstr(line, "librccl");
            if (!p) continue;
            const char* q = std::strchr(line, '/');
            if (q) { std::snprintf(path, sizeof(path), "%s", q); path[std::strcspn(path, "\n")] = 0; }
            break;
        }
        std::fclose(f);
    }
    std::snprintf(buf, buflen, "%d %s", ver, path);
    return GF_OK;
}

int gf_comm_unique_id(uint8_t id[GF_COMM_ID_BYTES])
{
    if (!id) return GF_ERR_INVALID_ARG;
    ncclUniqueId uid;
    GF_NCCL(ncclGetUniqueId(&uid));
    std::memcpy(id, &uid, GF_COMM_ID_BYTES);
    return GF_OK;
}

int gf_comm_create(const uint8_t id[GF_COMM_ID_BYTES], int rank, int nranks, int device, gf_comm** out)
{
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return GF_ERR_INVALID_ARG;
    *out = nullptr;
    gf_comm* c = new (std::nothrow) gf_comm();
    if (!c) return GF_ERR_ALLOC;
    c->rank = rank; c->nranks = nranks; c->device = device;
    ncclUniqueId uid;
    std::memcpy(&uid, id, GF_COMM_ID_BYTES);
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return comm_fail("gf_comm_create", hipGetErrorString(e)); }
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        (void)hipStreamDestroy(c->stream);
        delete c;
        return comm_fail("ncclCommInitRank", ncclGetErrorString(r));
    }
    *out = c;
    return GF_OK;
}

void gf_comm_destroy(gf_comm* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int gf_comm_broadcast(gf_comm* c, void* host_buf, size_t bytes, int root)
{
    if (!c || !host_buf || root < 0 || root >= c->nranks) return GF_ERR_INVALID_ARG;
    if (bytes == 0) return GF_OK;
    GF_CHIP(hipSetDevice(c->device));
    void* d = nullptr;
    GF_CHIP(hipMalloc(&d, bytes));
    int rc = GF_OK;
    hipError_t e = hipSuccess;
    if (c->rank == root) e = hipMemcpyAsync(d, host_buf, bytes, hipMemcpyHostToDevice, c->stream);
    // every rank enters the collective whatever its local copy did: a root that skipped it would leave the others
    // blocked inside ncclBroadcast; the copy error is reported after the exchange
    const hipError_t e_copy = e;
    ncclResult_t r = ncclBroadcast(d, d, bytes, ncclChar, root, c->comm, c->stream);
    e = hipSuccess;
    if (r == ncclSuccess) e = hipMemcpyAsync(host_buf, d, bytes, hipMemcpyDeviceToHost, c->stream);
    if (r == ncclSuccess && e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = e_copy;
    if (r != ncclSuccess) rc = comm_fail("ncclBroadcast", ncclGetErrorString(r));
    else if (e != hipSuccess) rc = comm_fail("gf_comm_broadcast", hipGetErrorString(e));
    (void)hipFree(d);
    return rc;
}

int gf_comm_allgather(gf_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank)
{
    if (!c || !d_send || !d_recv) return GF_ERR_INVALID_ARG;
    if (bytes_per_rank == 0) return GF_OK;
    GF_CHIP(hipSetDevice(c->device));
    GF_NCCL(ncclAllGather(d_send, d_recv, bytes_per_rank, ncclChar, c->comm, c->stream));
    GF_CHIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

// Gather to ONE rank: every rank's block lands in d_recv_on_root[rank * bytes_per_rank ...] on `root`; nobody else allocates
// or receives `world x` the block.  This is what the reference's N jobs writing N files to one place amount to
// (golemflavor/mcmc.py:108-126); an all-gather would put every chain of the scan on every GPU for nothing.  RCCL has no
// ncclGather: one group of point-to-point operations -- the root posts world - 1 receives, every other rank one send, and
// the root's own block is a device-to-device copy on the same stream.  xGMI is point to point, so the world - 1 transfers
// run on world - 1 different links into the root.
int gf_comm_gather(gf_comm* c, const void* d_send, void* d_recv_on_root, size_t bytes_per_rank, int root)
{
    if (!c || !d_send || root < 0 || root >= c->nranks) return GF_ERR_INVALID_ARG;
    if (c->rank == root && !d_recv_on_root) return GF_ERR_INVALID_ARG;
    if (bytes_per_rank == 0) return GF_OK;
    GF_CHIP(hipSetDevice(c->device));
    if (c->rank == root) {
        char* dst = static_cast<char*>(d_recv_on_root);
        hipError_t e = hipSuccess;
        if (dst + (size_t)root * bytes_per_rank != d_send)
            e = hipMemcpyAsync(dst + (size_t)root * bytes_per_rank, d_send, bytes_per_rank, hipMemcpyDeviceToDevice, c->stream);
        // the receives are posted whatever the local copy did: the peers are already inside their sends
        ncclResult_t r = ncclSuccess;
        if (c->nranks > 1) {
            r = ncclGroupStart();
            for (int p = 0; p < c->nranks && r == ncclSuccess; ++p)
                if (p != root) r = ncclRecv(dst + (size_t)p * bytes_per_rank, bytes_per_rank, ncclChar, p, c->comm, c->stream);
            const ncclResult_t r2 = ncclGroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (r != ncclSuccess) return comm_fail("ncclRecv (gather)", ncclGetErrorString(r));
        if (e != hipSuccess) return comm_fail("gf_comm_gather", hipGetErrorString(e));
    } else {
        GF_NCCL(ncclSend(d_send, bytes_per_rank, ncclChar, root, c->comm, c->stream));
    }
    GF_CHIP(hipStreamSynchronize(c->stream));
    return GF_OK;
}

int gf_comm_barrier(gf_comm* c)
{
    if (!c) return GF_ERR_INVALID_ARG;
    GF_CHIP(hipSetDevice(c->device));
    int* d = nullptr;
    GF_CHIP(hipMalloc((void**)&d, sizeof(int)));
    hipError_t e = hipMemsetAsync(d, 0, sizeof(int), c->stream);
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = ncclAllReduce(d, d, 1, ncclInt, ncclSum, c->comm, c->stream);
    if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (r != ncclSuccess) return comm_fail("ncclAllReduce", ncclGetErrorString(r));
    if (e != hipSuccess) return comm_fail("gf_comm_barrier", hipGetErrorString(e));
    return GF_OK;
}

int gf_comm_info(gf_comm* c, int* nranks, int* rank, int* device)
{
    if (!c || !c->comm) return GF_ERR_INVALID_ARG;
    int n = -1, r = -1, d = -1;
    GF_NCCL(ncclCommCount(c->comm, &n));
    GF_NCCL(ncclCommUserRank(c->comm, &r));
    GF_NCCL(ncclCommCuDevice(c->comm, &d));
    if (nranks) *nranks = n;
    if (rank) *rank = r;
    if (device) *device = d;
    return GF_OK;
}

int gf_device_malloc(int device, size_t bytes, void** dptr)
{
    if (!dptr || bytes == 0) return GF_ERR_INVALID_ARG;
    *dptr = nullptr;
    GF_CHIP(hipSetDevice(device));
    GF_CHIP(hipMalloc(dptr, bytes));
    return GF_OK;
}

int gf_device_release(int device, void* dptr)
{
    if (!dptr) return GF_OK;
    GF_CHIP(hipSetDevice(device));
    GF_CHIP(hipFree(dptr));
    return GF_OK;
}

// ---- the same gather without RCCL: device memory shared between the processes of ONE node (hipIpc) -----------------------------
// Fallback for a node where the RCCL communicator cannot be set up (and the only inter-process device path that can be
// exercised on a one-GPU box: RCCL refuses two ranks on one device, hipIpc does not).  Every rank exports a handle of its
// block (gf_ipc_export: 64 bytes, shipped over the control plane); the root opens the handles and copies the blocks into its
// receive buffer device to device -- over xGMI when the ranks sit on different GPUs (peer access is enabled lazily by the
// open) -- then closes them.  The senders keep their blocks alive until the root reports back (the caller's barrier).
int gf_ipc_export(const void* d_ptr, unsigned char* handle64)
{
    if (!d_ptr || !handle64) return GF_ERR_INVALID_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == GF_IPC_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    GF_CHIP(hipIpcGetMemHandle(&h, const_cast<void*>(d_ptr)));
    std::memcpy(handle64, &h, sizeof(h));
    return GF_OK;
}

int gf_ipc_gather(int device, const unsigned char* handles, int nranks, int self_rank, const void* d_own, void* d_recv,
                  size_t bytes_per_rank)
{
    if (!handles || nranks < 1 || self_rank < 0 || self_rank >= nranks || !d_own || !d_recv) return GF_ERR_INVALID_ARG;
    if (bytes_per_rank == 0) return GF_OK;
    GF_CHIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    GF_CHIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char* dst = static_cast<char*>(d_recv);
    void* opened[64] = {};
    hipError_t e = hipSuccess;
    if (nranks > 64) e = hipErrorInvalidValue;
    for (int r = 0; r < nranks && e == hipSuccess; ++r) {
        if (r == self_rank) {
            if (dst + (size_t)r * bytes_per_rank != d_own)
                e = hipMemcpyAsync(dst + (size_t)r * bytes_per_rank, d_own, bytes_per_rank, hipMemcpyDeviceToDevice, st);
            continue;
        }
        hipIpcMemHandle_t h;
        std::memcpy(&h, handles + (size_t)r * GF_IPC_HANDLE_BYTES, sizeof(h));
        e = hipIpcOpenMemHandle(&opened[r], h, hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipMemcpyAsync(dst + (size_t)r * bytes_per_rank, opened[r], bytes_per_rank, hipMemcpyDefault, st);
    }
    const hipError_t e2 = hipStreamSynchronize(st);
    for (int r = 0; r < nranks && r < 64; ++r)
        if (opened[r]) (void)hipIpcCloseMemHandle(opened[r]);
    (void)hipStreamDestroy(st);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return comm_fail("gf_ipc_gather", hipGetErrorString(e));
    return GF_OK;
}

}  // extern "C"
