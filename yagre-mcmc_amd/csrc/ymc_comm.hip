/*
 * ymc_comm.hip -- the ensemble's only inter-GPU step: one process per GPU,
 * RCCL over xGMI.  Chains are independent (nothing is exchanged while
 * sampling); at the end every rank contributes the packed per-chain summaries
 * of its shard to ONE ncclAllGather.  RCCL is loaded with dlopen on first use
 * so single-GPU runs never pay for (or depend on) it.
 */
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "ymc_kernels.h"

/* types and enums only: the entry points are resolved with dlsym, so the
 * library has no link-time dependency on librccl */
#include <rccl/rccl.h>

int ymc_internal_pack_summary(ymc_run *run, uint64_t **dev_out, uint64_t *words, hipStream_t *stream, int *device);
int ymc_internal_fail(int code, const char *msg);
int ymc_internal_ctx_device(const ymc_ctx *ctx, hipStream_t *stream);
uint64_t ymc_internal_run_chains(const ymc_run *run);
void ymc_internal_ctx_retain(ymc_ctx *ctx);
void ymc_internal_ctx_release(ymc_ctx *ctx);

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return YMC_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return ymc_internal_fail(YMC_ERR_COMM, "cannot load librccl.so");
    Rccl r;
    r.handle = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))dlsym(h, "ncclCommUserRank");
    r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.CommUserRank || !r.AllGather ||
        !r.AllReduce || !r.GetErrorString)
        return ymc_internal_fail(YMC_ERR_COMM, "librccl.so lacks a required symbol");
    g_rccl = r;
    return YMC_OK;
}

int nccl_fail(const char *what, ncclResult_t rc)
{
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return ymc_internal_fail(YMC_ERR_COMM, buf);
}

int hip_fail(const char *what, hipError_t e)
{
    char buf[256];
    std::snprintf(buf, sizeof buf, "%s failed: %s", what, hipGetErrorString(e));
    return ymc_internal_fail(YMC_ERR_DEVICE, buf);
}

}  // namespace

struct ymc_comm {
    ymc_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;
    double *scratch = nullptr;   /* one double on the device for barrier / max */
    uint64_t *gathered = nullptr;
    size_t gathered_bytes = 0;
};

extern "C" {

int ymc_comm_preload(void) { return load_rccl(); }

int ymc_comm_count(const ymc_comm *comm, int *nranks, int *rank)
{
    if (!comm) return ymc_internal_fail(YMC_ERR_INVALID, "null argument");
    /* what RCCL itself says, not what the caller passed to ymc_comm_create */
    if (nranks) {
        ncclResult_t r = g_rccl.CommCount(comm->comm, nranks);
        if (r != ncclSuccess) return nccl_fail("ncclCommCount", r);
    }
    if (rank) {
        ncclResult_t r = g_rccl.CommUserRank(comm->comm, rank);
        if (r != ncclSuccess) return nccl_fail("ncclCommUserRank", r);
    }
    return YMC_OK;
}

int ymc_comm_unique_id(void *id, size_t id_bytes)
{
    if (!id || id_bytes < YMC_UNIQUE_ID_BYTES) return ymc_internal_fail(YMC_ERR_INVALID, "id buffer too small");
    int rc = load_rccl();
    if (rc != YMC_OK) return rc;
    ncclUniqueId uid;
    ncclResult_t r = g_rccl.GetUniqueId(&uid);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    std::memcpy(id, uid.internal, sizeof uid.internal);
    return YMC_OK;
}

int ymc_comm_create(ymc_ctx *ctx, const void *id, size_t id_bytes, int nranks, int rank, ymc_comm **out)
{
    if (!ctx || !id || !out || id_bytes < YMC_UNIQUE_ID_BYTES)
        return ymc_internal_fail(YMC_ERR_INVALID, "bad argument to ymc_comm_create");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return ymc_internal_fail(YMC_ERR_INVALID, "bad rank / nranks");
    int rc = load_rccl();
    if (rc != YMC_OK) return rc;
    ymc_comm *c = new (std::nothrow) ymc_comm;
    if (!c) return ymc_internal_fail(YMC_ERR_DEVICE, "out of host memory");
    c->nranks = nranks;
    c->rank = rank;
    c->device = ymc_internal_ctx_device(ctx, &c->stream);
    hipError_t e = hipSetDevice(c->device);
    if (e == hipSuccess) e = hipMalloc((void **)&c->scratch, sizeof(double));
    if (e != hipSuccess) {
        delete c;
        return hip_fail("communicator setup", e);
    }
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, sizeof uid.internal);
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        (void)hipFree(c->scratch);
        delete c;
        return nccl_fail("ncclCommInitRank", r);
    }
    c->ctx = ctx;
    ymc_internal_ctx_retain(ctx); /* the communicator works on the context's stream */
    *out = c;
    return YMC_OK;
}

int ymc_comm_destroy(ymc_comm *comm)
{
    if (!comm) return YMC_OK;
    (void)hipSetDevice(comm->device);
    (void)hipStreamSynchronize(comm->stream);
    if (comm->comm) (void)g_rccl.CommDestroy(comm->comm);
    if (comm->scratch) (void)hipFree(comm->scratch);
    if (comm->gathered) (void)hipFree(comm->gathered);
    ymc_ctx *ctx = comm->ctx;
    delete comm;
    if (ctx) ymc_internal_ctx_release(ctx);
    return YMC_OK;
}

int ymc_comm_allreduce_max(ymc_comm *comm, double *value)
{
    if (!comm || !value) return ymc_internal_fail(YMC_ERR_INVALID, "null argument");
    hipError_t e = hipSetDevice(comm->device);
    if (e == hipSuccess) e = hipMemcpyAsync(comm->scratch, value, sizeof(double), hipMemcpyHostToDevice, comm->stream);
    if (e != hipSuccess) return hip_fail("allreduce staging", e);
    ncclResult_t r = g_rccl.AllReduce(comm->scratch, comm->scratch, 1, ncclFloat64, ncclMax, comm->comm, comm->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    e = hipMemcpyAsync(value, comm->scratch, sizeof(double), hipMemcpyDeviceToHost, comm->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(comm->stream);
    if (e != hipSuccess) return hip_fail("allreduce readback", e);
    return YMC_OK;
}

int ymc_comm_barrier(ymc_comm *comm)
{
    double v = 0.0;
    return ymc_comm_allreduce_max(comm, &v);
}

int ymc_comm_allgather_summary(ymc_comm *comm, ymc_run *run, uint64_t *dst, size_t dst_bytes)
{
    if (!comm || !run || !dst) return ymc_internal_fail(YMC_ERR_INVALID, "null argument");
    uint64_t *packed = nullptr, words = 0;
    hipStream_t stream = nullptr;
    int device = 0;
    int rc = ymc_internal_pack_summary(run, &packed, &words, &stream, &device);
    if (rc != YMC_OK) return rc;
    if (device != comm->device) return ymc_internal_fail(YMC_ERR_INVALID, "run and communicator are on different devices");
    const uint64_t count = ymc_internal_run_chains(run) * words;
    const size_t total = (size_t)count * comm->nranks * sizeof(uint64_t);
    if (dst_bytes < total) return ymc_internal_fail(YMC_ERR_INVALID, "destination too small for the gathered summaries");
    hipError_t e = hipSuccess;
    if (comm->gathered_bytes < total) {
        if (comm->gathered) (void)hipFree(comm->gathered);
        comm->gathered = nullptr;
        comm->gathered_bytes = 0;
        e = hipMalloc((void **)&comm->gathered, total);
        if (e != hipSuccess) return hip_fail("gather buffer", e);
        comm->gathered_bytes = total;
    }
    ncclResult_t r = g_rccl.AllGather(packed, comm->gathered, count, ncclUint64, comm->comm, stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
    e = hipMemcpyAsync(dst, comm->gathered, total, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail("gather readback", e);
    return YMC_OK;
}

}  // extern "C"
