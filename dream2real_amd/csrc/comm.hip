// comm.hip — the one collective of the path: an all-gather of the per-candidate logits across the
// GPUs of a node, on RCCL over xGMI, enqueued on the context's stream.
//
// The reference is single-GPU (README.md:27) and has no collective; candidates are independent
// through render, composite and CLIP (reference combined_rendering.py:118-155, clip_scoring.py:175-183)
// and only spatially_smooth_heatmap (geometry_utils.py:252-269) needs neighbouring grid cells, so the
// shards exchange their logits ONCE and smooth afterwards.
//
// RCCL is bound at run time (dlopen of its soname): libd2r.so stays loadable where no RCCL is
// installed, and inside a PyTorch process the copy torch already mapped is the one that is used, so
// the process holds a single RCCL/HIP runtime pair.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "d2r_internal.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    const char *(*GetLastError)(ncclComm_t) = nullptr;       // optional: RCCL's own description of what went wrong
    std::string error;
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *e = dlerror();
            api.error = std::string("RCCL not found: ") + (e ? e : "dlopen failed");
            return;
        }
        auto sym = [&](const char *s) {
            void *p = dlsym(api.handle, s);
            if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + s;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        api.GetLastError = (decltype(api.GetLastError))dlsym(api.handle, "ncclGetLastError");
    });
    return api;
}

int rccl_fail(d2r_ctx *ctx, const char *what, ncclResult_t r)
{
    RcclApi &R = rccl();
    std::string msg = std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(r) : "RCCL error");
    if (R.GetLastError) {                      // the line RCCL logged for the failure, when it keeps one
        const char *last = R.GetLastError(ctx ? (ncclComm_t)ctx->comm : nullptr);
        if (last && *last) msg += std::string(" [") + last + "]";
    }
    return d2r_fail(ctx, D2R_ERR_DEVICE, msg);
}

}  // namespace

static_assert(D2R_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "d2r.h promises an id blob of RCCL's size");

extern "C" {

int d2r_comm_get_unique_id(void *id_out)
{
    if (!id_out) return d2r_fail(nullptr, D2R_ERR_INVALID, "null id buffer");
    RcclApi &R = rccl();
    if (!R.error.empty()) return d2r_fail(nullptr, D2R_ERR_UNSUPPORTED, R.error);
    ncclUniqueId id;
    ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", r);
    memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return D2R_OK;
}

int d2r_comm_init(d2r_ctx *ctx, const void *id_blob, int rank, int world)
{
    if (!ctx) return d2r_fail(nullptr, D2R_ERR_INVALID, "null ctx");
    if (world < 1 || rank < 0 || rank >= world) return d2r_fail(ctx, D2R_ERR_INVALID, "bad rank / world size");
    if (ctx->comm) return d2r_fail(ctx, D2R_ERR_INVALID, "context already has a communicator");
    // (the context's rank / world change only when the call succeeds)
    if (world == 1 && !id_blob) {                    // single GPU: the gather is a device copy, no RCCL needed
        ctx->comm_rank = 0;
        ctx->comm_world = 1;
        return D2R_OK;
    }
    if (!id_blob) return d2r_fail(ctx, D2R_ERR_INVALID, "null id blob");
    RcclApi &R = rccl();
    if (!R.error.empty()) return d2r_fail(ctx, D2R_ERR_UNSUPPORTED, R.error);
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, id_blob, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t r = R.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return D2R_OK;
}

int d2r_comm_destroy(d2r_ctx *ctx)
{
    if (!ctx) return d2r_fail(nullptr, D2R_ERR_INVALID, "null ctx");
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        RcclApi &R = rccl();
        if (R.CommDestroy) (void)R.CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
    return D2R_OK;
}

int d2r_allgather_scores(d2r_ctx *ctx, const float *local_dev, size_t n_local, float *global_dev)
{
    if (!ctx || !global_dev || (!local_dev && n_local)) return d2r_fail(ctx, D2R_ERR_INVALID, "null argument");
    if (n_local == 0) return D2R_OK;
    D2R_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_world <= 1 && !ctx->comm) {
        if (global_dev != local_dev)
            D2R_HIP(ctx, hipMemcpyAsync(global_dev, local_dev, n_local * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return D2R_OK;
    }
    if (!ctx->comm) return d2r_fail(ctx, D2R_ERR_INVALID, "d2r_comm_init has not been called on this context");
    ncclResult_t r = rccl().AllGather(local_dev, global_dev, n_local, ncclFloat32, (ncclComm_t)ctx->comm, ctx->stream);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllGather", r);
    return D2R_OK;
}

}  // extern "C"
