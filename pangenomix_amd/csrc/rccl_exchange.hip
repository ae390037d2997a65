// libpgx: the record-sharded mode's exchange done by the library itself -- ncclAllGather (RCCL over xGMI) enqueued on the
// window's stream, no host language in the loop, graph-capturable like the kernels around it. Optional: the callback of
// pgx_cluster_params.exchange (e.g. torch.distributed) does the same job. RCCL is loaded with dlopen from the path the
// caller names (PyTorch-ROCm ships its own librccl.so next to the HIP runtime the process already uses), so libpgx has no
// link-time dependency on it; the header is used for its types only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "pgx_internal.h"

namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

template <typename F>
bool bind(F &fn, const char *name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    return fn != nullptr;
}
const char *err_text(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"; }
static_assert(sizeof(ncclUniqueId) == 128, "pgx.h hands the id around as 128 bytes");
}  // namespace

extern "C" {

int pgx_rccl_load(const char *path) {
    if (g_rccl.handle) return PGX_OK;
    const char *p = path && *path ? path : "librccl.so";
    void *h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
    if (!h) { pgx_set_error("pgx_rccl_load: %s", dlerror()); return PGX_ERR_INVALID; }
    g_rccl.handle = h;
    if (!(bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
          bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.AllGather, "ncclAllGather") &&
          bind(g_rccl.GetErrorString, "ncclGetErrorString"))) {
        pgx_set_error("pgx_rccl_load: %s does not export the collective API", p);
        g_rccl = RcclApi{};
        return PGX_ERR_INVALID;
    }
    return PGX_OK;
}

int pgx_rccl_unique_id(uint8_t *out128) {
    PGX_REQUIRE(out128, "NULL argument");
    PGX_REQUIRE(g_rccl.handle, "RCCL is not loaded (pgx_rccl_load)");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) { pgx_set_error("ncclGetUniqueId: %s", err_text(r)); return PGX_ERR_INTERNAL; }
    memcpy(out128, &id, 128);
    return PGX_OK;
}

int pgx_rccl_comm_create(pgx_ctx *ctx, const uint8_t *id128, int rank, int world) {
    PGX_REQUIRE(ctx && id128, "NULL argument");
    PGX_REQUIRE(g_rccl.handle, "RCCL is not loaded (pgx_rccl_load)");
    PGX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank must be in [0, world)");
    PGX_REQUIRE(!ctx->comm, "the context has a communicator already");
    PGX_HIP(hipSetDevice(ctx->device_id));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) { pgx_set_error("ncclCommInitRank: %s", err_text(r)); return PGX_ERR_INTERNAL; }
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return PGX_OK;
}

int pgx_rccl_comm_destroy(pgx_ctx *ctx) {
    if (!ctx || !ctx->comm) return PGX_OK;
    (void)hipSetDevice(ctx->device_id);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_world = 0;
    return PGX_OK;
}

}  // extern "C"

// (internal) all-gather of `count` uint64 per process on `stream`
int pgx_rccl_all_gather_u64(pgx_ctx *ctx, const void *send, void *recv, size_t count, hipStream_t stream) {
    PGX_REQUIRE(ctx && ctx->comm && g_rccl.AllGather, "no communicator");
    const ncclResult_t r = g_rccl.AllGather(send, recv, count, ncclUint64, static_cast<ncclComm_t>(ctx->comm), stream);
    if (r != ncclSuccess) { pgx_set_error("ncclAllGather: %s", err_text(r)); return PGX_ERR_INTERNAL; }
    return PGX_OK;
}
